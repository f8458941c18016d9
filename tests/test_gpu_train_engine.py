"""GPU tests of the hand-written training step (csrc/k_train.hip and its building blocks):
  * the grouped weight / bias gradient kernel against torch (the fused trunk's forward and input-gradient chain: test_gpu_trunk.py),
  * the whole step -- losses and the gradient of EVERY parameter -- against the reference's own autograd on both training
    fixtures (danbo_train: D-H36M; danbo_perfcap_train: BASELINE config 4's network),
  * the fused step against this package's autograd path on the same batch, Adam against torch.optim.Adam, HIP-graph replay.
All through the C ABI (ctypes)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import golden
from test_gpu_training import batch_of, build_trainer, T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_grouped_weight_gradients_match_torch():
    """danbo_dw16 on row-major operands: a two-input layer (the skip layer's shape), a 3-wide and a 1-wide layer in ONE launch,
    gradient magnitudes ~1e-7 (power-of-two pre-scale from the recorded max), device-side row count below the capacity"""
    from core import _hip
    g = torch.Generator(device="cpu").manual_seed(0)
    M, live = 1000, 700
    pe = torch.zeros(M, 196, device=DEV)
    pe[:, :195] = torch.randn(M, 195, generator=g).to(DEV)
    y4 = torch.relu(torch.randn(M, 256, generator=g)).to(DEV)
    dz = (torch.randn(M, 256, generator=g) * 1e-7).to(DEV)
    mx_in = dz[:live].abs().max().reshape(1).clone()
    d3 = torch.zeros(M, 4, device=DEV)
    d3[:, :3] = (torch.randn(M, 3, generator=g) * 1e-6).to(DEV)
    hv = torch.relu(torch.randn(M, 128, generator=g)).to(DEV)
    d1 = torch.zeros(M, 4, device=DEV)
    d1[:, 0] = (torch.randn(M, generator=g) * 1e-6).to(DEV)
    gw5, gb5 = torch.full((256, 451), 7., device=DEV), torch.full((256,), 7., device=DEV)
    gw3, gb3 = torch.full((3, 128), 7., device=DEV), torch.full((3,), 7., device=DEV)
    gw1, gb1 = torch.full((1, 256), 7., device=DEV), torch.full((1,), 7., device=DEV)
    mx3 = d3.abs().max().reshape(1).clone()
    L = (_hip.DanboDwLayer * 3)(
        _hip.DanboDwLayer(dy=P(dz), x1=P(pe), x2=P(y4), dy_maxabs=P(mx_in), gw=P(gw5), gb=P(gb5), ldy=256, ld1=196, ld2=256,
                          N=256, K1=195, K2=256),
        _hip.DanboDwLayer(dy=P(d3), x1=P(hv), dy_maxabs=P(mx3), gw=P(gw3), gb=P(gb3), ldy=4, ld1=128, N=3, K1=128),
        _hip.DanboDwLayer(dy=P(d1), x1=P(y4), dy_maxabs=P(mx3), gw=P(gw1), gb=P(gb1), ldy=4, ld1=256, N=1, K1=256))
    slices = 5
    scratch = torch.empty(_hip.lib().danbo_dw16_scratch_floats(L, 3, slices), device=DEV)
    n_live = torch.tensor([live - 13], dtype=torch.int32, device=DEV)       # device-side row count below the capacity
    _hip.check(_hip.lib().danbo_dw16(L, 3, live, P(n_live), slices, P(scratch), stream()), "dw16")
    rows = slice(0, live - 13)
    for gw, gb, dy, x in ((gw5, gb5, dz[rows], torch.cat([pe[rows, :195], y4[rows]], 1)), (gw3, gb3, d3[rows, :3], hv[rows]),
                          (gw1, gb1, d1[rows, :1], y4[rows])):
        ref_w = (dy.double().t() @ x.double()).float()
        ref_b = dy.double().sum(0).float()
        assert float((gw - ref_w).abs().max()) <= 2e-5 * float(ref_w.abs().max()), (tuple(gw.shape), float((gw - ref_w).abs().max()), float(ref_w.abs().max()))
        assert float((gb - ref_b).abs().max()) <= 1e-5 * float(ref_b.abs().max()) + 1e-12


# Gradient bounds against the reference's own autograd = 2x the worst deviation measured on the three training fixtures (MI355X,
# round 3; profiles/r03_parity_measured.txt): norms 5.6e-5 -> 2e-4; stored tensors (entry-wise, relative to the tensor's max) see below
# each path against float64 autograd on its own depths (test_fused_step_on_degenerate_batches): 2x measured, see profiles/r04_parity_measured.txt
F64_BOUND, F64_BOUND_OVERLAP, F64_BOUND_EVERY = 3e-5, 3e-5, 3e-5
NORM_TOL = 2e-4
TENSOR_TOL = 2e-3        # measured 9.7e-4 (danbo_train), 2.1e-4, 5.3e-4


def fused_step(fixture, graph=False, edit=None, model_edit=None):
    g = golden(fixture)
    args, caster, trainer, opt = build_trainer(g)
    if model_edit is not None:
        model_edit(caster)
    eng = trainer.fused_engine()
    assert eng is not None, trainer.fused_reason
    eng.use_graph = graph
    b = batch_of(g)
    if edit is not None:
        edit(b)
    G = b["N_uniques"]
    pp = caster._per_pose
    noise = {}
    if "draw/t_rand" in g.files:       # the reference's own random draws (perturb = 1, raw_noise_std = 1): replay them
        eng.fixed_draws = {k: T(g["draw/" + k]) for k in ("t_rand", "u_rand", "noise_c", "noise_f")}
        noise = dict(perturb=1.0, raw_noise_std=1.0)
    out = eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"], G), pp(b["bones"], G), pp(b["cyls"], G), b["cam_idxs"],
                               b["target_s"], b["bgs"], int(g["N_samples"]), int(g["N_importance"]), **noise)
    torch.cuda.synchronize()
    return g, args, caster, trainer, eng, out


@pytest.mark.parametrize("fixture", ["danbo_train", "danbo_perfcap_train", "danbo_perfcap_train_noise"])
def test_fused_step_matches_reference_autograd(fixture):
    """danbo_perfcap_train_noise: config 4's actual settings (perturb = 1, raw_noise_std = 1) on the reference's recorded draws --
    stratified depths, both density-noise tensors and the random inverse-CDF uniforms take part in the losses and gradients"""
    g, args, caster, trainer, eng, out = fused_step(fixture)
    if "alpha0" in g.files:
        assert np.abs(out["alpha0"].cpu().numpy() - g["alpha0"]).max() < 1e-4
    R, St = out["rgb_map"].shape[0], out["alpha"].shape[1]
    assert np.abs(out["rgb_map"].cpu().numpy() - g["rgb_map"]).max() < 5e-4
    assert np.abs(out["rgb0"].cpu().numpy() - g["rgb0"]).max() < 5e-5
    ls = out["loss"].cpu().numpy().astype(np.float64)
    ours = {"rgb_loss": ls[0], "rgb_loss0": ls[1], "soft_softmax_loss": ls[2] * args.soft_softmax_loss_coef / (R * St),
            "vol_scale_loss": ls[3]}
    ours["total_loss"] = sum(ours.values())
    for k, v in ours.items():
        ref = float(g["loss/" + k])
        assert abs(v - ref) <= 2e-4 * max(abs(ref), 1e-3), (k, v, ref)
    grads = {n: p.grad.detach().cpu().numpy() for n, p in caster.network.named_parameters()}
    worst = 0.0
    for key in g.files:
        if key.startswith("gnorm/"):
            n = key[len("gnorm/"):]
            o, ref = float(np.sqrt((grads[n].astype(np.float64) ** 2).sum())), float(g[key])
            worst = max(worst, abs(o - ref) / (ref + 1e-12))
            assert abs(o - ref) <= NORM_TOL * ref + 1e-9, (n, o, ref)
    worst_t = 0.0
    for key in g.files:
        if not key.startswith("grad/"):
            continue
        n = key[len("grad/"):]
        if "[" in n:
            base, sl = n.split("[", 1)
            o = eval("grads[base][" + sl)
        else:
            o = grads[n]
        ref = g[key]
        scale = np.abs(ref).max() + 1e-12
        worst_t = max(worst_t, float(np.abs(o - ref).max() / scale))
        assert np.abs(o - ref).max() <= TENSOR_TOL * scale, (n, np.abs(o - ref).max(), scale)
    print(f"{fixture}: worst gradient-norm deviation {worst:.2e} (bound {NORM_TOL:g}), worst stored-gradient deviation {worst_t:.2e} of "
          f"the tensor's max (bound {TENSOR_TOL:g})")


def test_fused_step_equals_autograd_path_on_every_parameter():
    """same batch through core/train_path.py (torch autograd + rocBLAS) and through danbo_train_step"""
    g = golden("danbo_perfcap_train")
    args, caster, trainer, opt = build_trainer(g)
    caster.train()
    b = batch_of(g)
    kw = {k: v for k, v in trainer.render_kwargs_train.items() if k not in ("ray_caster", "use_viewdirs")}
    preds = caster(trainer._ray_batch(b), kp_batch=b["kp3d"], skts=b["skts"], cyls=b["cyls"], bones=b["bones"], cams=b["cam_idxs"],
                   N_uniques=b["N_uniques"], **kw)
    loss = trainer.compute_loss(b, preds)
    caster.zero_grad()
    loss["total_loss"].backward()
    ref = {n: p.grad.detach().clone() for n, p in caster.network.named_parameters()}
    _, _, caster2, _, eng, out = fused_step("danbo_perfcap_train")
    for n, p in caster2.network.named_parameters():
        a, r = p.grad, ref[n]
        assert float((a - r).abs().max()) <= 2e-3 * float(r.abs().max()) + 1e-10, (n, float((a - r).abs().max()), float(r.abs().max()))
    assert float((out["rgb_map"] - preds["rgb_map"]).abs().max()) < 1e-4


def test_adam_kernel_matches_torch_adam_and_graph_replay_matches_eager():
    g = golden("danbo_train")
    args, caster, trainer, opt = build_trainer(g)
    eng = trainer.fused_engine()
    eng.use_graph = False
    b = batch_of(g)
    G = b["N_uniques"]
    pp = caster._per_pose
    run = lambda: eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"], G), pp(b["bones"], G), pp(b["cyls"], G),  # noqa: E731
                                       b["cam_idxs"], b["target_s"], b["bgs"], int(g["N_samples"]), int(g["N_importance"]))
    run()
    # a torch.optim.Adam on copies of the parameters, fed the same gradients for three steps
    names = list(eng.params)
    shadow = [torch.nn.Parameter(eng.params[n].detach().clone()) for n in names]
    ref_opt = torch.optim.Adam(shadow, lr=args.lrate, betas=(0.9, 0.999))
    for step in range(3):
        for s, n in zip(shadow, names):
            s.grad = eng.params[n].grad.detach().clone()
        ref_opt.step()
        eng.adam_step(args.lrate)
        for s, n in zip(shadow, names):
            if eng.params[n].requires_grad:
                d = float((s.detach() - eng.params[n].detach()).abs().max())
                assert d <= 2e-7 + 1e-6 * float(s.detach().abs().max()), (step, n, d)
        run()
    assert int(float(opt.state[eng.params[names[0]]]["step"])) == 3
    # HIP-graph capture of the step: replays give the eager gradients (deterministic batch: perturb = 0, noise = 0)
    eager = eng.flat_g.clone()
    eng.use_graph = True
    run()
    run()
    torch.cuda.synchronize()
    d = float((eng.flat_g - eager).abs().max())
    assert d <= 1e-5 * float(eager.abs().max()), d      # atomics: summation order differs between runs


def test_adam_scalars_survive_a_host_that_runs_ahead_of_the_gpu():
    """danbo_adam_step takes lr and the bias corrections BY VALUE.  (ABI 1 read them from one device buffer refreshed by a
    non-blocking copy from one pinned host buffer: with the host several steps ahead -- Trainer.train_batch(sync_stats=False) -- step
    t could be applied with step t+k's corrections.)  200 unsynchronised updates behind a long-running kernel == 200 updates
    with a synchronisation after each, bit for bit, and both == torch.optim.Adam to round-off."""
    g = golden("danbo_train")
    args, caster, trainer, opt = build_trainer(g)
    eng = trainer.fused_engine()
    gen = torch.Generator(device="cpu").manual_seed(3)
    p0 = eng.flat_p.clone()
    grads = [(torch.randn(eng.flat_g.numel(), generator=gen) * 1e-3).to(DEV) for _ in range(4)]
    big = torch.randn(8192, 8192, device=DEV)

    def run(sync):
        eng.flat_p.copy_(p0)
        eng.flat_m.zero_()
        eng.flat_v.zero_()
        eng.t = 0
        for _ in range(3):
            big @ big                                   # ~1.3 TFLOP each: the GPU is busy while the host enqueues all 200 updates
        for t in range(200):
            eng.flat_g.copy_(grads[t % 4])
            eng.adam_step(args.lrate * (0.999 ** t), 0.5)
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return eng.flat_p.clone()
    ahead, stepwise = run(False), run(True)
    assert torch.equal(ahead, stepwise)
    shadow = torch.nn.Parameter(p0[:eng.n_train].clone())
    ref_opt = torch.optim.Adam([shadow], lr=args.lrate, betas=(0.9, 0.999))
    for t in range(200):
        shadow.grad = grads[t % 4][:eng.n_train] * 0.5
        ref_opt.param_groups[0]["lr"] = args.lrate * (0.999 ** t)
        ref_opt.step()
    d = float((shadow.detach() - ahead[:eng.n_train]).abs().max())
    assert d <= 5e-6, d


def test_train_batch_runs_fused_with_noise_and_updates_eval_weights():
    g = golden("danbo_perfcap_train")
    args, caster, trainer, opt = build_trainer(g, extra=["--raw_noise_std", "1.0", "--perturb", "1.0"])
    before = {n: p.detach().clone() for n, p in caster.network.named_parameters()}
    torch.manual_seed(0)
    loss, stats = trainer.train_batch(batch_of(g), i=0, global_step=0)
    assert trainer.engine is not None and np.isfinite(stats["total_loss"]) and stats["lrate"] == pytest.approx(5e-4 * 0.1 ** (1 / 500000), rel=1e-9)
    moved = [n for n, p in caster.network.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert len(moved) == 43, sorted(set(before) - set(moved))
    for i in range(1, 4):
        loss, stats = trainer.train_batch(batch_of(g), i=i, global_step=i)
        assert np.isfinite(stats["total_loss"])
    # the eval path renders with the updated weights (parameter versions are bumped by the Adam kernel's wrapper)
    caster.eval()
    kw = {k: v for k, v in trainer.render_kwargs_test.items() if k not in ("ray_caster", "use_viewdirs")}
    b = batch_of(g)
    out1 = caster(trainer._ray_batch(b), kp_batch=b["kp3d"], skts=b["skts"], cyls=b["cyls"], bones=b["bones"], cams=b["cam_idxs"],
                  N_uniques=b["N_uniques"], **kw)
    trainer.train_batch(batch_of(g), i=4, global_step=4)
    caster.eval()
    out2 = caster(trainer._ray_batch(b), kp_batch=b["kp3d"], skts=b["skts"], cyls=b["cyls"], bones=b["bones"], cams=b["cam_idxs"],
                  N_uniques=b["N_uniques"], **kw)
    assert torch.isfinite(out2["rgb_map"]).all() and not torch.equal(out1["rgb_map"], out2["rgb_map"])


def test_graph_replays_are_repeatable():
    """2 000 replays of the captured step on the same deterministic batch (round 6: 300 until then; VERDICT r5 item 6 -- the in-suite form
    of tools/stress_replay.py): identical device counters, forward outputs and (up to the order of the atomics) gradients every time.  (A hipMemsetAsync node inside the captured graph used to be replayed wrongly --
    the counters stayed non-zero and the second replay wrote out of bounds; the step now zeroes with kernels.  Round 4: about one
    replay in 200 had one ray's colour off by 1e-3 while K2 ran beside the view-constant kernel of the prologue -- see the join
    in front of K2 in csrc/k_train.hip; tools/stress_replay.py is the long form of this test.)"""
    g, args, caster, trainer, eng, out = fused_step("danbo_perfcap_train", graph=True)
    b = batch_of(g)
    G = b["N_uniques"]
    pp = caster._per_pose
    ref_counts, ref_grad = out["counts"].clone(), eng.flat_g.clone()
    ref_maps = {k: out[k].clone() for k in ("rgb_map", "rgb0", "acc_map", "alpha")}
    for _ in range(2000):
        out = eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"], G), pp(b["bones"], G), pp(b["cyls"], G), b["cam_idxs"],
                                   b["target_s"], b["bgs"], int(g["N_samples"]), int(g["N_importance"]))
        torch.cuda.synchronize()
        assert torch.equal(out["counts"], ref_counts)
        for k, v in ref_maps.items():
            assert torch.equal(out[k], v), k           # the forward pass has no atomics in its arithmetic
        assert torch.isfinite(eng.flat_g).all()
        diff = (eng.flat_g - ref_grad).abs()
        if float(diff.max()) > 1e-5 * float(ref_grad.abs().max()):          # say WHERE before failing
            offs = sorted(eng.offsets.items(), key=lambda kv: kv[1])
            for (n, o), (_, o2) in zip(offs, offs[1:] + [("end", diff.numel())]):
                if o2 > o and float(diff[o:o2].max()) > 0:
                    print("replay", _, "differs in", n, "max", float(diff[o:o2].max()), "of", float(ref_grad[o:o2].abs().max()),
                          "entries", int((diff[o:o2] > 0).sum()), "/", o2 - o)
        assert float(diff.max()) <= 1e-5 * float(ref_grad.abs().max())


def _hip_graph_dag(raw_graph):
    """(kernel name | node type per node, set of (from, to) index pairs) of a captured hipGraph_t, through the HIP runtime's graph
    introspection (hipGraphGetNodes / hipGraphGetEdges / hipGraphKernelNodeGetParams / hipKernelNameRefByPtr)"""
    rt = ctypes.CDLL("libamdhip64.so")
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    rt.hipGraphGetNodes.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    rt.hipGraphGetEdges.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz)]
    rt.hipGraphNodeGetType.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
    rt.hipKernelNameRefByPtr.argtypes = [vp, vp]
    rt.hipKernelNameRefByPtr.restype = ctypes.c_char_p

    class Dim3(ctypes.Structure):
        _fields_ = [("x", ctypes.c_uint), ("y", ctypes.c_uint), ("z", ctypes.c_uint)]

    class KernelParams(ctypes.Structure):           # hipKernelNodeParams
        _fields_ = [("blockDim", Dim3), ("extra", vp), ("func", vp), ("gridDim", Dim3), ("kernelParams", vp), ("sharedMemBytes", ctypes.c_uint)]
    rt.hipGraphKernelNodeGetParams.argtypes = [vp, ctypes.POINTER(KernelParams)]
    g = vp(raw_graph)
    n = sz(0)
    assert rt.hipGraphGetNodes(g, None, ctypes.byref(n)) == 0 and n.value > 0
    nodes = (vp * n.value)()
    assert rt.hipGraphGetNodes(g, nodes, ctypes.byref(n)) == 0
    index = {int(nodes[i]): i for i in range(n.value)}
    ne = sz(0)
    assert rt.hipGraphGetEdges(g, None, None, ctypes.byref(ne)) == 0
    fr, to = (vp * max(ne.value, 1))(), (vp * max(ne.value, 1))()
    assert rt.hipGraphGetEdges(g, fr, to, ctypes.byref(ne)) == 0
    edges = {(index[int(fr[i])], index[int(to[i])]) for i in range(ne.value)}
    names = []
    for i in range(n.value):
        t = ctypes.c_int(-1)
        assert rt.hipGraphNodeGetType(nodes[i], ctypes.byref(t)) == 0
        if t.value == 0:                               # hipGraphNodeTypeKernel
            kp = KernelParams()
            assert rt.hipGraphKernelNodeGetParams(nodes[i], ctypes.byref(kp)) == 0
            nm = rt.hipKernelNameRefByPtr(kp.func, None)
            names.append((nm.decode() if nm else "kernel?") + f"<<<{kp.gridDim.x},{kp.blockDim.x},{kp.sharedMemBytes}>>>")
        else:
            names.append(f"node type {t.value}")
    return names, edges


def test_nothing_in_the_captured_step_can_run_beside_k2():
    """the non-repeatable view constants of round 4 (one term of one ray's sum wrong when K2 ran beside
    k_train_cview) are fenced by STREAM ORDER -- the step joins its side streams in front of the first K2 (csrc/k_train.hip).  That
    the fence is structural is checked on the captured graph itself: in the DAG of the replayed hipGraph every node is an ancestor or
    a descendant of each k_assign16 node -- no launch is concurrent with K2 -- and the view-constant kernel precedes the first K2."""
    g = golden("danbo_perfcap_train")
    args, caster, trainer, opt = build_trainer(g)
    eng = trainer.fused_engine()
    eng.keep_graph = True
    b = batch_of(g)
    G = b["N_uniques"]
    pp = caster._per_pose
    for _ in range(2):
        eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"], G), pp(b["bones"], G), pp(b["cyls"], G), b["cam_idxs"], b["target_s"],
                             b["bgs"], int(g["N_samples"]), int(g["N_importance"]))
    torch.cuda.synchronize()
    names, edges = _hip_graph_dag(eng.graph[1].raw_cuda_graph())
    n = len(names)
    assert n >= 30, names
    succ = [[] for _ in range(n)]
    for a, c in edges:
        succ[a].append(c)

    def reach(src):
        seen, stack = set(), [src]
        while stack:
            for y in succ[stack.pop()]:
                if y not in seen:
                    seen.add(y)
                    stack.append(y)
        return seen
    desc = [reach(i) for i in range(n)]
    k2 = [i for i, nm in enumerate(names) if "k_assign16" in nm and "pack" not in nm]
    cview = [i for i, nm in enumerate(names) if "k_train_cview" in nm]
    assert len(k2) == 2 and len(cview) == 1, (k2, cview, names)
    for k in k2:
        beside = [names[i] for i in range(n) if i != k and i not in desc[k] and k not in desc[i]]
        assert not beside, ("nodes that may run beside K2", beside)
        assert k in desc[cview[0]]
    # ... the step does use its side streams (the graph is not a chain): some pair of nodes IS unordered
    assert any(j not in desc[i] and i not in desc[j] for i in range(n) for j in range(i + 1, n))


def _autograd_grads(fixture, edit, model_edit=None, sampling=None):
    """sampling: a dict that receives the depths / merge order the path used (z_c, z_f, order), for the float64 reference"""
    from core import hip_ops
    g = golden(fixture)
    args, caster, trainer, opt = build_trainer(g)
    if model_edit is not None:
        model_edit(caster)
    caster.train()
    b = batch_of(g)
    edit(b)
    kw = {k: v for k, v in trainer.render_kwargs_train.items() if k not in ("ray_caster", "use_viewdirs")}
    orig = hip_ops.importance_samples

    def spy(z, w, n, u=None):
        out = orig(z, w, n, u)
        if sampling is not None:
            sampling.update(z_c=z.detach().clone(), z_f=out[1].detach().clone(), order=out[2].detach().clone())
        return out
    hip_ops.importance_samples = spy
    try:
        preds = caster(trainer._ray_batch(b), kp_batch=b["kp3d"], skts=b["skts"], cyls=b["cyls"], bones=b["bones"], cams=b["cam_idxs"],
                       N_uniques=b["N_uniques"], **kw)
    finally:
        hip_ops.importance_samples = orig
    loss = trainer.compute_loss(b, preds)
    caster.zero_grad()
    loss["total_loss"].backward()
    return ({n: (torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()) for n, p in caster.network.named_parameters()},
            preds, {k: float(v.detach()) for k, v in loss.items()})


def _fused_sampling(eng, R, G, S, Sf):
    """depths and merge order the fused step left in its workspace (danbo_train_workspace_view)"""
    from core import _hip
    v = _hip.DanboTrainView()
    _hip.check(_hip.lib().danbo_train_workspace_view(ctypes.byref(eng._model()), R, G, S, Sf, R, ctypes.c_void_p(eng._ws.data_ptr()),
                                                     ctypes.byref(v)), "danbo_train_workspace_view")
    base = eng._ws.data_ptr()

    def at(ptr, n, dt):
        return eng._ws[ptr - base:ptr - base + 4 * n].view(dt)
    return dict(z_c=at(v.z_coarse, R * S, torch.float32).view(R, S).clone(), z_f=at(v.z_fine, R * Sf, torch.float32).view(R, Sf).clone(),
                order=at(v.order, R * (S + Sf), torch.int32).view(R, S + Sf).clone())


def _f64_reference(g, args, caster, b, sampling, debug=None):
    """oracle/torch_f64_train.py on the batch `b`, the model as `caster` holds it, and the depths of the path under test"""
    import torch_f64_train as t64
    from core.utils import synthetic as syn
    cfg = syn.model_config(str(g["cfg_name"]))
    sd = {k: v.detach().cpu().numpy() for k, v in caster.network.state_dict().items()}
    coef = dict(loss_fn=args.loss_fn, use_background=bool(args.use_background), rgb_loss_coef=float(args.rgb_loss_coef),
                coarse_weight=float(args.coarse_weight), soft_softmax_loss_coef=float(args.soft_softmax_loss_coef),
                vol_scale_penalty=float(args.vol_scale_penalty) if args.opt_vol_scale else 0.0)
    nb = {k: b[k].detach().cpu().numpy() for k in ("rays_o", "rays_d", "skts", "bones", "target_s", "bgs", "cam_idxs")}
    return (t64.step if debug is not None else t64.step_bracketed)(cfg, coef, sd, caster.transforms[0].cpu().numpy(), caster.network.graph_net.init_scale.cpu().numpy(), nb,
                    sampling["z_c"].cpu().numpy(), sampling["z_f"].cpu().numpy(), sampling["order"].cpu().numpy(), int(b["N_uniques"]),
                    device=DEV, clamped_c=sampling["acc0"].detach().cpu().numpy() >= 1.0, clamped_f=sampling["acc_map"].detach().cpu().numpy() >= 1.0,
                    **({} if debug is None else dict(debug=debug)))


@pytest.mark.parametrize("case", ["no_sample_in_any_volume", "one_pose_misses", "odd_ray_count", "many_small_poses", "overlapping_volumes",
                                  "every_volume"])
def test_fused_step_on_degenerate_batches(case):
    """Batches the device-side row bookkeeping has to survive: no sample inside any bone volume (zero in-volume rows: only the
    per-ray empty-space rows carry a gradient), one pose whose rays all miss while the others hit, a ray count that is no
    multiple of any tile size, 48 poses of 4 rays each (a workgroup of the K2 adjoint keeps two poses' tables in LDS: its chunks
    then span many more and take the global-memory branch for the pose transforms, volumes and volume gradients), and bone volumes
    grown 4x / 12x (axis_scale is trainable) so that a sample lies in many volumes at
    once: more (row, bone) pairs than the K2 adjoint's launch grid has workgroups for, it has to stride -- against the autograd
    path on the same batch."""
    def model_edit(caster):
        if case in ("overlapping_volumes", "every_volume"):
            with torch.no_grad():
                # every_volume: x 12 puts (nearly) every sample into (nearly) all 24 volumes: > 8 pairs per row of CAPACITY, more than
                # the K2 adjoint's launch grid covers with 256 pairs per workgroup -- its workgroups take larger chunks in sub-batches
                caster.network.graph_net.axis_scale.mul_(4.0 if case == "overlapping_volumes" else 12.0)

    def edit(b):
        if case == "no_sample_in_any_volume":
            b["rays_o"] = b["rays_o"] + torch.tensor([40.0, 0.0, 0.0], device=DEV)
        elif case == "one_pose_misses":
            per = b["rays_o"].shape[0] // b["N_uniques"]
            b["rays_o"] = b["rays_o"].clone()
            b["rays_o"][:per] += torch.tensor([40.0, 0.0, 0.0], device=DEV)
        elif case == "many_small_poses":
            b["N_uniques"] = b["rays_o"].shape[0] // 4            # every 4 consecutive rays (of the same pose) become a "pose" of their own
        elif case == "odd_ray_count":
            G = b["N_uniques"]
            per = b["rays_o"].shape[0] // G
            keep = torch.cat([torch.arange(g0 * per, g0 * per + per - 3, device=DEV) for g0 in range(G)])   # 3 rays fewer per pose
            for k in ("rays_o", "rays_d", "target_s", "bgs", "kp3d", "skts", "bones", "cyls", "cam_idxs"):
                b[k] = b[k][keep].contiguous()
    samp_a = {}
    ref, preds, ref_loss = _autograd_grads("danbo_perfcap_train", edit, model_edit, sampling=samp_a)
    g, args, caster, trainer, eng, out = fused_step("danbo_perfcap_train", edit=edit, model_edit=model_edit)
    counts = out["counts"].cpu().numpy()
    # ---- each path against float64 autograd of the SAME batch on the path's own depths (oracle/torch_f64_train.py, pinned to the
    # reference's autograd in tests/test_oracle_configs.py): which of the two is how far from the truth
    b64 = batch_of(g)
    edit(b64)
    R64, G64 = b64["rays_o"].shape[0], int(b64["N_uniques"])
    samp_f = _fused_sampling(eng, R64, G64, int(g["N_samples"]), int(g["N_importance"]))
    samp_f.update(acc0=out["acc0"], acc_map=out["acc_map"])         # which rays sat on the constant branch of min(sum w, 1)
    samp_a.update(acc0=preds["acc0"], acc_map=preds["acc_map"])
    assert torch.equal(samp_f["z_c"], samp_a["z_c"])                # perturb = 0: the coarse depths are the same function of the rays
    f64_bound = {"overlapping_volumes": F64_BOUND_OVERLAP, "every_volume": F64_BOUND_EVERY}.get(case, F64_BOUND)
    brackets = {}
    for path, grads, samp in (("autograd", ref, samp_a), ("fused", {n: p.grad for n, p in caster.network.named_parameters()}, samp_f)):
        r64 = _f64_reference(g, args, caster, b64, samp)
        brackets[path] = r64["bracket"]
        worst64, name64 = 0.0, ""
        for n, gr in grads.items():
            t = r64["grads"][n]
            scale = float(np.abs(t).max())
            d = float(np.abs(gr.detach().cpu().numpy().astype(np.float64) - t).max())
            if d / (scale + 1e-30) > worst64:
                worst64, name64 = d / (scale + 1e-30), n
            # + what the ReLU units whose sign fp32 does not determine can move this gradient by (torch_f64_train.Kinks)
            # (x 1.5: the bracket is the effect of flipping ALL ambiguous units TOGETHER -- a signed sum in which the units' effects
            # partly cancel; a path that takes a SUBSET of them on the other side can be further from float64 than the whole set
            # is.  Round 5 measured up to 1.12 x on one tensor of the overlapping-volumes batch after one ulp of the view
            # directions' norm changed which units sit on the kink.)
            assert d <= f64_bound * scale + 1.5 * r64["bracket"][n] + 1e-9, (path, n, d, scale, r64["bracket"][n])
        kn = max(r64["bracket"], key=lambda k: r64["bracket"][k] / (float(np.abs(r64["grads"][k]).max()) + 1e-30))
        print(case, path, "vs float64: worst gradient deviation (of the tensor's max)", worst64, "in", name64, "| ambiguous ReLU units",
              r64["ambiguous"], "their bracket at most", r64["bracket"][kn] / (float(np.abs(r64["grads"][kn]).max()) + 1e-30), "in", kn)
    if case in ("overlapping_volumes", "every_volume"):
        pairs = int((preds["part_invalid"] == 0).sum())
        cap = out["rgb_map"].shape[0] * (out["alpha"].shape[1] + 1)
        print("(row, bone) pairs per in-volume row:", pairs / max(int(counts[5]), 1), "per row of capacity:", pairs / cap)
        assert pairs > 4 * int(counts[5])
        if case == "every_volume":
            assert pairs > 8.5 * cap                # beyond (capacity * 4 / 128) workgroups x 256 pairs
    R = out["rgb_map"].shape[0]
    assert torch.isfinite(out["loss"]).all() and torch.isfinite(eng.flat_g).all()
    if case == "no_sample_in_any_volume":
        assert counts[2] == R and counts[3] == 0, counts          # rows of the coarse pass = the R empty-space rows, fine pass: none
    assert float((out["rgb_map"] - preds["rgb_map"].detach()).abs().max()) < (1e-4 if case != "every_volume" else 1e-3)
    assert abs(float(out["loss"][0]) - ref_loss["rgb_loss"]) <= 2e-4 * max(abs(ref_loss["rgb_loss"]), 1e-3)
    worst, worst_name = 0.0, ""
    # fused step against autograd path: NOT the same function -- each resamples from its own coarse weights (importance depths are
    # a chaotic function of fp32 round-off) and each takes its own side of every ReLU kink; the tight statement is the one above,
    # each path against float64 on its own depths.  (Dropped (row, bone) pairs would show as tens of percent.)
    for n, p in caster.network.named_parameters():
        a, r = p.grad, ref[n]
        scale = float(r.abs().max())
        d = float((a - r).abs().max())
        assert d <= 2e-3 * scale + brackets["autograd"][n] + brackets["fused"][n] + 1e-9, (n, d, scale)
        if d / (scale + 1e-30) > worst:
            worst, worst_name = d / (scale + 1e-30), n
    print(case, "rows", counts[:6], "worst relative gradient deviation", worst, "in", worst_name)


def test_weight_gradients_from_fragment_order_operands_and_split_layers():
    """danbo_dw16 with dy / x in k_linear16's fragment order (what the trunk of the training step hands it), a layer whose two
    inputs are passed as two layers writing column ranges of the same weight gradient, ragged row count and device-side count"""
    from core import _hip, hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(11)
    M = 5000 - 37
    dy = (torch.randn(M, 256, generator=g) * 3e-6).to(DEV)
    xa = torch.randn(M, 196, generator=g).to(DEV)          # 195 used columns, row-major
    xb = torch.randn(M, 256, generator=g).to(DEV)
    mx = dy.abs().max().reshape(1)
    slack = lambda f: torch.cat([f.data, torch.zeros(128 * f.C, device=DEV)])
    fdy, fxb = slack(ops.FragBuffer.from_rows(dy)), slack(ops.FragBuffer.from_rows(xb))
    gw = torch.zeros(256, 451, device=DEV)
    gb = torch.zeros(256, device=DEV)
    gw2 = torch.zeros(256, 256, device=DEV)
    gb2 = torch.zeros(256, device=DEV)
    D = _hip.DanboDwLayer
    common = dict(dy_maxabs=P(mx), gw2=None, gb2=None, ldy=256, ld2=0, N=256, K2=0, split_n=0, x2=None)
    L = (D * 3)(D(dy=P(fdy), x1=P(xa), gw=P(gw), gb=P(gb), ld1=196, K1=195, frag=1, gw_ld=451, gw_col0=0, **common),
                D(dy=P(fdy), x1=P(fxb), gw=P(gw), gb=None, ld1=256, K1=256, frag=3, gw_ld=451, gw_col0=195, **common),
                D(dy=P(dy), x1=P(fxb), gw=P(gw2), gb=P(gb2), ld1=256, K1=256, frag=2, gw_ld=0, gw_col0=0, **common))
    slices = 7
    scratch = torch.empty(_hip.lib().danbo_dw16_scratch_floats(L, 3, slices), device=DEV)
    n_live = torch.tensor([M], dtype=torch.int32, device=DEV)
    _hip.check(_hip.lib().danbo_dw16(L, 3, M + 50, P(n_live), slices, P(scratch), stream()), "dw16")
    ref = (dy.double().t() @ torch.cat([xa[:, :195], xb], 1).double()).float()
    ref_b = dy.double().sum(0).float()
    assert float((gw - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float((gw2 - ref[:, 195:]).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float((gb - ref_b).abs().max()) <= 1e-5 * float(ref_b.abs().max()) and float((gb2 - ref_b).abs().max()) <= 1e-5 * float(ref_b.abs().max())


def test_fused_step_with_more_than_64_samples_per_ray():
    """S > 64 leaves the fused composite + resampling kernel (one wavefront per ray) for the general kernels: dense raw fill, plain
    composite, importance sampling by ranks.  72 + 24 samples against the autograd path."""
    g = golden("danbo_perfcap_train")
    S, Sf = 72, 24

    def run_autograd():
        args, caster, trainer, opt = build_trainer(g)
        caster.train()
        b = batch_of(g)
        kw = {k: v for k, v in trainer.render_kwargs_train.items() if k not in ("ray_caster", "use_viewdirs", "N_samples", "N_importance")}
        preds = caster(trainer._ray_batch(b), kp_batch=b["kp3d"], skts=b["skts"], cyls=b["cyls"], bones=b["bones"], cams=b["cam_idxs"],
                       N_uniques=b["N_uniques"], N_samples=S, N_importance=Sf, **kw)
        loss = trainer.compute_loss(b, preds)
        caster.zero_grad()
        loss["total_loss"].backward()
        return {n: p.grad.detach().clone() for n, p in caster.network.named_parameters()}, preds
    ref, preds = run_autograd()
    args, caster, trainer, opt = build_trainer(g)
    eng = trainer.fused_engine()
    eng.use_graph = False
    b = batch_of(g)
    G = b["N_uniques"]
    pp = caster._per_pose
    out = eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"], G), pp(b["bones"], G), pp(b["cyls"], G), b["cam_idxs"],
                               b["target_s"], b["bgs"], S, Sf)
    assert float((out["rgb_map"] - preds["rgb_map"].detach()).abs().max()) < 1e-4
    assert float((out["rgb0"] - preds["rgb0"].detach()).abs().max()) < 1e-4
    for n, p in caster.network.named_parameters():
        a, r = p.grad, ref[n]
        assert float((a - r).abs().max()) <= 2e-3 * float(r.abs().max()) + 1e-10, (n, float((a - r).abs().max()), float(r.abs().max()))


@pytest.mark.parametrize("fixture,extra", [("danbo_train", ["--loss_fn", "MSE"]), ("danbo_perfcap_train", ["--loss_fn", "MSE", "--coarse_weight", "0.3"]),
                                           ("danbo_train", ["--rgb_loss_coef", "2.5", "--soft_softmax_loss_coef", "0.0"])])
def test_fused_step_flag_variants_equal_the_autograd_path(fixture, extra):
    """the loss options the fused step covers (train_engine.supported): MSE instead of L1, coarse / rgb weights, no soft-softmax term"""
    g = golden(fixture)
    grads = {}
    outs = {}
    for path in ("autograd", "fused"):
        args, caster, trainer, opt = build_trainer(g, extra=extra)
        b = batch_of(g)
        if path == "autograd":
            caster.train()
            kw = {k: v for k, v in trainer.render_kwargs_train.items() if k not in ("ray_caster", "use_viewdirs")}
            preds = caster(trainer._ray_batch(b), kp_batch=b["kp3d"], skts=b["skts"], cyls=b["cyls"], bones=b["bones"], cams=b["cam_idxs"],
                           N_uniques=b["N_uniques"], **kw)
            loss = trainer.compute_loss(b, preds)
            caster.zero_grad()
            loss["total_loss"].backward()
            outs[path] = {k: float(v.detach()) for k, v in loss.items()}
        else:
            eng = trainer.fused_engine()
            assert eng is not None, trainer.fused_reason
            eng.use_graph = False
            G = b["N_uniques"]
            pp = caster._per_pose
            out = eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"], G), pp(b["bones"], G), pp(b["cyls"], G), b["cam_idxs"],
                                       b["target_s"], b["bgs"], int(g["N_samples"]), int(g["N_importance"]))
            outs[path] = out["loss"].cpu().numpy()
        grads[path] = {n: (torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()) for n, p in caster.network.named_parameters()}
    assert abs(float(outs["fused"][0]) - outs["autograd"]["rgb_loss"]) <= 2e-4 * max(abs(outs["autograd"]["rgb_loss"]), 1e-3)
    assert abs(float(outs["fused"][1]) - outs["autograd"]["rgb_loss0"]) <= 2e-4 * max(abs(outs["autograd"]["rgb_loss0"]), 1e-3)
    for n, r in grads["autograd"].items():      # 5e-3 of the tensor's max, as against the reference's autograd (axis_scale: 2e-3)
        a = grads["fused"][n]
        assert float((a - r).abs().max()) <= 5e-3 * float(r.abs().max()) + 1e-10, (n, float((a - r).abs().max()), float(r.abs().max()))


def test_random_draws_are_the_philox_streams_and_the_kernel_advances_its_counter():
    """danbo_random_draws: uniforms bit for bit the restated Philox4x32-10 stream (tests/helpers.py, pinned to the Random123
    known answers on the CPU), normals its Box-Muller pairs; the device-side counter moves by the quads drawn, so the next call
    -- or the next replay of a captured graph -- continues the stream; ragged counts; moments of a large draw"""
    from core import hip_ops as ops
    from helpers import philox_stream_words
    seed = 0x1234_5678_9ABC_DEF0
    state = torch.tensor([seed, 5, 0], dtype=torch.int64, device="cuda")
    counter = 5
    for nu, nn, std in ((1003, 2050, 1.5), (4096, 0, 1.0), (0, 777, 0.25), (3, 1, 1.0)):
        u, z = ops.random_draws(state, nu, nn, std)
        quads = (max(nu, nn) + 3) // 4
        if nu:
            want = (philox_stream_words(seed, counter, quads, 0).reshape(-1)[:nu] >> 8).astype(np.float32) * np.float32(2.0 ** -24)
            assert np.array_equal(u.cpu().numpy(), want)
            assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
        if nn:
            w = philox_stream_words(seed, counter, quads, 1).reshape(-1, 2).astype(np.float64)
            u1, u2 = ((w[:, 0].astype(np.uint64) >> 8) + 1) * 2.0 ** -24, (w[:, 1].astype(np.uint64) >> 8) * 2.0 ** -24
            rad = np.sqrt(-2.0 * np.log(u1)) * std
            want = np.stack([rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)], 1).reshape(-1)[:nn]
            assert np.abs(z.cpu().numpy() - want).max() < 2e-5 * std * 6
        counter += quads
        assert state.cpu().tolist() == [seed, counter, 0]
    state = torch.tensor([7, 0, 0], dtype=torch.int64, device="cuda")
    u, z = ops.random_draws(state, 1 << 22, 1 << 22, 2.0)
    u, z = u.double(), z.double()
    assert abs(float(u.mean()) - 0.5) < 1e-3 and abs(float(u.var()) - 1 / 12) < 1e-3
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.var()) - 4.0) < 2e-2 and abs(float((z ** 4).mean()) / 16.0 - 3.0) < 5e-2
    assert len(torch.unique(u)) > (1 << 22) * 0.85            # (24-bit grid: birthday collisions only)
    u2, _ = ops.random_draws(state, 1 << 22, 0, 1.0)
    assert not torch.equal(u2.double(), u) and state.cpu().tolist() == [7, 2 * (1 << 20), 0]
    assert ops._hip.lib().danbo_random_draws(None, 4, None, 0, 1.0, None, None) == -22


def test_gather_rows_collects_strided_rows_of_several_tensors():
    """danbo_gather_rows: one launch gathers rows of any stride (the trainer's per-pose slices of per-ray tensors), 8-byte elements
    as word pairs, broadcast rows (stride 0), into one flat buffer at the given word offsets"""
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn(3072, 24, 4, 4, generator=g).cuda()
    b = torch.randn(3072, 24, 3, generator=g).cuda()
    c = torch.randn(3072, 5, generator=g).cuda()
    d = torch.randint(0, 1 << 40, (3072,), generator=g).cuda()
    e = torch.randn(1, 3, generator=g).cuda().expand(3072, 3)
    f = torch.randn(50, 7, generator=g).cuda().t()                     # inner dim not contiguous: copied by the wrapper
    items = [a[::192], b[::192], c[::192], d, e, f, a[5:6]]
    offs, o = [], 0
    for t in items:
        offs.append(o)
        o += (t.numel() * (t.element_size() // 4) + 63) // 64 * 64
    dst = torch.full((o,), float("nan"), device="cuda")
    ops.gather_rows(dst, list(zip(items, offs)))
    for t, off in zip(items, offs):
        n = t.numel() * (t.element_size() // 4)
        got = dst[off:off + n].view(t.dtype).view(t.shape)
        assert torch.equal(got, t.contiguous())
        pad = dst[off + n:off + (n + 63) // 64 * 64]
        assert bool(torch.isnan(pad).all())                             # gaps are left alone
    assert ops._hip.lib().danbo_gather_rows(None, 1, None, None) == -22


def test_captured_step_draws_fresh_numbers_every_replay_and_follows_the_seed():
    """the step's random numbers come from danbo_random_draws inside the captured graph: every replay perturbs differently, the
    same seed gives the same sequence of steps (reseed), another seed another one"""
    g = golden("danbo_perfcap_train")
    args, caster, trainer, opt = build_trainer(g, extra=["--raw_noise_std", "1.0", "--perturb", "1.0"])
    assert trainer.fused_engine() is not None
    eng = trainer.engine
    b = batch_of(g)

    def steps(n):
        seq = []
        for i in range(n):
            trainer.train_batch(b, i=i, global_step=i, sync_stats=False)
            rnd = trainer.last_preds["_keep"][0]
            seq.append((rnd["t_rand"].clone(), rnd["noise_f"].clone(), trainer.last_preds["rgb_map"].clone()))
        return seq
    torch.manual_seed(3)
    s1 = steps(4)
    assert eng.graph is not None
    for i in range(1, 4):
        assert not torch.equal(s1[i][0], s1[i - 1][0]) and not torch.equal(s1[i][1], s1[i - 1][1])
    assert float(s1[0][0].min()) >= 0 and float(s1[0][0].max()) < 1 and 0.5 < float(s1[0][1].std()) < 2.0
    eng.reseed(torch.cuda.initial_seed())
    s2 = steps(2)
    assert torch.equal(s2[0][0], s1[0][0]) and torch.equal(s2[1][1], s1[1][1])          # the same draws (the weights have moved on)
    torch.manual_seed(4)
    s3 = steps(1)
    assert not torch.equal(s3[0][0], s1[0][0])


@pytest.mark.parametrize("S,Sf,mse,use_bg", [(48, 16, 0, 1), (32, 16, 1, 0), (64, 32, 0, 0), (96, 48, 0, 1), (200, 56, 1, 1)])
def test_fused_mid_step_equals_its_four_launches_bitwise(S, Sf, mse, use_bg):
    """danbo_train_mid (ABI 7: loss gradients + both composite adjoints + un-merge in one launch, what danbo_train_step runs) against
    danbo_train_loss_grad, danbo_composite_bwd_lazy x 2 and danbo_train_draw_unmerge on the same inputs: every tensor bit for bit,
    the loss sums (atomics in both) to rounding.  Several 64-sample chunks per ray, MSE / L1, with and without backgrounds."""
    from core import _hip
    lib = _hip.lib()
    g = torch.Generator(device="cpu").manual_seed(S * 131 + Sf)
    R, St, B = 157, S + Sf, 7.5
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)          # noqa: E731
    uni = lambda *s: torch.rand(*s, generator=g).to(DEV)           # noqa: E731
    rgb, rgb0, target, bgs = uni(R, 3), uni(R, 3), uni(R, 3), uni(R, 3)
    acc, acc0 = uni(R) * 1.2, uni(R) * 1.2                          # some rays saturate: acc >= 1 switches g_acc off
    raw_c, raw_empty, raw_sorted = rnd(R, S, 4) * 3, rnd(R, 4), rnd(R, St, 4) * 3
    bits_c = (torch.rand(R, S, generator=g) < 0.6).to(torch.int32).to(DEV) * 5
    bits_f = (torch.rand(R, Sf, generator=g) < 0.6).to(torch.int32).to(DEV) * 3
    z_c = torch.sort(uni(R, S) * 4 + 1, dim=1).values.contiguous()
    z_sorted = torch.sort(uni(R, St) * 4 + 1, dim=1).values.contiguous()
    rays_d = rnd(R, 3)
    noise_c, noise_f = rnd(R, S), rnd(R, St)
    order = torch.stack([torch.randperm(St, generator=g) for _ in range(R)]).to(torch.int32).to(DEV)
    weights = (uni(R, St) * (torch.rand(R, St, generator=g) < 0.7).to(DEV)).contiguous()
    alpha = uni(R, St)
    P = lambda t: ctypes.c_void_p(t.data_ptr())                    # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def outputs():
        return dict(g_rgb=torch.zeros(R, 3, device=DEV), g_acc=torch.zeros(R, device=DEV), g_rgb0=torch.zeros(R, 3, device=DEV),
                    g_acc0=torch.zeros(R, device=DEV), d_raw_c=torch.zeros(R, S, 4, device=DEV), d_raw_f=torch.zeros(R, Sf, 4, device=DEV),
                    d_raw_rows=torch.zeros(R, 4, device=DEV), label_c=torch.zeros(R, S, dtype=torch.uint8, device=DEV),
                    label_f=torch.zeros(R, Sf, dtype=torch.uint8, device=DEV), loss=torch.zeros(8, device=DEV), maxabs=torch.zeros(4, device=DEV))

    a, b = outputs(), outputs()
    d_sorted = torch.zeros(R, St, 4, device=DEV)
    wf, wc = 1.0, 0.5
    _hip.check(lib.danbo_train_loss_grad(P(rgb), P(acc), P(rgb0), P(acc0), P(target), P(bgs), use_bg, R, mse, wf, wc, P(a["g_rgb"]), P(a["g_acc"]),
                                         P(a["g_rgb0"]), P(a["g_acc0"]), P(a["loss"]), st), "loss_grad")
    _hip.check(lib.danbo_composite_bwd_lazy(P(raw_c), P(raw_empty), P(bits_c), P(z_c), P(rays_d), R, S, B, P(noise_c), P(a["g_rgb0"]),
                                            P(a["g_acc0"]), P(a["d_raw_c"]), st), "composite_bwd coarse")
    _hip.check(lib.danbo_composite_bwd_lazy(P(raw_sorted), None, None, P(z_sorted), P(rays_d), R, St, B, P(noise_f), P(a["g_rgb"]), P(a["g_acc"]),
                                            P(d_sorted), st), "composite_bwd merged")
    _hip.check(lib.danbo_train_draw_unmerge(P(a["d_raw_c"]), P(d_sorted), P(order), P(bits_c), P(bits_f), P(weights), P(alpha), R, S, Sf,
                                            P(a["d_raw_f"]), P(a["d_raw_rows"]), P(a["label_c"]), P(a["label_f"]), P(a["loss"]), P(a["maxabs"]), st),
               "draw_unmerge")
    _hip.check(lib.danbo_train_mid(P(rgb), P(acc), P(rgb0), P(acc0), P(target), P(bgs), use_bg, R, S, Sf, mse, wf, wc, B, P(b["g_rgb"]), P(b["g_acc"]),
                                   P(b["g_rgb0"]), P(b["g_acc0"]), P(raw_c), P(raw_empty), P(raw_sorted), P(bits_c), P(bits_f), P(z_c), P(z_sorted),
                                   P(rays_d), P(noise_c), P(noise_f), P(order), P(weights), P(alpha), P(b["d_raw_c"]), P(b["d_raw_f"]),
                                   P(b["d_raw_rows"]), P(b["label_c"]), P(b["label_f"]), P(b["loss"]), P(b["maxabs"]), st), "train_mid")
    torch.cuda.synchronize()
    for k in a:
        if k == "loss":
            assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), (k, a[k], b[k])
        else:
            assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))
    assert float(a["d_raw_c"].abs().max()) > 0 and float(a["d_raw_f"].abs().max()) > 0 and int(a["label_c"].sum()) > 0
    assert float(a["loss"][2]) > 0 and float(a["maxabs"][0]) > 0
    assert lib.danbo_train_mid(*([None] * 6), 0, R, 200, 100, *([0] * 1), 1.0, 1.0, 1.0, *([None] * 25)) == -22


def test_loss_terms_are_snapshotted_at_the_first_look_and_refuse_a_late_one():
    """Trainer.train_batch(sync_stats=False) returns the loss terms lazily: the replayed graph writes them into ONE static buffer, the
    dictionary snapshots it when somebody looks (no launch otherwise) -- and a look after the next step raises instead of handing out
    that step's numbers."""
    g = golden("danbo_perfcap_train")
    args, caster, trainer, opt = build_trainer(g)
    assert trainer.fused_engine() is not None
    b = batch_of(g)
    for i in range(3):                                       # graph built, steady state
        trainer.train_batch(b, i=i, global_step=i)
    loss_a, _ = trainer.train_batch(b, i=3, global_step=3, sync_stats=False)
    seen = float(loss_a['total_loss'])                       # first look: snapshot
    loss_b, _ = trainer.train_batch(b, i=4, global_step=4, sync_stats=False)
    assert float(loss_a['total_loss']) == seen               # the snapshot, not step 4's value
    assert float(loss_b['total_loss']) != seen
    loss_c, _ = trainer.train_batch(b, i=5, global_step=5, sync_stats=False)
    trainer.train_batch(b, i=6, global_step=6, sync_stats=False)
    with pytest.raises(RuntimeError, match="before the next step"):
        loss_c['total_loss']
    with pytest.raises(RuntimeError, match="before the next step"):     # still pending, still refusing
        len(loss_c)
    _, stats = trainer.train_batch(b, i=7, global_step=7, sync_stats=True)
    assert np.isfinite(stats['total_loss']) and np.isfinite(stats['psnr'])
    # an EAGER step (no graph) returns fresh tensors: its terms stay readable after later steps, like the reference's
    trainer.engine.use_graph = False
    loss_d, _ = trainer.train_batch(b, i=8, global_step=8, sync_stats=False)
    trainer.train_batch(b, i=9, global_step=9, sync_stats=False)
    trainer.train_batch(b, i=10, global_step=10, sync_stats=False)
    assert np.isfinite(float(loss_d['total_loss'])) and len(loss_d) >= 3
