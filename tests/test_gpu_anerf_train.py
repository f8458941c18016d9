"""GPU tests of A-NeRF (nerf_type = nerf) on the library's own kernels END TO END (round 6): the per-ray view constants of the render
path (danbo_anerf_view_consts_fwd replaces torch.bmm), the building blocks of the training step, `danbo_anerf_train_step` (one C
call, HIP graph) against the reference's own autograd (tests/golden/anerf_train.npz) and against the float64 arbiter
(oracle/torch_f64_anerf_train.py, itself pinned to the reference on CPU), the autograd path on the same kernels, and that NO library
GEMM runs in either (torch profiler: no rocBLAS / Tensile / at::native gemm kernel)."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(x, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(x), dtype=dtype, device=DEV)


def N(t):
    return t.detach().cpu().numpy()


def build(g, extra=(), perturb="0", noise="0"):
    from core.config import parse_args
    from core.raycasters import create_raycaster
    from core.trainer import Trainer
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import SMPLSkeleton
    args = parse_args(["--no_reload", "--N_samples", str(int(g["N_samples"])), "--N_importance", str(int(g["N_importance"])),
                       "--perturb", perturb, "--raw_noise_std", noise, *extra],
                      config=os.path.join(ROOT, "danbo-pytorch_amd", "configs", "h36m_zju", "anerf_base.txt"))
    n_codes = int(g["n_framecodes"])
    da = dict(skel_type=SMPLSkeleton, near=0., far=100., n_views=n_codes, rest_pose=syn.rest_pose(0.48), hwf=(64, 64, 80.))
    tr_kw, te_kw, start, grad_vars, opt, _ = create_raycaster(args, da, device=DEV)
    caster = tr_kw["ray_caster"]
    cfg = syn.model_config("anerf_base")
    sd = syn.make_state_dict(cfg, int(g["weight_seed"]), n_codes, syn.rest_pose(0.48))
    caster.network.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    return args, caster, Trainer(args, da, opt, None, tr_kw, te_kw, device=DEV), opt, cfg, sd


def batch_of(g):
    pose = g["pose_of_ray"]
    rb = g["ray_batch"]
    return dict(rays_o=T(rb[:, 0:3]), rays_d=T(rb[:, 3:6]), target_s=T(g["target"]), bgs=T(g["bgs"]),
                kp3d=T(g["kps"][pose]), skts=T(g["skts"][pose]), bones=T(g["bones"][pose]), cyls=T(g["cyls"][pose]),
                cam_idxs=T(g["cam_idx"], torch.int64), N_uniques=int(g["n_uniques"]))


def engine_step(trainer, batch, S, Sf, **kw):
    eng = trainer.fused_engine()
    assert eng is not None, trainer.fused_reason
    G = int(batch["N_uniques"])
    pp = lambda x: x[::x.shape[0] // G]  # noqa: E731
    return eng, eng.forward_backward(batch["rays_o"], batch["rays_d"], pp(batch["skts"]), pp(batch["bones"]), pp(batch["cyls"]), batch["cam_idxs"],
                                     batch["target_s"], batch["bgs"], S, Sf, **kw)


# ------------------------------------------------------------------------------------------------------------- building blocks
@pytest.mark.parametrize("Lv,VW", [(4, 224), (2, 64), (5, 256)])
def test_view_consts_kernel_and_its_adjoint_match_float64(Lv, VW):
    """danbo_anerf_view_consts_fwd / _bwd (the compile-time 27-term form of multires_views = 4 and the generic one) against a float64
    evaluation of the same products on danbo_anerf_view_pe_fwd's encodings: C[j, ray] = wj[j]^T E[ray, j]"""
    from core import hip_ops as ops
    gen = torch.Generator(device="cpu").manual_seed(5)
    R, G, W = 200, 4, 32
    nb = 1 + 2 * Lv
    rays_d = torch.randn(R, 3, generator=gen).to(DEV)
    skts = torch.randn(G, 24, 4, 4, generator=gen).to(DEV)
    views_w = (torch.randn(VW, W + 72 * nb + 16, generator=gen) * 0.1).to(DEV)
    wj = ops.anerf_view_wj(views_w, W, Lv)
    ref_wj = views_w[:, W:W + 72 * nb].reshape(VW, nb, 24, 3).permute(2, 1, 3, 0).reshape(24, nb * 3, VW)
    assert torch.equal(wj, ref_wj.contiguous())
    C = ops.anerf_view_consts(rays_d, skts, Lv, wj)
    E = ops.anerf_view_pe(rays_d, skts, Lv)                                    # [R, nb * 72], block-major
    Ej = E.reshape(R, nb, 24, 3).permute(2, 0, 1, 3).reshape(24, R, nb * 3).double()
    want = torch.einsum("jrk,jkc->jrc", Ej, ref_wj.double())
    scale = float(torch.einsum("jrk,jkc->jrc", Ej.abs(), ref_wj.double().abs()).max())
    assert float((C.double() - want).abs().max()) <= 2e-7 * scale
    dC = torch.randn(24, R, VW, generator=gen).to(DEV)
    g = torch.full_like(views_w, 7.0)
    ops.anerf_view_consts_bwd(rays_d, skts, Lv, dC, g, W)
    want_g = torch.einsum("jrk,jrc->jkc", Ej, dC.double()).reshape(24, nb, 3, VW).permute(3, 1, 0, 2).reshape(VW, nb * 72)
    gs = float(torch.einsum("jrk,jrc->jkc", Ej.abs(), dC.double().abs()).max())
    assert float((g[:, W:W + 72 * nb].double() - want_g).abs().max()) <= 3e-7 * gs
    assert bool((g[:, :W] == 7.0).all()) and bool((g[:, W + 72 * nb:] == 7.0).all())      # only the view columns are written
    g2 = torch.full_like(views_w, 7.0)
    ops.anerf_view_consts_bwd(rays_d, skts, Lv, dC, g2, W)
    assert torch.equal(g, g2)                                                  # fixed summation order


@pytest.mark.parametrize("S,VW,codes", [(48, 224, True), (16, 224, False), (32, 64, True), (16, 240, True)])
def test_head_layer_with_colour_epilogue_equals_head_layer_plus_colour_kernel(S, VW, codes):
    """danbo_linear16_fwd_color (round 6: A-NeRF's colour head as the EPILOGUE of its head layer -- the cutoff-weighted sum over the
    joints as one more k-step of the GEMM, table row as a 25th joint, ReLU, rgb_linear, raw stored) against the two launches it
    replaces (danbo_linear16_fwd_frag -> rows, danbo_anerf_color_fwd) and against float64; chunks that start at ray0 > 0, a partial
    last 128-row tile, rays without / with frame codes (the mean code for a negative index)"""
    from core import hip_ops as ops
    gen = torch.Generator(device="cpu").manual_seed(3)
    W, R_total, ray0, nrays = 448, 50, 7, 37
    n = nrays * S
    rnd = lambda *sh, sc=1.0: (torch.randn(*sh, generator=gen) * sc).to(DEV)  # noqa: E731
    h_rows = torch.relu(rnd(n, W))
    head_w, head_b = rnd(VW + 1, W, sc=0.05), rnd(VW + 1, sc=0.1)
    w = torch.rand(n, 24, generator=gen).to(DEV)
    C = rnd(24, R_total, VW, sc=0.3)
    n_codes = 5
    table = rnd(n_codes + 1, VW, sc=0.3)
    cam = torch.tensor([(i % 7) - 1 for i in range(R_total)], dtype=torch.int64, device=DEV) if codes else None    # -1: the mean row; 5: clamped
    rgb_w, rgb_b = rnd(3, VW, sc=0.2), rnd(3, sc=0.1)
    packed, shape = ops.linear16_pack(head_w, frag_in=(True, False))
    h = ops.FragBuffer.from_rows(h_rows)
    raw = torch.full((R_total, S, 4), 7.0, device=DEV)
    ops.linear16_color(h, packed, shape, head_b, w, C, table, cam, ray0, S, rgb_w, rgb_b, raw)
    # the two launches it replaces
    head = ops.linear16(h, packed, shape, head_b)
    raw2 = torch.full((R_total, S, 4), 7.0, device=DEV)
    ops.anerf_color(head[:, :VW], w, C, table, cam, ray0, nrays, S, rgb_w, rgb_b, head[:, VW], raw2)
    assert bool((raw[:ray0] == 7.0).all()) and bool((raw[ray0 + nrays:] == 7.0).all())         # only the chunk's rays are written
    # float64
    code = torch.full((R_total,), n_codes, dtype=torch.int64, device=DEV) if cam is None else torch.where(cam < 0, torch.full_like(cam, n_codes), cam.clamp(max=n_codes - 1))
    ray = ray0 + torch.arange(n, device=DEV) // S
    pre = h_rows.double() @ head_w.double().t() + head_b.double()
    Cr = C.double()[:, ray].permute(1, 0, 2)                                       # [n, 24, VW]
    x = torch.relu(pre[:, :VW] + table.double()[code[ray]] + torch.einsum("nj,njc->nc", w.double(), Cr))
    want = torch.cat([x @ rgb_w.double().t() + rgb_b.double(), pre[:, VW:]], 1).reshape(nrays, S, 4)
    scale = float(want.abs().max())
    got, old = raw[ray0:ray0 + nrays].double(), raw2[ray0:ray0 + nrays].double()
    assert float((got - want).abs().max()) <= 3e-6 * scale, float((got - want).abs().max()) / scale
    assert float((got - old).abs().max()) <= 3e-6 * scale
    assert torch.equal(got[..., 3], old[..., 3])                                   # the density logit: the same GEMM, the same bits


def test_small_matmul_reads_strided_operands_and_accumulates_in_float64():
    from core import hip_ops as ops
    gen = torch.Generator(device="cpu").manual_seed(9)
    a = torch.randn(37, 300, generator=gen).to(DEV)
    b = torch.randn(129, 300, generator=gen).to(DEV)
    bias = torch.randn(129, generator=gen).to(DEV)
    got = ops.small_matmul(a[:, 20:276], b[:, 20:276].t(), bias=bias)
    want = (a[:, 20:276].double() @ b[:, 20:276].double().t() + bias.double()).float()
    assert torch.allclose(got, want, rtol=2e-7, atol=1e-7)                     # float64 accumulation, one rounding


def test_relu_mask_recentres_by_powers_of_two():
    from core import _hip
    lib = _hip.lib()
    gen = torch.Generator(device="cpu").manual_seed(11)
    n = 1000 * 448
    t = (torch.randn(n, generator=gen) * 3e-7).to(DEV)
    y = torch.randn(n, generator=gen).to(DEV)
    prev_max = torch.tensor([3.1e-5], device=DEV)              # the previous layer sat at 3e-5: rho = 2^(6 + 15) = 2^21 -> [64, 128)
    sig_in = torch.tensor([4.0], device=DEV)
    sig_out, mx, dz = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV), torch.empty(n, device=DEV)
    P = lambda x: None if x is None else x.data_ptr()  # noqa: E731
    _hip.check(lib.danbo_anerf_relu_mask(P(t), P(y), n, P(prev_max), P(sig_in), P(sig_out), P(mx), P(dz), None), "relu_mask")
    torch.cuda.synchronize()
    rho = 2.0 ** (6 - np.floor(np.log2(3.1e-5)))
    assert float(sig_out) == 4.0 * rho
    want = torch.where(y > 0, t * rho, torch.zeros_like(t))
    assert torch.equal(dz, want) and float(mx) == float(want.abs().max())
    _hip.check(lib.danbo_anerf_relu_mask(P(t), P(y), n, None, P(sig_in), P(sig_out), P(mx), P(dz), None), "relu_mask")
    torch.cuda.synchronize()
    assert float(sig_out) == 4.0 and torch.equal(dz, torch.where(y > 0, t, torch.zeros_like(t)))


# ------------------------------------------------------------------------------------------------------------- the fused step
def _f64_reference(g, cfg, sd, eng, R, G, S, Sf, args, noise=None):
    """the float64 arbiter at the depths / merge order the step under test used (danbo_anerf_train_workspace_view)"""
    import torch_f64_anerf_train as f64
    from core.utils import synthetic as syn
    v = eng.workspace_view(R, G, S, Sf)
    torch.cuda.synchronize()
    base = eng._ws.data_ptr()
    at = lambda ptr, shape, dt: eng._ws[ptr - base:ptr - base + 4 * int(np.prod(shape))].view(dt).view(shape).cpu().numpy()  # noqa: E731
    z_c, z_f, order = at(v.z_coarse, (R, S), torch.float32), at(v.z_fine, (R, Sf), torch.float32), at(v.order, (R, S + Sf), torch.int32)
    assert v.bits_coarse is None and v.bits_fine is None
    batch = dict(rays_o=g["ray_batch"][:, 0:3], rays_d=g["ray_batch"][:, 3:6], skts=g["skts"], cam_idx=g["cam_idx"], target=g["target"],
                 bgs=g["bgs"])
    a = dict(loss_fn=args.loss_fn, use_background=bool(args.use_background), rgb_loss_coef=float(args.rgb_loss_coef),
             coarse_weight=float(args.coarse_weight), density_scale=float(args.density_scale), tau=20.0)
    kw = {} if noise is None else dict(noise_c=noise[0], noise_f=noise[1])
    return f64.step_bracketed(cfg, sd, syn.rest_pose(0.48), batch, z_c, z_f, order, a, **kw), (z_c, z_f, order)


@pytest.mark.parametrize("loss_fn", ["L1", "MSE"])
def test_fused_anerf_step_matches_the_reference_and_float64(loss_fn):
    """danbo_anerf_train_step on the reference's own training fixture (A-H36M, 96 rays = 4 poses x 24, 12 + 6 samples, perturb = 0,
    noise = 0): loss terms and every gradient norm / stored gradient against the reference's autograd (L1, the fixture's loss), and
    every gradient of every parameter against the float64 arbiter at the step's own depths (both losses) -- bounds = the ReLU-kink
    bracket + 2e-4 of the tensor's max (round 5's autograd path with library GEMMs was held to 2e-2)."""
    g = golden("anerf_train")
    args, caster, trainer, opt, cfg, sd = build(g, extra=("--loss_fn", loss_fn))
    batch = batch_of(g)
    S, Sf, R, G = int(g["N_samples"]), int(g["N_importance"]), 96, 4
    eng, out = engine_step(trainer, batch, S, Sf)
    torch.cuda.synchronize()
    assert type(eng).__name__ == "AnerfTrainEngine"
    grads = {n: N(p.grad).astype(np.float64) for n, p in caster.network.named_parameters() if p.requires_grad}
    loss = N(out["loss"])
    ref64, (z_c, z_f, order) = _f64_reference(g, cfg, sd, eng, R, G, S, Sf, args)
    assert np.all(np.sort(order, -1) == np.arange(S + Sf)[None]) and np.all(np.diff(np.take_along_axis(np.concatenate([z_c, z_f], 1), order, 1)) >= 0)
    for i, k in enumerate(("rgb_loss", "rgb_loss0")):
        assert abs(loss[i] - ref64["loss"][k]) <= 2e-5 * abs(ref64["loss"][k]), (k, loss[i], ref64["loss"][k])
    assert loss[2] == 0 and loss[3] == 0
    assert np.abs(N(out["rgb_map"]) - ref64["rgb_map"]).max() < 2e-5 and np.abs(N(out["rgb0"]) - ref64["rgb0"]).max() < 2e-5
    worst = 0.0
    for n, r in ref64["grads"].items():
        if not np.abs(r).max() > 0:
            assert not np.abs(grads[n]).max() > 0, n
            continue
        e = np.abs(grads[n] - r).max()
        worst = max(worst, (e - ref64["bracket"][n]) / np.abs(r).max())
        assert e <= 2e-4 * np.abs(r).max() + ref64["bracket"][n], (n, e, np.abs(r).max(), ref64["bracket"][n])
    print("fused A-NeRF step vs float64 (%s): worst (error - bracket) / max = %.2e; ambiguous units %d of %d" % (loss_fn, worst, *ref64["ambiguous"]))
    if loss_fn != "L1":
        return
    # ... and the reference's own numbers (its depths differ from ours in the last bits: importance depths are chaotic in round-off)
    for i, k in enumerate(("rgb_loss", "rgb_loss0")):
        ref = float(g["loss/" + k])
        assert abs(loss[i] - ref) <= 1e-4 * abs(ref), (k, loss[i], ref)
    assert np.abs(N(out["rgb_map"]) - g["rgb_map"]).max() < 1e-4
    for key in g.files:
        if key.startswith("gnorm/"):
            n = key[len("gnorm/"):]
            ours, ref = float(np.sqrt((grads[n] ** 2).sum())), float(g[key])
            assert abs(ours - ref) <= 2e-3 * ref + ref64["bracket"][n] * np.sqrt(grads[n].size), (n, ours, ref)
        if key.startswith("grad/"):
            n = key[len("grad/"):]
            base, _, sl = n.partition("[")
            ours = eval("grads[base][" + sl) if sl else grads[base]
            assert np.abs(ours - g[key]).max() <= 2e-3 * np.abs(g[key]).max() + ref64["bracket"][base], (n, np.abs(ours - g[key]).max())


def test_fused_anerf_step_replays_bitwise_and_trains():
    """config 5's training settings (perturb = 1, raw_noise_std = 1): the captured HIP graph replays draw fresh numbers; with the random
    stream rewound two replays of one batch give the SAME gradient bit for bit (every parameter-gradient sum of the step has a fixed
    order); Trainer.train_batch runs the step + danbo_adam_step and the loss of a fixed batch falls; tau travels as a device scalar"""
    g = golden("anerf_train")
    args, caster, trainer, opt, cfg, sd = build(g, perturb="1", noise="1")
    batch = batch_of(g)
    S, Sf = int(g["N_samples"]), int(g["N_importance"])
    eng, out = engine_step(trainer, batch, S, Sf, perturb=1.0, raw_noise_std=1.0)
    eng, out = engine_step(trainer, batch, S, Sf, perturb=1.0, raw_noise_std=1.0)      # steady state: the graph exists
    assert eng.graph is not None and eng.outputs_static
    state = eng._rng_state.clone()
    eng, out = engine_step(trainer, batch, S, Sf, perturb=1.0, raw_noise_std=1.0)
    g1, rgb1 = eng.flat_g.clone(), out["rgb_map"].clone()
    state2 = eng._rng_state.clone()
    assert int(state2[1]) > int(state[1])                                               # the kernel advanced the counter
    eng, out = engine_step(trainer, batch, S, Sf, perturb=1.0, raw_noise_std=1.0)
    assert not torch.equal(out["rgb_map"], rgb1)                                        # fresh draws
    eng._rng_state.copy_(state)
    eng, out = engine_step(trainer, batch, S, Sf, perturb=1.0, raw_noise_std=1.0)
    assert torch.equal(out["rgb_map"], rgb1) and torch.equal(eng.flat_g, g1)
    assert bool(torch.isfinite(eng.flat_g).all()) and float(eng.flat_g.abs().max()) > 0
    # a tau the graph did not see at capture time
    caster.network.pe_fn.tau = torch.tensor(200.0, device=DEV)
    caster.network.dirs_pe_fn.tau = torch.tensor(200.0, device=DEV)
    eng._rng_state.copy_(state)
    eng, out = engine_step(trainer, batch, S, Sf, perturb=1.0, raw_noise_std=1.0)
    assert not torch.equal(out["rgb_map"], rgb1)
    caster.network.pe_fn.tau = torch.tensor(20.0, device=DEV)
    caster.network.dirs_pe_fn.tau = torch.tensor(20.0, device=DEV)
    losses = []
    for i in range(40):
        loss, stats = trainer.train_batch(batch, i=i, global_step=i, sync_stats=(i % 13 == 0 or i == 39))
        if "total_loss" in stats:
            losses.append(stats["total_loss"])
            assert set(loss) == {"rgb_loss", "rgb_loss0", "total_loss"} and np.isfinite(stats["psnr"])
    assert losses[-1] < 0.8 * losses[0], losses
    # the optimizer state is torch.optim.Adam's (checkpoint layout): views of the engine's flat moments
    st = opt.state[caster.network.pts_linears[3].weight]
    assert float(st["step"]) == 40 and st["exp_avg"].data_ptr() >= eng.flat_m.data_ptr()


def test_autograd_path_runs_on_the_same_kernels_and_agrees_with_the_fused_step():
    """`caster.train(); caster(...)` for A-NeRF (core/train_path.forward_train_anerf: Linear16Fn on stacked / sliced weights,
    AnerfViewConstsFn, AnerfColorFn) against the fused step on the same batch: predictions to 1e-5, every gradient to 2e-4 of its max
    (two orderings of the same fp32 arithmetic + different depth round-off)"""
    g = golden("anerf_train")
    args, caster, trainer, opt, cfg, sd = build(g)
    batch = batch_of(g)
    S, Sf = int(g["N_samples"]), int(g["N_importance"])
    caster.train()
    kw = {k: v for k, v in trainer.render_kwargs_train.items() if k not in ("ray_caster", "use_viewdirs")}
    preds = caster(trainer._ray_batch(batch), kp_batch=batch["kp3d"], skts=batch["skts"], cyls=batch["cyls"], bones=batch["bones"],
                   cams=batch["cam_idxs"], N_uniques=batch["N_uniques"], **kw)
    assert preds["rgb_map"].grad_fn is not None and "confd" not in preds
    loss = trainer.compute_loss(batch, preds)
    caster.zero_grad()
    loss["total_loss"].backward()
    auto = {n: N(p.grad).astype(np.float64) for n, p in caster.network.named_parameters() if p.grad is not None}
    auto_rgb, auto_loss = N(preds["rgb_map"]), float(loss["total_loss"].detach())
    eng, out = engine_step(trainer, batch, S, Sf)
    torch.cuda.synchronize()
    assert np.abs(N(out["rgb_map"]) - auto_rgb).max() < 1e-5
    assert abs(float(out["loss"][0] + out["loss"][1]) - auto_loss) <= 2e-5 * auto_loss
    names = [n for n, p in caster.network.named_parameters() if p.requires_grad]
    assert set(auto) == set(names)
    for n, p in caster.network.named_parameters():
        if p.requires_grad:
            f = N(p.grad).astype(np.float64)
            assert np.abs(f - auto[n]).max() <= 2e-4 * max(np.abs(auto[n]).max(), 1e-30), (n, np.abs(f - auto[n]).max(), np.abs(auto[n]).max())


def _gemm_kernels(prof):
    bad = []
    for e in prof.events():
        name = e.name
        if any(s in name for s in ("Cijk_", "gemm", "Gemm", "GEMM", "rocblas", "hipblas", "at::native::(anonymous namespace)::gemv")):
            bad.append(name)
    return bad


def test_no_library_gemm_in_an_anerf_training_step_or_frame():
    """VERDICT r5 item 1, 'done' criterion: no Tensile / rocBLAS / at::native GEMM kernel in an A-NeRF training step (fused AND
    autograd path) nor in an A-NeRF render -- checked with torch's profiler on the kernels that actually ran"""
    from torch.profiler import ProfilerActivity, profile
    g = golden("anerf_train")
    args, caster, trainer, opt, cfg, sd = build(g)
    batch = batch_of(g)
    kw = {k: v for k, v in trainer.render_kwargs_train.items() if k not in ("ray_caster", "use_viewdirs")}
    trainer.train_batch(batch, i=0, global_step=0)                  # warm-up: graph capture, lazy initialisation
    caster.train()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        trainer.engine.use_graph = False                             # eager: the profiler sees every kernel by name
        trainer.train_batch(batch, i=1, global_step=1)
        preds = caster(trainer._ray_batch(batch), kp_batch=batch["kp3d"], skts=batch["skts"], cyls=batch["cyls"], bones=batch["bones"],
                       cams=batch["cam_idxs"], N_uniques=batch["N_uniques"], **kw)
        trainer.compute_loss(batch, preds)["total_loss"].backward()
        caster.eval()
        with torch.no_grad():
            te = {k: v for k, v in trainer.render_kwargs_test.items() if k not in ("ray_caster", "use_viewdirs")}
            caster(trainer._ray_batch(batch), kp_batch=batch["kp3d"], skts=batch["skts"], cyls=batch["cyls"], bones=batch["bones"],
                   cams=batch["cam_idxs"], N_uniques=batch["N_uniques"], **te)
        torch.cuda.synchronize()
    names = {e.name for e in prof.events()}
    assert any("k_linear16" in n for n in names) and any("k_dw16" in n for n in names) and any("k_anerf_color_bwd" in n for n in names), \
        "the profiler did not see the library's kernels"
    assert not _gemm_kernels(prof), sorted(set(_gemm_kernels(prof)))
