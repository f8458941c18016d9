"""Generate the golden vectors under tests/golden/ by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference; see oracle/ref_harness.py).
    python oracle/gen_golden.py            # regenerates every fixture
The reference never travels: what is committed are the inputs and the reference's outputs
(plain .npz data) plus this script.  Weights are NOT stored -- they come from the seeded
generator in danbo-pytorch_amd/core/utils/synthetic.py, which the tests call again with
the seed recorded in each fixture.

Fixtures
  danbo_stages.npz   D-H36M (danbo_base net), 48 rays = 2 poses x 24, 12+6 samples, framecode
                     idx per ray: every stage-boundary tensor of SURVEY §8(c)
  danbo_surreal.npz  D-Surr (box near/far, no frame codes), whole 64x64 frame as ONE 4096-ray
                     chunk incl. rays that miss the cylinder (NaN back-fill), 32+16: final maps
  danbo_perfcap.npz  D-Perf (relray/root_local view branch, box near/far), 256 rays, 32+16,
                     mean frame code (idx -1): raw + final maps
  danbo_train.npz    D-H36M, 128 rays = 4 poses x 32, 16+8 samples, training mode with perturb = 0 and
                     raw_noise_std = 0: the four loss terms of Trainer.compute_loss and gradients
  danbo_perfcap_train.npz  D-Perf (relray/root_local, box near/far), 192 rays = 4 poses x 48, 16+8 samples, training
                     mode (perturb = 0, noise = 0): loss terms, gradient norms and gradients of the reference's autograd
  danbo_h36m_fast.npz  BASELINE config 2: H36M danbo_fast (box near/far, 32 + 16, frame codes), 256 rays of 2 poses: bounds,
                     coarse raw, final maps
  danbo_mesh.npz     D-H36M, RayCaster.render_mesh_density at res = 16 (17^3 raw densities around the root joint)
  anerf_stages.npz   A-H36M (anerf_base net: cutoff PE, W = 448), 48 rays = 2 poses x 24, 12+6 samples, at
                     tau = 20 (step 0) with every stage tensor, and raw + final maps again at tau = 2000
  anerf_train.npz    A-H36M, 96 rays = 4 poses x 24, 12+6 samples, training mode (perturb = 0, noise = 0): loss terms and
                     gradients of the reference's autograd
  sequences.npz      bullet-time / interpolation / selected-frame camera and pose sequences of run_render.py
  render_path.npz    run_nerf.render_path end to end (valid-ray boxes, background blend), D-Surr, 4 images of 40 x 32
  pose_rot6d.npz     axis-angle -> rot6d incl. tiny angles (pytorch3d boundary, cross-checked
                     with scipy in the tests)
"""
import contextlib
import importlib.util
import io
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

spec = importlib.util.spec_from_file_location(
    "danbo_synthetic", os.path.join(ROOT, "danbo-pytorch_amd", "core", "utils", "synthetic.py"))
syn = importlib.util.module_from_spec(spec)
spec.loader.exec_module(syn)

CONFIGS = {
    "danbo_base": "configs/h36m_zju/danbo_base.txt",
    "danbo_fast": "configs/h36m_zju/danbo_fast.txt",
    "danbo_perfcap": "configs/perfcap/danbo_fast.txt",
    "danbo_surreal": "configs/surreal/danbo_fast.txt",
    "anerf_base": "configs/h36m_zju/anerf_base.txt",
}


def T(x, dtype=torch.float32):
    return torch.tensor(np.asarray(x), dtype=dtype)


def build(cfg_name, seed, n_framecodes=20):
    cfg = syn.model_config(cfg_name)
    args = rh.parse_reference_config(CONFIGS[cfg_name])
    rest = syn.rest_pose(cfg["rest_scale"])
    with contextlib.redirect_stdout(io.StringIO()):
        caster, kw_train, kw_test = rh.build_reference_caster(args, rest, n_framecodes, tempfile.mkdtemp())
    sd = syn.make_state_dict(cfg, seed=seed, n_framecodes=n_framecodes, rest=rest, lively=True)
    caster.network.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    caster.eval()
    return cfg, args, caster, kw_test, rest


def body_rays(scene, view, n, seed):
    """Pick n rays of a full-grid view whose pixels fall on the projected skeleton bbox."""
    ro, rd = scene["rays"][view]
    H, W = scene["H"], scene["W"]
    rng = np.random.default_rng(seed)
    js, is_ = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    sel = ((np.abs(is_ - W / 2) < W * 0.22) & (np.abs(js - H / 2) < H * 0.40)).reshape(-1)
    idx = rng.choice(np.nonzero(sel)[0], size=n, replace=False)
    idx.sort()
    return ro[idx], rd[idx]


def per_ray(scene, pose_of_ray):
    return (scene["kps"][pose_of_ray], scene["skts"][pose_of_ray], scene["bones"][pose_of_ray],
            scene["cyls"][pose_of_ray])


def call_caster(caster, kw_test, rb, kps, skts, cyls, bones, cams, N_samples, N_importance, n_uniques):
    kw = {k: v for k, v in kw_test.items() if k not in ("ray_caster", "use_viewdirs", "N_samples", "N_importance")}
    with torch.no_grad():
        out = caster(T(rb), N_samples=N_samples, kp_batch=T(kps), skts=T(skts), cyls=T(cyls),
                     bones=T(bones), cams=cams, N_importance=N_importance, N_uniques=n_uniques, **kw)
    return {k: v.numpy() for k, v in out.items()}


def gen_danbo_stages():
    seed = 11
    cfg, args, caster, kw_test, rest = build("danbo_base", seed)
    net = caster.network
    scene = syn.make_scene(n_poses=2, H=64, W=64, n_views=2, pose_seed=5)
    n_per = 24
    ro, rd, pose = [], [], []
    for p in range(2):
        o, d = body_rays(scene, p, n_per, seed=100 + p)
        ro.append(o); rd.append(d); pose += [p] * n_per
    ro, rd, pose = np.concatenate(ro), np.concatenate(rd), np.array(pose)
    rb = syn.ray_batch(ro, rd)
    kps, skts, bones, cyls = per_ray(scene, pose)
    cam_idx = (np.arange(len(pose)) % 7).astype(np.int64)
    S, Sf = 12, 6
    fin = call_caster(caster, kw_test, rb, kps, skts, cyls, bones, T(cam_idx, torch.long), S, Sf, 2)

    with torch.no_grad():
        rays_o, rays_d = T(ro), T(rd)
        near, far = caster.get_near_far(rays_o, rays_d, T(cyls), near=T(rb[:, 6:7]), far=T(rb[:, 7:8]), skts=T(skts))
        pts, z = caster.sample_pts(rays_o, rays_d, near, far, len(pose), S, 0., False)
        inputs = caster.get_nerf_inputs(pts, [rays_o[:, None, :], rays_d[:, None, :]], T(kps), T(skts), T(bones),
                                        cam_idxs=T(cam_idx, torch.long), N_uniques=2)
        enc0 = net.pts_embedder.encode_pts(pts, T(kps), T(skts), T(bones), inputs["align_transforms"], inputs["rest_pose"])
        w = net.pts_embedder.encode_graph_inputs(T(kps), T(bones), T(skts), 2)["w"]
        vols = net.forward_graph(net.graph_pe_fn(w)[0])
        part_feat, invalid = net.extract_graph_feat(vols, enc0["pts_t"])
        dens_in, enc = net.encode_pts(inputs)
        view_in, _ = net.encode_views(inputs, refs=enc["pts_t"], encoded_pts=enc)
        raw, _ = net(inputs)
        out0 = net.raw2outputs(raw, z, rays_d, B=1.0, act_fn=torch.relu)
        z_all, z_fine, order = __import__("core.utils.ray_utils", fromlist=["x"]).isample_from_lineseg(
            z, out0["weights"], Sf, det=True, is_only=True)
    np.savez_compressed(
        os.path.join(OUT, "danbo_stages.npz"),
        cfg_name="danbo_base", weight_seed=seed, n_framecodes=20, N_samples=S, N_importance=Sf, n_uniques=2,
        ray_batch=rb, kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], cyls=scene["cyls"],
        pose_of_ray=pose, cam_idx=cam_idx, rest_pose=rest,
        align=caster.transforms[0].numpy(), near=near.numpy(), far=far.numpy(), z_coarse=z.numpy(),
        pts=pts.numpy(), pts_t=enc0["pts_t"].numpy(), rot6d=w.numpy(), volumes=vols.numpy(),
        part_feat=part_feat.numpy(), invalid=invalid.numpy(), confd=enc["confd"].numpy(),
        agg_p=enc["agg_p"].numpy(), density_inputs=dens_in.numpy(), view_inputs=view_in.numpy()[::S].copy(),
        raw_coarse=raw.numpy(), weights_coarse=out0["weights"].numpy(), alpha_coarse=out0["alpha"].numpy(),
        rgb_coarse=out0["rgb_map"].numpy(), z_fine=z_fine.numpy(), z_sorted=z_all.numpy(),
        sorted_idxs=order.numpy(), **{"final_" + k: v for k, v in fin.items()})
    print("danbo_stages: valid frac", 1 - invalid.numpy().mean(), "acc mean", fin["acc_map"].mean())


def gen_danbo_surreal():
    seed = 12
    cfg, args, caster, kw_test, rest = build("danbo_surreal", seed, n_framecodes=4)
    scene = syn.make_scene(n_poses=1, H=64, W=64, n_views=3, pose_seed=2, rest_scale=cfg["rest_scale"], cam_dist=5.0)
    ro, rd = scene["rays"][1]
    rb = syn.ray_batch(ro, rd)
    pose = np.zeros(len(ro), dtype=np.int64)
    kps, skts, bones, cyls = per_ray(scene, pose)
    S, Sf = 32, 16
    fin = call_caster(caster, kw_test, rb, kps, skts, cyls, bones, None, S, Sf, 1)
    with torch.no_grad():
        near, far = caster.get_near_far(T(ro), T(rd), T(cyls), near=T(rb[:, 6:7]), far=T(rb[:, 7:8]), skts=T(skts))
        from core.utils.ray_utils import get_near_far_in_cylinder
        cn, cf = get_near_far_in_cylinder(T(ro), T(rd), T(cyls), near=T(rb[:, 6:7]), far=T(rb[:, 7:8]))
    np.savez_compressed(
        os.path.join(OUT, "danbo_surreal.npz"),
        cfg_name="danbo_surreal", weight_seed=seed, n_framecodes=4, N_samples=S, N_importance=Sf, n_uniques=1,
        pose_seed=2, cam_dist=5.0, view=1, H=64, W=64,
        near=near.numpy(), far=far.numpy(), cyl_near=cn.numpy(), cyl_far=cf.numpy(),
        **{"final_" + k: v for k, v in fin.items() if k in ("rgb_map", "disp_map", "acc_map", "rgb0", "acc0")})
    miss = np.isnan(np.sqrt(1.0)).sum()
    print("danbo_surreal: acc mean", fin["acc_map"].mean(), "rays", len(ro))


def gen_danbo_perfcap():
    seed = 13
    cfg, args, caster, kw_test, rest = build("danbo_perfcap", seed)
    scene = syn.make_scene(n_poses=1, H=96, W=96, n_views=4, pose_seed=9)
    ro, rd = body_rays(scene, 2, 256, seed=3)
    rb = syn.ray_batch(ro, rd)
    pose = np.zeros(len(ro), dtype=np.int64)
    kps, skts, bones, cyls = per_ray(scene, pose)
    S, Sf = 32, 16
    cams = T(-np.ones(len(ro)), torch.long)
    fin = call_caster(caster, kw_test, rb, kps, skts, cyls, bones, cams, S, Sf, 1)
    net = caster.network
    with torch.no_grad():
        rays_o, rays_d = T(ro), T(rd)
        near, far = caster.get_near_far(rays_o, rays_d, T(cyls), near=T(rb[:, 6:7]), far=T(rb[:, 7:8]), skts=T(skts))
        pts, z = caster.sample_pts(rays_o, rays_d, near, far, len(ro), S, 0., False)
        inputs = caster.get_nerf_inputs(pts, [rays_o[:, None, :], rays_d[:, None, :]], T(kps), T(skts), T(bones),
                                        cam_idxs=cams, N_uniques=1)
        raw, _ = net(inputs)
        dens_in, enc = net.encode_pts(inputs)
        view_in, _ = net.encode_views(inputs, refs=enc["pts_t"], encoded_pts=enc)
    np.savez_compressed(
        os.path.join(OUT, "danbo_perfcap.npz"),
        cfg_name="danbo_perfcap", weight_seed=seed, n_framecodes=20, N_samples=S, N_importance=Sf, n_uniques=1,
        ray_batch=rb, kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], cyls=scene["cyls"],
        rest_pose=rest, near=near.numpy(), far=far.numpy(), raw_coarse=raw.numpy(),
        view_inputs=view_in.numpy()[::S].copy(),
        **{"final_" + k: v for k, v in fin.items()})
    print("danbo_perfcap: acc mean", fin["acc_map"].mean())


def gen_danbo_train(stochastic=False):
    """one deterministic training forward/backward (perturb = 0, raw_noise_std = 0) of the reference:
    loss terms of Trainer.compute_loss and gradients of a representative parameter subset (stochastic=True would record the
    reference's draws as gen_danbo_perfcap_train does; no committed fixture uses it)"""
    seed = 14
    cfg, args, caster, kw_test, rest = build("danbo_base", seed)
    import types
    import core.trainer as rtr
    scene = syn.make_scene(n_poses=4, H=64, W=64, n_views=4, pose_seed=21)
    n_per = 32
    ro, rd, pose = [], [], []
    for p in range(4):
        o, d = body_rays(scene, p, n_per, seed=200 + p)
        ro.append(o); rd.append(d); pose += [p] * n_per
    ro, rd, pose = np.concatenate(ro), np.concatenate(rd), np.array(pose)
    rb = syn.ray_batch(ro, rd)
    kps, skts, bones, cyls = per_ray(scene, pose)
    cam_idx = (np.arange(len(pose)) % 5).astype(np.int64)
    rng = np.random.default_rng(7)
    target = rng.uniform(size=(len(pose), 3)).astype(np.float32)
    bgs = rng.uniform(size=(len(pose), 3)).astype(np.float32)
    S, Sf = 16, 8
    caster.train()
    kw = {k: v for k, v in kw_test.items() if k not in ("ray_caster", "use_viewdirs", "N_samples", "N_importance")}
    draws = {}
    if stochastic:
        kw.update(perturb=1.0, raw_noise_std=1.0)
        torch.manual_seed(4242)
        with RecordedDraws() as rec:
            preds = caster(T(rb), N_samples=S, kp_batch=T(kps), skts=T(skts), cyls=T(cyls), bones=T(bones),
                           cams=T(cam_idx, torch.long), N_importance=Sf, N_uniques=4, **kw)
        R = len(pose)
        kinds = [(k, tuple(t.shape)) for k, t in rec.calls]
        # the reference draws, in this order: stratified offsets, coarse density noise, inverse-CDF uniforms, fine density noise
        assert kinds == [("rand", (R, S)), ("randn", (R, S)), ("rand", (R, Sf)), ("randn", (R, S + Sf))], kinds
        B = float(kw["preproc_kwargs"]["density_scale"]) if "preproc_kwargs" in kw else 1.0
        draws = dict(t_rand=rec.calls[0][1].numpy(), noise_c=(rec.calls[1][1] * 1.0 * B).numpy(), u_rand=rec.calls[2][1].numpy(),
                     noise_f=(rec.calls[3][1] * 1.0 * B).numpy(), density_scale=np.float32(B))
    else:
        preds = caster(T(rb), N_samples=S, kp_batch=T(kps), skts=T(skts), cyls=T(cyls), bones=T(bones),
                       cams=T(cam_idx, torch.long), N_importance=Sf, N_uniques=4, **kw)
    wrap = types.SimpleNamespace(module=caster)
    tr = rtr.Trainer(args, dict(hwf=(64, 64, 80.0)), None, None, dict(ray_caster=wrap), dict(ray_caster=caster))
    loss_dict, stats = tr.compute_loss(dict(target_s=T(target), bgs=T(bgs)), preds, kp_opts=None, popt_detach=True)
    caster.zero_grad()
    loss_dict["total_loss"].backward()
    net = caster.network
    grads = {n: p.grad.numpy() for n, p in net.named_parameters() if p.grad is not None}
    keep = {"draw/" + k: v for k, v in draws.items()}
    if stochastic:
        keep["alpha"] = preds["alpha"].detach().numpy()
        keep["alpha0"] = preds["alpha0"].detach().numpy()
    for n in ("graph_net.axis_scale", "pts_linears.0.weight", "pts_linears.5.bias", "alpha_linear.weight",
              "rgb_linear.weight", "views_linears.0.bias", "framecodes.codes.weight", "prob_linears.layers.1.weight",
              "prob_linears.layers.0.adj_w", "prob_linears.layers.2.bias", "graph_net.layers.0.adj_w",
              "graph_net.layers.2.bias"):
        keep["grad/" + n] = grads[n]
    keep["grad/graph_net.layers.3.weight[:, ::16, ::8]"] = grads["graph_net.layers.3.weight"][:, ::16, ::8].copy()
    keep["grad/graph_net.layers.0.lin.weight[:, ::8, ::16]"] = grads["graph_net.layers.0.lin.weight"][:, ::8, ::16].copy()
    norms = {"gnorm/" + n: np.float64(np.sqrt((g.astype(np.float64) ** 2).sum())) for n, g in grads.items()}
    np.savez_compressed(
        os.path.join(OUT, "danbo_train.npz"),
        cfg_name="danbo_base", weight_seed=seed, n_framecodes=20, N_samples=S, N_importance=Sf, n_uniques=4,
        ray_batch=rb, kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], cyls=scene["cyls"],
        pose_of_ray=pose, cam_idx=cam_idx, target=target, bgs=bgs,
        rgb_map=preds["rgb_map"].detach().numpy(), acc_map=preds["acc_map"].detach().numpy(),
        rgb0=preds["rgb0"].detach().numpy(), part_invalid=preds["part_invalid"].detach().numpy(),
        **{"loss/" + k: np.float64(v.item()) for k, v in loss_dict.items()}, **keep, **norms)
    print("danbo_train:", {k: round(v.item(), 6) for k, v in loss_dict.items()})


class RecordedDraws:
    """While active, torch.rand / torch.randn hand out what they always do AND keep every draw, in call order -- so the
    reference's stochastic branches (stratified offsets ray_utils.py:240, density noise nerf.py:316, inverse-CDF uniforms
    ray_utils.py:171) can be replayed by feeding the same numbers to the HIP path."""
    def __enter__(self):
        self.calls = []
        self._rand, self._randn = torch.rand, torch.randn

        def rand(*a, **k):
            t = self._rand(*a, **k)
            self.calls.append(("rand", t.clone()))
            return t

        def randn(*a, **k):
            t = self._randn(*a, **k)
            self.calls.append(("randn", t.clone()))
            return t
        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn


def gen_danbo_perfcap_train(stochastic=False):
    """BASELINE config 4 in miniature: PerfCap danbo_fast (view_type relray + ray_tr_type root_local, per-bone box near/far,
    vol_scale_penalty as configured), one training forward/backward of the reference: loss terms of Trainer.compute_loss,
    gradient norms of every parameter and a representative set of gradients.
    stochastic=False -> danbo_perfcap_train.npz: perturb = 0, raw_noise_std = 0 (deterministic).
    stochastic=True  -> danbo_perfcap_train_noise.npz: config 4's ACTUAL settings perturb = 1, raw_noise_std = 1, with the
    reference's own random draws recorded (t_rand [R,S], noise_c [R,S] = randn * std * B as nerf.py:316 forms it, u [R,Sf],
    noise_f [R,S+Sf]) so the HIP step can be run on the same numbers."""
    seed = 17
    cfg, args, caster, kw_test, rest = build("danbo_perfcap", seed)
    import types
    import core.trainer as rtr
    scene = syn.make_scene(n_poses=4, H=64, W=64, n_views=4, pose_seed=31)
    n_per = 48
    ro, rd, pose = [], [], []
    for p in range(4):
        o, d = body_rays(scene, p, n_per, seed=300 + p)
        ro.append(o); rd.append(d); pose += [p] * n_per
    ro, rd, pose = np.concatenate(ro), np.concatenate(rd), np.array(pose)
    rb = syn.ray_batch(ro, rd)
    kps, skts, bones, cyls = per_ray(scene, pose)
    cam_idx = (np.arange(len(pose)) // n_per * 3 % 7).astype(np.int64)       # one camera per image, as the dataset hands over
    rng = np.random.default_rng(8)
    target = rng.uniform(size=(len(pose), 3)).astype(np.float32)
    bgs = rng.uniform(size=(len(pose), 3)).astype(np.float32)
    S, Sf = 16, 8
    caster.train()
    kw = {k: v for k, v in kw_test.items() if k not in ("ray_caster", "use_viewdirs", "N_samples", "N_importance")}
    draws = {}
    if stochastic:
        kw.update(perturb=1.0, raw_noise_std=1.0)
        torch.manual_seed(4242)
        with RecordedDraws() as rec:
            preds = caster(T(rb), N_samples=S, kp_batch=T(kps), skts=T(skts), cyls=T(cyls), bones=T(bones),
                           cams=T(cam_idx, torch.long), N_importance=Sf, N_uniques=4, **kw)
        R = len(pose)
        kinds = [(k, tuple(t.shape)) for k, t in rec.calls]
        # the reference draws, in this order: stratified offsets, coarse density noise, inverse-CDF uniforms, fine density noise
        assert kinds == [("rand", (R, S)), ("randn", (R, S)), ("rand", (R, Sf)), ("randn", (R, S + Sf))], kinds
        B = float(kw["preproc_kwargs"]["density_scale"]) if "preproc_kwargs" in kw else 1.0
        draws = dict(t_rand=rec.calls[0][1].numpy(), noise_c=(rec.calls[1][1] * 1.0 * B).numpy(), u_rand=rec.calls[2][1].numpy(),
                     noise_f=(rec.calls[3][1] * 1.0 * B).numpy(), density_scale=np.float32(B))
    else:
        preds = caster(T(rb), N_samples=S, kp_batch=T(kps), skts=T(skts), cyls=T(cyls), bones=T(bones),
                       cams=T(cam_idx, torch.long), N_importance=Sf, N_uniques=4, **kw)
    wrap = types.SimpleNamespace(module=caster)
    tr = rtr.Trainer(args, dict(hwf=(64, 64, 80.0)), None, None, dict(ray_caster=wrap), dict(ray_caster=caster))
    loss_dict, stats = tr.compute_loss(dict(target_s=T(target), bgs=T(bgs)), preds, kp_opts=None, popt_detach=True)
    caster.zero_grad()
    loss_dict["total_loss"].backward()
    net = caster.network
    grads = {n: p.grad.numpy() for n, p in net.named_parameters() if p.grad is not None}
    keep = {"draw/" + k: v for k, v in draws.items()}
    if stochastic:
        keep["alpha"] = preds["alpha"].detach().numpy()
        keep["alpha0"] = preds["alpha0"].detach().numpy()
    for n in ("graph_net.axis_scale", "pts_linears.0.weight", "pts_linears.5.bias", "pts_linears.7.bias", "alpha_linear.weight",
              "alpha_linear.bias", "feature_linear.bias", "rgb_linear.weight", "rgb_linear.bias", "views_linears.0.bias",
              "framecodes.codes.weight", "prob_linears.layers.0.bias", "prob_linears.layers.1.weight",
              "prob_linears.layers.1.bias", "prob_linears.layers.2.weight", "prob_linears.layers.0.adj_w",
              "prob_linears.layers.2.bias", "graph_net.layers.0.adj_w", "graph_net.layers.1.adj_w", "graph_net.layers.0.bias",
              "graph_net.layers.1.bias", "graph_net.layers.2.bias", "graph_net.layers.3.bias"):
        keep["grad/" + n] = grads[n]
    keep["grad/graph_net.layers.3.weight[:, ::16, ::8]"] = grads["graph_net.layers.3.weight"][:, ::16, ::8].copy()
    keep["grad/graph_net.layers.2.weight[:, ::16, ::16]"] = grads["graph_net.layers.2.weight"][:, ::16, ::16].copy()
    keep["grad/graph_net.layers.1.lin.weight[:, ::16, ::16]"] = grads["graph_net.layers.1.lin.weight"][:, ::16, ::16].copy()
    keep["grad/graph_net.layers.0.lin.weight[:, ::8, ::16]"] = grads["graph_net.layers.0.lin.weight"][:, ::8, ::16].copy()
    keep["grad/prob_linears.layers.0.lin.weight[:, ::3, ::4]"] = grads["prob_linears.layers.0.lin.weight"][:, ::3, ::4].copy()
    keep["grad/views_linears.0.weight[::4, ::8]"] = grads["views_linears.0.weight"][::4, ::8].copy()
    keep["grad/pts_linears.5.weight[::8, ::8]"] = grads["pts_linears.5.weight"][::8, ::8].copy()
    keep["grad/feature_linear.weight[::8, ::8]"] = grads["feature_linear.weight"][::8, ::8].copy()
    norms = {"gnorm/" + n: np.float64(np.sqrt((g.astype(np.float64) ** 2).sum())) for n, g in grads.items()}
    np.savez_compressed(
        os.path.join(OUT, "danbo_perfcap_train_noise.npz" if stochastic else "danbo_perfcap_train.npz"),
        cfg_name="danbo_perfcap", weight_seed=seed, n_framecodes=20, N_samples=S, N_importance=Sf, n_uniques=4,
        ray_batch=rb, kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], cyls=scene["cyls"],
        pose_of_ray=pose, cam_idx=cam_idx, target=target, bgs=bgs,
        rgb_map=preds["rgb_map"].detach().numpy(), acc_map=preds["acc_map"].detach().numpy(),
        rgb0=preds["rgb0"].detach().numpy(), acc0=preds["acc0"].detach().numpy(),
        part_invalid=preds["part_invalid"].detach().numpy(),
        **{"loss/" + k: np.float64(v.item()) for k, v in loss_dict.items()}, **keep, **norms)
    print("danbo_perfcap_train_noise:" if stochastic else "danbo_perfcap_train:", {k: round(v.item(), 6) for k, v in loss_dict.items()},
          "in-volume fraction", 1.0 - float(preds["part_invalid"].detach().numpy().all(-1).mean()))


def gen_danbo_mesh():
    """mesh-density path (reference raycasters.py:421-453): raw density on a (res+1)^3 grid around the root joint, res = 16,
    H36M danbo_base network; netchunk 1000 so that the chunked evaluation is exercised"""
    seed = 19
    cfg, args, caster, kw_test, rest = build("danbo_base", seed)
    scene = syn.make_scene(n_poses=1, H=32, W=32, n_views=1, pose_seed=41)
    kps, skts, bones = T(scene["kps"][:1]), T(scene["skts"][:1]), T(scene["bones"][:1])
    with torch.no_grad():
        dens = caster(kps, skts, bones, fwd_type='mesh', radius=0.9, res=16, netchunk=1000)
    np.savez_compressed(os.path.join(OUT, "danbo_mesh.npz"), cfg_name="danbo_base", weight_seed=seed, n_framecodes=20, res=16,
                        radius=0.9, kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], density=dens.numpy())
    print("danbo_mesh:", dens.shape, "non-constant fraction", float((dens != dens.flatten()[0]).float().mean()))


def gen_danbo_h36m_fast():
    """BASELINE config 2: H36M danbo_fast (world rays + identity view + frame codes, per-bone box near/far, 32 + 16 samples):
    bounds, coarse raw and the final maps of the reference's caster on 256 body rays of 2 poses"""
    seed = 21
    cfg, args, caster, kw_test, rest = build("danbo_fast", seed)
    scene = syn.make_scene(n_poses=2, H=96, W=96, n_views=4, pose_seed=51)
    ro, rd, pose = [], [], []
    for p_ in range(2):
        o, d = body_rays(scene, p_ + 1, 128, seed=400 + p_)
        ro.append(o); rd.append(d); pose += [p_] * 128
    ro, rd, pose = np.concatenate(ro), np.concatenate(rd), np.array(pose)
    rb = syn.ray_batch(ro, rd)
    kps, skts, bones, cyls = per_ray(scene, pose)
    S, Sf = 32, 16
    cam_idx = (np.arange(len(pose)) // 128 * 5 + 2).astype(np.int64)
    cams = T(cam_idx, torch.long)
    fin = call_caster(caster, kw_test, rb, kps, skts, cyls, bones, cams, S, Sf, 2)
    net = caster.network
    with torch.no_grad():
        rays_o, rays_d = T(ro), T(rd)
        near, far = caster.get_near_far(rays_o, rays_d, T(cyls), near=T(rb[:, 6:7]), far=T(rb[:, 7:8]), skts=T(skts))
        pts, z = caster.sample_pts(rays_o, rays_d, near, far, len(ro), S, 0., False)
        inputs = caster.get_nerf_inputs(pts, [rays_o[:, None, :], rays_d[:, None, :]], T(kps), T(skts), T(bones),
                                        cam_idxs=cams, N_uniques=2)
        raw, enc = net(inputs)
    invalid = enc["part_invalid"].numpy() if "part_invalid" in enc else None
    np.savez_compressed(
        os.path.join(OUT, "danbo_h36m_fast.npz"),
        cfg_name="danbo_fast", weight_seed=seed, n_framecodes=20, N_samples=S, N_importance=Sf, n_uniques=2,
        ray_batch=rb, kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], cyls=scene["cyls"], pose_of_ray=pose,
        cam_idx=cam_idx, rest_pose=rest, near=near.numpy(), far=far.numpy(), raw_coarse=raw.numpy(),
        **({} if invalid is None else {"in_volume_fraction": np.float64(1.0 - invalid.all(-1).mean())}),
        **{"final_" + k: v for k, v in fin.items()})
    print("danbo_h36m_fast: acc mean", fin["acc_map"].mean(), "rays hitting a box",
          float((np.abs(near.numpy() - near.numpy().mean()) > 0).mean()))


def gen_anerf_stages():
    seed = 15
    cfg, args, caster, kw_test, rest = build("anerf_base", seed)
    net = caster.network
    scene = syn.make_scene(n_poses=2, H=64, W=64, n_views=2, pose_seed=9)
    n_per = 24
    ro, rd, pose = [], [], []
    for p in range(2):
        o, d = body_rays(scene, p, n_per, seed=300 + p)
        ro.append(o); rd.append(d); pose += [p] * n_per
    ro, rd, pose = np.concatenate(ro), np.concatenate(rd), np.array(pose)
    rb = syn.ray_batch(ro, rd)
    kps, skts, bones, cyls = per_ray(scene, pose)
    cam_idx = (np.arange(len(pose)) % 7).astype(np.int64)
    S, Sf = 12, 6
    keep = {}
    for tag, tau in (("", 20.0), ("tau2000_", 2000.0)):
        for fn in (net.pe_fn, net.dirs_pe_fn):
            fn.tau.fill_(tau)
        fin = call_caster(caster, kw_test, rb, kps, skts, cyls, bones, T(cam_idx, torch.long), S, Sf, 2)
        with torch.no_grad():
            rays_o, rays_d = T(ro), T(rd)
            near, far = caster.get_near_far(rays_o, rays_d, T(cyls), near=T(rb[:, 6:7]), far=T(rb[:, 7:8]), skts=T(skts))
            pts, z = caster.sample_pts(rays_o, rays_d, near, far, len(pose), S, 0., False)
            inputs = caster.get_nerf_inputs(pts, [rays_o[:, None, :], rays_d[:, None, :]], T(kps), T(skts), T(bones),
                                            cam_idxs=T(cam_idx, torch.long), N_uniques=2)
            dens_in, enc = net.encode_pts(inputs)
            view_in, enc_v = net.encode_views(inputs, refs=enc["pts_t"], encoded_pts=enc)
            raw, _ = net(inputs)
        keep.update({tag + "raw_coarse": raw.numpy(), **{tag + "final_" + k: v for k, v in fin.items()}})
        if not tag:
            keep.update(near=near.numpy(), far=far.numpy(), z_coarse=z.numpy(), pts=pts.numpy(),
                        v=enc["v"].numpy(), r=enc["r"].numpy(), density_inputs=dens_in.numpy(),
                        view_dirs=enc_v["d"].numpy()[:, 0].copy(), view_inputs=view_in.numpy()[::5].copy())
            print("anerf has align:", inputs["align_transforms"] is not None)
    np.savez_compressed(
        os.path.join(OUT, "anerf_stages.npz"),
        cfg_name="anerf_base", weight_seed=seed, n_framecodes=20, N_samples=S, N_importance=Sf, n_uniques=2,
        ray_batch=rb, kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], cyls=scene["cyls"],
        pose_of_ray=pose, cam_idx=cam_idx, rest_pose=rest, **keep)
    print("anerf_stages: acc mean", keep["final_acc_map"].mean(), keep["tau2000_final_acc_map"].mean(),
          "raw alpha range", keep["raw_coarse"][..., 3].min(), keep["raw_coarse"][..., 3].max(),
          "rgb spread", keep["final_rgb_map"].std(0))


def gen_anerf_train():
    """A-NeRF: one deterministic training forward/backward of the reference (perturb = 0, raw_noise_std = 0)"""
    import types
    seed = 16
    cfg, args, caster, kw_test, rest = build("anerf_base", seed)
    import core.trainer as rtr
    scene = syn.make_scene(n_poses=4, H=64, W=64, n_views=4, pose_seed=23)
    n_per = 24
    ro, rd, pose = [], [], []
    for p in range(4):
        o, d = body_rays(scene, p, n_per, seed=400 + p)
        ro.append(o); rd.append(d); pose += [p] * n_per
    ro, rd, pose = np.concatenate(ro), np.concatenate(rd), np.array(pose)
    rb = syn.ray_batch(ro, rd)
    kps, skts, bones, cyls = per_ray(scene, pose)
    cam_idx = (np.arange(len(pose)) % 5).astype(np.int64)
    rng = np.random.default_rng(8)
    target = rng.uniform(size=(len(pose), 3)).astype(np.float32)
    bgs = rng.uniform(size=(len(pose), 3)).astype(np.float32)
    S, Sf = 12, 6
    caster.train()
    kw = {k: v for k, v in kw_test.items() if k not in ("ray_caster", "use_viewdirs", "N_samples", "N_importance")}
    preds = caster(T(rb), N_samples=S, kp_batch=T(kps), skts=T(skts), cyls=T(cyls), bones=T(bones),
                   cams=T(cam_idx, torch.long), N_importance=Sf, N_uniques=4, **kw)
    wrap = types.SimpleNamespace(module=caster)
    tr = rtr.Trainer(args, dict(hwf=(64, 64, 80.0)), None, None, dict(ray_caster=wrap), dict(ray_caster=caster))
    loss_dict, stats = tr.compute_loss(dict(target_s=T(target), bgs=T(bgs)), preds, kp_opts=None, popt_detach=True)
    caster.zero_grad()
    loss_dict["total_loss"].backward()
    grads = {n: p.grad.numpy() for n, p in caster.network.named_parameters() if p.grad is not None}
    keep = {"grad/" + n: grads[n] for n in ("alpha_linear.weight", "rgb_linear.weight", "views_linears.0.bias",
                                            "framecodes.codes.weight", "pts_linears.7.bias")}
    keep["grad/views_linears.0.weight[::8, ::16]"] = grads["views_linears.0.weight"][::8, ::16].copy()
    keep["grad/pts_linears.0.weight[::16, ::8]"] = grads["pts_linears.0.weight"][::16, ::8].copy()
    norms = {"gnorm/" + n: np.float64(np.sqrt((g.astype(np.float64) ** 2).sum())) for n, g in grads.items()}
    np.savez_compressed(
        os.path.join(OUT, "anerf_train.npz"),
        cfg_name="anerf_base", weight_seed=seed, n_framecodes=20, N_samples=S, N_importance=Sf, n_uniques=4,
        ray_batch=rb, kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], cyls=scene["cyls"],
        pose_of_ray=pose, cam_idx=cam_idx, target=target, bgs=bgs,
        rgb_map=preds["rgb_map"].detach().numpy(), acc_map=preds["acc_map"].detach().numpy(),
        **{"loss/" + k: np.float64(v.item()) for k, v in loss_dict.items()}, **keep, **norms)
    print("anerf_train:", {k: round(v.item(), 6) for k, v in loss_dict.items()}, "params with grad", len(grads))


def gen_ckpt_manifest():
    """wire format of the reference's checkpoints (trainer.py:597-618, raycasters.py:601-637): top-level keys of the
    saved dict and name -> shape of every tensor, for a DANBO and an A-NeRF caster"""
    import json
    out = {}
    for name in ("danbo_base", "anerf_base"):
        cfg, args, caster, kw_test, rest = build(name, 1)
        ck = caster.state_dict()
        out[name] = {top: {k: list(v.shape) for k, v in sub.items()} for top, sub in ck.items()}
    with open(os.path.join(OUT, "ckpt_manifest.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("ckpt_manifest:", {k: {t: len(v) for t, v in d.items()} for k, d in out.items()})


def gen_args_txt():
    """`args.txt` exactly as run_nerf.py:590-594 writes it for the danbo_base config, and the argv the reference's
    txt_to_argstring (core/utils/evaluation_helpers.py:221-255, lifted with ast: the module itself needs cv2 / imageio)
    turns it back into"""
    import ast as _ast
    import json
    args = rh.parse_reference_config(CONFIGS["danbo_base"])
    args.basedir, args.expname = "./logs", "danbo_base"
    text = "".join("{} = {}\n".format(k, getattr(args, k)) for k in sorted(vars(args)))
    src = open(os.path.join(rh.REF_ROOT, "core", "utils", "evaluation_helpers.py")).read()
    fn = next(n for n in _ast.parse(src).body if isinstance(n, _ast.FunctionDef) and n.name == "txt_to_argstring")
    ns = {}
    exec(compile(_ast.Module(body=[fn], type_ignores=[]), "txt_to_argstring", "exec"), ns)
    path = os.path.join(tempfile.mkdtemp(), "args.txt")
    open(path, "w").write(text)
    argv = ns["txt_to_argstring"](path)
    argv_noconf = ns["txt_to_argstring"](path, ignore_config=True)
    with open(os.path.join(OUT, "args_txt.json"), "w") as f:
        json.dump(dict(args_txt=text, argv=argv, argv_ignore_config=argv_noconf), f, indent=0)
    print("args_txt:", len(text.splitlines()), "lines ->", len(argv), "argv tokens")


def gen_valid_rays():
    """kp_to_valid_rays of the reference (core/utils/ray_utils.py:84-138) on a synthetic scene: 2 poses x 2 cameras"""
    rh.install_stubs()
    from core.utils.ray_utils import kp_to_valid_rays
    scene = syn.make_scene(n_poses=2, H=48, W=64, n_views=2, pose_seed=31, cam_dist=6.5)
    cams = np.stack([scene["cams"][0], scene["cams"][0], scene["cams"][1], scene["cams"][1]]).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        rays, idxs, cyl, boxes = kp_to_valid_rays(T(cams), 48, 64, float(scene["focal"]), kps=T(scene["kps"]), ext_scale=0.001)
    np.savez_compressed(
        os.path.join(OUT, "valid_rays.npz"), kps=scene["kps"], cams=cams, H=48, W=64, focal=scene["focal"],
        cyl=cyl.numpy(), boxes=np.array([[b[0], b[1]] for b in boxes]),
        n_valid=np.array([len(i) for i in idxs]), idx0=idxs[0].numpy(), idx3=idxs[3].numpy(),
        rays_o3=rays[3][0].numpy(), rays_d3=rays[3][1].numpy())
    print("valid_rays: boxes", [(tuple(b[0]), tuple(b[1])) for b in boxes], "n", [len(i) for i in idxs])


def _sequence_inputs():
    """4 poses / 4 cameras with off-axis camera translations and non-zero roots, so that every centring branch does something"""
    rest = syn.rest_pose(0.48)
    bones = syn.random_bones(4, seed=41).astype(np.float32)
    _, _, kps = syn.forward_kinematics(bones, rest)
    rng = np.random.default_rng(7)
    kps = (kps + rng.normal(0, 0.3, size=(4, 1, 3))).astype(np.float32)
    c2ws = syn.bullet_cameras(4, dist=3.0).astype(np.float32)
    c2ws[:, :3, 3] += rng.normal(0, 0.2, size=(4, 3)).astype(np.float32)
    focals = np.array([80., 82., 84., 86.], dtype=np.float32)
    centers = rng.uniform(28, 36, size=(4, 2)).astype(np.float32)
    return rest, bones, kps, c2ws, focals, centers


def gen_sequences():
    """Camera / pose sequence generators of the reference's run_render.py (:838-999) and load_data.py (:56-71), run from the
    reference source with the `refined=(kps, bones)` in-memory branch (the HDF5 branch needs deepdish)."""
    import math
    rh.install_stubs()
    from core.utils.skeleton_utils import get_smpl_l2ws, rotate_x, rotate_y, rotate_z
    (gbt,) = rh.lift_functions("core/load_data.py", ["generate_bullet_time"],
                               dict(np=np, math=math, rotate_x=rotate_x, rotate_y=rotate_y, rotate_z=rotate_z))
    _, bullet, interp, selected, bubble = rh.lift_functions(
        "run_render.py", ["find_idxs_with_map", "load_bullettime", "load_interpolate", "load_selected", "load_bubble"],
        dict(np=np, get_smpl_l2ws=get_smpl_l2ws, generate_bullet_time=gbt, rotate_x=rotate_x, rotate_y=rotate_y))
    rest, bones, kps, c2ws, focals, centers = _sequence_inputs()
    sel = np.array([2, 0, 3])
    out = dict(rest=rest, bones=bones, kps=kps, c2ws=c2ws, focals=focals, centers=centers, sel=sel,
               ring_x=gbt(c2ws[1], 5, 'x'), ring_y=gbt(c2ws[1], 5, 'y'), ring_z=gbt(c2ws[1], 5, 'z'))
    fresh = lambda: dict(refined=(kps.copy(), bones.copy()))  # noqa: E731  (the loaders edit their inputs in place)
    names = ("kps", "skts", "c2ws", "cam_idxs", "focals", "bones", "centers")
    for tag, kw in (("bt", dict()), ("bt_nokp", dict(center_kps=False)), ("bt_raw", dict(center_kps=False, center_cam=False, undo_rot=True))):
        r = bullet(None, c2ws.copy(), focals.copy(), rest, None, sel, n_bullet=3, centers=centers.copy(), **kw, **fresh())
        out.update({f"{tag}_{n}": v for n, v in zip(names, r)})
    for tag, kw in (("ip", dict()), ("ip_c", dict(center_cam=True)), ("ip_k", dict(center_kps=True))):
        r = interp(None, c2ws.copy(), focals.copy(), rest, None, sel, n_step=4, **kw, **fresh())
        out.update({f"{tag}_{n}": v for n, v in zip(names[:5], r)})
    r = selected(None, c2ws.copy(), focals.copy(), rest, None, sel, centers=centers.copy(), **fresh())
    out.update({f"sel_{n}": v for n, v in zip(names, r)})
    from core.utils.skeleton_utils import axisang_to_rot, rot_to_axisang
    (pose_rotate,) = rh.lift_functions("run_render.py", ["load_pose_rotate"],
                                       dict(np=np, torch=torch, get_smpl_l2ws=get_smpl_l2ws, generate_bullet_time=gbt,
                                            axisang_to_rot=axisang_to_rot, rot_to_axisang=rot_to_axisang,
                                            find_idxs_with_map=lambda s_, m_: s_))
    r = pose_rotate(None, c2ws.copy(), focals.copy(), rest, None, np.array([2]), n_bullet=9, **fresh())
    out.update({f"pr_{n}": v for n, v in zip(("kps", "skts", "bones", "c2ws", "cam_idxs", "focals"), r)})
    r = bubble(None, c2ws.copy(), focals.copy(), rest, None, sel, centers=centers.copy(), n_step=4, **fresh())
    out.update({f"bb_{n}": v for n, v in zip(names, r)})
    np.savez_compressed(os.path.join(OUT, "sequences.npz"), **out)
    print("sequences:", {k: v.shape for k, v in out.items() if k.startswith("bt_") and "raw" not in k and "nokp" not in k})


def gen_render_path():
    """`render_path` of the reference (run_nerf.py:29-147) end to end on CPU: D-Surr network, 2 poses x 2 cameras at 40 x 32,
    valid-ray boxes, background images blended by (1 - acc); and once more with a white background"""
    import torch.nn.functional as F
    seed = 17
    cfg, args, caster, kw_test, rest = build("danbo_surreal", seed, n_framecodes=4)
    from core.trainer import render
    from core.utils.ray_utils import kp_to_valid_rays
    (rp,) = rh.lift_functions("run_nerf.py", ["render_path"],
                              dict(np=np, torch=torch, F=F, time=__import__("time"), tqdm=lambda x: x, render=render,
                                   kp_to_valid_rays=kp_to_valid_rays))
    scene = syn.make_scene(n_poses=2, H=40, W=32, n_views=2, pose_seed=23, rest_scale=cfg["rest_scale"], cam_dist=7.0)
    cams = np.stack([scene["cams"][0], scene["cams"][1], scene["cams"][1], scene["cams"][0]]).astype(np.float32)
    rng = np.random.default_rng(3)
    bg_imgs = rng.uniform(size=(2, 20, 16, 3)).astype(np.float32)          # half resolution: exercises the bilinear resize
    bg_indices = np.array([1, 0, 0, 1])
    kw = dict(kw_test, N_samples=16, N_importance=8)
    res = {}
    for tag, extra in (("bg", dict(bg_imgs=bg_imgs, bg_indices=bg_indices)), ("white", dict(white_bkgd=True))):
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            rgbs, disps, accs, idxs, boxes = rp(T(cams), (40, 32, float(scene["focal"])), 4096, kw, kp=T(scene["kps"]),
                                                skts=T(scene["skts"]), bones=T(scene["bones"]), ret_acc=True, ext_scale=0.001, **extra)
        res.update({f"{tag}_rgbs": rgbs, f"{tag}_disps": disps, f"{tag}_accs": accs})
    np.savez_compressed(os.path.join(OUT, "render_path.npz"), cfg_name="danbo_surreal", weight_seed=seed, n_framecodes=4,
                        N_samples=16, N_importance=8, pose_seed=23, cam_dist=7.0, H=40, W=32, focal=scene["focal"], cams=cams,
                        kps=scene["kps"], skts=scene["skts"], bones=scene["bones"], bg_imgs=bg_imgs, bg_indices=bg_indices,
                        boxes=np.array([[b[0], b[1]] for b in boxes]), n_valid=np.array([len(i) for i in idxs]), **res)
    print("render_path: boxes", [(tuple(b[0]), tuple(b[1])) for b in boxes], "acc mean", res["bg_accs"].mean())


def gen_pose_rot6d():
    rh.install_stubs()
    from core.utils.skeleton_utils import axisang_to_rot6d
    rng = np.random.default_rng(0)
    aa = rng.normal(0, 0.6, size=(6, 24, 3)).astype(np.float32)
    aa[0, :4] = 0.0
    aa[0, 4] = [1e-8, -2e-8, 1e-9]
    aa[0, 5] = [3e-7, 0, 0]
    aa[1, 0] = [np.pi - 1e-3, 0, 0]
    out = axisang_to_rot6d(T(aa)).numpy()
    np.savez_compressed(os.path.join(OUT, "pose_rot6d.npz"), axis_angle=aa, rot6d=out)


def gen_confd_colours():
    """the two assignment visualisations of NeRF.raw2outputs (reference core/networks/misc.py:620-673) on random logits"""
    rh.install_stubs()
    from core.networks.misc import get_confidence_rgb, get_entropy_rgb
    rng = np.random.default_rng(12)
    confd = (rng.normal(size=(6, 5, 24)) * 3.0).astype(np.float32)
    confd[0, 0] = 0.0                         # uniform: maximal entropy
    confd[0, 1] = -50.0
    confd[0, 1, 7] = 50.0                     # one bone: zero entropy
    np.savez_compressed(os.path.join(OUT, "confd_colours.npz"), confd=confd,
                        confidence_rgb=get_confidence_rgb(T(confd), None).numpy(), entropy_rgb=get_entropy_rgb(T(confd), None).numpy())


def torch_norm_probe(n=200000, seed=3):
    """does torch.norm(x, dim=-1) of THIS torch build on THIS CPU equal the fma chain the oracle and the kernels restate
    (danbo_oracle.torch_norm, csrc/sample_math.hpp norm3_torch)?  ATen's contraction is compiler / ISA dependent:
    -> (rows equal, rows, max ulp distance)"""
    import danbo_oracle as o
    rng = np.random.default_rng(seed)
    x = (rng.normal(size=(n, 3)) * np.exp(rng.uniform(-6, 6, size=(n, 1)))).astype(np.float32)
    a = torch.norm(torch.tensor(x), dim=-1).numpy()
    b = o.torch_norm(x)
    ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    return int((ulp == 0).sum()), n, int(ulp.max())


def gen_provenance():
    """what the fixtures were generated WITH: the near / far goldens are bit-exact statements about torch.norm's fma contraction
    on the generating host"""
    import platform
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    eq, n, ulp = torch_norm_probe()
    doc = dict(torch=torch.__version__, numpy=np.__version__, python=platform.python_version(), machine=platform.machine(), cpu=cpu,
               torch_norm_equals_fma_chain=dict(rows_equal=eq, rows=n, max_ulp=ulp),
               note="tests/golden/*.npz were written by oracle/gen_golden.py importing /root/reference with this torch build on this "
                    "CPU; where torch.norm is not the fma chain (another build / ISA) regenerated near / far bounds may differ from "
                    "these by <= 1 ulp, which this network amplifies (DESIGN.md section 4)")
    import json
    with open(os.path.join(OUT, "PROVENANCE.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("provenance:", doc["torch"], doc["cpu"], doc["torch_norm_equals_fma_chain"])


# target name -> generator, in the default order (`python oracle/gen_golden.py` runs every one; tests/test_golden_recipe.py does the
# same into a scratch directory and compares the files with the committed ones)
TARGETS = {
    "stages": lambda: gen_danbo_stages(),
    "surreal": lambda: gen_danbo_surreal(),
    "perfcap": lambda: gen_danbo_perfcap(),
    "rot6d": lambda: gen_pose_rot6d(),
    "train": lambda: gen_danbo_train(),
    "anerf": lambda: gen_anerf_stages(),
    "anerf_train": lambda: gen_anerf_train(),
    "perfcap_train": lambda: gen_danbo_perfcap_train(),
    "confd_colours": lambda: gen_confd_colours(),
    "perfcap_train_noise": lambda: gen_danbo_perfcap_train(stochastic=True),
    "mesh": lambda: gen_danbo_mesh(),
    "h36m_fast": lambda: gen_danbo_h36m_fast(),
    "ckpt": lambda: gen_ckpt_manifest(),
    "args": lambda: gen_args_txt(),
    "valid_rays": lambda: gen_valid_rays(),
    "sequences": lambda: gen_sequences(),
    "render_path": lambda: gen_render_path(),
    "provenance": lambda: gen_provenance(),
}


def main(argv):
    """python oracle/gen_golden.py [--out DIR] [target ...]   (no target: all of TARGETS)"""
    global OUT
    argv = list(argv)
    if "--out" in argv:
        i = argv.index("--out")
        OUT = os.path.abspath(argv[i + 1])
        del argv[i:i + 2]
    assert rh.reference_available(), "needs /root/reference (build container only)"
    unknown = [t for t in argv if t not in TARGETS]
    if unknown:
        raise SystemExit(f"unknown target(s) {unknown}; known: {list(TARGETS)}")
    os.makedirs(OUT, exist_ok=True)
    for name, fn in TARGETS.items():
        if not argv or name in argv:
            torch.manual_seed(0)
            fn()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1:])
