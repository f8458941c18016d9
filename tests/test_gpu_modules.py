"""GPU tests of the drop-in module surface (core.networks / core.raycasters) against the
golden vectors captured from the reference's own `create_raycaster` / `caster(...)` calls."""
import os

import numpy as np
import pytest
import torch

import danbo_oracle as o
from helpers import ROOT, golden, max_err, rel_err, raw_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(x, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(x), dtype=dtype, device=DEV)


def N(t):
    return t.detach().cpu().numpy()


def build(cfg_file, g, rest_scale=0.48):
    from core.config import parse_args
    from core.raycasters import create_raycaster
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import SMPLSkeleton
    args = parse_args(["--no_reload"], config=os.path.join(ROOT, "danbo-pytorch_amd", "configs", cfg_file))
    n_codes = int(g["n_framecodes"])
    da = dict(skel_type=SMPLSkeleton, near=0., far=100., n_views=n_codes, rest_pose=syn.rest_pose(rest_scale),
              hwf=(64, 64, 80.))
    tr, te, *_ = create_raycaster(args, da, device=DEV)
    caster = te["ray_caster"].eval()
    cfg = syn.model_config(str(g["cfg_name"]))
    sd = syn.make_state_dict(cfg, int(g["weight_seed"]), n_codes, syn.rest_pose(rest_scale))
    caster.network.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    kw = {k: v for k, v in te.items() if k not in ("ray_caster", "use_viewdirs", "N_samples", "N_importance")}
    return caster, kw


def test_caster_call_matches_reference_caster_output():
    """exactly the call the reference's batchify_rays makes (trainer.py:83), per-ray replicated poses"""
    g = golden("danbo_stages")
    caster, kw = build("h36m_zju/danbo_base.txt", g)
    pose = g["pose_of_ray"]
    out = caster(T(g["ray_batch"]), N_samples=int(g["N_samples"]), kp_batch=T(g["kps"][pose]), skts=T(g["skts"][pose]),
                 cyls=T(g["cyls"][pose]), bones=T(g["bones"][pose]), cams=T(g["cam_idx"], torch.int64),
                 N_importance=int(g["N_importance"]), N_uniques=2, **kw)
    assert set(out) == {"rgb_map", "disp_map", "acc_map", "alpha", "T_i", "rgb0", "disp0", "acc0", "alpha0"}
    for k in ("rgb_map", "acc_map", "alpha", "T_i", "rgb0", "acc0", "alpha0"):
        assert max_err(N(out[k]), g["final_" + k]) < 3e-5, k            # measured 5.9e-6
    assert o.psnr(N(out["rgb_map"]), g["final_rgb_map"]) > 95.0


def test_model_forward_on_reference_nerf_inputs():
    """DANBO.forward(inputs) with the reference's nerf_inputs dict (raycasters.py:399-413)"""
    g = golden("danbo_stages")
    caster, kw = build("h36m_zju/danbo_base.txt", g)
    pose = g["pose_of_ray"]
    R = len(pose)
    rb = g["ray_batch"]
    inputs = dict(pts=T(g["pts"]), kps=T(g["kps"][pose]), skts=T(g["skts"][pose]), bones=T(g["bones"][pose]),
                  rest_pose=T(g["rest_pose"]).reshape(1, 1, 24, 3), align_transforms=caster.transforms[:1, None].to(DEV),
                  N_uniques=2, rays_o=T(rb[:, None, 0:3]), rays_d=T(rb[:, None, 3:6]), cam_idxs=T(g["cam_idx"], torch.int64))
    raw, enc = caster.network(inputs)
    assert raw.shape == (R, int(g["N_samples"]), 4)
    assert raw_err(N(raw), g["raw_coarse"]) < 1e-4
    # eval-mode `encoded` as the reference returns it (core/networks/danbo.py:341-346): formed on first access
    assert set(enc.keys()) == {"confd", "part_invalid"}
    assert enc["confd"].shape == g["confd"].shape and max_err(N(enc["confd"]), g["confd"]) < 2e-5
    assert np.array_equal(N(enc["part_invalid"]), g["invalid"].astype(np.float32).reshape(g["confd"].shape))
    p = caster.network.sigmoid(enc["confd"], enc["part_invalid"], mask_invalid=False)   # what trainer.py:521 does with them
    assert p.shape == enc["confd"].shape
    out = caster.network.raw2outputs(raw, T(g["z_coarse"]), T(rb[:, 3:6]), B=1.0)
    assert max_err(N(out["weights"]), g["weights_coarse"]) < 2e-5
    assert max_err(N(out["rgb_map"]), g["rgb_coarse"]) < 2e-5


def test_perfcap_caster_with_box_near_far_and_mean_framecode():
    g = golden("danbo_perfcap")
    caster, kw = build("perfcap/danbo_fast.txt", g)
    R = len(g["ray_batch"])
    z = np.zeros(R, np.int64)
    out = caster(T(g["ray_batch"]), N_samples=int(g["N_samples"]), kp_batch=T(g["kps"][z]), skts=T(g["skts"][z]),
                 cyls=T(g["cyls"][z]), bones=T(g["bones"][z]), cams=T(-np.ones(R), torch.int64),
                 N_importance=int(g["N_importance"]), N_uniques=1, **kw)
    # near/far are recomputed here -- and equal the reference's bit for bit since round 5 (measured: acc 1.2e-6; round 4's bounds:
    # 45 dB / 5e-2 "1-ulp differences move samples")
    assert o.psnr(N(out["rgb_map"]), g["final_rgb_map"]) > 95.0
    assert max_err(N(out["acc_map"]), g["final_acc_map"]) < 1e-5


def test_density_query_for_mesh_extraction():
    g = golden("danbo_stages")
    caster, kw = build("h36m_zju/danbo_base.txt", g)
    pts = g["pts"][:24].reshape(-1, 1, 3)                       # samples of pose 0
    dens = caster(T(pts), T(g["kps"][:1]), T(g["skts"][:1]), T(g["bones"][:1]), fwd_type="density")
    want = g["raw_coarse"][:24].reshape(-1, 4)[:, 3:4]
    assert raw_err(N(dens), want) < 1e-4


def test_long_rays_96_plus_48_samples_against_oracle():
    """SURVEY 8(d) config 3 (danbo_base recipe: 96 coarse + 48 importance samples): more than 64 samples per ray
    takes the chunked composite and the general importance kernel"""
    from helpers import oracle_for
    g = golden("danbo_stages")
    caster, kw = build("h36m_zju/danbo_base.txt", g)
    orc, cfg, sd, rest = oracle_for(g)
    pose = g["pose_of_ray"]
    out = caster(T(g["ray_batch"]), N_samples=96, kp_batch=T(g["kps"][pose]), skts=T(g["skts"][pose]),
                 cyls=T(g["cyls"][pose]), bones=T(g["bones"][pose]), cams=T(g["cam_idx"], torch.int64),
                 N_importance=48, N_uniques=2, **kw)
    ref = orc.render(g["ray_batch"], g["skts"][pose], g["bones"][pose], g["cyls"][pose], cam_idxs=g["cam_idx"],
                     n_uniques=2, N_samples=96, N_importance=48)
    assert out["T_i"].shape == (len(pose), 144)
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        assert max_err(N(out[k]), ref[k]) < 1e-5, k                    # measured 1.4e-6
    assert o.psnr(N(out["rgb_map"]), ref["rgb_map"]) > 100.0


def test_edge_cases_no_body_hit_single_ray_and_ragged_counts():
    """rays that never enter a bone volume (count = 0 everywhere), a single ray, and ray counts that are not multiples
    of any tile size: the compacted-row kernels must cope with 0 and ragged row counts"""
    from helpers import oracle_for
    from core.utils import synthetic as syn
    g = golden("danbo_stages")
    caster, kw = build("h36m_zju/danbo_base.txt", g)
    orc, cfg, sd, rest = oracle_for(g)
    scene = syn.make_scene(n_poses=1, H=16, W=16, n_views=1, pose_seed=4)
    ro, rd = scene["rays"][0]
    # (a) look away from the body: same origins, directions mirrored -> cylinder missed, no sample in any volume
    for R in (1, 7, 130):
        rb = syn.ray_batch(ro[:R], -rd[:R])
        z = np.zeros(R, np.int64)
        out = caster(T(rb), N_samples=16, kp_batch=T(scene["kps"][z]), skts=T(scene["skts"][z]), cyls=T(scene["cyls"][z]),
                     bones=T(scene["bones"][z]), cams=T(np.zeros(R), torch.int64), N_importance=8, N_uniques=1, **kw)
        ref = orc.render(rb, scene["skts"][z], scene["bones"][z], scene["cyls"][z], cam_idxs=np.zeros(R, np.int64),
                         n_uniques=1, N_samples=16, N_importance=8, chunk=R)
        assert out["rgb_map"].shape == (R, 3) and bool(torch.isfinite(out["rgb_map"]).all())
        assert max_err(N(out["acc_map"]), ref["acc_map"]) < 1e-5 and max_err(N(out["rgb_map"]), ref["rgb_map"]) < 1e-5
    # (b) ragged counts through the body
    for R in (1, 3, 65, 200):
        sel = (np.arange(R) + 16 * 4) % len(ro)   # starts in the rows through the torso
        rb = syn.ray_batch(ro[sel], rd[sel])
        z = np.zeros(R, np.int64)
        out = caster(T(rb), N_samples=16, kp_batch=T(scene["kps"][z]), skts=T(scene["skts"][z]), cyls=T(scene["cyls"][z]),
                     bones=T(scene["bones"][z]), cams=T(np.zeros(R), torch.int64), N_importance=8, N_uniques=1, **kw)
        ref = orc.render(rb, scene["skts"][z], scene["bones"][z], scene["cyls"][z], cam_idxs=np.zeros(R, np.int64),
                         n_uniques=1, N_samples=16, N_importance=8, chunk=R)
        assert max_err(N(out["rgb_map"]), ref["rgb_map"]) < 2e-3 and max_err(N(out["acc_map"]), ref["acc_map"]) < 2e-3, R


def test_whole_chunk_misses_the_cylinder_falls_back_to_the_placeholder_bounds():
    """every ray of the chunk misses the bounding cylinder (here: rays parallel to its axis, for which the 2-D
    intersection is 0/0): the nan-mean back-fill has nothing to average and the placeholder bounds (0, 1) stay
    (reference ray_utils.py:294-346 as restated by the oracle) -- no NaN, no out-of-bounds access on the way"""
    from helpers import oracle_for
    from core.utils import synthetic as syn
    g = golden("danbo_stages")
    caster, kw = build("h36m_zju/danbo_base.txt", g)
    orc, cfg, sd, rest = oracle_for(g)
    scene = syn.make_scene(n_poses=1, H=8, W=8, n_views=1, pose_seed=4)
    ro, rd = scene["rays"][0]
    R = 37
    up = np.zeros((R, 3), np.float32)
    up[:, 1] = 1.0
    rb = syn.ray_batch(ro[:R], up)
    z = np.zeros(R, np.int64)
    out = caster(T(rb), N_samples=16, kp_batch=T(scene["kps"][z]), skts=T(scene["skts"][z]), cyls=T(scene["cyls"][z]),
                 bones=T(scene["bones"][z]), cams=T(np.zeros(R), torch.int64), N_importance=8, N_uniques=1, **kw)
    ref = orc.render(rb, scene["skts"][z], scene["bones"][z], scene["cyls"][z], cam_idxs=np.zeros(R, np.int64), n_uniques=1,
                     N_samples=16, N_importance=8, chunk=R)
    assert out["rgb_map"].shape == (R, 3) and out["T_i"].shape == (R, 24)
    assert bool(torch.isfinite(out["rgb_map"]).all())
    assert max_err(N(out["rgb_map"]), ref["rgb_map"]) < 1e-5 and max_err(N(out["acc_map"]), ref["acc_map"]) < 1e-5


def test_small_chunks_replay_a_hip_graph_and_equal_the_eager_chain():
    """chunks of <= 8192 rays (the reference validates with 512-ray chunks) go through a captured HIP graph: same kernels,
    bit-identical outputs; a parameter update drops the graphs"""
    g = golden("danbo_stages")
    caster, kw = build("h36m_zju/danbo_base.txt", g)
    pose = g["pose_of_ray"]
    args = (T(g["ray_batch"]),)
    kwargs = dict(N_samples=int(g["N_samples"]), kp_batch=T(g["kps"][pose]), skts=T(g["skts"][pose]), cyls=T(g["cyls"][pose]),
                  bones=T(g["bones"][pose]), cams=T(g["cam_idx"], torch.int64), N_importance=int(g["N_importance"]), N_uniques=2, **kw)
    caster.use_graphs = False
    eager = {k: v.clone() for k, v in caster(*args, **kwargs).items()}
    caster.use_graphs = True
    first = {k: v.clone() for k, v in caster(*args, **kwargs).items()}          # captures
    again = caster(*args, **kwargs)                                              # replays
    assert len(caster._graphs.graphs) == 1
    for k in eager:
        assert torch.equal(eager[k], first[k]) and torch.equal(eager[k], again[k]), k
    # different inputs through the same graph
    rb2 = args[0].clone()
    rb2[:, 3:6] *= 1.01
    caster.use_graphs = False
    want = {k: v.clone() for k, v in caster(rb2, **kwargs).items()}
    caster.use_graphs = True
    got = caster(rb2, **kwargs)
    assert len(caster._graphs.graphs) == 1 and all(torch.equal(want[k], got[k]) for k in want)
    # a weight update invalidates the captured pointers
    with torch.no_grad():
        caster.network.alpha_linear.bias.add_(0.5)
    caster.use_graphs = False
    want = {k: v.clone() for k, v in caster(*args, **kwargs).items()}
    caster.use_graphs = True
    got = caster(*args, **kwargs)
    assert all(torch.equal(want[k], got[k]) for k in want) and not torch.equal(want["acc_map"], eager["acc_map"])


@pytest.mark.parametrize("cfg_file,fixture", [("h36m_zju/danbo_base.txt", "danbo_stages"), ("surreal/danbo_fast.txt", "danbo_surreal")])
def test_render_of_a_whole_image_equals_the_chunk_loop_bitwise(cfg_file, fixture):
    """core.trainer.render(..., chunk=c, rays=...) as run_render.py / render_path call it (reference run_nerf.py:64-91,
    core/trainer.py:96-161): the caster takes the whole ray set in ONE cast (RayCaster.render_rays_whole) -- every output
    bit-identical to the reference-shaped loop of chunk-sized casts, incl. the chunk-wide nan-mean of the cylinder bounds (the
    SURREAL frame has rays that miss the cylinder; chunk sizes that do and do not divide the ray count)"""
    from core import trainer
    from core.utils import synthetic as syn
    g = golden(fixture)
    rest_scale = syn.model_config(str(g["cfg_name"]))["rest_scale"]
    caster, kw = build(cfg_file, g, rest_scale=rest_scale)
    scene = syn.make_scene(n_poses=1, H=64, W=64, n_views=3, pose_seed=2, rest_scale=rest_scale, cam_dist=5.0)
    ro, rd = (T(x) for x in scene["rays"][1])
    n = len(ro)
    exp = lambda x, dt=torch.float32: T(x, dt)[:1].expand(n, *x.shape[1:])  # noqa: E731
    kwargs = dict(kp_batch=exp(scene["kps"]), skts=exp(scene["skts"]), cyls=exp(scene["cyls"]), bones=exp(scene["bones"]),
                  cams=torch.zeros(1, dtype=torch.int64, device=DEV).expand(n), ray_caster=caster, N_samples=24, N_importance=12, **kw)
    for chunk in (1024, 1000):
        whole = trainer.render(64, 64, 80., chunk=chunk, rays=(ro, rd), **kwargs)
        calls = []
        orig = caster.render_rays_whole
        caster.render_rays_whole = lambda *a, **k: calls.append(1)       # returns None: batchify_rays falls back to its loop
        try:
            loop = trainer.render(64, 64, 80., chunk=chunk, rays=(ro, rd), **kwargs)
        finally:
            caster.render_rays_whole = orig
        assert calls and set(whole) == set(loop)
        for k in loop:
            assert whole[k].shape == loop[k].shape and torch.equal(whole[k], loop[k]), (chunk, k)
        # the cast is bounded: with a budget of two chunks' worth of ray-samples the image goes through as super-chunks
        # of whole chunks -- the same bits
        caster.whole_cast_max_samples = 2 * chunk * 36
        try:
            capped = trainer.render(64, 64, 80., chunk=chunk, rays=(ro, rd), **kwargs)
        finally:
            del caster.whole_cast_max_samples
        assert all(torch.equal(capped[k], loop[k]) for k in loop), chunk
    # per-ray pose tensors that are not ONE expanded row (several poses may hide behind them): the caster declines, the loop runs
    kwargs["skts"] = T(np.repeat(scene["skts"], n, 0))
    rb = torch.cat([ro, rd, torch.zeros(n, 1, device=DEV), torch.ones(n, 1, device=DEV)], 1)
    assert caster.render_rays_whole(rb, 1024, **{k: v for k, v in kwargs.items() if k != "ray_caster"}) is None
    again = trainer.render(64, 64, 80., chunk=1024, rays=(ro, rd), **kwargs)
    whole = trainer.render(64, 64, 80., chunk=1024, rays=(ro, rd), **dict(kwargs, skts=exp(scene["skts"])))
    assert all(torch.equal(again[k], whole[k]) for k in whole)


def test_render_of_a_megapixel_image_is_bounded_and_equals_the_chunk_loop_bitwise():
    """1000 x 1000 rays x (48 + 16) through core.trainer.render: one cast at the default budget (64 M ray-samples <= 1 << 26), four
    super-chunks at a quarter of it, and the reference-shaped loop of 245 casts of 4 096 rays -- all bit-identical (the
    whole-image cast must not grow without limit, and the identity was only checked at 64 x 64)"""
    from core import trainer
    from core.utils import synthetic as syn
    g = golden("danbo_stages")
    rest_scale = syn.model_config(str(g["cfg_name"]))["rest_scale"]
    caster, kw = build("h36m_zju/danbo_base.txt", g, rest_scale=rest_scale)
    H = W = 1000
    scene = syn.make_scene(n_poses=1, H=H, W=W, n_views=2, pose_seed=5, rest_scale=rest_scale, cam_dist=3.5)
    ro, rd = (T(x) for x in scene["rays"][1])
    n = len(ro)
    assert n == H * W
    exp = lambda x, dt=torch.float32: T(x, dt)[:1].expand(n, *x.shape[1:])  # noqa: E731
    kwargs = dict(kp_batch=exp(scene["kps"]), skts=exp(scene["skts"]), cyls=exp(scene["cyls"]), bones=exp(scene["bones"]),
                  cams=torch.zeros(1, dtype=torch.int64, device=DEV).expand(n), ray_caster=caster, N_samples=48, N_importance=16, **kw)
    whole = trainer.render(H, W, 800., chunk=4096, rays=(ro, rd), **kwargs)
    assert float(whole["acc_map"].max()) > 0.5
    caster.whole_cast_max_samples = (1 << 26) // 4
    try:
        capped = trainer.render(H, W, 800., chunk=4096, rays=(ro, rd), **kwargs)
    finally:
        del caster.whole_cast_max_samples
    orig = caster.render_rays_whole
    caster.render_rays_whole = lambda *a, **k: None
    try:
        loop = trainer.render(H, W, 800., chunk=4096, rays=(ro, rd), **kwargs)
    finally:
        caster.render_rays_whole = orig
    for k in loop:
        assert torch.equal(whole[k], loop[k]) and torch.equal(capped[k], loop[k]), k


@pytest.mark.gpu
def test_eager_encoder_helpers_match_the_reference_formulas():
    """transform_batch_pts / transform_batch_rays (reference encoders.py:288-318) and Optcodes.forward (embedding.py:17-39) called on
    their own: stand-alone kernels of the library (csrc/k_encoders.hip), checked against the formulas in float64"""
    from core import encoders
    from core.networks.embedding import Optcodes
    rng = np.random.default_rng(11)
    N, S, J = 37, 5, 24
    pts = rng.normal(size=(N, S, 3)).astype(np.float32)
    skt = rng.normal(size=(N, J, 4, 4)).astype(np.float32)
    skt[..., 3, :] = (0, 0, 0, 1)
    hom = np.concatenate([pts, np.ones((N, S, 1), np.float32)], -1).astype(np.float64)
    want = np.einsum("njab,nsb->nsja", skt.astype(np.float64), hom)[..., :3]
    got = encoders.transform_batch_pts(torch.tensor(pts, device=DEV), torch.tensor(skt, device=DEV)).cpu().numpy()
    assert got.shape == (N, S, J, 3) and np.abs(got - want).max() < 2e-6 * np.abs(want).max()
    # one pose behind all rays (the reference expands skt)
    got1 = encoders.transform_batch_pts(torch.tensor(pts, device=DEV), torch.tensor(skt[:1], device=DEV)).cpu().numpy()
    want1 = np.einsum("jab,nsb->nsja", skt[0].astype(np.float64), hom)[..., :3]
    assert np.abs(got1 - want1).max() < 2e-6 * np.abs(want1).max()
    wantr = np.einsum("njab,nsb->nsja", skt[..., :3, :3].astype(np.float64), pts.astype(np.float64))
    gotr = encoders.transform_batch_rays(None, torch.tensor(pts, device=DEV), torch.tensor(skt, device=DEV)).cpu().numpy()
    assert np.abs(gotr - wantr).max() < 2e-6 * np.abs(wantr).max()
    with pytest.raises(RuntimeError):
        encoders.transform_batch_pts(torch.tensor(pts), torch.tensor(skt))            # no CPU path
    # Optcodes
    oc = Optcodes(9, 16).to(DEV)
    w = oc.codes.weight.detach().cpu().numpy().astype(np.float64)
    oc.eval()
    idx = torch.tensor(rng.integers(0, 9, size=(20, 1)), device=DEV)
    assert np.array_equal(oc(idx).cpu().numpy(), w[idx.cpu().numpy()[:, 0]].astype(np.float32))
    mean = oc(torch.full((7, 1), -1, device=DEV)).cpu().numpy()
    assert mean.shape == (7, 16) and np.abs(mean - w.mean(0)).max() < 1e-7
    mix = np.concatenate([rng.integers(0, 9, size=(11, 2)).astype(np.float32), rng.uniform(0, 1, size=(11, 1)).astype(np.float32)], 1)
    got = oc(torch.tensor(mix, device=DEV)).cpu().numpy()
    a, b, t = w[mix[:, 0].astype(int)], w[mix[:, 1].astype(int)], mix[:, 2:3].astype(np.float64)
    assert np.abs(got - (a + t * (b - a))).max() < 1e-6
    assert np.array_equal(oc(torch.full((3, 1), 12, device=DEV)).cpu().numpy(), np.repeat(w[8:9], 3, 0).astype(np.float32))   # clamped
    oc.train()
    with pytest.raises(RuntimeError):
        oc(idx)                                                                        # forward-only: training goes through the fused step
    with torch.no_grad():
        assert np.array_equal(oc(idx).cpu().numpy(), w[idx.cpu().numpy()[:, 0]].astype(np.float32))
