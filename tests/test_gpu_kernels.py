"""GPU parity tests: every HIP kernel (called through the C ABI) against the numpy oracle
and the committed golden vectors.  Run on the MI355X box with `pytest -m gpu`.

Tolerances: the in-volume mask, sample positions and merge order are bit-exact; everything
behind a transcendental or a GEMM is fp32 round-off level, with the 1e-4 relative bound of
BASELINE.json's north_star as the outer limit for RGB / sigma logits.
"""
import numpy as np
import pytest
import torch

import danbo_oracle as o
from helpers import golden, oracle_for, max_err, rel_err, raw_err

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(x, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(x), dtype=dtype, device=DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from core import hip_ops
    return hip_ops


@pytest.fixture(scope="module")
def stage():
    """golden stage fixture + oracle + engine with the same seeded weights"""
    from core.render_engine import DanboEngine
    g = golden("danbo_stages")
    orc, cfg, sd, rest = oracle_for(g)
    params = {k: T(v) for k, v in sd.items()}
    eng = DanboEngine(cfg, params, T(orc.align))
    eng.refresh()   # packed weights / tables exist even when a test uses the engine's buffers directly
    pose = g["pose_of_ray"]
    ret = orc.render(g["ray_batch"], g["skts"][pose], g["bones"][pose], g["cyls"][pose], cam_idxs=g["cam_idx"],
                     n_uniques=2, N_samples=int(g["N_samples"]), N_importance=int(g["N_importance"]), stages=True)
    return dict(g=g, orc=orc, cfg=cfg, sd=sd, eng=eng, ret=ret)


def test_library_and_device(ops):
    import ctypes
    from core import _hip
    l = _hip.lib()
    assert l.danbo_abi_version() == 9
    cu, lds = ctypes.c_int(), ctypes.c_int()
    arch = ctypes.create_string_buffer(64)
    assert l.danbo_device_info(ctypes.byref(cu), ctypes.byref(lds), arch, 64) == 0
    assert arch.value.decode().startswith("gfx950"), arch.value
    assert cu.value == 256


def test_coarse_samples_bit_exact(ops, stage):
    g = stage["g"]
    z = ops.coarse_samples(T(g["near"][:, 0]), T(g["far"][:, 0]), int(g["N_samples"]))
    assert np.array_equal(N(z), g["z_coarse"])


def test_cylinder_near_far(ops, stage):
    g = stage["g"]
    rb = g["ray_batch"]
    near, far = ops.near_far_cylinder(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["cyls"]), 0.0, 1.0, 4096)
    assert max_err(N(near), g["near"][:, 0]) < 2e-6
    assert max_err(N(far), g["far"][:, 0]) < 2e-6


def _surreal_inputs():
    from core.utils import synthetic as syn
    g = golden("danbo_surreal")
    orc, cfg, sd, rest = oracle_for(g)
    scene = syn.make_scene(n_poses=1, H=int(g["H"]), W=int(g["W"]), n_views=3, pose_seed=int(g["pose_seed"]),
                           rest_scale=cfg["rest_scale"], cam_dist=float(g["cam_dist"]))
    ro, rd = scene["rays"][int(g["view"])]
    return g, orc, cfg, sd, scene, ro, rd


def test_cylinder_nan_backfill_and_box_near_far(ops):
    from core.render_engine import DanboEngine
    g, orc, cfg, sd, scene, ro, rd = _surreal_inputs()
    near, far = ops.near_far_cylinder(T(ro), T(rd), T(scene["cyls"]), 0.0, 1.0, 4096)
    assert max_err(N(near), g["cyl_near"][:, 0]) < 2e-6 and max_err(N(far), g["cyl_far"][:, 0]) < 2e-6
    # rays that hit the cylinder: bit-equal to the reference (only the back-filled mean is summed in another order)
    vals, counts = np.unique(g["cyl_near"][:, 0], return_counts=True)
    hit = g["cyl_near"][:, 0] != vals[np.argmax(counts)]           # the back-fill value is the one that repeats
    assert 1000 < hit.sum() < len(ro) and counts.max() > 50
    assert np.array_equal(N(near)[hit], g["cyl_near"][hit, 0]) and np.array_equal(N(far)[hit], g["cyl_far"][hit, 0])
    # two chunks: each half gets its own nan-mean, exactly like two calls of the reference
    n2, f2 = ops.near_far_cylinder(T(ro), T(rd), T(scene["cyls"]), 0.0, 1.0, 2048)
    z = np.zeros(len(ro), dtype=np.int64)
    rb = np.concatenate([ro, rd, np.zeros((len(ro), 1), np.float32), np.ones((len(ro), 1), np.float32)], -1)
    on, of = o.near_far_cylinder(rb[:, 0:3], rb[:, 3:6], scene["cyls"][z], rb[:, 6:7], rb[:, 7:8], chunk=2048)
    assert max_err(N(n2), on[:, 0]) < 3e-6 and max_err(N(f2), of[:, 0]) < 3e-6
    # ragged chunks of the one-launch kernel (a workgroup per chunk) and a chunk beyond it (the three-launch path, atomics): the
    # same frame tiled 5 x so that chunks of 20 000 rays exist; a miss anywhere is back-filled with ITS chunk's mean
    ro5, rd5 = np.tile(ro, (5, 1)), np.tile(rd, (5, 1))
    rb5 = np.concatenate([ro5, rd5, np.zeros((len(ro5), 1), np.float32), np.ones((len(ro5), 1), np.float32)], -1)
    assert int(np.isnan(o.near_far_cylinder(rb5[:, 0:3], rb5[:, 3:6], scene["cyls"][np.zeros(len(ro5), dtype=np.int64)],
                                            rb5[:, 6:7], rb5[:, 7:8], chunk=10 ** 9)[0]).sum()) == 0
    for chunk in (1000, 1024, 3000, 16384, 20000):
        n5, f5 = ops.near_far_cylinder(T(ro5), T(rd5), T(scene["cyls"]), 0.0, 1.0, chunk)
        on, of = o.near_far_cylinder(rb5[:, 0:3], rb5[:, 3:6], scene["cyls"][np.zeros(len(ro5), dtype=np.int64)], rb5[:, 6:7], rb5[:, 7:8],
                                     chunk=chunk)
        assert max_err(N(n5), on[:, 0]) < 3e-6 and max_err(N(f5), of[:, 0]) < 3e-6, chunk
    eng = DanboEngine(cfg, {k: T(v) for k, v in sd.items()}, T(orc.align))
    nb, fb = eng.near_far(T(ro), T(rd), T(scene["cyls"]), T(scene["skts"]))
    # box bounds: the reference's tensors bit for bit (round 5: torch.norm's fma chain, the float32 bound 1.3f); rays without a box
    # keep the cylinder's bounds, rays that miss the cylinder too the chunk's nan-mean (another summation order: a few ulp)
    boxed = g["near"][:, 0] != g["cyl_near"][:, 0]
    assert boxed.sum() > 1000
    assert np.array_equal(N(nb)[boxed], g["near"][boxed, 0]) and np.array_equal(N(fb)[boxed], g["far"][boxed, 0])
    assert max_err(N(nb), g["near"][:, 0]) < 2e-6 and max_err(N(fb), g["far"][:, 0]) < 2e-6


def _stage_geo(ops, stage, z=None):
    g, eng = stage["g"], stage["eng"]
    eng.refresh()
    rb = g["ray_batch"]
    return ops.Geometry(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), eng.align, eng.axis_scale,
                        z=T(g["z_coarse"] if z is None else z))


def test_bone_cull_mask_bit_exact(ops, stage):
    g = stage["g"]
    geo = _stage_geo(ops, stage)
    bits, lst, cnt = ops.bone_cull(geo, compact=True)
    bits = N(bits).astype(np.uint32).reshape(g["invalid"].shape[:2])
    valid = ((bits[..., None] >> np.arange(24, dtype=np.uint32)) & 1).astype(bool)
    assert np.array_equal(~valid, g["invalid"].astype(bool))          # vs the reference itself
    assert np.array_equal(valid, stage["ret"]["enc"]["valid"])          # vs the oracle
    n = int(N(cnt)[0])
    want = np.nonzero(valid.any(-1).reshape(-1))[0]
    assert n == len(want)
    assert np.array_equal(np.sort(N(lst)[:n]), want)


def test_bone_cull_from_points(ops, stage):
    g, eng = stage["g"], stage["eng"]
    rb = g["ray_batch"]
    geo_p = ops.Geometry(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), eng.align, eng.axis_scale, pts=T(g["pts"]))
    b1, _, _ = ops.bone_cull(geo_p, compact=False)
    b2, _, _ = ops.bone_cull(_stage_geo(ops, stage), compact=False)
    assert torch.equal(b1, b2)


def test_pose_volumes(ops, stage):
    g, eng = stage["g"], stage["eng"]
    vol = eng.volumes(T(g["bones"]))
    assert max_err(N(vol), g["volumes"]) < 2e-5
    assert max_err(N(vol), stage["ret"]["enc"]["volumes"]) < 2e-5


def test_factorised_gather(ops, stage):
    g = stage["g"]
    geo = _stage_geo(ops, stage)
    pf = ops.bone_gather(geo, T(g["volumes"]))
    assert max_err(N(pf).reshape(g["part_feat"].shape), g["part_feat"]) < 5e-6
    # compacted list: rows follow the list order
    bits, lst, cnt = ops.bone_cull(geo, compact=True)
    n = int(N(cnt)[0])
    pfc = ops.bone_gather(geo, T(g["volumes"]), lst, cnt, geo.M)
    want = g["part_feat"].reshape(-1, 24, 15)[N(lst)[:n]]
    assert max_err(N(pfc)[:n], want) < 5e-6


def test_assign_blend_unfused_and_fused(ops, stage):
    g, eng, ret = stage["g"], stage["eng"], stage["ret"]
    geo = _stage_geo(ops, stage)
    bits, _, _ = ops.bone_cull(geo, compact=False)
    h, confd = ops.assign_blend(T(g["part_feat"].reshape(-1, 24, 15)), bits, eng.aw, want_confd=True)
    assert max_err(N(confd).reshape(g["confd"].shape), g["confd"]) < 2e-5
    h_ref = ret["enc"]["h"]
    assert max_err(N(h)[:, :15], h_ref) < 5e-6
    assert float(N(h)[:, 15].max()) == 0.0
    h2, confd2 = ops.gather_assign_blend(geo, T(g["volumes"]), bits, eng.aw, want_confd=True)
    assert max_err(N(h2)[:, :15], h_ref) < 5e-6
    assert max_err(N(confd2).reshape(g["confd"].shape), g["confd"]) < 2e-5
    # fast variant: fp16 hi/lo split MFMA, adjacency folded into the layer-0 weights
    h3, confd3 = ops.gather_assign_blend16(geo, T(g["volumes"]), bits, eng.aw, eng.assign16, want_confd=True)
    assert max_err(N(h3)[:, :15], h_ref) < 5e-6
    assert float(N(h3)[:, 15].max()) == 0.0
    assert max_err(N(confd3).reshape(g["confd"].shape), g["confd"]) < 2e-5
    # compacted rows, tail tile
    _, lst, cnt = ops.bone_cull(geo, compact=True)
    n = int(N(cnt)[0])
    h4, _ = ops.gather_assign_blend16(geo, T(g["volumes"]), bits, eng.aw, eng.assign16, lst, cnt, geo.M)
    assert max_err(N(h4)[:n, :15], h_ref[N(lst)[:n]]) < 5e-6


def test_view_constants_and_empty_raw(ops, stage):
    g, eng, orc, sd = stage["g"], stage["eng"], stage["orc"], stage["sd"]
    rb = g["ray_batch"]
    cview, raw_empty = eng.view_constants(T(rb[:, 3:6]), T(g["skts"]), T(g["cam_idx"], torch.int64))
    vin = g["view_inputs"]                                   # [R,155] from the reference
    # in the default (fp16-split) mode the bias also carries W_v[:, :256] b_feature (merged layers)
    vb = sd["views_linears.0.bias"] + sd["views_linears.0.weight"][:, :256] @ sd["feature_linear.bias"]
    # (and the exact power-of-two factor of the view layer in the engine's re-parametrisation, DanboEngine._equalized)
    rv = float(eng.view_scale)
    assert rv > 0 and np.log2(rv) == round(np.log2(rv))
    assert max_err(N(eng.views_b16), vb * rv) < 2e-6 * rv
    want = (vin @ sd["views_linears.0.weight"][:, 256:].T + vb) * rv
    assert max_err(N(cview), want) < 5e-6 * rv
    # empty-space raw: MLP on PE(0) with this ray's view vector
    dens0 = o.positional_encoding(np.zeros((1, 15), np.float32), 6)
    want_raw = o.mlp({k: np.asarray(v) for k, v in sd.items()}, np.repeat(dens0, len(vin), 0), vin)
    assert raw_err(N(raw_empty), want_raw) < 1e-4


def _packed_form(ops, eng, form):
    """the engine's MLP weights as fragments of K3's 16x16x32 (k_pe_mlp16) / 32x32x16 (k_pe_mlp32) form"""
    if form == eng.mlp_form:
        return eng.packed16
    q = eng._equalized(eng.p) if eng._built_mode[0] == "f16split" else eng.p      # the parameters of the engine's last refresh
    packed, vb = ops.mlp16_pack(eng.pts_w, q["feature_linear.weight"], q["feature_linear.bias"], q["views_linears.0.weight"],
                                q["views_linears.0.bias"], form=form)
    assert torch.equal(vb, eng.views_b16)
    return packed


def test_pe_mlp_on_golden_features(ops, stage):
    """the MLP kernels (exact fp32 MFMA; fp16x2-split MFMA in both forms) on the oracle's blended features"""
    g, eng, ret = stage["g"], stage["eng"], stage["ret"]
    rb = g["ray_batch"]
    S = int(g["N_samples"])
    eng.mlp_mode = "fp32"
    cview, raw_empty = eng.view_constants(T(rb[:, 3:6]), T(g["skts"]), T(g["cam_idx"], torch.int64))
    eng.mlp_mode = "f16split"
    h = np.zeros((ret["enc"]["h"].shape[0], 16), np.float32)
    h[:, :15] = ret["enc"]["h"]
    raw = torch.zeros(len(rb), S, 4, device=DEV)
    ops.pe_mlp(T(h), S, eng.packed, eng.pts_b, eng.alpha_w, eng.alpha_b, eng.feature_b, cview, eng.rgb_w, eng.rgb_b, raw)
    assert raw_err(N(raw), g["raw_coarse"]) < 1e-4      # north_star tolerance vs the reference
    assert raw_err(N(raw), ret["raw_coarse"]) < 1e-4
    cview16 = cview + (eng.views_b16 - eng.views_b)               # merged feature+view layer: bias moves to cview
    split = {}
    for form in (16, 32):
        raw16 = torch.zeros(len(rb), S, 4, device=DEV)
        ops.pe_mlp16(T(h), S, _packed_form(ops, eng, form), eng.pts_b, eng.alpha_w, eng.alpha_b, cview16, eng.rgb_w, eng.rgb_b, raw16, form=form)
        assert raw_err(N(raw16), g["raw_coarse"]) < 1e-4, form
        assert raw_err(N(raw16), N(raw)) < 2e-5, form      # split products ~ fp32 round-off class
        split[form] = raw16
    # the two forms add the same products in another order: the same round-off class
    assert raw_err(N(split[32]), N(split[16])) < 2e-5


def test_pe_mlp16_random_rows_and_tails(ops, stage):
    """row counts that are not multiples of the 128-row tile, compacted scatter, large |h|"""
    eng = stage["eng"]
    rng = np.random.default_rng(5)
    for n in (1, 31, 33, 129, 1000, 4133):
        h = np.zeros((n, 16), np.float32)
        h[:, :15] = rng.normal(0, 1.5, size=(n, 15))
        S = 4
        R = (n + S - 1) // S
        cview = T(rng.normal(0, 0.3, size=(R, 128)))
        lst = T(rng.permutation(R * S)[:n], torch.int32)
        a = torch.zeros(R * S, 4, device=DEV)
        b = torch.zeros(R * S, 4, device=DEV)
        ops.pe_mlp(T(h), S, eng.packed, eng.pts_b, eng.alpha_w, eng.alpha_b, eng.feature_b, cview, eng.rgb_w, eng.rgb_b,
                   a, lst=lst)
        for form in (16, 32):
            b.zero_()
            ops.pe_mlp16(T(h), S, _packed_form(ops, eng, form), eng.pts_b, eng.alpha_w, eng.alpha_b, cview + (eng.views_b16 - eng.views_b),
                         eng.rgb_w, eng.rgb_b, b, lst=lst, form=form)
            assert raw_err(N(b), N(a)) < 2e-5, (n, form)
            # rows outside the list stay untouched
            rest = torch.ones(R * S, dtype=torch.bool, device=DEV)
            rest[lst.long()] = False
            assert float(b[rest].abs().max()) == 0.0 if bool(rest.any()) else True
        assert float(N(a).__abs__().max()) > 0.1


@pytest.mark.parametrize("mode", ["f16split", "f16split-16", "fp32"])
def test_forward_dense_equals_culled_bitwise(stage, mode):
    g, eng = stage["g"], stage["eng"]
    eng.mlp_mode, eng.mlp_form = mode.split("-")[0], 16 if mode.endswith("-16") else 32
    rb = g["ray_batch"]
    args = (T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), T(g["bones"]), T(g["cam_idx"], torch.int64))
    raw_c, ex = eng.forward_samples(*args, z=T(g["z_coarse"]), dense=False)
    raw_d, _ = eng.forward_samples(*args, z=T(g["z_coarse"]), dense=True)
    assert torch.equal(raw_c, raw_d)
    assert raw_err(N(raw_c), g["raw_coarse"]) < 1e-4
    n = int(N(ex["count"])[0])
    assert 0 < n < raw_c.shape[0] * raw_c.shape[1]
    eng.mlp_mode, eng.mlp_form = "f16split", 32


def test_composite(ops, stage):
    g = stage["g"]
    rb = g["ray_batch"]
    out = ops.composite(T(g["raw_coarse"]), T(g["z_coarse"]), T(rb[:, 3:6]), 1.0)
    assert max_err(N(out["weights"]), g["weights_coarse"]) < 2e-6
    assert max_err(N(out["alpha"]), g["alpha_coarse"]) < 2e-6
    assert max_err(N(out["rgb_map"]), g["rgb_coarse"]) < 2e-6
    assert max_err(N(out["acc_map"]), g["final_acc0"]) < 2e-6
    assert raw_err(N(out["disp_map"]), g["final_disp0"]) < 1e-5


def test_composite_long_rays_and_noise(ops):
    rng = np.random.default_rng(0)
    R, S = 37, 144                                           # > 64 samples: multi-chunk scan
    raw = rng.normal(0, 2, size=(R, S, 4)).astype(np.float32)
    z = np.sort(rng.uniform(2, 5, size=(R, S)).astype(np.float32), -1)
    d = rng.normal(size=(R, 3)).astype(np.float32)
    noise = rng.normal(size=(R, S)).astype(np.float32)
    out = ops.composite(T(raw), T(z), T(d), 0.5, T(noise))
    ref = o.composite(raw, z, d, 0.5, noise)
    for k in ("weights", "alpha", "rgb_map", "acc_map"):
        assert max_err(N(out[k]), ref[k]) < 5e-6, k
    assert raw_err(N(out["disp_map"]), ref["disp_map"]) < 1e-5


def test_importance_samples_and_merge(ops, stage):
    g = stage["g"]
    Sf = int(g["N_importance"])
    zs, zf, idx = ops.importance_samples(T(g["z_coarse"]), T(g["weights_coarse"]), Sf)
    assert max_err(N(zf), g["z_fine"]) < 5e-6  # cdf summation order
    assert max_err(N(zs), g["z_sorted"]) < 5e-6
    assert np.array_equal(N(idx).astype(np.int64), g["sorted_idxs"])
    a = T(g["raw_coarse"])
    b = torch.randn(a.shape[0], Sf, 4, device=DEV)
    m = ops.merge_samples(a, b, idx)
    want = np.take_along_axis(np.concatenate([g["raw_coarse"], N(b)], 1), g["sorted_idxs"][..., None], 1)
    assert np.array_equal(N(m), want)


@pytest.mark.parametrize("S,Sf", [(48, 16), (32, 16), (96, 48), (7, 3), (200, 64), (272, 24), (80, 72)])
def test_importance_random_uniforms_and_long_rays(ops, S, Sf):
    """wave-per-ray kernel (S <= 64), its long-ray form (64 < S <= 256, Sf <= 64: chunks of 64 through LDS) and the per-thread
    fallback (S > 256 or Sf > 64), deterministic and random u"""
    rng = np.random.default_rng(S)
    R = 200
    z = np.sort(rng.uniform(2, 5, size=(R, S)).astype(np.float32), -1)
    w = (rng.uniform(size=(R, S)) ** 4).astype(np.float32)
    for u in (None, rng.uniform(size=(R, Sf)).astype(np.float32)):
        zs, zf, idx = ops.importance_samples(T(z), T(w), Sf, None if u is None else T(u))
        z_all, z_fine, order = o.importance_z(z, w, Sf, u=u)
        # t = (u - c0) / (c1 - c0) is ill-conditioned for the tiny bins of this w ~ U^4 test data
        assert max_err(N(zf), z_fine) < 1e-4
        assert np.mean(np.abs(N(zf) - z_fine)) < 2e-6
        cat = np.concatenate([z, N(zf)], -1)
        assert np.array_equal(N(zs), np.sort(cat, -1))                      # a permutation, sorted
        assert np.array_equal(np.take_along_axis(cat, N(idx).astype(np.int64), -1), N(zs))
        assert np.array_equal(N(idx).astype(np.int64), np.argsort(cat, -1, kind="stable"))


def test_render_stage_fixture_end_to_end(stage):
    g, eng = stage["g"], stage["eng"]
    rb = g["ray_batch"]
    out = eng.render(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), T(g["bones"]), T(g["cyls"]),
                     T(g["cam_idx"], torch.int64), int(g["N_samples"]), int(g["N_importance"]), keep=True)
    assert np.array_equal(N(out["z_coarse"]), g["z_coarse"])
    assert raw_err(N(out["raw_coarse"]), g["raw_coarse"]) < 1e-4
    for k in ("rgb_map", "acc_map", "alpha", "T_i", "rgb0", "acc0", "alpha0"):
        assert max_err(N(out[k]), g["final_" + k]) < 3e-5, k   # (bounds = ~4 x measured, round 5: with bit-exact bounds and the pack scales the maps sit at 1e-6 .. 1e-5 of the reference's)
    assert o.psnr(N(out["rgb_map"]), g["final_rgb_map"]) > 95.0


def test_render_surreal_full_frame(ops):
    """whole 64x64 frame, box near/far, no frame codes: final maps vs the reference"""
    from core.render_engine import DanboEngine
    g, orc, cfg, sd, scene, ro, rd = _surreal_inputs()
    eng = DanboEngine(cfg, {k: T(v) for k, v in sd.items()}, T(orc.align))
    nf = (T(g["near"][:, 0]), T(g["far"][:, 0]))             # the reference's own bounds (see oracle test)
    out = eng.render(T(ro), T(rd), T(scene["skts"]), T(scene["bones"]), T(scene["cyls"]), None,
                     int(g["N_samples"]), int(g["N_importance"]), near_far=nf)
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        assert max_err(N(out[k]), g["final_" + k]) < 2e-4, k            # measured 4.7e-5 (a 4 096-ray frame: more resampled depths at a kink)
    assert o.psnr(N(out["rgb_map"]), g["final_rgb_map"]) > 85.0


def test_render_perfcap_view_branch():
    from core.render_engine import DanboEngine
    g = golden("danbo_perfcap")
    orc, cfg, sd, rest = oracle_for(g)
    eng = DanboEngine(cfg, {k: T(v) for k, v in sd.items()}, T(orc.align))
    rb = g["ray_batch"]
    cam = T(-np.ones(len(rb)), torch.int64)
    cview, _ = eng.view_constants(T(rb[:, 3:6]), T(g["skts"]), cam)
    # (the fast kernels run a power-of-two re-parametrisation of the MLP: the view layer's pre-activations carry eng.view_scale)
    want = g["view_inputs"] @ sd["views_linears.0.weight"][:, 256:].T * float(eng.view_scale) + N(eng.views_b16)
    assert max_err(N(cview), want) < 5e-6
    nf = (T(g["near"][:, 0]), T(g["far"][:, 0]))
    out = eng.render(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), T(g["bones"]), T(g["cyls"]), cam,
                     int(g["N_samples"]), int(g["N_importance"]), near_far=nf, keep=True)
    assert raw_err(N(out["raw_coarse"]), g["raw_coarse"]) < 1e-4
    for k in ("rgb_map", "acc_map", "alpha", "T_i", "rgb0", "acc0"):
        assert max_err(N(out[k]), g["final_" + k]) < 1e-3, k
    assert o.psnr(N(out["rgb_map"]), g["final_rgb_map"]) > 70.0


def test_large_random_batch_against_oracle(ops):
    """8192 rays x 24 samples, 4 poses: mask bit-exact, raw within 1e-4 rel of the oracle"""
    from core.render_engine import DanboEngine
    from core.utils import synthetic as syn
    cfg = syn.model_config("danbo_base")
    rest = syn.rest_pose(cfg["rest_scale"])
    sd = syn.make_state_dict(cfg, seed=3, n_framecodes=10, rest=rest)
    orc = o.DanboOracle(cfg, sd, rest)
    scene = syn.make_scene(n_poses=4, H=64, W=64, n_views=4, pose_seed=40)
    ro = np.concatenate([scene["rays"][p][0][1024:3072] for p in range(4)])
    rd = np.concatenate([scene["rays"][p][1][1024:3072] for p in range(4)])
    pose = np.repeat(np.arange(4), 2048)
    rb = syn.ray_batch(ro, rd)
    near, far = orc.near_far(rb[:, :3], rb[:, 3:6], scene["cyls"][pose], scene["skts"][pose], rb[:, 6:7], rb[:, 7:8])
    z = o.coarse_z(near, far, 24)
    cam = (np.arange(len(ro)) % 10).astype(np.int64)
    raw_ref, enc = orc.forward(o.sample_points(ro, rd, z), rd, scene["skts"][pose], scene["bones"][pose], cam, 4)
    eng = DanboEngine(cfg, {k: T(v) for k, v in sd.items()}, T(orc.align))
    raw, ex = eng.forward_samples(T(ro), T(rd), T(scene["skts"]), T(scene["bones"]), T(cam, torch.int64), z=T(z),
                                  want_confd=True)
    bits = N(ex["valid_bits"]).astype(np.uint32).reshape(z.shape)
    valid = ((bits[..., None] >> np.arange(24, dtype=np.uint32)) & 1).astype(bool)
    assert np.array_equal(valid, enc["valid"])
    assert valid.any(-1).mean() > 0.02
    assert raw_err(N(raw), raw_ref) < 1e-4
    n = int(N(ex["count"])[0])
    rows = N(ex["list"])[:n]
    assert max_err(N(ex["confd_rows"])[:n], enc["confd"].reshape(-1, 24)[rows]) < 5e-5


@pytest.mark.parametrize("S,Sf,rand_u", [(48, 16, False), (12, 6, False), (64, 64, False), (32, 16, True), (5, 3, True)])
def test_fused_composite_importance_and_merged_composite_equal_the_separate_kernels(ops, S, Sf, rand_u):
    """one-launch coarse composite + resampling, and the final composite reading through the sorted order,
    are bit-identical to composite -> importance_samples -> merge_samples -> composite"""
    rng = np.random.default_rng(S * 100 + Sf)
    R = 777
    raw = T(rng.normal(0, 2.0, size=(R, S, 4)))
    raw_f = T(rng.normal(0, 2.0, size=(R, Sf, 4)))
    near = rng.uniform(1, 3, size=(R, 1))
    z = T(near + np.sort(rng.uniform(0, 2, size=(R, S)), -1))
    d = T(rng.normal(size=(R, 3)))
    u = T(rng.uniform(size=(R, Sf))) if rand_u else None
    noise = T(rng.normal(0, 0.2, size=(R, S))) if rand_u else None
    a = ops.composite(raw, z, d, 0.8, noise)
    z_all, z_fine, order = ops.importance_samples(z, a["weights"], Sf, u)
    b, z_all2, z_fine2, order2 = ops.composite_importance(raw, z, d, Sf, 0.8, noise, u)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(z_all, z_all2) and torch.equal(z_fine, z_fine2) and torch.equal(order, order2)
    srt = torch.sort(order.long(), -1).values
    assert bool((srt == torch.arange(S + Sf, device=DEV)).all()) and bool((z_all[:, 1:] >= z_all[:, :-1]).all())
    merged = ops.merge_samples(raw, raw_f, order)
    c = ops.composite(merged, z_all, d, 0.8)
    e = ops.composite_merged(raw, raw_f, order, z_all, d, 0.8, want_raw=True)
    for k in c:
        assert torch.equal(c[k], e[k]), k
    assert torch.equal(e["raw_sorted"], merged)
    # un-filled raw: rows whose in-volume word is 0 hold garbage and must be replaced by the ray's empty-space raw
    bits = T(rng.integers(0, 2, size=(R, S)) * rng.integers(1, 1 << 24, size=(R, S)), torch.int32)
    bits_f = T(rng.integers(0, 2, size=(R, Sf)) * 5, torch.int32)
    empty = T(rng.normal(size=(R, 4)))
    filled = torch.where((bits != 0)[..., None], raw, empty[:, None, :].expand(R, S, 4)).contiguous()
    filled_f = torch.where((bits_f != 0)[..., None], raw_f, empty[:, None, :].expand(R, Sf, 4)).contiguous()
    junk = torch.where((bits != 0)[..., None], raw, torch.full_like(raw, float("nan")))
    junk_f = torch.where((bits_f != 0)[..., None], raw_f, torch.full_like(raw_f, float("nan")))
    f1, zs1, zf1, o1 = ops.composite_importance(filled, z, d, Sf, 0.8, noise, u)
    f2, zs2, zf2, o2 = ops.composite_importance(junk, z, d, Sf, 0.8, noise, u, bits=bits, raw_empty=empty)
    assert all(torch.equal(f1[k], f2[k]) for k in f1) and torch.equal(zs1, zs2) and torch.equal(o1, o2)
    g1 = ops.composite_merged(filled, filled_f, o1, zs1, d, 0.8)
    g2 = ops.composite_merged(junk, junk_f, o1, zs1, d, 0.8, bits_a=bits, bits_b=bits_f, raw_empty=empty)
    assert all(torch.equal(g1[k], g2[k]) for k in g1)


def test_render_lazy_fill_equals_filled(ops, stage):
    """engine.render (no raw pre-fill, fused kernels) == the keep=True path that materialises everything"""
    g = golden("danbo_stages")
    eng = stage["eng"]
    pose = g["pose_of_ray"]
    rb = g["ray_batch"]
    args = (T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), T(g["bones"]), T(g["cyls"]), T(g["cam_idx"], torch.int64))
    a = eng.render(*args, N_samples=12, N_importance=6, keep=False)
    b = eng.render(*args, N_samples=12, N_importance=6, keep=True)
    for k in ("rgb_map", "disp_map", "acc_map", "alpha", "T_i", "rgb0", "disp0", "acc0", "alpha0"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("S,Sf", [(16, 8), (48, 16), (96, 32), (130, 64), (260, 16)])
def test_importance_with_descending_and_mixed_depth_order(ops, S, Sf):
    """a ray looking away from the body has its far bound before its near bound: descending coarse depths.  The merged
    order must still be the stable sort of cat([z, z_fine]) (torch.sort in the reference), for the wavefront kernel,
    the fused composite kernel and the general (S > 64) kernel"""
    rng = np.random.default_rng(S)
    R = 300
    near = rng.uniform(2, 4, size=(R, 1))
    far = near + rng.uniform(0.5, 2, size=(R, 1)) * np.where(np.arange(R)[:, None] % 3 == 0, -1.0, 1.0)   # every third ray descends
    t = np.linspace(0, 1, S)[None]
    z = (near * (1 - t) + far * t).astype(np.float32)
    w = rng.uniform(size=(R, S)).astype(np.float32)
    zs, zf, idx = ops.importance_samples(T(z), T(w), Sf)
    z_all, z_fine, order = o.importance_z(z, w, Sf)
    assert max_err(N(zf), z_fine) < 1e-4
    srt = np.sort(N(idx).astype(np.int64), -1)
    assert np.array_equal(srt, np.broadcast_to(np.arange(S + Sf), srt.shape))          # a permutation, always
    assert bool((N(zs)[:, 1:] >= N(zs)[:, :-1]).all())                                 # and sorted
    both = np.concatenate([z, N(zf)], 1)
    assert np.array_equal(np.take_along_axis(both, N(idx).astype(np.int64), 1), N(zs))
    asc = np.arange(R) % 3 != 0
    assert np.array_equal(N(idx)[asc], order[asc])
    if S <= 64:
        raw = T(rng.normal(0, 1, size=(R, S, 4)))
        d = T(rng.normal(size=(R, 3)))
        a = ops.composite(raw, T(z), d, 1.0)
        zs2, zf2, idx2 = ops.importance_samples(T(z), a["weights"], Sf)
        b, zs3, zf3, idx3 = ops.composite_importance(raw, T(z), d, Sf, 1.0)
        assert torch.equal(idx2, idx3) and torch.equal(zs2, zs3)
        srt = torch.sort(idx3.long(), -1).values
        assert bool((srt == torch.arange(S + Sf, device=DEV)).all())


def test_render_frame_c_entry_point_equals_the_python_chain():
    """danbo_render_frame (one C call, caller-provided workspace) enqueues the same kernels in the same order as
    DanboEngine.render: bit-identical maps; a workspace that is too small is refused"""
    import ctypes
    from core import _hip
    import bench
    eng, inp, _ = bench.build_workload(torch.device(DEV), 0)
    sl = slice(100 * 512, 100 * 512 + 6000)                       # 6000 rays across the body: ragged last chunk of 4096
    args = (inp["rays_o"][sl], inp["rays_d"][sl], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"][sl])
    want = eng.render(*args, 48, 16, chunk=4096)
    got = eng.render_frame_c(*args, 48, 16, chunk=4096)
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(want[k], got[k]), k
    assert float(want["acc_map"].max()) > 0.5
    eng.cfg["use_volume_near_far"] = True
    want, got = eng.render(*args, 32, 16, chunk=4096), eng.render_frame_c(*args, 32, 16, chunk=4096)
    assert all(torch.equal(want[k], got[k]) for k in want)
    eng.cfg["use_volume_near_far"] = False
    # rays of more than 64 coarse samples (BASELINE config 3: 96 + 32): the unfused composite / resampling pair, also behind the call
    for S, Sf in ((96, 32), (200, 64)):
        want, got = eng.render(*args, S, Sf, chunk=4096), eng.render_frame_c(*args, S, Sf, chunk=4096)
        assert all(torch.equal(want[k], got[k]) for k in want), (S, Sf)
    eng.skip_flat_rays = False
    try:
        want, got = eng.render(*args, 96, 32, chunk=4096), eng.render_frame_c(*args, 96, 32, chunk=4096)
    finally:
        eng.skip_flat_rays = True
    assert all(torch.equal(want[k], got[k]) for k in want)
    assert _hip.lib().danbo_render_frame(None, None, 257, 16, None, None, 0, None) == -22
    assert _hip.lib().danbo_render_frame_workspace(6000, 1, 48, 16, 4096, 128) > 6000 * 64 * 16
    assert _hip.lib().danbo_render_frame(None, None, 48, 16, None, None, 0, None) == -22


def test_group_rows_is_a_permutation_that_groups_rows_by_their_lowest_bones(ops):
    """danbo_group_rows (csrc/k_group.hip): inside each window of 16 384 compacted rows the list becomes a permutation of itself,
    ordered by the bin (lowest valid bone, second lowest valid bone or none) of each row's in-volume word; rows beyond the
    device-side count are untouched; the result is a pure function of the input (no atomics: two runs agree)"""
    W = 16384
    rng = np.random.default_rng(0)
    M = 5 * W + 777
    n = 3 * W + 1234                         # three full windows + a partial one; the rest of the list is capacity
    # ~40 distinct bone sets with a skewed frequency, as a frame has them, + a sprinkle of arbitrary words
    sets = np.unique(rng.integers(1, 1 << 24, size=60, dtype=np.int64) & rng.integers(1, 1 << 24, size=60, dtype=np.int64))
    sets = sets[sets != 0][:40]
    bits_np = sets[np.minimum((rng.exponential(6.0, size=M)).astype(np.int64), len(sets) - 1)]
    odd = rng.random(M) < 0.02
    bits_np[odd] = rng.integers(1, 1 << 24, size=int(odd.sum()))
    bits_np = bits_np.astype(np.uint32)
    lst_np = rng.permutation(M).astype(np.int32)             # row -> sample, any order
    bits, cnt = T(bits_np.astype(np.int64), torch.int64).to(torch.int32), T([n], torch.int32)
    lst, lst2 = T(lst_np, torch.int32), T(lst_np, torch.int32)
    ops.group_rows(bits, lst, cnt)
    ops.group_rows(bits, lst2, cnt)
    assert torch.equal(lst, lst2)
    out = N(lst)
    assert np.array_equal(out[n:], lst_np[n:])

    def bins(words):
        words = words.astype(np.int64)
        low = np.array([(int(w) & -int(w)).bit_length() - 1 for w in words])
        rest = words & (words - 1)
        second = np.array([(int(w) & -int(w)).bit_length() - 1 if w else 24 for w in rest])
        return low * 25 + second
    moved = 0
    for w0 in range(0, n, W):
        a, b = out[w0:min(w0 + W, n)], lst_np[w0:min(w0 + W, n)]
        assert np.array_equal(np.sort(a), np.sort(b))                                  # same rows
        key = bins(bits_np[a])
        assert np.all(np.diff(key) >= 0), w0                                           # ordered by bin
        assert np.array_equal(a, b[np.argsort(bins(bits_np[b]), kind="stable")]), w0   # ... and stable inside a bin
        moved += int((a != b).sum())
    assert moved > n // 2


def test_render_is_bitwise_independent_of_the_row_order(stage):
    """the re-ordering in front of K2 must not change a single bit of the frame (a row's result does not depend on the rows it
    shares a wavefront with)"""
    from core.utils import synthetic as syn
    eng = stage["eng"]
    scene = syn.make_scene(n_poses=2, H=96, W=96, n_views=2, pose_seed=4)
    ro = np.concatenate([scene["rays"][0][0], scene["rays"][1][0]])
    rd = np.concatenate([scene["rays"][0][1], scene["rays"][1][1]])
    args = (T(ro), T(rd), T(scene["skts"]), T(scene["bones"]), T(scene["cyls"]), torch.zeros(len(ro), dtype=torch.int64, device=DEV))
    assert eng.group_rows is False
    b = eng.render(*args, 24, 12, keep=True)
    eng.group_rows = True
    try:
        a = eng.render(*args, 24, 12, keep=True)
    finally:
        eng.group_rows = False
    assert int(a["count_coarse"].item()) == int(b["count_coarse"].item()) > 20000      # more than one window of 16 384 rows
    for k in ("rgb_map", "disp_map", "acc_map", "alpha", "T_i", "rgb0", "alpha0", "raw_coarse", "raw_fine", "z_fine"):
        assert torch.equal(a[k], b[k]), k


def test_render_without_resampling_of_rays_that_miss_every_volume_is_bitwise_the_full_render(stage):
    """render() flags the rays that cannot meet a volume (ray mask 0, empty-space density <= 0) in the coarse composite, gives
    them no importance resampling and writes their final maps as constants: every output equals, bit for bit, the render that
    resamples and composites every ray -- and the keep=True render that materialises everything"""
    from core.utils import synthetic as syn
    eng = stage["eng"]
    scene = syn.make_scene(n_poses=2, H=96, W=96, n_views=2, pose_seed=4)
    ro = np.concatenate([scene["rays"][0][0], scene["rays"][1][0]])
    rd = np.concatenate([scene["rays"][0][1], scene["rays"][1][1]])
    args = (T(ro), T(rd), T(scene["skts"]), T(scene["bones"]), T(scene["cyls"]), torch.zeros(len(ro), dtype=torch.int64, device=DEV))
    assert eng.skip_flat_rays is True
    a = eng.render(*args, 24, 12)
    eng.skip_flat_rays = False
    try:
        b = eng.render(*args, 24, 12)
    finally:
        eng.skip_flat_rays = True
    c = eng.render(*args, 24, 12, keep=True)
    for k in ("rgb_map", "disp_map", "acc_map", "alpha", "T_i", "rgb0", "disp0", "acc0", "alpha0"):
        assert torch.equal(a[k], b[k]), k
        assert torch.equal(a[k], c[k]), k
    # the flags themselves: most rays of the frame, none of them with an in-volume sample in either pass
    ops_ = eng_ops(eng)
    eng.refresh()
    near, far = eng.near_far(args[0], args[1], args[4], args[2])
    rm = ops_.ray_bone_mask(args[0], args[1], args[2], eng.align, eng.axis_scale, near, far, want_flat=True)
    R = rm[0].numel()
    assert torch.equal(rm[3] != 0, rm[0] == 0)                      # finite geometry: every ray that misses all volumes
    bits, _, _ = ops_.bone_cull(ops_.Geometry(args[0], args[1], args[2], eng.align, eng.axis_scale, z=c["z_coarse"], ray_mask=rm), True)
    assert torch.equal(bits, c["valid_bits"]) and torch.equal(rm[3] != 0, rm[0] == 0)      # the cull confirms them all
    view = eng.view_constants(args[1], args[2], args[5])
    B = eng.cfg["density_scale"]
    assert eng.flat_rays_ok and float(view[1][:, 3].max()) <= 0      # (the constants need an empty-space density <= 0: this network's is)
    fr = ops_.flat_rays(rm[1], rm[3], 24, 12, want_weights=True)
    n = int(fr["ray_count"].item())
    listed = torch.sort(fr["ray_list"][:n].long()).values
    flat = torch.ones(R, dtype=torch.bool, device=DEV)
    flat[listed] = False
    assert torch.equal(flat, rm[3] != 0) and int(flat.sum()) > R // 2 and len(torch.unique(listed)) == n
    assert not bool((c["valid_bits"].view(R, -1)[flat] != 0).any())
    out0, z_all, z_fine, order = ops_.composite_importance(c["raw_coarse"], c["z_coarse"], args[1], 12, B, bits=bits, raw_empty=view[1], flat=fr)
    assert torch.equal(z_fine[~flat], c["z_fine"][~flat]) and torch.equal(z_all[~flat], c["z_sorted"][~flat]) and torch.equal(order[~flat], c["sorted_idxs"][~flat])
    assert torch.equal(z_fine[flat], near.reshape(-1, 1)[flat].expand(-1, 12))
    for k, k0 in (("rgb_map", "rgb0"), ("disp_map", "disp0"), ("acc_map", "acc0"), ("alpha", "alpha0"), ("weights", "weights_coarse")):
        assert torch.equal(out0[k], c[k0]), k
    bits_f, _, _ = ops_.bone_cull(ops_.Geometry(args[0], args[1], args[2], eng.align, eng.axis_scale, z=z_fine, ray_mask=rm[:3]), True)
    assert not bool((bits_f.view(R, -1)[flat] != 0).any()) and torch.equal(bits_f.view(R, -1)[~flat], eng_bits(eng, args, c["z_fine"]).view(R, -1)[~flat])
    out = ops_.composite_merged(c["raw_coarse"], c["raw_fine"], order, z_all, args[1], B, bits_a=bits, bits_b=bits_f, raw_empty=view[1], flat=fr)
    for k, k0 in (("rgb_map", "rgb_map"), ("disp_map", "disp_map"), ("acc_map", "acc_map"), ("alpha", "alpha"), ("weights", "T_i")):
        assert torch.equal(out[k], c[k0]), k
    # the view constants of the listed rays alone: the same rows
    cv, re_ = eng.view_constants(args[1], args[2], args[5], fr["ray_list"], fr["ray_count"])
    assert torch.equal(cv[~flat], view[0][~flat]) and torch.equal(re_[~flat], view[1][~flat])
    # weights that do not allow it (an empty-space density > 0): the engine says so, and render() evaluates every ray
    sd = eng.p["alpha_linear.bias"]
    old = sd.clone()
    try:
        sd += 5.0
        eng.refresh()
        assert not eng.flat_rays_ok
        d1 = eng.render(*args, 24, 12)
        d2 = eng.render(*args, 24, 12, keep=True)
        for k in ("rgb_map", "disp_map", "acc_map", "alpha", "T_i", "rgb0", "disp0", "acc0", "alpha0"):
            assert torch.equal(d1[k], d2[k]), k
        assert float(d1["acc_map"].min()) > 0          # (empty space is not empty any more)
    finally:
        sd.copy_(old)
        eng.refresh()
    assert eng.flat_rays_ok
    # depths outside the interval the flags were made for: the cull takes the flags back
    rm_far = ops_.ray_bone_mask(args[0], args[1], args[2], eng.align, eng.axis_scale, near + 10.0, far + 10.0, want_flat=True)
    assert int(rm_far[3].sum()) > 0
    bits2, _, _ = ops_.bone_cull(ops_.Geometry(args[0], args[1], args[2], eng.align, eng.axis_scale, z=c["z_coarse"], ray_mask=rm_far), True)
    assert torch.equal(bits2, c["valid_bits"]) and int(rm_far[3].sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("S,Sf", [(24, 12), (96, 32)])
def test_frames_of_only_rays_of_constants_and_of_none(stage, S, Sf):
    """the two ends of the ray list: a batch in which EVERY ray is a ray of constants (empty list, no in-volume row in either pass:
    every per-row kernel is launched on a device-side count of 0) and a batch in which NONE is (the list holds every ray), ragged
    batch sizes (1, 63, 257 rays) included -- each bit-identical to the render that evaluates every ray, in Python and behind the
    C call"""
    from core.utils import synthetic as syn
    eng = stage["eng"]
    scene = syn.make_scene(n_poses=1, H=80, W=80, n_views=1, pose_seed=11)
    ro, rd = T(scene["rays"][0][0]), T(scene["rays"][0][1])
    skts, bones, cyls = T(scene["skts"]), T(scene["bones"]), T(scene["cyls"])
    ops_ = eng_ops(eng)
    eng.refresh()
    near, far = eng.near_far(ro, rd, cyls, skts)
    rm = ops_.ray_bone_mask(ro, rd, skts, eng.align, eng.axis_scale, near, far, want_flat=True)
    flat = rm[3] != 0
    assert int(flat.sum()) > 300 and int((~flat).sum()) > 300
    keys = ("rgb_map", "disp_map", "acc_map", "alpha", "T_i", "rgb0", "disp0", "acc0", "alpha0")
    for sel, n_flat_of in ((flat, lambda n: n), (~flat, lambda n: 0)):
        idx_all = torch.nonzero(sel).reshape(-1)
        for n in (1, 63, 257, len(idx_all)):
            idx = idx_all[:n]
            args = (ro[idx].contiguous(), rd[idx].contiguous(), skts, bones, cyls, torch.zeros(n, dtype=torch.int64, device=DEV))
            nf = (near[idx].contiguous(), far[idx].contiguous())        # the frame's bounds (a sub-batch has another chunk nan-mean)
            a = eng.render(*args, S, Sf, near_far=nf)
            eng.skip_flat_rays = False
            try:
                b = eng.render(*args, S, Sf, near_far=nf)
            finally:
                eng.skip_flat_rays = True
            c = eng.render(*args, S, Sf, near_far=nf, keep=True)
            for k in keys:
                assert torch.equal(a[k], b[k]), (k, n)
                assert torch.equal(a[k], c[k]), (k, n)
            want, got = eng.render(*args, S, Sf), eng.render_frame_c(*args, S, Sf)       # (their own bounds of the sub-batch)
            assert all(torch.equal(want[k], got[k]) for k in want), n
            rm_n = ops_.ray_bone_mask(args[0], args[1], skts, eng.align, eng.axis_scale, nf[0], nf[1], want_flat=True)
            assert int(rm_n[3].sum()) == n_flat_of(n)
            if n_flat_of(n) == n:
                assert int(c["count_coarse"].item()) == 0 and int(c["count_fine"].item()) == 0
                assert float(a["acc_map"].abs().max()) == 0.0 and float(a["alpha"].abs().max()) == 0.0
            elif n == len(idx_all):        # (a candidate bone is not yet an in-volume sample: not asked of the small batches)
                assert int(c["count_coarse"].item()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("S,Sf", [(96, 32), (130, 64), (256, 16)])
def test_rays_of_constants_of_long_rays_are_bitwise_the_full_render(stage, S, Sf):
    """rays of more than 64 coarse samples (BASELINE config 3: 96 + 32) go through the unfused composites; the rays of constants
    skip the raw pre-fill, the coarse composite, the resampling and the final composite there too -- every output equals the
    render that evaluates every ray and the keep=True render, bit for bit; the listed kernels one by one against the full ones"""
    from core.utils import synthetic as syn
    eng = stage["eng"]
    scene = syn.make_scene(n_poses=2, H=64, W=64, n_views=2, pose_seed=7)
    ro = np.concatenate([scene["rays"][0][0], scene["rays"][1][0]])
    rd = np.concatenate([scene["rays"][0][1], scene["rays"][1][1]])
    args = (T(ro), T(rd), T(scene["skts"]), T(scene["bones"]), T(scene["cyls"]), torch.zeros(len(ro), dtype=torch.int64, device=DEV))
    assert eng.skip_flat_rays is True and eng.flat_rays_ok
    a = eng.render(*args, S, Sf)
    eng.skip_flat_rays = False
    try:
        b = eng.render(*args, S, Sf)
    finally:
        eng.skip_flat_rays = True
    c = eng.render(*args, S, Sf, keep=True)
    d = eng.render(*args, S, Sf, dense=True)
    for k in ("rgb_map", "disp_map", "acc_map", "alpha", "T_i", "rgb0", "disp0", "acc0", "alpha0"):
        assert torch.equal(a[k], b[k]), k
        assert torch.equal(a[k], c[k]), k
        assert torch.equal(a[k], d[k]), k
    ops_ = eng_ops(eng)
    near, far = eng.near_far(args[0], args[1], args[4], args[2])
    rm = ops_.ray_bone_mask(args[0], args[1], args[2], eng.align, eng.axis_scale, near, far, want_flat=True)
    R = rm[0].numel()
    fr = ops_.flat_rays(rm[1], rm[3], S, Sf, want_weights=True)
    flat = rm[3] != 0
    assert int(flat.sum()) > R // 2 and int(fr["ray_count"].item()) == R - int(flat.sum())
    view = eng.view_constants(args[1], args[2], args[5])
    B = eng.cfg["density_scale"]
    out0 = ops_.composite(c["raw_coarse"], c["z_coarse"], args[1], B, flat=fr)
    for k, k0 in (("rgb_map", "rgb0"), ("disp_map", "disp0"), ("acc_map", "acc0"), ("alpha", "alpha0"), ("weights", "weights_coarse")):
        assert torch.equal(out0[k], c[k0]), k
    z_all, z_fine, order = ops_.importance_samples(c["z_coarse"], out0["weights"], Sf, flat=fr)
    assert torch.equal(z_fine[~flat], c["z_fine"][~flat]) and torch.equal(z_all[~flat], c["z_sorted"][~flat]) and torch.equal(order[~flat], c["sorted_idxs"][~flat])
    assert torch.equal(z_fine[flat], near.reshape(-1, 1)[flat].expand(-1, Sf))
    out = ops_.composite_merged(c["raw_coarse"], c["raw_fine"], order, z_all, args[1], B, flat=fr)
    for k, k0 in (("rgb_map", "rgb_map"), ("disp_map", "disp_map"), ("acc_map", "acc_map"), ("alpha", "alpha"), ("weights", "T_i")):
        assert torch.equal(out[k], c[k0]), k
    # un-filled raw (rows outside every volume hold garbage: NaN here) + the in-volume words: the same composite, with and without a list
    junk = torch.where((c["valid_bits"].view(R, S, 1) != 0), c["raw_coarse"], torch.full_like(c["raw_coarse"], float("nan")))
    lazy = ops_.composite(junk, c["z_coarse"], args[1], B, bits=c["valid_bits"], raw_empty=view[1])
    fr2 = ops_.flat_rays(rm[1], rm[3], S, Sf, want_weights=True)
    lazy_l = ops_.composite(junk, c["z_coarse"], args[1], B, bits=c["valid_bits"], raw_empty=view[1], flat=fr2)
    for k, k0 in (("rgb_map", "rgb0"), ("disp_map", "disp0"), ("acc_map", "acc0"), ("alpha", "alpha0"), ("weights", "weights_coarse")):
        assert torch.equal(lazy[k], c[k0]), k
        assert torch.equal(lazy_l[k], c[k0]), k


def eng_ops(eng):
    from core import hip_ops
    return hip_ops


def eng_bits(eng, args, z):
    from core import hip_ops
    return hip_ops.bone_cull(hip_ops.Geometry(args[0], args[1], args[2], eng.align, eng.axis_scale, z=z), False)[0]


def test_ray_bone_mask_is_conservative_and_never_changes_the_in_volume_mask(ops, stage):
    """k_ray_bone_mask only tells k_bone_cull what it may skip: with it, without it (the per-window rejection inside the kernel),
    and with an interval that does NOT hold the depths (every sample then falls back to all bones) the in-volume mask and the
    compacted list are the same; every bone a sample of a ray is inside is a candidate of that ray."""
    from core.utils import synthetic as syn
    eng = stage["eng"]
    eng.refresh()
    scene = syn.make_scene(n_poses=2, H=96, W=96, n_views=2, pose_seed=4)
    ro = T(np.concatenate([scene["rays"][0][0], scene["rays"][1][0]]))
    rd = T(np.concatenate([scene["rays"][0][1], scene["rays"][1][1]]))
    skts = T(scene["skts"])
    near, far = eng.near_far(ro, rd, T(scene["cyls"]), skts)
    R = ro.shape[0]
    for S, seed in ((48, None), (16, 0), (5, 1)):              # sorted depths, unsorted draws, fewer than 8 samples per ray
        if seed is None:
            z = ops.coarse_samples(near, far, S)
        else:
            u = torch.rand(R, S, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
            z = near.reshape(R, 1) + (far - near).reshape(R, 1) * u
        mask, lo, hi = ops.ray_bone_mask(ro, rd, skts, eng.align, eng.axis_scale, near, far)
        plain = ops.bone_cull(ops.Geometry(ro, rd, skts, eng.align, eng.axis_scale, z=z), True)
        pts = ops.bone_cull(ops.Geometry(ro, rd, skts, eng.align, eng.axis_scale, pts=(ro[:, None] + rd[:, None] * z[..., None])), True)
        assert torch.equal(plain[0], pts[0])        # (the per-window rejection against no rejection at all)
        n = int(plain[2].item())
        assert n > 1000
        wrong = (mask, lo + 10.0, hi + 10.0)        # no depth lies inside: the kernel must not trust the mask
        tight = ops.ray_bone_mask(ro, rd, skts, eng.align, eng.axis_scale, z.min(1).values, z.max(1).values)
        for rm in ((mask, lo, hi), wrong, tight, (torch.zeros_like(mask), lo + 10.0, hi + 10.0)):
            got = ops.bone_cull(ops.Geometry(ro, rd, skts, eng.align, eng.axis_scale, z=z, ray_mask=rm), True)
            assert torch.equal(got[0], plain[0])
            assert int(got[2].item()) == n
            assert torch.equal(torch.sort(got[1][:n]).values, torch.sort(plain[1][:n]).values)
        per_ray = torch.zeros(R, dtype=torch.int32, device=DEV)
        for s in range(S):
            per_ray |= plain[0].view(R, S)[:, s]
        assert int((per_ray & ~mask).count_nonzero()) == 0
        assert int((mask == 0).sum()) > R // 2       # most rays of the frame miss every volume
    # many poses with few rays each: the matrices no longer fit the workgroup's LDS (the kernel's second variant)
    Rs, G = 960, 96
    many = skts[torch.arange(G, device=DEV) % 2].contiguous()
    sel = torch.arange(Rs, device=DEV) * (R // Rs)
    o_s, d_s, lo_s, hi_s = ro[sel].contiguous(), rd[sel].contiguous(), near.reshape(-1)[sel].contiguous(), far.reshape(-1)[sel].contiguous()
    got = ops.ray_bone_mask(o_s, d_s, many, eng.align, eng.axis_scale, lo_s, hi_s)[0]
    per = Rs // G
    want = torch.cat([ops.ray_bone_mask(o_s[g * per:(g + 1) * per].contiguous(), d_s[g * per:(g + 1) * per].contiguous(),
                                        many[g:g + 1].contiguous(), eng.align, eng.axis_scale,
                                        lo_s[g * per:(g + 1) * per].contiguous(), hi_s[g * per:(g + 1) * per].contiguous())[0]
                      for g in range(G)])
    assert torch.equal(got, want) and int((got != 0).sum()) > 0
