"""GPU tests of the BASELINE configurations round 1 left without a reference comparison, and of the mesh-density path:
  * config 2 -- H36M `danbo_fast` (configs/h36m_zju/danbo_fast.txt: per-bone box near/far, 32 + 16 samples, world rays, frame
    codes) built through create_raycaster and compared with the reference's caster output (tests/golden/danbo_h36m_fast.npz),
  * RayCaster.render_mesh_density at res = 16 against the reference's grid (tests/golden/danbo_mesh.npz).
Both through the drop-in module surface (core.raycasters), i.e. through ctypes into the C ABI."""
import numpy as np
import pytest
import torch

import danbo_oracle as o
from helpers import golden, max_err, oracle_for, rel_err, raw_err
from test_gpu_modules import build, T, N

pytestmark = pytest.mark.gpu


def test_config2_h36m_danbo_fast_caster_matches_reference():
    g = golden("danbo_h36m_fast")
    caster, kw = build("h36m_zju/danbo_fast.txt", g)
    assert caster.use_volume_near_far is True
    pose, rb = g["pose_of_ray"], g["ray_batch"]
    S, Sf = int(g["N_samples"]), int(g["N_importance"])
    out = caster(T(rb), N_samples=S, kp_batch=T(g["kps"][pose]), skts=T(g["skts"][pose]), cyls=T(g["cyls"][pose]),
                 bones=T(g["bones"][pose]), cams=T(g["cam_idx"], torch.int64), N_importance=Sf, N_uniques=2, **kw)
    # the whole chain incl. our own box bounds: the final maps of the reference's caster
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        assert max_err(N(out[k]), g["final_" + k]) < 1e-5, k            # measured 1.4e-6: the bounds are the reference's, bit for bit
    assert o.psnr(N(out["rgb_map"]), g["final_rgb_map"]) > 100.0
    # stage-wise with the engine: bounds, then raw logits on the reference's own bounds (1 ulp of a bound moves every sample)
    eng = caster._engine()
    near, far = eng.near_far(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["cyls"]), T(g["skts"]), chunk=len(rb))
    # rays that meet a box: the reference's bounds bit for bit; the 48 that keep the cylinder's: <= 1 ulp (the reference's CPU
    # `.pow(0.5)` is Sleef's pow, not sqrt, on 1 % of the rays -- ray_utils.py:318; its GPU form is sqrt)
    nc, fc = o.near_far_cylinder(rb[:, 0:3], rb[:, 3:6], g["cyls"][pose], rb[:, 6:7], rb[:, 7:8])
    boxed = (np.abs(g["near"][:, 0] - nc[:, 0]) > 1e-6) | (np.abs(g["far"][:, 0] - fc[:, 0]) > 1e-6)
    assert boxed.sum() >= 200
    assert np.array_equal(N(near)[boxed], g["near"][boxed, 0]) and np.array_equal(N(far)[boxed], g["far"][boxed, 0])
    assert max_err(N(near), g["near"][:, 0]) < 2.4e-7 and max_err(N(far), g["far"][:, 0]) < 2.4e-7
    nf = (T(g["near"][:, 0]), T(g["far"][:, 0]))
    ret = eng.render(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), T(g["bones"]), T(g["cyls"]), T(g["cam_idx"], torch.int64), S, Sf,
                     near_far=nf, keep=True)
    assert raw_err(N(ret["raw_coarse"]), g["raw_coarse"]) < 1e-4          # north_star tolerance
    frac = float((ret["valid_bits"] != 0).float().mean())
    assert abs(frac - float(g["in_volume_fraction"])) < 1e-6                               # the in-volume mask, sample for sample
    for k in ("rgb_map", "acc_map", "alpha", "T_i", "rgb0", "acc0"):
        assert max_err(N(ret[k]), g["final_" + k]) < 3e-5, k            # measured 7.7e-6
    assert o.psnr(N(ret["rgb_map"]), g["final_rgb_map"]) > 100.0
    # and against the on-box oracle
    orc, cfg, sd, rest = oracle_for(g)
    ref = orc.render(rb, g["skts"][pose], g["bones"][pose], g["cyls"][pose], g["cam_idx"], 2, S, Sf, stages=True,
                     near_far=(g["near"], g["far"]))
    assert raw_err(N(ret["raw_coarse"]), ref["raw_coarse"]) < 1e-4
    # the maps against the on-box oracle (same bounds, same arithmetic order up to the fp16-split products): 1e-5
    for k in ("rgb0", "acc0", "rgb_map", "acc_map"):
        e = max_err(N(ret[k]), ref[k])
        print(f"config 2 {k} vs oracle: {e:.2e}")
        assert e < 1e-5, (k, e)


@pytest.mark.parametrize("case", ["weights_x1e-3", "weights_x1e3", "alternating", "activations_1e5"])
def test_fp16_split_kernels_on_checkpoints_outside_fp16_range(case):
    """The fast kernels split every fp32 operand into two fp16 halves (range 6e-5 .. 65504).  A checkpoint is not bound to the
    seeded generator's well-scaled weights: dense layers 1000x smaller / larger, alternating, or a layer whose bias drives the
    activations to 1e5.  The engine's exact power-of-two re-parametrisation (DanboEngine._equalized) has to keep the fast path
    within the north_star tolerance of the exact-fp32 kernels on the same parameters."""
    from core.render_engine import DanboEngine
    g = golden("danbo_perfcap")
    orc, cfg, sd, rest = oracle_for(g)
    sd = {k: np.array(v, dtype=np.float32) for k, v in sd.items()}
    if case == "activations_1e5":
        # one layer whose outputs are ~1e5 (weights and bias x 6e4), brought back by the next layer.  (Adding a constant 6e4 to the
        # bias instead would make the NETWORK ill-conditioned -- the next layer cancels the offset -- and then the exact-fp32
        # kernel itself is only good to ~1e-3: not a statement about the split.)
        sd["pts_linears.2.weight"] *= 6e4
        sd["pts_linears.2.bias"] *= 6e4
        sd["pts_linears.3.weight"] *= np.float32(1.0 / 6e4)
    else:
        # layer l's weights times f_l, its bias (and, in the skip layer, the columns that see the un-scaled input) times the
        # product of the factors so far: the same function up to the overall factor, which the heads divide out again
        total = 1.0
        for l in range(8):
            f = {"weights_x1e-3": 1e-3, "weights_x1e3": 1e3, "alternating": 1e3 if l % 2 else 1e-3}[case]
            w = sd[f"pts_linears.{l}.weight"].astype(np.float64)
            if l == 5:
                w[:, :195] *= total * f
                w[:, 195:] *= f
            else:
                w *= f
            total *= f
            sd[f"pts_linears.{l}.weight"] = w.astype(np.float32)
            sd[f"pts_linears.{l}.bias"] = (sd[f"pts_linears.{l}.bias"].astype(np.float64) * total).astype(np.float32)
        sd["alpha_linear.weight"] = (sd["alpha_linear.weight"].astype(np.float64) / total).astype(np.float32)
        sd["feature_linear.weight"] = (sd["feature_linear.weight"].astype(np.float64) / total).astype(np.float32)
    assert all(np.isfinite(v).all() for v in sd.values())
    rb = g["ray_batch"]
    cam = T(-np.ones(len(rb)), torch.int64)
    nf = (T(g["near"][:, 0]), T(g["far"][:, 0]))
    outs = {}
    for mode in ("fp32", "f16split"):
        eng = DanboEngine(cfg, {k: T(v) for k, v in sd.items()}, T(orc.align), mlp_mode=mode)
        outs[mode] = eng.render(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), T(g["bones"]), T(g["cyls"]), cam, int(g["N_samples"]),
                                int(g["N_importance"]), near_far=nf, keep=True)
    exact, fast = N(outs["fp32"]["raw_coarse"]), N(outs["f16split"]["raw_coarse"])
    assert np.isfinite(exact).all() and np.isfinite(fast).all()
    assert np.abs(exact[..., 3]).max() > 1e-3 and np.ptp(exact[..., :3]) > 1e-3          # a non-trivial network output
    assert raw_err(fast, exact) < 1e-4
    assert max_err(N(outs["f16split"]["rgb_map"]), N(outs["fp32"]["rgb_map"])) < 1e-5       # measured 6e-7


def test_custom_ops_through_torch_ops_namespace():
    """torch.ops.danbo.composite / bone_gather (core/custom_ops.py): schema + fake-tensor checks of torch.library.opcheck, the
    values of the direct C-ABI wrappers, and gradients that reach `raw` / the volumes through register_autograd"""
    from core import custom_ops, hip_ops as ops      # noqa: F401  (registers the operators)
    rng = np.random.default_rng(3)
    R, S = 40, 24
    raw = T(rng.normal(0, 1.5, size=(R, S, 4))).requires_grad_(True)
    z = T(np.sort(rng.uniform(2, 5, size=(R, S)), -1))
    d = T(rng.normal(size=(R, 3)))
    noise = T(rng.normal(0, 0.3, size=(R, S)))
    torch.library.opcheck(torch.ops.danbo.composite, (raw.detach(), z, d, 0.7, noise), test_utils=("test_schema", "test_faketensor"))
    rgb, disp, acc, w, al = torch.ops.danbo.composite(raw, z, d, 0.7, noise)
    direct = ops.composite(raw.detach(), z, d, 0.7, noise)
    assert torch.equal(rgb, direct["rgb_map"]) and torch.equal(w, direct["weights"]) and torch.equal(acc, direct["acc_map"])
    (rgb.sum() + 0.5 * acc.sum()).backward()
    assert raw.grad is not None and float(raw.grad.abs().max()) > 0 and torch.isfinite(raw.grad).all()
    # bone_gather on the stage fixture: values of K1b, gradient = the adjoint of a linear map (exact directional check)
    g = golden("danbo_stages")
    rb = g["ray_batch"]
    pts = T(g["pts"])
    vols = T(g["volumes"]).requires_grad_(True)
    sc = T(oracle_for(g)[2]["graph_net.axis_scale"])
    geo = ops.Geometry(T(rb[:, 0:3]), T(rb[:, 3:6]), T(g["skts"]), T(g["align"]), sc, pts=pts)
    bits, lst, cnt = ops.bone_cull(geo, True)
    rows = torch.sort(lst[: int(cnt.item())]).values.contiguous()
    args = (vols, sc, pts, T(g["skts"]), T(g["align"]), rows)
    torch.library.opcheck(torch.ops.danbo.bone_gather, tuple(a.detach() for a in args), test_utils=("test_schema", "test_faketensor"))
    pf = torch.ops.danbo.bone_gather(*args)
    assert torch.equal(pf, ops.bone_gather(geo, vols.detach(), rows, None, rows.shape[0]))
    wgt = T(rng.normal(size=tuple(pf.shape)))
    (pf * wgt).sum().backward()
    dv = T(rng.normal(size=tuple(vols.shape)))
    lin = float((torch.ops.danbo.bone_gather(dv, sc, pts, T(g["skts"]), T(g["align"]), rows) * wgt).sum())
    assert abs(float((vols.grad * dv).sum()) - lin) <= 1e-4 * abs(lin)


def test_mesh_density_grid_matches_reference():
    """RayCaster.render_mesh_density (reference raycasters.py:421-453): (res+1)^3 raw densities, evaluated netchunk points at a time"""
    g = golden("danbo_mesh")
    caster, kw = build("h36m_zju/danbo_base.txt", g)
    res = int(g["res"])
    dens = caster(T(g["kps"][:1]), T(g["skts"][:1]), T(g["bones"][:1]), fwd_type="mesh", radius=float(g["radius"]), res=res,
                  netchunk=1000)
    assert tuple(dens.shape) == (res + 1, res + 1, res + 1)
    assert raw_err(N(dens), g["density"]) < 1e-4
    whole = caster(T(g["kps"][:1]), T(g["skts"][:1]), T(g["bones"][:1]), fwd_type="mesh", radius=float(g["radius"]), res=res)
    assert torch.equal(whole, dens)                            # chunking does not change a value
