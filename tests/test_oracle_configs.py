"""CPU: the numpy oracle against the reference outputs of the round-2 fixtures -- BASELINE config 2 (H36M danbo_fast, per-bone box
near/far, 32 + 16) and the mesh-density grid.  (The oracle is the on-box checker of the GPU tests; these pin it.)"""
import numpy as np

import danbo_oracle as o
from helpers import golden, max_err, oracle_for, rel_err, raw_err


def test_oracle_on_config2_h36m_danbo_fast():
    g = golden("danbo_h36m_fast")
    orc, cfg, sd, rest = oracle_for(g)
    assert cfg["use_volume_near_far"] is True
    pose, rb = g["pose_of_ray"], g["ray_batch"]
    S, Sf = int(g["N_samples"]), int(g["N_importance"])
    n, f = orc.near_far(rb[:, 0:3], rb[:, 3:6], g["cyls"][pose], g["skts"][pose], rb[:, 6:7], rb[:, 7:8])
    assert max_err(n, g["near"]) < 5e-6 and max_err(f, g["far"]) < 5e-6
    ret = orc.render(rb, g["skts"][pose], g["bones"][pose], g["cyls"][pose], g["cam_idx"], 2, S, Sf, stages=True,
                     near_far=(g["near"], g["far"]))
    assert raw_err(ret["raw_coarse"], g["raw_coarse"]) < 1e-4
    valid = ret["enc"]["valid"]
    assert abs(float(valid.any(-1).mean()) - float(g["in_volume_fraction"])) < 1e-9
    for k in ("rgb_map", "acc_map", "alpha", "T_i", "rgb0", "acc0"):
        assert max_err(ret[k], g["final_" + k]) < 1e-3, k
    assert o.psnr(ret["rgb_map"], g["final_rgb_map"]) > 70.0


def test_torch_cpu_baseline_reproduces_the_reference_maps():
    """oracle/torch_cpu.py (bench.py's timed cpu_baseline) on config 2: the reference's final maps"""
    import torch_cpu
    g = golden("danbo_h36m_fast")
    orc, cfg, sd, rest = oracle_for(g)
    pose, rb = g["pose_of_ray"], g["ray_batch"]
    out = torch_cpu.DanboTorchCPU(cfg, sd, rest).render(rb, g["skts"][pose], g["bones"][pose], g["cyls"][pose], g["cam_idx"], 2,
                                                        int(g["N_samples"]), int(g["N_importance"]))
    for k in ("rgb_map", "acc_map", "rgb0"):
        assert max_err(out[k], g["final_" + k]) < 1e-3, k
    assert o.psnr(out["rgb_map"], g["final_rgb_map"]) > 70.0


def test_oracle_on_mesh_density_grid():
    g = golden("danbo_mesh")
    orc, cfg, sd, rest = oracle_for(g)
    res, radius = int(g["res"]), float(g["radius"])
    t = np.linspace(-radius, radius, res + 1)
    grid = np.stack(np.meshgrid(t, t, t), axis=-1).astype(np.float32)
    pts = (grid.reshape(-1, 3) + g["kps"][0, 0].astype(np.float32)).reshape(-1, 1, 3)
    M = pts.shape[0]
    z = np.zeros(M, dtype=np.int64)
    raw = orc.forward(pts, np.zeros((M, 3), np.float32), g["skts"][z], g["bones"][z], np.zeros(M, np.int64), 1)
    raw = raw[0] if isinstance(raw, tuple) else raw
    dens = raw[..., 3].reshape(res + 1, res + 1, res + 1).transpose(1, 0, 2)
    assert raw_err(dens, g["density"]) < 1e-4


def test_oracle_stochastic_training_branches_on_the_reference_draws():
    """BASELINE config 4's actual sampling settings (perturb = 1, raw_noise_std = 1): the reference's training-mode render with
    its own random draws recorded (oracle/gen_golden.py RecordedDraws -> danbo_perfcap_train_noise.npz).  The oracle on the same
    numbers: stratified depths (ray_utils.py:233-248), density noise (nerf.py:316), random inverse-CDF uniforms (ray_utils.py:171)."""
    g = golden("danbo_perfcap_train_noise")
    orc, cfg, sd, rest = oracle_for(g)
    pose, rb = g["pose_of_ray"], g["ray_batch"]
    S, Sf = int(g["N_samples"]), int(g["N_importance"])
    draws = {k: g["draw/" + k] for k in ("t_rand", "u_rand", "noise_c", "noise_f")}
    assert draws["t_rand"].shape == (len(rb), S) and draws["noise_f"].shape == (len(rb), S + Sf)
    ret = orc.render(rb, g["skts"][pose], g["bones"][pose], g["cyls"][pose], g["cam_idx"], int(g["n_uniques"]), S, Sf, stages=True,
                     draws=draws)
    # stratified depths stay inside their strata and differ from the even ones
    even = o.coarse_z(ret["near"], ret["far"], S)
    assert np.all(np.diff(ret["z_coarse"], axis=1) >= 0) and max_err(ret["z_coarse"], even) > 1e-3
    for k, tol in (("rgb0", 2e-5), ("acc0", 2e-5), ("alpha0", 5e-5), ("rgb_map", 1e-4), ("acc_map", 1e-4)):
        e = max_err(ret[k], g[k])
        print(k, e)
        assert e < tol, (k, e)
    assert o.psnr(ret["rgb_map"], g["rgb_map"]) > 65.0
    # without the draws the same call gives visibly different maps: the fixture does exercise the stochastic branches
    det = orc.render(rb, g["skts"][pose], g["bones"][pose], g["cyls"][pose], g["cam_idx"], int(g["n_uniques"]), S, Sf)
    assert max_err(det["rgb0"], g["rgb0"]) > 1e-3
