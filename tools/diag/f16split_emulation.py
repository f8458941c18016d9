#!/usr/bin/env python3
"""CPU-only emulation of K3's fp16 hi/lo-split arithmetic on the bench frame's in-volume rows (checker code, imports oracle/):
which part of the split path's distance from float64 comes from where.  Products are formed as the kernel forms them
(hi*hi + hi*lo + lo*hi on fp16 values incl. fp16 subnormals), accumulated EXACTLY (float64), rounded to fp32 per layer; the variants
switch single error sources off.  Usage: python tools/diag/f16split_emulation.py [n_rays]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("danbo-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import danbo_oracle as o  # noqa: E402
import torch_cpu  # noqa: E402
from core.utils import synthetic as syn  # noqa: E402

F = np.float32


def split(x, scale=1.0):
    """x (float32) -> hi, lo as float64 values of fp16 numbers; scale: power of two applied before the split"""
    xs = (x.astype(np.float64) * scale).astype(F)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(F)).astype(F).astype(np.float16)
    return hi.astype(np.float64) / scale, lo.astype(np.float64) / scale


def pow2_scale(w):
    m = np.abs(w).max()
    return 2.0 ** (13 - np.floor(np.log2(m)))         # max |w| -> [2^13, 2^14)


def layer(x, w, mode):
    """x [n,K] float32, w [N,K] float32 -> [n,N] float32 pre-bias, split products with exact accumulation"""
    if mode["exact_operands"]:
        return (x.astype(np.float64) @ w.astype(np.float64).T).astype(F)
    xh, xl = split(x)
    wh, wl = split(w, pow2_scale(w) if mode["scale_w"] else 1.0)
    acc = xh @ wh.T + xh @ wl.T + xl @ wh.T
    if mode["lolo"]:
        acc += xl @ wl.T
    return acc.astype(F)


def pe(h, L):
    out = [h]
    for l in range(L):
        a = h.astype(np.float64) * 2.0 ** l
        out += [np.sin(a).astype(F), np.cos(a).astype(F)]
    return np.concatenate(out, -1)


def mlp(sd, cfg, h, vin_c, mode):
    x0 = pe(h, cfg["multires_voxel"])
    y = x0
    for i in range(cfg["D"]):
        w, b = sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]
        if i == 5:
            pre = (layer(x0, w[:, :x0.shape[1]], mode).astype(np.float64) + layer(y, w[:, x0.shape[1]:], mode).astype(np.float64)).astype(F) \
                if not mode["exact_operands"] else layer(np.concatenate([x0, y], -1), w, mode)
        else:
            pre = layer(y, w, mode)
        y = np.maximum((pre + b).astype(F), 0)
    alpha = (y.astype(np.float64) @ sd["alpha_linear.weight"].astype(np.float64).T).astype(F) + sd["alpha_linear.bias"]
    wv = sd["views_linears.0.weight"]
    wfv = (wv[:, :256].astype(np.float64) @ sd["feature_linear.weight"].astype(np.float64)).astype(F)
    pre = layer(y, wfv, mode)
    hv = np.maximum((pre + vin_c).astype(F), 0)
    rgb = (hv.astype(np.float64) @ sd["rgb_linear.weight"].astype(np.float64).T).astype(F) + sd["rgb_linear.bias"]
    return np.concatenate([rgb, alpha], -1)


def main(n_rays=2048, S=48, H=512, W=512):
    cfg = syn.model_config("danbo_base")
    rest = syn.rest_pose(cfg["rest_scale"])
    sd = syn.make_state_dict(cfg, seed=0, n_framecodes=100, rest=rest)
    scene = syn.make_scene(n_poses=1, H=H, W=W, n_views=8, pose_seed=0, min_radius=1.25, cam_dist=3.0)
    ro, rd = scene["rays"][0]
    r0 = (H // 2) * W - n_rays // 2
    ro, rd = np.ascontiguousarray(ro[r0:r0 + n_rays]), np.ascontiguousarray(rd[r0:r0 + n_rays])
    m64 = torch_cpu.DanboTorchCPU(cfg, sd, rest, dtype=torch.float64)
    orc = m64.np_oracle
    z0 = np.zeros(n_rays, np.int64)
    rb = syn.ray_batch(ro, rd)
    near, far = orc.near_far(rb[:, 0:3], rb[:, 3:6], scene["cyls"][z0], scene["skts"][z0], rb[:, 6:7], rb[:, 7:8])
    z = o.coarse_z(near, far, S)
    pts = o.sample_points(ro, rd, z)
    pts_t = o.bone_local(pts, scene["skts"][z0], orc.align)
    _, valid = o.in_volume(pts_t, sd["graph_net.axis_scale"])
    t64 = lambda v: torch.tensor(np.ascontiguousarray(v)).double()  # noqa: E731
    # float64 chain with hooks: h and the per-ray view constants
    cap = {}
    orig = m64._mlp

    def hook(dens_in, vin):
        cap["dens_in"], cap["vin"] = dens_in.numpy(), vin.numpy()
        return orig(dens_in, vin)
    m64._mlp = hook
    m64.netchunk = 10 ** 9
    with torch.no_grad():
        raw64 = m64.forward(t64(pts), t64(rd), t64(scene["skts"][z0]), m64._volumes(scene["bones"][:1]), torch.zeros(n_rays, dtype=torch.long),
                            np.zeros(n_rays, np.int64), valid=torch.as_tensor(valid)).numpy().reshape(-1, 4)
    rows = np.nonzero(valid.any(-1).reshape(-1))[0]
    h = cap["dens_in"][rows, :15].astype(F)
    vin = cap["vin"][rows]
    sdf = {k: np.asarray(v, F) for k, v in sd.items() if np.asarray(v).dtype != np.int64}
    wv = sdf["views_linears.0.weight"]
    b_eff = sdf["views_linears.0.bias"].astype(np.float64) + wv[:, :256].astype(np.float64) @ sdf["feature_linear.bias"].astype(np.float64)
    vin_c = (vin @ wv[:, 256:].astype(np.float64).T + b_eff).astype(F)
    raw64 = raw64[rows]
    cmax = np.abs(raw64).max(0)
    big = np.abs(raw64) > 0.1 * cmax
    print(f"{len(rows)} in-volume rows of {n_rays} rays; channel max {cmax}")

    def report(tag, raw):
        e = np.abs(raw - raw64)
        fl = e / np.maximum(np.abs(raw64), 0.05 * cmax)
        print(f"{tag:58s} floored-5% {fl.max():.3e} (rgb {fl[:, :3].max():.3e} sigma {fl[:, 3].max():.3e})  un-floored {(e / np.maximum(np.abs(raw64), 1e-30))[big].max():.3e}"
              f"  rms floored {np.sqrt((fl ** 2).mean()):.3e}")
    base = dict(exact_operands=False, scale_w=False, lolo=False)
    report("fp32 operands, exact accumulation (floor of this harness)", mlp(sdf, cfg, h, vin_c, dict(base, exact_operands=True)))
    report("split as the kernel does (weights' lo in fp16 subnormals)", mlp(sdf, cfg, h, vin_c, base))
    report("split, weights scaled to [2^13, 2^14) before the split", mlp(sdf, cfg, h, vin_c, dict(base, scale_w=True)))
    report("split, scaled weights, + lo*lo", mlp(sdf, cfg, h, vin_c, dict(base, scale_w=True, lolo=True)))
    rng = np.random.default_rng(0)
    for amp in (1e-8, 3e-8, 1e-7):     # how much of the budget a difference in K2's output h takes (K2 fast vs exact: 3e-8)
        report(f"fp32 operands, exact accumulation, h + U(-{amp:g}, {amp:g})",
               mlp(sdf, cfg, (h + rng.uniform(-amp, amp, h.shape)).astype(F), vin_c, dict(base, exact_operands=True)))
    print("|h| max", np.abs(h).max(), "rms", np.sqrt((h ** 2).mean()))
    report("split, unscaled weights, + lo*lo", mlp(sdf, cfg, h, vin_c, dict(base, lolo=True)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2048)
