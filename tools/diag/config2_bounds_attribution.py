#!/usr/bin/env python3
"""CPU-only attribution of the config-2 full-size parity gap: how far are the host-compiled kernel body's box
bounds (tests/host_emu, the arithmetic of k_box_bounds) from the numpy oracle's on the bench frame's centre rays, and how far do the
oracle's OWN raw logits move when it samples at the kernel's bounds instead of its own?  (checker code: imports oracle/)"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("danbo-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import danbo_oracle as o  # noqa: E402
import torch_cpu  # noqa: E402
from core.utils import synthetic as syn  # noqa: E402

fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))  # noqa: E731


def main(n_rays=8192, S=32, H=512, W=512):
    d = os.path.join(ROOT, "tests", "host_emu")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    emu = ctypes.CDLL(os.path.join(d, "libdanbo_emu.so"))
    cfg = dict(syn.model_config("danbo_base"), use_volume_near_far=True)
    rest = syn.rest_pose(cfg["rest_scale"])
    sd = syn.make_state_dict(cfg, seed=0, n_framecodes=100, rest=rest)
    scene = syn.make_scene(n_poses=1, H=H, W=W, n_views=8, pose_seed=0, min_radius=1.25, cam_dist=3.0)
    ro, rd = scene["rays"][0]
    r0 = (H // 2) * W - n_rays // 2
    sl = slice(r0, r0 + n_rays)
    ro, rd = np.ascontiguousarray(ro[sl]), np.ascontiguousarray(rd[sl])
    model = torch_cpu.DanboTorchCPU(cfg, sd, rest)
    orc = model.np_oracle
    z0 = np.zeros(n_rays, np.int64)
    rb = syn.ray_batch(ro, rd)
    skts, cyls, bones = scene["skts"][z0], scene["cyls"][z0], scene["bones"][z0]
    res = {}
    for a in range(0, n_rays, 4096):
        c = slice(a, a + 4096)
        n_o, f_o = orc.near_far(rb[c, 0:3], rb[c, 3:6], cyls[c], skts[c], rb[c, 6:7], rb[c, 7:8])
        nc, fc = o.near_far_cylinder(rb[c, 0:3], rb[c, 3:6], cyls[c], rb[c, 6:7], rb[c, 7:8])
        nr, fr = np.ascontiguousarray(nc[:, 0]), np.ascontiguousarray(fc[:, 0])
        emu.emu_boxes(fp(np.ascontiguousarray(ro[c])), fp(np.ascontiguousarray(rd[c])), fp(scene["skts"]), fp(orc.align),
                      fp(np.ascontiguousarray(sd["graph_net.axis_scale"])), 4096, 1, fp(nr), fp(fr))
        for k, v in (("n_o", n_o[:, 0]), ("f_o", f_o[:, 0]), ("n_k", nr), ("f_k", fr)):
            res.setdefault(k, []).append(v)
    res = {k: np.concatenate(v) for k, v in res.items()}
    dn, df = np.abs(res["n_o"] - res["n_k"]), np.abs(res["f_o"] - res["f_k"])
    print("near: max %.3e  >2e-6: %d  >1e-5: %d   far: max %.3e  >2e-6: %d >1e-5: %d  of %d rays" %
          (dn.max(), (dn > 2e-6).sum(), (dn > 1e-5).sum(), df.max(), (df > 2e-6).sum(), (df > 1e-5).sum(), n_rays))
    # the oracle's raw at its own depths vs at the kernel's depths
    t = lambda v: torch.tensor(np.ascontiguousarray(v, dtype=np.float32))  # noqa: E731
    vols = model._volumes(scene["bones"][:1])
    raws = {}
    for tag in ("o", "k"):
        out = []
        for a in range(0, n_rays, 2048):
            c = slice(a, a + 2048)
            z = o.coarse_z(res["n_" + tag][c, None], res["f_" + tag][c, None], S)
            pts = t(ro[c])[:, None] + t(rd[c])[:, None] * t(z)[..., None]
            with torch.no_grad():
                out.append(model.forward(pts, t(rd[c]), t(skts[c]), vols, torch.zeros(2048, dtype=torch.long), np.zeros(2048, np.int64)).numpy())
        raws[tag] = np.concatenate(out)
    a, b = raws["k"], raws["o"]
    cmax = np.abs(b).reshape(-1, 4).max(0)
    floored = np.abs(a - b) / np.maximum(np.abs(b), 0.05 * cmax)
    big = np.abs(b) > 0.1 * cmax
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
    print("oracle raw at the kernel's bounds vs at its own: floored-5%% %.3e, un-floored %.3e (sigma %.3e)" %
          (floored.max(), rel[big].max(), rel[..., 3][big[..., 3]].max()))
    worst = np.unravel_index(np.argmax(floored), floored.shape)
    print("worst entry: ray %d sample %d ch %d; |dnear| %.3e |dfar| %.3e of that ray" % (worst[0], worst[1], worst[2], dn[worst[0]], df[worst[0]]))
    # rays with equal bounds only
    same = (dn == 0) & (df == 0)
    print("rays with bit-equal bounds: %d; floored err on those: %.3e" % (same.sum(), floored[same].max() if same.any() else 0))


if __name__ == "__main__":
    main()
