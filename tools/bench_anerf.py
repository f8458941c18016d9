"""A-NeRF frame timing for SURVEY §8(d) config 5: anerf_base network, 512 x 512 rays x (48 + 16) samples,
tau = 20 (step 0) or 2000 (converged).  Every sample is evaluated (A-NeRF has no in-volume mask).
    python tools/bench_anerf.py [--steps 3] [--warmup 1] [--tau 20] [--hw 512]
Prints one JSON line."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tau", type=float, default=20.0)
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--rows-per-chunk", type=int, default=1 << 20)
    a = ap.parse_args()
    from core.anerf_engine import AnerfEngine
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import bone_align_transforms
    dev = torch.device("cuda:0")
    cfg = syn.model_config("anerf_base")
    rest = syn.rest_pose(cfg["rest_scale"])
    sd = syn.make_state_dict(cfg, seed=0, n_framecodes=100, rest=rest)
    sd["pe_fn.tau"] = np.array(a.tau, np.float32)
    sd["dirs_pe_fn.tau"] = np.array(a.tau, np.float32)
    scene = syn.make_scene(n_poses=1, H=a.hw, W=a.hw, n_views=8, pose_seed=0, min_radius=1.25)
    ro, rd = scene["rays"][0]
    T = lambda x, dt=torch.float32: torch.tensor(np.ascontiguousarray(x), dtype=dt, device=dev)  # noqa: E731
    eng = AnerfEngine(cfg, {k: T(v) for k, v in sd.items()}, T(bone_align_transforms(rest)), rows_per_chunk=a.rows_per_chunk)
    inp = dict(rays_o=T(ro), rays_d=T(rd), skts=T(scene["skts"]), bones=T(scene["bones"]), cyls=T(scene["cyls"]),
               cam_idx=torch.zeros(len(ro), dtype=torch.int64, device=dev))
    S, Sf = 48, 16
    run = lambda: eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], S, Sf)  # noqa: E731
    for _ in range(a.warmup):
        out = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        out = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    n = len(ro) * (S + Sf)
    W, inc, VW = cfg["W"], 432, cfg["view_W"]
    mac = inc * W + 4 * W * W + (inc + W) * W + 2 * W * W + W + W * VW + 24 * VW + 3 * VW
    print(json.dumps(dict(metric="ray-samples/s", value=n / ms * 1e3, ms_per_frame=ms, rays=len(ro), samples_per_ray=S + Sf,
                          tau=a.tau, executed_mac_per_sample=mac, reference_mac_per_sample=2268000,
                          tflops_executed=n * mac * 2 / ms / 1e9, acc_mean=float(out["acc_map"].mean()),
                          config="h36m_zju/anerf_base (SURVEY 8d config 5), k_linear16 trunk (fp16-split MFMA)")))


if __name__ == "__main__":
    main()
