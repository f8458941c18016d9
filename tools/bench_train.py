"""Training-step timing for SURVEY §8(d) config 4: PerfCap `danbo_fast`, N_rand = 3072 rays =
16 poses x 192, 32 + 16 samples, perturb = 1, raw_noise_std = 1, L1 loss, Adam.
`--rays-per-pose 24` gives the per-rank share of an 8-rank job (384 rays).  Prints one JSON line.
    python tools/bench_train.py [--steps 20] [--warmup 5] [--rays-per-pose 192]
"""
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--poses", type=int, default=16)
    ap.add_argument("--rays-per-pose", type=int, default=192)
    ap.add_argument("--path", choices=["fused", "autograd"], default="fused",
                    help="fused: danbo_train_step + danbo_adam_step (HIP); autograd: core/train_path.py (torch autograd + rocBLAS)")
    ap.add_argument("--no-graph", action="store_true", help="fused path without HIP-graph capture of the step")
    ap.add_argument("--sync-stats", action="store_true", help="copy the loss terms to the host after every step (as the reference does)")
    a = ap.parse_args()
    if a.path == "autograd":
        os.environ["DANBO_TRAIN_PATH"] = "autograd"
    from core.config import parse_args
    from core.raycasters import create_raycaster
    from core.trainer import Trainer
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import SMPLSkeleton
    dev = "cuda:0"
    args = parse_args(["--no_reload"], config=os.path.join(ROOT, "danbo-pytorch_amd", "configs", "perfcap", "danbo_fast.txt"))
    rest = syn.rest_pose(0.48)
    da = dict(skel_type=SMPLSkeleton, near=0., far=100., n_views=20, rest_pose=rest, hwf=(128, 128, 160.))
    tr_kw, te_kw, start, grad_vars, opt, _ = create_raycaster(args, da, device=dev)
    caster = tr_kw["ray_caster"]
    sd = syn.make_state_dict(syn.model_config("danbo_perfcap"), 3, 20, rest)
    caster.network.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    trainer = Trainer(args, da, opt, None, tr_kw, te_kw, device=dev)
    scene = syn.make_scene(n_poses=a.poses, H=128, W=128, n_views=a.poses, pose_seed=5)
    rng = np.random.default_rng(0)
    ro, rd, pose = [], [], []
    H = W = 128
    js, is_ = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    sel = np.nonzero(((np.abs(is_ - W / 2) < W * 0.22) & (np.abs(js - H / 2) < H * 0.40)).reshape(-1))[0]
    for p in range(a.poses):
        idx = np.sort(rng.choice(sel, size=a.rays_per_pose, replace=False))
        ro.append(scene["rays"][p][0][idx]); rd.append(scene["rays"][p][1][idx]); pose += [p] * a.rays_per_pose
    pose = np.array(pose)
    t = lambda x, d=torch.float32: torch.tensor(np.ascontiguousarray(x), dtype=d, device=dev)  # noqa: E731
    R = len(pose)
    batch = dict(rays_o=t(np.concatenate(ro)), rays_d=t(np.concatenate(rd)), target_s=t(rng.uniform(size=(R, 3))),
                 bgs=t(rng.uniform(size=(R, 3))), kp3d=t(scene["kps"][pose]), skts=t(scene["skts"][pose]),
                 bones=t(scene["bones"][pose]), cyls=t(scene["cyls"][pose]), cam_idxs=t(pose % 20, torch.int64),
                 N_uniques=a.poses)
    if a.path == "fused":
        assert trainer.fused_engine() is not None, trainer.fused_reason
        trainer.engine.use_graph = not a.no_graph
    sync = a.sync_stats or a.path == "autograd"
    for i in range(a.warmup):
        trainer.train_batch(batch, i=i, global_step=i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.steps):
        loss, stats = trainer.train_batch(batch, i=i, global_step=a.warmup + i, sync_stats=sync)
    e1.record()
    torch.cuda.synchronize()
    if "total_loss" not in stats:
        stats["total_loss"] = float(loss["total_loss"])
    counts = trainer.last_preds["counts"].cpu().tolist() if a.path == "fused" else None
    ms = e0.elapsed_time(e1) / a.steps
    S = args.N_samples + args.N_importance
    print(json.dumps(dict(metric="training ray-samples/s", value=R * S / ms * 1e3, ms_per_step=ms, rays=R, samples_per_ray=S,
                          poses=a.poses, loss=stats["total_loss"], path=a.path, hip_graph=a.path == "fused" and not a.no_graph,
                          in_volume_rows=None if counts is None else counts[5], rows=None if counts is None else counts[4],
                          config="perfcap/danbo_fast (SURVEY 8d config 4)")))


if __name__ == "__main__":
    main()
