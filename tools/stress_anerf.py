"""dev tool: render one A-NeRF frame (config 5's network, a smaller image) many times and report every frame whose outputs are
not bit-identical to the first one's -- the k_linear16 trunk (row loads and weight fragments in pinned registers, hand-counted
waits) has no atomics in its arithmetic: every frame must be.    python tools/stress_anerf.py [frames] [hw]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
from core.anerf_engine import AnerfEngine  # noqa: E402
from core.utils import synthetic as syn  # noqa: E402
from core.utils.skeleton_utils import bone_align_transforms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 192
dev = torch.device("cuda:0")
cfg = syn.model_config("anerf_base")
rest = syn.rest_pose(cfg["rest_scale"])
sd = syn.make_state_dict(cfg, seed=0, n_framecodes=100, rest=rest)
scene = syn.make_scene(n_poses=1, H=hw, W=hw, n_views=8, pose_seed=0, min_radius=1.25)
ro, rd = scene["rays"][0]
T = lambda x, dt=torch.float32: torch.tensor(np.ascontiguousarray(x), dtype=dt, device=dev)  # noqa: E731
eng = AnerfEngine(cfg, {k: T(v) for k, v in sd.items()}, T(bone_align_transforms(rest)), rows_per_chunk=1 << 19)
args = (T(ro), T(rd), T(scene["skts"]), T(scene["bones"]), T(scene["cyls"]), torch.zeros(len(ro), dtype=torch.int64, device=dev), 48, 16)
ref = {k: v.clone() for k, v in eng.render(*args).items() if torch.is_tensor(v)}
torch.cuda.synchronize()
bad = 0
for i in range(n):
    out = eng.render(*args)
    torch.cuda.synchronize()
    for k, v in ref.items():
        if not torch.equal(out[k], v):
            d = (out[k] - v).abs()
            print("frame", i, k, "differs: max", float(d.max()), "entries", int((d > 0).sum()))
            bad += 1
            break
print("done:", n, "frames of", hw, "x", hw, ",", bad, "differ; acc mean", float(ref["acc_map"].mean()))
