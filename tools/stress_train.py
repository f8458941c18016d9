"""dev tool: run the fused training step many times with noise / perturb and report the first non-finite gradient tensor"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("danbo-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from helpers import golden  # noqa: E402
from test_gpu_training import batch_of, build_trainer  # noqa: E402

g = golden("danbo_perfcap_train")
extra = [] if "--no-noise" in sys.argv else ["--raw_noise_std", "1.0", "--perturb", "1.0"]
args, caster, trainer, opt = build_trainer(g, extra=extra)
eng = trainer.fused_engine()
eng.use_graph = "--no-graph" not in sys.argv
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
b = batch_of(g)
torch.manual_seed(0)
bad = 0
for i in range(steps):
    if "--no-adam" in sys.argv:
        G = b["N_uniques"]
        pp = caster._per_pose
        trainer.last_preds = eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"], G), pp(b["bones"], G), pp(b["cyls"], G),
                                                  b["cam_idxs"], b["target_s"], b["bgs"], int(g["N_samples"]), int(g["N_importance"]),
                                                  perturb=0. if "--no-noise" in sys.argv else 1., raw_noise_std=0. if "--no-noise" in sys.argv else 1.)
    else:
        trainer.train_batch(b, i=i, global_step=i, sync_stats=False)
    if "--sync" in sys.argv:
        torch.cuda.synchronize()
        print("step", i, "ok", trainer.last_preds["counts"].tolist(), flush=True)
    if not bool(torch.isfinite(eng.flat_g).all()) or not bool(torch.isfinite(eng.flat_p).all()):
        bad += 1
        names = [n for n, p in eng.params.items() if not bool(torch.isfinite(p.grad).all())]
        pn = [n for n, p in eng.params.items() if not bool(torch.isfinite(p.detach()).all())]
        print(f"step {i}: non-finite gradients in {names[:12]} ({len(names)} tensors); non-finite parameters: {len(pn)}")
        out = trainer.last_preds
        print("   loss", out["loss"].tolist(), "counts", out["counts"].tolist(), "rgb finite", bool(torch.isfinite(out["rgb_map"]).all()))
        break
print("done", steps, "steps,", bad, "bad")
