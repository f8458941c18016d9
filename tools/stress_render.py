"""dev tool: render the bench frame many times and report every frame whose outputs are not bit-identical to the first one's
(the render path has no atomics in its arithmetic: every frame must be)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
args = (inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], 48, 16)
ref = {k: v.clone() for k, v in eng.render(*args).items()}
torch.cuda.synchronize()
bad = 0
for i in range(n):
    out = eng.render(*args)
    torch.cuda.synchronize()
    for k, v in ref.items():
        if not torch.equal(out[k], v):
            d = (out[k] - v).abs()
            rows = torch.nonzero(d.reshape(d.shape[0], -1).sum(1) > 0).reshape(-1)
            print("frame", i, k, "differs: max", float(d.max()), "rays", rows.numel(), rows[:6].tolist())
            bad += 1
            break
print("done:", n, "frames,", bad, "differ")
