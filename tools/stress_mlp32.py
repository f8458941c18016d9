"""stress of K3 in the 32x32x16 form (dev tool): random row counts (partial rounds, wavefronts without rows, ragged last groups),
random scatter lists; every launch twice (bitwise repeatable) and against the 16x16x32 form (round-off class)
    python tools/stress_mlp32.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import numpy as np, torch
import bench
from core import hip_ops as ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
eng, inp, _ = bench.build_workload(dev, 0)
eng.refresh()
q = eng._equalized(eng.p)
packed16, _ = ops.mlp16_pack(eng.pts_w, q["feature_linear.weight"], q["feature_linear.bias"], q["views_linears.0.weight"], q["views_linears.0.bias"], form=16)
assert eng.mlp_form == 32
rng = np.random.default_rng(7)
bad, worst = 0, 0.0
for it in range(iters):
    n = int(rng.choice([1, 31, 32, 33, 127, 128, 129, int(rng.integers(1, 6000)), int(rng.integers(30000, 140000))]))
    S = int(rng.choice([1, 4, 16, 48]))
    R = (n + S - 1) // S
    h = torch.zeros(n, 16, device=dev)
    h[:, :15] = torch.from_numpy(rng.normal(0, 1.0, size=(n, 15)).astype(np.float32)).to(dev)
    cview = torch.from_numpy(rng.normal(0, 0.3, size=(R, 128)).astype(np.float32)).to(dev)
    lst = torch.from_numpy(rng.permutation(R * S)[:n].astype(np.int32)).to(dev)
    out = [torch.full((R * S, 4), -7.0, device=dev) for _ in range(3)]
    for k, (pk, form) in enumerate(((eng.packed16, 32), (eng.packed16, 32), (packed16, 16))):
        ops.pe_mlp16(h, S, pk, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, out[k], lst, None, n, form=form)
    torch.cuda.synchronize()
    rep = bool(torch.equal(out[0], out[1]))
    d = float((out[0] - out[2]).abs().max() / out[2].abs().max())
    worst = max(worst, d)
    untouched = torch.ones(R * S, dtype=torch.bool, device=dev); untouched[lst.long()] = False
    clean = bool((out[0][untouched] == -7.0).all()) if bool(untouched.any()) else True
    if not rep or d > 2e-5 or not clean or bool(torch.isnan(out[0]).any()):
        bad += 1
        print("iteration", it, "n", n, "S", S, "repeatable", rep, "vs 16-form", d, "rows outside the list untouched", clean)
print("done:", iters, "iterations,", bad, "bad; worst distance from the 16x16x32 form (of the output's max)", worst)
