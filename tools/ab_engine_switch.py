"""dev tool: the bench frame with and without one engine switch (default: group_rows -- danbo_group_rows in front of K2), interleaved
blocks in one process.  usage: ab_engine_switch.py [attribute]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import numpy as np, torch
import bench
ATTR = sys.argv[1] if len(sys.argv) > 1 else "group_rows"
eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
for _ in range(100):
    bench.render(eng, inp)
torch.cuda.synchronize()
res = {True: [], False: []}
for rep in range(6):
    for flag in (True, False):
        setattr(eng, ATTR, flag)
        for _ in range(5):
            bench.render(eng, inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            bench.render(eng, inp)
        torch.cuda.synchronize()
        res[flag].append((time.perf_counter() - t0) / 20 * 1e3)
for flag in (True, False):
    print(ATTR, flag, "ms/frame median", round(float(np.median(res[flag])), 4), [round(x, 3) for x in res[flag]])
