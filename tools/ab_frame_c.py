import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch, numpy as np
import bench
eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
args = (inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], 48, 16)
for _ in range(60):
    eng.render(*args); eng.render_frame_c(*args)
torch.cuda.synchronize()
res = {"python chain": [], "danbo_render_frame": []}
for rep in range(6):
    for name, fn in (("python chain", eng.render), ("danbo_render_frame", eng.render_frame_c)):
        for _ in range(5): fn(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn(*args)
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    print(k, "ms/frame median", round(float(np.median(v)), 4), [round(x, 3) for x in v])
