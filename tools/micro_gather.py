"""HBM roofline of the unfused transform + gather stage (K1a cull, K1b gather): achieved GB/s vs algorithmic bytes (dev tool)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch
import bench
from core import hip_ops as ops

eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
eng.refresh()
near, far = eng.near_far(inp["rays_o"], inp["rays_d"], inp["cyls"], inp["skts"])
z = ops.coarse_samples(near, far, 48)
vols = eng.volumes(inp["bones"])
geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

bits, lst, cnt = ops.bone_cull(geo, True)
n = int(cnt.item())
M = geo.M
t_cull = timeit(lambda: ops.bone_cull(geo, True))
# dense gather over a 2M-row window (1440 B/row of output: 2.9 GB) and the compacted rows
rows_dense = torch.arange(0, min(M, 2_000_000), device="cuda", dtype=torch.int32)
t_dense = timeit(lambda: ops.bone_gather(geo, vols, rows_dense, None, rows_dense.shape[0]), 5)
t_comp = timeit(lambda: ops.bone_gather(geo, vols, lst, cnt, n), 5)
out = dict(
    cull=dict(samples=M, ms=t_cull, algorithmic_bytes_per_sample=12, GBps=M * 12 / t_cull / 1e6, samples_per_s=M / t_cull * 1e3),
    gather_dense=dict(rows=int(rows_dense.shape[0]), ms=t_dense, algorithmic_bytes_per_row=1448, GBps=rows_dense.shape[0] * 1448 / t_dense / 1e6,
                      frac_of_8TBps=rows_dense.shape[0] * 1448 / t_dense / 1e6 / 8000),
    gather_compacted=dict(rows=n, ms=t_comp, GBps=n * 1448 / t_comp / 1e6, frac_of_8TBps=n * 1448 / t_comp / 1e6 / 8000))
print(json.dumps(out))
# write-bandwidth ceiling of this GPU for the same byte count (plain fill / copy)
buf = torch.empty(rows_dense.shape[0] * 362, device="cuda")
t_fill = timeit(lambda: buf.fill_(1.0), 5)
src = torch.empty_like(buf)
t_copy = timeit(lambda: buf.copy_(src), 5)
print(json.dumps(dict(fill_GBps=buf.numel() * 4 / t_fill / 1e6, copy_GBps_rw=2 * buf.numel() * 4 / t_copy / 1e6)))
