"""micro-benchmark of the assignment kernels on the bench workload (dev tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch, numpy as np
import bench
from core import hip_ops as ops

eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
eng.refresh()
near, far = eng.near_far(inp["rays_o"], inp["rays_d"], inp["cyls"], inp["skts"])
z = ops.coarse_samples(near, far, 48)
vols = eng.volumes(inp["bones"])
geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z)
bits, lst, cnt = ops.bone_cull(geo, True)
n = int(cnt.item())
print("rows", n)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print("assign16        ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lst, cnt, geo.M)))
print("assign16 n-host ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lst, None, n)))
zero = torch.zeros_like(bits)
print("assign16 nobits ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, zero, eng.aw, eng.assign16, lst, None, n)))
print("assign v1       ms", timeit(lambda: ops.gather_assign_blend(geo, vols, bits, eng.aw, lst, None, n)))
# sorted list (spatially coherent rows)
ls = torch.sort(lst[:n]).values.contiguous()
print("assign16 sorted ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, ls, None, n)))
h = torch.zeros(n, 16, device="cuda")
raw = torch.zeros(geo.M, 4, device="cuda")
cview, raw_empty = eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"])
print("mlp16           ms", timeit(lambda: ops.pe_mlp16(h, 48, eng.packed16, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, raw, ls, None, n)))
print("mlp16 cap=M     ms", timeit(lambda: ops.pe_mlp16(h, 48, eng.packed16, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, raw, lst, cnt, geo.M)))

# grouped rows (k_group.hip) and the s_memtime trace of one wavefront
lg = lst.clone()
print("group_rows      ms", timeit(lambda: ops.group_rows(bits, lg, cnt)))
print("assign16 grouped ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lg, cnt, geo.M)))
if "--trace" in sys.argv:
    import ctypes
    from core import _hip
    for name, l_ in (("cull order", lst), ("grouped", lg)):
        buf = torch.zeros(256, dtype=torch.int64, device="cuda")
        _hip.lib().danbo_assign16_set_trace(ctypes.c_void_p(buf.data_ptr()))
        ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, l_, cnt, geo.M)
        torch.cuda.synchronize()
        _hip.lib().danbo_assign16_set_trace(None)
        t = buf.cpu().numpy().reshape(-1, 2)
        t = t[t[:, 1] != 0]
        print("trace", name, "(tag, delta of s_memtime ticks [100 MHz: 10 ns each])")
        prev = None
        line = []
        for tag, tm in t:
            if tag == 0 and line:
                print("   ", " ".join(line)); line = []
            line.append(f"{tag}:{'' if prev is None else tm - prev}")
            prev = tm
        print("   ", " ".join(line))
