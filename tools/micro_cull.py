"""micro-benchmark of k_bone_cull on the bench frame (dev tool): the frame's rays, rays that miss every volume (the kernel's fixed
cost per workgroup), and the copy floor of its 100 MB of depth reads and mask writes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch
import bench
from core import hip_ops as ops

eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
out = eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], 48, 16, keep=True)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


rm = ops.ray_bone_mask(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, out["near"], out["far"])
print("rays", rm[0].numel(), "with a candidate bone", int((rm[0] != 0).sum()), " ray_bone_mask us",
      round(timeit(lambda: ops.ray_bone_mask(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, out["near"], out["far"])), 1))
for name, z in (("coarse 48", out["z_coarse"]), ("fine 16", out["z_fine"])):
    geo_m = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z, ray_mask=rm)
    print(name, " bone_cull + list with the ray mask us", round(timeit(lambda: ops.bone_cull(geo_m, True)), 1))
    none = (torch.zeros_like(rm[0]), rm[1], rm[2])
    geo_0 = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z, ray_mask=none)
    print(name, " every workgroup leaves early (mask 0 everywhere) us", round(timeit(lambda: ops.bone_cull(geo_0, True)), 1))
    full = (torch.full_like(rm[0], (1 << 24) - 1), rm[1], rm[2])
    geo_f = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z, ray_mask=full)
    print(name, " every bone a candidate of every ray us", round(timeit(lambda: ops.bone_cull(geo_f, True)), 1))
    geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z)
    bits, lst, cnt = ops.bone_cull(geo, True)
    print(name, "rows", int(cnt.item()), "of", bits.numel(), " bone_cull + list us", round(timeit(lambda: ops.bone_cull(geo, True)), 1),
          " mask only us", round(timeit(lambda: ops.bone_cull(geo, False)), 1))
    away = ops.Geometry(inp["rays_o"], -inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z)
    print(name, " rays pointing away (every bone rejected per ray) us", round(timeit(lambda: ops.bone_cull(away, True)), 1))
    dst = torch.empty_like(z)
    print(name, " copy of the depths (read + write 4 B per sample) us", round(timeit(lambda: dst.copy_(z)), 1))
