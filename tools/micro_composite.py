"""micro-benchmark of the fused composite kernels on the bench frame (dev tool): the general path against the constant path of
rays without an in-volume sample, and the unfused kernels beside them"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch
import bench
from core import hip_ops as ops

eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
out = eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], 48, 16, keep=True)
cview, raw_empty = eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"])
z, raw, bits = out["z_coarse"], out["raw_coarse"], out["valid_bits"]
R = z.shape[0]
print("rays", R, "rays with >= 1 in-volume coarse sample", int((bits.view(R, 48) != 0).any(1).sum()))


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


ci = lambda b, re: ops.composite_importance(raw, z, inp["rays_d"], 16, 1.0, bits=b, raw_empty=re, want_weights=False)
print("composite_importance, frame's bits          us", timeit(lambda: ci(bits, raw_empty)))
print("composite_importance, all rays empty        us", timeit(lambda: ci(torch.zeros_like(bits), raw_empty)))
print("composite_importance, no ray empty          us", timeit(lambda: ci(torch.ones_like(bits), raw_empty)))
print("composite_importance, dense raw (no bits)   us", timeit(lambda: ci(None, None)))
pos = raw_empty.clone(); pos[:, 3] = 1.0
print("composite_importance, empty density > 0     us", timeit(lambda: ci(bits, pos)))
print("composite (alone)                           us", timeit(lambda: ops.composite(raw, z, inp["rays_d"], 1.0)))
w = ops.composite(raw, z, inp["rays_d"], 1.0)["weights"]
print("importance_samples (alone)                  us", timeit(lambda: ops.importance_samples(z, w, 16)))
cm = lambda ba, bb, re: ops.composite_merged(raw, out["raw_fine"], out["sorted_idxs"], out["z_sorted"], inp["rays_d"], 1.0, bits_a=ba, bits_b=bb, raw_empty=re)
geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=out["z_fine"])
bits_f, _, _ = ops.bone_cull(geo, True)
print("composite_merged, frame's bits              us", timeit(lambda: cm(bits, bits_f, raw_empty)))
print("composite_merged, all empty                 us", timeit(lambda: cm(torch.zeros_like(bits), torch.zeros_like(bits_f), raw_empty)))
print("composite_merged, none empty                us", timeit(lambda: cm(torch.ones_like(bits), torch.ones_like(bits_f), raw_empty)))
print("composite_merged, dense                     us", timeit(lambda: cm(None, None, None)))
# rays that cannot meet a volume (ray_bone_mask) skip the resampling
rm = ops.ray_bone_mask(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, out["near"], out["far"], want_flat=True)
print("empty-space density pre-activation: min", float(raw_empty[:, 3].min()), "max", float(raw_empty[:, 3].max()),
      " rays with mask 0:", int((rm[0] == 0).sum()), " flagged:", int(rm[3].sum()))
fr = ops.flat_rays(rm[1], rm[3], 48, 16)
print("flat_rays us", timeit(lambda: ops.flat_rays(rm[1], rm[3], 48, 16)), " listed rays", int(fr["ray_count"].item()))
print("composite_importance, listed rays only us", timeit(lambda: ops.composite_importance(raw, z, inp["rays_d"], 16, 1.0, bits=bits, raw_empty=raw_empty, want_weights=False, flat=fr)))
print("composite_merged, listed rays only     us", timeit(lambda: ops.composite_merged(raw, out["raw_fine"], out["sorted_idxs"], out["z_sorted"], inp["rays_d"], 1.0, bits_a=bits, bits_b=bits_f, raw_empty=raw_empty, flat=fr)))
print("view constants, all rays    us", timeit(lambda: eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"])))
print("view constants, listed rays us", timeit(lambda: eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"], fr["ray_list"], fr["ray_count"])))
