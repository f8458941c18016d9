"""micro-benchmark of the f16-split MLP kernel on the bench workload's coarse pass (dev tool)
    python tools/micro_mlp16.py [reps] [--zeros]     --zeros: same rows, all-zero weights and inputs (no operand toggling: power probe)
    --rows=K: only the first K rows"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch
import bench
from core import hip_ops as ops

zeros = "--zeros" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(args[0]) if args else 10
eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
eng.refresh()
near, far = eng.near_far(inp["rays_o"], inp["rays_d"], inp["cyls"], inp["skts"])
z = ops.coarse_samples(near, far, 48)
vols = eng.volumes(inp["bones"])
geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z)
bits, lst, cnt = ops.bone_cull(geo, True)
n = int(cnt.item())
h = ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lst, cnt, geo.M)[0]
raw = torch.zeros(geo.M, 4, device="cuda")
cview, raw_empty = eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"])
if zeros:
    with torch.no_grad():
        for name, prm in eng.p.items():
            if 'adj' not in name.split('.')[-1] or 'adj_w' in name:     # (the 0/1 adjacency BUFFERS stay: the fast K2 kernel checks them)
                prm.zero_()
    eng.refresh()
    h.zero_(), cview.zero_()
for a_ in sys.argv[1:]:
    if a_.startswith("--rows="):        # the same kernel on the first k in-volume rows only (the training step's sizes: 41 000 / 18 000)
        n = min(n, int(a_.split("=")[1]))
fn = lambda: ops.pe_mlp16(h, 48, eng.packed16, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, raw, lst, None, n, form=eng.mlp_form)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("rows", n, "mlp16 ms", round(ms, 4), "TFLOP/s executed", round(n * 611840 * 2 / ms / 1e9, 1), "of 833")
