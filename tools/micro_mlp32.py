"""K3 in its two forms on the bench workload's coarse pass (dev tool): k_pe_mlp16 (16x16x32, two wavefronts per SIMD) against
k_pe_mlp32 (32x32x16, one wavefront per SIMD) -- same rows, same weights; results compared, launches timed alternately
    python tools/micro_mlp32.py [reps] [--rows=K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch
import bench
from core import hip_ops as ops

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
cview, raw_empty = eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"])
for a_ in sys.argv[1:]:
    if a_.startswith("--rows="):
        n = min(n, int(a_.split("=")[1]))
q = eng._equalized(eng.p)
pack = lambda form: ops.mlp16_pack(eng.pts_w, q["feature_linear.weight"], q["feature_linear.bias"], q["views_linears.0.weight"],
                                   q["views_linears.0.bias"], form=form)
(packed16, vb16), (packed32, vb32) = pack(16), pack(32)      # (the engine's own buffer is in the form of its mlp_form)
print("views_b_eff equal:", bool(torch.equal(vb32, eng.views_b16) and torch.equal(vb16, vb32)))
raw16 = torch.zeros(geo.M, 4, device="cuda")
raw32 = torch.zeros(geo.M, 4, device="cuda")
f16 = lambda aux=False: ops.pe_mlp16(h, 48, packed16, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, raw16, lst, None, n, aux)
f32 = lambda aux=False: ops.pe_mlp16(h, 48, packed32, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, raw32, lst, None, n, aux, form=32)
a16 = f16(True); a32 = f32(True); torch.cuda.synchronize()
rows = lst[:n].long()
d = (raw16[rows] - raw32[rows]).abs()
scale = raw16[rows].abs().amax(0)
print("rows", n, "max |raw16 - raw32| per channel", d.amax(0).tolist(), "channel max", scale.tolist())
print("rel to channel max", (d.amax(0) / scale).tolist(), " aux max diff", float((a16 - a32).abs().max()), "aux max", float(a16.abs().max()))
print("nan:", bool(torch.isnan(raw32).any()))


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(3):
    t16, t32 = timeit(f16), timeit(f32)
    print("k_pe_mlp16 %.4f ms   k_pe_mlp32 %.4f ms   (%.1f / %.1f of 833)" % (t16, t32, n * 611840 * 2 / t16 / 1e9, n * 611840 * 2 / t32 / 1e9))
if "--debug" in sys.argv:
    dcol = (a16 - a32).abs().amax(0)
    print("aux diff per column (x1e3):", [round(float(v) * 1e3, 2) for v in dcol])
    r = 0
    print("row 0 aux16[:16]", a16[r, :16].tolist()); print("row 0 aux32[:16]", a32[r, :16].tolist())
    print("row 1 aux16[60:76]", a16[1, 60:76].tolist()); print("row 1 aux32[60:76]", a32[1, 60:76].tolist())
    # is a32[:, 64:128] a permutation / another row's values of a16?
    x = a32[1, 64:128]
    best = ((a16[:64, 64:128] - x[None]).abs().amax(1)).min(0)
    print("closest a16 row to a32 row 1 (cols 64:128):", int(best.indices), float(best.values))
    for c in (64, 65, 70):
        dd = (a16[1, :128] - a32[1, c]).abs()
        print("a32[1,%d] closest a16[1,:] column" % c, int(dd.argmin()), float(dd.min()))
if "--debug" in sys.argv:
    import numpy as np
    pk = packed32[:74 * 32768].view(torch.float16).view(74, 32, 64, 8).float()
    for ch in (0, 7, 62, 70, 71, 73):
        print("chunk", ch, "per-piece max |w|:", [round(float(v), 1) for v in pk[ch].abs().amax((1, 2))])
if "--rows-diff" in sys.argv:
    dd = (raw16[rows] - raw32[rows]).abs()
    bad = (dd[:, 3] > 1e-4).nonzero().flatten()
    print("rows with a wrong density:", int(bad.numel()), "of", n, " first:", bad[:16].tolist(), " row %32:", sorted(set((bad % 32).tolist()))[:40])
if "--trace" in sys.argv:
    # dev variant of the library (tools/ab/build_variant.sh m32_TRACE k_mlp32.hip -DM32_TRACE): s_memtime stamps of one wavefront
    import ctypes
    from core import _hip
    lib = _hip.lib()
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    lib.danbo_dev_m32_trace.argtypes = [ctypes.c_void_p]
    assert lib.danbo_dev_m32_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    f32(); torch.cuda.synchronize(); f32(); torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(4, 16)
    names = ["layer 0 (13 k-substeps: 9 984 MFMA cycles; the encoding's slices under them)", "(two stamps back to back: what a stamp costs)",
             "layers 1-6 (109 k-substeps: 83 712)", "layer 7 (16: 12 288)", "view chunks 0-2 (12 k-substeps of 12 MFMAs: 4 608)",
             "view chunk 3 (1 536)", "drain + colour head"]
    for r in range(3):
        d = [int(t[r, i + 1] - t[r, i]) for i in range(7)]
        print("tile", r, "total", int(t[r, 7] - t[r, 0]), "| to next tile's start", int(t[r + 1, 0] - t[r, 7]))
        for n_, v in zip(names, d):
            print("    %-84s %8d" % (n_, v))

