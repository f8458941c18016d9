"""probe (dev): do CU-masked streams exist here, does a kernel on a masked stream keep to its CUs, and do two kernels on
complementary masks run side by side?  K3 (every CU's registers) on `big` CUs, K2 of the same frame's fine pass on the rest.
usage: DANBO_NUM_CU=<big> python cu_mask_probe.py <big> [interleaved|blocked]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch
import bench
from core import hip_ops as ops

big = int(sys.argv[1]) if len(sys.argv) > 1 else 224
layout = sys.argv[2] if len(sys.argv) > 2 else "interleaved"
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(cus):
    words = (ctypes.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


if layout == "interleaved":
    big_cus, small_cus = list(range(big)), list(range(big, 256))
else:           # the same share of every group of 32
    per = big // 8
    big_cus = [32 * x + i for x in range(8) for i in range(per)]
    small_cus = [c for c in range(256) if c not in big_cus]
s_big, s_small = masked_stream(big_cus), masked_stream(small_cus)

eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
out = eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], 48, 16, keep=True)
vols = eng.volumes(inp["bones"])
view = eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"])
geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=out["z_coarse"])
bits, lst, cnt = ops.bone_cull(geo, True)
n = int(cnt.item())
h, _ = ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lst, cnt, geo.M)
raw = torch.empty(geo.R, 48, 4, device="cuda")
torch.cuda.synchronize()


def k3():
    eng._mlp(h, 48, view[0], raw, lst, cnt, geo.M)


def k2():
    ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lst, cnt, geo.M)


def timed(fn, stream, reps=5):
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cur = torch.cuda.current_stream()
print("rows", n, "big", len(big_cus), "small", len(small_cus), layout, "DANBO_NUM_CU", os.environ.get("DANBO_NUM_CU"))
print("K3 plain stream ms", round(timed(k3, cur), 3), " K3 on the big mask ms", round(timed(k3, s_big), 3))
print("K2 plain stream ms", round(timed(k2, cur), 3), " K2 on the small mask ms", round(timed(k2, s_small), 3))
# side by side: K3 on big, K2 x N on small, both started together
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record(cur)
s_big.wait_event(e0); s_small.wait_event(e0)
with torch.cuda.stream(s_big):
    k3()
    e1.record(s_big)
with torch.cuda.stream(s_small):
    for _ in range(2):
        k2()
    e2.record(s_small)
torch.cuda.synchronize()
print("side by side: K3 (big) done after ms", round(e0.elapsed_time(e1), 3), " 2 x K2 (small) done after ms", round(e0.elapsed_time(e2), 3))
