"""dev diagnostic (GPU): where the fp16-split path's distance from float64 comes from on the bench frame's centre rays -- K2 (h) or
K3 (MLP) -- by mixing the exact-fp32 and the fp16-split kernels of each stage"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("danbo-pytorch_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import bench, torch_cpu, danbo_oracle as o
from core import hip_ops as ops
from core.utils import synthetic as syn

dev = torch.device("cuda:0")
eng, inp, (cfg, sd, rest, scene, ro, rd) = bench.build_workload(dev, 0)
H = W = 512
nr = 4096
r0 = (H // 2) * W - nr // 2
sl = slice(r0, r0 + nr)
S = 48
eng.refresh()
near, far = eng.near_far(inp["rays_o"][sl], inp["rays_d"][sl], inp["cyls"], inp["skts"])
z = ops.coarse_samples(near, far, S)
geo = ops.Geometry(inp["rays_o"][sl], inp["rays_d"][sl], inp["skts"], eng.align, eng.axis_scale, z=z)
vols = eng.volumes(inp["bones"])
bits, lst, cnt = ops.bone_cull(geo, True)
n = int(cnt.item())
rows = torch.sort(lst[:n]).values.contiguous()
h16, _ = ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, rows, None, n)
h32, _ = ops.gather_assign_blend(geo, vols, bits, eng.aw, rows, None, n)
print("rows", n, "|h| max", float(h32.abs().max()), " h16 - h32 max abs", float((h16 - h32).abs().max()))

# float64 restatement on the same points / mask
m64 = torch_cpu.DanboTorchCPU(cfg, sd, rest, dtype=torch.float64)
zz = z.cpu().numpy()
ron, rdn = ro[sl], rd[sl]
pts = o.sample_points(ron, rdn, zz)
pts_t = o.bone_local(pts, np.repeat(scene["skts"], nr, 0), m64.np_oracle.align)
_, valid = o.in_volume(pts_t, sd["graph_net.axis_scale"])
t64 = lambda v: torch.tensor(np.ascontiguousarray(v)).double()
raw64, lg64, _ = m64.forward(t64(pts), t64(rdn), t64(np.repeat(scene["skts"], nr, 0)), m64._volumes(scene["bones"]), torch.zeros(nr, dtype=torch.long),
                             np.zeros(nr, np.int64), return_enc=True, valid=torch.as_tensor(valid))
raw64 = raw64.numpy().reshape(-1, 4)[rows.cpu().numpy()]
cmax = np.abs(raw64).max(0)


def err(raw):
    r = raw.cpu().numpy().astype(np.float64)
    big = np.abs(raw64) > 0.1 * cmax
    return (float((np.abs(r - raw64) / np.maximum(np.abs(raw64), 1e-30))[big].max()), float((np.abs(r - raw64) / np.maximum(np.abs(raw64), 0.05 * cmax)).max()))


cam = torch.zeros(nr, dtype=torch.int64, device=dev)
for mlp_mode in ("f16split", "fp32"):
    eng.mlp_mode = mlp_mode
    eng.refresh()
    cview, raw_empty = eng.view_constants(geo.rays_d, geo.skts, cam)
    for hname, h in (("h from k_assign16", h16), ("h from exact K2", h32)):
        raw = torch.zeros(nr * S, 4, device=dev)
        eng._mlp(h, S, cview, raw, rows, None, n)
        print(f"K3 {mlp_mode:8s} | {hname:18s}: un-floored max rel {err(raw[rows.long()])[0]:.3e}   floored 5 % {err(raw[rows.long()])[1]:.3e}")
