"""dev diagnostic (GPU): danbo_trunk_bwd's dz_l against float64 autograd, every layer printed (the test stops at the first miss)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("danbo-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import test_gpu_trunk as T
from core import _hip

R, S, Sf, n_c, n_f, gscale = 64, 8, 4, 0, 0, 1e-4
if len(sys.argv) > 1:
    R, S, Sf, n_c, n_f = (int(x) for x in sys.argv[1:6])
net = T.Net(seed=R + 1)
t = T.make_rows(net, R, S, Sf, n_c, n_f, seed=n_c + 3)
g = torch.Generator(device="cpu").manual_seed(5)
n, first_f = t["n"], R + n_c
row_gain = torch.exp(torch.randn(t["cap"], 1, generator=g) * 2.0)
t["d_raw_c"] = (torch.randn(R * S, 4, generator=g) * gscale).to(T.DEV)
t["d_raw_f"] = (torch.randn(R * Sf, 4, generator=g) * gscale).to(T.DEV)
t["d_raw_rows"] = (torch.randn(t["cap"], 4, generator=g) * gscale * row_gain).to(T.DEV)
w, r, cview = T.run_forward(net, t)
r = T.rows_struct(t, cview)
rs = t["row_sample"][:n].long()
G = t["d_raw_rows"][:n].clone()
G[R:first_f] = t["d_raw_c"][rs[R:first_f]]
G[first_f:] = t["d_raw_f"][rs[first_f:]]
_hip.check(_hip.lib().danbo_trunk_bwd(ctypes.byref(w), ctypes.byref(r), T.stream()), "trunk_bwd")
torch.cuda.synchronize()
ray, ref = T.reference_rows(net, t, keep_graph=True)
(ref["raw"] * G.double()).sum().backward()
rows_p = (n + 15) // 16 * 16
dv = T.frag_to_rows(t["dpre_v"], rows_p, 128)[:n].double()
print("d pre_v: max rel", ((dv - ref["pre_v"].grad).abs().max() / ref["pre_v"].grad.abs().max()).item())
for l in range(7, -1, -1):
    dz = T.frag_to_rows(t["dz"][l], rows_p, 256)[:n].double()
    refz = ref["zs"][l].grad
    bad = torch.isnan(dz).any(1)
    err = (dz - refz).abs().amax(1) / (refz.abs().amax() + 1e-300)
    print("layer", l, "tensor-relative", float(torch.nan_to_num(err, nan=9e9).max()), "rows with NaN", int(bad.sum()), "of", n,
          "worst rows", torch.topk(torch.nan_to_num(err, nan=9e9), 4).indices.tolist())
