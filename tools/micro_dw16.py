"""danbo_dw16 on the training step's layer mix at M rows: trunk operands row-major against fragment order (dev tool)
    python tools/micro_dw16.py [rows] [slices] [--trunk-only]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch
from core import _hip, hip_ops as ops

args = [a for a in sys.argv[1:] if not a.startswith("--")]
M = int(args[0]) if len(args) > 0 else 50000
slices = int(args[1]) if len(args) > 1 else 10
trunk_only = "--trunk-only" in sys.argv
dev = "cuda:0"
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
slack = lambda f: torch.cat([f.data, torch.zeros(128 * f.C, device=dev)])
y = [torch.randn(M, 256, device=dev) for _ in range(8)]            # y0 .. y7
dz = [torch.randn(M, 256, device=dev) * 1e-6 for _ in range(8)]
pe = torch.randn(M, 196, device=dev); fa = torch.randn(M, 260, device=dev); vinr = torch.randn(M, 156, device=dev)
hv = torch.randn(M, 128, device=dev); dpre = torch.randn(M, 128, device=dev) * 1e-6; draw = torch.randn(M, 4, device=dev) * 1e-6
dx5 = torch.randn(M, 452, device=dev) * 1e-6
mx = torch.tensor([1e-6 * 5], device=dev)
G = lambda *s: torch.zeros(*s, device=dev)
for mode in ("rows", "frag"):
    fr = mode == "frag"
    yb = [slack(ops.FragBuffer.from_rows(t)) if fr and i < 7 else t for i, t in enumerate(y)]
    zb = [slack(ops.FragBuffer.from_rows(t)) if fr and i not in (4, 5) else t for i, t in enumerate(dz)]
    D = _hip.DanboDwLayer
    def layer(dy, ldy, N, x1, ld1, K1, x2=None, ld2=0, K2=0, frag=0, gw_ld=0, gw_col0=0, gb=True):
        return D(dy=P(dy), x1=P(x1), x2=P(x2), dy_maxabs=P(mx), gw=P(G(N, gw_ld or K1 + K2)), gw2=None, gb=P(G(N)) if gb else None, gb2=None,
                 ldy=ldy, ld1=ld1, ld2=ld2, N=N, K1=K1, K2=K2, split_n=0, frag=frag, gw_ld=gw_ld, gw_col0=gw_col0)
    Ls, keep = [], []
    for l in range(8):
        dyl, fa_ = (dx5, 0) if l == 4 else (zb[l], 1 if fr and l != 5 else 0)
        ldy = 452 if l == 4 else 256
        if l == 0: Ls.append(layer(dyl, ldy, 256, pe, 196, 195, frag=fa_))
        elif l == 5:
            Ls.append(layer(dyl, ldy, 256, pe, 196, 195, frag=fa_, gw_ld=451))
            Ls.append(layer(dyl, ldy, 256, yb[4], 256, 256, frag=fa_ | (2 if fr else 0), gw_ld=451, gw_col0=195, gb=False))
        else: Ls.append(layer(dyl, ldy, 256, yb[l - 1], 256, 256, frag=fa_ | (2 if fr else 0)))
    if not trunk_only:
        Ls += [layer(dz[7], 256, 256, y[7], 256, 256), layer(draw, 4, 1, y[7], 256, 256), layer(dpre, 128, 128, fa, 260, 256, vinr, 156, 155),
               layer(draw, 4, 3, hv, 128, 128)]
    L = (D * len(Ls))(*Ls)
    scratch = torch.empty(_hip.lib().danbo_dw16_scratch_floats(L, len(Ls), slices), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: _hip.check(_hip.lib().danbo_dw16(L, len(Ls), M, None, slices, P(scratch), st), "dw16")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(mode, len(Ls), "layers: ms", round(e0.elapsed_time(e1) / 20, 4))
