"""dev tool: the training step's per-ray view-gradient launch pair (danbo_train_view_grads: k_train_ray_grad + k_train_view_grad) ALONE on an
idle device -- in the step it runs on a side stream beside the K2 adjoint, where rocprof shows 160 - 190 us for 61 MFLOP."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
from core import _hip  # noqa: E402

dev = "cuda:0"
R, S, Cv, ldv, n_codes = 3072, 48, 155, 156, 20
rows = 53000
rows_pad = (rows + 127) // 128 * 128 + 128
g = torch.Generator(device="cpu").manual_seed(0)
dpre_v = torch.randn(rows_pad * 128, generator=g).to(dev) * 1e-4
row_ray = torch.sort(torch.randint(0, R, (rows,), generator=g)).values.int().to(dev)
cnt = torch.zeros(8, dtype=torch.int32, device=dev)
cnt[4] = rows
vin = torch.randn(R, ldv, generator=g).to(dev)
cam = (torch.arange(R) // 192 % n_codes).to(dev)
d_cview = torch.zeros(R, 128, device=dev)
csum = torch.zeros(n_codes, 128, device=dev)
g_vw = torch.zeros(128, 256 + Cv, device=dev)
part = torch.zeros(32 * 128 * 160, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
lib = _hip.lib()


def run():
    d_cview.zero_(); csum.zero_()
    _hip.check(lib.danbo_train_view_grads(P(dpre_v), P(row_ray), P(cnt), rows, R, P(vin), ldv, Cv, P(cam), n_codes, P(d_cview), P(csum), P(g_vw),
                                          P(part), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "view_grads")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
print("danbo_train_view_grads alone (incl. two zeroing launches): %.1f us per call" % (e0.elapsed_time(e1) / 50 * 1e3))
