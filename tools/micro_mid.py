"""dev tool: microseconds per launch of danbo_train_mid (the training step's fused loss-gradient / composite-adjoint / un-merge kernel) at
the PerfCap batch shape, back to back"""
import ctypes, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
from core import _hip
lib = _hip.lib()
DEV = "cuda:0"
R, S, Sf, B = 3072, 32, 16, 7.5
St = S + Sf
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
uni = lambda *s: torch.rand(*s, generator=g).to(DEV)
rgb, rgb0, target, bgs = uni(R, 3), uni(R, 3), uni(R, 3), uni(R, 3)
acc, acc0 = uni(R), uni(R)
raw_c, raw_empty, raw_sorted = rnd(R, S, 4), rnd(R, 4), rnd(R, St, 4)
bits_c = (torch.rand(R, S, generator=g) < 0.4).to(torch.int32).to(DEV)
bits_f = (torch.rand(R, Sf, generator=g) < 0.4).to(torch.int32).to(DEV)
z_c = torch.sort(uni(R, S) * 4 + 1, dim=1).values.contiguous()
z_s = torch.sort(uni(R, St) * 4 + 1, dim=1).values.contiguous()
rays_d = rnd(R, 3); noise_c, noise_f = rnd(R, S), rnd(R, St)
order = torch.stack([torch.randperm(St, generator=g) for _ in range(R)]).to(torch.int32).to(DEV)
weights, alpha = uni(R, St), uni(R, St)
o = dict(g_rgb=torch.zeros(R, 3, device=DEV), g_acc=torch.zeros(R, device=DEV), g_rgb0=torch.zeros(R, 3, device=DEV), g_acc0=torch.zeros(R, device=DEV),
         d_raw_c=torch.zeros(R, S, 4, device=DEV), d_raw_f=torch.zeros(R, Sf, 4, device=DEV), d_raw_rows=torch.zeros(R, 4, device=DEV),
         label_c=torch.zeros(R, S, dtype=torch.uint8, device=DEV), label_f=torch.zeros(R, Sf, dtype=torch.uint8, device=DEV),
         loss=torch.zeros(8, device=DEV), maxabs=torch.zeros(4, device=DEV))
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    lib.danbo_train_mid(P(rgb), P(acc), P(rgb0), P(acc0), P(target), P(bgs), 1, R, S, Sf, 0, 1.0, 0.5, B, P(o["g_rgb"]), P(o["g_acc"]), P(o["g_rgb0"]), P(o["g_acc0"]),
                        P(raw_c), P(raw_empty), P(raw_sorted), P(bits_c), P(bits_f), P(z_c), P(z_s), P(rays_d), P(noise_c), P(noise_f), P(order), P(weights), P(alpha),
                        P(o["d_raw_c"]), P(o["d_raw_f"]), P(o["d_raw_rows"]), P(o["label_c"]), P(o["label_f"]), P(o["loss"]), P(o["maxabs"]), st)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("DANBO_HIP_LIB", "tree"), "us per launch", e0.elapsed_time(e1) / 200 * 1e3)
