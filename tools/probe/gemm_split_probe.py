"""Is a K-concatenated fp16 GEMM with fp32 output (hi|hi|lo x hi|lo|hi) a usable fp32-accurate trunk layer? (dev probe)"""
import torch, time
dev = "cuda:0"
torch.manual_seed(0)
n, K, N = 262144, 448, 448
x = torch.randn(n, K, device=dev).relu_()
w = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev) * 0.1
def split(t):
    hi = t.half(); lo = (t - hi.float()).half(); return hi, lo
xh, xl = split(x); wh, wl = split(w)
xc = torch.cat([xh, xh, xl], 1).contiguous()
wc = torch.cat([wh, wl, wh], 1).contiguous()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
ref = (x.double() @ w.double().t()).float()
try:
    y = torch.mm(xc, wc.t(), out_dtype=torch.float32)
    print("mm.dtype ok; rel err vs fp64", float((y - ref).abs().max() / ref.abs().max()))
    ms = timeit(lambda: torch.mm(xc, wc.t(), out_dtype=torch.float32))
    print("fp16 concat GEMM ms", ms, "executed fp32-equiv TFLOP/s", 2 * n * K * N / ms / 1e9)
except Exception as e:
    print("mm.dtype failed:", type(e).__name__, str(e)[:300])
y32 = x @ w.t()
print("fp32 GEMM rel err", float((y32 - ref).abs().max() / ref.abs().max()))
ms = timeit(lambda: torch.addmm(b, x, w.t()))
print("fp32 addmm ms", ms, "TFLOP/s", 2 * n * K * N / ms / 1e9)
ms = timeit(lambda: torch._addmm_activation(b, x, w.t()))
print("fp32 addmm+relu epilogue ms", ms)
yh = (xh @ wh.t())
ms = timeit(lambda: xh @ wh.t())
print("plain fp16 GEMM (K) ms", ms, "TFLOP/s", 2 * n * K * N / ms / 1e9)
ms = timeit(lambda: split(x))
print("torch split ms", ms)
