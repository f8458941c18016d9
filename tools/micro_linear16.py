"""Timing of k_linear16 over layer shapes (dev tool): ms per call, fp32-equivalent TFLOP/s, HBM GB/s of the algorithmic bytes.
    python tools/micro_linear16.py [--rows 262144]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
from core import hip_ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262144)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--zeros", action="store_true", help="all-zero inputs and weights: no operand toggling (power probe)")
    ap.add_argument("--trace", action="store_true", help="s_memtime stamps of one wavefront over one row tile (448 x 448)")
    a = ap.parse_args()
    dev = "cuda:0"
    if a.trace:
        import ctypes
        from core import _hip
        M, K, N = a.rows, 448, 448
        x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5
        packed, shape = ops.linear16_pack(w)
        y = torch.empty(M, N, device=dev)
        ops.linear16(x, packed, shape, None, relu=True, out=y)
        buf = torch.zeros(256, dtype=torch.int64, device=dev)
        _hip.lib().danbo_linear16_set_trace(ctypes.c_void_p(buf.data_ptr()))
        ops.linear16(x, packed, shape, None, relu=True, out=y)
        torch.cuda.synchronize()
        _hip.lib().danbo_linear16_set_trace(None)
        t = buf.cpu().numpy()
        t = t[t > 0]
        d = (t[1:] - t[:-1])
        # per k-step: [start, before h0, after h0, before h1, after h1, end] -> 5 intervals + gap to the next k-step
        print("stamps", len(t), "total cycles", int(t[-1] - t[0]))
        names = ["take+split+request", "h0 vmcnt wait", "h0 barrier", "h0 issue", "chunk 0 (16 tiles)", "h1 vmcnt wait", "h1 barrier", "h1 issue",
                 "chunk 1 (12 tiles)", "(to next k-step)"]
        per = d[:10 * 14].reshape(-1, 10) if len(d) >= 140 else d
        print(names)
        print(per)
        print("mean", per.mean(0) if hasattr(per, "mean") else per)
        return
    for M, K, N in [(a.rows, 448, 448), (a.rows, 448, 256), (a.rows, 448, 512), (a.rows, 896, 448), (a.rows, 64, 448),
                    (a.rows, 448, 64), (a.rows // 8, 448, 448), (a.rows * 4, 448, 448), (a.rows, 256, 256)]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        if a.zeros:
            x.zero_(), w.zero_(), b.zero_()
        packed, shape = ops.linear16_pack(w)
        y = torch.empty(M, N, device=dev)
        for _ in range(3):
            ops.linear16(x, packed, shape, b, relu=True, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ops.linear16(x, packed, shape, b, relu=True, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        wt = torch.randn(K, N, device=dev)
        for _ in range(3):
            torch.relu_(torch.addmm(b, x, wt))
        e0.record()
        for _ in range(a.reps):
            torch.relu_(torch.addmm(b, x, wt))
        e1.record()
        torch.cuda.synchronize()
        ms_lib = e0.elapsed_time(e1) / a.reps
        print(f"M={M:8d} K={K:4d} N={N:4d}: {ms:7.3f} ms  {2e-9 * M * K * N / ms:7.1f} TFLOP/s  {4e-6 * M * (K + N) / ms:7.1f} GB/s"
              f"   | library fp32 GEMM + relu {ms_lib:7.3f} ms {2e-9 * M * K * N / ms_lib:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
