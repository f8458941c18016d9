"""Dev probe: the HBM rates torch's own streaming kernels reach on this box (copy = read + write, fill = write, sum = read),
the yardstick for the "mixed read / write" rate the A-NeRF trunk layers run at."""
import torch

dev = "cuda:0"
n = 1 << 29                      # 2 GiB of float32
a = torch.empty(n, device=dev)
b = torch.empty(n, device=dev)
a.normal_()


def timed(f, reps=10):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: b.copy_(a))
print(f"copy  2 GiB -> 2 GiB: {ms:.3f} ms  {2 * 4 * n / ms / 1e9:.2f} TB/s (read + write)")
ms = timed(lambda: b.fill_(1.0))
print(f"fill  2 GiB        : {ms:.3f} ms  {4 * n / ms / 1e9:.2f} TB/s (write)")
ms = timed(lambda: a.sum())
print(f"sum   2 GiB        : {ms:.3f} ms  {4 * n / ms / 1e9:.2f} TB/s (read)")
