"""dev tool: where the HOST time of a fused training step goes (cProfile over N steps of tools/bench_train's loop, sync_stats off)."""
import cProfile
import os
import pstats
import sys
import time

sys.argv = [sys.argv[0], "--steps", "0", "--warmup", "0"] + sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_train  # noqa: E402
import torch  # noqa: E402

# re-create bench_train's setup with its own code, then profile the loop
src = open(os.path.join(ROOT, "tools", "bench_train.py")).read()
setup = src[src.index("def main():"):src.index("    sync = a.sync_stats")]
ns = dict(bench_train.__dict__)
exec(setup + "    return trainer, batch\n", ns)
trainer, batch = ns["main"]()
for i in range(20):
    trainer.train_batch(batch, i=i, global_step=i, sync_stats=False)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    trainer.train_batch(batch, i=i, global_step=20 + i, sync_stats=False)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {1e3 * (t1 - t0) / N:.3f} ms (with cProfile), until GPU done {1e3 * (t2 - t0) / N:.3f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
# the same without the profiler
t0 = time.perf_counter()
for i in range(N):
    trainer.train_batch(batch, i=i, global_step=400 + i, sync_stats=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"no profiler: host enqueue per step {1e3 * (t1 - t0) / N:.3f} ms, until GPU done {1e3 * (t2 - t0) / N:.3f} ms")
