"""which torch operators launch kernels around the fused training step (dev tool): one eager step under torch.profiler, the
operators that own a device kernel in launch order
    python tools/diag/step_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "3"]
import torch
from torch.profiler import profile, ProfilerActivity
import bench_train

orig = None
def patched_main():
    import core.trainer as T
    tb = T.Trainer.train_batch
    state = dict(n=0)
    def wrapped(self, *a, **k):
        state["n"] += 1
        if state["n"] == 4:
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
                r = tb(self, *a, **k)
                torch.cuda.synchronize()
            evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
            for e in sorted(evs, key=lambda e: e.time_range.start):
                print(e.name.ljust(40), [k.name[:50] for k in e.kernels][:3])
                for fr in (e.stack or [])[:6]:
                    if 'danbo' in fr or 'tools' in fr: print('      ', fr)
            return r
        return tb(self, *a, **k)
    T.Trainer.train_batch = wrapped
    bench_train.main()
patched_main()
