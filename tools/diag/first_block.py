"""dev probe: why the first timed block of bench.py's config 1 read 5.1 - 7.6 ms per frame against 4.7 for the other four.
    python tools/diag/first_block.py MODE
render_fn: bench.bench_render as a function; set_device: torch.cuda.set_device + the K3 event profile from the first timed block
on (the outlier); no_profile: the same without the events (none); prime_frame: instrumented frames in front of the timed region
(2 - 3 frames: the outlier stays on most leases; bench.py runs a whole block's worth, which removes it -- the runtime's pool of
profiling signals is sized by the events of one block); prime_sleep: an idle gap instead (outlier stays); nodev / dev_late: without /
with a late torch.cuda.set_device (no difference: the outlier is intermittent in all of them)"""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
mode = sys.argv[1]
if mode == "render_fn":
    args = argparse.Namespace(steps=20, warmup=3, mlp="f16split", box_near_far=False, config=1, no_cpu_baseline=True, no_dense=True, no_sweep=True,
                              debug_single_device=False, gpus=1)
    r = bench.bench_render(args, 0, 1, dev, None)
    print(mode, [round(b, 3) for b in r["block_ms"]])
elif mode == "set_device":
    torch.cuda.set_device(0)
    eng, inp, _ = bench.build_workload(dev, view=0, mlp_mode="f16split")
    eng.cfg["use_volume_near_far"] = False
    med, out, info = bench.timed(lambda: bench.render(eng, inp), 20, 3, None, dev, False, on_timed_start=lambda: setattr(eng, "profile", {}))
    print(mode, [round(b, 3) for b in info["block_ms"]])
elif mode in ("prime_frame", "no_profile", "prime_sleep"):
    torch.cuda.set_device(0)
    eng, inp, _ = bench.build_workload(dev, view=0, mlp_mode="f16split")
    def hook():
        if mode == "prime_frame":
            eng.profile = {}
            for _ in range(3):
                bench.render(eng, inp)
            torch.cuda.synchronize()
            eng.profile = {}
        elif mode == "prime_sleep":
            import time
            torch.cuda.synchronize(); time.sleep(0.2)
            eng.profile = {}
    med, out, info = bench.timed(lambda: bench.render(eng, inp), 20, 3, None, dev, False, on_timed_start=hook)
    print(mode, [round(b, 3) for b in info["block_ms"]])
elif mode in ("nodev", "dev_late"):
    eng, inp, _ = bench.build_workload(dev, view=0, mlp_mode="f16split")
    if mode == "dev_late":
        torch.cuda.set_device(0)
    eng.cfg["use_volume_near_far"] = False
    med, out, info = bench.timed(lambda: bench.render(eng, inp), 20, 3, None, dev, False, on_timed_start=lambda: setattr(eng, "profile", {}))
    print(mode, [round(b, 3) for b in info["block_ms"]])
