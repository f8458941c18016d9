"""Shader clock and socket power (hwmon of the visible GPU) while one kernel runs back to back (dev tool):
    python tools/clock_trace.py [mlp32|mlp32z|mlp16|mlp16z|linear16|copy|idle] [seconds]      (mlp32 / mlp16: K3 in its two forms)
Samples freq1_input (sclk) and power1_input every ~2 ms from a thread while the main thread keeps the queue full."""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import numpy as np
import torch


def hwmon():
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if os.path.exists(d + "/freq1_input") and os.path.exists(d + "/power1_input"):
            return d
    return None


def rd(p):
    with open(p) as f:
        return float(f.read().strip())


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "mlp16"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    d = hwmon()
    print("hwmon", d, "cap W", rd(d + "/power1_cap") / 1e6 if d else None)
    dev = torch.device("cuda:0")
    zeros = what in ("mlp16z", "mlp32z")  # same launch, all-zero weights and inputs: identical instruction stream, no operand toggling
    form = 16 if what.startswith("mlp16") else 32
    if what.startswith("mlp"):
        what = "mlp"
    if what == "mlp":
        import bench
        from core import hip_ops as ops
        eng, inp, _ = bench.build_workload(dev, 0)
        eng.mlp_form = form
        eng.refresh()
        near, far = eng.near_far(inp["rays_o"], inp["rays_d"], inp["cyls"], inp["skts"])
        z = ops.coarse_samples(near, far, 48)
        vols = eng.volumes(inp["bones"])
        geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z)
        bits, lst, cnt = ops.bone_cull(geo, True)
        n = int(cnt.item())
        h = ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lst, cnt, geo.M)[0]
        raw = torch.zeros(geo.M, 4, device=dev)
        cview, _ = eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"])
        if zeros:
            with torch.no_grad():
                for name, prm in eng.p.items():
                    if 'adj' not in name.split('.')[-1] or 'adj_w' in name:
                        prm.zero_()
            eng.refresh()
            h.zero_(), cview.zero_()
        fn = lambda: ops.pe_mlp16(h, 48, eng.packed16, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, raw, lst, None, n, form=eng.mlp_form)
        flops = n * 611840 * 2 * 3
    elif what == "linear16":
        from core import hip_ops as ops
        M = 1 << 20
        x, w = torch.randn(M, 448, device=dev), torch.randn(448, 448, device=dev) / 21
        packed, shape = ops.linear16_pack(w)
        y = torch.empty(M, 448, device=dev)
        fn = lambda: ops.linear16(x, packed, shape, None, relu=True, out=y)
        flops = M * 448 * 448 * 2 * 3
    elif what == "copy":
        a = torch.empty(1 << 28, device=dev)
        b = torch.empty_like(a)
        fn = lambda: b.copy_(a)
        flops = 0
    else:
        fn, flops = None, 0
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), rd(d + "/freq1_input") / 1e6, rd(d + "/power1_input") / 1e6))
            time.sleep(0.002)
    if fn is not None:
        fn(); torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); reps = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < secs:
        if fn is None:
            time.sleep(0.05)
        else:
            for _ in range(20): fn()
            reps += 20
            torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    s = np.array(samples)[len(samples) // 5:]          # the first fifth: ramp
    ms = e0.elapsed_time(e1) / max(reps, 1)
    print(what, "reps", reps, "ms/launch", round(ms, 4), "executed f16 TFLOP/s", round(flops / ms / 1e9, 1) if flops else None)
    print("samples", len(s), "sclk MHz mean/min/max", round(s[:, 1].mean()), s[:, 1].min(), s[:, 1].max(),
          "| power W mean/min/max", round(s[:, 2].mean()), s[:, 2].min(), s[:, 2].max())


main()
