"""dev tool: replay the captured training step on ONE deterministic batch many times and report every replay whose gradients
differ from the first one's (and in which parameter) -- the long form of tests/test_gpu_train_engine.py::test_graph_replays_are_repeatable.
usage: stress_replay.py [replays] [--busy]   (--busy: a render frame between replays, to move the GPU's timing around)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("danbo-pytorch_amd", "oracle", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
from helpers import golden  # noqa: E402
from test_gpu_train_engine import fused_step  # noqa: E402
from test_gpu_training import batch_of  # noqa: E402

LD_VIN = int(os.environ.get('LD_VIN', '156'))
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
g, args, caster, trainer, eng, out = fused_step("danbo_perfcap_train", graph=True)
b = batch_of(g)
G = b["N_uniques"]
pp = caster._per_pose
ref_counts, ref = out["counts"].clone(), eng.flat_g.clone()
ref_out = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
torch.cuda.synchronize()
ref_ws = eng._ws.clone() if "--ws" in sys.argv else None
busy = None
if "--busy" in sys.argv:
    import bench
    busy = bench.build_workload(torch.device("cuda:0"), 0)
offs = sorted(eng.offsets.items(), key=lambda kv: kv[1])
bad = 0
junk = torch.empty(96 << 20, device="cuda") if "--flush" in sys.argv else None      # 384 MB: larger than the L2s and the MALL
for i in range(n):
    if junk is not None:
        junk.add_(1.0)
    if busy is not None and i % 3 == 0:
        e, inp, _ = busy
        e.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], 48, 16)
    out = eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"], G), pp(b["bones"], G), pp(b["cyls"], G), b["cam_idxs"],
                               b["target_s"], b["bgs"], int(g["N_samples"]), int(g["N_importance"]))
    torch.cuda.synchronize()
    if ref_ws is not None:
        al = lambda x: (x + 255) & ~255  # noqa: E731
        Rr, Wg = b["rays_o"].shape[0], eng._model().graph_width
        o0 = 240896 + (Rr * (int(g["N_samples"]) + int(g["N_importance"]))) * 4          # behind `order` (offsets printed at the end)
        o1 = al(o0) + 3 * G * 24 * Wg * 4
        o2 = al(o1) + G * 24 * 240 * 4
        o3 = al(o2) + (Rr * LD_VIN + 8) * 4
        o4 = al(o3) + Rr * 128 * 4
        o5 = al(o4) + 3 * 576 * 4
        for name, a0, a1 in (("vol_scratch", al(o0), o1), ("volumes", al(o1), o2), ("vin", al(o2), o3), ("cview", al(o3), o4), ("adj_prod", al(o4), o5)):
            cur_f, ref_f = eng._ws[a0:a1].view(torch.float32), ref_ws[a0:a1].view(torch.float32)
            if not torch.equal(cur_f, ref_f):
                d = (cur_f - ref_f).abs()
                idx = torch.nonzero(d > 0).reshape(-1)
                print("replay", i, name, "differs in", idx.numel(), "floats, max", float(d.max()), "first at float", idx[:6].tolist(), "last", int(idx[-1]))
                if name == "cview":
                    i0 = int(idx[0])
                    print("      now ", [round(x, 5) for x in cur_f[i0:i0 + 16].tolist()])
                    print("      ref ", [round(x, 5) for x in ref_f[i0:i0 + 16].tolist()])
                    print("      same columns of the rows before / after (ref):", [round(x, 5) for x in ref_f[i0 - 128:i0 - 128 + 4].tolist()], [round(x, 5) for x in ref_f[i0 + 128:i0 + 132].tolist()])
                    # does the wrong segment exist anywhere else in the workspace?
                    wsf = eng._ws[: eng._ws.numel() // 4 * 4].view(torch.float32)
                    hit = torch.nonzero((wsf[:-1] == cur_f[i0]) & (wsf[1:] == cur_f[i0 + 1])).reshape(-1)
                    print("      first two wrong values found at float offsets", hit[:8].tolist(), "(cview starts at float", a0 // 4, ")")
    diff = (eng.flat_g - ref).abs()
    if not torch.equal(out["counts"], ref_counts) or float(diff.max()) > 1e-5 * float(ref.abs().max()):
        bad += 1
        print("replay", i, "counts", out["counts"].tolist(), "ref", ref_counts.tolist())
        if ref_ws is not None:          # which bytes of the step's workspace differ from the first replay's (runs of 4 KB pages)
            ne = (eng._ws != ref_ws)
            pages = torch.nonzero(ne.view(-1)[: ne.numel() // 4096 * 4096].view(-1, 4096).any(1)).reshape(-1).tolist()
            runs, start, prev = [], None, None
            for pg in pages:
                if start is None:
                    start = prev = pg
                elif pg == prev + 1:
                    prev = pg
                else:
                    runs.append((start, prev)); start = prev = pg
            if start is not None:
                runs.append((start, prev))
            print("    workspace of", eng._ws.numel(), "bytes at", hex(eng._ws.data_ptr()), ": differing 4-KB page runs", [(a * 4096, (b + 1) * 4096, int(ne.view(-1)[a * 4096:(b + 1) * 4096].sum())) for a, b in runs][:40])
        for k, v in ref_out.items():
            if not torch.equal(out[k], v):
                d = (out[k].float() - v.float()).abs()
                print("    output", k, "differs: max", float(d.max()), "entries", int((d > 0).sum()), "/", d.numel(),
                      "rows", torch.nonzero(d.reshape(d.shape[0], -1).sum(1) > 0).reshape(-1).tolist()[:8] if d.dim() > 1 else "")
        for (name, o), (_, o2) in zip(offs, offs[1:] + [("end", diff.numel())]):
            if o2 > o and float(diff[o:o2].max()) > 1e-6 * float(ref.abs().max()):
                print("   ", name, "max diff", float(diff[o:o2].max()), "of", float(ref[o:o2].abs().max()), "entries", int((diff[o:o2] > 0).sum()), "/", o2 - o)
print("done:", n, "replays,", bad, "differ")
try:
    import ctypes, struct
    from core import _hip
    buf = (ctypes.c_uint * 8)()
    f = _hip.lib().danbo_dbg_cview
    f.argtypes = [ctypes.c_void_p]
    rc = f(ctypes.byref(buf))
    tof = lambda u: struct.unpack("f", struct.pack("I", u))[0]  # noqa: E731
    print("cview LDS check: mismatches", buf[0], "last: ray", buf[1], "k", buf[2], "want", tof(buf[3]), "got", tof(buf[4]), "wg", buf[5], "rc", rc, " recomputed sums that differ:", buf[6], "last (ray * 1000 + feature)", buf[7])
except AttributeError:
    pass
if ref_ws is not None:
    import ctypes
    from core import _hip
    v = _hip.DanboTrainView()
    R = b["rays_o"].shape[0]
    _hip.check(_hip.lib().danbo_train_workspace_view(ctypes.byref(eng._model()), R, G, int(g["N_samples"]), int(g["N_importance"]), R,
                                                     ctypes.c_void_p(eng._ws.data_ptr()), ctypes.byref(v)), "view")
    print("offsets: z_coarse", v.z_coarse - eng._ws.data_ptr(), "z_fine", v.z_fine - eng._ws.data_ptr(), "z_sorted", v.z_sorted - eng._ws.data_ptr(),
          "order", v.order - eng._ws.data_ptr(), "bits_coarse", v.bits_coarse - eng._ws.data_ptr(), "bits_fine", v.bits_fine - eng._ws.data_ptr())
