"""dev diagnostic (GPU): the degenerate training batches of tests/test_gpu_train_engine.py, every path variant against the float64
reference, per parameter.  usage: python tools/diag/degenerate_vs_f64.py [case ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("danbo-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_train_engine as tt  # noqa: E402
from helpers import golden  # noqa: E402
from test_gpu_training import batch_of  # noqa: E402

DEV = "cuda:0"


def edits(case):
    def model_edit(caster):
        if case in ("overlapping_volumes", "every_volume"):
            with torch.no_grad():
                caster.network.graph_net.axis_scale.mul_(4.0 if case == "overlapping_volumes" else 12.0)

    def edit(b):
        pass
    return edit, model_edit


def report(tag, grads, r64, top=4):
    rows = []
    for n, gr in grads.items():
        t = r64["grads"][n]
        rows.append((float(np.abs(gr.detach().cpu().numpy().astype(np.float64) - t).max()) / (float(np.abs(t).max()) + 1e-30), n))
    rows.sort(reverse=True)
    print(f"  {tag:34s}", "  ".join(f"{n}={e:.2e}" for e, n in rows[:top]))


for case in (sys.argv[1:] or ["overlapping_volumes", "every_volume"]):
    print("====", case)
    edit, model_edit = edits(case)
    from core import train_path
    import torch_layerwise                       # tests/torch_layerwise.py: the layer-by-layer torch route (test infrastructure)
    saved_fwd = train_path.forward_train
    orig_comp = train_path.composite
    kept = []

    def comp_spy(raw, z, rays_d, B=1.0, noise=None):
        raw1 = raw * 1.0
        raw1.retain_grad()
        kept.append(raw1)
        return orig_comp(raw1, z, rays_d, B, noise)
    train_path.composite = comp_spy
    orig_pe = train_path.positional_encoding
    hkept = []

    def pe_spy(x, L):
        if x.shape[-1] == 15 and x.requires_grad:
            x = x * 1.0
            x.retain_grad()
            hkept.append(x)
        return orig_pe(x, L)
    train_path.positional_encoding = pe_spy
    for assign in ("library",):
        for mlp in ("library",):
            train_path.forward_train = torch_layerwise.forward_train if assign == "library" else saved_fwd
            samp = {}
            ref, preds, loss = tt._autograd_grads("danbo_perfcap_train", edit, model_edit, sampling=samp)
            g = golden("danbo_perfcap_train")
            args, caster, trainer, opt = tt.build_trainer(g)
            model_edit(caster)
            b = batch_of(g)
            samp.update(acc0=preds["acc0"], acc_map=preds["acc_map"])
            print("   rays at acc = 1:", int((preds["acc0"] >= 1).sum()), int((preds["acc_map"] >= 1).sum()), "of", len(preds["acc0"]))
            dbg = {}
            r64 = tt._f64_reference(g, args, caster, b, samp, debug=dbg)
            for name, mine, want in (("d_raw coarse", kept[-2].grad, dbg["d_raw_coarse"]), ("d_raw sorted", kept[-1].grad, dbg["d_raw_sorted"])):
                mine = mine.detach().cpu().numpy().astype(np.float64)
                dev = np.abs(mine - want)
                ray = dev.reshape(dev.shape[0], -1).max(1)
                worst = np.argsort(ray)[-5:]
                print(f"   {name}: max |.| {np.abs(want).max():.3e}  max dev {dev.max():.3e}  rays with dev > 1e-3 of max: {(ray > 1e-3 * np.abs(want).max()).sum()}  worst rays {worst.tolist()}")
                r = int(worst[-1])
                np.set_printoptions(linewidth=200, precision=3)
                print("     worst ray mine  d sigma:", mine[r, :, 3])
                print("     worst ray f64   d sigma:", want[r, :, 3])
                print("     raw sigma:", (dbg["raw_coarse"] if "coarse" in name else dbg["raw_sorted"])[r, :, 3])
                print("     acc f64", (dbg["acc0"] if "coarse" in name else dbg["acc"])[r], "path acc", float((preds["acc0"] if "coarse" in name else preds["acc_map"])[r]))
            kept.clear()
            for name, hk, hh, dh, anyv in (("coarse", hkept[0], dbg["h_c"], dbg["d_h_c"], dbg["any_c"]), ("fine", hkept[1], dbg["h_f"], dbg["d_h_f"], dbg["any_f"])):
                rows = np.nonzero(anyv)[0]
                mine_h, mine_dh = hk.detach().cpu().numpy().astype(np.float64), hk.grad.detach().cpu().numpy().astype(np.float64)
                print(f"   {name}: rows {len(rows)} (path {mine_h.shape[0]})  |h| max {np.abs(hh[rows]).max():.3f}  h dev {np.abs(mine_h - hh[rows]).max():.3e}  "
                      f"d_h max {np.abs(dh[rows]).max():.3e} dev {np.abs(mine_dh - dh[rows]).max():.3e}")
                bad = np.argsort(np.abs(mine_dh - dh[rows]).max(1))[-3:]
                for r_ in bad:
                    print("      row", r_, "sample", rows[r_], "mine", mine_dh[r_][:5], "f64", dh[rows][r_][:5], "h", hh[rows][r_][:3])
            hkept.clear()
            print(f" autograd assign={assign} mlp={mlp}: loss {loss['total_loss']:.8f} f64 {r64['loss']['total_loss']:.8f}  "
                  f"rgb_map dev {np.abs(preds['rgb_map'].detach().cpu().numpy() - r64['rgb_map']).max():.2e}")
            report("grads vs f64", ref, r64)
    train_path.composite = orig_comp
    train_path.positional_encoding = orig_pe
    train_path.forward_train = saved_fwd
    g, args, caster, trainer, eng, out = tt.fused_step("danbo_perfcap_train", edit=edit, model_edit=model_edit)
    b = batch_of(g)
    R, G = b["rays_o"].shape[0], int(b["N_uniques"])
    samp_f = tt._fused_sampling(eng, R, G, int(g["N_samples"]), int(g["N_importance"]))
    samp_f.update(acc0=out["acc0"], acc_map=out["acc_map"])
    print("   rays at acc = 1:", int((out["acc0"] >= 1).sum()), int((out["acc_map"] >= 1).sum()), "of", len(out["acc0"]))
    r64 = tt._f64_reference(g, args, caster, b, samp_f)
    print(f" fused: rgb_map dev {np.abs(out['rgb_map'].cpu().numpy() - r64['rgb_map']).max():.2e}")
    report("grads vs f64", {n: p.grad for n, p in caster.network.named_parameters()}, r64)
    import torch_f64_train as t64
    t64.F64 = torch.float32
    r32 = tt._f64_reference(g, args, caster, b, samp_f)
    t64.F64 = torch.float64
    report("SAME restatement in fp32 vs f64", {n: torch.tensor(v) for n, v in r32["grads"].items()}, r64)
    for _ in range(1):      # run-to-run
        g2, args2, caster2, trainer2, eng2, out2 = tt.fused_step("danbo_perfcap_train", edit=edit, model_edit=model_edit)
        report("fused again vs f64", {n: p.grad for n, p in caster2.network.named_parameters()}, r64)
