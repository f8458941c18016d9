"""dev diagnostic (GPU): both training paths against the float64 reference for a range of axis_scale multipliers of the degenerate
batches (tests/test_gpu_train_engine.py: overlapping_volumes / every_volume) -- finds multipliers at which no ReLU of a
high-gradient row sits within fp32 round-off of its kink (there the gradient is discontinuous and fp32 paths legitimately land on
either side: x 12 has one such (sample, unit) pair worth 4 % of pts_linears.6.weight)."""
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


def worst(grads, r64):
    rows = []
    for n, gr in grads.items():
        t = r64["grads"][n]
        rows.append((float(np.abs(gr.detach().cpu().numpy().astype(np.float64) - t).max()) / (float(np.abs(t).max()) + 1e-30), n))
    return max(rows)


for mult in [float(x) for x in (sys.argv[1:] or [3, 3.5, 4, 4.5, 5, 6, 8, 10, 11, 12, 13, 14, 16, 20])]:
    def model_edit(caster):
        with torch.no_grad():
            caster.network.graph_net.axis_scale.mul_(mult)
    edit = lambda b: None  # noqa: E731
    samp = {}
    ref, preds, loss = tt._autograd_grads("danbo_perfcap_train", edit, model_edit, sampling=samp)
    samp.update(acc0=preds["acc0"], acc_map=preds["acc_map"])
    g, args, caster, trainer, eng, out = tt.fused_step("danbo_perfcap_train", edit=edit, model_edit=model_edit)
    b = batch_of(g)
    R, G = b["rays_o"].shape[0], int(b["N_uniques"])
    sf = tt._fused_sampling(eng, R, G, int(g["N_samples"]), int(g["N_importance"]))
    sf.update(acc0=out["acc0"], acc_map=out["acc_map"])
    ra, rf = tt._f64_reference(g, args, caster, b, samp), tt._f64_reference(g, args, caster, b, sf)
    pairs = int((preds["part_invalid"] == 0).sum())
    cap = out["rgb_map"].shape[0] * (out["alpha"].shape[1] + 1)
    counts = out["counts"].cpu().numpy()
    wa, wf = worst(ref, ra), worst({n: p.grad for n, p in caster.network.named_parameters()}, rf)
    print(f"x{mult:5.1f}: pairs/row {pairs / max(int(counts[5]), 1):5.2f} pairs/cap {pairs / cap:5.2f}   autograd vs f64 {wa[0]:.2e} ({wa[1]})   fused vs f64 {wf[0]:.2e} ({wf[1]})",
          flush=True)
