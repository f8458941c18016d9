"""dev diagnostic (GPU): what a k_assign16 wavefront (32 consecutive compacted rows) evaluates on the bench frame -- bones with >= 1
valid sample (the assignment net runs for those, for all 32 rows) and bones whose features they need (those + tree neighbours) --
against the (row, valid bone) pairs it serves; for the compaction order the cull kernel produces and for candidate re-orderings."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from core import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
PARENT = [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
nb = [1 << j for j in range(24)]
for j, p in enumerate(PARENT):
    if j != p:
        nb[j] |= 1 << p
        nb[p] |= 1 << j
NB = torch.tensor(nb, device=dev, dtype=torch.int64)


def popcount(x):
    c = torch.zeros_like(x)
    for j in range(24):
        c += (x >> j) & 1
    return c


def stats(name, bits_rows, group=32):
    n = bits_rows.shape[0]
    if group != 32:        # evaluated (row, bone) slots per valid pair if a wavefront served `group` rows instead of 32
        pad = (-n) % group
        bg = torch.cat([bits_rows, bits_rows.new_zeros(pad)]).view(-1, group)
        g = torch.zeros(bg.shape[0], dtype=torch.int64, device=dev)
        for k in range(group):
            g |= bg[:, k]
        print(f"{name:42s} groups of {group} rows: bones per group {popcount(g).float().mean():.2f}, evaluated slots / valid pairs = "
              f"{popcount(g).sum().item() * group / popcount(bits_rows).sum().item():.2f}")
        return
    pad = (-n) % 32
    b = torch.cat([bits_rows, bits_rows.new_zeros(pad)]).view(-1, 32)
    gnn = torch.zeros(b.shape[0], dtype=torch.int64, device=dev)
    for k in range(32):
        gnn |= b[:, k]
    feat = torch.zeros_like(gnn)
    for j in range(24):
        feat |= torch.where(((gnn >> j) & 1) == 1, NB[j], torch.zeros_like(gnn))
    pairs_feat = torch.zeros_like(gnn)
    for i in range(12):
        pairs_feat += (((feat >> (2 * i)) & 3) != 0).long()
    valid_pairs = popcount(bits_rows).sum().item()
    print(f"{name:42s} rows {n}  valid pairs/row {valid_pairs / n:.2f}  bones evaluated per wavefront {popcount(gnn).float().mean():.2f} "
          f"(x32 rows / valid pairs = {popcount(gnn).sum().item() * 32 / valid_pairs:.2f})  feature bone-PAIRS per wavefront "
          f"{pairs_feat.float().mean():.2f}  feature bones {popcount(feat).float().mean():.2f}")


for S, Sf, box in ((48, 16, False),):
    eng, inp, _ = bench.build_workload(dev, view=0)
    eng.cfg["use_volume_near_far"] = box
    out = eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], S, Sf, keep=True)
    for tag, z in (("coarse", out["z_coarse"]), ("fine", out["z_fine"])):
        geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z)
        bits, lst, cnt = ops.bone_cull(geo, True)
        n = int(cnt.item())
        rows = lst[:n].long()
        b = bits[rows].long() & 0xFFFFFF
        print(f"--- {S}+{Sf} box={box} {tag}")
        stats("cull order (as launched)", b)
        for grp in (16, 8):
            stats("cull order (as launched)", b, group=grp)
        stats("ray-major (sorted sample index)", bits[torch.sort(rows).values].long() & 0xFFFFFF)
        low = torch.full_like(b, 24)
        for j in reversed(range(24)):
            low = torch.where(((b >> j) & 1) == 1, torch.full_like(b, j), low)
        stats("stable sort by lowest valid bone", b[torch.sort(low, stable=True).indices])
        rest = b & ~(1 << low)
        second = torch.full_like(b, 24)
        for j in reversed(range(24)):
            second = torch.where(((rest >> j) & 1) == 1, torch.full_like(b, j), second)
        stats("stable sort by (lowest, second lowest)", b[torch.sort(low * 25 + second, stable=True).indices])
        stats("sort by the whole bit set", b[torch.sort(b, stable=True).indices])
        for win in (1024, 4096, 16384):
            pad = (-n) % win
            bw = torch.cat([b, b.new_full((pad,), 1 << 30)]).view(-1, win)
            bs = torch.sort(bw, dim=1, stable=True).values.reshape(-1)
            stats(f"cull order, sorted by bit set inside windows of {win}", bs[bs < (1 << 30)])
            key = torch.cat([low * 25 + second, low.new_full((pad,), 1 << 30)]).view(-1, win)
            idx = torch.sort(key, dim=1, stable=True).indices
            bs = torch.gather(bw, 1, idx).reshape(-1)
            stats(f"   ... by (lowest, second) inside windows of {win}", bs[bs < (1 << 30)])
