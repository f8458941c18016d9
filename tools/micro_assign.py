"""micro-benchmark of the assignment kernels on the bench workload (dev tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
import torch, numpy as np
import bench
from core import hip_ops as ops

eng, inp, _ = bench.build_workload(torch.device("cuda:0"), 0)
eng.refresh()
near, far = eng.near_far(inp["rays_o"], inp["rays_d"], inp["cyls"], inp["skts"])
z = ops.coarse_samples(near, far, 48)
vols = eng.volumes(inp["bones"])
geo = ops.Geometry(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, z=z)
bits, lst, cnt = ops.bone_cull(geo, True)
n = int(cnt.item())
print("rows", n)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print("assign16        ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lst, cnt, geo.M)))
print("assign16 n-host ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lst, None, n)))
zero = torch.zeros_like(bits)
print("assign16 nobits ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, zero, eng.aw, eng.assign16, lst, None, n)))
print("assign v1       ms", timeit(lambda: ops.gather_assign_blend(geo, vols, bits, eng.aw, lst, None, n)))
# sorted list (spatially coherent rows)
ls = torch.sort(lst[:n]).values.contiguous()
print("assign16 sorted ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, ls, None, n)))
h = torch.zeros(n, 16, device="cuda")
raw = torch.zeros(geo.M, 4, device="cuda")
cview, raw_empty = eng.view_constants(inp["rays_d"], inp["skts"], inp["cam_idx"])
print("mlp16           ms", timeit(lambda: ops.pe_mlp16(h, 48, eng.packed16, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, raw, ls, None, n, form=eng.mlp_form)))
print("mlp16 cap=M     ms", timeit(lambda: ops.pe_mlp16(h, 48, eng.packed16, eng.pts_b, eng.alpha_w, eng.alpha_b, cview, eng.rgb_w, eng.rgb_b, raw, lst, cnt, geo.M, form=eng.mlp_form)))

# grouped rows (k_group.hip) and the s_memtime trace of one wavefront
lg = lst.clone()
print("group_rows      ms", timeit(lambda: ops.group_rows(bits, lg, cnt)))
print("assign16 grouped ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, lg, cnt, geo.M)))
# reference orders made with torch (same box, same run): by the whole bit set / by (lowest, second lowest) inside windows of 16 384
b = bits[lst[:n].long()].long() & 0xFFFFFF
W = 16384
pad = (-n) % W
def windowed(key):
    k = torch.cat([key, key.new_full((pad,), 1 << 40)]).view(-1, W)
    idx = torch.sort(k, dim=1, stable=True).indices + (torch.arange(k.shape[0], device=k.device) * W)[:, None]
    idx = idx.reshape(-1)
    return lst[:n][idx[idx < n]].contiguous()
low = torch.full_like(b, 24)
for j in reversed(range(24)):
    low = torch.where(((b >> j) & 1) == 1, torch.full_like(b, j), low)
rest = b & ~(1 << low)
second = torch.full_like(b, 24)
for j in reversed(range(24)):
    second = torch.where(((rest >> j) & 1) == 1, torch.full_like(b, j), second)
for name, key in (("whole set", b), ("(low, second)", low * 25 + second), ("lowest", low)):
    l_ = windowed(key)
    print(f"assign16 torch-sorted by {name:14s} ms", timeit(lambda: ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, l_, None, n)))
print("same rows as the kernel's order:", bool(torch.equal(torch.sort(lg[:n]).values, torch.sort(lst[:n]).values)),
      " kernel order == torch (low, second) order:", bool(torch.equal(lg[:n], windowed(low * 25 + second))))
if "--trace" in sys.argv:
    import ctypes
    from core import _hip
    for name, l_ in (("cull order", lst), ("grouped", lg)):
        buf = torch.zeros(256, dtype=torch.int64, device="cuda")
        _hip.lib().danbo_assign16_set_trace(ctypes.c_void_p(buf.data_ptr()))
        ops.gather_assign_blend16(geo, vols, bits, eng.aw, eng.assign16, l_, cnt, geo.M)
        torch.cuda.synchronize()
        _hip.lib().danbo_assign16_set_trace(None)
        t = buf.cpu().numpy().reshape(-1, 2)
        t = t[t[:, 1] != 0]
        print("trace", name, "(tag, delta of s_memtime ticks [100 MHz: 10 ns each])")
        prev = None
        line = []
        for tag, tm in t:
            if tag == 0 and line:
                print("   ", " ".join(line)); line = []
            line.append(f"{tag}:{'' if prev is None else tm - prev}")
            prev = tm
        print("   ", " ".join(line))

# cost model of the round-4 kernel per wavefront (s_memtime trace: ~2 600 ticks per feature pair-step, ~1 800 per bone for the three
# layers, ~1 500 per tile of fixed work) summed along each wavefront's tile sequence: average vs slowest wavefront, for each order
if "--model" in sys.argv:
    DEG = torch.tensor([3, 2, 2, 2, 2, 2, 2, 2, 2, 4, 1, 1, 2, 2, 2, 1, 2, 2, 2, 2, 2, 2, 1, 1], device="cuda")
    bone_cost = ((DEG + 2) // 2) * 2600 + 1800           # ceil((deg + 1) / 2) pair-steps
    NWG = 512
    def model(name, l_, RUN=4, sched=None):
        bb = bits[l_[:n].long()].long() & 0xFFFFFF
        padr = (-n) % 128
        bw = torch.cat([bb, bb.new_zeros(padr)]).view(-1, 4, 32)           # [tile, wave, row]
        need = torch.zeros(bw.shape[:2], dtype=torch.int64, device="cuda")
        for k in range(32):
            need |= bw[:, :, k]
        cost = torch.full(need.shape, 1500, dtype=torch.int64, device="cuda")
        nb = torch.zeros_like(need)
        for j in range(24):
            on = (need >> j) & 1
            cost += on * bone_cost[j]
            nb += on
        ntiles = cost.shape[0]
        wg = (torch.arange(ntiles, device="cuda") // RUN) % NWG
        if sched is not None:
            wg = sched(ntiles)
        per_wave = torch.zeros(NWG, 4, dtype=torch.int64, device="cuda").index_add_(0, wg, cost)
        lock = cost.max(1).values.float().sum().item() / NWG
        print(f"model RUN={RUN} {name:28s} dynamic WG tickets, waves in lockstep per tile: {lock:.0f} ticks (+ <= one tile {cost.max().item()}) |  bones/wave {nb.float().mean():.2f}  ticks: mean wavefront {per_wave.float().mean():.0f}  slowest {per_wave.max().item()}  "
              f"(x {per_wave.max().item() / per_wave.float().mean():.2f})")
    for RUN in (1, 2, 4):
        model("cull order", lst, RUN)
        model("kernel grouped (low, second)", lg, RUN)
        model("torch whole set", windowed(b), RUN)
    for name, key in (("whole set", b), ("(low, second)", low * 25 + second), ("lowest", low)):
        model("torch " + name, windowed(key))

    import math
    def golden(ntiles, nwg=512):
        P = int(ntiles * 0.6180339887) | 1
        while math.gcd(P, ntiles) != 1:
            P += 2
        visit = (torch.arange(ntiles, device="cuda") * P) % ntiles          # i-th visited tile
        wg = torch.empty(ntiles, dtype=torch.int64, device="cuda")
        wg[visit] = torch.arange(ntiles, device="cuda") % nwg
        return wg
    for nm, l_ in (("cull order", lst), ("kernel grouped", lg), ("torch whole set", windowed(b))):
        model(nm + " stride 509", l_, 1, lambda nt: torch.arange(nt, device="cuda") % 509)
        model(nm + " golden perm", l_, 1, golden)
    # greedy list scheduling (what the ticket counter does): 512 workgroups, tiles in list order, chunks of 1 / 2 / 4 tiles per draw
    import heapq
    def simulate(name, l_):
        bb = bits[l_[:n].long()].long() & 0xFFFFFF
        padr = (-n) % 128
        bw = torch.cat([bb, bb.new_zeros(padr)]).view(-1, 4, 32)
        need = torch.zeros(bw.shape[:2], dtype=torch.int64, device="cuda")
        for k in range(32):
            need |= bw[:, :, k]
        cost = torch.full(need.shape, 1500, dtype=torch.int64, device="cuda")
        for j in range(24):
            cost += ((need >> j) & 1) * bone_cost[j]
        tile_cost = cost.max(1).values.cpu().tolist()
        out = []
        for chunk in (1, 2, 4):
            heap = [0] * NWG
            for i in range(0, len(tile_cost), chunk):
                t = heapq.heappop(heap)
                heapq.heappush(heap, t + sum(tile_cost[i:i + chunk]))
            out.append(f"chunk {chunk}: {max(heap)}")
        print(f"greedy {name:28s} sum/512 = {sum(tile_cost) / NWG:.0f}  makespan " + "  ".join(out))
    simulate("cull order", lst)
    simulate("kernel grouped", lg)
    simulate("torch whole set", windowed(b))
