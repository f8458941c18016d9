#!/usr/bin/env python3
"""Benchmark of the MI355X DANBO path.  `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.

Default workload = BASELINE.json's metric: ray-samples/s of the render path at 512 x 512 rays x 64 samples (48 coarse + 16
importance: the reference cannot run N_importance = 0), H36M `danbo_base` network, seeded synthetic weights / pose /
camera, everything resident in HBM.  A *step* is one full frame: bounds -> depths -> cull -> gather / assignment / blend ->
PE + MLP -> composite -> importance resampling -> second network pass -> composite.  For N > 1 there is one rank (process) per
GPU: either the caller started them (RANK / WORLD_SIZE / LOCAL_RANK in the environment, e.g. the driver's
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`), or -- `python bench.py --gpus N` from a plain shell --
bench.py starts them itself (launch_ranks: a `torch.distributed.run` child, BEFORE this process touches the GPU) and relays rank
0's JSON line.  Ranks render different camera views (rays shard data-parallel, no collective on the data path) => weak scaling;
value = N * K * units / max-over-ranks time.  `ranks_seen` = dist.get_world_size() as the ranks saw it.

--config selects the other BASELINE configurations, each with its own `roofline` and `cpu_baseline`:
  2  H36M danbo_fast: 512^2 x (32 + 16), per-bone box near/far           (render)
  3  H36M danbo_base: 512^2 x (96 + 32) = 128 samples per ray              (render)
  4  PerfCap danbo_fast TRAINING step: 3072 rays = 16 poses x 192, 32 + 16, perturb, noise, L1, Adam; danbo_train_step +
     danbo_adam_step (one C call each), RCCL all-reduce of the flat gradient for N > 1.  --scaling weak (default): every rank
     its own 3072 rays; --scaling strong: the reference's ONE 16-pose / 3072-ray batch split by whole poses over the ranks
     (8 ranks x 2 poses x 192 rays), value = 3072 x 48 / step time
  5  A-NeRF anerf_base: 512^2 x (48 + 16), cutoff PE, W = 448             (render)

`value` of the render configs is measured with exact in-volume culling (bit-identical raw to evaluating every sample:
`dense_equals_culled`); `dense_value` pushes every sample through every kernel (the reference's executed work).
`roofline` is for the dominant kernel, EXECUTED flops only, its launch durations taken from HIP events inside the timed region.
`roofline.traffic` is filled from the newest profiles/rNN_pmc_hbm.json that was measured on the kernel sources being timed
(sha recorded by tools/pmc_hbm.sh), else null.  `cpu_baseline`: oracle/torch_cpu.py (a multi-threaded torch-CPU restatement
doing the reference's executed work) on a bounded sample of the same workload, best of 3.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

H = W = 512
N_SAMPLES, N_IMPORTANCE = 48, 16
MAC_PER_ROW_REF = 677376       # reference network: trunk 558592 + alpha 256 + feature 65536 + view 52608 + rgb 384
# executed by each kernel (per-ray part of the view layer is hoisted into k_view_consts in both):
MAC_PER_ROW = {"fp32": 558592 + 256 + 65536 + 32768 + 384,      # k_pe_mlp : trunk, alpha, feature, view[:, :256], rgb
               "f16split": 558592 + 256 + 32768 + 384}          # k_pe_mlp32 / k_pe_mlp16: feature+view merged into one 256->128
PEAK_FP32_MFMA = 157.3e12      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_FP16_MFMA = 2500e12       # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (2:1-sparse figures are not used)
SPLIT_NOTE = ("dense fp16 MFMA peak 2500 TFLOP/s / 3: every fp32-accurate product is three half-precision MFMA products "
              "(hi*hi + hi*lo + lo*hi); executed MFMA rate = 3 x achieved")
CONFIGS = {1: (48, 16, False), 2: (32, 16, True), 3: (96, 32, False)}      # (coarse, importance, per-bone box bounds)


def T(x, device, dt=torch.float32):
    return torch.tensor(np.ascontiguousarray(x), dtype=dt, device=device)


def build_workload(device, view, mlp_mode="f16split", cam_dist=3.0):
    from core.render_engine import DanboEngine
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import bone_align_transforms
    cfg = syn.model_config("danbo_base")
    rest = syn.rest_pose(cfg["rest_scale"])
    sd = syn.make_state_dict(cfg, seed=0, n_framecodes=100, rest=rest)
    scene = syn.make_scene(n_poses=1, H=H, W=W, n_views=8, pose_seed=0, min_radius=1.25, cam_dist=cam_dist)
    ro, rd = scene["rays"][view % 8]
    align = bone_align_transforms(rest)
    eng = DanboEngine(cfg, {k: T(v, device) for k, v in sd.items()}, T(align, device), mlp_mode=mlp_mode)
    inputs = dict(rays_o=T(ro, device), rays_d=T(rd, device), skts=T(scene["skts"], device), bones=T(scene["bones"], device),
                  cyls=T(scene["cyls"], device), cam_idx=torch.zeros(len(ro), dtype=torch.int64, device=device))
    return eng, inputs, (cfg, sd, rest, scene, ro, rd)


def render(eng, inp, dense=False):
    return eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"],
                      N_SAMPLES, N_IMPORTANCE, chunk=4096, dense=dense)


K3_SOURCES = ("k_mlp32.hip", "mlp32_regs.inc", "k_mlp16.hip", "mlp16_core.hpp", "common.hpp")     # tools/pmc_hbm.sh hashes the same list


def pmc_record(kind, *files):
    """the newest profiles/rNN_pmc_<kind>.json whose recorded hash equals the hash of the kernel sources being timed, or None: a
    traffic figure is only quoted for the code it was measured on"""
    import glob
    want = sha16(*files)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{kind}.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except ValueError:
            continue
        if rec.get("kernel_src_sha16") == want:
            rec["_file"] = os.path.relpath(path, ROOT)
            return rec
    return None


def sha16(*files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "danbo-pytorch_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------- CPU baselines (checker code)
def best_of(fn, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


PARITY_BOUND = 1e-4         # north_star: RGB / sigma logits within 1e-4 relative of the reference
MAPS_BOUND = 5e-5           # final rgb / acc maps of the TIMED frame against the oracle's (measured 1e-6 .. 1e-5: resampled depths amplify round-off)
BOUNDS_BOUND = 2e-6         # near / far against the oracle (the box bounds are bit-equal to the reference's; the cylinder's nan-mean
                            # back-fill is a sum in another order)


def raw_measures(raw, ref):
    """(floored, un-floored): max |a - b| / max(|b|, 5 % of the channel's largest |b|) -- tests/helpers.raw_err's measure -- and the
    max relative error without a floor over the entries above 10 % of their channel's largest |b|"""
    cmax = np.abs(ref).reshape(-1, 4).max(0)
    big = np.abs(ref) > 0.1 * cmax
    d = np.abs(raw - ref)
    return float((d / np.maximum(np.abs(ref), 0.05 * cmax)).max()), float((d / np.maximum(np.abs(ref), 1e-30))[big].max())


def parity_block(model, extra, sl, ref, frame, stages, eng_inp=None):
    """The HIP path against oracle/torch_cpu.py on the rays `sl` of the bench frame (checker code; bench.py's `parity`, and
    tests/test_gpu_benchframe.py asserts the same dictionary).  Coarse-pass logits are compared TWICE: at the HIP path's own depths
    (`stages`: the timed frame's near / far / raw / in-volume words) and -- `eng_inp` = (engine, inputs) -- with the ORACLE's near /
    far fed to the HIP path; one ulp of a bound moves every sample of the ray, and this network turns 5e-7 of depth into 5e-4 of a
    logit (tools/diag/config2_bounds_attribution.py), so the first comparison is only meaningful where the bounds are bit-equal --
    `bounds_bit_equal_rays` says for how many rays they are."""
    import danbo_oracle as o
    import torch_cpu
    from core.utils import synthetic as syn
    cfg, sd, rest, scene, ro, rd = extra
    cfg = model.cfg
    n = sl.stop - sl.start
    rgb, acc = frame["rgb_map"][sl].cpu().numpy(), frame["acc_map"][sl].cpu().numpy()
    parity = dict(against="oracle/torch_cpu.py on the cpu_baseline sample", rays=n, psnr_rgb_db=float(o.psnr(rgb, ref["rgb_map"])),
                  max_abs_rgb=float(np.abs(rgb - ref["rgb_map"]).max()), max_abs_acc=float(np.abs(acc - ref["acc_map"]).max()))
    if stages is None:
        return parity
    S = ref["raw_coarse"].shape[1]
    bits = stages["valid_bits"].view(H * W, S)[sl].cpu().numpy().astype(np.uint32)
    valid = ((bits[..., None] >> np.arange(24, dtype=np.uint32)) & 1).astype(bool)
    rr = ref["raw_coarse"]
    # the same restatement in float64 on the float32 points and mask: the exact result of the reference's graph
    m64 = torch_cpu.DanboTorchCPU(cfg, sd, rest, dtype=torch.float64)
    t64 = lambda v: torch.tensor(np.ascontiguousarray(v)).double()  # noqa: E731
    rb_all = syn.ray_batch(ro[sl], rd[sl])
    raw64, a0 = [], 0
    for nrows, pose_of_ray, bones_g in ref["chunks"]:
        c = slice(a0, a0 + nrows)
        z0 = np.zeros(nrows, dtype=np.int64)
        raw64.append(m64.forward(t64(ref["pts_coarse"][c]), t64(rb_all[c, 3:6]), t64(scene["skts"][z0]), m64._volumes(bones_g),
                                 torch.as_tensor(pose_of_ray), np.zeros(nrows, np.int64), valid=torch.as_tensor(ref["valid_coarse"][c])).numpy())
        a0 += nrows
    r64 = np.concatenate(raw64)
    near_o, far_o = ref["near"][:, 0], ref["far"][:, 0]
    sides = {}
    if "near" in stages:
        near_g, far_g = stages["near"][sl].cpu().numpy(), stages["far"][sl].cpu().numpy()
        same = (near_g == near_o) & (far_g == far_o)
        parity.update(max_abs_near=float(np.abs(near_g - near_o).max()), max_abs_far=float(np.abs(far_g - far_o).max()),
                      bounds_bit_equal_rays=int(same.sum()))
    sides["own_depths"] = (stages["raw_coarse"][sl].cpu().numpy(), valid)
    if eng_inp is not None:       # the HIP path at the oracle's bounds
        eng, inp = eng_inp
        dev = inp["rays_o"].device
        nf = (torch.tensor(near_o, device=dev), torch.tensor(far_o, device=dev))
        k2 = eng.render(inp["rays_o"][sl], inp["rays_d"][sl], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"][sl], S,
                        stages["Sf"], chunk=4096, near_far=nf, keep=True)
        b2 = k2["valid_bits"].view(n, S).cpu().numpy().astype(np.uint32)
        sides["oracle_depths"] = (k2["raw_coarse"].cpu().numpy(), ((b2[..., None] >> np.arange(24, dtype=np.uint32)) & 1).astype(bool))
    worst = 0.0
    for tag, (raw, v) in sides.items():
        f32_fl, f32_un = raw_measures(raw, rr)
        f64_fl, f64_un = raw_measures(raw, r64)
        parity[tag] = dict(mask_mismatches=int((v != ref["valid_coarse"]).sum()), max_rel_raw_floored_5pct=f32_fl, max_rel_raw=f32_un,
                           max_rel_raw_floored_5pct_vs_float64=f64_fl, max_rel_raw_vs_float64=f64_un)
        worst = max(worst, f32_fl, f64_fl)
    # the measure that decides: the HIP path at the oracle's depths when it was run, else at its own
    key = "oracle_depths" if "oracle_depths" in sides else "own_depths"
    decided = parity[key]
    ok = (decided["mask_mismatches"] == 0 and decided["max_rel_raw_floored_5pct"] <= PARITY_BOUND
          and decided["max_rel_raw_floored_5pct_vs_float64"] <= PARITY_BOUND)
    if "max_abs_near" in parity:
        ok = ok and parity["max_abs_near"] <= BOUNDS_BOUND and parity["max_abs_far"] <= BOUNDS_BOUND
        if parity["bounds_bit_equal_rays"] == n:      # bit-equal bounds: the own-depths comparison is the same statement, hold it too
            own = parity["own_depths"]
            ok = ok and own["mask_mismatches"] == 0 and own["max_rel_raw_floored_5pct"] <= PARITY_BOUND \
                and own["max_rel_raw_floored_5pct_vs_float64"] <= PARITY_BOUND
    # ... and the TIMED frame itself (`decided` may be the re-run at the oracle's bounds): its logits on the rays whose bounds
    # are bit-equal to the oracle's -- there the own-depths comparison is the same statement -- and its final maps on every ray
    timed_ok = parity["max_abs_rgb"] <= MAPS_BOUND and parity["max_abs_acc"] <= MAPS_BOUND
    if "near" in stages:
        if same.any():
            own_raw = sides["own_depths"][0]
            t32, _ = raw_measures(own_raw[same], rr[same])
            t64, _ = raw_measures(own_raw[same], r64[same])
            parity["timed_frame_bit_equal_bound_rays"] = dict(rays=int(same.sum()), max_rel_raw_floored_5pct=t32, max_rel_raw_floored_5pct_vs_float64=t64,
                                                              mask_mismatches=int((valid[same] != ref["valid_coarse"][same]).sum()))
            timed_ok = timed_ok and t32 <= PARITY_BOUND and t64 <= PARITY_BOUND and parity["timed_frame_bit_equal_bound_rays"]["mask_mismatches"] == 0
        timed_ok = timed_ok and int(same.sum()) >= n // 2      # a frame most of whose bounds differ is not the oracle's frame
    parity.update(parity_ok_timed_frame=bool(timed_ok), maps_bound=MAPS_BOUND)
    ok = ok and timed_ok
    parity.update(mask_entries=int(valid.size), restatement_fp32_vs_float64_floored_5pct=raw_measures(rr, r64)[0],
                  # top-level copies of the deciding comparison (the names earlier rounds' lines carry)
                  mask_mismatches=decided["mask_mismatches"], max_rel_raw=decided["max_rel_raw"],
                  max_rel_raw_floored_5pct=decided["max_rel_raw_floored_5pct"],
                  max_rel_raw_vs_float64=decided["max_rel_raw_vs_float64"],
                  max_rel_raw_floored_5pct_vs_float64=decided["max_rel_raw_floored_5pct_vs_float64"],
                  decided_by=key, bound=PARITY_BOUND, bounds_bound=BOUNDS_BOUND, parity_ok=bool(ok),
                  raw_note="coarse-pass logits; *_floored_5pct = max |a - b| / max(|b|, 5 % of the channel's largest |b|) "
                           "(tests/helpers.raw_err), max_rel_raw = no floor, entries with |b| > 10 % of the channel's largest; b = the "
                           "float32 CPU restatement, *_vs_float64: the same restatement in float64 on the float32 points and mask; "
                           "own_depths: the timed frame (its own near / far), oracle_depths: the HIP path fed the oracle's near / far")
    return parity


def cpu_baseline_render(extra, box_bounds, frame=None, stages=None, budget_s=6.0, eng_inp=None):
    """oracle/torch_cpu.py on centre rays of the same frame; the sample is grown until one repetition takes ~budget_s"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import danbo_oracle as o
    import torch_cpu
    cfg, sd, rest, scene, ro, rd = extra
    from core.utils import synthetic as syn
    cfg = dict(cfg, use_volume_near_far=box_bounds)
    model = torch_cpu.DanboTorchCPU(cfg, sd, rest)

    def run(n_rays):
        r0 = (H // 2) * W - n_rays // 2
        sl = slice(r0, r0 + n_rays)
        z = np.zeros(n_rays, dtype=np.int64)
        rb = syn.ray_batch(ro[sl], rd[sl])
        return sl, model.render(rb, scene["skts"][z], scene["bones"][z], scene["cyls"][z], np.zeros(n_rays, np.int64), 1, N_SAMPLES,
                                N_IMPORTANCE, stages=stages is not None)
    n = 4096
    run(512)                                                    # thread pools, allocator
    # torch's default of one thread per hardware thread is far from the fastest setting for 4096-ray chunks on a 128-core
    # host: use the thread count that is fastest on a probe, and report it
    best_t, best_dt = torch.get_num_threads(), None
    for nt in sorted({8, 16, 32, 64, torch.get_num_threads()}):
        if nt > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        run(1024)
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best_t, best_dt = nt, dt
    torch.set_num_threads(best_t)
    t0 = time.perf_counter()
    run(n)                                                      # a first estimate at the full chunk size
    est = time.perf_counter() - t0
    while est * 2 < budget_s and n * 2 <= 65536:
        n, est = n * 2, est * 2
    dt, (sl, ref) = best_of(lambda: run(n))
    parity = None
    if frame is not None:   # the timed HIP path's image on the same rays (checker only: nothing here is timed or shipped)
        parity = parity_block(model, extra, sl, ref, frame, stages, eng_inp)
    return dict(value=n * (N_SAMPLES + N_IMPORTANCE) / dt, unit="ray-samples/s", cores=int(torch.get_num_threads()),
                host_cpu_count=os.cpu_count(), kind="port",
                sample=f"{n} centre rays x {N_SAMPLES}+{N_IMPORTANCE} samples of the same frame, every sample through every bone and "
                       f"the full MLP (the reference's executed work), oracle/torch_cpu.py (torch CPU kernels, "
                       f"{torch.get_num_threads()} threads), 4096-ray chunks, best of 3: {dt:.2f} s"), parity


# ---------------------------------------------------------------------------------------------- timing harness
N_BLOCKS = 5               # timed blocks of --steps steps each; ms_per_step = their MEDIAN
RANK_TIMES = []            # per block (settle blocks included): the seconds every rank measured for it
SETTLE_MIN_S, SETTLE_MAX_S, SETTLE_TOL = 0.5, 8.0, 0.02


def _sync():
    if torch.cuda.is_available():          # (--dry-run runs the same harness where there is no GPU)
        torch.cuda.synchronize()


def _block(fn, steps, dist, device, cpu_dist, before=None):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides -> (seconds, MAX over the ranks; last output).
    before(): untimed preparation of the block (the training bench restores its weights there)"""
    if before is not None:
        before()
    _sync()
    if dist is not None:
        dist.barrier()
    _sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = fn()
    _sync()
    own = time.perf_counter() - t0          # this rank's steps alone, before it waits for the others
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        # what counts: the MAX over the ranks of the bracketed time; beside it every rank's OWN time (before the closing barrier),
        # gathered so that the line shows which rank was slow
        mine = torch.tensor([elapsed, own], device="cpu" if cpu_dist else device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        RANK_TIMES.append([float(x[1].item()) for x in every])
        elapsed = max(float(x[0].item()) for x in every)
    return elapsed, out


def timed(fn, steps, warmup, dist, device, cpu_dist, settle_block=10, on_timed_start=None, before_block=None, after_block=None):
    """The number must not depend on where in the process it is taken (the first tens of milliseconds after idle
    run ~10 % slower -- clocks, caches, the allocator's pool -- and a 0.12 s timed region sat inside them):
      1. settle: blocks of `settle_block` steps until >= SETTLE_MIN_S of work has run BEHIND THE FIRST BLOCK (which carries the
         one-time initialisations: round 4 saw it take 1 s and satisfy the criterion alone) AND two consecutive blocks agree within
         SETTLE_TOL (give up after SETTLE_MAX_S and say so); block times are max-over-ranks, so every rank takes the same
         decisions and runs the same number of steps (the training step contains collectives);
      2. the CLI's `warmup` untimed steps;
      3. N_BLOCKS timed blocks of EXACTLY `steps` steps, each bracketed by barrier + synchronize, each the MAX over the ranks.
    -> (median block seconds, last output, info for the JSON line)"""
    spent, steady, prev, settled, n_settle = 0.0, 0.0, None, False, 0
    settle_ms = []
    while spent < SETTLE_MAX_S:
        dt, _ = _block(fn, settle_block, dist, device, cpu_dist)
        spent += dt
        settle_ms.append(1e3 * dt / settle_block)
        if n_settle > 0:
            steady += dt            # the first block carries the one-time initialisations (a second of them on a fresh box): not "work"
        n_settle += 1
        if prev is not None and steady >= SETTLE_MIN_S and abs(dt - prev) <= SETTLE_TOL * max(dt, prev):
            settled = True
            break
        prev = dt
    if before_block is not None:
        before_block()
    for _ in range(warmup):
        fn()
    if on_timed_start is not None:
        on_timed_start()
    blocks, out = [], None
    for _ in range(N_BLOCKS):
        dt, out = _block(fn, steps, dist, device, cpu_dist, before=before_block)
        blocks.append(dt)
        if after_block is not None:
            after_block(out)
    med = float(np.median(blocks))
    per_rank = {}
    if dist is not None and len(RANK_TIMES) >= N_BLOCKS:      # the timed blocks are the last N_BLOCKS entries
        last = np.array(RANK_TIMES[-N_BLOCKS:])               # [block, rank]
        per_rank = dict(ms_per_step_ranks=[float(x) for x in 1e3 * np.median(last, 0) / steps],
                        slowest_rank_per_block=[int(i) for i in last.argmax(1)])
    info = dict(block_ms=[1e3 * b / steps for b in blocks], spread=(max(blocks) - min(blocks)) / med, settle_ms=settle_ms, **per_rank,
                timing=f"median of {N_BLOCKS} blocks of --steps steps (each: barrier + synchronize on both sides, max over ranks) after "
                       f"a settle phase of {n_settle} x {settle_block} steps = {spent:.2f} s "
                       f"({'two consecutive blocks within 2 %' if settled else 'NOT settled within %.0f s' % SETTLE_MAX_S}) and "
                       f"--warmup untimed steps")
    return med, out, info


def ops_mod():
    from core import hip_ops
    return hip_ops


# ---------------------------------------------------------------------------------------------- render configs 1-3
def bench_render(args, rank, world, device, dist):
    eng, inp, extra = build_workload(device, view=rank, mlp_mode=args.mlp)
    eng.cfg["use_volume_near_far"] = bool(args.box_near_far)

    def start_profile():                    # HIP events around K3 from the first timed block on (settle / warm-up frames carry none)
        # The first block that records them pays for the runtime's pool of profiling signals (4 timing events per frame: the first
        # timed block read 5.1 - 7.6 ms per frame against 4.7 for the other four, on every lease; recording events alone, or two
        # instrumented frames, did not cure it: tools/diag/first_block.py).  One block's worth of instrumented frames runs -- and
        # is waited for, its events dropped -- in front of the timed region: the measurement's own instrumentation is not frame work.
        eng.profile = {}
        for _ in range(args.steps):
            render(eng, inp)
        torch.cuda.synchronize()
        eng.profile = {}
    elapsed, out, tinfo = timed(lambda: render(eng, inp), args.steps, args.warmup, dist, device, args.debug_single_device,
                                on_timed_start=start_profile)
    samples_per_frame = H * W * (N_SAMPLES + N_IMPORTANCE)
    value = world * args.steps * samples_per_frame / elapsed

    # ---- roofline of the dominant kernel from the HIP events recorded inside the timed region
    prof = eng.profile["k_pe_mlp"]
    eng.profile = None
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in prof)
    rows = sum(int(c.item()) if torch.is_tensor(c) else int(c) for _, _, c in prof)
    mac = MAC_PER_ROW[args.mlp]
    achieved = 2.0 * mac * rows / (ms * 1e-3)                # EXECUTED multiply-adds of in-volume rows only
    if args.mlp == "f16split":
        kernel, peak, peak_note = {16: "k_pe_mlp16", 32: "k_pe_mlp32"}[eng.mlp_form], PEAK_FP16_MFMA / 3.0, SPLIT_NOTE
    else:
        kernel, peak, peak_note = "k_pe_mlp", PEAK_FP32_MFMA, "fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)"
    traffic, traffic_note = None, "no PMC profile of this build of the kernel under profiles/ (tools/pmc_hbm.sh)"
    rec = pmc_record("hbm", *K3_SOURCES) if (args.mlp == "f16split" and args.config == 1) else None
    if rec is not None:
        traffic = rec["kernels"].get("danbo::" + kernel, {}).get("hbm_bytes_per_launch")
        traffic_note = f"HBM bytes per launch from rocprofv3 PMC passes of THIS kernel source ({rec['_file']})"
    # algorithmic HBM bytes of a launch: 84 B per row (64 B h + 4 B list entry + 16 B raw) + the 512-byte per-ray view constants of
    # every ray that owns >= 1 row (counted on the coarse pass of one extra, untimed frame)
    keep = eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], N_SAMPLES, N_IMPORTANCE,
                      chunk=4096, keep=True)
    rays_hit = int((keep["valid_bits"].view(H * W, N_SAMPLES) != 0).any(1).sum())
    algo_bytes = 84.0 * rows / len(prof) + 512.0 * rays_hit
    # what the exact sparsity of the frame amounts to (untimed): rays that cannot meet a bone volume between their bounds are rays of
    # constants when the weights allow it (include/danbo_hip.h: danbo_flat_rays) -- no view constants, resampling or composite for them
    rm = ops_mod().ray_bone_mask(inp["rays_o"], inp["rays_d"], inp["skts"], eng.align, eng.axis_scale, keep["near"], keep["far"], want_flat=True)
    sparsity = dict(rays=H * W, rays_with_a_candidate_bone=int((rm[0] != 0).sum()), rays_with_an_in_volume_coarse_sample=rays_hit,
                    rays_of_constants=int(rm[3].sum()) if (eng.flat_rays_ok and eng.skip_flat_rays and N_SAMPLES <= 256 and N_IMPORTANCE <= 64) else 0,
                    note="every output of the timed frame is compared bitwise with the render that evaluates every sample of every ray "
                         "(dense_equals_culled)")
    keep = dict(raw_coarse=keep["raw_coarse"], valid_bits=keep["valid_bits"], near=keep["near"], far=keep["far"],
                Sf=N_IMPORTANCE)                                                       # for the parity block below
    roofline = dict(bound="mfma", kernel=kernel, achieved=achieved / 1e12, peak=peak / 1e12, unit="TFLOP/s", frac=achieved / peak,
                    traffic=traffic, algorithmic_bytes=algo_bytes, launches=len(prof), avg_launch_ms=ms / len(prof),
                    rows_per_launch=rows / len(prof), flop_per_row=2 * mac, flop_per_row_reference=2 * MAC_PER_ROW_REF,
                    peak_note=peak_note,
                    note="executed flops of rows inside >=1 bone volume only; algorithmic bytes per launch = 84 B per row + 512 B of "
                         "per-ray view constants for every ray with >= 1 row; " + traffic_note)
    names = {1: "H36M danbo_base network", 2: "H36M danbo_fast (BASELINE config 2)", 3: "H36M danbo_base (BASELINE config 3)"}
    result = {
        "metric": f"ray-samples/sec at 512x512x{N_SAMPLES + N_IMPORTANCE} samples", "value": value, "unit": "ray-samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.mlp == "fp32" else "f32 (fp16x2-split MFMA products, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": f"{names.get(args.config, names[1])}, 512x512 rays x ({N_SAMPLES} coarse + {N_IMPORTANCE} importance) "
                               f"samples, 1 pose / 1 camera per rank, {'per-bone box' if args.box_near_far else 'cylinder'} near/far, "
                               "exact in-volume culling",
                   "rays": H * W, "samples_per_ray": N_SAMPLES + N_IMPORTANCE, "parallelism": f"rays-dp{world}"},
        "in_volume_fraction": rows / (N_BLOCKS * args.steps * samples_per_frame), "sparsity": sparsity,
        "roofline": roofline, **tinfo,
    }
    if rank == 0 and world == 1 and not args.no_api:
        result.update(bench_through_caster(args, device, extra, inp, out))
    if rank == 0 and world == 1:
        if not args.no_dense:
            render(eng, inp, dense=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nd = 2
            for _ in range(nd):
                out_d = render(eng, inp, dense=True)
            torch.cuda.synchronize()
            td = (time.perf_counter() - t1) / nd
            result["dense_value"] = samples_per_frame / td
            result["dense_ms_per_step"] = 1e3 * td
            result["dense_mlp_tflops_lower_bound"] = 2.0 * mac * samples_per_frame / td / 1e12
            # every output of the timed (culled, rays-of-constants) frame against the frame that evaluates every sample of every ray
            result["dense_equals_culled"] = bool(all(torch.equal(out_d[k], out[k]) for k in out if torch.is_tensor(out[k]) and k in out_d))
            result["dense_equals_culled_outputs"] = sorted(k for k in out if torch.is_tensor(out[k]) and k in out_d)
        if sparsity["rays_of_constants"] > 0:
            # the headline leans on a property of the weights (empty-space density <= 0 makes 63 % of the rays "rays of
            # constants").  The same frame with that shortcut OFF -- what a checkpoint whose PE(0) density is slightly positive costs:
            # every ray gets view constants, resampling and both composites (DanboEngine.skip_flat_rays; bit-identical outputs)
            eng.skip_flat_rays = False
            for _ in range(5):
                out_nf = render(eng, inp)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                out_nf = render(eng, inp)
            torch.cuda.synchronize()
            tnf = (time.perf_counter() - t1) / args.steps
            eng.skip_flat_rays = True
            result["ms_no_flat_rays"] = 1e3 * tnf
            result["value_no_flat_rays"] = samples_per_frame / tnf
            result["no_flat_rays_equals_timed"] = bool(all(torch.equal(out_nf[k], out[k]) for k in out if torch.is_tensor(out[k]) and k in out_nf))
            result["no_flat_rays_note"] = ("the same frame with DanboEngine.skip_flat_rays = False (one block of --steps frames): every ray is "
                                           "resampled and composited, as for weights whose empty-space density is positive")
        if not args.no_sweep:
            # the headline rides on how much of the frame the body fills: the same frame from nearer / farther cameras
            sweep = []
            for dist_ in (2.0, 3.0, 4.5):
                e2, i2, _ = build_workload(device, view=0, mlp_mode=args.mlp, cam_dist=dist_)
                e2.cfg["use_volume_near_far"] = bool(args.box_near_far)

                def start2(e=e2):
                    e.profile = {}
                dt, _, ti = timed(lambda: render(e2, i2), 10, 0, None, device, False, on_timed_start=start2)
                dt /= 10
                r2 = sum(int(c.item()) for _, _, c in e2.profile["k_pe_mlp"]) / (10 * N_BLOCKS)
                sweep.append(dict(camera_distance=dist_, in_volume_fraction=r2 / samples_per_frame, value=samples_per_frame / dt,
                                  ms_per_step=1e3 * dt, spread=ti["spread"]))
                del e2, i2
            result["occupancy_sweep"] = sweep
        if not args.no_cpu_baseline:
            result["cpu_baseline"], parity = cpu_baseline_render(extra, bool(args.box_near_far), frame=out, stages=keep, eng_inp=(eng, inp))
            if parity is not None:
                result["parity"] = parity
                result["parity_ok"] = parity.get("parity_ok")
    return result


def bench_through_caster(args, device, extra, inp, engine_out):
    """The same frame through the DROP-IN call path (missing 1): `core.trainer.render(H, W, focal, chunk=4096,
    rays=..., kp_batch=..., skts=..., **render_kwargs_test)` exactly as run_render.py / render_path call it (reference
    run_nerf.py:64-91 -> core/trainer.py:96-161 -> batchify_rays -> ray_caster(...)), on a caster built by create_raycaster from the
    shipped config file and loaded with the same weights.  -> api_ms_per_step / api_value (median of N_BLOCKS blocks of --steps
    frames, same bracketing as the headline) and whether its maps are bit-identical to the engine-level frame that was timed."""
    from core import trainer
    from core.config import parse_args
    from core.raycasters import create_raycaster
    from core.utils.skeleton_utils import SMPLSkeleton
    cfg, sd, rest, scene, ro, rd = extra
    cfg_file = "danbo_fast.txt" if args.config == 2 else "danbo_base.txt"
    targs = parse_args(["--no_reload"], config=os.path.join(ROOT, "danbo-pytorch_amd", "configs", "h36m_zju", cfg_file))
    da = dict(skel_type=SMPLSkeleton, near=0., far=100., n_views=100, rest_pose=rest, hwf=(H, W, 0.))
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the constructors print; stdout carries the JSON line only
        _, te_kw, *_ = create_raycaster(targs, da, device=device)
    caster = te_kw["ray_caster"].eval()
    caster.network.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    caster.use_volume_near_far = bool(args.box_near_far)
    kw = {k: v for k, v in te_kw.items() if k not in ("N_samples", "N_importance")}
    n = H * W
    exp = lambda x: x[:1].expand(n, *x.shape[1:])  # noqa: E731   (render_path's reuse_input(x, expand): one pose behind every ray)
    kps = torch.tensor(np.ascontiguousarray(scene["kps"]), dtype=torch.float32, device=device)
    call = dict(chunk=4096, rays=(inp["rays_o"], inp["rays_d"]), kp_batch=exp(kps), skts=exp(inp["skts"]), cyls=exp(inp["cyls"]),
                bones=exp(inp["bones"]), cams=inp["cam_idx"][:1].expand(n), N_samples=N_SAMPLES, N_importance=N_IMPORTANCE, **kw)

    def frame():
        return trainer.render(H, W, scene.get("focal", 0.), **call)
    elapsed, out, tinfo = timed(frame, args.steps, args.warmup, None, device, False)
    same = all(torch.equal(out[k].reshape(engine_out[k].shape), engine_out[k]) for k in engine_out if torch.is_tensor(engine_out[k]) and k in out)
    del caster
    return dict(api_ms_per_step=1e3 * elapsed / args.steps, api_value=args.steps * n * (N_SAMPLES + N_IMPORTANCE) / elapsed,
                api_spread=tinfo["spread"], api_equals_engine=bool(same),
                api_note="the same frame through core.trainer.render(H, W, focal, chunk=4096, rays=..., **render_kwargs_test) on a caster "
                         "from create_raycaster(configs/h36m_zju/" + cfg_file + "): the call run_render.py makes per image; the caster "
                         "takes the image's rays in one cast (RayCaster.render_rays_whole), bit-identical to the 64-chunk loop")


# ---------------------------------------------------------------------------------------------- config 4: training step
def bench_train(args, rank, world, device, dist):
    from core.config import parse_args
    from core.raycasters import create_raycaster
    from core.trainer import Trainer
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import SMPLSkeleton
    targs = parse_args(["--no_reload"], config=os.path.join(ROOT, "danbo-pytorch_amd", "configs", "perfcap", "danbo_fast.txt"))
    rest = syn.rest_pose(0.48)
    da = dict(skel_type=SMPLSkeleton, near=0., far=100., n_views=20, rest_pose=rest, hwf=(128, 128, 160.))
    torch.manual_seed(0)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the constructors print; stdout carries the JSON line only
        tr_kw, te_kw, start, grad_vars, opt, _ = create_raycaster(targs, da, device=device)
    caster = tr_kw["ray_caster"]
    sd = syn.make_state_dict(syn.model_config("danbo_perfcap"), 3, 20, rest)
    caster.network.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    trainer = Trainer(targs, da, opt, None, tr_kw, te_kw, device=device)
    trainer.collectives_at_world_1 = bool(args.nccl_world_1)
    strong = args.scaling == "strong"
    poses, rpp = 16, 192
    if strong and (poses % world != 0):
        raise SystemExit(f"--scaling strong splits the 16-pose batch by whole poses: {world} ranks do not divide 16")
    # weak: every rank draws its own 16-pose batch; strong: every rank builds the SAME batch and keeps its 16 / world poses
    scene = syn.make_scene(n_poses=poses, H=128, W=128, n_views=poses, pose_seed=5 + (0 if strong else rank))
    rng = np.random.default_rng(0 if strong else rank)
    js, is_ = np.meshgrid(np.arange(128), np.arange(128), indexing="ij")
    sel = np.nonzero(((np.abs(is_ - 64) < 128 * 0.22) & (np.abs(js - 64) < 128 * 0.40)).reshape(-1))[0]
    ro, rd, pose = [], [], []
    for p in range(poses):
        idx = np.sort(rng.choice(sel, size=rpp, replace=False))
        ro.append(scene["rays"][p][0][idx]); rd.append(scene["rays"][p][1][idx]); pose += [p] * rpp
    pose = np.array(pose)
    R_global = len(pose) * (1 if strong else world)
    ro, rd = np.concatenate(ro), np.concatenate(rd)
    target, bgs = rng.uniform(size=(len(pose), 3)), rng.uniform(size=(len(pose), 3))
    if strong:          # rank r keeps poses [r * 16 / world, (r + 1) * 16 / world): whole poses, so N_uniques stays an integer
        mine = slice(rank * (poses // world) * rpp, (rank + 1) * (poses // world) * rpp)
        ro, rd, target, bgs, pose = ro[mine], rd[mine], target[mine], bgs[mine], pose[mine]
        poses = poses // world
    R = len(pose)
    batch = dict(rays_o=T(ro, device), rays_d=T(rd, device), target_s=T(target, device),
                 bgs=T(bgs, device), kp3d=T(scene["kps"][pose], device), skts=T(scene["skts"][pose], device),
                 bones=T(scene["bones"][pose], device), cyls=T(scene["cyls"][pose], device), cam_idxs=T(pose % 20, device, torch.int64),
                 N_uniques=poses)
    if trainer.fused_engine() is None:
        raise SystemExit(f"the fused training step does not cover this configuration: {trainer.fused_reason}")
    torch.manual_seed(1 + rank)
    step = [0]

    def one():
        out = trainer.train_batch(batch, i=step[0], global_step=step[0], sync_stats=False)
        step[0] += 1
        return out
    # The workload must not depend on how long a box takes to settle (every step is a real Adam update, the bone volumes
    # the model learns change the row count, and the settle phase runs a box-dependent number of steps): one step builds the engine,
    # its state is snapshotted, and the snapshot is restored in front of the warm-up steps and of EVERY timed block -- each block
    # trains steps 1 .. K from the same weights, moments and random stream.  rows_per_block: what the last step of each block saw.
    one()
    torch.cuda.synchronize()
    eng = trainer.engine
    snap = eng.snapshot() if eng is not None else None
    rows_per_block = []

    def restore():
        if snap is not None:
            eng.restore(snap)

    def note_rows(_):
        rows_per_block.append(int(trainer.last_preds["counts"][4].item()))
    coll = []

    def start_coll():                      # HIP events around the two all-reduces of every step of the timed blocks
        if dist is not None or args.nccl_world_1:
            trainer.collective_events = coll
    elapsed, (loss, stats), tinfo = timed(one, args.steps, args.warmup, dist, device, args.debug_single_device, settle_block=50,
                                          before_block=restore, after_block=note_rows, on_timed_start=start_coll)
    trainer.collective_events = None
    collective = None
    if coll:
        torch.cuda.synchronize()
        e_ms = np.array([[a.elapsed_time(b), c.elapsed_time(d), a.elapsed_time(d)] for a, b, c, d in coll])
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as exc:  # noqa: BLE001
            rccl = f"unknown ({exc})"
        collective = dict(early_allreduce_ms=float(np.median(e_ms[:, 0])), late_allreduce_ms=float(np.median(e_ms[:, 1])),
                          first_start_to_last_end_ms=float(np.median(e_ms[:, 2])), steps_measured=len(coll), rccl_version=rccl,
                          bytes=[int(x.numel() * 4) for x in trainer.engine.grad_buckets()],
                          note="HIP events on the streams the collectives run on (the early one on the comm side stream under the "
                               "weight-gradient kernels: its duration includes waiting for nothing but the wire and the other ranks); "
                               "medians over the steps of the timed blocks, this rank")
    S = targs.N_samples + targs.N_importance
    counts = trainer.last_preds["counts"].cpu().tolist()
    rows, in_vol = counts[4], counts[5]
    # dense-layer work of a step as EXECUTED per row: forward (pts 0..7, alpha, the merged feature / view matrix W_fv 256 -> 128, rgb),
    # input gradients (the transposed matrices), weight gradients (one N x K product per matrix and row)
    mac_fwd = 195 * 256 + 4 * 256 * 256 + 451 * 256 + 2 * 256 * 256 + 256 + 256 * 128 + 128 * 3
    mac_dx = 3 * 128 + 128 * 256 + 256 + 6 * 256 * 256 + 256 * 451 + 256 * 195    # rgb^T, W_fv^T, alpha, trunk 7..1 (5: 451 out), layer 0
    mac_ref = 195 * 256 + 4 * 256 * 256 + 451 * 256 + 2 * 256 * 256 + 257 * 256 + 411 * 128 + 128 * 3   # the reference's forward per row
    ms = 1e3 * elapsed / args.steps
    if dist is not None:                 # whole-job executed rows: sum over the ranks
        rt = torch.tensor([float(rows), float(in_vol)], dtype=torch.float64, device="cpu" if args.debug_single_device else device)
        dist.all_reduce(rt)
        rows, in_vol = int(rt[0].item()), int(rt[1].item())
    flops = 2.0 * rows * (2 * mac_fwd + mac_dx)
    achieved = flops / (ms * 1e-3) / world           # per GPU: the roofline is one GPU's
    peak = PEAK_FP16_MFMA / 3.0
    # HBM bytes of a step from the PMC passes of tools/pmc_train.sh, quoted only for the kernel sources they were measured on
    traffic, hbm = None, None
    rec = pmc_record("train", "k_train.hip", "k_mlp16.hip", "k_mlp16_bwd.hip", "mlp16_core.hpp", "k_dw16.hip", "k_assign_bwd.hip",
                     "k_train_rows.hip", "k_train_head.hip", "common.hpp")
    if rec is not None:
        if True:
            traffic = rec["step_hbm_bytes"]
            hbm = dict(bytes_per_step=traffic, achieved=traffic / (ms * 1e-3) / 1e12, peak=8.0, unit="TB/s", frac=traffic / (ms * 1e-3) / 8e12,
                       note="steady-state steps only (tools/pmc_train.sh); every activation and gradient of the trunk is written once "
                            "(forward / input-gradient chain, fragment order: 0.59 + 0.52 GB) and read ONCE by the weight-gradient kernel "
                            "(k_dw16: 1.07 GB per step against 1.0 GB if no operand were read twice; it streams them at the ~3.8 TB/s a "
                            "reduction reaches on this part)")
    result = {
        "metric": "training ray-samples/sec (PerfCap danbo_fast step: forward + losses + backward + Adam)", "value": R_global * S / (ms * 1e-3),
        "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32 (fp16x2-split MFMA products, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "BASELINE config 4: PerfCap danbo_fast training step, " +
                               (f"ONE batch of 3072 rays = 16 poses x 192 split by whole poses: {poses} poses = {R} rays per rank, " if strong
                                else "3072 rays = 16 poses x 192 per rank, ") +
                               "32 + 16 samples, perturb = 1, raw_noise_std = 1, L1 + soft-softmax + volume-scale losses, Adam; "
                               "danbo_train_step (fused trunk: one forward kernel per pass, one input-gradient chain) + "
                               "danbo_adam_step, HIP graph; flat-gradient all-reduce (RCCL) for N > 1",
                   "rays": R_global, "rays_per_rank": R, "samples_per_ray": S, "parallelism": f"rays-dp{world}"},
        "rows_per_step": rows, "in_volume_fraction": in_vol / (R_global * S), "loss": float(loss["total_loss"]),
        "rows_per_block": rows_per_block,
        "rows_note": "every timed block (and the warm-up) starts from the SAME snapshot of weights, Adam moments and random stream, taken "
                     "after the engine's first step: each block trains steps 2 .. K + 1 of the same run, whatever the settle phase did "
                     "(round 4 timed a model that had trained through a box-dependent number of settle steps: 50 k .. 64 k rows)",
        "roofline": dict(bound="mfma", kernel="k_train_mlp_fwd x 2 + k_train_mlp_bwd + k_dw16 (whole step)", achieved=achieved / 1e12,
                         peak=peak / 1e12, unit="TFLOP/s", frac=achieved / peak, traffic=traffic, flop_per_row=2 * (2 * mac_fwd + mac_dx),
                         flop_per_row_reference=2 * (3 * mac_ref),
                         peak_note=SPLIT_NOTE,
                         note="STEP-level lower bound, per GPU: executed dense-layer flops of the step (forward + input gradients + weight "
                              "gradients on the compacted rows) divided by the WHOLE step time incl. every non-GEMM kernel and Adam; "
                              "per-kernel durations: profiles/r04_train_kernel_stats.csv"),
        **tinfo,
        "collective_ms": collective,
        "collectives": ("none (one rank, no process group)" if dist is None and not args.nccl_world_1 else
                        f"two in-place all-reduces of the flat gradient per step ({'gloo' if args.debug_single_device else 'nccl = RCCL'}, "
                        f"world {world}{', forced at world 1: --nccl-world-1' if args.nccl_world_1 else ''}), the first on a side stream "
                        "under the weight-gradient kernels; split step = two HIP graphs"),
        "strong_scaling_bound": "the reference's ONE 3072-ray batch is small for eight MI355X: a 384-ray shard (8 ranks x 2 poses x 192) "
                                "still takes 0.90 ms of the 1.53 ms the whole batch takes on one GPU (tile rounds of 128 rows x 256 "
                                "CUs are the granule), so --scaling strong is bounded at ~1.7x on 8 GPUs before the collective; "
                                "--scaling weak (3072 rays per rank, the default) is the scaling configuration",
    }
    if hbm is not None:
        result["hbm"] = hbm
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_train(targs, caster, batch, poses)
        restore()
        result["parity"] = train_parity_block(targs, caster, trainer, batch, poses, device)
        result["parity_ok"] = result["parity"]["parity_ok"]
        restore()
    return result


def train_parity_block(targs, caster, trainer, batch, poses, device, n_poses=2):
    """Config 4 against the float64 arbiter (the line had no parity field): ONE fused step (danbo_train_step, eager,
    the snapshot's weights) on a 2-pose / 384-ray shard of the bench batch with the step's random draws supplied (fixed_draws: the
    arbiter needs the same stratified offsets, inverse-CDF uniforms and density noise), and oracle/torch_f64_train.py on the same
    shard at the depths / merge order / min(sum w, 1) branches the step took: the four loss terms and the gradient norms of one tensor
    per sub-network, plus the worst entry-wise gradient error over ALL tensors relative to the tensor's largest entry (one float64
    evaluation, ReLU kinks not bracketed: a unit within fp32 round-off of zero moves a tensor by ~1e-3 of its max at this batch size).
    Checker code: the float64 graph runs as torch kernels on the GPU beside the path under test; nothing here is timed."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_f64_train as t64
    from core import _hip
    from core.utils import synthetic as syn
    eng = trainer.engine
    R = n_poses * 192
    S, Sf = int(targs.N_samples), int(targs.N_importance)
    b = {k: (v[:R] if torch.is_tensor(v) else v) for k, v in batch.items()}
    gen = torch.Generator(device="cpu").manual_seed(4)
    B = float(targs.density_scale)
    dr = lambda *shape: torch.rand(*shape, generator=gen).to(device)  # noqa: E731
    dn = lambda *shape: (torch.randn(*shape, generator=gen) * float(targs.raw_noise_std) * B).to(device)  # noqa: E731
    draws = dict(t_rand=dr(R, S), u_rand=dr(R, Sf), noise_c=dn(R, S), noise_f=dn(R, S + Sf))
    pp = lambda x: x[::192]  # noqa: E731
    eng.fixed_draws = draws
    try:
        out = eng.forward_backward(b["rays_o"], b["rays_d"], pp(b["skts"]), pp(b["bones"]), pp(b["cyls"]), b["cam_idxs"], b["target_s"], b["bgs"],
                                   S, Sf, perturb=float(targs.perturb), raw_noise_std=float(targs.raw_noise_std))
        torch.cuda.synchronize()
    finally:
        eng.fixed_draws = None
    grads = {n: p.grad.detach().double().cpu().numpy() for n, p in caster.network.named_parameters() if p.requires_grad}
    v = _hip.DanboTrainView()
    _hip.check(_hip.lib().danbo_train_workspace_view(ctypes.byref(eng._model()), R, n_poses, S, Sf, R, ctypes.c_void_p(eng._ws.data_ptr()),
                                                     ctypes.byref(v)), "danbo_train_workspace_view")
    base = eng._ws.data_ptr()
    at = lambda ptr, shape, dt: eng._ws[ptr - base:ptr - base + 4 * int(np.prod(shape))].view(dt).view(shape).cpu().numpy()  # noqa: E731
    z_c, z_f, order = at(v.z_coarse, (R, S), torch.float32), at(v.z_fine, (R, Sf), torch.float32), at(v.order, (R, S + Sf), torch.int32)
    sd = {k: t.detach().cpu().numpy() for k, t in caster.network.state_dict().items()}
    coef = dict(loss_fn=targs.loss_fn, use_background=bool(targs.use_background), rgb_loss_coef=float(targs.rgb_loss_coef),
                coarse_weight=float(targs.coarse_weight), soft_softmax_loss_coef=float(targs.soft_softmax_loss_coef),
                vol_scale_penalty=float(targs.vol_scale_penalty) if targs.opt_vol_scale else 0.0)
    nb = {k: b[k].detach().cpu().numpy() for k in ("rays_o", "rays_d", "skts", "bones", "target_s", "bgs", "cam_idxs")}
    ref = t64.step(syn.model_config("danbo_perfcap"), coef, sd, caster.transforms[0].cpu().numpy(), caster.network.graph_net.init_scale.cpu().numpy(),
                   nb, z_c, z_f, order, n_poses, noise_c=draws["noise_c"].cpu().numpy(), noise_f=draws["noise_f"].cpu().numpy(), device=str(device),
                   clamped_c=out["acc0"].cpu().numpy() >= 1.0, clamped_f=out["acc_map"].cpu().numpy() >= 1.0)
    ls = out["loss"].double().cpu().numpy()
    ours = {"rgb_loss": ls[0], "rgb_loss0": ls[1], "soft_softmax_loss": ls[2] * coef["soft_softmax_loss_coef"] / (R * (S + Sf)), "vol_scale_loss": ls[3]}
    loss = {k: dict(hip=float(ours[k]), float64=float(ref["loss"][k]), rel=float(abs(ours[k] - ref["loss"][k]) / max(abs(ref["loss"][k]), 1e-12)))
            for k in ours if k in ref["loss"]}
    norms = {}
    for n in ("pts_linears.4.weight", "graph_net.layers.3.weight", "prob_linears.layers.1.weight"):
        a, r = float(np.sqrt((grads[n] ** 2).sum())), float(np.sqrt((ref["grads"][n] ** 2).sum()))
        norms[n] = dict(hip=a, float64=r, rel=abs(a - r) / max(r, 1e-30))
    worst, worst_name = 0.0, None
    for n, r in ref["grads"].items():
        if n in grads and np.abs(r).max() > 0:
            e = float(np.abs(grads[n] - r).max() / np.abs(r).max())
            if e > worst:
                worst, worst_name = e, n
    LOSS_BOUND, NORM_BOUND, TENSOR_BOUND = 2e-4, 2e-3, 2e-2
    ok = all(x["rel"] <= LOSS_BOUND for x in loss.values()) and all(x["rel"] <= NORM_BOUND for x in norms.values()) and worst <= TENSOR_BOUND
    return dict(against="oracle/torch_f64_train.py (float64 autograd, the reference's graph) on a 2-pose / 384-ray shard of the bench batch, at the "
                        "fused step's own depths, merge order and random draws", rays=R, loss_terms=loss, gradient_norms=norms,
                worst_gradient_entry_rel_to_tensor_max=worst, worst_gradient_tensor=worst_name,
                max_abs_rgb=float(np.abs(out["rgb_map"].cpu().numpy() - ref["rgb_map"]).max()),
                bounds=dict(loss_rel=LOSS_BOUND, gradient_norm_rel=NORM_BOUND, gradient_entry_rel_to_max=TENSOR_BOUND), parity_ok=bool(ok))


def cpu_baseline_train(targs, caster, batch, poses, n_poses=2):
    """the same step (forward, L1 losses of both passes, backward, Adam) as a torch-CPU restatement with autograd, on a shard of
    n_poses x 192 rays, dense like the reference"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_cpu_train
    R = n_poses * 192
    b = {k: (v[:R].cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
    sd = {k: v.detach().cpu() for k, v in caster.network.state_dict().items()}
    from core.utils import synthetic as syn
    step = torch_cpu_train.make_step(targs, syn.model_config("danbo_perfcap"), sd, caster.transforms[0].cpu(), b, n_poses)
    step()
    dt, _ = best_of(step)
    S = targs.N_samples + targs.N_importance
    return dict(value=R * S / dt, unit="ray-samples/s", cores=int(torch.get_num_threads()), host_cpu_count=os.cpu_count(), kind="port",
                sample=f"{R} rays = {n_poses} poses x 192 of the same batch, forward + L1 losses + backward + Adam with torch autograd on "
                       f"CPU kernels ({torch.get_num_threads()} threads), every sample through every bone and the full MLP, best of 3: "
                       f"{dt:.2f} s per step")


# ---------------------------------------------------------------------------------------------- config 5: A-NeRF frame
def bench_anerf(args, rank, world, device, dist):
    from core.anerf_engine import AnerfEngine
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import bone_align_transforms
    cfg = syn.model_config("anerf_base")
    rest = syn.rest_pose(cfg["rest_scale"])
    sd = syn.make_state_dict(cfg, seed=0, n_framecodes=100, rest=rest)
    scene = syn.make_scene(n_poses=1, H=H, W=W, n_views=8, pose_seed=0, min_radius=1.25)
    ro, rd = scene["rays"][rank % 8]
    eng = AnerfEngine(cfg, {k: T(v, device) for k, v in sd.items()}, T(bone_align_transforms(rest), device))
    inp = dict(rays_o=T(ro, device), rays_d=T(rd, device), skts=T(scene["skts"], device), bones=T(scene["bones"], device),
               cyls=T(scene["cyls"], device), cam_idx=torch.zeros(len(ro), dtype=torch.int64, device=device))
    S, Sf = 48, 16
    run = lambda: eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"], S, Sf)  # noqa: E731
    elapsed, out, tinfo = timed(run, args.steps, args.warmup, dist, device, args.debug_single_device, settle_block=2)
    n = len(ro) * (S + Sf)
    Wd, inc, VW = cfg["W"], 432, cfg["view_W"]
    mac = inc * Wd + 4 * Wd * Wd + (inc + Wd) * Wd + 2 * Wd * Wd + Wd + Wd * VW + 24 * VW + 3 * VW
    ms = 1e3 * elapsed / args.steps
    achieved = 2.0 * n * mac / (ms * 1e-3)
    peak = PEAK_FP16_MFMA / 3.0
    # HBM bytes of a frame from the PMC passes of tools/pmc_anerf.sh, quoted only for the kernel sources they were measured on
    traffic, hbm = None, None
    rec = pmc_record("anerf", "k_linear16.hip", "k_anerf.hip", "common.hpp")
    if rec is not None:
        if True:
            traffic = rec["frame_hbm_bytes"]
            # algorithmic: every dense layer reads its input row(s) and writes its output row once (4 (K + N) bytes per row), the
            # encoder writes the 432 inputs + 24 cutoff weights, the colour kernel reads the 225-wide head rows + weights
            per_row = 4 * ((inc + Wd) + 4 * 2 * Wd + (inc + 2 * Wd) + 2 * 2 * Wd + (Wd + VW + 1)) + 4 * (inc + 24) + 4 * (VW + 1 + 24) + 16
            hbm = dict(bytes_per_frame=traffic, algorithmic_bytes_per_frame=per_row * n, achieved=traffic / (ms * 1e-3) / 1e12, peak=8.0,
                       unit="TB/s", frac=traffic / (ms * 1e-3) / 8e12,
                       note="the frame moves its activations through HBM once per layer (fragment order, no re-reads): it runs at the "
                            "HBM rate this GPU sustains for mixed read/write streams (3.4 TB/s, the same as the training step's "
                            "weight-gradient kernel) AND at the MFMA fraction of the register-resident DANBO kernel")
    result = {
        "metric": "ray-samples/sec at 512x512x64 samples (A-NeRF)", "value": world * n / (ms * 1e-3), "unit": "ray-samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (fp16x2-split MFMA products, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "BASELINE config 5: A-NeRF anerf_base (cutoff PE, W = 448), 512x512 rays x (48 + 16) samples, tau = 20, "
                               "every sample evaluated (A-NeRF has no in-volume mask)",
                   "rays": len(ro), "samples_per_ray": S + Sf, "parallelism": f"rays-dp{world}"},
        "roofline": dict(bound="mfma", kernel="A-NeRF trunk (whole frame)", achieved=achieved / 1e12, peak=peak / 1e12, unit="TFLOP/s",
                         frac=achieved / peak, traffic=traffic, flop_per_row=2 * mac, flop_per_row_reference=2 * 2268000, peak_note=SPLIT_NOTE,
                         note="FRAME-level lower bound: executed flops per sample x samples / whole frame time; traffic = HBM bytes of "
                              "the whole frame (tools/pmc_anerf.sh), see `hbm`"),
        **tinfo,
    }
    if hbm is not None:
        result["hbm"] = hbm
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # oracle/torch_cpu.AnerfTorchCPU: the same multi-threaded torch-CPU restatement the DANBO configs are timed against (the
        # numpy oracle stays the parity checker: it is not a fair clock), thread count chosen on a probe, best of 3
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import danbo_oracle as o
        import torch_cpu
        model = torch_cpu.AnerfTorchCPU(cfg, sd, rest)

        def run(n_rays):
            r0 = (H // 2) * W - n_rays // 2
            rb = syn.ray_batch(ro[r0:r0 + n_rays], rd[r0:r0 + n_rays])
            z = np.zeros(n_rays, dtype=np.int64)
            return r0, model.render(rb, scene["skts"][z], scene["bones"][z], scene["cyls"][z], np.zeros(n_rays, np.int64), 1, S, Sf)
        run(256)
        best_t, best_dt = torch.get_num_threads(), None
        for nt in sorted({8, 16, 32, 64, torch.get_num_threads()}):
            if nt > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            run(512)
            dt = time.perf_counter() - t0
            if best_dt is None or dt < best_dt:
                best_t, best_dt = nt, dt
        torch.set_num_threads(best_t)
        n_rays = 4096
        dt, (r0, ref) = best_of(lambda: run(n_rays))
        rgb = out["rgb_map"][r0:r0 + n_rays].cpu().numpy()
        parity5 = anerf_parity_block(cfg, sd, rest, scene, ro, rd, r0, n_rays, S, Sf, eng, inp, out, model)
        result["cpu_baseline"] = dict(value=n_rays * (S + Sf) / dt, unit="ray-samples/s", cores=int(torch.get_num_threads()),
                                      host_cpu_count=os.cpu_count(), kind="port",
                                      sample=f"{n_rays} centre rays x {S}+{Sf} samples of the same frame through oracle/torch_cpu.AnerfTorchCPU "
                                             f"(torch CPU kernels, {torch.get_num_threads()} threads), 4096-ray chunks, best of 3: {dt:.2f} s")
        result["parity"] = dict(against="oracle/torch_cpu.AnerfTorchCPU on the cpu_baseline sample", rays=n_rays,
                                psnr_rgb_db=float(o.psnr(rgb, ref["rgb_map"])), max_abs_rgb=float(np.abs(rgb - ref["rgb_map"]).max()), **parity5)
        result["parity_ok"] = parity5["parity_ok"]
    return result


def anerf_parity_block(cfg, sd, rest, scene, ro, rd, r0, n_rays, S, Sf, eng, inp, frame, model):
    """Config 5 at the RAW level (its line carried maps only): the HIP path's coarse-pass logits on the cpu_baseline
    sample against oracle/torch_cpu.AnerfTorchCPU in float32 and in float64 (the same graph on the float32 points), at the HIP path's
    own depths and with the oracle's near / far fed in -- the dictionary configs 1 - 3 carry (parity_block), minus the in-volume mask
    A-NeRF does not have.  Checker code: nothing here is timed."""
    import torch_cpu
    from core.utils import synthetic as syn
    sl = slice(r0, r0 + n_rays)
    z0 = np.zeros(n_rays, dtype=np.int64)
    rb = syn.ray_batch(ro[sl], rd[sl])
    ref = model.render(rb, scene["skts"][z0], scene["bones"][z0], scene["cyls"][z0], np.zeros(n_rays, np.int64), 1, S, Sf, stages=True)
    m64 = torch_cpu.AnerfTorchCPU(cfg, sd, rest, dtype=torch.float64)
    t = lambda v: torch.tensor(np.ascontiguousarray(v, dtype=np.float32))  # noqa: E731
    with torch.no_grad():
        r64 = np.concatenate([m64.forward(t(ref["pts_coarse"][a:a + 1024]), t(rb[a:a + 1024, 3:6]), t(scene["skts"][z0[a:a + 1024]]),
                                          np.zeros(min(1024, n_rays - a), np.int64)).numpy() for a in range(0, n_rays, 1024)])
    rr = ref["raw_coarse"]
    dev = inp["rays_o"].device
    args5 = (inp["rays_o"][sl], inp["rays_d"][sl], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"][sl], S, Sf)
    own = eng.render(*args5, keep=True)
    near_o, far_o = ref["near"][:, 0], ref["far"][:, 0]
    near_g, far_g = own["near"].cpu().numpy().reshape(-1), own["far"].cpu().numpy().reshape(-1)
    same = (near_g == near_o) & (far_g == far_o)
    at_oracle = eng.render(*args5, near_far=(torch.tensor(near_o, device=dev), torch.tensor(far_o, device=dev)), keep=True)
    p = dict(max_abs_near=float(np.abs(near_g - near_o).max()), max_abs_far=float(np.abs(far_g - far_o).max()),
             bounds_bit_equal_rays=int(same.sum()),
             max_abs_acc=float(np.abs(frame["acc_map"][sl].cpu().numpy() - ref["acc_map"]).max()),
             sample_render_equals_frame=bool(torch.equal(own["rgb_map"], frame["rgb_map"][sl])))
    for tag, o_ in (("own_depths", own), ("oracle_depths", at_oracle)):
        raw = o_["raw_coarse"].cpu().numpy()
        f32_fl, f32_un = raw_measures(raw, rr)
        f64_fl, f64_un = raw_measures(raw, r64)
        p[tag] = dict(max_rel_raw_floored_5pct=f32_fl, max_rel_raw=f32_un, max_rel_raw_floored_5pct_vs_float64=f64_fl, max_rel_raw_vs_float64=f64_un)
    d = p["oracle_depths"]
    ok = d["max_rel_raw_floored_5pct"] <= PARITY_BOUND and d["max_rel_raw_floored_5pct_vs_float64"] <= PARITY_BOUND \
        and p["max_abs_near"] <= BOUNDS_BOUND and p["max_abs_far"] <= BOUNDS_BOUND
    timed_ok = float(np.abs(frame["rgb_map"][sl].cpu().numpy() - ref["rgb_map"]).max()) <= MAPS_BOUND and p["max_abs_acc"] <= MAPS_BOUND
    if same.any():
        t32, _ = raw_measures(own["raw_coarse"].cpu().numpy()[same], rr[same])
        t64_, _ = raw_measures(own["raw_coarse"].cpu().numpy()[same], r64[same])
        p["timed_frame_bit_equal_bound_rays"] = dict(rays=int(same.sum()), max_rel_raw_floored_5pct=t32, max_rel_raw_floored_5pct_vs_float64=t64_)
        timed_ok = timed_ok and t32 <= PARITY_BOUND and t64_ <= PARITY_BOUND
    p.update(restatement_fp32_vs_float64_floored_5pct=raw_measures(rr, r64)[0], max_rel_raw=d["max_rel_raw"],
             max_rel_raw_floored_5pct=d["max_rel_raw_floored_5pct"], max_rel_raw_vs_float64=d["max_rel_raw_vs_float64"],
             max_rel_raw_floored_5pct_vs_float64=d["max_rel_raw_floored_5pct_vs_float64"], decided_by="oracle_depths", bound=PARITY_BOUND,
             bounds_bound=BOUNDS_BOUND, maps_bound=MAPS_BOUND, parity_ok_timed_frame=bool(timed_ok), parity_ok=bool(ok and timed_ok),
             raw_note="coarse-pass logits of every sample (A-NeRF has no in-volume mask); measures as configs 1 - 3 (bench.parity_block)")
    return p


def launch_ranks(args, argv):
    """`python bench.py --gpus N` from a shell that is not already a rank: start N ranks of this script, one per GPU, as a child
    `python -m torch.distributed.run` -- from a process that has not touched the GPU (never re-exec one that has) -- and relay
    rank 0's JSON line.  Exit code = the child's."""
    import socket
    import subprocess
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or not lines:
        raise SystemExit(proc.returncode or 1)
    print(lines[-1])


def bench_dry_run(args, rank, world, device, dist):
    """--dry-run: the launch / rendezvous / timing harness -- timed() itself: settle phase, barriers, max over the ranks, per-rank
    times -- with a host sleep in place of the step (no HIP call at all).  --dry-run-ms a,b,..: the sleep of rank 0, 1, .. in
    milliseconds (unequal rank speeds: tests/test_multiprocess.py checks value = N x K x units / MAX-over-ranks time with them).
    Its line says so and is not a measurement: `units` per step and rank are a made-up 1000."""
    ms = [float(x) for x in args.dry_run_ms.split(",")]
    mine = ms[rank % len(ms)] * 1e-3

    def one():
        time.sleep(mine)
    elapsed, _, tinfo = timed(one, args.steps, args.warmup, dist, "cpu", True, settle_block=5)
    units = 1000
    return {"metric": "DRY RUN of the launch path (no GPU work, not a measurement)", "value": world * args.steps * units / elapsed,
            "unit": "made-up units/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling or "weak", "vs_baseline": None, "dtype": "none", "data": "none", **tinfo,
            "config": {"workload": f"dry run of --config {args.config}", "parallelism": f"rays-dp{world}", "units_per_step_and_rank": units}}


def main():
    global N_SAMPLES, N_IMPORTANCE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4, 5],
                    help="1: BASELINE metric (default); 2: danbo_fast 32+16 box bounds; 3: danbo_base 96+32; 4: training step; 5: A-NeRF")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="config 4 only: weak = 3072 rays per rank (default); strong = the reference's one 16-pose / 3072-ray batch "
                         "split by whole poses over the ranks (SURVEY 8e)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the timing of the same frame through core.trainer.render (api_value)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the occupancy sweep (camera distance) of the render configs")
    ap.add_argument("--coarse", type=int, default=None, help="dev: coarse samples per ray")
    ap.add_argument("--fine", type=int, default=None, help="dev: importance samples per ray")
    ap.add_argument("--box-near-far", action="store_true", help="dev: per-bone box near/far (config 2 sets it)")
    ap.add_argument("--debug-single-device", action="store_true",
                    help="dev: every rank uses cuda:0 and the gloo backend (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--nccl-world-1", action="store_true",
                    help="config 4 with --gpus 1: create a ONE-rank nccl (= RCCL) process group and run the data-parallel form of the "
                         "step (split phases, both in-place all-reduces, comm side stream) -- everything of the N > 1 path but the wire")
    ap.add_argument("--dry-run-ms", default="1", help="--dry-run: milliseconds the step of rank 0, 1, .. sleeps (comma separated)")
    ap.add_argument("--dry-run", action="store_true",
                    help="dev: launch / rendezvous / timing harness only, a host no-op as the step (gloo, no GPU needed); not a measurement")
    ap.add_argument("--mlp", choices=["f16split", "fp32"], default="f16split",
                    help="f16split: fp32-accurate products as 3 fp16 MFMAs (default); fp32: exact fp32 MFMA kernels")
    args = ap.parse_args()
    if args.scaling == "strong" and args.config != 4:
        raise SystemExit("--scaling strong is defined for --config 4 (the training batch); the render configs shard views: weak")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- not a rank yet?  Then become the launcher: nothing above or in this branch touches the GPU.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])

    if args.config in CONFIGS:
        N_SAMPLES, N_IMPORTANCE, box = CONFIGS[args.config]
        args.box_near_far = args.box_near_far or box
    N_SAMPLES = args.coarse or N_SAMPLES
    N_IMPORTANCE = args.fine or N_IMPORTANCE

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    device = None
    if not args.dry_run:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: libdanbo_hip has no CPU path")
        if args.debug_single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.debug_single_device or args.dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)   # "nccl" is RCCL on ROCm
    elif args.nccl_world_1:
        if args.config != 4:
            raise SystemExit("--nccl-world-1 is defined for --config 4 (the only path with a collective)")
        import socket
        import torch.distributed as dist1
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist1.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
    ranks_seen = dist.get_world_size() if dist is not None else 1

    fn = bench_dry_run if args.dry_run else {4: bench_train, 5: bench_anerf}.get(args.config, bench_render)
    result = fn(args, rank, world, device, dist)
    result["ranks_seen"] = ranks_seen
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    elif args.nccl_world_1:
        dist1.destroy_process_group()
    if rank == 0 and result.get("parity_ok") is False:
        # a fast frame whose logits are not the reference's is not a result: the line is printed (it says what missed), the exit
        # code says the run failed
        print("bench.py: PARITY MISS -- see `parity` in the line above", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
