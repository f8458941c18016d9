#!/usr/bin/env python3
"""Headline benchmark: ray-samples/s of the DANBO render path at 512x512 rays x 64 samples.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1
the driver launches one rank per GPU with torch.distributed.run.  A *step* is one full frame:
262 144 rays x (48 coarse + 16 importance) network evaluations = 16.78 M ray-samples through
near/far -> sampling -> transform+cull -> gather/assign/blend -> PE+MLP -> composite ->
importance resampling -> fine pass -> composite, with rays / pose / weights already in HBM.
Ranks render different camera views of the same pose (rays shard data-parallel, no collective
on the data path) => weak scaling; value = N * K * samples_per_frame / max-over-ranks time.

Workload (BASELINE.json configs[2] network at the metric's 64 samples):
D-H36M `danbo_base` network (FGNNcat + vox_MIXGNN, W=256, D=8, 128-d frame codes), seeded
synthetic weights / SMPL pose / bullet-time camera.  `value` is measured with exact in-volume
culling (identical raw to evaluating every sample -- tests/test_gpu_kernels.py checks bitwise);
`dense_value` is the same frame with every sample pushed through every kernel (the reference's
executed work).  `roofline` is for the dominant kernel (k_pe_mlp, fp32 MFMA) with EXECUTED
flops only.
"""
import argparse
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
               "f16split": 558592 + 256 + 32768 + 384}          # k_pe_mlp16: feature+view merged into one 256->128
PEAK_FP32_MFMA = 157.3e12      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_FP16_MFMA = 2500e12       # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (2:1-sparse figures are not used)


def build_workload(device, view, mlp_mode="f16split"):
    from core.render_engine import DanboEngine
    from core.utils import synthetic as syn
    cfg = syn.model_config("danbo_base")
    rest = syn.rest_pose(cfg["rest_scale"])
    sd = syn.make_state_dict(cfg, seed=0, n_framecodes=100, rest=rest)
    scene = syn.make_scene(n_poses=1, H=H, W=W, n_views=8, pose_seed=0, min_radius=1.25)
    ro, rd = scene["rays"][view % 8]
    T = lambda x, dt=torch.float32: torch.tensor(np.ascontiguousarray(x), dtype=dt, device=device)  # noqa: E731
    from core.utils.skeleton_utils import bone_align_transforms
    align = bone_align_transforms(rest)
    eng = DanboEngine(cfg, {k: T(v) for k, v in sd.items()}, T(align), mlp_mode=mlp_mode)
    inputs = dict(rays_o=T(ro), rays_d=T(rd), skts=T(scene["skts"]), bones=T(scene["bones"]), cyls=T(scene["cyls"]),
                  cam_idx=torch.zeros(len(ro), dtype=torch.int64, device=device))
    return eng, inputs, (cfg, sd, rest, scene, ro, rd)


def render(eng, inp, dense=False):
    return eng.render(inp["rays_o"], inp["rays_d"], inp["skts"], inp["bones"], inp["cyls"], inp["cam_idx"],
                      N_SAMPLES, N_IMPORTANCE, chunk=4096, dense=dense)


def cpu_baseline(extra, n_rays=4096, frame=None):
    """The numpy oracle (a CPU port of the reference path) on one 4096-ray chunk of the frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import danbo_oracle as o
    cfg, sd, rest, scene, ro, rd = extra
    from core.utils import synthetic as syn
    r0 = (H // 2) * W - n_rays // 2          # rows around the image centre (body pixels)
    sl = slice(r0, r0 + n_rays)
    rb = syn.ray_batch(ro[sl], rd[sl])
    z = np.zeros(n_rays, dtype=np.int64)
    orc = o.DanboOracle(cfg, sd, rest)
    t0 = time.perf_counter()
    ref = orc.render(rb, scene["skts"][z], scene["bones"][z], scene["cyls"][z], np.zeros(n_rays, np.int64), 1,
                     N_SAMPLES, N_IMPORTANCE)
    dt = time.perf_counter() - t0
    parity = None
    if frame is not None:   # the timed HIP path's image on the same rays (checker only: nothing here is timed or shipped)
        rgb, acc = frame["rgb_map"][sl].cpu().numpy(), frame["acc_map"][sl].cpu().numpy()
        parity = dict(against="oracle/danbo_oracle.py on the cpu_baseline sample", rays=n_rays,
                      psnr_rgb_db=float(o.psnr(rgb, ref["rgb_map"])), max_abs_rgb=float(np.abs(rgb - ref["rgb_map"]).max()),
                      max_abs_acc=float(np.abs(acc - ref["acc_map"]).max()))
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return dict(value=n_rays * (N_SAMPLES + N_IMPORTANCE) / dt, unit="ray-samples/s", cores=int(cores), kind="port",
                sample=f"{n_rays} centre rays x {N_SAMPLES}+{N_IMPORTANCE} samples of the same frame through "
                       f"oracle/danbo_oracle.py (numpy, BLAS threads), {dt:.1f} s"), parity


def main():
    global N_SAMPLES, N_IMPORTANCE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--coarse", type=int, default=N_SAMPLES, help="dev: coarse samples per ray (metric: 48)")
    ap.add_argument("--fine", type=int, default=N_IMPORTANCE, help="dev: importance samples per ray (metric: 16)")
    ap.add_argument("--box-near-far", action="store_true",
                    help="dev: per-bone box near/far as in the danbo_fast configs (SURVEY 8d config 2: with --coarse 32 --fine 16)")
    ap.add_argument("--debug-single-device", action="store_true",
                    help="dev: every rank uses cuda:0 and the gloo backend (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--mlp", choices=["f16split", "fp32"], default="f16split",
                    help="f16split: fp32-accurate products as 3 fp16 MFMAs (default); fp32: exact fp32 MFMA kernels")
    args = ap.parse_args()
    N_SAMPLES, N_IMPORTANCE = args.coarse, args.fine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libdanbo_hip has no CPU path")
    if args.debug_single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.debug_single_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)   # "nccl" is RCCL on ROCm

    eng, inp, extra = build_workload(device, view=rank, mlp_mode=args.mlp)
    eng.cfg["use_volume_near_far"] = bool(args.box_near_far)
    for _ in range(args.warmup):
        render(eng, inp)
    torch.cuda.synchronize()

    eng.profile = {}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = render(eng, inp)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if args.debug_single_device else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_frame = H * W * (N_SAMPLES + N_IMPORTANCE)
    value = world * args.steps * samples_per_frame / elapsed

    # ---- roofline of the dominant kernel from the HIP events recorded inside the timed region
    prof = eng.profile["k_pe_mlp"]
    eng.profile = None
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in prof)
    rows = sum(int(c.item()) if torch.is_tensor(c) else int(c) for _, _, c in prof)
    mac = MAC_PER_ROW[args.mlp]
    flops = 2.0 * mac * rows                          # EXECUTED multiply-adds of in-volume rows only
    achieved = flops / (ms * 1e-3)
    if args.mlp == "f16split":
        kernel, peak = "k_pe_mlp16", PEAK_FP16_MFMA / 3.0
        peak_note = ("dense fp16 MFMA peak 2500 TFLOP/s / 3: every fp32-accurate product is three half-precision "
                     "MFMA products (hi*hi + hi*lo + lo*hi); executed MFMA rate = 3 x achieved")
    else:
        kernel, peak = "k_pe_mlp", PEAK_FP32_MFMA
        peak_note = "fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)"
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01r_pmc_hbm.json")
    if os.path.exists(pmc) and args.mlp == "f16split":
        traffic = json.load(open(pmc))["kernels"].get("danbo::k_pe_mlp16", {}).get("hbm_bytes_per_launch")
    roofline = dict(bound="mfma", kernel=kernel, achieved=achieved / 1e12, peak=peak / 1e12, unit="TFLOP/s",
                    frac=achieved / peak, traffic=traffic, launches=len(prof), avg_launch_ms=ms / len(prof),
                    rows_per_launch=rows / len(prof), flop_per_row=2 * mac, flop_per_row_reference=2 * MAC_PER_ROW_REF,
                    peak_note=peak_note,
                    note="executed flops of rows inside >=1 bone volume only; traffic = HBM bytes per launch from "
                         "rocprofv3 PMC passes (profiles/r01r_pmc_hbm.json, tools/pmc_hbm.sh), algorithmic bytes = 84 B per row")

    result = {
        "metric": f"ray-samples/sec at 512x512x{N_SAMPLES + N_IMPORTANCE} samples", "value": value, "unit": "ray-samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.mlp == "fp32" else "f32 (fp16x2-split MFMA products, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"H36M danbo_base network, 512x512 rays x ({N_SAMPLES} coarse + {N_IMPORTANCE} importance) samples, "
                               f"1 pose / 1 camera per rank, {'per-bone box' if args.box_near_far else 'cylinder'} near/far, "
                               "exact in-volume culling",
                   "rays": H * W, "samples_per_ray": N_SAMPLES + N_IMPORTANCE, "parallelism": f"rays-dp{world}"},
        "in_volume_fraction": rows / (args.steps * samples_per_frame),
        "roofline": roofline,
    }
    if rank == 0:
        if not args.no_dense and world == 1:
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
            result["dense_equals_culled"] = bool(torch.equal(out_d["rgb_map"], out["rgb_map"]))
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"], parity = cpu_baseline(extra, frame=out)
            if parity is not None:
                result["parity"] = parity
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
