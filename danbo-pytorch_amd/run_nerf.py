"""Training entry point (reference: run_nerf.py -- `render_path` :29-147, `render_testset` :150-184, `train` :575-737).

    python danbo-pytorch_amd/run_nerf.py --config danbo-pytorch_amd/configs/surreal/danbo_fast.txt --basedir logs --expname demo
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 danbo-pytorch_amd/run_nerf.py --config ...

Same flags, log directory (`args.txt`, `config.txt`, `NNNNNN.tar` checkpoints in the reference's layout) and loop as the
reference; the renderer underneath is the HIP path.  One process drives one GPU: under `torch.distributed.run` every rank takes
whole images of each batch (core/load_data.py) and the flat gradient is all-reduced once per step (core/trainer.py) -- the
reference's nn.DataParallel replicas are gone.  Data: `--dataset_type synthetic` (default; seeded poses, teacher-rendered
images) or `npz`; TensorBoard / video writers are not in this image, scalars go to stdout and `scalars.jsonl`.
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from core.config import config_parser, parse_args  # noqa: E402,F401
from core.load_data import load_data  # noqa: E402
from core.raycasters import create_raycaster  # noqa: E402
from core.trainer import Trainer, render  # noqa: E402
from core.utils.evaluation_helpers import evaluate_metric  # noqa: E402
from core.utils.ray_utils import kp_to_valid_rays  # noqa: E402


def render_path(render_poses, hwf, chunk, render_kwargs, centers=None, kp=None, skts=None, cyls=None, bones=None, gt_imgs=None,
                bg_imgs=None, bg_indices=None, cams=None, subject_idxs=None, render_factor=0, white_bkgd=False, ret_acc=False,
                ext_scale=0.00035, base_bg=1.0):
    """One image per camera in `render_poses`: only the pixels inside the 2-D box of the pose's bounding cylinder are cast,
    the rest of the image is the background (`bg_imgs[bg_indices[i]]` resized, white, or black).  Pose tensors with fewer
    entries than cameras are cycled (`i % n`).  -> rgbs [N,H,W,3], disps [N,H,W,1], accs, valid_idxs, bboxes"""
    H, W, focal = hwf
    if render_factor != 0:
        H, W = H // render_factor, W // render_factor
        focal = focal / render_factor
        centers = centers / render_factor if centers is not None else None
    rays, valid_idxs, cyls, bboxes = kp_to_valid_rays(render_poses, H, W, focal, kps=kp, cylinder_params=cyls, skts=skts,
                                                      ext_scale=ext_scale, centers=centers)
    dev = kp.device

    def per_image(x, i, n_rays):
        if x is None:
            return None
        y = x[i % x.shape[0]:i % x.shape[0] + 1] if x.shape[0] > 1 else x
        return y.to(dev).expand(n_rays, *x.shape[1:])

    rgbs, disps, accs = [], [], []
    for i, c2w in enumerate(render_poses):
        h = H if isinstance(H, int) else int(H[i])
        w = W if isinstance(W, int) else int(W[i])
        rays_o, rays_d = (r.to(dev) for r in rays[i])
        n, idx = len(rays_o), valid_idxs[i].to(dev)
        if bg_imgs is not None and not white_bkgd:
            bg = torch.as_tensor(bg_imgs[bg_indices[i] if bg_indices is not None else 0], dtype=torch.float32)
            rgb_img = F.interpolate(bg.permute(2, 0, 1)[None], size=(h, w), mode='bilinear', align_corners=False)[0]
            rgb_img = rgb_img.permute(1, 2, 0).reshape(h * w, 3).to(dev)
        else:
            rgb_img = torch.full((h * w, 3), 1. if white_bkgd else 0., device=dev)
        disp_img, acc_img = torch.zeros(h * w, device=dev), torch.zeros(h * w, device=dev)
        if n > 0:
            ret = render(h, w, focal, rays=(rays_o, rays_d), chunk=chunk, c2w=c2w[:3, :4], kp_batch=per_image(kp, i, n),
                         skts=per_image(skts, i, n), cyls=per_image(cyls, i, n), cams=per_image(cams, i, n),
                         subject_idxs=per_image(subject_idxs, i, n), bones=per_image(bones, i, n), **render_kwargs)
            rgb_img[idx] = ret['rgb_map'] + (1. - ret['acc_map'][..., None]) * rgb_img[idx]
            disp_img[idx], acc_img[idx] = ret['disp_map'], ret['acc_map']
        rgbs.append(rgb_img.view(h, w, 3).cpu().numpy())
        disps.append(disp_img.view(h, w, 1).cpu().numpy())
        if ret_acc:
            accs.append(acc_img.view(h, w, 1).cpu().numpy())
    rgbs, disps = np.stack(rgbs, 0), np.nan_to_num(np.stack(disps, 0), nan=0.)
    return rgbs, disps, (np.stack(accs, 0) if ret_acc else accs), valid_idxs, bboxes


def render_testset(poses, hwf, args, render_kwargs, kps=None, skts=None, cyls=None, cams=None, bones=None, subject_idxs=None,
                   gt_imgs=None, gt_masks=None, bg_imgs=None, bg_indices=None, vid_base=None, eval_metrics=False,
                   render_factor=0, eval_postfix="", eval_both=False, centers=None):
    """Validation render (+ PSNR / SSIM) with the caster switched to eval mode for its duration."""
    caster = render_kwargs["ray_caster"]
    was_training = caster.training
    caster.eval()
    rgbs, disps, _, valid_idxs, _ = render_path(poses, hwf, args.chunk // 8, render_kwargs, bg_imgs=bg_imgs, bg_indices=bg_indices,
                                                centers=centers, kp=kps, skts=skts, cyls=cyls, bones=bones, cams=cams,
                                                subject_idxs=subject_idxs, render_factor=args.render_factor,
                                                ext_scale=args.ext_scale, white_bkgd=args.white_bkgd)
    caster.train(was_training)
    if not eval_metrics:
        return rgbs, disps
    if gt_masks is not None and gt_masks.sum() < 1:
        gt_masks = None
    metrics = evaluate_metric(rgbs, gt_imgs, disps, gt_masks, valid_idxs, poses, kps, hwf, centers, args.ext_scale,
                              vid_base=vid_base, eval_postfix=eval_postfix, eval_both=eval_both and gt_masks is not None,
                              white_bkgd=args.white_bkgd, render_factor=args.render_factor)
    return metrics, rgbs, disps


def _dist_env():
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    return rank, world, local


def sync_replicas(module, src=0):
    """parameters and buffers of every rank := rank `src`'s (no-op without a process group)"""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            torch.distributed.broadcast(t.data, src=src)


def validate(args, render_data, render_kwargs_test, device, vid_base):
    t = lambda x: torch.tensor(np.ascontiguousarray(x)).to(device)  # noqa: E731
    gt, fg, bgs, bg_idx = render_data["imgs"], render_data["fgs"], render_data["bgs"], render_data.get("bg_idxs")
    masked = gt * fg + (1 - fg) * (bgs[bg_idx] if bg_idx is not None else bgs)
    H, W, focals = render_data["hwf"]
    hwf = (int(H[0]), int(W[0]), focals.astype(np.float32))
    cams = t(render_data["cam_idxs"]) if args.opt_framecode else None
    return render_testset(t(render_data["c2ws"]), hwf, args, render_kwargs_test, cams=cams, kps=t(render_data["kp3d"]),
                          skts=t(render_data["skts"]), bones=t(render_data["bones"]), gt_imgs=masked, gt_masks=fg, vid_base=vid_base,
                          centers=render_data["center"], bg_imgs=bgs, bg_indices=bg_idx, eval_metrics=True, eval_both=True)


def train(argv=None):
    args = parse_args(argv)
    rank, world, local = _dist_env()
    if not torch.cuda.is_available():
        raise RuntimeError("run_nerf.py drives the HIP render path: no GPU visible")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    np.random.seed(0)
    # every rank builds the SAME network: replicas are kept identical by averaging gradients only (core/trainer.py), so the
    # initial weights must agree -- one seed while the model is created, a broadcast from rank 0 for good measure, and only
    # then per-rank streams for the stratified offsets / density noise
    torch.manual_seed(0)

    train_iter, render_data, data_attrs = load_data(args, device=device, rank=rank, world=world)
    logdir = os.path.join(args.basedir, args.expname)
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        with open(os.path.join(logdir, 'args.txt'), 'w') as f:
            for k in sorted(vars(args)):
                f.write(f'{k} = {getattr(args, k)}\n')
        if args.config is not None:
            with open(os.path.join(logdir, 'config.txt'), 'w') as f:
                f.write(open(args.config).read())

    kw_train, kw_test, start, grad_vars, optimizer, loaded = create_raycaster(args, data_attrs, device=device)
    sync_replicas(kw_train['ray_caster'])
    torch.manual_seed(rank + 1)
    trainer = Trainer(args, data_attrs, optimizer, None, kw_train, kw_test, None, device=device)
    if isinstance(loaded, dict):     # a resumed run continues the fused step's random stream: rank 0 wrote its (seed, counter); the
        trainer.resume_rng_state = loaded.get('danbo_rng_state')     # counter is every rank's, the seed each rank's own (load_rng_state_dict)
    global_step = start
    log = open(os.path.join(logdir, 'scalars.jsonl'), 'a') if rank == 0 else None
    t0 = time.time()
    for i in range(start + 1, args.n_iters + 1):
        # statistics cost a device-to-host copy and a stream synchronisation: only the iterations that print ask for them
        want_stats = rank == 0 and i % args.i_print == 0
        loss, stats = trainer.train_batch(next(train_iter), i, global_step, sync_stats=want_stats)
        if rank == 0 and i % args.i_weights == 0:
            trainer.save_nerf(os.path.join(logdir, f'{i:06d}.tar'), global_step)
        if rank == 0 and i % args.i_testset == 0:
            metrics, _, _ = validate(args, render_data, kw_test, device, os.path.join(logdir, f'{args.expname}_val_{i:06d}_'))
            print(f"[VAL] Iter: {i} PSNR: {metrics['psnr']} SSIM: {metrics['ssim']} PSNR_FG: {metrics['psnr_fg']}")
            log.write(json.dumps(dict(iter=i, **{f'Val/{k}': v for k, v in metrics.items()})) + "\n")
        if want_stats:
            print(f"[TRAIN] Iter: {i} Loss: {stats['total_loss']}  PSNR: {stats['psnr']}, Alpha: {stats['alpha']}, "
                  f"{(time.time() - t0) / (i - start):.3f} s/iter")
            log.write(json.dumps(dict(iter=i, **{f'Stats/{k}': v for k, v in stats.items()})) + "\n")
            log.flush()
        global_step += 1
    if world > 1:
        torch.distributed.barrier()
    return trainer


if __name__ == '__main__':
    train()
