"""Training step (reference: core/trainer.py `Trainer.train_batch`, losses :396-422,507-553,
learning-rate decay :189-200) for one process per GPU.

Data-parallel training = every rank renders its own shard of the ray batch (whole images, so
`N_uniques` stays an integer per rank) and the flat fp32 gradient is summed with ONE RCCL
all-reduce per step and divided by the world size; Adam then runs redundantly
on every rank.  No nn.DataParallel, no DistributedDataParallel hooks.

Two step implementations:
  * fused (default for the shipped DANBO configurations): core/train_engine.py -- forward, losses and backward of a batch are
    ONE C call (`danbo_train_step`, ~75 hand-written HIP kernels, captured in a HIP graph), parameters / gradients / Adam
    moments are views of flat buffers, the all-reduce runs on the flat gradient in place, Adam is one kernel;
  * autograd (A-NeRF, exotic flags, or DANBO_TRAIN_PATH=autograd): the caster's differentiable forward (core/train_path.py)
    with torch.autograd and torch.optim.Adam.
"""
import os

import torch
import torch.distributed as dist

from . import train_path
from .utils.lazy import LazyDict


def batchify_rays(rays_flat, chunk=1024 * 64, ray_caster=None, **kwargs):
    """Cast rays chunk by chunk and concatenate every returned tensor (reference :75-90).  Per-ray tensors in `kwargs` are
    sliced with the rays; everything else is passed through.

    A caster of this package in eval mode takes the WHOLE ray set in one call when that is the same computation
    (`RayCaster.render_rays_whole`: one pose behind all rays, DANBO engine; the only thing `chunk` decides in the reference's
    chain -- the chunk-wide nan-mean of the cylinder bounds -- is handed to the bounds kernel): bit-identical to the loop, without
    its per-chunk slicing, replay and concatenation (tests/test_gpu_modules.py)."""
    whole = getattr(ray_caster, 'render_rays_whole', None)
    if whole is not None and rays_flat.shape[0] > chunk:
        out = whole(rays_flat, chunk, **kwargs)
        if out is not None:
            return out
    parts = {}
    for i in range(0, rays_flat.shape[0], chunk):
        kw = {k: (v[i:i + chunk] if torch.is_tensor(v) else v) for k, v in kwargs.items()}
        for k, v in ray_caster(rays_flat[i:i + chunk], **kw).items():
            parts.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in parts.items()}


def render(H, W, focal, chunk=1024 * 64, rays=None, c2w=None, near=0., far=1., center=None, use_viewdirs=False, **kwargs):
    """Rays (given, or the full image of camera `c2w`) -> dict of per-ray maps shaped like the ray grid (reference :96-161).
    near / far are the 0 / 1 placeholders the caster replaces by the cylinder (or per-bone box) bounds."""
    from .utils.ray_utils import get_rays
    if rays is None:
        rays_o, rays_d = get_rays(H, W, focal, c2w, center=None if center is None else center.ravel())
    else:
        rays_o, rays_d = rays
    sh = rays_d.shape
    rays_o, rays_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
    # a caster of this package in eval mode takes the image's rays as they are -- no [R, 8 | 11] ray batch to build and to slice
    # apart again (four strided copies per image): RayCaster.render_rays_whole(..., rays=(o, d), near_far0=(near, far))
    whole = getattr(kwargs.get('ray_caster'), 'render_rays_whole', None)
    if whole is not None and rays_o.shape[0] > chunk and not torch.is_tensor(near) and not torch.is_tensor(far):
        out = whole(None, chunk, rays=(rays_o, rays_d), near_far0=(float(near), float(far)),
                    **{k: v for k, v in kwargs.items() if k != 'ray_caster'})
        if out is not None:
            return {k: (v if v.dim() >= 4 else v.reshape(list(sh[:-1]) + list(v.shape[1:]))) for k, v in out.items()}
    cols = [rays_o, rays_d, near * torch.ones_like(rays_d[:, :1]), far * torch.ones_like(rays_d[:, :1])]
    if use_viewdirs:
        cols.append(rays_d / torch.norm(rays_d, dim=-1, keepdim=True))
    out = batchify_rays(torch.cat(cols, -1), chunk, **kwargs)
    return {k: (v if v.dim() >= 4 else v.reshape(list(sh[:-1]) + list(v.shape[1:]))) for k, v in out.items()}


def decay_optimizer_lrate(lrate, lrate_decay, decay_rate=0.1, optimizer=None, global_step=None, decay_unit=1000):
    """lr = lrate * decay_rate^((optimizer step // unit) / lrate_decay), the step read from the optimizer's state as the
    reference does (core/trainer.py:189-200): it counts the update that has just been made, survives a resume and starts from 0
    after --finetune; `global_step` is only the fall-back before the first update."""
    state = optimizer.state.get(optimizer.param_groups[0]['params'][0], {}) if optimizer is not None else {}
    step = float(state['step']) if 'step' in state else float(global_step or 0)
    new_lrate = lrate * (decay_rate ** ((step // decay_unit) / lrate_decay))
    for g in optimizer.param_groups:
        g['lr'] = new_lrate
    return new_lrate, None


def allreduce_gradients(params, world=None):
    """one flat-bucket all-reduce (sum) of every gradient, then / world"""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = world or dist.get_world_size()
    if world == 1:
        return 0
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        p.grad = flat[off:off + n].view_as(p).clone()
        off += n
    return flat.numel()


class LazyLossDict(LazyDict):
    """The loss terms of a fused step as device tensors, formed on first access: the step leaves them in one [4] tensor (rgb,
    coarse rgb, sum (label - q)^2, volume scale); a loop that does not look at them (every iteration that does not print)
    launches nothing for them."""

    def __init__(self, ls_in, ss_coef, with_ss, with_vol, still_valid=None):
        def fill():
            # `ls` may be the step's STATIC output (a replayed HIP graph writes the same buffer every step): it is snapshotted
            # here, at the first look -- a loop that never looks launches nothing -- and a look AFTER the next step fails loudly
            # instead of returning that step's numbers
            if still_valid is not None and not still_valid():
                raise RuntimeError("the loss terms of this training step were not read before the next step overwrote them: "
                                   "read them (or pass sync_stats=True) before the next train_batch call")
            ls = ls_in.clone() if still_valid is not None else ls_in
            terms = {'rgb_loss': ls[0], 'rgb_loss0': ls[1]}
            if with_ss:
                terms['soft_softmax_loss'] = ls[2] * ss_coef
            if with_vol:
                terms['vol_scale_loss'] = ls[3]
            terms['total_loss'] = sum(terms.values())
            return terms
        super().__init__(fill)


class Trainer:
    def __init__(self, args, data_attrs, optimizer, pose_optimizer=None, render_kwargs_train=None,
                 render_kwargs_test=None, popt_kwargs=None, device=None):
        self.args, self.optimizer, self.device = args, optimizer, device
        self.render_kwargs_train, self.render_kwargs_test = render_kwargs_train, render_kwargs_test
        self.hwf, self.data_attrs = data_attrs.get('hwf'), data_attrs
        self.engine, self.fused_reason, self._comm_stream = None, None, None
        # run the data-parallel form of the step (split phases, both in-place all-reduces, the comm side stream) whenever a
        # process group exists, also at world size 1: how the RCCL branch is exercised on a one-GPU box
        self.collectives_at_world_1 = False
        self.collective_events = None      # a list: train_batch_fused appends the HIP events around its two all-reduces (bench.py)

    def fused_engine(self):
        """the HIP training engine for this caster / optimizer, or None with `self.fused_reason` saying why not"""
        if self.engine is None and self.fused_reason is None:
            from . import anerf_train_engine, train_engine
            caster = self.render_kwargs_train['ray_caster']
            # DANBO -> danbo_train_step, A-NeRF (NeRF with the cutoff encoders) -> danbo_anerf_train_step
            mod, cls = ((anerf_train_engine, anerf_train_engine.AnerfTrainEngine) if type(caster.network).__name__ == 'NeRF'
                        else (train_engine, train_engine.DanboTrainEngine))
            self.fused_reason = ('DANBO_TRAIN_PATH=autograd' if os.environ.get('DANBO_TRAIN_PATH') == 'autograd'
                                 else mod.supported(self.args, caster))
            if self.fused_reason is None:
                self.engine = cls(self.args, caster, self.optimizer)
                self.engine.load_rng_state_dict(getattr(self, 'resume_rng_state', None))
        return self.engine

    def train_batch_fused(self, batch, i=0, global_step=0, sync_stats=True):
        """Trainer.train_batch through danbo_train_step + danbo_adam_step (core/train_engine.py)"""
        args, eng = self.args, self.engine
        caster = self.render_kwargs_train['ray_caster']
        kw = self.render_kwargs_train
        batch = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        G = int(batch['N_uniques'])
        # per-pose rows of the loader's per-ray tensors as strided VIEWS: the engine gathers them (one launch for all inputs)
        pp = lambda x, g: x if x.shape[0] == g else x[::max(x.shape[0] // g, 1)]  # noqa: E731
        S, Sf = int(kw['N_samples']), int(kw['N_importance'])
        grouped = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if grouped else 1
        parallel = world > 1 or (grouped and self.collectives_at_world_1)
        out = eng.forward_backward(batch['rays_o'], batch['rays_d'], pp(batch['skts'], G), pp(batch['bones'], G), pp(batch['cyls'], G),
                                   batch.get('cam_idxs'), batch['target_s'], batch.get('bgs'), S, Sf,
                                   perturb=float(kw['perturb']), raw_noise_std=float(kw['raw_noise_std']), split=parallel)
        if parallel:
            # Two in-place all-reduces on the flat gradient (no packing, no copies).  The first -- pose GNN, assignment net, axis
            # scales: 7 of the 10 MB, final once the pose-GNN adjoint has run -- is launched on a side stream and overlaps the
            # weight-gradient GEMMs of the dense layers (0.4 ms), whose 2.5 MB follow as the second ("overlapped
            # with the tail of backward").
            early, late = eng.grad_buckets()
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            self._comm_stream.wait_stream(cur)
            prof = self.collective_events            # bench.py: a list -> (start, end) events of both all-reduces, per step
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if prof is not None else None
            with torch.cuda.stream(self._comm_stream):
                if ev:
                    ev[0].record()
                dist.all_reduce(early, op=dist.ReduceOp.SUM)
                if ev:
                    ev[1].record()
            eng.finish_backward()
            if ev:
                ev[2].record()
            if late.numel():         # (the A-NeRF step is one phase: its whole gradient travels in the first collective)
                dist.all_reduce(late, op=dist.ReduceOp.SUM)
            if ev:
                ev[3].record()
                prof.append(ev)
            cur.wait_stream(self._comm_stream)
        lr = self.optimizer.param_groups[0]['lr']
        eng.adam_step(lr, 1.0 / world)
        lr, _ = decay_optimizer_lrate(args.lrate, args.lrate_decay, args.lrate_decay_rate, self.optimizer, global_step, args.decay_unit)
        caster.update_embed_fns(global_step, args)
        R = out['rgb_map'].shape[0]
        # out['loss'] is the HIP graph's static output, overwritten by the next replay: snapshotted at the first look (an 11 us copy
        # between every two steps until round 5: 0.8 % of the step for a dictionary the loop reads once in i_print iterations).  An
        # EAGER step (fixed_draws, use_graph off) returns fresh tensors: its terms stay readable for good, like the reference's
        gen = eng.generation
        danbo = type(caster.network).__name__ == 'DANBO'       # (A-NeRF has neither an assignment nor a volume-scale term)
        loss = LazyLossDict(out['loss'], args.soft_softmax_loss_coef / (R * (S + Sf)), danbo and args.agg_type == 'sigmoid',
                            danbo and bool(args.opt_vol_scale),
                            still_valid=(lambda: eng.generation == gen) if eng.outputs_static else None)
        stats = dict(lrate=lr)
        if sync_stats:      # one device-to-host copy; the loop asks for it only when it prints
            bgs = batch.get('bgs', 1.0)
            mse = torch.mean((out['rgb_map'] + (1. - out['acc_map'][..., None]) * bgs - batch['target_s']) ** 2)
            keys = list(loss)
            vals = torch.stack([loss[k] for k in keys] + [out['acc_map'].mean(), mse]).cpu().tolist()
            stats.update({k: v for k, v in zip(keys, vals)}, alpha=vals[-2], psnr=float(-10. * torch.log10(torch.tensor(vals[-1]))))
        self.last_preds = out
        return loss, stats

    def _ray_batch(self, batch):
        ro, rd = batch['rays_o'].float(), batch['rays_d'].float()
        vd = rd / torch.norm(rd, dim=-1, keepdim=True)
        near, far = torch.zeros_like(rd[..., :1]), torch.ones_like(rd[..., :1])
        return torch.cat([ro, rd, near, far, vd], -1)

    def compute_loss(self, batch, preds):
        args = self.args
        caster = self.render_kwargs_train['ray_caster']
        model = caster.network
        bgs = batch.get('bgs', 1.0)
        loss = {'rgb_loss': train_path.nerf_loss(args, preds['rgb_map'], preds['acc_map'], batch['target_s'], bgs)}
        if 'rgb0' in preds:
            loss['rgb_loss0'] = train_path.nerf_loss(args, preds['rgb0'], preds['acc0'], batch['target_s'], bgs,
                                                     loss_weight=args.coarse_weight)
        if 'confd' in preds and args.agg_type == 'sigmoid':
            loss['soft_softmax_loss'] = train_path.soft_softmax_loss(args, model, preds)
        if args.opt_vol_scale:
            loss['vol_scale_loss'] = train_path.volume_scale_loss(args, model)
        loss['total_loss'] = sum(loss.values())
        return loss

    def train_batch(self, batch, i=0, global_step=0, sync_stats=True):
        if self.fused_engine() is not None:
            return self.train_batch_fused(batch, i, global_step, sync_stats)
        args = self.args
        kw = {k: v for k, v in self.render_kwargs_train.items() if k not in ('ray_caster', 'use_viewdirs')}
        caster = self.render_kwargs_train['ray_caster']
        caster.train()
        batch = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        preds = caster(self._ray_batch(batch), kp_batch=batch['kp3d'], skts=batch['skts'], cyls=batch['cyls'],
                       bones=batch['bones'], cams=batch.get('cam_idxs'), N_uniques=batch['N_uniques'], **kw)
        loss = self.compute_loss(batch, preds)
        self.optimizer.zero_grad()
        loss['total_loss'].backward()
        params = [p for g in self.optimizer.param_groups for p in g['params']]
        allreduce_gradients(params)
        self.optimizer.step()
        lr, _ = decay_optimizer_lrate(args.lrate, args.lrate_decay, args.lrate_decay_rate, self.optimizer,
                                      global_step, args.decay_unit)
        caster.update_embed_fns(global_step, args)
        stats = {k: float(v.detach()) for k, v in loss.items()}
        with torch.no_grad():
            mse = torch.mean((preds['rgb_map'] + (1. - preds['acc_map'][..., None]) * batch.get('bgs', 1.0) - batch['target_s']) ** 2)
        stats.update(lrate=lr, alpha=float(preds['acc_map'].mean().detach()), psnr=float(-10. * torch.log10(mse)))
        return loss, stats

    def save_nerf(self, path, global_step):
        """checkpoint in the reference's layout (:597-618): step, optimizer and one state dict per caster sub-module; the
        pose-optimisation entries the reference writes as None are kept so its loader finds every key"""
        caster = self.render_kwargs_train['ray_caster']
        extra = {}
        if self.engine is not None and self.engine.rng_state_dict() is not None:
            # one key more than the reference writes (its loader ignores unknown keys): the fused step's random stream
            extra['danbo_rng_state'] = self.engine.rng_state_dict()
        torch.save({'global_step': global_step, 'optimizer_state_dict': self.optimizer.state_dict(),
                    'poseopt_layer_state_dict': None, 'pose_optimizer_state_dict': None, 'poseopt_anchors': None,
                    **caster.state_dict(), **extra}, path)
        print('Saved checkpoints at', path)
