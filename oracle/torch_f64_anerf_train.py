"""TEST INFRASTRUCTURE -- never imported by the product (danbo-pytorch_amd/).

The A-NeRF (nerf_type = nerf) training step -- both passes, the two rgb losses of Trainer.compute_loss, the gradient of every
parameter -- as a float64 torch-autograd restatement: the arbiter for `danbo_anerf_train_step` and the autograd path
(tests/test_gpu_anerf_train.py).  Follows the reference's training forward: SamplePointsEmbedder.encode_pts with RelDist / VecNorm
(core/encoders.py:424-450,630-651,774-795), CutoffEmbedder._embed (core/cutoff_embedder.py:151-214), NeRF.forward / inference /
encode_views (core/networks/nerf.py:107-122,176-209,252-279), raw2outputs (:281-347), nerf_loss (core/trainer.py:396-422).

As in torch_f64_train.py (the DANBO arbiter), what is not a property of the differentiated function comes from the path under test:
the sample depths `z_c` / `z_f` and the merge `order` (detached in the reference, chaotic in the coarse weights' round-off), and the
density noise if there is any.  The bone-local coordinates are formed from the float32 points in the oracle's float32 operation order
and promoted; everything behind them -- encodings, trunk, heads, composites, losses -- is float64.  ReLU kinks are bracketed with
torch_f64_train.Kinks (a unit whose sign fp32 round-off does not determine moves the gradient by its sample's whole contribution).
Pinned against the reference's own autograd on tests/golden/anerf_train.npz (tests/test_oracle_anerf.py).
"""
import numpy as np
import torch

import danbo_oracle as o
import torch_f64_train as t64
from torch_f64_train import Kinks, _relu

J = 24


def _dtype():
    return t64.F64


def _cutoff_pe(x, v, cutoff, tau, L, dist):
    """dist: the joint-distance encoding of v [M,24]; else the view encoding of directions x [M,72] (weights from v repeated x 3)"""
    if dist:
        inp = cutoff - v
        base = inp * (2.0 / cutoff) - 1.0
        w = 1.0 - torch.sigmoid(tau * (v - cutoff))
    else:
        inp = base = x
        w = 1.0 - torch.sigmoid(tau * (v.repeat_interleave(3, -1) - cutoff.repeat_interleave(3, -1)))
    blocks = [inp]
    for l in range(L):
        f = base * float(2 ** l)
        blocks += [torch.sin(f), torch.cos(f)]
    return (torch.stack(blocks, -2) * w[..., None, :]).reshape(x.shape[0], -1)


def network(cfg, p, pt, rays_d, skts_ray, cam_idx, tau, kinks=None):
    """NeRF.forward on bone-local aligned points pt [R,S,24,3] (float32 values, promoted) -> raw [R,S,4]"""
    dt = _dtype()
    R, S = pt.shape[:2]
    M = R * S
    lin = lambda n, x: torch.nn.functional.linear(x, p[n + '.weight'], p[n + '.bias'])  # noqa: E731
    v = pt.norm(dim=-1).reshape(M, J)
    r = torch.nn.functional.normalize(pt, dim=-1).reshape(M, 3 * J)
    cut = p['pe_fn.cutoff_dist']
    x0 = torch.cat([_cutoff_pe(v, v, cut, tau, cfg['multires'], True), r], -1)
    d = torch.nn.functional.normalize(torch.einsum('rjab,rb->rja', skts_ray[:, :, :3, :3], rays_d), dim=-1).reshape(R, 3 * J)
    vin = _cutoff_pe(d.repeat_interleave(S, 0), v, p['dirs_pe_fn.cutoff_dist'], tau, cfg['multires_views'], False)
    if cfg['use_framecode']:
        code = p['framecodes.codes.weight'][torch.as_tensor(np.asarray(cam_idx).reshape(-1), device=pt.device).long()]
        vin = torch.cat([vin, code.repeat_interleave(S, 0)], -1)
    h = x0
    for i in range(cfg['D']):
        h = _relu(lin(f'pts_linears.{i}', h), kinks)
        if i in cfg['skips']:
            h = torch.cat([x0, h], -1)
    alpha = lin('alpha_linear', h)
    hv = _relu(lin('views_linears.0', torch.cat([lin('feature_linear', h), vin], -1)), kinks)
    return torch.cat([lin('rgb_linear', hv), alpha], -1).reshape(R, S, 4).to(dt)


def composite(raw, z, rays_d, B, noise=None, clamped=None):
    """NeRF.raw2outputs (nerf.py:281-347) -> rgb_map, acc_map (the two outputs with a gradient), weights, alpha.  clamped [R] bool or
    None: the branch of acc = min(sum w, 1) the path under test took per ray (an opaque ray sits exactly on that kink, see
    torch_f64_train.step); None: float64's own min()"""
    d = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * torch.norm(rays_d, dim=-1, keepdim=True)
    rgb = torch.sigmoid(raw[..., :3]) * 1.002 - 0.001
    dens = raw[..., 3] / B
    if noise is not None:
        dens = dens + noise
    alpha = 1.0 - torch.exp(-torch.relu(dens) * d)
    w = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    sw = w.sum(-1)
    acc = torch.clamp(sw, max=1.0) if clamped is None else torch.where(clamped, torch.ones_like(sw), sw)
    return dict(rgb_map=(w[..., None] * rgb).sum(-2), acc_map=acc, weights=w, alpha=alpha)


def step(cfg, sd, rest_pose, batch, z_c, z_f, order, args, noise_c=None, noise_f=None, kinks=None, device='cpu', clamped_c=None,
         clamped_f=None):
    """batch: dict(rays_o, rays_d [R,3], skts [G,24,4,4] per pose, cam_idx [R], target [R,3], bgs [R,3] or None); z_c [R,S],
    z_f [R,Sf], order [R,S+Sf] from the path under test; args: dict(loss_fn, use_background, rgb_loss_coef, coarse_weight,
    density_scale, tau) -> dict(loss={rgb_loss, rgb_loss0, total_loss}, grads={name: float64 array}, rgb_map, acc_map, rgb0)"""
    dt = _dtype()
    T = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=dt, device=device)  # noqa: E731
    as_mask = lambda m: None if m is None else torch.as_tensor(np.asarray(m, dtype=bool), device=device)  # noqa: E731
    names = [k for k, v in sd.items() if np.asarray(v).dtype.kind == 'f' and not k.endswith('.tau') and 'cutoff_dist' not in k]
    p = {k: T(sd[k]).requires_grad_(True) for k in names}
    p['pe_fn.cutoff_dist'], p['dirs_pe_fn.cutoff_dist'] = T(sd['pe_fn.cutoff_dist']), T(sd['dirs_pe_fn.cutoff_dist'])
    ro, rd = np.asarray(batch['rays_o'], np.float32), np.asarray(batch['rays_d'], np.float32)
    R = ro.shape[0]
    G = batch['skts'].shape[0]
    pose = np.arange(R) // (R // G)
    skts_ray = np.asarray(batch['skts'], np.float32)[pose]
    align = o.bone_align_transforms(rest_pose).astype(np.float32)
    tau, B = float(args['tau']), float(args['density_scale'])

    def pass_(z):
        z = np.asarray(z, np.float32)
        pts = o.sample_points(ro, rd, z)                                       # float32, the kernels' two roundings
        pt = o.bone_local(pts, skts_ray, align)                              # [R,S,24,3] float32: the bit-exact chain
        return network(cfg, p, T(pt), T(rd), T(skts_ray), batch.get('cam_idx'), tau, kinks)

    raw_c, raw_f = pass_(z_c), pass_(z_f)
    out0 = composite(raw_c, T(z_c), T(rd), B, None if noise_c is None else T(noise_c), as_mask(clamped_c))
    idx = torch.as_tensor(np.asarray(order), device=device).long()
    raw_all = torch.gather(torch.cat([raw_c, raw_f], 1), 1, idx[..., None].expand(-1, -1, 4))
    z_all = torch.gather(torch.cat([T(z_c), T(z_f)], 1), 1, idx)
    out = composite(raw_all, z_all, T(rd), B, None if noise_f is None else T(noise_f), as_mask(clamped_f))
    target = T(batch['target'])
    bgs = T(batch['bgs']) if batch.get('bgs') is not None else torch.ones((), dtype=dt, device=device)

    def nerf_loss(rgb, acc, w):
        if args['use_background']:
            rgb = rgb + (1.0 - acc)[..., None] * bgs
        d = rgb - target
        return (d.abs().mean() if args['loss_fn'] == 'L1' else (d * d).mean()) * w * float(args['rgb_loss_coef'])
    loss = {'rgb_loss': nerf_loss(out['rgb_map'], out['acc_map'], 1.0),
            'rgb_loss0': nerf_loss(out0['rgb_map'], out0['acc_map'], float(args['coarse_weight']))}
    loss['total_loss'] = loss['rgb_loss'] + loss['rgb_loss0']
    loss['total_loss'].backward()
    grads = {k: (v.grad.cpu().numpy().astype(np.float64) if v.grad is not None else np.zeros(v.shape)) for k, v in p.items() if v.requires_grad}
    return dict(loss={k: float(v.detach()) for k, v in loss.items()}, grads=grads, rgb_map=out['rgb_map'].detach().cpu().numpy(),
                acc_map=out['acc_map'].detach().cpu().numpy(), rgb0=out0['rgb_map'].detach().cpu().numpy())


def step_bracketed(*args, **kw):
    """step() in float64 + the bracket of the ReLU-kink decisions (torch_f64_train.Kinks): `bracket`: name -> max |g_on - g_off|,
    `ambiguous`: (units whose sign fp32 does not determine, units)"""
    k32, k64 = Kinks(), Kinks()
    t64.F64 = torch.float32
    try:
        step(*args, kinks=k32, **kw)
    finally:
        t64.F64 = torch.float64
    ret = step(*args, kinks=k64, **kw)
    masks = []
    for a, b in zip(k32.z, k64.z):
        a = a.to(torch.float64)
        masks.append(((a > 0) != (b > 0)) | (b.abs() < Kinks.KAPPA * (b - a).abs()))
    g_on = step(*args, kinks=Kinks(masks, 1), **kw)['grads']
    g_off = step(*args, kinks=Kinks(masks, 0), **kw)['grads']
    ret['bracket'] = {n: float(np.abs(g_on[n] - g_off[n]).max()) for n in g_on}
    ret['ambiguous'] = (int(sum(int(m.sum()) for m in masks)), int(sum(m.numel() for m in masks)))
    return ret
