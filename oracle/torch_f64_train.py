"""TEST INFRASTRUCTURE -- never imported by the product (danbo-pytorch_amd/).

The DANBO training step (render of both passes, the loss terms of Trainer.compute_loss, gradients of every parameter) as a
float64 torch-autograd restatement: the arbiter when two fp32 paths disagree (tests/test_gpu_train_engine.py: the fused
`danbo_train_step` and the autograd path on degenerate batches).  Follows the reference's training forward
(core/networks/danbo.py:219-346, gnn_backbone.py:567-629,683-704,787-828, nerf.py:164-209,281-347, core/trainer.py:396-422,
507-553), dense: every sample through every bone and the full MLP.

What is NOT recomputed in float64, deliberately: the sampling.  Sample depths are detached in the reference
(core/utils/ray_utils.py:287) and importance depths are a chaotic function of fp32 round-off in the coarse weights, so the
caller passes the depths `z_c` / `z_f` and the merge `order` THE PATH UNDER TEST USED; bone-local coordinates and the in-volume
mask are formed from them in the oracle's bit-exact float32 order (danbo_oracle.bone_local / in_volume -- the arithmetic the
kernels reproduce bit for bit), then promoted.  Everything that carries a gradient -- pose GNN, interpolation weights (through
x = pts_t / |axis_scale|), assignment net, masked sigmoid, blend, positional encoding, MLP, both composites, the losses -- is
float64.  Pinned against the reference's own autograd on tests/golden/danbo_perfcap_train.npz and danbo_train.npz
(tests/test_oracle_configs.py::test_f64_training_step_reproduces_the_reference_losses_and_gradients).
"""
import numpy as np
import torch
import torch.nn.functional as F

import danbo_oracle as o

J = 24
F64 = torch.float64        # (tools/diag sets this to torch.float32 to see what plain fp32 arithmetic gives on the same graph)


class Kinks:
    """ReLU units whose SIGN is not determined at fp32 precision.  A ReLU network's gradient is discontinuous where a
    pre-activation crosses zero; when a (sample, unit) pair sits within round-off of that kink, two correct fp32 evaluations land
    on different sides and their gradients differ by that sample's whole contribution through the unit (measured: one pair worth
    4 % of pts_linears.6.weight on a 192-ray batch).  No float64 number is "the" reference there -- but float64 can BRACKET it:
    record every pre-activation in float64 and in float32 (the same graph in both precisions), call a unit ambiguous when the
    two disagree in sign or |z64| < KAPPA |z64 - z32| (KAPPA = 16: the paths under test do not round like this restatement;
    measured on the pin fixture: the reference's own fp32 autograd sits on the other side of a unit at 4 .. 16 x), and differentiate twice more with the ambiguous units' derivative forced to
    1 and to 0.  |g_on - g_off| is what the kink decisions can move each gradient by (step_bracketed)."""
    KAPPA = 16.0

    def __init__(self, masks=None, side=0):
        self.z, self.masks, self.side, self.i = [], masks, side, 0


class _ForcedRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, ambiguous, side):
        ctx.save_for_backward((torch.where(ambiguous, torch.full_like(z, float(side)), (z > 0).to(z.dtype))))
        return torch.relu(z)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.saved_tensors[0], None, None


def _relu(z, kinks):
    if kinks is None:
        return F.relu(z)
    i = kinks.i
    kinks.i += 1
    if kinks.masks is None:
        kinks.z.append(z.detach())
        return F.relu(z)
    return _ForcedRelu.apply(z, kinks.masks[i], kinks.side)


def _pe(x, L):
    out = [x]
    for l in range(L):
        out += [torch.sin(x * float(2 ** l)), torch.cos(x * float(2 ** l))]
    return torch.cat(out, -1)


def _rot6d(aa):
    """pytorch3d axis_angle_to_matrix via quaternion (Taylor branch below 1e-6) -> first two columns, row-major"""
    ang = torch.norm(aa, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - ang * ang / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    q = torch.cat([torch.cos(half), aa * s], -1)
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    return torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * j + k * r),
                        1 - two_s * (i * i + k * k), two_s * (i * k - j * r), two_s * (j * k + i * r)], -1)


def pose_volumes(p, bones, L, kinks=None):
    """FactorizeGNN forward (gnn_backbone.py:683-704) incl. the skip_gcn=False doubling of layer 0"""
    n = _pe(_rot6d(bones), L)
    mask = torch.ones(1, J, 1, dtype=F64, device=bones.device)
    mask[:, 0] = 0.
    n = n * mask
    g = 'graph_net.layers.'
    for i in range(4):
        if i < 2:
            out = torch.einsum('bkl,klj->bkj', n, p[f'{g}{i}.lin.weight'])
            out = torch.matmul((p[f'{g}{i}.adj_w'] * p[f'{g}{i}.adj'])[0], out) + p[f'{g}{i}.bias']
        else:
            out = torch.einsum('bkl,klj->bkj', n, p[f'{g}{i}.weight']) + p[f'{g}{i}.bias']
        if i == 0:
            out = out + out
        n = _relu(out, kinks) if i < 3 else out
    return n


def network(cfg, p, pts_t, valid, vols, pose_of_ray, vin, keep=None, kinks=None):
    """DANBO.forward on [R,S] samples given their aligned bone-local coordinates pts_t [R,S,24,3] (float64 copies of the float32
    values) and the in-volume mask valid [R,S,24] -> raw [R,S,4], logits [R,S,24]"""
    R, S = pts_t.shape[:2]
    x = pts_t / p['graph_net.axis_scale'].abs()
    win = torch.exp(-2.0 * (x.detach() ** 6).sum(-1))                      # .detach(): gnn_backbone.py:804
    Fc, res = cfg['voxel_feat'], cfg['voxel_res']
    vol = vols.reshape(-1, J, Fc, res, 3)[pose_of_ray]                     # [R,24,F,res,3]
    iy = ((x + 1.0) * res - 1.0) / 2.0
    y0 = torch.floor(iy.detach())
    w1 = iy - y0
    y0 = y0.long()
    feat = []
    for k in range(3):
        vk = vol[..., k].permute(0, 1, 3, 2)                               # [R,24,res,F]
        acc = 0.
        for yy, ww in ((y0[..., k], 1.0 - w1[..., k]), (y0[..., k] + 1, w1[..., k])):
            ok = ((yy >= 0) & (yy < res)).to(F64)
            idx = yy.clamp(0, res - 1).permute(0, 2, 1)[..., None].expand(-1, -1, -1, Fc)
            v = torch.gather(vk, 2, idx).permute(0, 2, 1, 3)               # [R,S,24,F]
            acc = acc + v * (ww * ok)[..., None]
        feat.append(acc)
    pf = (torch.stack(feat, -1).reshape(R, S, J, Fc * 3) * win[..., None]).reshape(R * S, J, Fc * 3)
    p0 = 'prob_linears.layers.0'
    y = torch.einsum('bkl,klj->bkj', pf, p[p0 + '.lin.weight'])
    y = _relu(torch.matmul((p[p0 + '.adj_w'] * p[p0 + '.adj'])[0], y) + p[p0 + '.bias'], kinks)
    y = _relu(torch.einsum('bkl,klj->bkj', y, p['prob_linears.layers.1.weight']) + p['prob_linears.layers.1.bias'], kinks)
    logits = (torch.einsum('bkl,klj->bkj', y, p['prob_linears.layers.2.weight']) + p['prob_linears.layers.2.bias'])[..., 0]
    pr = (torch.sigmoid(logits) * 1.002 - 0.001) * valid.reshape(R * S, J).to(F64)
    h = (pf * pr[..., None]).sum(-2)
    if keep is not None:
        keep.append(h)
    x0 = _pe(h, cfg['multires_voxel'])
    lin = lambda n, t: F.linear(t, p[n + '.weight'], p[n + '.bias'])  # noqa: E731
    t = x0
    for i in range(cfg['D']):
        t = _relu(lin(f'pts_linears.{i}', t), kinks)
        if i in cfg['skips']:
            t = torch.cat([x0, t], -1)
    alpha = lin('alpha_linear', t)
    hv = _relu(lin('views_linears.0', torch.cat([lin('feature_linear', t), vin.repeat_interleave(S, 0)], -1)), kinks)
    raw = torch.cat([lin('rgb_linear', hv), alpha], -1)
    return raw.reshape(R, S, 4), logits.reshape(R, S, J)


def composite(raw, z, rays_d, B, noise=None, clamped=None):
    d = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * torch.norm(rays_d, dim=-1, keepdim=True)
    rgb = torch.sigmoid(raw[..., :3]) * 1.002 - 0.001
    s = raw[..., 3] / B
    if noise is not None:
        s = s + noise
    alpha = 1.0 - torch.exp(-F.relu(s) * d)
    w = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    sw = w.sum(-1)
    if clamped is None:
        acc = torch.minimum(sw, torch.ones((), dtype=F64, device=raw.device))
    else:       # the branch of min(sum w, 1) the path under test took, per ray (see step())
        acc = torch.where(clamped, torch.ones_like(sw), sw)
    return (w[..., None] * rgb).sum(-2), acc, w, alpha


def step(cfg, coef, sd, align, init_scale, batch, z_c, z_f, order, n_uniques, noise_c=None, noise_f=None, device='cpu', Sf=None,
         u_rand=None, clamped_c=None, clamped_f=None, debug=None, kinks=None):
    """One forward + backward.
    cfg: synthetic.model_config(...); coef: dict(loss_fn 'L1' | 'MSE', use_background, rgb_loss_coef, coarse_weight,
    soft_softmax_loss_coef, vol_scale_penalty (0 = term off)); sd: name -> numpy array (float32 parameters and 0/1 adjacency
    buffers); align [24,4,4] float32; init_scale [24,3]; batch: numpy rays_o, rays_d [R,3], skts [R,24,4,4], bones [R,24,3] (per
    ray), target, bgs [R,3], cam_idxs [R]; z_c [R,S], z_f [R,Sf] float32 depths and order [R,S+Sf] (argsort of [z_c | z_f]) of the
    path under test.
    clamped_c / clamped_f [R] bool or None: which rays of the coarse / final composite the path under test saw at acc_map = 1,
    i.e. on the constant branch of acc = min(sum w, 1) (nerf.py:344).  An opaque ray sits EXACTLY on that kink: its last sample
    has delta = 1e10, so alpha = 1 and sum w = 1 + O(1e-10) in exact arithmetic (the +1e-10 inside the transmittance), which
    float64 resolves (always clamped: no gradient through acc) and float32 does not (1 - a + 1e-10 == 1 - a; sum w lands on
    either side of 1 by round-off, ray by ray).  Which side is taken is not a property of the function being differentiated;
    like the sampling, the decision is taken from the path under test.  None: float64's own min().
    -> dict(loss: name -> float, grads: name -> float64 numpy, rgb_map, rgb0, acc_map, labels)"""
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    T = lambda a: torch.tensor(np.asarray(a), dtype=F64, device=device)  # noqa: E731
    ro, rd, skts, bones = f32(batch['rays_o']), f32(batch['rays_d']), f32(batch['skts']), f32(batch['bones'])
    R = ro.shape[0]
    skip = R // n_uniques
    pose_of_ray = torch.arange(R, device=device) // skip
    p = {}
    for k, v in sd.items():
        v = np.asarray(v)
        if v.dtype.kind != 'f':
            continue
        leaf = k.split('.')[-1] != 'adj' and 'init_scale' not in k and 'tau' not in k
        p[k] = T(v).requires_grad_(leaf)
    scale32 = f32(sd['graph_net.axis_scale'])

    def geometry(z):
        pts = o.sample_points(ro, rd, f32(z))
        pts_t = o.bone_local(pts, skts, f32(align))
        _, valid = o.in_volume(pts_t, scale32)
        return T(pts_t), torch.tensor(valid, device=device)

    d = rd
    if cfg['ray_tr_type'] == 'root_local':
        d = np.einsum('rab,rb->ra', skts[:, 0, :3, :3].astype(np.float64), rd.astype(np.float64))
    d = T(d)
    if cfg['view_type'] == 'relray':
        d = F.normalize(d, dim=-1)
    vin = _pe(d, cfg['multires_views'])
    if cfg['use_framecode']:
        vin = torch.cat([vin, p['framecodes.codes.weight'][torch.as_tensor(np.asarray(batch['cam_idxs']).reshape(-1), device=device).long()]], -1)
    vols = pose_volumes(p, T(bones[::skip]), cfg['multires_graph'], kinks)
    B = float(cfg['density_scale'])
    rdt = T(rd)
    zc, zf = T(z_c), (None if z_f is None else T(z_f))
    pt_c, va_c = geometry(z_c)
    hs = [] if debug is not None else None
    raw_c, lg_c = network(cfg, p, pt_c, va_c, vols, pose_of_ray, vin, hs, kinks)
    as_mask = lambda m: None if m is None else torch.as_tensor(np.asarray(m), device=device).bool()  # noqa: E731
    raw_c1 = raw_c * 1.0            # (own graph nodes for the two composites' inputs: `debug` reads the gradients arriving there)
    rgb0, acc0, w0, _ = composite(raw_c1, zc, rdt, B, None if noise_c is None else T(noise_c), as_mask(clamped_c))
    if z_f is None:     # pin mode only (no path under test): resample from this restatement's own coarse weights, rounded to float32
        _, z_f, order = o.importance_z(f32(z_c), w0.detach().cpu().numpy().astype(np.float32), int(Sf), u=u_rand)
        zf = T(z_f)
    pt_f, va_f = geometry(z_f)
    raw_f, lg_f = network(cfg, p, pt_f, va_f, vols, pose_of_ray, vin, hs, kinks)
    idx = torch.as_tensor(np.asarray(order), device=device).long()
    take = lambda a, b: torch.gather(torch.cat([a, b], 1), 1, idx[..., None].expand(-1, -1, a.shape[-1]))  # noqa: E731
    z_all = torch.gather(torch.cat([zc, zf], 1), 1, idx)
    raw_all1 = take(raw_c, raw_f) * 1.0
    rgb, acc, w, alpha = composite(raw_all1, z_all, rdt, B, None if noise_f is None else T(noise_f), as_mask(clamped_f))
    target, bgs = T(batch['target_s']), T(batch['bgs'])
    fn = {'L1': F.l1_loss, 'MSE': F.mse_loss}[coef.get('loss_fn', 'L1')]
    bg = (lambda c, a: c + (1.0 - a)[:, None] * bgs) if coef.get('use_background', True) else (lambda c, a: c)
    loss = {'rgb_loss': fn(bg(rgb, acc), target) * coef['rgb_loss_coef'],
            'rgb_loss0': fn(bg(rgb0, acc0), target) * coef['rgb_loss_coef'] * coef['coarse_weight']}
    labels = ((w * alpha) > 0).to(F64)
    valid = take(va_c.to(F64), va_f.to(F64))
    pr = torch.sigmoid(take(lg_c, lg_f)) * 1.002 - 0.001
    loss['soft_softmax_loss'] = coef['soft_softmax_loss_coef'] * (labels - (pr * valid).sum(-1)).pow(2).mean()
    if coef.get('vol_scale_penalty', 0.):
        sc = p['graph_net.axis_scale'].abs().clamp(min=T(init_scale) * 0.05)
        loss['vol_scale_loss'] = coef['vol_scale_penalty'] * torch.prod(sc, -1).sum()
    loss['total_loss'] = sum(loss.values())
    leaves = {k: v for k, v in p.items() if v.requires_grad}
    if debug is not None:
        d1, d2 = torch.autograd.grad(loss['total_loss'], [raw_c1, raw_all1], retain_graph=True)
        debug.update(d_raw_coarse=d1.cpu().numpy(), d_raw_sorted=d2.cpu().numpy(), raw_coarse=raw_c.detach().cpu().numpy(),
                     raw_sorted=raw_all1.detach().cpu().numpy(), acc0=acc0.detach().cpu().numpy(), acc=acc.detach().cpu().numpy())
    if debug is not None:
        gh = torch.autograd.grad(loss['total_loss'], hs, retain_graph=True)
        debug.update(h_c=hs[0].detach().cpu().numpy(), h_f=hs[1].detach().cpu().numpy(), d_h_c=gh[0].cpu().numpy(), d_h_f=gh[1].cpu().numpy(),
                     any_c=va_c.any(-1).reshape(-1).cpu().numpy(), any_f=va_f.any(-1).reshape(-1).cpu().numpy())
    grads = torch.autograd.grad(loss['total_loss'], list(leaves.values()), allow_unused=True)
    return dict(loss={k: float(v.detach()) for k, v in loss.items()},
                grads={k: (np.zeros(tuple(v.shape)) if g is None else g.cpu().numpy()) for (k, v), g in zip(leaves.items(), grads)},
                rgb_map=rgb.detach().cpu().numpy(), rgb0=rgb0.detach().cpu().numpy(), acc_map=acc.detach().cpu().numpy(),
                labels=labels.cpu().numpy(), weights0=w0.detach().cpu().numpy(), z_f=np.asarray(z_f), order=np.asarray(order))


def step_bracketed(*args, **kw):
    """step() in float64 + the bracket of the ReLU-kink decisions (class Kinks): -> step()'s dict with two more entries,
    `bracket`: name -> max |g_on - g_off| of the tensor, `ambiguous`: (units whose sign fp32 does not determine, units)."""
    global F64
    k32, k64 = Kinks(), Kinks()
    F64 = torch.float32
    try:
        step(*args, kinks=k32, **kw)
    finally:
        F64 = torch.float64
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
