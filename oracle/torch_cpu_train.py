"""TEST / BENCH INFRASTRUCTURE -- never imported by the product (danbo-pytorch_amd/).

The training step of BASELINE config 4 as a torch-CPU restatement with autograd: what the reference's Trainer.train_batch does
(core/trainer.py:257-302: render with perturb / raw_noise_std, L1 losses of both passes + soft-softmax + volume-scale term,
loss.backward(), Adam) when it is given no GPU -- dense, every sample through every bone and the full MLP.  Used only as the
timed `cpu_baseline` of `bench.py --config 4`.
"""
import numpy as np
import torch
import torch.nn.functional as F

import danbo_oracle as o
import torch_cpu

J = 24


def pose_volumes(sd, bones, L):
    """FactorizeGNN forward (reference gnn_backbone.py:683-704), differentiable"""
    r6 = torch.tensor(o.rot6d(bones.numpy()))
    n = torch_cpu._pe(r6, L)
    mask = torch.ones(1, J, 1)
    mask[:, 0] = 0.
    n = n * mask
    g = 'graph_net.layers.'
    for i in range(4):
        if i < 2:
            out = torch.einsum('bkl,klj->bkj', n, sd[f'{g}{i}.lin.weight'])
            out = torch.matmul((sd[f'{g}{i}.adj_w'] * sd[f'{g}{i}.adj'])[0], out) + sd[f'{g}{i}.bias']
        else:
            out = torch.einsum('bkl,klj->bkj', n, sd[f'{g}{i}.weight']) + sd[f'{g}{i}.bias']
        if i == 0:
            out = out + out
        n = F.relu(out) if i < 3 else out
    return n


def make_step(args, cfg, sd, align, batch, n_poses):
    """-> closure running one optimisation step on `batch` (per-ray tensors on the CPU)"""
    model = torch_cpu.DanboTorchCPU(cfg, {k: v.numpy() for k, v in sd.items()}, np.zeros((24, 3)))
    model.align = align.float()
    params = {k: v.clone().float().requires_grad_(v.dtype.is_floating_point and 'adj' != k.split('.')[-1] and 'init' not in k)
              for k, v in sd.items()}
    model.sd = params
    opt = torch.optim.Adam([p for p in params.values() if p.requires_grad], lr=args.lrate)
    S, Sf, B = args.N_samples, args.N_importance, float(args.density_scale)
    ro, rd = batch['rays_o'].float(), batch['rays_d'].float()
    R = ro.shape[0]
    skip = R // n_poses
    pose_of_ray = torch.arange(R) // skip
    skts, bones, cyls = batch['skts'].float(), batch['bones'].float(), batch['cyls'].float()
    target, bgs, cams = batch['target_s'].float(), batch['bgs'].float(), batch['cam_idxs']
    init_scale = torch.tensor(np.abs(sd['graph_net.axis_scale'].numpy()))

    def composite(raw, z, noise):
        d = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * torch.norm(rd, dim=-1, keepdim=True)
        rgb = torch.sigmoid(raw[..., :3]) * 1.002 - 0.001
        alpha = 1.0 - torch.exp(-F.relu(raw[..., 3] / B + noise) * d)
        w = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
        return (w[..., None] * rgb).sum(-2), torch.minimum(w.sum(-1), torch.ones(())), w, alpha

    def step():
        model.np_oracle.sd['graph_net.axis_scale'] = params['graph_net.axis_scale'].detach().numpy()
        near, far = model.np_oracle.near_far(ro.numpy(), rd.numpy(), cyls.numpy(), skts.numpy(), np.zeros((R, 1), np.float32),
                                             np.ones((R, 1), np.float32))
        z = o.coarse_z(near, far, S)
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        lo, hi = np.concatenate([z[:, :1], mids], -1), np.concatenate([mids, z[:, -1:]], -1)
        z = (lo + (hi - lo) * np.random.rand(R, S)).astype(np.float32)
        zt = torch.tensor(z)
        vols = pose_volumes(params, bones[::skip], cfg['multires_graph'])
        raw, lg, va = model.forward(ro[:, None] + rd[:, None] * zt[..., None], rd, skts, vols, pose_of_ray, cams.numpy(), True)
        rgb0, acc0, w0, _ = composite(raw, zt, torch.randn(R, S) * args.raw_noise_std * B)
        z_all, z_fine, order = o.importance_z(z, w0.detach().numpy(), Sf, u=np.random.rand(R, Sf).astype(np.float32))
        zf = torch.tensor(z_fine)
        raw_f, lg_f, va_f = model.forward(ro[:, None] + rd[:, None] * zf[..., None], rd, skts, vols, pose_of_ray, cams.numpy(), True)
        idx = torch.as_tensor(order).long()
        take = lambda a, b: torch.gather(torch.cat([a, b], 1), 1, idx[..., None].expand(-1, -1, a.shape[-1]))  # noqa: E731
        rgb, acc, w, alpha = composite(take(raw, raw_f), torch.tensor(z_all), torch.randn(R, S + Sf) * args.raw_noise_std * B)
        loss = F.l1_loss(rgb + (1 - acc)[:, None] * bgs, target) + F.l1_loss(rgb0 + (1 - acc0)[:, None] * bgs, target)
        labels = ((w * alpha) > 0).float()
        valid = take(va.float(), va_f.float())
        p = torch.sigmoid(take(lg, lg_f)) * 1.002 - 0.001
        loss = loss + args.soft_softmax_loss_coef * (labels - (p * valid).sum(-1)).pow(2).mean()
        sc = params['graph_net.axis_scale'].abs().clamp(min=init_scale * 0.05)
        loss = loss + args.vol_scale_penalty * torch.prod(sc, -1).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return float(loss)

    return step
