"""TEST / BENCH INFRASTRUCTURE -- never imported by the product (danbo-pytorch_amd/).

A multi-threaded torch-CPU restatement of the DANBO eval path, used ONLY as the timed `cpu_baseline` of bench.py (SURVEY 8d:
"the build's CPU restatement (C++/OpenMP or torch-CPU, our code) ... nproc and thread count stated, 3 repetitions, best-of").
It executes the work the reference executes -- every sample through all 24 bones, the assignment net and the full MLP
(reference core/networks/danbo.py:219-339, nerf.py:164-209, gnn_backbone.py:567-629,787-828), in 4096-ray chunks with the
network evaluated `netchunk` = 65 536 rows at a time (core/trainer.py:75-90, danbo.py:201-205) -- on torch's CPU kernels
(MKL GEMMs, OpenMP element-wise ops), which is also what the reference runs on when it is given no GPU.
The cheap per-ray stages (bounds, inverse-CDF resampling) are taken from the numpy oracle.  Checked against the numpy oracle
(and thereby against the reference's goldens) in tests/test_oracle_configs.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

import danbo_oracle as o

J = 24


def _pe(x, L):
    if L == 0:
        return x
    out = [x]
    for l in range(L):
        out += [torch.sin(x * float(2 ** l)), torch.cos(x * float(2 ** l))]
    return torch.cat(out, -1)


class DanboTorchCPU:
    def __init__(self, cfg, sd, rest_pose, netchunk=65536, dtype=torch.float32):
        """dtype = torch.float64: the same graph in double precision on the float32 parameters (bench.py's parity block measures
        the GPU's logits against it: both fp32 sides -- the GPU and this restatement in float32 -- carry their own round-off)"""
        self.cfg, self.netchunk, self.dtype = cfg, netchunk, dtype
        self.np_oracle = o.DanboOracle(cfg, sd, rest_pose)
        self.sd = {k: torch.tensor(np.asarray(v, dtype=np.float32)).to(dtype) for k, v in sd.items() if np.asarray(v).dtype != np.int64}
        self.align = torch.tensor(self.np_oracle.align.astype(np.float32)).to(dtype)

    # ---- network on R x S points
    def _volumes(self, bones):
        return torch.tensor(o.pose_volumes(self.np_oracle.sd, bones, self.cfg['multires_graph'])).to(self.dtype)      # tiny: per pose

    def _assign(self, pf):
        sd = self.sd
        p0 = 'prob_linears.layers.0'
        A = (sd[p0 + '.adj_w'] * sd[p0 + '.adj'])[0]
        out = []
        for a in range(0, pf.shape[0], self.netchunk):
            x = pf[a:a + self.netchunk]
            y = torch.einsum('bkl,klj->bkj', x, sd[p0 + '.lin.weight'])
            y = F.relu(torch.matmul(A, y) + sd[p0 + '.bias'])
            y = F.relu(torch.einsum('bkl,klj->bkj', y, sd['prob_linears.layers.1.weight']) + sd['prob_linears.layers.1.bias'])
            out.append((torch.einsum('bkl,klj->bkj', y, sd['prob_linears.layers.2.weight']) + sd['prob_linears.layers.2.bias'])[..., 0])
        return torch.cat(out, 0)

    def _mlp(self, dens_in, vin):
        sd, cfg = self.sd, self.cfg
        lin = lambda n, x: F.linear(x, sd[n + '.weight'], sd[n + '.bias'])  # noqa: E731
        out = []
        for a in range(0, dens_in.shape[0], self.netchunk):
            x0, v = dens_in[a:a + self.netchunk], vin[a:a + self.netchunk]
            h = x0
            for i in range(cfg['D']):
                h = F.relu(lin(f'pts_linears.{i}', h))
                if i in cfg['skips']:
                    h = torch.cat([x0, h], -1)
            alpha = lin('alpha_linear', h)
            hv = F.relu(lin('views_linears.0', torch.cat([lin('feature_linear', h), v], -1)))
            out.append(torch.cat([lin('rgb_linear', hv), alpha], -1))
        return torch.cat(out, 0)

    def forward(self, pts, rays_d, skts, vols, pose_of_ray, cam_idxs, return_enc=False, valid=None):
        """valid [R,S,24] bool: take the in-volume mask from the caller (a float64 run on float32 points uses the float32 mask)"""
        cfg, sd = self.cfg, self.sd
        R, S = pts.shape[:2]
        # world -> bone -> aligned
        pl = torch.einsum('rjab,rsb->rsja', skts[:, :, :3, :3], pts) + skts[:, None, :, :3, 3]
        pt = torch.einsum('jab,rsjb->rsja', self.align[:, :3, :3], pl) + self.align[None, None, :, :3, 3]
        x = pt / sd['graph_net.axis_scale'].abs()
        if valid is None:
            valid = ~(x.abs() > 1).any(-1)
        win = torch.exp(-2.0 * (x ** 6).sum(-1))
        # factorised 1-D interpolation (zero padding)
        Fc, res = cfg['voxel_feat'], cfg['voxel_res']
        vol = vols.reshape(-1, J, Fc, res, 3)[pose_of_ray]                       # [R,24,F,res,3]
        iy = ((x + 1.0) * res - 1.0) / 2.0
        y0 = torch.floor(iy)
        w1 = iy - y0
        y0 = y0.long()
        feat = torch.zeros(R, S, J, Fc, 3, dtype=self.dtype)
        for k in range(3):
            vk = vol[..., k].permute(0, 1, 3, 2)                                 # [R,24,res,F]
            for yy, ww in ((y0[..., k], 1.0 - w1[..., k]), (y0[..., k] + 1, w1[..., k])):
                ok = (yy >= 0) & (yy < res)
                idx = yy.clamp(0, res - 1).permute(0, 2, 1)[..., None].expand(-1, -1, -1, Fc)        # [R,24,S,F]
                v = torch.gather(vk, 2, idx).permute(0, 2, 1, 3)                                     # [R,S,24,F]
                feat[..., k] += torch.where(ok[..., None], v * ww[..., None], torch.zeros((), dtype=self.dtype))
        pf = (feat.reshape(R, S, J, Fc * 3) * win[..., None]).reshape(R * S, J, Fc * 3)
        logits = self._assign(pf)
        p = (torch.sigmoid(logits) * 1.002 - 0.001) * valid.reshape(R * S, J).to(self.dtype)
        h = (pf * p[..., None]).sum(-2)
        dens_in = _pe(h, cfg['multires_voxel'])
        d = rays_d
        if cfg['ray_tr_type'] == 'root_local':
            d = torch.einsum('rab,rb->ra', skts[:, 0, :3, :3], rays_d)
        if cfg['view_type'] == 'relray':
            d = F.normalize(d, dim=-1)
        vin = _pe(d, cfg['multires_views'])
        if cfg['use_framecode']:
            codes = sd['framecodes.codes.weight']
            idx = torch.as_tensor(np.asarray(cam_idxs).reshape(-1))
            code = codes.mean(0, keepdim=True).expand(R, -1) if int(idx.max()) < 0 else codes[idx.long()]
            vin = torch.cat([vin, code], -1)
        raw = self._mlp(dens_in, vin.repeat_interleave(S, 0)).reshape(R, S, 4)
        if return_enc:
            return raw, logits.reshape(R, S, J), valid
        return raw

    @staticmethod
    def _composite(raw, z, rays_d, B):
        d = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * torch.norm(rays_d, dim=-1, keepdim=True)
        rgb = torch.sigmoid(raw[..., :3]) * 1.002 - 0.001
        alpha = 1.0 - torch.exp(-F.relu(raw[..., 3] / B) * d)
        w = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
        acc = w.sum(-1)
        return dict(rgb_map=(w[..., None] * rgb).sum(-2), acc_map=torch.minimum(acc, torch.ones(())), weights=w, alpha=alpha)

    @torch.no_grad()
    def render(self, ray_batch, skts, bones, cyls, cam_idxs, n_uniques, S, Sf, chunk=4096, stages=False):
        """-> dict(rgb_map, acc_map) as numpy; per-ray skts / bones / cyls like the reference's caster call.
        stages: also the coarse pass' raw [R,S,4] and in-volume mask [R,S,24] (bench.py's parity block)"""
        outs, st = [], []
        R_all = ray_batch.shape[0]
        for a in range(0, R_all, chunk):
            sl = slice(a, min(a + chunk, R_all))
            rb = ray_batch[sl]
            R = rb.shape[0]
            G = max(1, n_uniques * R // R_all) if R_all > chunk else n_uniques
            near, far = self.np_oracle.near_far(rb[:, 0:3], rb[:, 3:6], cyls[sl], skts[sl], rb[:, 6:7], rb[:, 7:8])
            z = o.coarse_z(near, far, S)
            skip = R // G
            pose_of_ray = torch.arange(R) // skip
            vols = self._volumes(bones[sl][::skip])
            t = lambda v: torch.tensor(np.ascontiguousarray(v, dtype=np.float32))  # noqa: E731
            ro, rd, sk = t(rb[:, 0:3]), t(rb[:, 3:6]), t(skts[sl])
            cam = None if cam_idxs is None else cam_idxs[sl]
            zt = t(z)
            raw = self.forward(ro[:, None] + rd[:, None] * zt[..., None], rd, sk, vols, pose_of_ray, cam, return_enc=stages)
            if stages:
                raw, _, valid = raw
                st.append((raw.numpy(), valid.numpy(), (ro[:, None] + rd[:, None] * zt[..., None]).numpy(), pose_of_ray.numpy(),
                           bones[sl][::skip], near, far))
            out0 = self._composite(raw, zt, rd, self.cfg['density_scale'])
            z_all, z_fine, order = o.importance_z(z, out0['weights'].numpy(), Sf)
            zf = t(z_fine)
            raw_f = self.forward(ro[:, None] + rd[:, None] * zf[..., None], rd, sk, vols, pose_of_ray, cam)
            raw_all = torch.gather(torch.cat([raw, raw_f], 1), 1, torch.as_tensor(order)[..., None].expand(-1, -1, 4).long())
            out = self._composite(raw_all, t(z_all), rd, self.cfg['density_scale'])
            outs.append((out['rgb_map'].numpy(), out['acc_map'].numpy(), out0['rgb_map'].numpy()))
        ret = dict(rgb_map=np.concatenate([x[0] for x in outs]), acc_map=np.concatenate([x[1] for x in outs]),
                   rgb0=np.concatenate([x[2] for x in outs]))
        if stages:
            ret.update(raw_coarse=np.concatenate([x[0] for x in st]), valid_coarse=np.concatenate([x[1] for x in st]),
                       pts_coarse=np.concatenate([x[2] for x in st]), chunks=[(len(x[0]), x[3], x[4]) for x in st],
                       near=np.concatenate([x[5] for x in st]), far=np.concatenate([x[6] for x in st]))
        return ret


class AnerfTorchCPU:
    """A-NeRF (nerf_type = nerf: joint-distance cutoff PE + unit bone-local directions -> W = 448 trunk; cutoff-weighted PE of the
    bone-local ray directions + frame code -> colour head; reference core/networks/nerf.py:107-122,222-279,
    core/cutoff_embedder.py:151-214) on torch's CPU kernels -- bench.py's timed cpu_baseline of config 5.  Every sample is
    evaluated (A-NeRF has no in-volume mask).  Checked against the numpy AnerfOracle in tests/test_oracle_anerf.py."""

    def __init__(self, cfg, sd, rest_pose, netchunk=65536, dtype=torch.float32):
        """dtype=torch.float64: the same graph evaluated in float64 on the float32 inputs (bench.py's parity block: the exact result
        of the reference's arithmetic on those points)"""
        self.cfg, self.netchunk, self.dtype = cfg, netchunk, dtype
        self.np_oracle = o.AnerfOracle(cfg, sd, rest_pose)
        self.sd = {k: torch.tensor(np.asarray(v, dtype=np.float32)).to(dtype) for k, v in sd.items() if np.asarray(v).dtype != np.int64}
        self.align = torch.tensor(self.np_oracle.align.astype(np.float32)).to(dtype)

    def _cutoff_pe(self, x, v, cutoff, tau, L, dist):
        """dist: x = v -> [cutoff - v, sin / cos of 2^l (shifted)] * w;  else x = directions [M,72], w from v repeated x 3"""
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

    def _mlp(self, dens_in, vin):
        sd, cfg = self.sd, self.cfg
        lin = lambda n, x: F.linear(x, sd[n + '.weight'], sd[n + '.bias'])  # noqa: E731
        out = []
        for a in range(0, dens_in.shape[0], self.netchunk):
            x0, v = dens_in[a:a + self.netchunk], vin[a:a + self.netchunk]
            h = x0
            for i in range(cfg['D']):
                h = F.relu(lin(f'pts_linears.{i}', h))
                if i in cfg['skips']:
                    h = torch.cat([x0, h], -1)
            alpha = lin('alpha_linear', h)
            hv = F.relu(lin('views_linears.0', torch.cat([lin('feature_linear', h), v], -1)))
            out.append(torch.cat([lin('rgb_linear', hv), alpha], -1))
        return torch.cat(out, 0)

    def forward(self, pts, rays_d, skts, cam_idxs):
        cfg, sd = self.cfg, self.sd
        pts, rays_d, skts = pts.to(self.dtype), rays_d.to(self.dtype), skts.to(self.dtype)
        R, S = pts.shape[:2]
        M = R * S
        pl = torch.einsum('rjab,rsb->rsja', skts[:, :, :3, :3], pts) + skts[:, None, :, :3, 3]
        pt = torch.einsum('jab,rsjb->rsja', self.align[:, :3, :3], pl) + self.align[None, None, :, :3, 3]
        v = pt.norm(dim=-1).reshape(M, J)
        r = F.normalize(pt, dim=-1).reshape(M, 3 * J)
        dens_in = torch.cat([self._cutoff_pe(v, v, sd['pe_fn.cutoff_dist'], float(sd['pe_fn.tau']), cfg['multires'], True), r], -1)
        d = F.normalize(torch.einsum('rjab,rb->rja', skts[:, :, :3, :3], rays_d), dim=-1).reshape(R, 3 * J)
        vin = self._cutoff_pe(d.repeat_interleave(S, 0), v, sd['dirs_pe_fn.cutoff_dist'], float(sd['dirs_pe_fn.tau']),
                              cfg['multires_views'], False)
        if cfg['use_framecode']:
            codes = sd['framecodes.codes.weight']
            idx = torch.as_tensor(np.asarray(cam_idxs).reshape(-1))
            code = codes.mean(0, keepdim=True).expand(R, -1) if int(idx.max()) < 0 else codes[idx.long()]
            vin = torch.cat([vin, code.repeat_interleave(S, 0)], -1)
        return self._mlp(dens_in, vin).reshape(R, S, 4)

    @torch.no_grad()
    def render(self, ray_batch, skts, bones, cyls, cam_idxs, n_uniques, S, Sf, chunk=4096, stages=False):
        """stages: also near / far [R,1], z_coarse [R,S], pts_coarse [R,S,3] (float32) and the coarse pass' raw [R,S,4]"""
        outs, st = [], []
        t = lambda v: torch.tensor(np.ascontiguousarray(v, dtype=np.float32))  # noqa: E731
        for a in range(0, ray_batch.shape[0], chunk):
            sl = slice(a, min(a + chunk, ray_batch.shape[0]))
            rb = ray_batch[sl]
            near, far = o.near_far_cylinder(rb[:, 0:3], rb[:, 3:6], cyls[sl], rb[:, 6:7], rb[:, 7:8], None)
            z = o.coarse_z(near, far, S)
            ro, rd, sk, zt = t(rb[:, 0:3]), t(rb[:, 3:6]), t(skts[sl]), t(z)
            cam = None if cam_idxs is None else cam_idxs[sl]
            pts_c = ro[:, None] + rd[:, None] * zt[..., None]
            raw = self.forward(pts_c, rd, sk, cam).float()
            if stages:
                st.append((near, far, z, pts_c.numpy(), raw.numpy()))
            out0 = DanboTorchCPU._composite(raw, zt, rd, self.cfg['density_scale'])
            z_all, z_fine, order = o.importance_z(z, out0['weights'].numpy(), Sf)
            zf = t(z_fine)
            raw_f = self.forward(ro[:, None] + rd[:, None] * zf[..., None], rd, sk, cam).float()
            raw_all = torch.gather(torch.cat([raw, raw_f], 1), 1, torch.as_tensor(order)[..., None].expand(-1, -1, 4).long())
            out = DanboTorchCPU._composite(raw_all, t(z_all), rd, self.cfg['density_scale'])
            outs.append((out['rgb_map'].numpy(), out['acc_map'].numpy(), out0['rgb_map'].numpy()))
        ret = dict(rgb_map=np.concatenate([x[0] for x in outs]), acc_map=np.concatenate([x[1] for x in outs]),
                   rgb0=np.concatenate([x[2] for x in outs]))
        if stages:
            for i, k in enumerate(("near", "far", "z_coarse", "pts_coarse", "raw_coarse")):
                ret[k] = np.concatenate([x[i] for x in st])
        return ret
