"""Orchestration of the A-NeRF render path (nerf_type = nerf: joint-distance cutoff PE, W = 448 trunk).

Per chunk of whole rays:  k_anerf_encode (HIP) -> 8 dense trunk layers, then alpha + merged feature/view layer as one
225-wide layer (k_linear16: fp32-accurate hi/lo-split products on the fp16 matrix cores, bias + ReLU in the epilogue, skip
layer fed from two buffers without a concatenated copy) -> k_anerf_color (HIP).  Sampling,
compositing and importance resampling are the kernels the DANBO path uses.  A-NeRF has no in-volume mask:
every sample is evaluated, exactly as in the reference (core/networks/nerf.py:107-122).
"""
import torch

from . import hip_ops as ops


class AnerfEngine:
    def __init__(self, cfg, params, align, rows_per_chunk=1 << 22):
        self.cfg, self.p = cfg, params
        self.align = align.float().contiguous()
        self.rows_per_chunk = rows_per_chunk
        self._key_built = None

    def _key(self):
        return tuple((k, v.data_ptr(), v._version) for k, v in sorted(self.p.items()))

    def refresh(self):
        key = self._key()
        if key == self._key_built:
            return
        p, cfg = self.p, self.cfg
        W, VW = cfg["W"], cfg["view_W"]
        self.L, self.Lv = cfg["multires"], cfg["multires_views"]
        self.in_ch = (1 + 2 * self.L) * 24 + 72
        self.b = [p[f"pts_linears.{i}.bias"].contiguous() for i in range(cfg["D"])]
        # layer after a skip: input = [x0 | h], read from the two buffers (no concatenated copy)
        self.skip_into = {i + 1 for i in cfg["skips"]}
        # activations between the layers in k_linear16's fragment order (every load / store instruction one contiguous KB) when
        # the widths allow it; the first layer reads the encoder's rows, the head writes rows for k_anerf_color
        self.frag = W % 32 == 0 and VW + 1 <= 256
        # the density inputs recomputed inside the first and the skip layer from the encoder's compact table (576 instead of
        # 1 728 B per sample, written once and read twice): the kernel instantiation exists for the shipped width
        self.fused_enc = self.frag and W == 448 and 1 <= self.L <= 7 and tuple(cfg["skips"]) == (4,) and self.in_ch == 24 * (1 + 2 * self.L) + 72
        self.layers = [ops.linear16_pack_enc(p[f"pts_linears.{i}.weight"], self.L) if self.fused_enc and (i == 0 or i in self.skip_into)
                       else ops.linear16_pack(p[f"pts_linears.{i}.weight"], K1=self.in_ch if i in self.skip_into else None,
                                              frag_in=(self.frag and i > 0 and i not in self.skip_into, self.frag and i in self.skip_into))
                       for i in range(cfg["D"])]
        self._frag_store = None
        wv = p["views_linears.0.weight"].float()                        # [VW, W + view_ch + code]
        wf, bf = p["feature_linear.weight"].float(), p["feature_linear.bias"].float()
        view_ch = (1 + 2 * self.Lv) * 72
        # feature_linear (no activation) folded into the view layer: one W -> VW GEMM per sample
        # ... and stacked on alpha_linear: rows [0, VW) = view features, row VW = density logit
        # (the parameter-sized products of this refresh on danbo_small_matmul -- float64 accumulation, operands read through their
        # strides -- instead of float64 torch matmuls: no library GEMM anywhere on the A-NeRF path)
        head_w = torch.empty(VW + 1, W, device=wv.device, dtype=torch.float32)
        ops.small_matmul(wv[:, :W], wf, out=head_w[:VW])
        head_w[VW:].copy_(p["alpha_linear.weight"].float())
        self.head = ops.linear16_pack(head_w, frag_in=(self.frag, False))
        self.head_b = torch.cat([torch.zeros(VW, device=head_w.device), p["alpha_linear.bias"].float()]).contiguous()
        self.VW = VW
        b_eff = ops.small_matmul(bf[None, :], wv[:, :W].t(), bias=p["views_linears.0.bias"].float())      # [1, VW]
        # per-joint slices of the view weights: [24, 3 nb, VW] with k = block * 3 + axis (danbo_anerf_view_wj_pack)
        self.w_view_j = ops.anerf_view_wj(p["views_linears.0.weight"], W, self.Lv)
        if cfg["use_framecode"]:
            codes = p["framecodes.codes.weight"].float()
            codes = torch.cat([codes, codes.double().mean(0, keepdim=True).float()], 0)  # last row: mean code (eval, idx < 0)
            self.table = ops.small_matmul(codes, wv[:, W + view_ch:].t(), bias=b_eff[0])
        else:
            self.table = b_eff.reshape(1, VW).contiguous()
        self.rgb_w = p["rgb_linear.weight"].contiguous()
        self.rgb_b = p["rgb_linear.bias"].contiguous()
        self.cutoff = p["pe_fn.cutoff_dist"].contiguous()
        self.cutoff_v = p["dirs_pe_fn.cutoff_dist"].contiguous()
        if not torch.equal(self.cutoff, self.cutoff_v):
            raise NotImplementedError("distance and view cutoffs differ: the shared cutoff weight no longer applies")
        self._key_built = key

    def taus(self):
        t, tv = float(self.p["pe_fn.tau"]), float(self.p["dirs_pe_fn.tau"])
        if t != tv:
            raise NotImplementedError("pe_fn.tau != dirs_pe_fn.tau")
        return t

    # ------------------------------------------------------------------ network forward
    def view_constants(self, rays_d, skts):
        """C [24, R, VW]: per-ray, per-joint part of the view layer (before the per-sample cutoff weight)."""
        self.refresh()
        # one kernel: the cutoff view encoding of (ray, joint) is formed in LDS and multiplied with the joint's [3 nb, VW] weight
        # slice (round 5: danbo_anerf_view_pe_fwd + a permuted copy + torch.bmm, 3.1 ms of library kernels per frame)
        return ops.anerf_view_consts(rays_d, skts, self.Lv, self.w_view_j)

    def _trunk(self, x0):
        if self.frag:
            return self._trunk_frag(x0)
        h = None
        for i, (packed, shape) in enumerate(self.layers):
            if i == 0:
                h = ops.linear16(x0, packed, shape, self.b[i], relu=True)
            elif i in self.skip_into:
                h = ops.linear16(x0, packed, shape, self.b[i], relu=True, x2=h)
            else:
                h = ops.linear16(h, packed, shape, self.b[i], relu=True)
        return h

    def _trunk_frag(self, x0):
        n, W = x0.shape[0], self.cfg["W"]
        need = (n + 127) // 128 * 128 * W
        if self._frag_store is None or self._frag_store[0].numel() < need or self._frag_store[0].device != x0.device:
            self._frag_store = tuple(torch.empty(need, device=x0.device, dtype=torch.float32) for _ in range(2))
        h = None
        for i, (packed, shape) in enumerate(self.layers):
            out = ops.FragBuffer(n, W, x0.device, storage=self._frag_store[i & 1])
            if self.fused_enc and (i == 0 or i in self.skip_into):      # x0: the encoder's compact table
                ops.linear16_enc(x0, packed, shape, self.L, self.b[i], relu=True, x2=h if i else None, out=out)
            elif i == 0:
                ops.linear16(x0, packed, shape, self.b[i], relu=True, out=out)
            elif i in self.skip_into:
                ops.linear16(x0, packed, shape, self.b[i], relu=True, x2=h, out=out)
            else:
                ops.linear16(h, packed, shape, self.b[i], relu=True, out=out)
            h = out
        return h

    def forward_samples(self, rays_o, rays_d, skts, cam_idx=None, z=None, pts=None, view=None, density_only=False):
        """NeRF.forward on R x S samples -> raw [R,S,4] (density_only: [R*S,1])."""
        self.refresh()
        tau = self.taus()
        if pts is not None:
            R, S = pts.shape[:2]
            dev = pts.device
        else:
            R, S = z.shape
            dev = z.device
        raw = torch.empty(R, S, 4, device=dev, dtype=torch.float32)
        dens = torch.empty(R * S, 1, device=dev, dtype=torch.float32) if density_only else None
        C = None if density_only else (self.view_constants(rays_d, skts) if view is None else view)
        # whole rays per chunk, the chunks equally long: 262 144 rays x 48 samples at 1 M rows per chunk are 12.0002 chunks -- thirteen
        # with a last one of 4 rays, each a full chain of launches, when the chunk is simply rows_per_chunk // S rays
        n_chunks = max(1, -(-(R * S) // self.rows_per_chunk))
        rays_per_chunk = max(1, -(-R // n_chunks))
        n_max = min(R, rays_per_chunk) * S
        buf = (torch.empty(n_max, ops.ANERF_ENC_FLOATS if self.fused_enc else self.in_ch, device=dev), torch.empty(n_max, 24, device=dev))
        head_buf = torch.empty(n_max, (self.VW + 4) // 4 * 4, device=dev)     # [view features | density logit | pad to 16 B]
        for r0 in range(0, R, rays_per_chunk):
            nr = min(rays_per_chunk, R - r0)
            n = nr * S
            if self.fused_enc:
                x0, w = ops.anerf_encode_compact(rays_o, rays_d, skts, self.align, self.cutoff, tau, r0 * S, n, z=z, pts=pts, out=buf)
            else:
                x0, w = ops.anerf_encode(rays_o, rays_d, skts, self.align, self.cutoff, tau, self.L, r0 * S, n, z=z, pts=pts,
                                         out=buf)
            h = self._trunk(x0)
            if self.frag and not density_only and S % 16 == 0 and self.VW % 16 == 0 and self.VW <= 240:
                # the colour head as the EPILOGUE of the head layer: its (VW + 1)-wide rows never reach memory (round 5: written by
                # the layer and read back by k_anerf_color, 1.8 GB per 1 M-row chunk and 6 % of the frame)
                ops.linear16_color(h, self.head[0], self.head[1], self.head_b, w, C, self.table,
                                   cam_idx if self.cfg["use_framecode"] else None, r0, S, self.rgb_w, self.rgb_b, raw)
                continue
            head = ops.linear16(h, self.head[0], self.head[1], self.head_b, out=head_buf[:n, :self.VW + 1])
            if density_only:
                dens[r0 * S:r0 * S + n] = head[:, self.VW:]
                continue
            ops.anerf_color(head[:, :self.VW], w, C, self.table, cam_idx if self.cfg["use_framecode"] else None, r0, nr, S,
                            self.rgb_w, self.rgb_b, head[:, self.VW], raw)
        return dens if density_only else raw

    def density(self, pts, skts, bones=None, netchunk=1024 * 64):
        pts = pts.reshape(-1, 1, 3)
        return torch.cat([self.forward_samples(None, None, skts, pts=pts[a:a + netchunk], density_only=True)
                          for a in range(0, pts.shape[0], netchunk)], 0)

    # ------------------------------------------------------------------ RayCaster.render_rays (eval)
    def near_far(self, rays_o, rays_d, cyls, skts=None, near0=0.0, far0=1.0, chunk=4096):
        return ops.near_far_cylinder(rays_o, rays_d, cyls, near0, far0, chunk)

    def render(self, rays_o, rays_d, skts, bones, cyls, cam_idx=None, N_samples=None, N_importance=None, chunk=4096,
               near_far=None, keep=False, **_):
        cfg = self.cfg
        S = N_samples or cfg["N_samples"]
        Sf = N_importance or cfg["N_importance"]
        B = cfg["density_scale"]
        self.refresh()
        near, far = self.near_far(rays_o, rays_d, cyls, skts, 0.0, 1.0, chunk) if near_far is None else near_far
        z = ops.coarse_samples(near, far, S)
        C = self.view_constants(rays_d, skts)
        raw = self.forward_samples(rays_o, rays_d, skts, cam_idx, z=z, view=C)
        out0 = ops.composite(raw, z, rays_d, B)
        z_all, z_fine, order = ops.importance_samples(z, out0["weights"], Sf)
        raw_f = self.forward_samples(rays_o, rays_d, skts, cam_idx, z=z_fine, view=C)
        raw_all = ops.merge_samples(raw, raw_f, order)
        out = ops.composite(raw_all, z_all, rays_d, B)
        ret = dict(rgb_map=out["rgb_map"], disp_map=out["disp_map"], acc_map=out["acc_map"], alpha=out["alpha"],
                   T_i=out["weights"], rgb0=out0["rgb_map"], disp0=out0["disp_map"], acc0=out0["acc_map"],
                   alpha0=out0["alpha"])
        if keep:
            ret.update(near=near, far=far, z_coarse=z, raw_coarse=raw, weights_coarse=out0["weights"], z_fine=z_fine,
                       z_sorted=z_all, sorted_idxs=order, raw_fine=raw_f, raw_sorted=raw_all)
        return ret
