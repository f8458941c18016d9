"""Orchestration of the HIP kernels for the DANBO render path (eval forward).

`DanboEngine` owns nothing but *views* of the model parameters plus two derived buffers
(MFMA-packed MLP weights, empty-space constants) that are rebuilt whenever a parameter's
version counter changes.  All work is enqueued on the current HIP stream; no call in here
synchronises with the host.

Stage map (reference file:line in include/danbo_hip.h):
    pose_volumes -> [near/far -> coarse z] -> K1a cull+compact -> K1b+K2 gather/assign/blend
    -> view consts + raw fill -> K3 PE+MLP (scatter) -> K4 composite -> importance -> fine pass
"""
import torch

from . import hip_ops as ops


class DanboEngine:
    def __init__(self, cfg, params, align, buffers=None, mlp_mode="f16split"):
        """cfg: dict (see core/utils/synthetic.model_config); params: name -> CUDA tensor using the
        reference's state_dict names; align: [24,4,4] bone-align transforms."""
        self.cfg = cfg
        self.p = params
        self.align = align.float().contiguous()
        self._packed_key = None
        self.packed = self.wrt = self.empty_consts = None
        self.mean_code = None
        # optional per-kernel timing (bench.py): name -> list of (start_event, end_event, count_tensor)
        self.profile = None
        # render() is ONE stream-ordered chain (round 4 measured the side-stream form -- pose volumes / view constants beside the
        # bounds -> cull chain -- slower: 4.696 vs 4.634 ms; round 5 removed it: nothing of the frame runs beside K2)
        self.skip_flat_rays = True   # render(): no resampling for rays that cannot meet a volume (tests switch it off to compare)
        # "f16split": k_pe_mlp32 (3 fp16 MFMAs per fp32-accurate product; mlp_form 16: k_pe_mlp16, the 16x16x32 form of the same
        # arithmetic -- kept for A/B, `render_frame_c` needs 32); "fp32": k_pe_mlp (exact fp32 MFMA)
        self.mlp_form = 32
        assert mlp_mode in ("f16split", "fp32")
        self.mlp_mode = mlp_mode
        # True: re-order the compacted rows by bone set in front of K2 (k_group.hip).  Off since the end of round 4: K2 gains 2 x 20 us
        # from it, the grouping costs 2 x 27 us and scatters K3's rows (tools/ab_engine_switch.py: 4.734 ms with, 4.712 ms without)
        self.group_rows = False

    # ------------------------------------------------------------------ derived buffers
    def _key(self):
        return tuple((k, v.data_ptr(), v._version) for k, v in sorted(self.p.items()))

    def refresh(self):
        key = self._key()
        if key == self._packed_key and getattr(self, "_built_mode", None) == (self.mlp_mode, self.mlp_form):
            return
        p = self.p
        dev = p["alpha_linear.weight"].device
        q = self._equalized(p) if self.mlp_mode == "f16split" else p
        self.pts_w = [q[f"pts_linears.{i}.weight"] for i in range(8)]
        self.pts_b = [q[f"pts_linears.{i}.bias"].contiguous() for i in range(8)]
        self.packed, self.wrt = ops.mlp_pack(self.pts_w, q["feature_linear.weight"], q["views_linears.0.weight"])
        self.packed16, self.views_b16 = ops.mlp16_pack(self.pts_w, q["feature_linear.weight"],
                                                       q["feature_linear.bias"], q["views_linears.0.weight"],
                                                       q["views_linears.0.bias"], form=self.mlp_form)
        self.alpha_w = q["alpha_linear.weight"].reshape(-1).contiguous()
        self.alpha_b = q["alpha_linear.bias"].contiguous()
        self.feature_b = q["feature_linear.bias"].contiguous()
        self.views_b = q["views_linears.0.bias"].contiguous()
        self.rgb_w = q["rgb_linear.weight"].contiguous()
        self.rgb_b = q["rgb_linear.bias"].contiguous()
        # power of two the view layer's pre-activations (cview, empty_consts[:128]) are multiplied by, see _equalized
        self.view_scale = q.get("_view_scale", torch.ones((), device=dev))
        g = "graph_net.layers."
        self.gw = dict(
            w0=p[g + "0.lin.weight"].contiguous(), adjw0=(p[g + "0.adj_w"] * p[g + "0.adj"])[0].contiguous(),
            b0=p[g + "0.bias"].contiguous(),
            w1=p[g + "1.lin.weight"].contiguous(), adjw1=(p[g + "1.adj_w"] * p[g + "1.adj"])[0].contiguous(),
            b1=p[g + "1.bias"].contiguous(),
            w2=p[g + "2.weight"].contiguous(), b2=p[g + "2.bias"].reshape(24, -1).contiguous(),
            w3=p[g + "3.weight"].contiguous(), b3=p[g + "3.bias"].reshape(24, -1).contiguous())
        a = "prob_linears.layers."
        self.aw = dict(
            w0=p[a + "0.lin.weight"].contiguous(), adjw=(p[a + "0.adj_w"] * p[a + "0.adj"])[0].contiguous(),
            b0=p[a + "0.bias"].contiguous(), w1=p[a + "1.weight"].contiguous(),
            b1=p[a + "1.bias"].reshape(24, -1).contiguous(), w2=p[a + "2.weight"].reshape(24, -1).contiguous(),
            b2=p[a + "2.bias"].reshape(-1).contiguous())
        if self.mlp_mode == "f16split":
            ops.check_smpl_adjacency(p[a + "0.adj"])
        self.assign16 = ops.assign16_pack(self.aw)
        self.axis_scale = p["graph_net.axis_scale"].contiguous()
        if self.cfg["use_framecode"]:
            self.framecodes = p["framecodes.codes.weight"].contiguous()
            self.mean_code = self.framecodes.mean(0).contiguous()
            vb = self.views_b16 if self.mlp_mode == "f16split" else self.views_b
            self.code_table = ops.view_code_table(self.framecodes, self.mean_code, self.cfg["multires_views"],
                                                  self.wrt, vb)
        else:
            self.framecodes = self.mean_code = self.code_table = None
        # empty-space constants: one zero row through the MLP without the per-ray view term
        h0 = torch.zeros(1, ops.H_STRIDE, device=dev)
        scratch_raw = torch.empty(1, 4, device=dev)
        self.empty_consts = self._mlp(h0, 1, None, scratch_raw, aux=True).reshape(-1).contiguous()
        self.flat_rays_ok = self._flat_rays_ok()
        self._built_mode = (self.mlp_mode, self.mlp_form)
        self._packed_key = key

    def _flat_rays_ok(self):
        """The two statements danbo_flat_rays (include/danbo_hip.h) needs about these weights, once per weight update (one host
        sync): (a) the empty-space density -- the MLP's density for a zero feature row -- is <= 0; (b) the empty-space colour
        logits are finite for every flagged ray: danbo_ray_bone_mask flags no ray whose view-direction inputs can exceed
        DANBO_RAY_FLAT_VMAX in magnitude (normalised directions, sines and cosines: 1), so |cview_c| <= vmax * sum_i |W_ci| +
        |b_c| (or the largest entry of the per-camera table), the hidden row is at most |empty_pre_c| + that, and the logits at
        most sum_c |rgb_w_c| * hidden_c + |rgb_b| -- finite and far from overflow means no inf - inf, no NaN.  Then a ray that
        cannot meet a volume is a ray of constants."""
        cfg = self.cfg
        if self.mlp_mode != "f16split":
            return False
        ec = self.empty_consts
        Cpe = 3 * (1 + 2 * cfg["multires_views"])
        vmax = 1.0 if cfg["view_type"] == "relray" else ops.RAY_FLAT_VMAX
        a_max = self.wrt[:Cpe].abs().sum(0) * vmax
        if self.code_table is not None:
            a_max = a_max + self.code_table.abs().max(0).values
        elif self.wrt.shape[0] > Cpe:
            return False
        else:
            a_max = a_max + self.views_b16.abs()
        hidden = ec[:128].abs() + a_max
        logit = ((self.rgb_w.abs() * hidden[None, :]).sum(-1) + self.rgb_b.abs()).max()
        ok = (ec[128] / float(cfg["density_scale"]) <= 0) & (logit < 1e30)
        return bool(ok.item())

    @staticmethod
    def _equalized(p):
        """Range guard of the fp16 hi/lo-split kernels (k_pe_mlp16): an EXACT re-parametrisation of the MLP by powers of two.

        A ReLU layer commutes with positive scaling, relu(r (W x + b)) = r relu(W x + b), so with per-layer factors r_l = 2^k:
            W'_l = r_l W_l / r_{l-1},  b'_l = r_l b_l   =>   y'_l = r_l y_l,
        and the heads divide the factor out again (alpha_linear, feature_linear by r_7; rgb_linear by the view layer's r_v).
        Powers of two commute with every fp32 rounding and with the hi/lo split, so as long as nothing leaves fp16's range the
        kernel's result is bit-identical; what changes is WHERE the numbers sit: r_l is chosen so that a layer's gain on its
        scaled input (RMS row norm x 0.7 for the ReLU) is ~1 and its largest bias stays below 2^10, which keeps activations near
        the size of the positional encoding whatever the checkpoint's scale is (weights of 1e-3 or 1e3 times the usual size would
        otherwise run the activations out of fp16's 6e-5 .. 65504).  Everything stays on the device (no host sync); the
        exact-fp32 kernels (mlp_mode = 'fp32') take the parameters as they are."""
        q = dict(p)
        one = torch.ones((), device=p["alpha_linear.weight"].device)
        tiny = torch.finfo(torch.float32).tiny

        def pow2(x):   # power of two nearest to x (> 0) in log scale, clamped to 2^+-100 (fp32 itself ends at 2^+-126)
            return torch.exp2(torch.clamp(torch.round(torch.log2(torch.clamp(x, min=tiny))), -100, 100))

        def factor(w_eff, b):
            gain = torch.sqrt((w_eff.double() ** 2).sum(1).mean()).float() * 0.7
            r = pow2(1.0 / torch.clamp(gain, min=tiny))
            cap = torch.exp2(torch.clamp(torch.floor(torch.log2(1024.0 / torch.clamp(b.abs().max(), min=tiny))), -100, 100))
            return torch.minimum(r, cap)

        r_prev = one
        n_in = p["pts_linears.0.weight"].shape[1]
        for l in range(8):
            w, b = p[f"pts_linears.{l}.weight"], p[f"pts_linears.{l}.bias"]
            if l == 0:
                w_eff = w
            elif w.shape[1] > p["pts_linears.1.weight"].shape[1]:      # the skip layer: [input | h]
                w_eff = torch.cat([w[:, :n_in], w[:, n_in:] / r_prev], 1)
            else:
                w_eff = w / r_prev
            r = factor(w_eff, b)
            q[f"pts_linears.{l}.weight"], q[f"pts_linears.{l}.bias"] = (w_eff * r).contiguous(), (b * r).contiguous()
            r_prev = r
        q["alpha_linear.weight"] = (p["alpha_linear.weight"] / r_prev).contiguous()
        q["feature_linear.weight"] = (p["feature_linear.weight"] / r_prev).contiguous()
        wv, bv = p["views_linears.0.weight"], p["views_linears.0.bias"]
        rv = factor(wv, bv)
        q["views_linears.0.weight"], q["views_linears.0.bias"] = (wv * rv).contiguous(), (bv * rv).contiguous()
        q["rgb_linear.weight"] = (p["rgb_linear.weight"] / rv).contiguous()
        q["_view_scale"] = rv
        return q

    def _mlp(self, h, S, cview, raw, lst=None, cnt=None, n=None, aux=False):
        if self.mlp_mode == "f16split":
            return ops.pe_mlp16(h, S, self.packed16, self.pts_b, self.alpha_w, self.alpha_b, cview,
                                self.rgb_w, self.rgb_b, raw, lst, cnt, n, aux, form=self.mlp_form)
        return ops.pe_mlp(h, S, self.packed, self.pts_b, self.alpha_w, self.alpha_b, self.feature_b, cview,
                          self.rgb_w, self.rgb_b, raw, lst, cnt, n, aux)

    # ------------------------------------------------------------------ network forward
    def volumes(self, bones):
        self.refresh()
        return ops.pose_volumes(bones, self.gw, self.cfg["multires_graph"])

    def view_constants(self, rays_d, skts, cam_idx, ray_list=None, ray_count=None):
        self.refresh()
        cfg = self.cfg
        ray_mode = {"world": 0, "root_local": 1}[cfg["ray_tr_type"]]
        normalise = 1 if cfg["view_type"] == "relray" else 0
        return ops.view_consts(rays_d, skts, ray_mode, normalise, cfg["multires_views"], self.framecodes,
                               self.mean_code, cam_idx, self.wrt,
                               self.views_b16 if self.mlp_mode == "f16split" else self.views_b, self.rgb_w,
                               self.rgb_b, self.empty_consts, {16: 1, 32: 2}[self.mlp_form] if self.mlp_mode == "f16split" else 0,
                               self.code_table, ray_list, ray_count)

    def forward_samples(self, rays_o, rays_d, skts, bones, cam_idx=None, z=None, pts=None, dense=False,
                        want_confd=False, volumes=None, view=None, fill=True, ray_mask=None, count=None):
        """DANBO.forward on R x S samples -> raw [R,S,4] (+ dict of extras).

        dense=False: only samples inside >= 1 bone volume go through K1b/K2/K3; all others take
                     the per-ray empty-space raw (identical values).
        dense=True : every sample goes through every kernel (the reference's executed work).
        ray_mask: ops.ray_bone_mask() of these rays over an interval that holds every depth of z (render: [near, far]).
        count: zeroed [1] int32 for the row count (render() fills both passes' counters at once)."""
        self.refresh()
        geo = ops.Geometry(rays_o, rays_d, skts, self.align, self.axis_scale, z=z, pts=pts, ray_mask=ray_mask)
        vols = self.volumes(bones) if volumes is None else volumes
        cview, raw_empty = self.view_constants(geo.rays_d, geo.skts, cam_idx) if view is None else view
        S = geo.S
        bits, lst, cnt = ops.bone_cull(geo, compact=not dense, cnt=count)
        if lst is not None and self.mlp_mode == "f16split" and self.group_rows:
            ops.group_rows(bits, lst, cnt)
        if self.mlp_mode == "f16split":
            h, confd = ops.gather_assign_blend16(geo, vols, bits, self.aw, self.assign16, lst, cnt, geo.M, want_confd)
        else:
            h, confd = ops.gather_assign_blend(geo, vols, bits, self.aw, lst, cnt, geo.M, want_confd)
        # fill=False: rows outside every volume stay unwritten; the consumer reads raw_empty for them (valid_bits == 0)
        raw = ops.fill_raw(raw_empty, S) if fill or dense else torch.empty(geo.R, S, 4, device=raw_empty.device)
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self._mlp(h, S, cview, raw, lst, cnt, geo.M)
        if self.profile is not None:
            e1.record()
            self.profile.setdefault("k_pe_mlp", []).append((e0, e1, cnt if cnt is not None else geo.M))
        extras = dict(valid_bits=bits, list=lst, count=cnt, confd_rows=confd, h_rows=h, volumes=vols)
        return raw, extras

    def density(self, pts, skts, bones, netchunk=1024 * 64):
        """NeRF.forward_pts (reference nerf.py:150-154): raw density of arbitrary points [M,1,3]."""
        self.refresh()
        M = pts.shape[0]
        out = torch.empty(M, 1, device=pts.device, dtype=torch.float32)
        vols = self.volumes(bones)
        # netchunk points at a time, as the reference (raycasters.py:421-453): a 256^3 grid in one piece would need > 10 GB of
        # per-point buffers.  The colour head's per-ray inputs are irrelevant for the density: zero rows, allocated per chunk.
        for a in range(0, M, netchunk):
            p = pts[a:a + netchunk].contiguous()
            n = p.shape[0]
            dummy = torch.zeros(n, 3, device=pts.device)
            raw_empty = torch.zeros(n, 4, device=pts.device)
            raw_empty[:, 3] = self.empty_consts[128]
            cview = torch.zeros(n, ops.VIEW_W, device=pts.device)
            raw, _ = self.forward_samples(dummy, dummy, skts, bones, pts=p, volumes=vols, view=(cview, raw_empty))
            out[a:a + n] = raw[..., 3:4].reshape(n, 1)
        return out

    # ------------------------------------------------------------------ the same chain behind ONE C call
    def render_frame_c(self, rays_o, rays_d, skts, bones, cyls, cam_idx=None, N_samples=None, N_importance=None, chunk=4096):
        """`render()` through `danbo_render_frame` (include/danbo_hip.h): the library enqueues the whole chain itself, out of
        one workspace buffer -- the entry point a C host binds.  Same kernels, same order: bit-identical outputs."""
        import ctypes
        from . import _hip
        assert self.mlp_mode == "f16split" and self.mlp_form == 32
        self.refresh()
        cfg = self.cfg
        S, Sf = N_samples or cfg["N_samples"], N_importance or cfg["N_importance"]
        f32 = lambda t: None if t is None else t.float().contiguous()  # noqa: E731
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        rays_o, rays_d, skts, bones, cyls = (f32(t) for t in (rays_o, rays_d, skts, bones, cyls))
        cam = None if (cam_idx is None or not cfg["use_framecode"]) else cam_idx.reshape(-1).to(torch.int64).contiguous()
        R, G, dev = rays_o.shape[0], skts.shape[0], rays_o.device
        gw, aw = self.gw, self.aw
        m = _hip.DanboModel()
        for k in ("w0", "adjw0", "b0", "w1", "adjw1", "b1", "w2", "b2", "w3", "b3"):
            setattr(m, "g_" + k, gw[k].data_ptr())
        m.L_graph, m.graph_width = cfg["multires_graph"], gw["w1"].shape[-1]
        m.align, m.axis_scale, m.assign16 = self.align.data_ptr(), self.axis_scale.data_ptr(), self.assign16.data_ptr()
        m.a_b0, m.a_b1, m.a_w2, m.a_b2 = (aw[k].data_ptr() for k in ("b0", "b1", "w2", "b2"))
        m.mlp16 = self.packed16.data_ptr()
        for i in range(8):
            m.pts_b[i] = self.pts_b[i].data_ptr()
        m.alpha_w, m.alpha_b, m.rgb_w, m.rgb_b = (t.data_ptr() for t in (self.alpha_w, self.alpha_b, self.rgb_w, self.rgb_b))
        m.views_w_ray_t, m.views_b_eff, m.empty_consts = self.wrt.data_ptr(), self.views_b16.data_ptr(), self.empty_consts.data_ptr()
        if self.framecodes is not None:
            m.framecodes, m.mean_code, m.code_table = (t.data_ptr() for t in (self.framecodes, self.mean_code, self.code_table))
            m.n_codes, m.code_size = self.framecodes.shape
        m.L_view = cfg["multires_views"]
        m.ray_mode, m.normalise = {"world": 0, "root_local": 1}[cfg["ray_tr_type"]], 1 if cfg["view_type"] == "relray" else 0
        m.density_scale, m.use_volume_near_far = float(cfg["density_scale"]), int(bool(cfg["use_volume_near_far"]))
        m.flat_rays_ok = int(self.flat_rays_ok and self.skip_flat_rays)
        r = _hip.DanboRays(rays_o=rays_o.data_ptr(), rays_d=rays_d.data_ptr(), skts=skts.data_ptr(), bones=bones.data_ptr(),
                           cyls=cyls.data_ptr(), cam_idx=None if cam is None else cam.data_ptr(), near_in=None, far_in=None,
                           R=R, G=G, chunk=int(chunk))
        shapes = dict(rgb_map=(R, 3), disp_map=(R,), acc_map=(R,), alpha=(R, S + Sf), weights=(R, S + Sf), rgb0=(R, 3),
                      disp0=(R,), acc0=(R,), alpha0=(R, S))
        out = {k: torch.empty(v, device=dev, dtype=torch.float32) for k, v in shapes.items()}
        o = _hip.DanboFrameOut(**{k: v.data_ptr() for k, v in out.items()})
        nbytes = _hip.lib().danbo_render_frame_workspace(R, G, S, Sf, int(chunk), m.graph_width)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        _hip.check(_hip.lib().danbo_render_frame(ctypes.byref(m), ctypes.byref(r), S, Sf, ctypes.byref(o), ptr(ws), nbytes,
                                                 ops._stream()), "danbo_render_frame")
        out["T_i"] = out.pop("weights")
        return out

    # ------------------------------------------------------------------ RayCaster.render_rays (eval)
    def near_far(self, rays_o, rays_d, cyls, skts, near0=0.0, far0=1.0, chunk=4096):
        self.refresh()
        near, far = ops.near_far_cylinder(rays_o, rays_d, cyls, near0, far0, chunk)
        if self.cfg["use_volume_near_far"]:
            ops.near_far_boxes(rays_o, rays_d, skts, self.align, self.axis_scale, near, far)
        return near, far

    def render(self, rays_o, rays_d, skts, bones, cyls, cam_idx=None, N_samples=None, N_importance=None,
               chunk=4096, near_far=None, dense=False, keep=False):
        cfg = self.cfg
        S = N_samples or cfg["N_samples"]
        Sf = N_importance or cfg["N_importance"]
        B = cfg["density_scale"]
        self.refresh()
        fused = S <= 64 and Sf <= 64
        lazy = not dense and not keep      # skip the raw pre-fill: consumers read raw_empty where bits == 0
        # lazy: nobody outside this function sees z_fine, the sorted order or the per-ray view constants.  If the weights allow it
        # (_flat_rays_ok), the rays that cannot meet a volume anywhere in [near, far] -- flagged with the ray mask -- get their
        # constants from ops.flat_rays and nothing else: no view constants, no resampling, no composite (danbo_hip.h).  The
        # coarse depths are this function's own: near (1 - t) + far t lies in [near, far] up to a few ulps, far inside the slack
        # the flags allow, so they need no confirmation by the cull.
        # Longer rays (S > 64, up to 256 with Sf <= 64: the unfused composites) take the same constants: the coarse composite,
        # the resampling and the final composite walk the list there too.
        flat_mode = lazy and (fused or (S <= 256 and Sf <= 64)) and self.skip_flat_rays and self.flat_rays_ok
        near, far = self.near_far(rays_o, rays_d, cyls, skts, 0.0, 1.0, chunk) if near_far is None else near_far
        z = ops.coarse_samples(near, far, S)
        # candidate bones of every ray over [near, far] (coarse and importance depths both lie inside): the two culls skip the
        # rays, and whole workgroups, that miss every volume -- most of a frame
        ray_mask = None if dense else ops.ray_bone_mask(rays_o, rays_d, skts, self.align, self.axis_scale, near, far,
                                                        want_flat=flat_mode)
        counts = torch.zeros(3, device=rays_o.device, dtype=torch.int32)       # rows of the two passes, listed rays: one fill
        flat = None
        vols = self.volumes(bones)
        if flat_mode:
            flat = ops.flat_rays(ray_mask[1], ray_mask[3], S, Sf, want_weights=not fused, cnt=counts[2:3])
            view = self.view_constants(rays_d, skts, cam_idx, flat["ray_list"], flat["ray_count"])
        else:
            view = self.view_constants(rays_d, skts, cam_idx)
        if ray_mask is not None:
            ray_mask = ray_mask[:3]
        raw, ex = self.forward_samples(rays_o, rays_d, skts, bones, cam_idx, z=z, dense=dense, volumes=vols, view=view,
                                       fill=not lazy, ray_mask=ray_mask, count=counts[0:1])
        if fused:
            out0, z_all, z_fine, order = ops.composite_importance(
                raw, z, rays_d, Sf, B, bits=ex["valid_bits"] if lazy else None, raw_empty=view[1] if lazy else None,
                want_weights=keep, flat=flat)
        else:
            out0 = ops.composite(raw, z, rays_d, B, bits=ex["valid_bits"] if lazy else None, raw_empty=view[1] if lazy else None,
                                 flat=flat)
            z_all, z_fine, order = ops.importance_samples(z, out0["weights"], Sf, flat=flat)
        raw_f, ex_f = self.forward_samples(rays_o, rays_d, skts, bones, cam_idx, z=z_fine, dense=dense,
                                           volumes=vols, view=view, fill=not lazy,
                                           ray_mask=ray_mask, count=counts[1:2])
        out = ops.composite_merged(raw, raw_f, order, z_all, rays_d, B, bits_a=ex["valid_bits"] if lazy else None,
                                   bits_b=ex_f["valid_bits"] if lazy else None, raw_empty=view[1] if lazy else None,
                                   want_raw=keep, flat=flat)
        raw_all = out.get("raw_sorted")
        ret = dict(rgb_map=out["rgb_map"], disp_map=out["disp_map"], acc_map=out["acc_map"], alpha=out["alpha"],
                   T_i=out["weights"], rgb0=out0["rgb_map"], disp0=out0["disp_map"], acc0=out0["acc_map"],
                   alpha0=out0["alpha"])
        if keep:
            ret.update(near=near, far=far, z_coarse=z, raw_coarse=raw, weights_coarse=out0["weights"], z_fine=z_fine,
                       z_sorted=z_all, sorted_idxs=order, raw_fine=raw_f, raw_sorted=raw_all,
                       count_coarse=ex["count"], count_fine=ex_f["count"], valid_bits=ex["valid_bits"])
        return ret
