"""NeRF / A-NeRF module (reference: core/networks/nerf.py).

Parameters keep the reference's names and shapes (`pts_linears.N`, `alpha_linear`,
`feature_linear`, `views_linears.0`, `rgb_linear`, `framecodes.codes`), so reference
checkpoints load with `load_state_dict`.  Arithmetic: libdanbo_hip (include/danbo_hip.h).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .embedding import Optcodes
from .. import hip_ops as ops


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, input_ch=3, input_ch_bones=0, input_ch_views=3, output_ch=4, skips=[4],
                 use_viewdirs=False, use_framecode=False, framecode_ch=16, n_framecodes=0, pts_embedder=None,
                 pe_fn=None, bones_pe_fn=None, dirs_pe_fn=None, skel_type=None, view_W=None, density_scale=1.0):
        super().__init__()
        self.D, self.W = D, W
        self.view_W = W // 2 if view_W is None else view_W
        self.input_ch, self.input_ch_bones, self.input_ch_views = input_ch, input_ch_bones, input_ch_views
        self.skips, self.use_viewdirs = list(skips), use_viewdirs
        self.use_framecode, self.framecode_ch, self.n_framecodes = use_framecode, framecode_ch, n_framecodes
        self.cam_ch = 1 if use_framecode else 0
        self.N_joints, self.output_ch = 24, output_ch
        self.skel_type, self.density_scale = skel_type, density_scale
        self.pts_embedder, self.pe_fn, self.bones_pe_fn, self.dirs_pe_fn = pts_embedder, pe_fn, bones_pe_fn, dirs_pe_fn
        if not use_viewdirs:
            raise NotImplementedError("use_viewdirs=False is not used by any shipped config")
        self.init_density_net()
        self.init_radiance_net()
        self._engine = None

    # ---- layer sizes (reference :61-105) ----
    @property
    def dnet_input(self):
        return self.input_ch + self.input_ch_bones

    @property
    def vnet_input(self):
        return self.input_ch_views + (self.framecode_ch if self.use_framecode else 0) + self.view_W * 2

    def _trunk(self, in_ch):
        layers = [nn.Linear(in_ch, self.W)]
        for i in range(self.D - 1):
            layers.append(nn.Linear(self.W + in_ch if i in self.skips else self.W, self.W))
        return nn.ModuleList(layers)

    def init_density_net(self):
        self.pts_linears = self._trunk(self.dnet_input)
        self.alpha_linear = nn.Linear(self.W, 1)

    def init_radiance_net(self):
        self.views_linears = nn.ModuleList([nn.Linear(self.vnet_input, self.view_W)])
        self.feature_linear = nn.Linear(self.W, self.view_W * 2)
        self.rgb_linear = nn.Linear(self.view_W, 3)
        if self.use_framecode:
            self.framecodes = Optcodes(self.n_framecodes, self.framecode_ch)

    # ---- compositing (reference :281-347) ----
    def raw2outputs(self, raw, z_vals, rays_d, raw_noise_std=0, pytest=False, B=0.01, rgb_act=torch.sigmoid,
                    act_fn=F.relu, rgb_eps=0.001, alpha_w=None, render_confd=False, render_entropy=False, **kwargs):
        if act_fn is not F.relu and getattr(act_fn, "__name__", "") != "relu":
            raise NotImplementedError("only density_type=relu is implemented in danbo_composite_fwd")
        if alpha_w is not None or rgb_act is not torch.sigmoid:
            raise NotImplementedError("alpha_w / a colour activation other than sigmoid are out of scope")
        noise = None
        if raw_noise_std > 0.:
            noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std * B
        if render_confd or render_entropy:
            # the reference's visualisations (nerf.py:306-311): the per-sample colour is replaced by the colour of the most
            # confident bone / by a blue -> red ramp of the assignment entropy, read from raw[..., 4:]; everything else of
            # raw2outputs is unchanged.  The composite kernel applies sigmoid(.) * (1 + 2 eps) - eps to its colour channels, so
            # it is handed the logit of that map's inverse (a visualisation: not on the hot path, plain torch element-wise ops).
            assert raw.shape[-1] > 4, 'Needs to have confidence/prob logit when render_confd=True'
            from .misc import get_confidence_rgb, get_entropy_rgb
            rgb = (get_confidence_rgb if render_confd else get_entropy_rgb)(raw[..., 4:], kwargs.get('encoded'))
            y = ((rgb + rgb_eps) / (1. + 2. * rgb_eps)).clamp(1e-6, 1. - 1e-6)
            raw = torch.cat([torch.log(y) - torch.log1p(-y), raw[..., 3:4]], -1).contiguous()
        else:
            raw = raw[..., :4]
        if raw.requires_grad:
            from .. import train_path
            return train_path.composite(raw, z_vals, rays_d, B, noise)
        return ops.composite(raw, z_vals, rays_d.reshape(-1, 3), B, noise)

    def update_embed_fns(self, global_step, args):
        for fn in (self.pe_fn, self.dirs_pe_fn, self.bones_pe_fn):
            if fn is not None and hasattr(fn, 'update_threshold'):
                fn.update_threshold(global_step, args.cutoff_step, args.cutoff_rate,
                                    getattr(args, 'freq_schedule_step', 0), args.multires - 1)

    def collect_encoded(self, encoded_pts, encoded_views):
        return {}

    # ---- A-NeRF (cutoff PE) engine plumbing ----
    def engine_config(self):
        emb = self.pts_embedder
        names = (emb.pts_tr_fn.encoder_name, emb.kp_input_fn.encoder_name, emb.bone_input_fn.encoder_name,
                 emb.view_input_fn.encoder_name, getattr(emb.ray_tr_fn, 'encoder_name', 'local'))
        pe, dpe, bpe = self.pe_fn, self.dirs_pe_fn, self.bones_pe_fn
        ok = (names == ('W2LEncoder', 'RelDist', 'VecNorm', 'VecNorm', 'local')
              and type(pe).__name__ == 'CutoffEmbedder' and type(dpe).__name__ == 'CutoffEmbedder'
              and pe.cut_to_cutoff and pe.shift_inputs and pe.cutoff_inputs and not pe.dist_inputs
              and dpe.dist_inputs and dpe.cutoff_inputs and getattr(bpe, 'num_freqs', 0) == 0
              and type(bpe).__name__ != 'CutoffEmbedder' and tuple(self.skips) == (4,) and self.view_W <= 256)
        if not ok:
            raise NotImplementedError(f"NeRF variant {names}: the HIP path implements the shipped A-NeRF configuration "
                                      "(reldist + reldir + relray, local rays, use_cutoff, cutoff_viewdir, cutoff_inputs, "
                                      "cut_to_dist, cutoff_shift)")
        return dict(nerf_type='nerf', W=self.W, D=self.D, skips=tuple(self.skips), view_W=self.view_W,
                    multires=pe.num_freqs, multires_views=dpe.num_freqs, use_framecode=self.use_framecode,
                    density_scale=self.density_scale, use_volume_near_far=False, N_samples=None, N_importance=None)

    # ---- parameters as the engines see them: detached views, rebuilt only when storage moved ----
    def _engine_params(self):
        """{name: detached tensor} of every parameter and buffer.  Walking the module tree costs ~0.3 ms, which is most of the
        host time of a 512-ray validation chunk, so the walk is cached; in-place updates (optimizer steps, load_state_dict)
        keep the storage and are seen through the shared version counters, moves (`.to()`) change the pointers."""
        cache = self.__dict__.get('_engine_params_cache')
        if cache is None:
            named = list(self.named_parameters()) + list(self.named_buffers())
            cache = self.__dict__['_engine_params_cache'] = [named, None, None]
        ptrs = tuple(v.data_ptr() for _, v in cache[0])
        if ptrs != cache[1]:
            cache[1], cache[2] = ptrs, {k: v.detach() for k, v in cache[0]}
        return cache[2]

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .float(): buffers are replaced by new tensors
        self.__dict__.pop('_engine_params_cache', None)
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__.pop('_engine_params_cache', None)
        return super().load_state_dict(*args, **kwargs)

    def engine(self, align):
        from ..anerf_engine import AnerfEngine
        params = self._engine_params()
        key = (next(self.parameters()).device, align.data_ptr())
        if self._engine is None or self._engine_key != key:
            self._engine = AnerfEngine(self.engine_config(), params, align.to(key[0]))
            self._engine_key = key
        else:
            self._engine.p = params
        return self._engine

    def forward(self, inputs, netchunk=1024 * 64):
        """inputs: the reference's nerf_inputs dict (core/raycasters.py:399-413) -> raw [R,S,4], encoded"""
        if self.training:
            from .. import train_path
            self.engine_config()   # raises for unsupported encoder combinations
            return train_path.forward_train_anerf(self, inputs)
        pts = inputs['pts']
        R = pts.shape[0]
        G = int(inputs.get('N_uniques', 1))
        skts = inputs['skts']
        if skts.shape[0] != R:
            G = skts.shape[0]
        skts_g = skts if skts.shape[0] == G else skts[::max(skts.shape[0] // G, 1)].contiguous()
        eng = self.engine(inputs['align_transforms'].reshape(-1, 24, 4, 4)[0])
        raw = eng.forward_samples(None, inputs['rays_d'].reshape(R, 3), skts_g, inputs.get('cam_idxs'), pts=pts)
        return raw, {}

    def forward_pts(self, inputs, **kwargs):
        """density of arbitrary points (reference nerf.py:150-154)"""
        eng = self.engine(inputs['align_transforms'].reshape(-1, 24, 4, 4)[0])
        return eng.density(inputs['pts'], inputs['skts'][:1])
