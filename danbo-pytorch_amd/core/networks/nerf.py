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
        if render_confd or render_entropy or alpha_w is not None or rgb_act is not torch.sigmoid:
            raise NotImplementedError("render_confd / render_entropy / alpha_w are out of scope")
        noise = None
        if raw_noise_std > 0.:
            noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std * B
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

    def forward(self, inputs, netchunk=1024 * 64):
        raise NotImplementedError("plain NeRF / A-NeRF forward: the cutoff-PE kernel is not built yet "
                                  "(DESIGN.md, 'not yet built'); DANBO configs are supported")
