"""DANBO model (reference: core/networks/danbo.py) on the gfx950 kernels.

Same constructor, sub-module names (`graph_net`, `prob_linears`, `pts_linears`, ...) and
method surface as the reference; `forward(inputs)` takes the reference's `nerf_inputs`
dictionary (core/raycasters.py:399-413) and returns `(raw [R,S,4], encoded)`.
"""
import torch

from .nerf import NeRF
from .gnn_backbone import get_gnn_backbone, get_volume_gnn_backbone
from ..render_engine import DanboEngine


class DANBO(NeRF):
    def __init__(self, *args, node_W=128, input_ch_graph=44, input_ch_voxel=44, voxel_feat=4, voxel_res=4, gcn_D=4,
                 gcn_fc_D=0, gcn_sep_bias=False, graph_pe_fn=None, voxel_pe_fn=None, backbone='PNBGNN', agg_W=16,
                 agg_D=3, rest_pose=None, mask_root=False, align_corners=False, agg_backbone=None,
                 adj_self_one=False, gnn_concat=False, aggregate_dim=None, detach_agg_grad=False, init_adj_w=0.05,
                 attenuate_feat=False, attenuate_invalid=False, opt_scale=False, base_scale=0.5, gnn_n_basis=32,
                 no_adj=False, skel_profile=None, mask_vol_prob=False, use_posecode=False, agg_type='sigmoid',
                 **kwargs):
        unsupported = dict(gnn_concat=gnn_concat, use_posecode=use_posecode, detach_agg_grad=detach_agg_grad,
                           adj_self_one=adj_self_one, no_adj=no_adj)
        bad = [k for k, v in unsupported.items() if v]
        if bad or agg_type != 'sigmoid' or not mask_vol_prob or not mask_root or agg_backbone != 'vox_MIXGNN':
            raise NotImplementedError(f"unsupported DANBO variant {bad or (agg_type, agg_backbone)}: the HIP path "
                                      "implements the shipped configs (FGNNcat + vox_MIXGNN + sigmoid + mask_root)")
        self.node_W, self.agg_W, self.agg_D = node_W, agg_W, agg_D
        self.skel_type, self.rest_pose = kwargs['skel_type'], rest_pose
        self.input_ch_graph, self.input_ch_voxel = input_ch_graph, input_ch_voxel
        self.voxel_feat, self.voxel_res, self.backbone = voxel_feat, voxel_res, backbone
        self.gcn_D, self.gcn_fc_D, self.gcn_sep_bias = gcn_D, gcn_fc_D, gcn_sep_bias
        self.agg_backbone, self.mask_root, self.align_corners = agg_backbone, mask_root, align_corners
        self.opt_scale, self.base_scale, self.skel_profile = opt_scale, base_scale, skel_profile
        self.attenuate_feat, self.attenuate_invalid = attenuate_feat, attenuate_invalid
        self.mask_vol_prob, self.agg_type, self.init_adj_w = mask_vol_prob, agg_type, init_adj_w
        n_joints = len(self.skel_type.joint_trees)
        self.volume_shape = [n_joints, voxel_feat] + 3 * [voxel_res]
        # per-bone view encodings (72 -> 3 after blending): the reference's `% 24` rule (danbo.py:82-83)
        if kwargs['input_ch_views'] % 24 == 0:
            kwargs['input_ch_views'] = kwargs['input_ch_views'] // n_joints
        super().__init__(*args, **kwargs)
        self.graph_pe_fn, self.voxel_pe_fn = graph_pe_fn, voxel_pe_fn
        self.graph_net = get_volume_gnn_backbone(
            input_ch_graph, skel_type=self.skel_type, gcn_D=gcn_D, node_W=node_W, gcn_fc_D=gcn_fc_D,
            gcn_sep_bias=gcn_sep_bias, voxel_res=voxel_res, voxel_feat=voxel_feat, skip_gcn=False,
            backbone=backbone, mask_root=mask_root, opt_scale=opt_scale, base_scale=base_scale,
            skel_profile=skel_profile, aggregate_dim=aggregate_dim, attenuate_feat=attenuate_feat,
            attenuate_invalid=attenuate_invalid, align_corners=align_corners, init_adj_w=init_adj_w)
        self.prob_linears = get_gnn_backbone(
            voxel_feat * 3, skel_type=self.skel_type, gcn_D=agg_D, node_W=agg_W, gcn_sep_bias=gcn_sep_bias,
            output_ch=1, skip_gcn=False, backbone='_'.join(agg_backbone.split('_')[1:]), init_adj_w=init_adj_w)
        if (self.W, self.D, self.view_W, tuple(self.skips), voxel_feat, voxel_res, agg_W) != (256, 8, 128, (4,), 5, 16, 32):
            raise NotImplementedError("k_pe_mlp / k_assign are specialised for W=256, D=8, view_W=128, "
                                      "voxel 5x16, agg_W=32 (all shipped DANBO configs)")

    # ---- input width of the density trunk: PE of the blended 15-d voxel feature ----
    @property
    def pts_input_ch(self):
        return self.input_ch_voxel

    def init_density_net(self):
        self.pts_linears = self._trunk(self.pts_input_ch)
        self.alpha_linear = torch.nn.Linear(self.W, 1)

    # ---- engine plumbing ----
    def engine_config(self):
        view_type = 'relray' if self.pts_embedder.view_input_fn.encoder_name == 'VecNorm' else 'identity'
        ray_tr = {'RLEncoder': 'root_local', 'world': 'world'}[self.pts_embedder.ray_tr_fn.encoder_name]
        return dict(multires_graph=self.graph_pe_fn.num_freqs, multires_voxel=self.voxel_pe_fn.num_freqs,
                    multires_views=self.dirs_pe_fn.num_freqs, use_framecode=self.use_framecode,
                    view_type=view_type, ray_tr_type=ray_tr, density_scale=self.density_scale,
                    use_volume_near_far=False, N_samples=None, N_importance=None)

    def engine(self, align):
        """The kernel orchestrator bound to this module's parameters (rebuilt if they move)."""
        params = self._engine_params()
        key = (next(self.parameters()).device, align.data_ptr())
        if self._engine is None or self._engine_key != key:
            self._engine = DanboEngine(self.engine_config(), params, align.to(key[0]))
            self._engine_key = key
        else:
            self._engine.p = params
        return self._engine

    @staticmethod
    def _unique(x, n_uniques):
        return x[::max(x.shape[0] // n_uniques, 1)].contiguous()

    def forward(self, inputs, netchunk=1024 * 64):
        """inputs: pts [R,S,3], skts [R|1,24,4,4], bones [R|1,24,3], align_transforms [..,24,4,4],
        N_uniques, rays_d [R,1,3], cam_idxs [R] | None  ->  raw [R,S,4], encoded"""
        if self.training:
            from .. import train_path
            return train_path.forward_train(self, inputs)
        pts = inputs['pts']
        R = pts.shape[0]
        G = int(inputs.get('N_uniques', 1))
        skts, bones = inputs['skts'], inputs['bones']
        if skts.shape[0] != R:
            G = skts.shape[0]
        skts_g = skts if skts.shape[0] == G else self._unique(skts, G)
        bones_g = bones if bones.shape[0] == G else self._unique(bones, G)
        align = inputs['align_transforms'].reshape(-1, 24, 4, 4)[0]
        eng = self.engine(align)
        rays_d = inputs['rays_d'].reshape(R, 3)
        rays_o = inputs['rays_o'].reshape(R, 3) if inputs.get('rays_o') is not None else rays_d
        raw, ex = eng.forward_samples(rays_o, rays_d, skts_g, bones_g, inputs.get('cam_idxs'), pts=pts)
        return raw, self.collect_encoded(dict(ex, eng=eng, pts=pts, rays_d=rays_d, skts=skts_g), None)

    def collect_encoded(self, encoded_pts, encoded_views):
        """`encoded` of the reference's DANBO.forward (core/networks/danbo.py:341-346): confd [R,S,24] = the assignment logits of
        EVERY sample and bone, part_invalid [R,S,24] = 1 outside the bone's volume -- in eval mode too.  The reference's caster
        reads them in training only (raycasters.py:710-716), so they are formed on first access: the render path (which culls
        and never evaluates the assignment net outside the volumes) pays nothing for them."""
        from .. import hip_ops as ops
        from ..utils.lazy import LazyDict
        ex = encoded_pts

        def fill():
            eng, pts = ex['eng'], ex['pts']
            R, S = pts.shape[:2]
            geo = ops.Geometry(ex['rays_d'], ex['rays_d'], ex['skts'], eng.align, eng.axis_scale, pts=pts.contiguous().float())
            bits = ex['valid_bits']
            # dense: all R x S rows, all 24 bones (the culled pass evaluates only bones some sample of the wavefront is inside)
            if eng.mlp_mode == "f16split":
                _, confd = ops.gather_assign_blend16(geo, ex['volumes'], bits, eng.aw, eng.assign16, want_confd=True)
            else:
                _, confd = ops.gather_assign_blend(geo, ex['volumes'], bits, eng.aw, want_confd=True)
            shifts = torch.arange(24, device=bits.device, dtype=torch.int32)
            valid = ((bits.reshape(R, S, 1) >> shifts) & 1).float()
            return dict(confd=confd.reshape(R, S, 24), part_invalid=1.0 - valid)
        return LazyDict(fill)

    # ---- helpers the trainer calls on the module (reference danbo.py:382-415) ----
    def get_adjw(self):
        return self.graph_net.get_adjw() + self.prob_linears.get_adjw()

    def sigmoid(self, logit, invalid, mask_invalid=True, clamp=True, eps=1e-7, sigmoid_eps=0.001):
        p = torch.sigmoid(logit) * (1 + 2 * sigmoid_eps) - sigmoid_eps
        if mask_invalid:
            p = p * (1 - invalid.flatten(end_dim=-2))
        return p
