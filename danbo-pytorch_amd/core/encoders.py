"""Encoder selection (reference: core/encoders.py:10-282).

The reference builds a tree of small nn.Modules (WorldToLocalEncoder, RelDistEncoder,
VecNormEncoder, ...) and chains them in Python.  Here the same flags pick a *fused kernel
variant*; the classes below only carry the names / dimensions the model constructors and the
trainer's logging read.  Supported combinations = the shipped configs:
    pts_tr_type=local, kp_dist_type=reldist, bone_type in {Nope, reldir},
    (view_type, ray_tr_type) in {(identity, world), (relray, root_local), (relray, local)},
    graph_input_type=rot6d.
"""
from copy import deepcopy

import torch.nn as nn

from .cutoff_embedder import get_embedder
from .utils.skeleton_utils import SMPLSkeleton


class _Spec(nn.Module):
    """Named, dimensioned description of one encoder; never executed in Python."""

    def __init__(self, name, dims):
        super().__init__()
        self._name, self._dims = name, dims

    @property
    def encoder_name(self):
        return self._name

    @property
    def dims(self):
        return self._dims

    def forward(self, *a, **k):
        raise RuntimeError(f"{self._name}: fused into libdanbo_hip; there is no eager path")


def transform_batch_pts(pts, skt):
    """reference encoders.py:288-303: pts [N_rays, N_samples, 3] into every joint's local frame -> [N_rays, N_samples, NJ, 3]
    (skt [N_rays or fewer, NJ, 4, 4]: broadcast over the rays as the reference's expand does) -- danbo_transform_batch_pts"""
    from . import hip_ops as ops
    return ops.transform_batch(pts, skt, rot_only=False)


def transform_batch_rays(rays_o, rays_d, skt):
    """reference encoders.py:305-318: the rotational part of skt applied to the directions rays_d [N_rays, N_samples, 3]"""
    from . import hip_ops as ops
    return ops.transform_batch(rays_d, skt, rot_only=True)


def _pick(kind, table, key):
    if key not in table:
        raise NotImplementedError(f"{kind}={key} is not used by any shipped config (supported: {sorted(table)})")
    return table[key]


class SamplePointsEmbedder(nn.Module):
    def __init__(self, pts_tr_fn, ray_tr_fn, kp_input_fn=None, bone_input_fn=None, view_input_fn=None,
                 graph_input_fn=None, skel_type=SMPLSkeleton):
        super().__init__()
        self.pts_tr_fn, self.ray_tr_fn = pts_tr_fn, ray_tr_fn
        self.kp_input_fn, self.bone_input_fn = kp_input_fn, bone_input_fn
        self.view_input_fn, self.graph_input_fn = view_input_fn, graph_input_fn
        self.skel_type = skel_type

    def forward(self, *a, **k):
        raise RuntimeError("SamplePointsEmbedder: fused into libdanbo_hip; there is no eager path")


def get_pts_embedder(args, data_attrs):
    skel_type = data_attrs['skel_type']
    J = len(skel_type.joint_names)
    pts_tr = _pick('pts_tr_type', {'local': _Spec('W2LEncoder', J * 3)}, args.pts_tr_type)
    ray_tr = _pick('ray_tr_type', {'local': _Spec('local', J * 3), 'root_local': _Spec('RLEncoder', 3),
                                   'world': _Spec('world', 3)}, args.ray_tr_type)
    kp = _pick('kp_dist_type', {'reldist': _Spec('RelDist', J)}, args.kp_dist_type)
    bone = _pick('bone_type', {'reldir': _Spec('VecNorm', J * 3), 'Nope': _Spec('Empty', 0)}, args.bone_type)
    view = _pick('view_type', {'relray': _Spec('VecNorm', J * 3), 'identity': _Spec('Identity', 3)}, args.view_type)
    if args.view_type == 'relray' and args.ray_tr_type == 'root_local':
        view = _Spec('VecNorm', J * 3)  # dims stay 72; DANBO folds them with its `% 24` rule (danbo.py:82-83)
    embed_dims = dict(input_dims=kp.dims, cutoff_dims=J, bone_dims=bone.dims, view_dims=view.dims)
    graph = None
    if args.nerf_type in ('graph', 'danbo'):
        graph = _pick('graph_input_type', {'rot6d': _Spec('Rot6D', 6)}, args.graph_input_type)
        embed_dims['graph_dims'] = graph.dims
    print(f'PPE: {pts_tr.encoder_name}, KPE: {kp.encoder_name},BPE: {bone.encoder_name}, VPE: {view.encoder_name}')
    emb = SamplePointsEmbedder(pts_tr, ray_tr, kp_input_fn=kp, bone_input_fn=bone, view_input_fn=view,
                               graph_input_fn=graph, skel_type=skel_type)
    return emb, embed_dims


def get_pe_embedder(args, data_attrs, embed_dims):
    """Sizes every positional encoding exactly like the reference (encoders.py:45-163)."""
    skel_type = data_attrs['skel_type']
    J = len(skel_type.joint_trees)
    input_dims, cutoff_dims = embed_dims['input_dims'], embed_dims['cutoff_dims']
    base = dict(cutoff=args.use_cutoff, normalize_cutoff=args.normalize_cutoff,
                cutoff_dist=args.cutoff_mm * args.ext_scale, cutoff_inputs=args.cutoff_inputs,
                opt_cutoff=args.opt_cutoff, cutoff_dim=cutoff_dims, dist_inputs=not (input_dims == cutoff_dims))

    def cut(**over):
        kw = deepcopy(base)
        kw.update(over)
        kw['normalize'] = kw.pop('normalize_cutoff')
        return kw

    common = dict(skel_type=skel_type, freq_schedule=args.freq_schedule, init_alpha=args.init_freq)
    chs, fns = {}, {}
    fns['pe_fn'], chs['input_ch'] = get_embedder(
        args.multires, args.i_embed, input_dims=input_dims,
        cutoff_kwargs=cut(cut_to_cutoff=args.cut_to_dist, shift_inputs=args.cutoff_shift), **common)
    fns['bones_pe_fn'], chs['input_ch_bones'] = get_embedder(
        args.multires_bones, args.i_embed, input_dims=embed_dims['bone_dims'],
        cutoff_kwargs=cut(dist_inputs=True) if args.cutoff_bones else {"cutoff": False}, **common)
    chs['input_ch_views'], fns['dirs_pe_fn'] = 0, None
    if args.use_viewdirs:
        vk = cut(dist_inputs=True) if args.cutoff_viewdir else {"cutoff": False}
        vk["cutoff_dim"] = J
        if not vk["cutoff"]:
            vk = {"cutoff": False}
        fns['dirs_pe_fn'], chs['input_ch_views'] = get_embedder(
            args.multires_views, args.i_embed, input_dims=embed_dims['view_dims'], cutoff_kwargs=vk, **common)
    if args.nerf_type in ('graph', 'danbo'):
        fns['graph_pe_fn'], chs['input_ch_graph'] = get_embedder(
            args.multires_graph, args.i_embed, input_dims=embed_dims['graph_dims'], **common)
        vox = args.voxel_feat * (3 if args.gnn_backbone.endswith('cat') else 1)
        fns['voxel_pe_fn'], chs['input_ch_voxel'] = get_embedder(
            args.multires_voxel, args.i_embed, input_dims=vox, **common)
    chs['output_ch'] = 5 if args.N_importance > 0 else 4
    return fns, chs
