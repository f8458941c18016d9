"""Skeleton-graph networks of DANBO as parameter containers
(reference: core/networks/gnn_backbone.py).  Module / parameter / buffer names follow the
reference so checkpoints load unchanged:
    graph_net  = FactorizeGNN : layers.{0,1} DensePNGCN (bias, adj_w, adj, lin.weight),
                                layers.{2,3} ParallelLinear, axis_scale
    prob_linears = MixGNN     : layers.0 DensePNGCN, layers.{1,2} ParallelLinear
The forward arithmetic lives in csrc/k_pose.hip (pose -> volumes), csrc/k_sample.hip
(cull / gather) and csrc/k_assign.hip (assignment); only the shipped structure
(gcn_D=4, gcn_fc_D=1, agg_D=3, FGNNcat, vox_MIXGNN) is supported.
"""
import numpy as np
import torch
import torch.nn as nn

from .misc import ParallelLinear, init_volume_scale, _FUSED


def skeleton_to_graph(skel=None, edges=None):
    """adjacency = I + symmetric parent/child edges; also returns the edge list."""
    if skel is not None:
        edges = [[int(p), i] for i, p in enumerate(skel.joint_trees) if i != p]
    n = int(np.max(edges)) + 1
    adj = np.eye(n, dtype=np.float32)
    for a, b in edges:
        adj[a, b] = adj[b, a] = 1.0
    return adj, edges


class DenseWGCN(nn.Module):
    """Dense graph convolution with a learnable weighted adjacency (masked by `adj`)."""

    def __init__(self, adj, in_channels, out_channels, init_adj_w=0.05, bias=True, sep_bias=False,
                 normalize_adj=False, bound_adj=False, adj_self_one=False, aggregate_dim=None, no_adj=False,
                 skel_type=None, **kwargs):
        super().__init__()
        if normalize_adj or bound_adj or adj_self_one or aggregate_dim is not None or no_adj or sep_bias:
            raise NotImplementedError("adjacency variants are not used by any shipped config")
        self.in_channels, self.out_channels = in_channels, out_channels
        adj = adj.clone().float()
        n = adj.shape[-1]
        eye = torch.arange(n)
        adj[:, eye, eye] = 1
        w = adj * (init_adj_w + (torch.rand_like(adj) - 0.5) * 0.1).clamp(min=0.01, max=1.0)
        w[:, eye, eye] = 1.0
        self.lin = nn.Linear(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.register_buffer('adj', adj)
        self.adj_w = nn.Parameter(w)

    def get_adjw(self):
        return self.adj_w * self.adj

    def forward(self, x):
        raise RuntimeError(_FUSED.format(type(self).__name__))


class DensePNGCN(DenseWGCN):
    """Graph convolution whose feature transform is per-bone (ParallelLinear, no bias)."""

    def __init__(self, adj, in_channel, out_channel, *args, **kwargs):
        super().__init__(adj, in_channel, out_channel, *args, **kwargs)
        self.lin = ParallelLinear(adj.shape[-1], in_channel, out_channel, bias=False)


class BasicGNN(nn.Module):
    def __init__(self, adj_matrix, per_node_input, W=64, D=4, skip_gcn=10, gcn_module=DensePNGCN,
                 gcn_module_kwargs=None, exclude_root=False, mask_root=False, output_ch=None, skel_type=None,
                 **kwargs):
        super().__init__()
        if exclude_root:
            raise NotImplementedError("exclude_root is not used by any shipped config")
        self.adj_matrix, self.skel_type = adj_matrix, skel_type
        self.per_node_input, self.W, self.D = per_node_input, W, D
        self.skip_gcn, self.mask_root = skip_gcn, mask_root
        self.gcn_module_kwargs = dict(gcn_module_kwargs or {})
        self.output_ch = W + 1 if output_ch is None else output_ch
        self.init_network(gcn_module)

    def _adj(self):
        n = self.adj_matrix.shape[-1]
        return torch.tensor(self.adj_matrix).view(1, n, n)

    def get_adjw(self):
        return [m.get_adjw() for m in self.modules() if hasattr(m, 'adj_w')]

    def forward(self, *args, **kwargs):
        raise RuntimeError(_FUSED.format(type(self).__name__))


class MixGNN(BasicGNN):
    """D - N_P - 1 extra graph-conv layers after the first one, then N_P per-bone linears."""

    def __init__(self, *args, N_P=2, **kwargs):
        self.N_P = N_P
        super().__init__(*args, **kwargs)

    def init_network(self, gcn_module):
        n = self.adj_matrix.shape[-1]
        layers = [gcn_module(self._adj(), self.per_node_input, self.W, skel_type=self.skel_type,
                             **self.gcn_module_kwargs)]
        for _ in range(self.D - self.N_P - 1):
            layers.append(gcn_module(self._adj(), self.W, self.W, skel_type=self.skel_type, **self.gcn_module_kwargs))
        for _ in range(self.N_P - 1):
            layers.append(ParallelLinear(n, self.W, self.W))
        layers.append(ParallelLinear(n, self.W, self.output_ch))
        self.layers = nn.ModuleList(layers)
        if len(layers) != 3 or not isinstance(layers[1], ParallelLinear):
            raise NotImplementedError("k_assign.hip implements GCN -> PerBone -> PerBone (agg_D=3)")


class BodyGNN(BasicGNN):
    def __init__(self, *args, voxel_res=4, voxel_feat=4, fc_D=0, align_corners=False, last_module=ParallelLinear,
                 **kwargs):
        self.voxel_res, self.voxel_feat, self.fc_D = voxel_res, voxel_feat, fc_D
        self.align_corners, self.last_module = align_corners, last_module
        if align_corners:
            raise NotImplementedError("align_corners=True is not used by any shipped config")
        super().__init__(*args, **kwargs)

    @property
    def output_size(self):
        return self.voxel_res ** 3 * self.voxel_feat

    def init_network(self, gcn_module):
        n = self.adj_matrix.shape[-1]
        W = self.W
        layers = [gcn_module(self._adj(), self.per_node_input, W, skel_type=self.skel_type, **self.gcn_module_kwargs)]
        for _ in range(self.D - self.fc_D - 2):
            layers.append(gcn_module(self._adj(), W, W, skel_type=self.skel_type, **self.gcn_module_kwargs))
        for _ in range(self.fc_D):
            layers.append(ParallelLinear(n, W, W))
        if self.last_module is not ParallelLinear:
            raise NotImplementedError("only ParallelLinear output layers are supported")
        layers.append(ParallelLinear(n, W, self.output_size))
        self.layers = nn.ModuleList(layers)
        kinds = [isinstance(l, ParallelLinear) for l in layers]
        if kinds != [False, False, True, True]:
            raise NotImplementedError("k_pose.hip implements GCN, GCN, PerBone, PerBone (gcn_D=4, gcn_fc_D=1)")
        self.volume_shape = [n, self.voxel_feat] + 3 * [self.voxel_res]


class FactorizeGNN(BodyGNN):
    """Per-bone volumes factorised into three 1-D feature lines (voxel_feat x voxel_res x 3)."""

    def __init__(self, *args, factorize_type='sum', pred_residual=False, opt_scale=False, base_scale=0.5,
                 attenuate_feat=False, attenuate_invalid=False, skel_profile=None, **kwargs):
        if factorize_type != 'cat' or not attenuate_feat or attenuate_invalid:
            raise NotImplementedError("shipped configs use FGNNcat with attenuate_feat=True")
        self.factorize_type, self.opt_scale = factorize_type, opt_scale
        self.attenuate_feat, self.attenuate_invalid = attenuate_feat, attenuate_invalid
        self.skel_profile, self.base_scale = skel_profile, base_scale
        super().__init__(*args, **kwargs)
        n = len(self.skel_type.joint_names)
        scale = torch.ones(n, 3) * base_scale
        if skel_profile is not None:
            scale = init_volume_scale(base_scale, skel_profile, self.skel_type)
        self.init_scale = scale.clone()
        self.axis_scale = nn.Parameter(scale, requires_grad=opt_scale)

    @property
    def output_size(self):
        return self.voxel_res * self.voxel_feat * 3

    @property
    def sample_feat_size(self):
        return self.voxel_feat * 3

    def get_axis_scale(self):
        return self.axis_scale


def _gcn_kwargs(init_adj_w, gcn_sep_bias, **flags):
    return dict(init_adj_w=init_adj_w, sep_bias=gcn_sep_bias, **flags)


def get_gnn_backbone(per_node_input, backbone='PNGCN', skel_type=None, gcn_D=4, gcn_sep_bias=False, node_W=64,
                     skip_gcn=10, output_ch=None, init_adj_w=0.05, no_adj=False, aggregate_dim=None, **kwargs):
    if backbone != 'MIXGNN':
        raise NotImplementedError(f"assignment backbone {backbone}: shipped configs use vox_MIXGNN")
    adj, _ = skeleton_to_graph(skel_type)
    return MixGNN(adj, per_node_input, W=node_W, D=gcn_D, skip_gcn=skip_gcn, skel_type=skel_type,
                  output_ch=output_ch, gcn_module=DensePNGCN,
                  gcn_module_kwargs=_gcn_kwargs(init_adj_w, gcn_sep_bias, no_adj=no_adj, aggregate_dim=aggregate_dim))


def get_volume_gnn_backbone(per_node_input, backbone='PNBGCN', skel_type=None, gcn_D=4, gcn_fc_D=0,
                            gcn_sep_bias=False, node_W=64, skip_gcn=10, voxel_res=4, voxel_feat=4, mask_root=False,
                            opt_scale=False, base_scale=0.5, attenuate_feat=False, attenuate_invalid=False,
                            no_adj=False, skel_profile=None, align_corners=False, aggregate_dim=None,
                            adj_self_one=False, init_adj_w=0.05, **kwargs):
    if not backbone.startswith('FGNN'):
        raise NotImplementedError(f"volume backbone {backbone}: shipped configs use FGNNcat")
    adj, _ = skeleton_to_graph(skel_type)
    return FactorizeGNN(adj, per_node_input, W=node_W, D=gcn_D, skip_gcn=skip_gcn, skel_type=skel_type,
                        fc_D=gcn_fc_D, voxel_res=voxel_res, voxel_feat=voxel_feat, mask_root=mask_root,
                        align_corners=align_corners, gcn_module=DensePNGCN, last_module=ParallelLinear,
                        factorize_type=backbone.split('FGNN')[-1] or 'sum', opt_scale=opt_scale,
                        base_scale=base_scale, skel_profile=skel_profile, attenuate_feat=attenuate_feat,
                        attenuate_invalid=attenuate_invalid,
                        gcn_module_kwargs=_gcn_kwargs(init_adj_w, gcn_sep_bias, no_adj=no_adj,
                                                      aggregate_dim=aggregate_dim, adj_self_one=adj_self_one))
