"""Drop-in for the reference's core/networks package (same public names)."""
from .nerf import NeRF
from .danbo import DANBO
from .embedding import Optcodes
from .misc import ParallelLinear, init_volume_scale
from .gnn_backbone import (BasicGNN, BodyGNN, DensePNGCN, DenseWGCN, FactorizeGNN, MixGNN, get_gnn_backbone,
                           get_volume_gnn_backbone, skeleton_to_graph)


def create_nerf(args, shared_nerf_kwargs, data_attrs):
    """-> (model, model_fine, caster_class); reference core/networks/__init__.py:7-69."""
    if args.nerf_type == 'nerf':
        cls, caster_class, extra = NeRF, None, {}
        shared_nerf_kwargs = {k: v for k, v in shared_nerf_kwargs.items() if k not in ('mask_vol_prob', 'agg_type')}
    elif args.nerf_type in ('graph', 'danbo'):
        cls, caster_class = DANBO, 'graph'
        extra = dict(node_W=args.node_W, voxel_res=args.voxel_res, voxel_feat=args.voxel_feat,
                     rest_pose=data_attrs['rest_pose'], backbone=args.gnn_backbone, gcn_D=args.gcn_D,
                     align_corners=args.align_corners, agg_backbone=args.agg_backbone, mask_root=args.mask_root,
                     adj_self_one=args.adj_self_one, gnn_concat=args.gnn_concat, opt_scale=args.opt_vol_scale,
                     aggregate_dim=args.aggregate_dim, init_adj_w=args.init_adj_w,
                     attenuate_feat=args.attenuate_feat, attenuate_invalid=args.attenuate_invalid,
                     agg_W=args.agg_W, agg_D=args.agg_D, gcn_fc_D=args.gcn_fc_D, no_adj=args.no_adj,
                     gcn_sep_bias=args.gcn_sep_bias, detach_agg_grad=args.detach_agg_grad,
                     use_posecode=args.opt_posecode, base_scale=0.4)
        if args.vol_cal_scale:
            extra['skel_profile'] = data_attrs['skel_profile']
    else:
        raise NotImplementedError(f'nerf class {args.nerf_type} is not implemented.')
    model = cls(**shared_nerf_kwargs, **extra)
    model_fine = None
    if args.N_importance > 0:
        model_fine = model if args.single_net else cls(**shared_nerf_kwargs, **extra)
    return model, model_fine, caster_class
