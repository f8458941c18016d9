"""Ray casters (reference: core/raycasters.py): model + optimizer construction, checkpoint
load, and the two-pass volumetric render of a ray batch.

The render loop is the reference's (`render_rays`, reference :245-377) re-expressed as one
stream-ordered chain of HIP kernels without host synchronisation:
    cylinder (+ per-bone box) bounds -> coarse z -> [cull -> gather/assign/blend -> PE+MLP]
    -> composite -> importance z + merge order -> [same network on the new samples] -> merge
    -> composite.
One process drives one GPU; multi-GPU runs shard rays across processes (no nn.DataParallel).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops as ops
from .encoders import get_pe_embedder, get_pts_embedder
from .render_engine import DanboEngine
from .networks import create_nerf
from .utils.skeleton_utils import SMPLSkeleton, bone_align_transforms, get_skel_profile_from_rest_pose


def get_density_fn(args):
    if args.density_type == 'relu':
        return F.relu
    raise NotImplementedError(f'density activation {args.density_type}: only relu is implemented')


def get_grad_vars(args, ray_caster):
    """Trainable parameters of the coarse (and, unless single_net, fine) network."""
    net, net_fine = ray_caster.get_networks()
    if getattr(args, 'finetune_light', False):
        for n, p in net.named_parameters():
            p.requires_grad = 'framecodes' in n
    out = [p for p in net.parameters() if p.requires_grad]
    if net_fine is not None and not args.single_net:
        out += [p for p in net_fine.parameters() if p.requires_grad]
    return out


def load_ckpt_from_path(ray_caster, optimizer, ckpt_path, finetune=False):
    ckpt = torch.load(ckpt_path, map_location='cpu')
    ray_caster.load_state_dict(ckpt)
    if optimizer is not None and not finetune and "optimizer_state_dict" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return ckpt["global_step"], ray_caster, optimizer, ckpt


def filter_state_dict(model_sd, ckpt_sd):
    """Non-strict reload rules (reference core/utils/run_nerf_helpers.py:23-50): an entry is taken from the checkpoint when
    name and shape match; a cutoff-embedder entry (`*pe_fn*`) missing from an older checkpoint keeps the model's value, except
    `tau`, which is set to 1000 (a converged, hard cutoff); frame codes of a different count are replaced by the mean code in
    every row; any other mismatch is left out (the model keeps its initialisation)."""
    out = {}
    for k, cur in model_sd.items():
        if k in ckpt_sd:
            v = ckpt_sd[k]
        elif 'pe_fn' in k:
            v = torch.tensor(1000.) if 'tau' in k else cur
            print(f'checkpoint has no {k}: using {v}')
        else:
            continue
        if tuple(v.shape) == tuple(cur.shape):
            out[k] = v
        elif 'framecodes' in k:
            print(f'{k}: {tuple(v.shape)} in the checkpoint, {tuple(cur.shape)} in the model -- loading the mean code')
            out[k] = v.mean(dim=0, keepdim=True).repeat(cur.shape[0], 1)
        else:
            print(f'{k}: {tuple(v.shape)} in the checkpoint, {tuple(cur.shape)} in the model -- not loaded')
    return out


def create_raycaster(args, data_attrs, device=None):
    """-> (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, loaded_ckpt)
    (reference :17-143)."""
    skel_type = data_attrs["skel_type"]
    n_framecodes = data_attrs["n_views"] if args.n_framecodes is None else args.n_framecodes
    pts_embedder, embed_dims = get_pts_embedder(args, data_attrs)
    pe_fns, dims = get_pe_embedder(args, data_attrs, embed_dims)
    if args.vol_cal_scale:
        data_attrs['skel_profile'] = get_skel_profile_from_rest_pose(data_attrs['rest_pose'], skel_type=skel_type)
    nerf_kwargs = dict(D=args.netdepth, W=args.netwidth, **dims, **pe_fns, pts_embedder=pts_embedder,
                       use_viewdirs=args.use_viewdirs, use_framecode=args.opt_framecode,
                       framecode_ch=args.framecode_size, n_framecodes=n_framecodes, skel_type=skel_type,
                       density_scale=args.density_scale, view_W=args.netwidth_view,
                       mask_vol_prob=args.mask_vol_prob, agg_type=args.agg_type)
    model, model_fine, caster_class = create_nerf(args, nerf_kwargs, data_attrs)
    kw = {}
    cls = RayCaster
    if caster_class is not None and caster_class.startswith('graph'):
        cls, kw = GraphCaster, dict(use_volume_near_far=args.use_volume_near_far)
    ray_caster = cls(model, network_fine=model_fine, rest_poses=data_attrs['rest_pose'], single_net=args.single_net,
                     align_bones=args.align_bones, skel_type=skel_type, **kw)
    if device is not None:
        ray_caster = ray_caster.to(device)
    grad_vars = get_grad_vars(args, ray_caster)
    if args.weight_decay is None:
        optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    else:
        optimizer = torch.optim.AdamW(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999),
                                      weight_decay=args.weight_decay)
    start, loaded = 0, None
    if args.ft_path is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f and 'pose' not in f] \
            if os.path.isdir(d) else []
    print('Found ckpts', ckpts)
    if ckpts and not args.no_reload:
        fin = getattr(args, 'finetune', False) or getattr(args, 'finetune_light', False)
        start, ray_caster, optimizer, loaded = load_ckpt_from_path(ray_caster, optimizer, ckpts[-1], fin)
        start = 0 if fin else start
    preproc = dict(density_scale=args.density_scale, density_fn=get_density_fn(args))
    train_kw = dict(ray_caster=ray_caster, perturb=args.perturb, N_importance=args.N_importance,
                    N_samples=args.N_samples, use_viewdirs=args.use_viewdirs, raw_noise_std=args.raw_noise_std,
                    ray_noise_std=args.ray_noise_std, ext_scale=args.ext_scale, preproc_kwargs=preproc,
                    lindisp=args.lindisp, nerf_type=args.nerf_type)
    test_kw = dict(train_kw, preproc_kwargs=dict(preproc), perturb=False, raw_noise_std=0., ray_noise_std=0.)
    print(f"#parameters: {sum(p.numel() for p in model.parameters() if p.requires_grad)}")
    optimizer.zero_grad()
    return train_kw, test_kw, start, grad_vars, optimizer, loaded


class RayCaster(nn.Module):
    def __init__(self, network, network_fine=None, single_net=False, rest_poses=None, align_bones=None,
                 skel_type=None, **kwargs):
        super().__init__()
        self.network, self.network_fine = network, network_fine
        self.rest_poses, self.skel_type, self.align_bones = rest_poses, skel_type or SMPLSkeleton, align_bones
        self.single_net = single_net
        if not single_net and network_fine is not None:
            raise NotImplementedError("single_net=False (separate fine network) is out of scope")
        if align_bones is not None:
            self.init_bone_align_transforms()

    # ---- HIP-graph replay of the eval chain for small chunks (see render_rays) ----
    use_graphs = True
    graph_max_rays = 8192
    whole_cast_max_samples = 1 << 26      # ray-samples per cast of render_rays_whole (512 x 512 x 64 is 1 << 24)

    @property
    def _graphs(self):
        if '_graph_cache' not in self.__dict__:
            self.__dict__['_graph_cache'] = _GraphCache()
        return self.__dict__['_graph_cache']

    def init_bone_align_transforms(self):
        if self.align_bones != 'align':
            raise NotImplementedError("align_bones must be 'align'")
        rest = np.asarray(self.rest_poses).reshape(-1, len(self.skel_type.joint_trees), 3)
        self.transforms = torch.tensor(np.stack([bone_align_transforms(r, self.skel_type) for r in rest]))
        self.child_idxs = None

    def get_networks(self):
        return self.network, self.network_fine

    def update_embed_fns(self, global_step, args):
        self.network.update_embed_fns(global_step, args)

    # ---- checkpoint format of the reference (:601-637): one sub-dict per sub-module ----
    @staticmethod
    def _ckpt_key(k):
        if k.endswith("_fine"):
            return f"{k}_state_dict"
        if k.endswith("_fn"):
            return f"{k.split('_fn')[0]}_state_dict"
        return "network_fn_state_dict" if k == "network" else f"{k}_state_dict"

    def state_dict(self):
        return {self._ckpt_key(k): m.state_dict() for k, m in self._modules.items() if m is not None}

    def load_state_dict(self, ckpt, strict=True):
        for k, m in self._modules.items():
            if m is None:
                continue
            key = self._ckpt_key(k)
            try:
                m.load_state_dict(ckpt[key], strict=strict)
            except (KeyError, RuntimeError):
                if k.startswith('network') and key in ckpt:
                    print('Error occur when loading state dict for network. Try loading with strict=False now')
                    m.load_state_dict(filter_state_dict(m.state_dict(), ckpt[key]), strict=False)
                else:
                    print(f'Error occurr when loading state dict for {key}. The entity is not in the state dict?')

    # ---- forward dispatch (reference :233-243) ----
    def forward(self, *args, fwd_type='', **kwargs):
        if fwd_type == 'density':
            return self.render_pts_density(*args, **kwargs)
        if fwd_type == 'mesh':
            return self.render_mesh_density(*args, **kwargs)
        if fwd_type:
            raise NotImplementedError(fwd_type)
        if self.training:
            return self.render_rays_train(*args, **kwargs)
        with torch.no_grad():
            return self.render_rays(*args, **kwargs)

    def _engine(self, refresh=True):
        dev = next(self.network.parameters()).device
        if self.transforms.device != dev:
            self.transforms = self.transforms.to(dev)
        eng = self.network.engine(self.transforms[0])
        eng.cfg['use_volume_near_far'] = bool(getattr(self, 'use_volume_near_far', False))
        if refresh:      # re-packs the kernels' weight buffers when a parameter version changed: the eval path needs that
            eng.refresh()
        return eng

    @staticmethod
    def _per_pose(x, G):
        return x if x.shape[0] == G else x[::max(x.shape[0] // G, 1)].contiguous()

    def render_rays(self, ray_batch, N_samples, kp_batch, skts=None, cyls=None, bones=None, cams=None,
                    subject_idxs=None, retraw=False, lindisp=False, perturb=0., N_importance=0, network_fine=None,
                    raw_noise_std=0., ray_noise_std=0., verbose=False, ext_scale=0.001, pytest=False, N_uniques=1,
                    render_confd=False, render_entropy=False, preproc_kwargs={}, netchunk=1024 * 64,
                    nerf_type="nerf", **kwargs):
        if N_importance <= 0:
            raise NotImplementedError("N_importance=0 raises in the reference too (raycasters.py:377)")
        if perturb or raw_noise_std or ray_noise_std or lindisp:
            raise NotImplementedError("stochastic sampling belongs to the training path")
        # render_confd / render_entropy: accepted and -- exactly as in the reference -- without effect here: its render_rays takes
        # the two flags (raycasters.py:265-266) but never hands them to raw2outputs (:334-337, :373-375), whose colourings
        # (nerf.py:306-311) would moreover need raw[..., 4:], which no shipped network produces.  The colourings themselves are
        # available on NeRF.raw2outputs (core/networks/nerf.py) for a caller that passes the assignment logits in raw[..., 4:].
        eng = self._engine()
        G = int(N_uniques)
        skts_g, bones_g, cyls_g = self._per_pose(skts, G), self._per_pose(bones, G), self._per_pose(cyls, G)
        eng.cfg['density_scale'] = preproc_kwargs.get('density_scale', eng.cfg['density_scale'])
        R = ray_batch.shape[0]

        def chain(rb, skts_g, bones_g, cyls_g, cams):
            rays_o, rays_d = rb[:, 0:3].contiguous(), rb[:, 3:6].contiguous()
            near, far = ops.near_far_cylinder(rays_o, rays_d, cyls_g, 0., 1., R, rb[:, 6], rb[:, 7])
            if eng.cfg['use_volume_near_far']:
                ops.near_far_boxes(rays_o, rays_d, skts_g, eng.align, eng.axis_scale, near, far)
            return eng.render(rays_o, rays_d, skts_g, bones_g, cyls_g, cams, N_samples, N_importance, near_far=(near, far))

        # Small ray chunks (the reference casts `chunk // 8` = 512 rays at a time during validation) are launch-bound: the
        # ~25 kernels of the chain are captured once per chunk shape as a HIP graph and replayed.
        if self.use_graphs and R <= self.graph_max_rays and isinstance(eng, DanboEngine):
            key = (R, G, int(N_samples), int(N_importance), cams is not None, eng.cfg['use_volume_near_far'],
                   float(eng.cfg['density_scale']))
            return self._graphs.run(eng, key, chain, ray_batch, skts_g, bones_g, cyls_g, cams)
        return chain(ray_batch, skts_g, bones_g, cyls_g, cams)

    @staticmethod
    def _one_pose(x):
        """x [R, ...] is ONE row behind every ray: an expanded view (stride 0, what render_path builds) or R == 1"""
        return x is not None and (x.shape[0] == 1 or x.stride(0) == 0)

    @torch.no_grad()
    def render_rays_whole(self, ray_batch, chunk, N_samples=None, kp_batch=None, skts=None, cyls=None, bones=None, cams=None,
                          lindisp=False, perturb=0., N_importance=0, raw_noise_std=0., ray_noise_std=0., N_uniques=1,
                          preproc_kwargs={}, fwd_type='', rays=None, near_far0=None, **kwargs):
        """`trainer.batchify_rays`' loop over `chunk`-ray casts as ONE cast of all rays, when that is the same computation, else
        None (the caller loops).  It is the same when every ray belongs to one pose (so a chunk's N_uniques = 1 whatever the
        chunking) and the engine is the DANBO engine: every stage is per ray or per sample except the cylinder bounds' nan-mean
        back-fill, which the bounds kernel takes per `chunk` rays -- exactly one reference call each (ray_utils.py:330-344)."""
        if (self.training or fwd_type or N_importance <= 0 or perturb or raw_noise_std or ray_noise_std or lindisp
                or int(N_uniques) != 1 or not all(self._one_pose(x) for x in (skts, bones, cyls))):
            return None
        eng = self._engine()
        if not isinstance(eng, DanboEngine) or N_samples > 256 or N_importance > 64:
            return None
        eng.cfg['density_scale'] = preproc_kwargs.get('density_scale', eng.cfg['density_scale'])
        skts_g, bones_g, cyls_g = skts[:1].contiguous(), bones[:1].contiguous(), cyls[:1].contiguous()
        chunk = int(chunk)
        R = rays[0].shape[0] if rays is not None else ray_batch.shape[0]
        # `chunk` bounds the reference's memory; here it only sets the nan-mean group, so the cast itself is bounded: super-chunks
        # of whole `chunk`s with at most `whole_cast_max_samples` ray-samples each (the engine keeps ~100 B per ray-sample alive:
        # 6.4 GB at the default) -- bit-identical to one cast, every stage being per ray, per sample or per `chunk`
        sub = max(chunk, self.whole_cast_max_samples // (int(N_samples) + int(N_importance)) // chunk * chunk)

        def cast(a, b):
            if rays is not None:       # trainer.render hands the rays over as they are (scalar placeholder bounds near_far0)
                rays_o, rays_d = rays[0][a:b].contiguous(), rays[1][a:b].contiguous()
                near, far = ops.near_far_cylinder(rays_o, rays_d, cyls_g, near_far0[0], near_far0[1], chunk)
            else:
                rb = ray_batch[a:b]
                rays_o, rays_d = rb[:, 0:3].contiguous(), rb[:, 3:6].contiguous()
                near, far = ops.near_far_cylinder(rays_o, rays_d, cyls_g, 0., 1., chunk, rb[:, 6], rb[:, 7])
            if eng.cfg['use_volume_near_far']:
                ops.near_far_boxes(rays_o, rays_d, skts_g, eng.align, eng.axis_scale, near, far)
            return eng.render(rays_o, rays_d, skts_g, bones_g, cyls_g, None if cams is None else cams[a:b], N_samples, N_importance,
                              near_far=(near, far))

        if R <= sub:
            return cast(0, R)
        parts = [cast(a, min(a + sub, R)) for a in range(0, R, sub)]
        return {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}

    def render_rays_train(self, ray_batch, N_samples, kp_batch, skts=None, cyls=None, bones=None, cams=None,
                          subject_idxs=None, lindisp=False, perturb=0., N_importance=0, raw_noise_std=0.,
                          ray_noise_std=0., N_uniques=1, preproc_kwargs={}, netchunk=1024 * 64, **kwargs):
        """Differentiable two-pass render (reference render_rays :245-377 in training mode): sampling is
        detached, the network and the compositing carry gradients (core/train_path.py)."""
        if N_importance <= 0 or lindisp or ray_noise_std:
            raise NotImplementedError("training needs N_importance > 0, lindisp=False, ray_noise_std=0")
        # the training forward reads the parameters themselves: no re-pack of the eval kernels' buffers after every optimizer step
        eng = self._engine(refresh=False)
        G = int(N_uniques)
        R = ray_batch.shape[0]
        rays_o, rays_d = ray_batch[:, 0:3].contiguous(), ray_batch[:, 3:6].contiguous()
        skts_g, bones_g, cyls_g = self._per_pose(skts, G), self._per_pose(bones, G), self._per_pose(cyls, G)
        B = preproc_kwargs.get('density_scale', 1.0)
        with torch.no_grad():
            near, far = ops.near_far_cylinder(rays_o, rays_d, cyls_g, 0., 1., R, ray_batch[:, 6], ray_batch[:, 7])
            if eng.cfg.get('use_volume_near_far'):
                ops.near_far_boxes(rays_o, rays_d, skts_g, eng.align, self.network.graph_net.axis_scale.detach().contiguous(),
                                   near, far)
            t_rand = torch.rand(R, N_samples, device=rays_o.device) if perturb > 0. else None
            z = ops.coarse_samples(near, far, N_samples, t_rand)
        align = self.transforms[:1, None].to(rays_o.device)

        shared = {}   # per-pose volumes, per-ray view inputs and the empty-space evaluation are the same in both passes

        def net(zv):
            pts = rays_o[:, None, :] + rays_d[:, None, :] * zv[:, :, None]
            inputs = dict(pts=pts, kps=None, skts=skts_g, bones=bones_g, align_transforms=align, N_uniques=G,
                          rays_o=rays_o[:, None], rays_d=rays_d[:, None], cam_idxs=cams, shared=shared)
            return self.network(inputs)

        raw, enc = net(z)
        out0 = self.network.raw2outputs(raw, z, rays_d, raw_noise_std=raw_noise_std, B=B)
        with torch.no_grad():
            u = torch.rand(R, N_importance, device=rays_o.device) if perturb > 0. else None
            z_all, z_fine, order = ops.importance_samples(z, out0['weights'], N_importance, u)
        raw_f, enc_f = net(z_fine)
        idx = order.long().clamp_(0, N_samples + N_importance - 1)   # a permutation unless depths are NaN
        take = lambda a, b: torch.gather(torch.cat([a, b], 1), 1, idx[..., None].expand(-1, -1, a.shape[-1]))  # noqa: E731
        raw_all = take(raw, raw_f)
        out = self.network.raw2outputs(raw_all, z_all, rays_d, raw_noise_std=raw_noise_std, B=B)
        ret = dict(rgb_map=out['rgb_map'], disp_map=out['disp_map'], acc_map=out['acc_map'], alpha=out['alpha'],
                   T_i=out['weights'], rgb0=out0['rgb_map'], disp0=out0['disp_map'], acc0=out0['acc_map'],
                   alpha0=out0['alpha'])
        if 'confd' in enc:   # DANBO: the assignment logits feed the soft-softmax loss (reference :710-716)
            ret.update(confd=take(enc['confd'], enc_f['confd']), part_invalid=take(enc['part_invalid'], enc_f['part_invalid']))
            if 'p_valid' in enc and 'p_valid' in enc_f:      # torch.ops.danbo.assign_blend: the differentiable masked probabilities
                ret['p_valid'] = take(enc['p_valid'], enc_f['p_valid'])
        return ret

    def render_pts_density(self, pts, kps, skts, bones, netchunk=1024 * 64, network=None):
        assert kps.shape[0] == 1, f'Assuming only one pose is provided, got {kps.shape[0]} instead'
        eng = self._engine()
        return eng.density(pts.reshape(-1, 1, 3), skts[:1], bones[:1], netchunk=int(netchunk))

    @torch.no_grad()
    def render_mesh_density(self, kps, skts, bones, subject_idxs=None, radius=1.0, res=64, render_kwargs=None,
                            netchunk=1024 * 64, v=None):
        # the reference's grid (raycasters.py:425-429): float64 linspace rounded to float32, 'xy' meshgrid, + the root joint
        t = np.linspace(-radius, radius, res + 1)
        grid = torch.tensor(np.stack(np.meshgrid(t, t, t), axis=-1).astype(np.float32), device=kps.device)
        dens = self.render_pts_density(grid.reshape(-1, 3) + kps[0, 0], kps, skts, bones, netchunk)[..., :1]
        return dens.reshape(*grid.shape[:-1]).transpose(1, 0)      # x-y swapped, as the mesh extraction expects


class _GraphCache:
    """One captured HIP graph per (chunk shape, sampling, engine state).  The engine's packed weights are part of the
    captured pointers, so every graph is dropped when the engine has been refreshed with new parameter versions."""

    def __init__(self, max_graphs=8):
        self.max_graphs, self.engine_key, self.graphs = max_graphs, None, {}

    def run(self, eng, key, chain, rb, skts_g, bones_g, cyls_g, cams):
        state = (eng._packed_key, eng.mlp_mode)
        if self.engine_key != state:
            self.graphs.clear()
            self.engine_key = state
        g = self.graphs.get(key)
        if g is None:
            if len(self.graphs) >= self.max_graphs:
                self.graphs.pop(next(iter(self.graphs)))
            g = self.graphs[key] = _ChainGraph(chain, rb, skts_g, bones_g, cyls_g, cams)
        return g(rb, skts_g, bones_g, cyls_g, cams)


class _ChainGraph:
    def __init__(self, chain, rb, skts_g, bones_g, cyls_g, cams):
        self.inputs = [x.clone().contiguous() if x is not None else None for x in (rb.float(), skts_g.float(), bones_g.float(),
                                                                                   cyls_g.float(), cams)]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):       # eager warm-up off the capture: lazy initialisations, allocator pools
            chain(*self.inputs)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out = chain(*self.inputs)
            self.keys = list(out)
            self.shapes = [tuple(out[k].shape) for k in self.keys]
            R = rb.shape[0]
            self.packed = torch.cat([out[k].reshape(R, -1) for k in self.keys], 1)   # one buffer to copy out per replay

    def __call__(self, rb, skts_g, bones_g, cyls_g, cams):
        for dst, src in zip(self.inputs, (rb, skts_g, bones_g, cyls_g, cams)):
            if dst is not None:
                dst.copy_(src)
        self.graph.replay()
        flat = self.packed.clone()
        out, c = {}, 0
        for k, shp in zip(self.keys, self.shapes):
            w = 1
            for d in shp[1:]:
                w *= d
            out[k] = flat[:, c:c + w].reshape(shp)
            c += w
        return out


class GraphCaster(RayCaster):
    def __init__(self, *args, use_volume_near_far=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.use_volume_near_far = use_volume_near_far


def merge_samples(x, x_is, gather_idxs, N_total_samples):
    """Interleave coarse / importance tensors by the sorted order (reference :745-761);
    gather_idxs here is the int32 sorted index [R, S+Sf] returned by isample_from_lineseg."""
    if x is None or x.shape[-1] == 0:
        return None
    return ops.merge_samples(x, x_is, gather_idxs)
