"""The differentiable HIP stages as `torch.library` custom operators ("Python wraps each pair as a
torch.library.custom_op with register_autograd"), so that a caller -- the reference's trainer included -- can trace or
`torch.compile` around them:

    torch.ops.danbo.pose_volumes(bones, L_graph, params)                  rot6d + PE + FactorizeGNN (reference core/networks/gnn_backbone.py:683-704)
        -> volumes [G,24,240] (+ the adjoint's scratch)                  backward: danbo_pose_volumes_bwd -> every parameter's gradient
    torch.ops.danbo.composite(raw, z, rays_d, B, noise)                  NeRF.raw2outputs (reference core/networks/nerf.py:281-347)
        -> rgb_map, disp_map, acc_map, weights, alpha                     backward: danbo_composite_bwd (d rgb_map, d acc_map -> d raw)
    torch.ops.danbo.bone_gather(volumes, axis_scale, pts, skts, align, rows)
        -> part_feat [n,24,15]                                            FactorizeGNN.sample_from_volume (gnn_backbone.py:787-828)
                                                                          backward: danbo_bone_gather_bwd (d volumes, d axis_scale)

    torch.ops.danbo.assign_blend(volumes, axis_scale, pts, skts, align, rows, bits, [8 parameter tensors])
        -> h [n,16], p [n,24], confd [n,24]                               sample_from_volume + MixGNN + DANBO.sigmoid / blend fused
                                                                          (gnn_backbone.py:567-629,787-828, danbo.py:299-300,406-415);
                                                                          backward: danbo_assign_blend_bwd (forward recomputed)
    torch.ops.danbo.anerf_cutoff_pe_mlp(pts, rays_d, skts, align, cam_idx, params, tau, multires, multires_views)
        -> raw [R,S,4]                                                    A-NeRF's NeRF.forward (nerf.py:107-122,176-209: joint-distance
                                                                          cutoff PE -> W-wide trunk -> view layer -> colour); forward
                                                                          only, as SURVEY 8(b) lists it
    torch.ops.danbo.pe_mlp(h, row_ray, vin, [20 parameter tensors])       Embedder + NeRF.inference (cutoff_embedder.py:62-73,
        -> raw [n,4]                                                      nerf.py:176-209) on the fused trunk kernels; backward:
                                                                          danbo_trunk_bwd + danbo_dw16 + the view / head chain

All enqueue on the current HIP stream through the C ABI (include/danbo_hip.h); there is no CPU implementation (calling them
with CPU tensors raises), only shape-propagating fake kernels for tracing.  core/train_path.py (the autograd training path)
is built on composite / bone_gather; the fused training step (core/train_engine.py) does not need autograd at all.
"""
import ctypes
from typing import List, Optional, Tuple

import torch

from . import _hip
from . import hip_ops as ops


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------------------------------ composite
@torch.library.custom_op("danbo::composite", mutates_args=())
def composite(raw: torch.Tensor, z: torch.Tensor, rays_d: torch.Tensor, B: float,
              noise: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    out = ops.composite(raw, z, rays_d, B, noise)
    return out["rgb_map"], out["disp_map"], out["acc_map"], out["weights"], out["alpha"]


@composite.register_fake
def _(raw, z, rays_d, B, noise):
    R, S = z.shape
    f = lambda *s: raw.new_empty(s, dtype=torch.float32)  # noqa: E731
    return f(R, 3), f(R), f(R), f(R, S), f(R, S)


@torch.library.custom_op("danbo::composite_bwd", mutates_args=())
def composite_bwd(raw: torch.Tensor, z: torch.Tensor, rays_d: torch.Tensor, B: float, noise: Optional[torch.Tensor],
                  g_rgb: torch.Tensor, g_acc: torch.Tensor) -> torch.Tensor:
    R, S = z.shape
    raw, z, rays_d = (ops._f32(t, n) for t, n in ((raw, "raw"), (z, "z"), (rays_d, "rays_d")))
    d_raw = torch.empty(R, S, 4, dtype=torch.float32, device=raw.device)
    _hip.check(_hip.lib().danbo_composite_bwd(_p(raw), _p(z), _p(rays_d), R, S, float(B), _p(ops._f32(noise, "noise")),
                                              _p(ops._f32(g_rgb, "g_rgb")), _p(ops._f32(g_acc, "g_acc")), _p(d_raw), ops._stream()),
               "danbo_composite_bwd")
    return d_raw


@composite_bwd.register_fake
def _(raw, z, rays_d, B, noise, g_rgb, g_acc):
    return raw.new_empty(tuple(z.shape) + (4,), dtype=torch.float32)


def _composite_setup(ctx, inputs, output):
    raw, z, rays_d, B, noise = inputs
    ctx.save_for_backward(raw, z, rays_d, noise)
    ctx.B = B


def _composite_backward(ctx, g_rgb, g_disp, g_acc, g_w, g_alpha):
    # disp_map, weights and alpha carry no gradient in the reference's losses (core/trainer.py:396-422,507-536)
    raw, z, rays_d, noise = ctx.saved_tensors
    R = z.shape[0]
    g_rgb = g_rgb if g_rgb is not None else raw.new_zeros(R, 3)
    g_acc = g_acc if g_acc is not None else raw.new_zeros(R)
    return torch.ops.danbo.composite_bwd(raw, z, rays_d, ctx.B, noise, g_rgb.contiguous(), g_acc.contiguous()).reshape(raw.shape), \
        None, None, None, None


composite.register_autograd(_composite_backward, setup_context=_composite_setup)


# ---------------------------------------------------------------------------------------------------------------- bone gather
@torch.library.custom_op("danbo::bone_gather", mutates_args=())
def bone_gather(volumes: torch.Tensor, axis_scale: torch.Tensor, pts: torch.Tensor, skts: torch.Tensor, align: torch.Tensor,
                rows: torch.Tensor) -> torch.Tensor:
    R = pts.shape[0]
    dummy = pts.new_zeros(R, 3)
    geo = ops.Geometry(dummy, dummy, skts, align, axis_scale, pts=pts)
    return ops.bone_gather(geo, volumes.contiguous(), rows, None, rows.shape[0])


@bone_gather.register_fake
def _(volumes, axis_scale, pts, skts, align, rows):
    return volumes.new_empty(rows.shape[0], ops.J, ops.FEAT, dtype=torch.float32)


@torch.library.custom_op("danbo::bone_gather_bwd", mutates_args=())
def bone_gather_bwd(volumes: torch.Tensor, axis_scale: torch.Tensor, pts: torch.Tensor, skts: torch.Tensor, align: torch.Tensor,
                    rows: torch.Tensor, g: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    volumes, axis_scale = volumes.contiguous().float(), axis_scale.contiguous().float()
    pts, skts, align = ops._f32(pts, "pts"), ops._f32(skts, "skts"), ops._f32(align, "align")
    R, S, G, n = pts.shape[0], pts.shape[1], skts.shape[0], rows.shape[0]
    d_vol = torch.zeros(volumes.shape, dtype=torch.float32, device=volumes.device)
    d_sc = torch.zeros(axis_scale.shape, dtype=torch.float32, device=volumes.device)
    if n > 0:
        _hip.check(_hip.lib().danbo_bone_gather_bwd(None, None, None, _p(pts), R, S, G, _p(skts), _p(align), _p(axis_scale), _p(volumes),
                                                    _p(rows), n, _p(g.contiguous().float()), _p(d_vol), _p(d_sc), ops._stream()),
                   "danbo_bone_gather_bwd")
    return d_vol, d_sc


@bone_gather_bwd.register_fake
def _(volumes, axis_scale, pts, skts, align, rows, g):
    return volumes.new_empty(volumes.shape, dtype=torch.float32), axis_scale.new_empty(axis_scale.shape, dtype=torch.float32)


def _gather_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _gather_backward(ctx, g):
    volumes, axis_scale, pts, skts, align, rows = ctx.saved_tensors
    d_vol, d_sc = torch.ops.danbo.bone_gather_bwd(volumes, axis_scale, pts, skts, align, rows, g)
    # window detached, in-volume mask not differentiable (gnn_backbone.py:804,808); no gradient to points / transforms (opt_pose off)
    return d_vol, d_sc, None, None, None, None


bone_gather.register_autograd(_gather_backward, setup_context=_gather_setup)


# ----------------------------------------------------------------------------------------------------------------- pose volumes
# torch.ops.danbo.pose_volumes(bones [G,24,3], L_graph, params) -> volumes [G,24,240], scratch
#   params = graph_net.layers.{0.lin.weight [24,6(1+2L),W], 0.adj_w [1,24,24], 0.adj [1,24,24], 0.bias [W],
#                              1.lin.weight [24,W,W], 1.adj_w, 1.adj, 1.bias [W], 2.weight [24,W,W], 2.bias [1,24,W],
#                              3.weight [24,W,240], 3.bias [1,24,240]}       (the reference's state_dict tensors)
#   = encode_graph_inputs (core/encoders.py:460-473) + AxisAngtoRot6DEncoder (:859-877) + Embedder (cutoff_embedder.py:62-73) +
#     FactorizeGNN / BodyGNN.forward (core/networks/gnn_backbone.py:683-704) incl. mask_root and the doubled first layer.
#   forward: danbo_pose_volumes_fwd (csrc/k_pose.hip); backward: danbo_pose_volumes_bwd (csrc/k_pose_bwd.hip) -> the gradient of
#   every parameter (the 0/1 adjacency buffers get None; no gradient to the pose: opt_pose is off).
#   `scratch` [3 G 24 W]: the layer activations the forward leaves for the adjoint -- an output so that autograd keeps it alive;
#   not differentiable, ignore it.
def _pose_params(params):
    w0, aw0, a0, b0, w1, aw1, a1, b1, w2, b2, w3, b3 = params
    W = w1.shape[-1]
    f = lambda t, *shape: ops._f32(t.detach().reshape(*shape), "graph_net")  # noqa: E731
    return dict(w0=f(w0, 24, -1, W), aw0=f(aw0, 24, 24), a0=f(a0, 24, 24), b0=f(b0, W), w1=f(w1, 24, W, W), aw1=f(aw1, 24, 24),
                a1=f(a1, 24, 24), b1=f(b1, W), w2=f(w2, 24, W, W), b2=f(b2, 24, W), w3=f(w3, 24, W, ops.VOL), b3=f(b3, 24, ops.VOL)), W


@torch.library.custom_op("danbo::pose_volumes", mutates_args=())
def pose_volumes(bones: torch.Tensor, L_graph: int, params: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    if not bones.is_cuda:
        raise RuntimeError("danbo::pose_volumes runs on the HIP path only (no CPU fallback)")
    a, W = _pose_params(params)
    if a["w0"].shape[1] != 6 * (1 + 2 * int(L_graph)):
        raise ValueError(f"danbo::pose_volumes: layer 0 takes {a['w0'].shape[1]} inputs, the rot6d encoding with L = {L_graph} has "
                         f"{6 * (1 + 2 * int(L_graph))}")
    bones = ops._f32(bones.detach(), "bones")
    G = bones.shape[0]
    scratch = torch.empty(3 * G * ops.J * W, device=bones.device, dtype=torch.float32)
    vol = torch.empty(G, ops.J, ops.VOL, device=bones.device, dtype=torch.float32)
    # (named: a temporary handed to _p() is freed -- and its block re-used by the next temporary -- before the launch)
    adjw0, adjw1 = (a["aw0"] * a["a0"]).contiguous(), (a["aw1"] * a["a1"]).contiguous()
    _hip.check(_hip.lib().danbo_pose_volumes_fwd(_p(bones), G, int(L_graph), W, _p(a["w0"]), _p(adjw0), _p(a["b0"]), _p(a["w1"]), _p(adjw1),
                                                 _p(a["b1"]), _p(a["w2"]), _p(a["b2"]), _p(a["w3"]), _p(a["b3"]), _p(scratch), _p(vol),
                                                 ops._stream()), "danbo_pose_volumes_fwd")
    return vol, scratch


@pose_volumes.register_fake
def _(bones, L_graph, params):
    G, W = bones.shape[0], params[4].shape[-1]
    return bones.new_empty(G, ops.J, ops.VOL, dtype=torch.float32), bones.new_empty(3 * G * ops.J * W, dtype=torch.float32)


@torch.library.custom_op("danbo::pose_volumes_bwd", mutates_args=())
def pose_volumes_bwd(bones: torch.Tensor, L_graph: int, params: List[torch.Tensor], scratch: torch.Tensor,
                     g_vol: torch.Tensor) -> List[torch.Tensor]:
    """-> [d w0, d adj_w0, d b0, d w1, d adj_w1, d b1, d w2, d b2, d w3, d b3] in the parameters' own shapes"""
    a, W = _pose_params(params)
    bones = ops._f32(bones.detach(), "bones")
    G, dev = bones.shape[0], bones.device
    names = ("w0", "aw0", "b0", "w1", "aw1", "b1", "w2", "b2", "w3", "b3")
    g = {k: torch.zeros(a[k].shape, device=dev, dtype=torch.float32) for k in names}     # (adj_w / b0 / b1 are accumulated into)
    bwd_scratch = torch.empty(2 * G * ops.J * W, device=dev, dtype=torch.float32)
    _hip.check(_hip.lib().danbo_pose_volumes_bwd(
        _p(bones), G, int(L_graph), W, _p(a["w0"]), _p(a["aw0"]), _p(a["a0"]), _p(a["b0"]), _p(a["w1"]), _p(a["aw1"]), _p(a["a1"]), _p(a["b1"]),
        _p(a["w2"]), _p(a["w3"]), _p(ops._f32(scratch, "scratch")), _p(ops._f32(g_vol, "g_vol")), _p(g["w0"]), _p(g["aw0"]), _p(g["b0"]),
        _p(g["w1"]), _p(g["aw1"]), _p(g["b1"]), _p(g["w2"]), _p(g["b2"]), _p(g["w3"]), _p(g["b3"]), _p(bwd_scratch), ops._stream()),
        "danbo_pose_volumes_bwd")
    w0, aw0, a0, b0, w1, aw1, a1, b1, w2, b2, w3, b3 = params
    return [g[k].reshape(t.shape) for k, t in zip(names, (w0, aw0, b0, w1, aw1, b1, w2, b2, w3, b3))]


@pose_volumes_bwd.register_fake
def _(bones, L_graph, params, scratch, g_vol):
    w0, aw0, a0, b0, w1, aw1, a1, b1, w2, b2, w3, b3 = params
    return [torch.empty_like(t, dtype=torch.float32) for t in (w0, aw0, b0, w1, aw1, b1, w2, b2, w3, b3)]


def _pose_setup(ctx, inputs, output):
    bones, L_graph, params = inputs
    ctx.L_graph = L_graph
    ctx.save_for_backward(bones, output[1], *params)
    ctx.mark_non_differentiable(output[1])


def _pose_backward(ctx, g_vol, g_scratch):
    bones, scratch, *params = ctx.saved_tensors
    d = torch.ops.danbo.pose_volumes_bwd(bones, ctx.L_graph, list(params), scratch, g_vol.contiguous())
    d_w0, d_aw0, d_b0, d_w1, d_aw1, d_b1, d_w2, d_b2, d_w3, d_b3 = d
    return None, None, [d_w0, d_aw0, None, d_b0, d_w1, d_aw1, None, d_b1, d_w2, d_b2, d_w3, d_b3]


pose_volumes.register_autograd(_pose_backward, setup_context=_pose_setup)


# ------------------------------------------------------------------------------------------------------- gather + assign + blend
# torch.ops.danbo.assign_blend(volumes [G,24,240], axis_scale [24,3], pts [R,S,3], skts [G,24,4,4], align [24,4,4],
#                              rows [n] int32 (ascending sample ids inside >= 1 volume), bits [R*S] int32 (bone j valid: bit j),
#                              params = prob_linears.layers.{0.lin.weight [24,15,32], 0.adj_w [1,24,24], 0.adj [1,24,24], 0.bias [32],
#                                                            1.weight [24,32,32], 1.bias [24,1,32], 2.weight [24,32,1], 2.bias [24,1,1]})
#   -> h [n,16]     blended voxel feature sum_j p_j f_j (15 channels + zero pad)                     differentiable
#      p [n,24]     p_j = (1.002 sigmoid(logit_j) - 0.001) valid_j: DANBO.sigmoid(mask_invalid=False) x part_valid, what the
#                   soft-softmax loss sums (core/trainer.py:507-536)                                   differentiable
#      confd [n,24] the raw logits the reference returns in `encoded` (all bones, valid or not): a monitoring output, NOT
#                   differentiable -- take gradients through p
# The 1 440 B / row part_feat tensor of the unfused forward is never written; the backward recomputes the gather and the GNN from
# the 23 KB pose volumes (csrc/k_assign_bwd.hip) and returns d volumes, d axis_scale and the parameter gradients (d adj is None: a
# buffer).  No gradient to pts / skts / align (window detached, mask not differentiable, opt_pose off: gnn_backbone.py:804,808).
def _assign_params(params):
    w0, adj_w, adj, b0, w1, b1, w2, b2 = params
    f = lambda t, *shape: ops._f32(t.detach().reshape(*shape), "prob_linears")  # noqa: E731
    return dict(w0=f(w0, 24, ops.FEAT, 32), adj_w=f(adj_w, 24, 24), adj=f(adj, 24, 24), b0=f(b0, 32), w1=f(w1, 24, 32, 32),
                b1=f(b1, 24, 32), w2=f(w2, 24, 32), b2=f(b2, 24))


def _valid_mask(bits, rows):
    shifts = torch.arange(ops.J, device=bits.device, dtype=torch.int32)
    return ((bits.reshape(-1)[rows.long()].to(torch.int32).unsqueeze(-1) >> shifts) & 1).float()


@torch.library.custom_op("danbo::assign_blend", mutates_args=())
def assign_blend(volumes: torch.Tensor, axis_scale: torch.Tensor, pts: torch.Tensor, skts: torch.Tensor, align: torch.Tensor,
                 rows: torch.Tensor, bits: torch.Tensor, params: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    if not pts.is_cuda:
        raise RuntimeError("danbo::assign_blend runs on the HIP path only (no CPU fallback)")
    a = _assign_params(params)
    R = pts.shape[0]
    n = rows.shape[0]
    dev = pts.device
    if n == 0:
        return pts.new_zeros(0, ops.H_STRIDE), pts.new_zeros(0, ops.J), pts.new_zeros(0, ops.J)
    dummy = pts.new_zeros(R, 3)
    geo = ops.Geometry(dummy, dummy, ops._f32(skts, "skts"), ops._f32(align, "align"), ops._f32(axis_scale.detach(), "axis_scale"),
                       pts=ops._f32(pts, "pts"))
    aw = dict(w0=a["w0"], adjw=(a["adj_w"] * a["adj"]).contiguous(), b0=a["b0"], w1=a["w1"], b1=a["b1"], w2=a["w2"], b2=a["b2"])
    cnt = torch.tensor([n], device=dev, dtype=torch.int32)
    # exact-fp32 K1b + K2 (csrc/k_assign.hip): reads the adjacency it is given, any tree
    h, confd = ops.gather_assign_blend(geo, ops._f32(volumes.detach(), "volumes"), bits.reshape(-1).contiguous(), aw,
                                       rows.to(torch.int32).contiguous(), cnt, n, want_confd=True)
    p = (torch.sigmoid(confd) * 1.002 - 0.001) * _valid_mask(bits, rows)
    return h, p, confd


@assign_blend.register_fake
def _(volumes, axis_scale, pts, skts, align, rows, bits, params):
    n = rows.shape[0]
    f = lambda *s: pts.new_empty(s, dtype=torch.float32)  # noqa: E731
    return f(n, ops.H_STRIDE), f(n, ops.J), f(n, ops.J)


@torch.library.custom_op("danbo::assign_blend_bwd", mutates_args=())
def assign_blend_bwd(volumes: torch.Tensor, axis_scale: torch.Tensor, pts: torch.Tensor, skts: torch.Tensor, align: torch.Tensor,
                     rows: torch.Tensor, bits: torch.Tensor, params: List[torch.Tensor], g_h: torch.Tensor,
                     g_p: torch.Tensor) -> List[torch.Tensor]:
    """-> [d volumes, d axis_scale, d w0, d adj_w, d b0, d w1, d b1, d w2, d b2] (the parameters' own shapes)"""
    a = _assign_params(params)
    dev = pts.device
    R, S = pts.shape[0], pts.shape[1]
    G, n, M = skts.shape[0], rows.shape[0], pts.shape[0] * pts.shape[1]
    z32 = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)  # noqa: E731
    vol = ops._f32(volumes.detach(), "volumes")
    g = dict(vol=z32(G, ops.J, vol.shape[-1] if vol.dim() == 3 else vol.numel() // (G * ops.J)), scale=z32(ops.J, 3),
             w0=z32(24, ops.FEAT, 32), adj_w=z32(24, 24), b0=z32(32), w1=z32(24, 32, 32), b1=z32(24, 32), w2=z32(24, 32), b2=z32(24))
    if n > 0:
        # the kernel's row tables, with every sample its own "ray" (origin = the sample point, direction 0, depth 0): row i is
        # list entry i, all rows belong to the first ("coarse") pass
        rows32 = rows.to(torch.int32).contiguous()
        cnt = torch.zeros(8, device=dev, dtype=torch.int32)
        cnt[2] = n
        cnt[5] = n
        bits32 = bits.reshape(-1).contiguous()
        lists = torch.empty(ops.J, n, device=dev, dtype=torch.int32)
        cntb = torch.zeros(ops.J, device=dev, dtype=torch.int32)
        _hip.check(_hip.lib().danbo_train_bone_lists(_p(bits32), _p(bits32), _p(rows32), _p(cnt), 0, n, _p(lists), _p(cntb), ops._stream()),
                   "danbo_train_bone_lists")
        zeros_m = z32(M)
        ab = _hip.DanboAssignBwd()
        keep = dict(o=ops._f32(pts, "pts").reshape(M, 3), d=z32(M, 3), z=zeros_m, lab=torch.zeros(M, device=dev, dtype=torch.uint8),
                    skts=ops._f32(skts, "skts"), align=ops._f32(align, "align"), sc=ops._f32(axis_scale.detach(), "axis_scale"),
                    h=z32(n, ops.H_STRIDE), dh=ops._f32(g_h, "g_h"), dp=ops._f32(g_p, "g_p"), loss=z32(4))
        for k, v in dict(rays_o=keep["o"], rays_d=keep["d"], z_c=keep["z"], z_f=keep["z"], skts=keep["skts"], align=keep["align"],
                         axis_scale=keep["sc"], volumes=vol, row_sample=rows32, row_ray=rows32, cnt=cnt, lists=lists, cntb=cntb,
                         h_rows=keep["h"], d_h=keep["dh"], label_c=keep["lab"], label_f=keep["lab"], bits_c=bits32, bits_f=bits32,
                         w0=a["w0"], adj_w=a["adj_w"], adj=a["adj"], b0=a["b0"], w1=a["w1"], b1=a["b1"], w2=a["w2"], b2=a["b2"],
                         g_w0=g["w0"], g_adj_w=g["adj_w"], g_b0=g["b0"], g_w1=g["w1"], g_b1=g["b1"], g_w2=g["w2"], g_b2=g["b2"],
                         g_vol=g["vol"], g_scale=g["scale"], loss=keep["loss"], d_p=keep["dp"]).items():
            setattr(ab, k, v.data_ptr())
        ab.R, ab.S, ab.Sf, ab.G, ab.rows_cap, ab.c_ss = M, 1, 1, G, n, 0.0
        _hip.check(_hip.lib().danbo_assign_blend_bwd(ctypes.byref(ab), ops._stream()), "danbo_assign_blend_bwd")
    w0, adj_w, adj, b0, w1, b1, w2, b2 = params
    return [g["vol"].reshape(volumes.shape), g["scale"].reshape(axis_scale.shape), g["w0"].reshape(w0.shape), g["adj_w"].reshape(adj_w.shape),
            g["b0"].reshape(b0.shape), g["w1"].reshape(w1.shape), g["b1"].reshape(b1.shape), g["w2"].reshape(w2.shape),
            g["b2"].reshape(b2.shape)]


@assign_blend_bwd.register_fake
def _(volumes, axis_scale, pts, skts, align, rows, bits, params, g_h, g_p):
    w0, adj_w, adj, b0, w1, b1, w2, b2 = params
    return [torch.empty_like(t, dtype=torch.float32) for t in (volumes, axis_scale, w0, adj_w, b0, w1, b1, w2, b2)]


def _assign_setup(ctx, inputs, output):
    volumes, axis_scale, pts, skts, align, rows, bits, params = inputs
    ctx.save_for_backward(volumes, axis_scale, pts, skts, align, rows, bits, *params)


def _assign_backward(ctx, g_h, g_p, g_confd):
    volumes, axis_scale, pts, skts, align, rows, bits, *params = ctx.saved_tensors
    n = rows.shape[0]
    g_h = g_h if g_h is not None else pts.new_zeros(n, ops.H_STRIDE)
    g_p = g_p if g_p is not None else pts.new_zeros(n, ops.J)
    d = torch.ops.danbo.assign_blend_bwd(volumes, axis_scale, pts, skts, align, rows, bits, list(params), g_h.contiguous(), g_p.contiguous())
    d_vol, d_sc, d_w0, d_adjw, d_b0, d_w1, d_b1, d_w2, d_b2 = d
    return d_vol, d_sc, None, None, None, None, None, [d_w0, d_adjw, None, d_b0, d_w1, d_b1, d_w2, d_b2]


assign_blend.register_autograd(_assign_backward, setup_context=_assign_setup)


# ------------------------------------------------------------------------------------------------------------------- pe + MLP
# torch.ops.danbo.pe_mlp(h, row_ray, vin, <20 parameter tensors>) -> raw [n, 4]
#   h [n, 15] blended voxel features of n rows, row_ray [n] int32 ray of each row, vin [R, view_ch] per-ray view inputs
#   (PE(dir) | frame code); parameters in the reference's layouts: pts_linears.{0..7}.{weight, bias}, alpha_linear, feature_linear,
#   views_linears.0, rgb_linear.  = Embedder (cutoff_embedder.py:62-73) + NeRF.inference (core/networks/nerf.py:176-209).
#   forward: danbo_trunk_pack + danbo_train_cview + danbo_trunk_fwd;  backward: danbo_trunk_bwd + danbo_dw16 +
#   danbo_train_view_grads + danbo_train_head_chain -> d h, d vin, the gradient of every parameter.
_PE_MLP_PARAMS = ([f"pts_w{i}" for i in range(8)] + [f"pts_b{i}" for i in range(8)]
                  + ["alpha_w", "alpha_b", "feature_w", "feature_b", "views_w", "views_b", "rgb_w", "rgb_b"])


def _trunk_structs(h, row_ray, vin, params, need_bwd):
    """device buffers + the two C structs for n rows handed over one by one (R = 0 empty-space rows, S = 1: row_sample = ray)"""
    dev = h.device
    n, R = h.shape[0], vin.shape[0]
    pad = (n + 127) // 128 * 128 + 128
    f32 = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)  # noqa: E731
    buf = dict(packed=torch.empty(_hip_trunk_bytes(), device=dev, dtype=torch.uint8), wfv=f32(128 * 256), b_eff=f32(128),
               wmax=torch.zeros(16, device=dev), winv=f32(16), cnt=torch.zeros(8, device=dev, dtype=torch.int32),
               h16=torch.zeros(n, 16, device=dev), cview=f32(R, 128), y=f32(8, pad * 256), pe=f32(pad * 224),
               relu=torch.empty(8, pad * 4, device=dev, dtype=torch.int64), hv=f32(pad * 128),
               hv_bits=torch.empty(pad * 4, device=dev, dtype=torch.int32), raw_rows=f32(n, 4), raw_dense=f32(max(R, 1), 4),
               row_ray_out=torch.empty(n, device=dev, dtype=torch.int32))
    buf["h16"][:, :15] = h
    buf["cnt"][0] = n
    buf["row_sample"] = row_ray.to(torch.int32).contiguous()
    w = _hip.DanboTrunkWeights()
    for i in range(8):
        w.pts_w[i], w.pts_b[i] = params[i].data_ptr(), params[8 + i].data_ptr()
    for k, t in zip(("alpha_w", "alpha_b", "feature_w", "feature_b", "views_w", "views_b", "rgb_w", "rgb_b"), params[16:]):
        setattr(w, k, t.data_ptr())
    for k in ("packed", "wfv", "b_eff", "wmax", "winv"):
        setattr(w, k, buf[k].data_ptr())
    w.view_ch = vin.shape[1]
    r = _hip.DanboTrunkRows()
    r.cnt, r.row_sample, r.h_rows, r.cview = (buf[k].data_ptr() for k in ("cnt", "row_sample", "h16", "cview"))
    r.R, r.S, r.Sf, r.rows_cap, r.rows_pad = 0, 1, 1, n, pad
    for k in ("y", "pe", "relu", "hv", "hv_bits", "raw_rows"):
        setattr(r, k, buf[k].data_ptr())
    r.raw_c = r.raw_f = r.raw_empty = buf["raw_dense"].data_ptr()
    r.row_ray = buf["row_ray_out"].data_ptr()
    if need_bwd:
        buf.update(dz=f32(8, pad * 256), dpre_v=f32(pad * 128), d_alpha4=f32(n, 4), d_h=f32(n, 16), maxabs=torch.zeros(16, device=dev))
        for k in ("dz", "dpre_v", "d_alpha4", "d_h", "maxabs"):
            setattr(r, k, buf[k].data_ptr())
    return buf, w, r


def _hip_trunk_bytes():
    return (74 + 76) * 32768       # DANBO_TRUNK_PACKED_BYTES


def _vin_padded(vin):
    """[R, view_ch] -> row stride a multiple of 4 floats with 8 floats of slack (the view-gradient kernel reads 8 columns at a time)"""
    R, C = vin.shape
    ld = (C + 3) // 4 * 4 + 4
    out = torch.zeros(R * ld + 8, device=vin.device, dtype=torch.float32)
    out[:R * ld].view(R, ld)[:, :C] = vin
    return out, ld


@torch.library.custom_op("danbo::pe_mlp", mutates_args=())
def pe_mlp(h: torch.Tensor, row_ray: torch.Tensor, vin: torch.Tensor, params: List[torch.Tensor]) -> torch.Tensor:
    if not h.is_cuda:
        raise RuntimeError("danbo::pe_mlp runs on the HIP path only (no CPU fallback)")
    params = [p.detach().float().contiguous() for p in params]
    h, vin = h.detach().float().contiguous(), vin.detach().float().contiguous()
    buf, w, r = _trunk_structs(h, row_ray, vin, params, need_bwd=False)
    lib, st = _hip.lib(), ops._stream()
    _hip.check(lib.danbo_trunk_pack(ctypes.byref(w), st), "danbo_trunk_pack")
    vp, ld = _vin_padded(vin)
    _hip.check(lib.danbo_train_cview(_p(vp), ld, vin.shape[1], _p(params[20]), _p(buf["b_eff"]), vin.shape[0], _p(buf["cview"]), st),
               "danbo_train_cview")
    _hip.check(lib.danbo_trunk_fwd(ctypes.byref(w), ctypes.byref(r), 0, st), "danbo_trunk_fwd")
    return buf["raw_rows"]


@pe_mlp.register_fake
def _(h, row_ray, vin, params):
    return h.new_empty(h.shape[0], 4, dtype=torch.float32)


@torch.library.custom_op("danbo::pe_mlp_bwd", mutates_args=())
def pe_mlp_bwd(h: torch.Tensor, row_ray: torch.Tensor, vin: torch.Tensor, params: List[torch.Tensor],
               g_raw: torch.Tensor) -> List[torch.Tensor]:
    """-> [d h [n,15], d vin [R, view_ch], d param_0 .. d param_19]  (the forward is re-run: a custom op's saved state are tensors,
    and the activations are cheaper to recompute than to keep alive across the autograd graph)"""
    params = [p.detach().float().contiguous() for p in params]
    h, vin = h.detach().float().contiguous(), vin.detach().float().contiguous()
    n, R, C = h.shape[0], vin.shape[0], vin.shape[1]
    dev = h.device
    buf, w, r = _trunk_structs(h, row_ray, vin, params, need_bwd=True)
    lib, st = _hip.lib(), ops._stream()
    _hip.check(lib.danbo_trunk_pack(ctypes.byref(w), st), "danbo_trunk_pack")
    vp, ld = _vin_padded(vin)
    _hip.check(lib.danbo_train_cview(_p(vp), ld, C, _p(params[20]), _p(buf["b_eff"]), R, _p(buf["cview"]), st), "danbo_train_cview")
    _hip.check(lib.danbo_trunk_fwd(ctypes.byref(w), ctypes.byref(r), 0, st), "danbo_trunk_fwd")
    buf["cnt"][4] = n                                  # total rows (the step's second pass would have set it)
    # ---- input-gradient chain on d raw per row
    d_raw_rows = torch.zeros(n, 4, device=dev)
    d_raw_rows.copy_(g_raw.float())
    r.d_raw_rows = d_raw_rows.data_ptr()
    r.d_raw_c = r.d_raw_f = None
    _hip.check(lib.danbo_trunk_bwd(ctypes.byref(w), ctypes.byref(r), st), "danbo_trunk_bwd")
    # ---- weight gradients (k_dw16): the layer list of the training step (csrc/k_train.hip describe_dw)
    g = [torch.zeros_like(p) for p in params]
    g_wfv, g_beff = torch.zeros(128, 256, device=dev), torch.zeros(128, device=dev)
    pad = buf["y"].shape[1] // 256
    D = _hip.DanboDwLayer
    y_of = lambda l: buf["y"][l].data_ptr()       # noqa: E731
    dz_of = lambda l: buf["dz"][l].data_ptr()     # noqa: E731
    mx = buf["maxabs"]
    layers = []
    for l in range(8):
        common = dict(dy=dz_of(l), ldy=256, N=256, dy_maxabs=mx[l:].data_ptr(), gw=g[l].data_ptr(), frag=3)
        if l in (0, 5):
            layers.append(D(x1=buf["pe"].data_ptr(), ld1=224, K1=224, x1_pe=1, gw_ld=195 if l == 0 else 451, gw_col0=0, gb=g[8 + l].data_ptr(),
                            **common))
            if l == 5:
                layers.append(D(x1=y_of(4), ld1=256, K1=256, gw_ld=451, gw_col0=195, gb=None, **common))
        else:
            layers.append(D(x1=y_of(l - 1), ld1=256, K1=256, gb=g[8 + l].data_ptr(), **common))
    layers.append(D(dy=buf["dpre_v"].data_ptr(), ldy=128, N=128, dy_maxabs=mx[8:].data_ptr(), frag=3, x1=y_of(7), ld1=256, K1=256,
                    gw=g_wfv.data_ptr(), gb=g_beff.data_ptr()))
    layers.append(D(dy=buf["d_alpha4"].data_ptr(), ldy=4, N=1, dy_maxabs=mx[9:].data_ptr(), frag=2, x1=y_of(7), ld1=256, K1=256,
                    gw=g[16].data_ptr(), gb=g[17].data_ptr()))
    mraw = g_raw.detach().abs().max().reshape(1).float().contiguous()
    layers.append(D(dy=d_raw_rows.data_ptr(), ldy=4, N=3, dy_maxabs=mraw.data_ptr(), frag=2, x1=buf["hv"].data_ptr(), ld1=128, K1=128,
                    gw=g[22].data_ptr(), gb=g[23].data_ptr()))
    L = (D * len(layers))(*layers)
    slices = 8
    scratch = torch.empty(lib.danbo_dw16_scratch_floats(L, len(layers), slices), device=dev)
    _hip.check(lib.danbo_dw16(L, len(layers), n, _p(buf["cnt"][4:]), slices, _p(scratch), st), "danbo_dw16")
    # ---- per-ray view gradients, then the chain rule of the merged feature / view layer
    d_cview = torch.zeros(R, 128, device=dev)
    _hip.check(lib.danbo_train_view_grads(_p(buf["dpre_v"]), _p(buf["row_ray_out"]), _p(buf["cnt"]), n, R, _p(vp), ld, C, None, 0, _p(d_cview),
                                          None, _p(g[20]), None, st), "danbo_train_view_grads")
    _hip.check(lib.danbo_train_head_chain(_p(g_wfv), _p(g_beff), None, _p(params[18]), _p(params[19]), _p(params[20]), C, 0, 0, 0,
                                          _p(g[18]), _p(g[19]), _p(g[20]), _p(g[21]), None, None, st), "danbo_train_head_chain")
    d_vin = ops.small_matmul(d_cview, params[20][:, 256:])     # [R,128] x [128, view_ch]: per RAY, tiny (danbo_small_matmul)
    return [buf["d_h"][:, :15].contiguous(), d_vin] + g


@pe_mlp_bwd.register_fake
def _(h, row_ray, vin, params, g_raw):
    return [h.new_empty(h.shape[0], 15), vin.new_empty(vin.shape)] + [p.new_empty(p.shape) for p in params]


def _pe_mlp_setup(ctx, inputs, output):
    h, row_ray, vin, params = inputs
    ctx.save_for_backward(h, row_ray, vin, *params)


def _pe_mlp_backward(ctx, g_raw):
    h, row_ray, vin, *params = ctx.saved_tensors
    out = torch.ops.danbo.pe_mlp_bwd(h, row_ray, vin, list(params), g_raw.contiguous())
    return out[0], None, out[1], out[2:]


pe_mlp.register_autograd(_pe_mlp_backward, setup_context=_pe_mlp_setup)


# ------------------------------------------------------------------------------------------------- A-NeRF cutoff PE + MLP (fwd)
# torch.ops.danbo.anerf_cutoff_pe_mlp(pts [R,S,3], rays_d [R,3], skts [G,24,4,4], align [24,4,4], cam_idx [R] int64 or None,
#                                     params, tau, multires, multires_views) -> raw [R,S,4]
# params: pts_linears.{0..D-1}.weight, pts_linears.{0..D-1}.bias, alpha_linear.{weight,bias}, feature_linear.{weight,bias},
#         views_linears.0.{weight,bias}, rgb_linear.{weight,bias}, pe_fn.cutoff_dist [24], dirs_pe_fn.cutoff_dist [24]
#         (+ framecodes.codes.weight when the model has frame codes) -- the reference's state_dict tensors, skip after layer 4.
# The pair SURVEY 8(b) lists without a backward (`anerf_cutoff_pe_mlp_fwd`): k_anerf_encode -> k_linear16 x (D + 1) ->
# k_anerf_color through core/anerf_engine.AnerfEngine, whose packed weights are cached per parameter storage / version.
# A-NeRF TRAINING differentiates core/train_path.forward_train_anerf instead (HIP encoders, dense layers recorded by autograd).
_ANERF_ENGINES = {}


def _anerf_engine(params, align, tau, multires, multires_views):
    from .anerf_engine import AnerfEngine
    n = len(params)
    coded = n % 2 == 1                                  # 2 D + 10 tensors, + 1 with frame codes
    D = (n - 10 - (1 if coded else 0)) // 2
    if D < 5 or 2 * D + 10 + (1 if coded else 0) != n:
        raise ValueError(f"danbo::anerf_cutoff_pe_mlp: {n} parameter tensors do not form D weights + D biases + 10 (+ codes)")
    names = ([f"pts_linears.{i}.weight" for i in range(D)] + [f"pts_linears.{i}.bias" for i in range(D)]
             + ["alpha_linear.weight", "alpha_linear.bias", "feature_linear.weight", "feature_linear.bias", "views_linears.0.weight",
                "views_linears.0.bias", "rgb_linear.weight", "rgb_linear.bias", "pe_fn.cutoff_dist", "dirs_pe_fn.cutoff_dist"]
             + (["framecodes.codes.weight"] if coded else []))
    p = {k: ops._f32(v.detach(), k) for k, v in zip(names, params)}
    p["pe_fn.tau"] = p["dirs_pe_fn.tau"] = torch.tensor(float(tau))
    key = (align.data_ptr(), float(tau), int(multires), int(multires_views)) + tuple(v.data_ptr() for v in params)
    eng = _ANERF_ENGINES.get(key)
    if eng is None:
        if len(_ANERF_ENGINES) >= 4:
            _ANERF_ENGINES.pop(next(iter(_ANERF_ENGINES)))
        W = p["pts_linears.1.weight"].shape[0]
        cfg = dict(nerf_type="nerf", W=W, D=D, skips=(4,), view_W=p["views_linears.0.weight"].shape[0], multires=int(multires),
                   multires_views=int(multires_views), use_framecode=coded)
        eng = _ANERF_ENGINES[key] = AnerfEngine(cfg, p, ops._f32(align, "align"))
    else:
        eng.p = p                                       # same storages: refresh() re-packs only when a version counter moved
    return eng


@torch.library.custom_op("danbo::anerf_cutoff_pe_mlp", mutates_args=())
def anerf_cutoff_pe_mlp(pts: torch.Tensor, rays_d: torch.Tensor, skts: torch.Tensor, align: torch.Tensor,
                        cam_idx: Optional[torch.Tensor], params: List[torch.Tensor], tau: float, multires: int,
                        multires_views: int) -> torch.Tensor:
    if not pts.is_cuda:
        raise RuntimeError("danbo::anerf_cutoff_pe_mlp runs on the HIP path only (no CPU fallback)")
    eng = _anerf_engine(params, align, tau, multires, multires_views)
    R = pts.shape[0]
    return eng.forward_samples(None, ops._f32(rays_d.reshape(R, 3), "rays_d"), ops._f32(skts, "skts"), cam_idx, pts=ops._f32(pts, "pts")).clone()


@anerf_cutoff_pe_mlp.register_fake
def _(pts, rays_d, skts, align, cam_idx, params, tau, multires, multires_views):
    return pts.new_empty(pts.shape[0], pts.shape[1], 4, dtype=torch.float32)
