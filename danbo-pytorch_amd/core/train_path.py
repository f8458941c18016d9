"""Differentiable (training) forward of DANBO / A-NeRF on the MI355X path: the AUTOGRAD path (`caster.train(); caster(...)` ->
predictions with `grad_fn`, what the reference's trainer expects; the fused step of core/train_engine.py is the default for
the shipped DANBO configurations).

Every stage runs on the library's kernels in both directions, as torch.library custom operators (core/custom_ops.py):
  torch.ops.danbo.pose_volumes   rot6d + PE + skeleton GNN -> per-bone volumes            (k_pose_layer / k_pose_*_bwd)
  torch.ops.danbo.assign_blend   gather + assignment GNN + masked sigmoid + blend          (k_assign_blend / k_assign_bwd)
  torch.ops.danbo.pe_mlp         voxel PE + density trunk + colour head                     (k_train_mlp_fwd / _bwd, k_dw16)
  torch.ops.danbo.composite      NeRF.raw2outputs                                           (k_composite / k_composite_bwd)
  Linear16Fn                     A-NeRF's dense layers                                      (k_linear16 both ways, k_dw16)
  AnerfViewConstsFn, AnerfColorFn  A-NeRF's per-ray view constants, cutoff-weighted view sum + colour head (k_anerf_train.hip)
and the in-volume cull, sampling, importance sampling and merge order without gradient (the reference detaches them).
What torch computes here: element-wise glue (no GEMM: A-NeRF's heads and per-ray view products, library routes until round 5, are
`Linear16Fn` on stacked / sliced weights, `AnerfViewConstsFn` and `AnerfColorFn`).  A network of ANOTHER shape than the shipped ones is
not evaluated by a library fallback:
`forward_train` raises (the constructors raise for those shapes already).  The layer-by-layer torch restatement the operators are
tested against lives with the tests (tests/torch_layerwise.py).

Samples outside every bone volume share, per ray, one "empty-space" MLP evaluation (h = 0), which keeps the gradient path of
the reference (every sample reaches the MLP weights) at a fraction of the rows.
Gradient semantics follow the reference: `window` is detached and `invalid` is not differentiable
(core/networks/gnn_backbone.py:802-808); sample depths are detached (core/utils/ray_utils.py:287); no gradient reaches
pts / skts / bones (opt_pose is off).
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _hip
from . import custom_ops  # noqa: F401  (registers torch.ops.danbo.*)
from . import hip_ops as ops


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------------------
# custom autograd functions around the HIP kernels
# --------------------------------------------------------------------------------------
class GatherFn(torch.autograd.Function):
    """part_feat [n,24,15] = factorised gather of `volumes` at the listed samples (K1b)."""

    @staticmethod
    def forward(ctx, volumes, axis_scale, geo, rows):
        geo.axis_scale = axis_scale.detach().float().contiguous()
        n = rows.shape[0]
        out = ops.bone_gather(geo, volumes.detach().contiguous(), rows, None, n)
        ctx.geo, ctx.n = geo, n
        ctx.save_for_backward(volumes.detach(), geo.axis_scale, rows)
        return out

    @staticmethod
    def backward(ctx, g):
        volumes, axis_scale, rows = ctx.saved_tensors
        geo = ctx.geo
        # explicit shapes: zeros_like would inherit the permuted strides of an einsum output
        d_vol = torch.zeros(volumes.shape, dtype=torch.float32, device=volumes.device)
        d_sc = torch.zeros(axis_scale.shape, dtype=torch.float32, device=volumes.device)
        g = g.contiguous().float()
        if ctx.n > 0:
            _hip.check(_hip.lib().danbo_bone_gather_bwd(
                _p(geo.rays_o), _p(geo.rays_d), _p(geo.z), _p(geo.pts), geo.R, geo.S, geo.G, _p(geo.skts),
                _p(geo.align), _p(axis_scale), _p(volumes.contiguous()), _p(rows), ctx.n, _p(g), _p(d_vol), _p(d_sc),
                ops._stream()), "danbo_bone_gather_bwd")
        return d_vol, d_sc, None, None


def _packed(weight, K1=None, transposed=False, cols=None):
    """fragment packing of a dense-layer weight for Linear16Fn; cols = (c0, c1): of the column slice weight[:, c0:c1].  Cached ON
    the parameter object, per version counter (an optimizer step re-packs; a new tensor at a recycled address cannot hit)"""
    cache = getattr(weight, "_danbo_packs", None)
    if cache is None or cache[0] != weight._version:
        cache = (weight._version, {})
        try:
            weight._danbo_packs = cache
        except AttributeError:
            pass
    key = (K1, transposed, cols)
    hit = cache[1].get(key)
    if hit is None:
        w = weight.detach()
        if cols is not None:
            w = w[:, cols[0]:cols[1]]
        hit = cache[1][key] = ops.linear16_pack(w, K1=K1, transposed=transposed)
    return hit


class Linear16Fn(torch.autograd.Function):
    """y = act([x1 | x2] W^T + b) on k_linear16 (fp16 hi/lo-split MFMA products, fp32 accumulate -- the kernel of A-NeRF's eval
    trunk) with a hand-written backward on the same family: dX = dZ W through the transposed packing of the weight
    (danbo_linear16_fwd on W^T), dW = dZ^T X and db through danbo_dw16.  Gradients are ~1e-6 .. 1e-9: dZ is multiplied by the
    power of two that brings its largest entry to ~1 before the split (fp16's lo halves would otherwise be subnormal) and the
    product by the exact inverse.  x1 carries no gradient when `x1_grad` is False (the network input); x2: optional second
    input of a skip layer.  Reference: nn.Linear + F.relu of NeRF.inference (core/networks/nerf.py:176-209)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, relu, x1_grad):
        x1 = x1.contiguous()
        K1 = x1.shape[1]
        K2 = 0 if x2 is None else x2.shape[1]
        if x2 is not None:
            x2 = x2.contiguous()
        packed, shape = _packed(weight, K1=K1 if K2 else None)
        y = ops.linear16(x1, packed, shape, bias.detach() if bias is not None else None, relu=relu, x2=x2)
        ctx.relu, ctx.x1_grad, ctx.K1, ctx.K2 = relu, x1_grad, K1, K2
        ctx.save_for_backward(x1, x2 if x2 is not None else x1.new_empty(0), weight, y if relu else x1.new_empty(0))
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        import ctypes
        x1, x2, weight, y = ctx.saved_tensors
        K1, K2 = ctx.K1, ctx.K2
        N, M = weight.shape[0], x1.shape[0]
        dz = (g * (y > 0)) if ctx.relu else g
        dz = dz.contiguous().float()
        mx = dz.abs().max().reshape(1).clamp_min(torch.finfo(torch.float32).tiny)
        scale = torch.exp2(-torch.floor(torch.log2(mx)))                      # power of two: exact both ways
        dzs = dz * scale
        dx1 = dx2 = None
        if ctx.x1_grad:
            packed_t, shape_t = _packed(weight, transposed=True, cols=(0, K1) if K2 else None)
            dx1 = ops.linear16(dzs, packed_t, shape_t, None) / scale
        if K2:
            packed_t, shape_t = _packed(weight, transposed=True, cols=(K1, K1 + K2))
            dx2 = ops.linear16(dzs, packed_t, shape_t, None) / scale
        gw = torch.empty(N, K1 + K2, device=dz.device, dtype=torch.float32)
        gb = torch.empty(N, device=dz.device, dtype=torch.float32)
        D = _hip.DanboDwLayer
        common = dict(dy=dz.data_ptr(), x2=None, dy_maxabs=mx.data_ptr(), gw=gw.data_ptr(), gw2=None, gb2=None, ldy=N, ld2=0, N=N, K2=0,
                      split_n=0, frag=0, gw_ld=K1 + K2, x1_pe=0)
        layers = [D(x1=x1.data_ptr(), ld1=K1, K1=K1, gw_col0=0, gb=gb.data_ptr(), **common)]
        if K2:      # the second input as a layer of its own writing its column range of the same weight gradient
            layers.append(D(x1=x2.data_ptr(), ld1=K2, K1=K2, gw_col0=K1, gb=None, **common))
        L = (D * len(layers))(*layers)
        slices = 8
        lib = _hip.lib()
        scratch = torch.empty(lib.danbo_dw16_scratch_floats(L, len(layers), slices), device=dz.device)
        _hip.check(lib.danbo_dw16(L, len(layers), M, None, slices, _p(scratch), ops._stream()), "danbo_dw16")
        return dx1, dx2, gw, (gb if ctx.has_bias else None), None, None


def _linear16_ok(layer, K):
    """k_linear16 / k_dw16 take 16-byte aligned rows: input and output widths multiples of 4, N <= 512"""
    N = layer.weight.shape[0]
    return layer.weight.is_cuda and K % 4 == 0 and N % 4 == 0 and 4 <= N <= 512 and layer.weight.shape[1] == K


def linear16(layer, x, relu=False, x2=None, x_grad=True):
    """layer([x | x2]) (+ ReLU) on the HIP kernels both ways; a shape they do not take (widths not multiples of 4, N > 512) raises"""
    K = x.shape[1] + (0 if x2 is None else x2.shape[1])
    if not (_linear16_ok(layer, K) and x.shape[1] % 4 == 0):
        raise NotImplementedError(f"k_linear16 / k_dw16 take 16-byte aligned rows, N <= 512: layer {tuple(layer.weight.shape)} on K = {K}")
    if x.shape[0] == 0:
        return x.new_zeros(0, layer.weight.shape[0]) + 0.0 * (layer.weight.sum() + layer.bias.sum())     # keeps the graph connected
    return Linear16Fn.apply(x, x2, layer.weight, layer.bias, relu, x_grad)


class AnerfViewConstsFn(torch.autograd.Function):
    """C [24, R, VW]: the per-ray, per-joint part of A-NeRF's view layer, views_linears.0[:, view columns of joint j] . PE(unit local
    ray direction) (reference nerf.py:252-279 + cutoff_embedder.py:156-166 before the per-sample cutoff weight) on
    danbo_anerf_view_consts_fwd; backward: the view columns' gradient on danbo_anerf_view_consts_bwd (ray slices added in a fixed
    order).  No gradient reaches the ray directions or the skeleton (opt_pose is off)."""

    @staticmethod
    def forward(ctx, views_w, rays_d, skts, col0, Lv):
        wj = ops.anerf_view_wj(views_w.detach().contiguous(), col0, Lv)
        ctx.save_for_backward(rays_d, skts)
        ctx.col0, ctx.Lv, ctx.shape = col0, Lv, tuple(views_w.shape)
        return ops.anerf_view_consts(rays_d, skts, Lv, wj)

    @staticmethod
    def backward(ctx, dC):
        rays_d, skts = ctx.saved_tensors
        g = torch.zeros(ctx.shape, device=dC.device, dtype=torch.float32)
        ops.anerf_view_consts_bwd(rays_d, skts, ctx.Lv, dC.contiguous().float(), g, ctx.col0)
        return g, None, None, None, None


class AnerfColorFn(torch.autograd.Function):
    """raw [R, S, 4] = (rgb_linear(relu(featv + table_ray[ray] + sum_j w_j C[j, ray])), alpha): the view layer's cutoff-weighted sum,
    its ReLU and the colour head (reference nerf.py:196-209) on danbo_anerf_color_train_fwd / danbo_anerf_color_bwd -- one wavefront
    per ray both ways.  The cutoff weights w carry no gradient (the encoders have no trainable parameter)."""

    @staticmethod
    def forward(ctx, featv, alpha, C, table_ray, rgb_w, rgb_b, w, R, S):
        VW = featv.shape[1]
        n = R * S
        head = torch.empty(n, VW + 4, device=featv.device, dtype=torch.float32)
        head[:, :VW] = featv
        head[:, VW] = alpha.reshape(-1)
        hv = torch.empty(n, VW, device=featv.device, dtype=torch.float32)
        raw = torch.empty(R, S, 4, device=featv.device, dtype=torch.float32)
        lib = _hip.lib()
        _hip.check(lib.danbo_anerf_color_train_fwd(_p(head), VW + 4, _p(w), _p(C.contiguous()), _p(table_ray.contiguous()), R, R, S, VW,
                                                   _p(rgb_w.detach().contiguous()), _p(rgb_b.detach().contiguous()), head[:, VW:].data_ptr(), VW + 4,
                                                   _p(hv), _p(raw), ops._stream()), "danbo_anerf_color_train_fwd")
        ctx.save_for_backward(hv, w, rgb_w.detach().contiguous())
        ctx.R, ctx.S, ctx.VW = R, S, VW
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        hv, w, rgb_w = ctx.saved_tensors
        R, S, VW = ctx.R, ctx.S, ctx.VW
        n = R * S
        dev = d_raw.device
        d_raw = d_raw.contiguous().float()
        mx = d_raw.abs().max().reshape(1)
        sig = torch.empty(1, device=dev)
        d_featv = torch.empty(n, VW, device=dev)
        d_alpha4 = torch.empty(n, 4, device=dev)
        dC = torch.empty(24, R, VW, device=dev)
        d_pre_ray = torch.empty(R, VW, device=dev)
        lib = _hip.lib()
        nf = lib.danbo_anerf_color_bwd_part_floats(R, VW)
        part = torch.empty(nf, device=dev)
        _hip.check(lib.danbo_anerf_color_bwd(_p(d_raw), _p(hv), _p(w), R, R, S, VW, _p(rgb_w), _p(mx), _p(sig), _p(d_featv), _p(d_alpha4), 4,
                                             _p(dC), _p(d_pre_ray), 0, _p(part), ops._stream()), "danbo_anerf_color_bwd")
        g_rgb_w, g_rgb_b = torch.empty(3, VW, device=dev), torch.empty(3, device=dev)
        _hip.check(lib.danbo_anerf_rgb_reduce(_p(part), nf, VW, _p(g_rgb_w), _p(g_rgb_b), ops._stream()), "danbo_anerf_rgb_reduce")
        return d_featv / sig, (d_alpha4[:, 0] / sig).reshape(n, 1), dC, d_pre_ray, g_rgb_w, g_rgb_b, None, None, None


def composite(raw, z, rays_d, B=1.0, noise=None):
    """NeRF.raw2outputs with gradients for rgb_map and acc_map (K4): torch.ops.danbo.composite (core/custom_ops.py)"""
    rgb, disp, acc, w, al = torch.ops.danbo.composite(raw.contiguous().float(), z.contiguous().float(),
                                                      rays_d.reshape(-1, 3).contiguous().float(), float(B), noise)
    return dict(rgb_map=rgb, disp_map=disp, acc_map=acc, weights=w, alpha=al)


# --------------------------------------------------------------------------------------
# small differentiable pieces (element-wise glue)
# --------------------------------------------------------------------------------------
_PE_FREQS = {}


def positional_encoding(x, L):
    """[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...] (reference cutoff_embedder.py:62-73) in five tensor ops instead
    of 3 L + 1: the training step is launch-bound, and the scaling by powers of two is exact either way"""
    if L == 0:
        return x
    key = (L, x.device, x.dtype)
    if key not in _PE_FREQS:
        _PE_FREQS[key] = torch.tensor([float(2 ** l) for l in range(L)], device=x.device, dtype=x.dtype)
    freqs = _PE_FREQS[key]
    xf = x.unsqueeze(-2) * freqs[:, None]                                   # [..., L, d]
    sc = torch.stack([torch.sin(xf), torch.cos(xf)], -2)                   # [..., L, 2, d]
    return torch.cat([x, sc.flatten(-3)], -1)


def axis_angle_to_rot6d(aa):
    """pytorch3d axis_angle_to_matrix (via quaternion, Taylor branch < 1e-6) -> first two columns."""
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = ang * 0.5
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    q = torch.cat([torch.cos(half), aa * s], -1)
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    return torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * j + k * r),
                        1 - two_s * (i * i + k * k), two_s * (i * k - j * r), two_s * (j * k + i * r)], -1)


def _pose_op_params(model):
    """the 12 tensors of torch.ops.danbo.pose_volumes if graph_net has the structure k_pose_layer is built for (two graph
    convolutions, two per-bone linears, 240 outputs, width <= 256: every shipped DANBO config), else None"""
    gn = model.graph_net
    try:
        l0, l1, l2, l3 = gn.layers
        ok = (hasattr(l0, "adj_w") and hasattr(l1, "adj_w") and not hasattr(l2, "adj_w") and not hasattr(l3, "adj_w")
              and l3.weight.shape[-1] == ops.VOL and l1.lin.weight.shape[-1] <= 256
              and l0.lin.weight.shape[1] == 6 * (1 + 2 * model.graph_pe_fn.num_freqs))
    except (ValueError, AttributeError):
        return None
    if not ok:
        return None
    return [l0.lin.weight, l0.adj_w, l0.adj, l0.bias, l1.lin.weight, l1.adj_w, l1.adj, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias]


def pose_volumes(model, bones_g):
    """FactorizeGNN forward (reference gnn_backbone.py:683-704) -> [G,24,240], differentiable: torch.ops.danbo.pose_volumes
    (k_pose_layer both ways, core/custom_ops.py)"""
    params = _pose_op_params(model)
    if params is None:
        raise NotImplementedError("the pose-GNN kernels cover two graph convolutions + two per-bone linears -> 240 outputs, width <= 256 "
                                  "(every shipped DANBO config); this graph_net has another structure")
    return torch.ops.danbo.pose_volumes(bones_g.contiguous().float(), int(model.graph_pe_fn.num_freqs), params)[0]


def _assign_op_applies(model):
    """torch.ops.danbo.assign_blend covers the shipped assignment net (MixGNN: one graph convolution 15 -> 32, two per-bone
    linears 32 -> 32 -> 1)"""
    try:
        l0, l1, l2 = model.prob_linears.layers
        return (tuple(l0.lin.weight.shape) == (24, ops.FEAT, 32) and tuple(l1.weight.shape) == (24, 32, 32)
                and tuple(l2.weight.shape) == (24, 32, 1))
    except (ValueError, AttributeError):
        return False


def _fused_mlp_params(model):
    """the 24 parameter tensors of torch.ops.danbo.pe_mlp if the network has the shape the fused trunk kernels are built for
    (D = 8, W = 256, skip after layer 4, 6 voxel octaves, view_W = 128), else None (forward_train raises)"""
    try:
        ok = (len(model.pts_linears) == 8 and list(model.skips) == [4] and model.voxel_pe_fn.num_freqs == 6
              and model.pts_linears[0].weight.shape == (256, 195) and model.views_linears[0].weight.shape[0] == 128
              and model.feature_linear.weight.shape == (256, 256) and model.views_linears[0].weight.shape[1] <= 256 + 160)
    except AttributeError:
        return None
    if not ok:
        return None
    from . import custom_ops  # noqa: F401  (registers the operator)
    return ([l.weight for l in model.pts_linears] + [l.bias for l in model.pts_linears]
            + [model.alpha_linear.weight, model.alpha_linear.bias, model.feature_linear.weight, model.feature_linear.bias,
               model.views_linears[0].weight, model.views_linears[0].bias, model.rgb_linear.weight, model.rgb_linear.bias])


def view_inputs(model, rays_d, skts_g, cam_idxs, rays_per_pose):
    """per-ray [PE(dir) | frame code] (reference nerf.py:252-279, encoders.py:179-189,570-578)."""
    name = model.pts_embedder.ray_tr_fn.encoder_name
    d = rays_d
    if name == "RLEncoder":
        pose = torch.arange(rays_d.shape[0], device=rays_d.device) // rays_per_pose
        d = torch.einsum("rij,rj->ri", skts_g[pose, 0, :3, :3], rays_d)
    if model.pts_embedder.view_input_fn.encoder_name == "VecNorm":
        d = F.normalize(d, dim=-1, p=2)
    v = positional_encoding(d, model.dirs_pe_fn.num_freqs)
    if model.use_framecode:
        idx = cam_idxs.reshape(-1).long()
        if (not model.training) and int(idx.max()) < 0:
            code = model.framecodes.codes.weight.mean(0, keepdim=True).expand(idx.shape[0], -1)
        else:
            code = model.framecodes.codes(idx)
        v = torch.cat([v, code], -1)
    return v


# --------------------------------------------------------------------------------------
# DANBO.forward in training mode
# --------------------------------------------------------------------------------------
def forward_train(model, inputs):
    """-> raw [R,S,4] (differentiable), encoded {confd [R,S,24], part_invalid [R,S,24]}"""
    pts = inputs["pts"].contiguous().float()
    R, S = pts.shape[:2]
    G = int(inputs.get("N_uniques", 1))
    skts, bones = inputs["skts"], inputs["bones"]
    skts_g = (skts if skts.shape[0] == G else skts[:: max(skts.shape[0] // G, 1)]).contiguous().float()
    bones_g = (bones if bones.shape[0] == G else bones[:: max(bones.shape[0] // G, 1)]).contiguous().float()
    align = inputs["align_transforms"].reshape(-1, 24, 4, 4)[0].contiguous().float().to(pts.device)
    rays_d = inputs["rays_d"].reshape(R, 3).contiguous().float()
    axis_scale = model.graph_net.axis_scale

    geo = ops.Geometry(rays_d, rays_d, skts_g, align, axis_scale.detach(), pts=pts)
    bits, lst, cnt = ops.bone_cull(geo, compact=True)
    n = int(cnt.item())                      # one host sync per pass: sizes the autograd graph
    rows = torch.sort(lst[:n]).values.contiguous()
    shared = inputs.get("shared")
    shared = shared if shared is not None else {}
    if "vols" not in shared:
        shared["vols"] = pose_volumes(model, bones_g)
    vols = shared["vols"]
    shifts = torch.arange(24, device=pts.device, dtype=torch.int32)
    fused = _fused_mlp_params(model)
    if not _assign_op_applies(model) or fused is None:
        raise NotImplementedError("the autograd path evaluates the shipped network shapes on the HIP operators (assignment net 15 -> 32 "
                                  "-> 32 -> 1; D = 8, W = 256, skip after layer 4, 6 voxel octaves, view_W = 128); there is no library "
                                  "fallback for another shape")
    # torch.ops.danbo.assign_blend: gather + assignment GNN + masked sigmoid + blend in one HIP kernel each way (core/custom_ops.py);
    # part_feat [n,24,15] is never materialised
    l0, l1, l2 = model.prob_linears.layers
    if n > 0:
        h16, p_rows, logits = torch.ops.danbo.assign_blend(vols, axis_scale, pts, skts_g, align, rows, bits,
                                                           [l0.lin.weight, l0.adj_w, l0.adj, l0.bias, l1.weight, l1.bias, l2.weight, l2.bias])
        h = h16[:, :15]
    else:     # no sample of the batch inside a volume: only the per-ray empty-space rows below reach the MLP
        h = pts.new_zeros(0, 15)
        p_rows = logits = pts.new_zeros(0, 24)
    if "vin" not in shared:
        shared["vin"] = view_inputs(model, rays_d, skts_g, inputs.get("cam_idxs"), R // G)
    vin = shared["vin"]
    ray_of_row = (rows // S).long()
    # in-volume rows and the one empty-space row per ray through torch.ops.danbo.pe_mlp (encoding + trunk + heads on the fused HIP
    # kernels) in ONE batch; a caller that runs several passes over the same rays -- coarse and importance samples -- shares the
    # empty-space rows
    if "raw_empty" in shared:
        raw_empty = shared["raw_empty"]
        raw_rows = torch.ops.danbo.pe_mlp(h, ray_of_row.int(), vin, fused) if n > 0 else h.new_zeros(0, 4)
    else:
        raw_both = torch.ops.danbo.pe_mlp(torch.cat([h, h.new_zeros(R, h.shape[1])], 0),
                                          torch.cat([ray_of_row.int(), torch.arange(R, device=pts.device, dtype=torch.int32)]), vin, fused)
        raw_rows, raw_empty = raw_both[:n], raw_both[n:]
        shared["raw_empty"] = raw_empty
    raw = raw_empty[:, None, :].expand(R, S, 4).reshape(R * S, 4).index_copy(0, rows.long(), raw_rows)
    confd = torch.zeros(R * S, 24, device=pts.device).index_copy(0, rows.long(), logits)
    # the differentiable route to the assignment net for the soft-softmax loss (the operator's confd carries no gradient)
    p_valid = torch.zeros(R * S, 24, device=pts.device).index_copy(0, rows.long(), p_rows).reshape(R, S, 24)
    # confd of samples outside every volume: the reference evaluates the assignment net there too; those
    # logits never reach a loss (they are multiplied by part_valid = 0), so they are left at zero
    all_valid = ((bits.unsqueeze(-1) >> shifts) & 1).float()
    encoded = dict(confd=confd.reshape(R, S, 24), part_invalid=(1.0 - all_valid).reshape(R, S, 24))
    encoded["p_valid"] = p_valid
    return raw.reshape(R, S, 4), encoded


def forward_train_anerf(model, inputs):
    """A-NeRF (nerf_type = nerf) training forward -> raw [R,S,4] (differentiable), {}.

    The encoders have no trainable parameter (`cutoff_dist` is `requires_grad=False`, core/cutoff_embedder.py:139) and
    no gradient flows to the sample positions (`z_samples` detached, core/utils/ray_utils.py:287), so the HIP encode
    kernels run as they do in evaluation; every dense layer is `Linear16Fn` (k_linear16 both ways, k_dw16): the trunk, feature_linear
    and alpha_linear stacked as ONE W + 1 wide layer, views_linears.0's feature columns, its frame-code columns on the rays' code rows.
    The view layer uses the factorisation of the eval kernels: per-ray, per-joint products of views_linears.0 with the direction
    encoding (`AnerfViewConstsFn`), then the 24-term cutoff-weighted sum, the ReLU and rgb_linear per sample (`AnerfColorFn`).
    No torch / rocBLAS GEMM: the library route of rounds 1 - 5 (alpha_linear, rgb_linear, the view products) is gone.  (The fused
    step, core/anerf_train_engine.py, is what Trainer.train_batch runs; this is `caster.train(); caster(...)` for a caller that
    wants predictions with a grad_fn.)"""
    pts = inputs["pts"].contiguous().float()
    R, S = pts.shape[:2]
    G = int(inputs.get("N_uniques", 1))
    skts = inputs["skts"]
    skts_g = (skts if skts.shape[0] == G else skts[:: max(skts.shape[0] // G, 1)]).contiguous().float()
    align = inputs["align_transforms"].reshape(-1, 24, 4, 4)[0].contiguous().float().to(pts.device)
    rays_d = inputs["rays_d"].reshape(R, 3).contiguous().float()
    L, Lv = model.pe_fn.num_freqs, model.dirs_pe_fn.num_freqs
    tau = float(model.pe_fn.tau)
    W = model.W
    if model.feature_linear.weight.shape[0] != W or len(model.skips) > 1:
        raise NotImplementedError("the A-NeRF training kernels take feature_linear of width W (the reference: 2 view_W = W) and one skip")
    with torch.no_grad():
        x0, w = ops.anerf_encode(None, None, skts_g, align, model.pe_fn.cutoff_dist.detach(), tau, L, 0, R * S, pts=pts)
    h, skip_in = x0, None
    for i, l in enumerate(model.pts_linears):
        if skip_in is not None:       # the layer behind a skip: [input | h] as two operands, no concatenated copy
            h = linear16(l, x0, relu=True, x2=h, x_grad=False)
            skip_in = None
        else:
            h = linear16(l, h, relu=True, x_grad=i > 0)
        if i in model.skips:
            skip_in = h
    if skip_in is not None:
        raise NotImplementedError("a skip after the last trunk layer")
    # [feature_linear ; alpha_linear ; 0 0 0] as one layer of W + 4 outputs (16-byte rows): column W is the density logit
    pad = h.new_zeros(3, W)
    w_head = torch.cat([model.feature_linear.weight, model.alpha_linear.weight, pad], 0)
    b_head = torch.cat([model.feature_linear.bias, model.alpha_linear.bias, h.new_zeros(3)])
    head = Linear16Fn.apply(h, None, w_head, b_head, False, True)                  # [n, W + 4]
    wv, bv = model.views_linears[0].weight, model.views_linears[0].bias           # [VW, W + 72 nb + code]
    view_ch = 72 * (1 + 2 * Lv)
    featv = Linear16Fn.apply(head[:, :W], None, wv[:, :W].contiguous(), None, False, True)
    C = AnerfViewConstsFn.apply(wv, rays_d, skts_g, W, Lv)                          # [24, R, VW]
    if model.use_framecode:
        idx = inputs.get("cam_idxs").reshape(-1).long()
        table_ray = Linear16Fn.apply(model.framecodes.codes(idx), None, wv[:, W + view_ch:].contiguous(), bv, False, True)   # [R, VW]
    else:
        table_ray = bv[None, :].expand(R, -1)
    return AnerfColorFn.apply(featv, head[:, W:W + 1], C, table_ray, model.rgb_linear.weight, model.rgb_linear.bias, w, R, S), {}


# --------------------------------------------------------------------------------------
# losses (reference core/trainer.py:396-422, 507-553)
# --------------------------------------------------------------------------------------
def nerf_loss(args, rgb_pred, acc_pred, target, bgs=1.0, loss_weight=1.0):
    if args.use_background:
        rgb_pred = rgb_pred + (1. - acc_pred)[..., None] * bgs
    fn = {"L1": F.l1_loss, "MSE": F.mse_loss}[args.loss_fn]
    return fn(rgb_pred, target, reduction="mean") * loss_weight * args.rgb_loss_coef


def soft_softmax_loss(args, model, preds):
    labels = ((preds["T_i"] * preds["alpha"]) > 0).float()
    part_valid = 1 - preds["part_invalid"]
    if preds.get("p_valid") is not None:       # torch.ops.danbo.assign_blend's masked probabilities: the same p x part_valid
        return args.soft_softmax_loss_coef * (labels - preds["p_valid"].sum(-1)).pow(2.).mean()
    p = model.sigmoid(preds["confd"], preds["part_invalid"], mask_invalid=False, clamp=False)
    return args.soft_softmax_loss_coef * (labels - (p * part_valid).sum(-1)).pow(2.).mean()


def volume_scale_loss(args, model):
    gn = model.graph_net
    scale = gn.axis_scale.abs().clamp(min=gn.init_scale.to(gn.axis_scale.device) * 0.05)
    return torch.prod(scale, dim=-1).sum() * args.vol_scale_penalty
