"""The differentiable HIP stages as `torch.library` custom operators ("Python wraps each pair as a
torch.library.custom_op with register_autograd"), so that a caller -- the reference's trainer included -- can trace or
`torch.compile` around them:

    torch.ops.danbo.composite(raw, z, rays_d, B, noise)                  NeRF.raw2outputs (reference core/networks/nerf.py:281-347)
        -> rgb_map, disp_map, acc_map, weights, alpha                     backward: danbo_composite_bwd (d rgb_map, d acc_map -> d raw)
    torch.ops.danbo.bone_gather(volumes, axis_scale, pts, skts, align, rows)
        -> part_feat [n,24,15]                                            FactorizeGNN.sample_from_volume (gnn_backbone.py:787-828)
                                                                          backward: danbo_bone_gather_bwd (d volumes, d axis_scale)

Both enqueue on the current HIP stream through the C ABI (include/danbo_hip.h); there is no CPU implementation (calling them
with CPU tensors raises), only shape-propagating fake kernels for tracing.  core/train_path.py (the autograd training path)
is built on these two operators; the fused training step (core/train_engine.py) does not need autograd at all.
"""
import ctypes
from typing import Optional, Tuple

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
