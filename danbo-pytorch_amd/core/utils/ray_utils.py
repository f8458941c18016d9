"""Ray helpers (reference: core/utils/ray_utils.py).  Ray generation stays a few torch ops
(it is per-frame set-up, not part of the hot path); bounds and sampling call the HIP kernels."""
import numpy as np
import torch

from .. import hip_ops as ops


def get_rays(H, W, focal, c2w, center=None):
    """Pinhole rays, camera looks down -z: dir = [(i-cx)/fx, -(j-cy)/fy, -1] rotated by c2w."""
    if isinstance(focal, float) or (len(np.reshape(focal, -1)) < 2):
        fx = fy = float(np.reshape(focal, -1)[0])
    else:
        fx, fy = focal
    cx, cy = (W * 0.5, H * 0.5) if center is None else center
    dev = c2w.device if torch.is_tensor(c2w) else None
    c2w = torch.as_tensor(c2w, dtype=torch.float32, device=dev)
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev),
                          torch.arange(W, dtype=torch.float32, device=dev), indexing='ij')
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    rays_d = (dirs[..., None, :] * c2w[:3, :3]).sum(-1)
    return c2w[:3, -1].expand(rays_d.shape), rays_d


def get_near_far_in_cylinder(rays_o, rays_d, cyl, near=0., far=1., chunk=None):
    """cyl [R,5] per ray (the reference's layout) or [G,5] per pose (R % G == 0).  All R rays
    form one nan-mean chunk unless `chunk` is given (the reference is called once per chunk)."""
    R = rays_o.shape[0]
    ni = near.reshape(-1) if torch.is_tensor(near) else None
    fi = far.reshape(-1) if torch.is_tensor(far) else None
    n, f = ops.near_far_cylinder(rays_o, rays_d, cyl, 0. if ni is not None else near, 1. if fi is not None else far,
                                 chunk or R, ni, fi)
    return n[:, None], f[:, None]


def sample_from_lineseg(near, far, N_lines, N_samples, perturb=0., lindisp=False, pytest=False):
    if lindisp:
        raise NotImplementedError("lindisp is not used by any shipped config")
    t_rand = torch.rand(N_lines, N_samples, device=near.device) if perturb > 0. else None
    return ops.coarse_samples(near.reshape(-1), far.reshape(-1), N_samples, t_rand)


def isample_from_lineseg(z_vals, weights, N_importance, det=False, pytest=False, is_only=False, alpha_base=0.01):
    if not is_only:
        raise NotImplementedError("two-network importance sampling (single_net=False) is out of scope")
    u = None if det else torch.rand(z_vals.shape[0], N_importance, device=z_vals.device)
    z_all, z_fine, idx = ops.importance_samples(z_vals, weights.detach(), N_importance, u)
    return z_all, z_fine, idx


def kp_to_valid_rays(poses, H, W, focal, kps=None, cylinder_params=None, skts=None, centers=None, ext_scale=0.00035):
    """Rays of the pixels inside the image-space box of each pose's bounding cylinder (reference :84-138): render-time
    ray selection, so that only pixels that can see the body are cast.  poses: camera-to-world matrices, one per image
    (images of the same pose consecutive, `cyl_idx = i % n_poses`).  -> rays [(o, d)], flat pixel indices, cylinders, boxes"""
    from .skeleton_utils import cylinder_to_box_2d, get_kp_bounding_cylinder, nerf_c2w_to_extrinsic
    if cylinder_params is None:
        assert kps is not None
        cyl = get_kp_bounding_cylinder(kps.cpu().numpy(), ext_scale=ext_scale, extend_mm=250, top_expand_ratio=1.60,
                                       bot_expand_ratio=1.10, head='-y')
        cylinder_params = torch.tensor(cyl, dtype=torch.float32, device=kps.device)
    rays, valid_idxs, bboxes = [], [], []
    for i, c2w in enumerate(poses):
        cyl = cylinder_params[i % kps.shape[0]]
        f = focal if isinstance(focal, float) else focal[i]
        center = None if centers is None else centers[i]
        h = H if isinstance(H, int) else H[i]
        w = W if isinstance(W, int) else W[i]
        ray_o, ray_d = get_rays(h, w, f, c2w, center=center)
        ray_o, ray_d = ray_o.cpu(), ray_d.cpu()
        tl, br, _ = cylinder_to_box_2d(cyl.cpu().numpy(), [h, w, f], nerf_c2w_to_extrinsic(c2w.cpu().numpy()), center=center)
        vh, vw = torch.meshgrid(torch.arange(tl[1], br[1]), torch.arange(tl[0], br[0]), indexing='ij')
        idx = (vh * w + vw).reshape(-1)
        rays.append((ray_o.reshape(-1, 3)[idx], ray_d.reshape(-1, 3)[idx]))
        valid_idxs.append(idx)
        bboxes.append((tl, br))
    return rays, valid_idxs, cylinder_params, bboxes
