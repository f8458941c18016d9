"""Ray batches and camera sequences for the entry points (reference: core/load_data.py, core/dataset.py).

The reference reads its images and poses from per-dataset HDF5 files (h5py / deepdish -- neither is in this image, nor
are the licensed datasets).  What the render path needs from that layer is small and is kept here with the reference's
names and dictionary keys:

  * `generate_bullet_time`                 camera ring around the subject            (reference load_data.py:56-71)
  * `PoseImageDataset.get_meta()`          -> `data_attrs` of `create_raycaster`     (reference dataset.py:469-525)
  * `PoseImageDataset.get_render_data()`   -> validation set of `render_testset`     (reference dataset.py:527-597)
  * `PoseImageDataset.sample_batch()`      -> the per-step ray batch: `N_sample_images` images x `N_rand / N_sample_images`
                                              pixels each, pose tensors replicated per ray, `N_uniques` = number of images
                                              (reference dataset.py:61-125 + `RayImageSampler` / `ray_collate_fn`)
  * `load_data(args)`                      -> (train_iterator, render_data, data_attrs)   (reference load_data.py:82-110)

Two array sources: `--dataset_type npz` (a file with the arrays listed in `PoseImageDataset.KEYS`, the layout of the
reference's h5 files) and `--dataset_type synthetic` (seeded poses and cameras of `core/utils/synthetic.py`; the target images
are rendered once by a *teacher* network through the same HIP path, so a training run has a ground truth to converge to).
Under `torch.distributed` every rank draws the same images and keeps its contiguous share of them (whole images, so
`N_uniques` stays an integer per rank -- SURVEY.md §8e).
"""
import math

import numpy as np
import torch

from .utils import synthetic as syn
from .utils.skeleton_utils import SMPLSkeleton, get_kp_bounding_cylinder, rotate_x, rotate_y, rotate_z


def generate_bullet_time(c2w, n_views=20, axis='y'):
    """[4,4] -> [n_views,4,4]: the camera rotated about a world axis in n_views equal steps of a full turn"""
    if axis not in 'xyz':
        raise NotImplementedError(f'rotate axis {axis} is not defined')
    rot = {'x': rotate_x, 'y': rotate_y, 'z': rotate_z}[axis]
    return np.array([rot(a) @ c2w for a in np.linspace(0, math.radians(360), n_views + 1)[:-1]])


class PoseImageDataset:
    """In-memory image / pose / camera arrays, one entry per image."""
    KEYS = ('imgs', 'fgs', 'bgs', 'bg_idxs', 'c2ws', 'focals', 'kp3d', 'bones', 'skts', 'rest_pose')

    def __init__(self, imgs, fgs, bgs, bg_idxs, c2ws, focals, kp3d, bones, skts, rest_pose, cam_idxs=None, centers=None,
                 ext_scale=0.001, skel_type=SMPLSkeleton, N_render=15, render_skip=1, seed=0):
        self.imgs = np.asarray(imgs, dtype=np.float32)                      # [N,H,W,3] in [0,1]
        self.fgs = np.asarray(fgs, dtype=np.float32).reshape(*self.imgs.shape[:3], 1)
        self.bgs = np.asarray(bgs, dtype=np.float32).reshape(-1, *self.imgs.shape[1:])
        self.bg_idxs = np.asarray(bg_idxs, dtype=np.int64)
        self.c2ws = np.asarray(c2ws, dtype=np.float32)
        N, self.H, self.W = self.imgs.shape[:3]
        self.focals = np.full(N, focals, dtype=np.float32) if np.isscalar(focals) else np.asarray(focals, dtype=np.float32)
        self.kp3d, self.bones, self.skts = (np.asarray(x, dtype=np.float32) for x in (kp3d, bones, skts))
        self.rest_pose = np.asarray(rest_pose, dtype=np.float64)
        self.cam_idxs = np.arange(N) if cam_idxs is None else np.asarray(cam_idxs, dtype=np.int64)
        self.centers, self.ext_scale, self.skel_type = centers, ext_scale, skel_type
        self.cyls = get_kp_bounding_cylinder(self.kp3d, ext_scale=ext_scale, extend_mm=250, top_expand_ratio=1.60,
                                             bot_expand_ratio=1.10, head='-y').astype(np.float32)
        self.N_render, self.render_skip = N_render, render_skip
        self.rng = np.random.default_rng(seed)
        # pixels a training ray may be drawn from: the image-space box of the pose's bounding cylinder, i.e. the pixels
        # `render_path` casts for this frame (stands in for the reference's precomputed `sampling_masks`, dataset.py:238-262)
        from .utils.skeleton_utils import cylinder_to_box_2d, nerf_c2w_to_extrinsic
        self.sampling_idxs = []
        for i in range(N):
            center = None if centers is None else centers[i]
            tl, br, _ = cylinder_to_box_2d(self.cyls[i], [self.H, self.W, float(self.focals[i])],
                                           nerf_c2w_to_extrinsic(self.c2ws[i]), center=center)
            ys, xs = np.meshgrid(np.arange(tl[1], br[1]), np.arange(tl[0], br[0]), indexing='ij')
            self.sampling_idxs.append((ys * self.W + xs).reshape(-1))

    def __len__(self):
        return len(self.imgs)

    def get_meta(self):
        N = len(self)
        hwf = (np.repeat([self.H], N), np.repeat([self.W], N), self.focals)
        return {'hwf': hwf, 'center': self.centers, 'c2ws': self.c2ws, 'near': 60., 'far': 100., 'n_views': N,
                'skel_type': self.skel_type, 'rest_pose': self.rest_pose, 'gt_kp3d': None, 'kp3d': self.kp3d,
                'skts': self.skts, 'bones': self.bones, 'betas': None, 'kp_map': None, 'kp_uidxs': None}

    def get_render_data(self):
        sel = np.arange(len(self))[::self.render_skip][:self.N_render]
        return {'imgs': self.imgs[sel], 'fgs': self.fgs[sel], 'bgs': self.bgs, 'bg_idxs': self.bg_idxs[sel],
                'bg_idxs_len': len(self.bgs), 'cam_idxs': self.cam_idxs[sel], 'cam_idxs_len': len(self.c2ws),
                'c2ws': self.c2ws[sel], 'hwf': (np.repeat([self.H], len(sel)), np.repeat([self.W], len(sel)), self.focals[sel]),
                'center': None if self.centers is None else self.centers[sel], 'kp_idxs': sel, 'kp_idxs_len': len(self.kp3d),
                'kp3d': self.kp3d[sel], 'skts': self.skts[sel], 'bones': self.bones[sel]}

    def _rays(self, i, pix):
        """pinhole rays of flat pixel indices of image i (reference ray_utils.py:7-29: unnormalised directions)"""
        f = self.focals[i]
        cx, cy = (self.W * 0.5, self.H * 0.5) if self.centers is None else self.centers[i]
        x, y = (pix % self.W).astype(np.float32), (pix // self.W).astype(np.float32)
        dirs = np.stack([(x - cx) / f, -(y - cy) / f, -np.ones_like(x)], -1)
        rays_d = (dirs[:, None, :] * self.c2ws[i, :3, :3]).sum(-1)
        return np.broadcast_to(self.c2ws[i, :3, 3], rays_d.shape), rays_d

    def sample_batch(self, N_images, N_rand, rank=0, world=1):
        """One training batch: every rank draws the same images / pixels from the shared generator and keeps the images
        [rank * N_images / world, (rank + 1) * N_images / world)."""
        assert N_images % world == 0 and N_rand % N_images == 0, "images must split evenly over ranks and rays over images"
        per = N_rand // N_images
        picks = self.rng.choice(len(self), size=N_images, replace=len(self) < N_images)
        pixels = [self.rng.choice(self.sampling_idxs[i], size=per, replace=len(self.sampling_idxs[i]) < per) for i in picks]
        lo, hi = rank * N_images // world, (rank + 1) * N_images // world
        out = {k: [] for k in ('rays_o', 'rays_d', 'target_s', 'fgs', 'bgs', 'kp3d', 'bones', 'skts', 'cyls', 'cam_idxs', 'kp_idx')}
        for i, pix in list(zip(picks, pixels))[lo:hi]:
            ro, rd = self._rays(i, pix)
            out['rays_o'].append(ro), out['rays_d'].append(rd)
            out['target_s'].append(self.imgs[i].reshape(-1, 3)[pix])
            out['fgs'].append(self.fgs[i].reshape(-1, 1)[pix])
            out['bgs'].append(self.bgs[self.bg_idxs[i]].reshape(-1, 3)[pix])
            for k, src in (('kp3d', self.kp3d), ('bones', self.bones), ('skts', self.skts), ('cyls', self.cyls)):
                out[k].append(np.broadcast_to(src[i], (per,) + src[i].shape))
            out['cam_idxs'].append(np.full(per, self.cam_idxs[i])), out['kp_idx'].append(np.full(per, i))
        batch = {k: torch.tensor(np.concatenate(v)) for k, v in out.items()}
        batch['N_uniques'] = hi - lo
        return batch


def batch_iterator(dataset, args, rank=0, world=1):
    while True:
        yield dataset.sample_batch(args.N_sample_images, args.N_rand, rank, world)


def synthetic_arrays(n_poses=8, n_cams=4, H=64, W=64, pose_seed=0, rest_scale=0.48, cam_dist=3.0):
    """Seeded poses x a bullet-time ring of cameras; images are filled in by the caller."""
    rest = syn.rest_pose(rest_scale)
    bones = syn.random_bones(n_poses, seed=pose_seed)
    _, skts, kps = syn.forward_kinematics(bones, rest)
    base = np.eye(4, dtype=np.float32)
    base[2, 3] = cam_dist
    ring = generate_bullet_time(base, n_cams)
    N = n_poses * n_cams
    pose_of, cam_of = np.repeat(np.arange(n_poses), n_cams), np.tile(np.arange(n_cams), n_poses)
    c2ws = ring[cam_of].copy()
    c2ws[:, :3, 3] += kps[pose_of, 0]                     # orbit each pose's pelvis
    return dict(c2ws=c2ws, focals=np.full(N, 1.25 * H, dtype=np.float32), kp3d=kps[pose_of], bones=bones[pose_of],
                skts=skts[pose_of], rest_pose=rest, cam_idxs=np.arange(N), H=H, W=W)


def render_targets(ray_caster, arrays, N_samples, N_importance, device, chunk=65536, bg_color=1.0):
    """Images of the teacher network for `synthetic_arrays`: every pixel cast through the caster in eval mode and composited
    over a constant background; foreground = accumulated opacity > 0.5, hard-matted."""
    from .utils.ray_utils import get_rays
    H, W = arrays['H'], arrays['W']
    t = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float32, device=device)  # noqa: E731
    cyls = get_kp_bounding_cylinder(arrays['kp3d'], ext_scale=0.001, extend_mm=250, top_expand_ratio=1.60,
                                    bot_expand_ratio=1.10, head='-y')
    imgs, fgs = [], []
    ray_caster.eval()
    with torch.no_grad():
        for i in range(len(arrays['c2ws'])):
            ro, rd = get_rays(H, W, float(arrays['focals'][i]), t(arrays['c2ws'][i]))
            ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
            vd = rd / rd.norm(dim=-1, keepdim=True)
            rb = torch.cat([ro, rd, torch.zeros_like(rd[:, :1]), torch.ones_like(rd[:, :1]), vd], -1)
            R = rb.shape[0]
            rep = lambda x: t(x[i])[None].expand(R, *x[i].shape)  # noqa: E731
            out = ray_caster(rb, N_samples=N_samples, N_importance=N_importance, kp_batch=rep(arrays['kp3d']),
                             skts=rep(arrays['skts']), bones=rep(arrays['bones']), cyls=rep(cyls),
                             cams=torch.full((R,), int(arrays['cam_idxs'][i]), dtype=torch.int64, device=device), N_uniques=1)
            acc = out['acc_map'].reshape(H, W, 1)
            img = (out['rgb_map'].reshape(H, W, 3) + (1. - acc) * bg_color).clamp(0, 1)
            fg = (acc > 0.5).float()
            # hard matte, like the photographs + binary masks of the real datasets: outside the mask the image IS the background
            imgs.append((img * fg + (1. - fg) * bg_color).cpu().numpy())
            fgs.append(fg.cpu().numpy())
    return np.stack(imgs), np.stack(fgs)


def teacher_caster(args, arrays, device):
    """The network that paints the synthetic targets: the architecture `args` describes (one of the shipped configs) with the
    seeded, structure-producing weights of `synthetic.make_state_dict`."""
    import copy
    from .raycasters import create_raycaster
    name = 'anerf_base' if args.nerf_type == 'nerf' else ('danbo_base' if args.opt_framecode else 'danbo_surreal')
    cfg = dict(syn.model_config(name), view_type=args.view_type, ray_tr_type=args.ray_tr_type)
    N = len(arrays['c2ws'])
    targs = copy.copy(args)
    targs.no_reload, targs.ft_path = True, None
    attrs = dict(skel_type=SMPLSkeleton, near=60., far=100., n_views=N, rest_pose=arrays['rest_pose'],
                 hwf=(arrays['H'], arrays['W'], arrays['focals']))
    caster = create_raycaster(targs, attrs, device=device)[1]['ray_caster']
    n_codes = N if args.n_framecodes is None else args.n_framecodes
    sd = syn.make_state_dict(cfg, seed=getattr(args, 'syn_seed', 0) + 1, n_framecodes=n_codes, rest=arrays['rest_pose'])
    caster.network.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    return caster


def get_dataset(args, device=None, images=None):
    kind = getattr(args, 'dataset_type', 'synthetic')
    if kind == 'npz':
        d = np.load(args.datadir)
        opt = {k: d[k] for k in ('cam_idxs', 'centers') if k in d}
        return PoseImageDataset(*[d[k] for k in PoseImageDataset.KEYS], ext_scale=args.ext_scale, **opt)
    if kind != 'synthetic':
        raise NotImplementedError(f"dataset_type '{kind}': the reference's HDF5 datasets need h5py and the licensed data; "
                                  "export the arrays to .npz (PoseImageDataset.KEYS) and use --dataset_type npz")
    res = getattr(args, 'syn_res', 64)
    arrays = synthetic_arrays(getattr(args, 'syn_poses', 8), getattr(args, 'syn_cams', 4), res, res,
                              pose_seed=getattr(args, 'syn_seed', 0), rest_scale=getattr(args, 'syn_rest_scale', 0.48))
    if images is None:
        images = render_targets(teacher_caster(args, arrays, device), arrays, args.N_samples, args.N_importance, device)
    imgs, fgs = images
    N, H, W = imgs.shape[:3]
    return PoseImageDataset(imgs, fgs, np.ones((1, H, W, 3), np.float32), np.zeros(N, np.int64), arrays['c2ws'],
                            arrays['focals'], arrays['kp3d'], arrays['bones'], arrays['skts'], arrays['rest_pose'],
                            cam_idxs=arrays['cam_idxs'], ext_scale=args.ext_scale)


def load_data(args, device=None, images=None, rank=0, world=1):
    """-> (train batch iterator, render_data, data_attrs); `images` = (imgs, fgs) replaces the teacher render (CPU tests)"""
    dataset = get_dataset(args, device=device, images=images)
    return batch_iterator(dataset, args, rank, world), dataset.get_render_data(), dataset.get_meta()
