"""Rendering entry point (reference: run_render.py -- parser :31-97, `load_nerf` :99-134, sequence loaders :838-999,
`evaluate_metric` :1178-1263, `render_mesh` :1266-1281, `run_render` :1283-1347).

    python danbo-pytorch_amd/run_render.py --nerf_args logs/demo/args.txt --ckptpath logs/demo/001000.tar \
        --dataset synthetic --entry val --render_type bullet --runname demo_bullet --render_res 256 256 [--eval]

A trained model (the reference's `args.txt` + `.tar` checkpoint formats) is rendered along a generated camera / pose
sequence -- `bullet` (camera ring around selected poses), `interpolate` (axis-angle blend between selected poses), `bubble`
(camera wobbling around its position), `val` /
`selected` (the data's own cameras) -- or sampled on a density grid (`--render_mesh`).  The sequence generators take arrays
instead of the reference's HDF5 paths (no h5py / deepdish here); `--dataset synthetic` builds them from seeded poses, `--dataset
npz --entry file.npz` reads them.  Outputs: `image/`, `acc/` as .npy stacks (no imageio here), `bboxes.npy`, and with
`--eval` `scores.npy` + `score_final.txt` like the reference.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from core.config import config_parser as nerf_config_parser, txt_to_argstring  # noqa: E402
from core.load_data import PoseImageDataset, generate_bullet_time, get_dataset  # noqa: E402
from core.raycasters import create_raycaster  # noqa: E402
from core.utils.evaluation_helpers import evaluate_in_boxes  # noqa: E402
from core.utils.skeleton_utils import get_smpl_l2ws  # noqa: E402
from run_nerf import render_path  # noqa: E402


def config_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--nerf_args', type=str, required=True, help='args.txt in the training log directory')
    p.add_argument('--ckptpath', type=str, required=True, help='checkpoint (.tar)')
    p.add_argument('--render_res', nargs='+', type=int, default=[1000, 1000], help='(H, W) of the rendered images')
    p.add_argument('--dataset', type=str, required=True, help="'synthetic' or 'npz'")
    p.add_argument('--entry', type=str, required=True, help='catalog entry: a split name (synthetic) or an .npz path')
    p.add_argument('--white_bkgd', action='store_true')
    p.add_argument('--render_type', type=str, default='bullet', help='bullet | interpolate | bubble | pose_rotate | selected | val')
    p.add_argument('--render_mesh', action='store_true', help='sample the density grid instead of rendering images')
    p.add_argument('--mesh_res', type=int, default=255)
    p.add_argument('--mesh_radius', type=float, default=1.8)
    p.add_argument('--render_confd', action='store_true')
    p.add_argument('--render_entropy', action='store_true')
    p.add_argument('--selected_idxs', nargs='+', type=int, default=None)
    p.add_argument('--selected_framecode', type=int, default=None)
    p.add_argument('--n_bullet', type=int, default=10)
    p.add_argument('--n_step', type=int, default=10)
    p.add_argument('--outputdir', type=str, default='render_output/')
    p.add_argument('--runname', type=str, required=True)
    p.add_argument('--eval', action='store_true', help='PSNR / SSIM inside the bounding boxes against the data images')
    p.add_argument('--no_save', action='store_true')
    return p


# ---------------------------------------------------------------------------------------------------- sequences
def _select_cameras(c2ws, focals, selected_idxs):
    focals = np.array([focals] * len(selected_idxs)) if isinstance(focals, float) else focals[selected_idxs]
    return c2ws[selected_idxs].copy(), focals


def _pose_chain(bones, rest_pose, root_locs):
    """axis-angle poses + root positions -> joint positions, world-to-bone matrices"""
    l2ws = np.array([get_smpl_l2ws(b, rest_pose, 1.0) for b in bones])
    l2ws[..., :3, -1] += root_locs
    return l2ws[..., :3, -1], np.linalg.inv(l2ws)


def load_bullettime(kps, bones, c2ws, focals, rest_pose, selected_idxs, n_bullet=30, centers=None, undo_rot=False,
                    center_cam=True, center_kps=True):
    """Every selected pose seen from `n_bullet` cameras on a ring about the world y axis (reference :895-968).  The camera is
    first moved onto the axis (`center_cam`) and the pose onto the origin (`center_kps`), so the ring orbits the subject.
    -> kps, skts, c2ws, cam_idxs, focals, bones, centers, one entry per (pose, view)"""
    selected_idxs = np.asarray(selected_idxs)
    c2ws, focals = _select_cameras(c2ws, focals, selected_idxs)
    if center_cam:
        shift = c2ws[..., :2, -1].copy()
        c2ws[..., :2, -1] = 0.
    c2ws = np.array([generate_bullet_time(c, n_bullet) for c in c2ws]).reshape(-1, 4, 4)
    rep = lambda x: np.repeat(x, n_bullet, axis=0)  # noqa: E731
    kps, bones = kps[selected_idxs].copy(), bones[selected_idxs].copy()
    if center_kps:
        kps -= kps[..., :1, :].copy()
    elif center_cam:
        kps[..., :2] -= shift[:, None]
    if undo_rot:
        bones[..., 0, :] = np.array([1.5708, 0., 0.], dtype=np.float32)
    kps, skts = _pose_chain(bones, rest_pose, kps[..., :1, :])
    centers = rep(centers[selected_idxs]) if centers is not None else None
    return rep(kps), rep(skts), c2ws, rep(selected_idxs), rep(focals), rep(bones), centers


def load_interpolate(kps, bones, c2ws, focals, rest_pose, selected_idxs, n_step=10, undo_rot=False, center_cam=False,
                     center_kps=False):
    """`n_step` linear blends of the axis-angle parameters between consecutive selected poses, all placed at the first pose's
    root and seen from the first selected camera (reference :838-893).  -> kps, skts, c2ws, cam_idxs, focals, bones"""
    selected_idxs = np.asarray(selected_idxs)
    c2ws, focals = _select_cameras(c2ws, focals, selected_idxs)
    if center_cam:
        shift = c2ws[..., :2, -1].copy()
        c2ws[..., :2, -1] = 0.
    kps, bones = kps[selected_idxs].copy(), bones[selected_idxs].copy()
    if center_kps:
        kps -= kps[..., :1, :].copy()
    elif center_cam:
        kps[..., :2] -= shift[:, None]
    if undo_rot:
        bones[..., 0, :] = np.array([1.5708, 0., 0.], dtype=np.float32)
    w = np.linspace(0, 1.0, n_step, endpoint=False).reshape(-1, 1, 1)
    seq = [bones[i:i + 1] * (1 - w) + bones[i + 1:i + 2] * w for i in range(len(bones) - 1)] + [bones[-1:]]
    seq = np.concatenate(seq, axis=0)
    kps, skts = _pose_chain(seq, rest_pose, kps[:1, :1, :])
    n = len(kps)
    return kps, skts, c2ws[:1].repeat(n, 0), selected_idxs[:1].repeat(n, 0), focals[:1].repeat(n, 0), seq


def load_selected(kps, bones, c2ws, focals, rest_pose, selected_idxs, centers=None):
    """The selected frames with their own cameras (reference :971-999)."""
    selected_idxs = np.asarray(selected_idxs)
    c2ws, focals = _select_cameras(c2ws, focals, selected_idxs)
    kps, bones = kps[selected_idxs], bones[selected_idxs]
    kps, skts = _pose_chain(bones, rest_pose, kps[..., :1, :])
    return kps, skts, c2ws, selected_idxs, focals, bones, (centers[selected_idxs] if centers is not None else None)


def load_bubble(kps, bones, c2ws, focals, rest_pose, selected_idxs, x_deg=15., y_deg=25., z_t=0.1, centers=None, n_step=5):
    """Every selected pose seen from a camera that wobbles around its own position: `n_step` points of a closed curve, rotation
    about x by (cos t - 1) x_deg and about y by sin t y_deg, pushed back along z by (sin t + 1) z_t x (distance of the first
    camera) (reference :1001-1073).  Poses are centred on their root.  Like the reference, `bones` comes back once per selected
    pose (NOT repeated per step; `render_path` cycles shorter tensors), everything else once per (pose, step)."""
    from core.utils.skeleton_utils import rotate_x, rotate_y
    selected_idxs = np.asarray(selected_idxs)
    c2ws, focals = _select_cameras(c2ws, focals, selected_idxs)
    z_t = z_t * c2ws[0, 2, -1]
    t = np.linspace(0., 2 * np.pi, n_step, endpoint=True)
    motions = [rotate_x(xm) @ rotate_y(ym) for xm, ym in zip((np.cos(t) - 1.) * np.deg2rad(x_deg), np.sin(t) * np.deg2rad(y_deg))]
    z_trans = (np.sin(t) + 1.) * z_t
    out = []
    for c2w in c2ws:
        for m, dz in zip(motions, z_trans):
            c = c2w.copy()
            c[2, -1] += dz
            out.append(m @ c)
    rep = lambda x: np.repeat(x, n_step, axis=0)  # noqa: E731
    kps, bones = kps[selected_idxs].copy(), bones[selected_idxs].copy()
    kps -= kps[..., :1, :].copy()
    kps, skts = _pose_chain(bones, rest_pose, kps[..., :1, :])
    centers = rep(centers[selected_idxs]) if centers is not None else None
    return rep(kps), rep(skts), np.array(out).reshape(-1, 4, 4), rep(selected_idxs), rep(focals), bones, centers


def load_pose_rotate(kps, bones, c2ws, focals, rest_pose, selected_idxs, n_bullet=30):
    """ONE selected pose whose root joint is turned through full circles about the world y, x and z axes (`n_bullet // 3` steps
    each) in front of its fixed camera (reference :800-836; axis-angle <-> matrix there is pytorch3d, here scipy's Rotation).
    -> kps, skts, bones, c2ws, cam_idxs, focals (the reference's order for this loader)"""
    from scipy.spatial.transform import Rotation
    selected_idxs = np.asarray(selected_idxs)
    if len(selected_idxs) != 1:
        raise ValueError("pose_rotate renders one selected pose (the reference's array shapes only fit one)")
    kps, bones = kps[selected_idxs], bones[selected_idxs].copy()
    root = np.eye(4, dtype=np.float32)
    root[:3, :3] = Rotation.from_rotvec(bones[0, 0].astype(np.float64)).as_matrix()
    rots = np.concatenate([generate_bullet_time(root, n_bullet // 3, axis) for axis in 'yxz'], 0)
    n = len(rots)
    bones = bones.repeat(n, 0)
    bones[:, 0, :] = Rotation.from_matrix(rots[:, :3, :3].astype(np.float64)).as_rotvec().astype(bones.dtype)
    c2ws, focals = _select_cameras(c2ws, focals, selected_idxs)
    kps_out, skts = _pose_chain(bones, rest_pose, kps[..., :1, :])
    return kps_out, skts, bones, c2ws.repeat(n, 0), selected_idxs.repeat(n, 0), focals.repeat(n, 0)


# ---------------------------------------------------------------------------------------------------- model / data
def load_nerf(args, nerf_args, device):
    """Network of `nerf_args` with the checkpoint's weights, frozen, in eval mode; the frame-code table takes its size from the
    checkpoint (reference :99-134)."""
    nerf_args.ft_path, nerf_args.finetune = args.ckptpath, True
    ckpt = torch.load(args.ckptpath, map_location='cpu')
    dataset = get_render_dataset(args, nerf_args, device)
    attrs = dataset.get_meta()
    codes = ckpt['network_fn_state_dict'].get('framecodes.codes.weight')
    if codes is not None:
        attrs['n_views'] = codes.shape[0]
    kw_train, kw_test, _, grad_vars, _, _ = create_raycaster(nerf_args, attrs, device=device)
    for p in grad_vars:
        p.requires_grad = False
    kw_test['ray_caster'] = kw_train['ray_caster'].eval()
    return kw_test, dataset


def get_render_dataset(args, nerf_args, device):
    if args.dataset == 'npz':
        d = np.load(args.entry)
        opt = {k: d[k] for k in ('cam_idxs', 'centers') if k in d}
        return PoseImageDataset(*[d[k] for k in PoseImageDataset.KEYS], ext_scale=nerf_args.ext_scale, **opt)
    if args.dataset != 'synthetic':
        raise NotImplementedError(f"dataset '{args.dataset}': HDF5 catalogs are not readable here; export to .npz")
    nerf_args.dataset_type = 'synthetic'
    return get_dataset(nerf_args, device=device)


def load_render_data(args, nerf_args, dataset):
    """-> (render_data for `render_path`, gt_dict for evaluation)"""
    H0, W0 = dataset.H, dataset.W
    H, W = args.render_res if args.render_res is not None else (H0, W0)
    scale = float(H) / float(H0)
    focals = dataset.focals * scale
    centers = dataset.centers * scale if dataset.centers is not None else None
    sel = np.asarray(args.selected_idxs if args.selected_idxs is not None else np.arange(len(dataset))[:dataset.N_render])
    src = (dataset.kp3d, dataset.bones, dataset.c2ws, focals, dataset.rest_pose, sel)
    gt = dict(gt_paths=None, gt_mask_paths=None, is_gt_paths=False, bg_imgs=None, bg_indices=None)
    if args.render_type == 'bullet':
        kps, skts, c2ws, cam_idxs, focals, bones, centers = load_bullettime(*src, n_bullet=args.n_bullet, centers=centers)
    elif args.render_type == 'interpolate':
        kps, skts, c2ws, cam_idxs, focals, bones = load_interpolate(*src, n_step=args.n_step)
        centers = None
    elif args.render_type == 'pose_rotate':
        kps, skts, bones, c2ws, cam_idxs, focals = load_pose_rotate(*src, n_bullet=args.n_bullet)
        centers = None
    elif args.render_type == 'bubble':
        kps, skts, c2ws, cam_idxs, focals, bones, centers = load_bubble(*src, centers=centers, n_step=args.n_step)
        bones = np.repeat(bones, args.n_step, axis=0)     # the pose GNN needs the bones of every frame, in frame order
    elif args.render_type in ('selected', 'val'):
        kps, skts, c2ws, cam_idxs, focals, bones, centers = load_selected(*src, centers=centers)
        if scale == 1.0:
            gt.update(gt_paths=dataset.imgs[sel], gt_mask_paths=dataset.fgs[sel], bg_imgs=dataset.bgs, bg_indices=dataset.bg_idxs[sel])
    else:
        raise NotImplementedError(f"render_type '{args.render_type}'")
    if args.selected_framecode is not None:
        cam_idxs = np.full_like(cam_idxs, args.selected_framecode)
    data = dict(render_poses=c2ws, hwf=(int(H), int(W), focals.astype(np.float32)), centers=centers, kp=kps, skts=skts,
                bones=bones, cams=cam_idxs if nerf_args.opt_framecode else None)
    return data, gt


def to_tensors(data, device):
    out = {}
    for k, v in data.items():
        if isinstance(v, np.ndarray) and k != 'centers':
            out[k] = torch.tensor(v, dtype=torch.int64 if k == 'cams' else torch.float32, device=device)
        else:
            out[k] = v
    return out


def evaluate_metric(rgbs, accs, bboxes, gt_dict, basedir):
    scores = evaluate_in_boxes(rgbs, accs, bboxes, gt_dict['gt_paths'], gt_dict['gt_mask_paths'], gt_dict['bg_imgs'],
                               gt_dict['bg_indices'])
    np.save(os.path.join(basedir, 'scores.npy'), scores, allow_pickle=True)
    with open(os.path.join(basedir, 'score_final.txt'), 'w') as f:
        for k, v in scores.items():
            f.write(f'{k}: {np.mean(v)}\n')
    return scores


@torch.no_grad()
def render_mesh(basedir, render_kwargs, tensor_data, chunk=4096, radius=1.80, res=255):
    """Density on a (res+1)^3 grid around every pose (reference :1266-1281).  Marching cubes (PyMCubes / trimesh) is a host
    step outside this image: the clamped grids are saved as `meshes/NNN_sigma.npy` for it."""
    caster = render_kwargs['ray_caster']
    os.makedirs(os.path.join(basedir, 'meshes'), exist_ok=True)
    kps, skts, bones = tensor_data['kp'], tensor_data['skts'], tensor_data['bones']
    for i in range(len(kps)):
        raw = caster(kps=kps[i:i + 1], skts=skts[i:i + 1], bones=bones[i:i + 1], radius=radius,
                     render_kwargs=render_kwargs['preproc_kwargs'], res=res, netchunk=chunk, fwd_type='mesh')
        np.save(os.path.join(basedir, 'meshes', f'{i:03d}_sigma.npy'), np.maximum(raw.cpu().numpy(), 0))


def run_render(argv=None):
    args = config_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("run_render.py drives the HIP render path: no GPU visible")
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    nerf_args, unknown = nerf_config_parser().parse_known_args(txt_to_argstring(args.nerf_args))
    print(f'UNKNOWN ARGS: {unknown}')
    render_kwargs, dataset = load_nerf(args, nerf_args, device)
    render_data, gt_dict = load_render_data(args, nerf_args, dataset)
    tensor_data = to_tensors(render_data, device)
    basedir = os.path.join(args.outputdir, args.runname)
    os.makedirs(basedir, exist_ok=True)
    if args.render_mesh:
        render_mesh(basedir, render_kwargs, tensor_data, res=args.mesh_res, radius=args.mesh_radius)
        return None
    render_kwargs = dict(render_kwargs, render_confd=args.render_confd, render_entropy=args.render_entropy)
    rgbs, _, accs, _, bboxes = render_path(render_kwargs=render_kwargs, chunk=nerf_args.chunk, ext_scale=nerf_args.ext_scale,
                                           ret_acc=True, white_bkgd=args.white_bkgd, **tensor_data)
    scores = None
    if gt_dict['gt_paths'] is not None and args.eval:
        scores = evaluate_metric(rgbs, accs, bboxes, gt_dict, basedir)
    if not args.no_save:
        np.save(os.path.join(basedir, 'image.npy'), (rgbs * 255).astype(np.uint8))
        np.save(os.path.join(basedir, 'acc.npy'), (accs * 255).astype(np.uint8))
        np.save(os.path.join(basedir, 'bboxes.npy'), np.array(bboxes), allow_pickle=True)
    return rgbs, accs, bboxes, scores


if __name__ == '__main__':
    run_render()
