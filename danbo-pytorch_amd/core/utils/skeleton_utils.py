"""Skeleton helpers needed by the render path (subset of the reference's
core/utils/skeleton_utils.py; plotting / dataset helpers are out of scope).

When this package is overlaid on a full checkout of the reference, keep the reference's own
skeleton_utils.py instead of this file: every name defined here exists there with the same
meaning.
"""
from collections import namedtuple

import numpy as np

from .synthetic import JOINT_NAMES, JOINT_TREES, SMPL_REST_POSE

Skeleton = namedtuple("Skeleton", ["joint_names", "joint_trees", "root_id", "nonroot_id", "cutoffs", "end_effectors"])

# 24-joint SMPL tree (reference: skeleton_utils.py:83-110)
SMPLSkeleton = Skeleton(
    joint_names=list(JOINT_NAMES),
    joint_trees=np.array(JOINT_TREES),
    root_id=0,
    nonroot_id=list(range(1, 24)),
    cutoffs={'hip': 200, 'spine': 300, 'knee': 70, 'ankle': 70, 'foot': 40, 'collar': 100,
             'neck': 100, 'head': 120, 'shoulder': 70, 'elbow': 70, 'wrist': 60, 'hand': 60},
    end_effectors=[10, 11, 15, 22, 23],
)
CMUSkeleton = SMPLSkeleton
smpl_rest_pose = SMPL_REST_POSE


def get_children_joints(skel_type=SMPLSkeleton):
    """children[p] lists every i with parent p -- the root lists itself first, as in the reference."""
    children = [[] for _ in skel_type.joint_trees]
    for i, p in enumerate(skel_type.joint_trees):
        children[p].append(i)
    return children


def _ry(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]], dtype=np.float32)


def _rx(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float32)


def _acos(a):
    return np.arccos(np.clip(a, -1. + 1e-8, 1. - 1e-8))


def get_axis_aligned_rotation(vec):
    """4x4 whose rotation block R satisfies R @ vec = |vec| * e_z (reference :544-566):
    yaw about y until the xz-projection lies on +z, then pitch about x."""
    v = np.asarray(vec)
    xz = v[[0, 2]] / np.linalg.norm(v[[0, 2]])
    ry = _ry(_acos(xz[1]) * np.sign(xz[0]))
    v1 = ry @ v
    yz = v1[1:3] / np.linalg.norm(v1[1:3])
    rx = _rx(_acos(yz[1]) * np.sign(yz[0]))
    out = np.eye(4, dtype=np.float32)
    # the reference returns inv(rx ry)^T; for a rotation that is rx ry itself up to round-off,
    # so reproduce the inverse-transpose literally to stay bit-identical
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = rx @ ry
    out[:] = np.linalg.inv(m).T
    return out


def bone_align_transforms(rest_pose, skel_type=SMPLSkeleton):
    """[24,4,4] float32: for bones with exactly one child, rotate the rest-pose bone onto +z and
    shift by -|bone|/2 along z; identity otherwise (reference raycasters.py:548-591, 'align')."""
    rest = np.asarray(rest_pose).reshape(len(skel_type.joint_trees), 3)
    T = np.tile(np.eye(4, dtype=np.float32), (len(rest), 1, 1))
    for parent, c in enumerate(get_children_joints(skel_type)):
        if len(c) != 1:
            continue
        d = rest[c[0]] - rest[parent]
        m = get_axis_aligned_rotation(d).astype(np.float64)
        m[:3, 3] = -0.5 * np.linalg.norm(d) * np.array([0., 0., 1.], dtype=np.float32)
        T[parent] = m.astype(np.float32)
    return T


def calculate_bone_length(kp, skel_type=SMPLSkeleton, to_child=False):
    tree = skel_type.joint_trees
    if not to_child:
        return np.array([np.sqrt(((kp[i] - kp[tree[i]]) ** 2).sum()) for i in range(1, kp.shape[0])])
    out = []
    for i, c in enumerate(get_children_joints(skel_type)):
        if len(c) < 1:
            out.append(-1.)
            continue
        cc = c[1:] if i == 0 else c
        out.append(np.sqrt(((kp[i:i + 1] - kp[cc]) ** 2).sum(-1)).mean())
    return np.array(out)


def get_skel_profile_from_rest_pose(rest_pose, skel_type=SMPLSkeleton):
    """Body proportions used to initialise the per-bone volume extents (reference :1515-1571)."""
    rp = np.asarray(rest_pose)
    if rp.ndim == 2:
        rp = rp[None]
    names = skel_type.joint_names
    prof = {}
    for key in ('shoulder', 'hip', 'collar', 'knee'):
        idx = [i for i, n in enumerate(names) if key in n]
        prof[f'{key}_width'] = np.linalg.norm(rp[:, idx[0]] - rp[:, idx[1]], axis=-1)
    prof['bone_lens'] = np.concatenate(
        [np.zeros((len(rp), 1)), np.array([calculate_bone_length(r, skel_type) for r in rp])], -1)
    prof['bone_lens_to_child'] = np.array([calculate_bone_length(r, skel_type, to_child=True) for r in rp])
    groups = dict(torso=('shoulder', 'spine', 'collar', 'neck', 'pelvis'), arm=('elbow', 'wrist', 'hand'),
                  leg=('hip', 'knee', 'ankle', 'foot'), head=('head',))
    for g, keys in groups.items():
        prof[f'{g}_idxs'] = np.array([i for i, n in enumerate(names) if any(k in n for k in keys)])
    return prof


def get_kp_bounding_cylinder(kp, skel_type=None, ext_scale=0.001, extend_mm=250, top_expand_ratio=1.,
                             bot_expand_ratio=0.25, head=None):
    """(cx, cz|cy, radius, top, bot) per pose (reference :568-618)."""
    assert head is not None
    g_axes, h_axis = ([0, 1], 2) if head.endswith('z') else ([0, 2], 1)
    flip = -1.0 if head.startswith('-') else 1.0
    kp = np.asarray(kp)
    batched = kp.ndim == 3
    k = kp if batched else kp[None]
    root = k[:, 0]
    dist = np.linalg.norm(k[..., g_axes] - root[:, None, g_axes], axis=-1).max(-1)
    hi = (flip * k[..., h_axis]).max(-1)
    lo = (flip * k[..., h_axis]).min(-1)
    ext = extend_mm * ext_scale
    cyl = np.stack([root[:, g_axes[0]], root[:, g_axes[1]], dist + ext,
                    flip * (hi + ext * top_expand_ratio), flip * (lo - ext * bot_expand_ratio)], -1)
    return cyl if batched else cyl[0]


# ---------------------------------------------------------------------------------------------
# camera helpers of the render-time ray selection (reference :445-446, 633-707, 1431-1450)
# ---------------------------------------------------------------------------------------------
def swap_mat(mat):
    """NeRF camera convention [right, up, back] <-> [right, -up, -forward]: negate columns 1 and 2."""
    return np.concatenate([mat[..., 0:1], -mat[..., 1:2], -mat[..., 2:3], mat[..., 3:]], axis=-1)


def nerf_c2w_to_extrinsic(c2w):
    return np.linalg.inv(swap_mat(c2w))


def focal_to_intrinsic_np(focal):
    fx, fy = (focal, focal) if isinstance(focal, float) or np.asarray(focal).size < 2 else focal
    return np.array([[fx, 0, 0, 0], [0, fy, 0, 0], [0, 0, 1, 0]], dtype=np.float32)


def cylinder_to_box_2d(cylinder_params, hwf, w2c=None, scale=1.0, center=None, make_int=True):
    """image-space bounding box of a bounding cylinder: project 50 points on each cap rim and take the extrema
    -> (top-left [N,2], bottom-right [N,2], projected points)."""
    H, W, focal = hwf
    cyl = np.asarray(cylinder_params)
    cyl = cyl[None] if cyl.ndim == 1 else cyl
    root, radius, top, bot = cyl[:, :2], cyl[:, 2:3], cyl[:, 3:4], cyl[:, 4:5]
    rads = np.linspace(0., 2 * np.pi, 50)
    x = root[:, 0:1] + np.cos(rads)[None] * radius
    z = root[:, 1:2] + np.sin(rads)[None] * radius
    ones = np.ones_like(x)
    caps = np.concatenate([np.stack([x, top * ones, z, ones], -1), np.stack([x, bot * ones, z, ones], -1)], axis=-2)
    pts = caps.reshape(-1, 4)
    if w2c is not None:
        pts = pts @ w2c.T
    pts = (pts @ focal_to_intrinsic_np(focal).T).reshape(cyl.shape[0], -1, 3)
    p2 = pts[..., :2] / pts[..., 2:3]
    lo, hi = p2.min(1), p2.max(1)
    if make_int:
        lo, hi = np.floor(lo).astype(np.int32), np.ceil(hi).astype(np.int32)
    off = np.array([int(W * .5), int(H * .5)] if center is None else [int(center[0]), int(center[1])])
    tl, br = lo + off, hi + off
    if scale != 1.0:
        raise NotImplementedError("box scaling is not used on the render path")
    tl[:, 0], br[:, 0] = np.clip(tl[:, 0], 0, W - 1), np.clip(br[:, 0], 0, W - 1)
    tl[:, 1], br[:, 1] = np.clip(tl[:, 1], 0, H - 1), np.clip(br[:, 1], 0, H - 1)
    return (tl[0], br[0], p2[0]) if cyl.shape[0] == 1 else (tl, br, p2)


def _axis_rotation(axis, angle):
    """float32 4x4 rotation about a coordinate axis (reference :20-42; the y rotation has the sign of sin flipped,
    i.e. it turns the camera ring clockwise seen from +y)."""
    c, s = np.cos(angle), np.sin(angle)
    m = np.eye(4, dtype=np.float32)
    a, b = {'x': (1, 2), 'y': (0, 2), 'z': (0, 1)}[axis]
    m[a, a] = m[b, b] = c
    m[a, b], m[b, a] = -s, s
    return m


def rotate_x(phi):
    return _axis_rotation('x', phi)


def rotate_y(theta):
    return _axis_rotation('y', theta)


def rotate_z(psi):
    return _axis_rotation('z', psi)


def get_smpl_l2ws(pose, rest_pose=None, scale=1., skel_type=SMPLSkeleton):
    """Forward kinematics of ONE pose: axis-angle [24,3] -> joint-local-to-world matrices [24,4,4] (reference :334-376):
    root at rest_pose[0], every other joint = parent . [R_j | rest_j - rest_parent]."""
    from .synthetic import rodrigues
    rest = (smpl_rest_pose if rest_pose is None else np.asarray(rest_pose)) * scale
    rots = rodrigues(np.asarray(pose).reshape(-1, 3))
    l2ws = []
    for j, parent in enumerate(skel_type.joint_trees):
        m = np.eye(4, dtype=np.result_type(rest.dtype, np.float32))
        m[:3, :3] = rots[j]
        m[:3, 3] = rest[j] if j == 0 else rest[j] - rest[parent]
        l2ws.append(m if j == 0 else l2ws[parent] @ m)
    return np.array(l2ws)
