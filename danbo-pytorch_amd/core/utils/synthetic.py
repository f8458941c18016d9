"""Seeded synthetic skeleton poses, cameras, rays and network weights (numpy only).

No licensed dataset ships with this repo (reference README.md:41), so tests, the golden
generator and bench.py all draw their inputs from here.  This module deliberately imports
nothing from the rest of the package so that `oracle/gen_golden.py` can load it by file
path next to the *reference's* own `core` package.

What is mirrored from the reference (file:line under /root/reference):
  * SMPL 24-joint tree and rest pose table             core/utils/skeleton_utils.py:83-110,259-282
  * forward kinematics local->world, skts = inv(l2w)   core/utils/skeleton_utils.py:334-376
  * bounding cylinder of a pose                        core/utils/skeleton_utils.py:568-618
  * pinhole rays  dir = [(i-W/2)/f, -(j-H/2)/f, -1]    core/utils/ray_utils.py:7-29
  * bullet-time camera ring                            core/load_data.py:56-71
  * parameter names / shapes / init distributions      SURVEY.md §3.2, §8(b)
"""
import math
import numpy as np

N_JOINTS = 24

JOINT_NAMES = [
    'pelvis', 'left_hip', 'right_hip', 'spine1',
    'left_knee', 'right_knee', 'spine2', 'left_ankle',
    'right_ankle', 'spine3', 'left_foot', 'right_foot',
    'neck', 'left_collar', 'right_collar', 'head',
    'left_shoulder', 'right_shoulder', 'left_elbow', 'right_elbow',
    'left_wrist', 'right_wrist', 'left_hand', 'right_hand',
]

JOINT_TREES = np.array([0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8,
                        9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21])

# SMPL template joints in the (x, y, z) convention used by the reference (data table).
SMPL_REST_POSE = np.array([
    [0.00000000e+00, 2.30003661e-09, -9.86228770e-08],
    [1.63832515e-01, -2.17391014e-01, -2.89178602e-02],
    [-1.57855421e-01, -2.14761734e-01, -2.09642015e-02],
    [-7.04505108e-03, 2.50450850e-01, -4.11837511e-02],
    [2.42021069e-01, -1.08830070e+00, -3.14962119e-02],
    [-2.47206554e-01, -1.10715497e+00, -3.06970738e-02],
    [3.95125849e-03, 5.94849110e-01, -4.03754264e-02],
    [2.12680623e-01, -1.99382353e+00, -1.29327580e-01],
    [-2.10857525e-01, -2.01218796e+00, -1.23002514e-01],
    [9.39484313e-03, 7.19204426e-01, 2.06931755e-02],
    [2.63385147e-01, -2.12222481e+00, 1.46775618e-01],
    [-2.51970559e-01, -2.12153077e+00, 1.60450473e-01],
    [3.83779174e-03, 1.22592449e+00, -9.78838727e-02],
    [1.91201791e-01, 1.00385976e+00, -6.21964522e-02],
    [-1.77145526e-01, 9.96228695e-01, -7.55542740e-02],
    [1.68482102e-02, 1.38698268e+00, 2.44048554e-02],
    [4.01985168e-01, 1.07928419e+00, -7.47655183e-02],
    [-3.98825467e-01, 1.07523870e+00, -9.96334553e-02],
    [1.00236952e+00, 1.05217218e+00, -1.35129794e-01],
    [-9.86728609e-01, 1.04515052e+00, -1.40235111e-01],
    [1.56646240e+00, 1.06961894e+00, -1.37338534e-01],
    [-1.56946480e+00, 1.05935931e+00, -1.53905824e-01],
    [1.75282109e+00, 1.04682994e+00, -1.68231070e-01],
    [-1.75758195e+00, 1.04255080e+00, -1.77773550e-01]], dtype=np.float32)


def rest_pose(scale=0.48):
    """float64 rest pose (the reference needs float64)."""
    return SMPL_REST_POSE.astype(np.float64) * scale


# ----------------------------------------------------------------------------- poses
def rodrigues(rotvec):
    """axis-angle [...,3] -> rotation matrices [...,3,3], float64."""
    rv = np.asarray(rotvec, dtype=np.float64)
    th = np.linalg.norm(rv, axis=-1, keepdims=True)
    safe = np.where(th < 1e-12, 1.0, th)
    k = rv / safe
    kx, ky, kz = k[..., 0], k[..., 1], k[..., 2]
    z = np.zeros_like(kx)
    K = np.stack([np.stack([z, -kz, ky], -1),
                  np.stack([kz, z, -kx], -1),
                  np.stack([-ky, kx, z], -1)], -2)
    s = np.sin(th)[..., None]
    c = np.cos(th)[..., None]
    eye = np.broadcast_to(np.eye(3), K.shape)
    return eye + s * K + (1.0 - c) * (K @ K)


def random_bones(n_poses, seed=0, std=0.2):
    """Axis-angle joint rotations ~ N(0, std^2) rad; one independent stream per pose."""
    out = np.zeros((n_poses, N_JOINTS, 3), dtype=np.float64)
    for i in range(n_poses):
        out[i] = np.random.default_rng(seed + i).normal(0.0, std, size=(N_JOINTS, 3))
    return out


def forward_kinematics(bones, rest):
    """bones [P,24,3] axis-angle, rest [24,3] -> l2ws, skts [P,24,4,4], kps [P,24,3] (fp64)."""
    bones = np.asarray(bones, dtype=np.float64)
    P = bones.shape[0]
    R = rodrigues(bones)
    l2ws = np.zeros((P, N_JOINTS, 4, 4), dtype=np.float64)
    for p in range(P):
        for j in range(N_JOINTS):
            loc = np.eye(4)
            loc[:3, :3] = R[p, j]
            if j == 0:
                loc[:3, 3] = rest[0]
                l2ws[p, j] = loc
            else:
                par = JOINT_TREES[j]
                loc[:3, 3] = rest[j] - rest[par]
                l2ws[p, j] = l2ws[p, par] @ loc
    skts = np.linalg.inv(l2ws)
    kps = l2ws[..., :3, 3].copy()
    return l2ws, skts, kps


def bounding_cylinder(kps, ext_scale=0.001, extend_mm=250.0, top_ratio=1.6, bot_ratio=1.1,
                      min_radius=None):
    """[P,24,3] -> [P,5] (cx, cz, radius, top, bot), head direction '-y'."""
    kps = np.asarray(kps, dtype=np.float64)
    root = kps[:, 0]
    dist = np.linalg.norm(kps[..., [0, 2]] - root[:, None, [0, 2]], axis=-1)
    flip = -1.0
    hmax = (flip * kps[..., 1]).max(-1)
    hmin = (flip * kps[..., 1]).min(-1)
    ext = extend_mm * ext_scale
    radius = dist.max(-1) + ext
    if min_radius is not None:
        radius = np.maximum(radius, min_radius)
    top = flip * (hmax + ext * top_ratio)
    bot = flip * (hmin - ext * bot_ratio)
    return np.stack([root[:, 0], root[:, 2], radius, top, bot], -1)


# ----------------------------------------------------------------------------- cameras
def _rotate_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def bullet_cameras(n_views, dist=3.0, center=(0.0, 0.0, 0.0)):
    """c2w ring around `center` at distance `dist`, camera looking down its -z axis."""
    base = np.eye(4)
    base[2, 3] = dist
    cams = []
    for i in range(n_views):
        c2w = _rotate_y(2.0 * math.pi * i / n_views) @ base
        c2w[:3, 3] += np.asarray(center)
        cams.append(c2w)
    return np.stack(cams)


def pinhole_rays(H, W, focal, c2w):
    """-> rays_o, rays_d [H*W,3] float32, row-major over (j, i); rays_d is NOT normalised."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - W * 0.5) / focal, -(j - H * 0.5) / focal, -np.ones_like(i)], -1).astype(np.float32)
    R = c2w[:3, :3].astype(np.float32)
    rays_d = (dirs[..., None, :] * R).sum(-1)
    rays_o = np.broadcast_to(c2w[:3, 3].astype(np.float32), rays_d.shape)
    return rays_o.reshape(-1, 3).copy(), rays_d.reshape(-1, 3).astype(np.float32).copy()


def ray_batch(rays_o, rays_d, near=0.0, far=1.0):
    """[R,11] = (o, d, near, far, unit view dir) as built by trainer.py:141-149."""
    vd = rays_d / np.linalg.norm(rays_d, axis=-1, keepdims=True)
    R = rays_o.shape[0]
    return np.concatenate([rays_o, rays_d, np.full((R, 1), near, np.float32),
                           np.full((R, 1), far, np.float32), vd], -1).astype(np.float32)


# ----------------------------------------------------------------------------- configs
def model_config(name):
    """Hot-path hyper-parameters of the shipped configs."""
    danbo = dict(nerf_type='danbo', W=256, D=8, skips=(4,), view_W=128, node_W=128,
                 voxel_res=16, voxel_feat=5, gcn_D=4, gcn_fc_D=1, agg_W=32, agg_D=3,
                 multires_graph=5, multires_voxel=6, multires_views=4, framecode_ch=128,
                 use_framecode=True, view_type='identity', ray_tr_type='world',
                 base_scale=0.4, density_scale=1.0, use_volume_near_far=False,
                 N_samples=48, N_importance=16, rest_scale=0.48)
    if name == 'danbo_base':
        return danbo
    if name == 'danbo_fast':
        return dict(danbo, use_volume_near_far=True, N_samples=32, N_importance=16)
    if name == 'danbo_perfcap':
        return dict(danbo, use_volume_near_far=True, N_samples=32, N_importance=16,
                    view_type='relray', ray_tr_type='root_local')
    if name == 'danbo_surreal':
        return dict(danbo, use_volume_near_far=True, N_samples=32, N_importance=16,
                    use_framecode=False, rest_scale=0.714)
    if name == 'anerf_base':
        return dict(nerf_type='nerf', W=448, D=8, skips=(4,), view_W=224, multires=7,
                    multires_views=4, multires_bones=0, framecode_ch=128, use_framecode=True,
                    view_type='relray', ray_tr_type='local', cutoff_dist=0.5, tau=20.0,
                    density_scale=1.0, use_volume_near_far=False, N_samples=48,
                    N_importance=16, rest_scale=0.48)
    raise KeyError(name)


def adjacency():
    """I + symmetric parent/child edges (gnn_backbone.py:18-34): 70 non-zeros."""
    adj = np.eye(N_JOINTS, dtype=np.float32)
    for i, p in enumerate(JOINT_TREES):
        if i != p:
            adj[i, p] = 1.0
            adj[p, i] = 1.0
    return adj


def init_axis_scale(rest, base_scale=0.4):
    """Per-bone half extents from body proportions (misc.py:675-724 + skeleton_utils.py:1515-1571)."""
    rest = np.asarray(rest, dtype=np.float64)
    names = JOINT_NAMES

    def width(key):
        idx = [i for i, n in enumerate(names) if key in n]
        return np.linalg.norm(rest[idx[0]] - rest[idx[1]])

    shoulder_w, knee_w = width('shoulder'), width('knee')
    collar_w = knee_w  # sic: the reference reads knee_width for the collar (misc.py:691)
    children = [[] for _ in range(N_JOINTS)]
    for i, p in enumerate(JOINT_TREES):
        if i != p:
            children[p].append(i)
    to_child = np.zeros(N_JOINTS)
    for i, c in enumerate(children):
        if len(c) < 1:
            to_child[i] = -1.0
            continue
        # the reference's child list of the root starts with the root itself and drops it
        # again (skeleton_utils.py:1494-1497); ours never contains it
        to_child[i] = np.sqrt(((rest[i:i + 1] - rest[c]) ** 2).sum(-1)).mean()
    torso = [i for i, n in enumerate(names) if any(k in n for k in ('shoulder', 'spine', 'collar', 'neck', 'pelvis'))]
    arms = [i for i, n in enumerate(names) if any(k in n for k in ('elbow', 'wrist', 'hand'))]
    legs = [i for i, n in enumerate(names) if any(k in n for k in ('hip', 'knee', 'ankle', 'foot'))]
    head = [i for i, n in enumerate(names) if 'head' in n]
    x = np.full(N_JOINTS, base_scale, dtype=np.float32)
    y = np.full(N_JOINTS, base_scale, dtype=np.float32)
    x[legs] = y[legs] = np.float32(knee_w * 0.5)
    x[torso] = y[torso] = np.float32(shoulder_w * 0.70)
    x[head] = y[head] = np.float32(shoulder_w * 0.60)
    x[arms] = y[arms] = np.float32(collar_w * 0.60)
    z = to_child.astype(np.float32) * np.float32(0.8)
    z[z < 0] = z.max()
    z[head] = z.max() * np.float32(1.1)
    return np.stack([x, y, z], -1).astype(np.float32)


def pe_dim(d, L):
    return d * (1 + 2 * L)


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def make_state_dict(cfg, seed=0, n_framecodes=100, rest=None, lively=True):
    """Seeded parameters with the reference's names, shapes and init distributions.

    `lively=True` additionally rescales a few layers so that a randomly initialised model
    produces a structured image (non-trivial volumes, assignment logits, densities and
    colours) -- otherwise parity tests would compare near-constant tensors.  The result is
    loaded into the reference model by `oracle/gen_golden.py` and into ours by the tests,
    so no weight file is committed.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    W, D, view_W = cfg['W'], cfg['D'], cfg['view_W']
    if cfg['nerf_type'] == 'danbo':
        in_ch = pe_dim(cfg['voxel_feat'] * 3, cfg['multires_voxel'])
        view_ch = pe_dim(3, cfg['multires_views'])
    else:
        in_ch = pe_dim(N_JOINTS, cfg['multires']) + N_JOINTS * 3
        view_ch = pe_dim(N_JOINTS * 3, cfg['multires_views'])
    fc = cfg['framecode_ch'] if cfg['use_framecode'] else 0

    def linear(name, fin, fout, gain=1.0):
        b = 1.0 / math.sqrt(fin)
        sd[name + '.weight'] = _uniform(rng, (fout, fin), b) * np.float32(gain)
        sd[name + '.bias'] = _uniform(rng, (fout,), b)

    g = 1.15 if lively else 1.0
    linear('pts_linears.0', in_ch, W, g)
    for i in range(D - 1):
        fin = W + in_ch if i in cfg['skips'] else W
        linear(f'pts_linears.{i + 1}', fin, W, g)
    linear('alpha_linear', W, 1, 4.0 if lively else 1.0)
    linear('views_linears.0', view_ch + fc + view_W * 2, view_W, g)
    linear('feature_linear', W, view_W * 2, g)
    linear('rgb_linear', view_W, 3, 12.0 if lively else 1.0)
    if cfg['use_framecode']:
        std = math.sqrt(2.0 / (n_framecodes + fc))
        sd['framecodes.codes.weight'] = (rng.standard_normal((n_framecodes, fc)) * std).astype(np.float32)

    if cfg['nerf_type'] == 'danbo':
        adj = adjacency()
        J = N_JOINTS

        def adj_w():
            w = adj * np.clip(0.05 + (rng.uniform(size=(J, J)) - 0.5) * 0.1, 0.01, 1.0)
            w[np.arange(J), np.arange(J)] = 1.0
            return w.astype(np.float32)[None]

        def plin(name, fin, fout, bias, gain=1.0):
            b = 1.0 / math.sqrt(fin)
            sd[name + '.weight'] = _uniform(rng, (J, fin, fout), b) * np.float32(gain)
            if bias:
                sd[name.replace('.lin', '') + '.bias'] = np.zeros((1, J, fout), np.float32)

        def gcn(prefix, fin, fout, gain=1.0):
            sd[prefix + '.bias'] = np.zeros((fout,), np.float32)
            sd[prefix + '.adj_w'] = adj_w()
            sd[prefix + '.adj'] = adj[None].copy()
            plin(prefix + '.lin', fin, fout, False, gain)

        nW = cfg['node_W']
        g_in = pe_dim(6, cfg['multires_graph'])
        vol = cfg['voxel_res'] * cfg['voxel_feat'] * 3
        if rest is None:
            rest = rest_pose(cfg['rest_scale'])
        sd['graph_net.axis_scale'] = init_axis_scale(rest, cfg['base_scale'])
        n_gcn = cfg['gcn_D'] - cfg['gcn_fc_D'] - 1
        li = 0
        gcn(f'graph_net.layers.{li}', g_in, nW, g); li += 1
        for _ in range(n_gcn - 1):
            gcn(f'graph_net.layers.{li}', nW, nW, g); li += 1
        for _ in range(cfg['gcn_fc_D']):
            plin(f'graph_net.layers.{li}', nW, nW, True, g); li += 1
        plin(f"graph_net.layers.{li}", nW, vol, True, 2.5 if lively else 1.0)
        aW = cfg['agg_W']
        gcn('prob_linears.layers.0', cfg['voxel_feat'] * 3, aW, 2.0 if lively else 1.0)
        plin('prob_linears.layers.1', aW, aW, True, 2.0 if lively else 1.0)
        plin('prob_linears.layers.2', aW, 1, True, 3.0 if lively else 1.0)
        if lively:
            # biases of the per-bone layers are zero-initialised in the reference; perturb them
            # so a bias-indexing bug cannot hide
            for k in list(sd):
                if k.endswith('.bias') and sd[k].ndim == 3:
                    sd[k] = (rng.standard_normal(sd[k].shape) * 0.05).astype(np.float32)
            for k in ('graph_net.layers.0.bias', 'graph_net.layers.1.bias', 'prob_linears.layers.0.bias'):
                if k in sd:
                    sd[k] = (rng.standard_normal(sd[k].shape) * 0.05).astype(np.float32)
        if lively:
            _calibrate_density(sd, cfg, rng)
    else:
        sd['pe_fn.cutoff_dist'] = np.full((N_JOINTS,), cfg['cutoff_dist'], np.float32)
        sd['dirs_pe_fn.cutoff_dist'] = np.full((N_JOINTS,), cfg['cutoff_dist'], np.float32)
        sd['pe_fn.tau'] = np.array(cfg['tau'], np.float32)
        sd['dirs_pe_fn.tau'] = np.array(cfg['tau'], np.float32)
        if lively:
            _calibrate_anerf(sd, cfg, rng)
    return sd


def _pe64(x, L):
    outs = [x]
    for l in range(L):
        outs += [np.sin(x * 2.0 ** l), np.cos(x * 2.0 ** l)]
    return np.concatenate(outs, -1)


def _trunk64(sd, x, D, skips):
    h = x
    for i in range(D):
        h = np.maximum(h @ sd[f'pts_linears.{i}.weight'].astype(np.float64).T
                       + sd[f'pts_linears.{i}.bias'].astype(np.float64), 0.0)
        if i in skips:
            h = np.concatenate([x, h], -1)
    return h


def _calibrate_density(sd, cfg, rng):
    """Re-aim alpha_linear so a random-init model looks 'trained': empty space (blended
    feature h = 0) gets raw density -2 (transparent) and a probe set of in-body features
    +6 on average.  The weight is the random init plus a component along the direction in
    which the trunk output actually moves between "empty" and "body" (a plain rescale of
    the random weight needs a gain of 100+ and cancels catastrophically in fp32).
    Pure function of the seeded weights, so reference and build agree."""
    d = cfg['voxel_feat'] * 3
    probe = rng.standard_normal((256, d))
    x = _pe64(np.concatenate([np.zeros((1, d)), probe], 0), cfg['multires_voxel'])
    h = _trunk64(sd, x, cfg['D'], cfg['skips'])
    u = h[1:].mean(0) - h[0]
    w = sd['alpha_linear.weight'].astype(np.float64)[0] * 2.0
    gap = float(u @ w)
    w = w + (8.0 - gap) * u / float(u @ u)
    sd['alpha_linear.weight'] = w[None].astype(np.float32)
    sd['alpha_linear.bias'] = np.array([-2.0 - float(h[0] @ w)], dtype=np.float32)


def _cutoff_pe64(v, cutoff, tau, L):
    """A-NeRF distance encoding (cut_to_dist, cutoff_shift, cutoff_inputs) in float64: [n,24] -> [n,(1+2L)*24]."""
    inp = cutoff - v
    sh = inp * (2.0 / cutoff) - 1.0
    w = 1.0 - 1.0 / (1.0 + np.exp(-tau * (v - cutoff)))
    blocks = [inp]
    for l in range(L):
        blocks += [np.sin(sh * 2.0 ** l), np.cos(sh * 2.0 ** l)]
    return (np.stack(blocks, -2) * w[..., None, :]).reshape(v.shape[0], -1)


def _calibrate_anerf(sd, cfg, rng):
    """The A-NeRF counterpart of `_calibrate_density`: samples far from every joint (all cutoff weights
    ~0) get raw density -2, samples with a few joints within 5-25 cm get +6 on average."""
    n = 256
    unit = rng.standard_normal((2 * n, N_JOINTS, 3))
    unit /= np.linalg.norm(unit, axis=-1, keepdims=True)
    v_far = rng.uniform(0.9, 1.6, size=(n, N_JOINTS))
    v_near = rng.uniform(0.45, 1.2, size=(n, N_JOINTS))
    for i in range(n):
        j = rng.choice(N_JOINTS, size=3, replace=False)
        v_near[i, j] = rng.uniform(0.05, 0.25, size=3)
    v = np.concatenate([v_far, v_near], 0)
    x = np.concatenate([_cutoff_pe64(v, cfg['cutoff_dist'], cfg['tau'], cfg['multires']), unit.reshape(2 * n, -1)], -1)
    h = _trunk64(sd, x, cfg['D'], cfg['skips'])
    h_far, h_near = h[:n].mean(0), h[n:].mean(0)
    u = h_near - h_far
    w = sd['alpha_linear.weight'].astype(np.float64)[0] * 0.5
    w = w + (8.0 - float(u @ w)) * u / float(u @ u)
    sd['alpha_linear.weight'] = w[None].astype(np.float32)
    sd['alpha_linear.bias'] = np.array([-2.0 - float(h_far @ w)], dtype=np.float32)


# ----------------------------------------------------------------------------- scenes
def make_scene(n_poses=1, H=64, W=64, n_views=1, pose_seed=0, rest_scale=0.48, cam_dist=3.0,
               min_radius=None, focal=None):
    """A complete synthetic input set: per-pose skeleton tensors + per-view full-grid rays."""
    rest = rest_pose(rest_scale)
    bones = random_bones(n_poses, seed=pose_seed)
    l2ws, skts, kps = forward_kinematics(bones, rest)
    cyls = bounding_cylinder(kps, min_radius=min_radius)
    focal = 1.25 * H if focal is None else focal
    cams = bullet_cameras(n_views, dist=cam_dist, center=kps[0, 0])
    rays = [pinhole_rays(H, W, focal, c) for c in cams]
    return dict(rest_pose=rest, bones=bones.astype(np.float32), skts=skts.astype(np.float32),
                kps=kps.astype(np.float32), cyls=cyls.astype(np.float32), cams=cams,
                rays=rays, H=H, W=W, focal=focal)
