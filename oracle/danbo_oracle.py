"""CPU oracle: numpy restatement of the DANBO / A-NeRF rendering hot path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product package
(`danbo-pytorch_amd/`) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  Every stage below is checked in `tests/test_oracle_golden.py`
against golden vectors captured from the imported reference implementation
(`oracle/gen_golden.py`, run in the build container; vectors under `tests/golden/`).
One boundary is unpinned *by the reference itself*: axis-angle -> rotation comes from the
un-vendored third-party `pytorch3d` (README.md:31-33, ~v0.6); `axis_angle_to_matrix` below
restates its published algorithm and is cross-checked against scipy.

All arithmetic is float32 with an explicit, documented operation order.  The two chained
rigid transforms and the in-volume test use only correctly-rounded elementwise float32
mul/add (no FMA contraction), so the HIP kernels -- which use `__fmul_rn/__fadd_rn` in the
same order -- reproduce the in-volume mask BIT-EXACTLY.  Everything downstream of a GEMM
or a transcendental is compared with a stated tolerance.

Citations are file:line under /root/reference.
"""
import math
import numpy as np

F32 = np.float32
J = 24
JOINT_TREES = np.array([0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8,
                        9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21])


def torch_norm(x, keepdims=False):
    """torch.norm(x, p=2, dim=-1) on fp32: ATen's reduction step is acc + x*x contracted to ONE fma (NormTwoOps), so the result is
    sqrt(fma(x2, x2, fma(x1, x1, x0*x0))) -- pinned on the imported reference: bit-equal to torch on every row of the box-bound, cylinder
    and direction norms tried, where the separately rounded sum differs on ~5 % of them.  The fma is emulated in float64 (the product
    of two fp32 values is exact there; the double rounding of the sum can only matter on an exact tie of the 53-bit sum)."""
    x64 = np.asarray(x, dtype=F32).astype(np.float64)
    acc = (x64[..., 0] * x64[..., 0]).astype(F32)
    for k in range(1, x64.shape[-1]):
        acc = (x64[..., k] * x64[..., k] + acc.astype(np.float64)).astype(F32)
    with np.errstate(invalid='ignore'):
        n = np.sqrt(acc).astype(F32)
    return n[..., None] if keepdims else n


# =============================================================================
# a1 / a2  skeleton constants
# =============================================================================
def adjacency():
    """skeleton_to_graph, core/networks/gnn_backbone.py:18-34."""
    adj = np.eye(J, dtype=F32)
    for i, p in enumerate(JOINT_TREES):
        if i != p:
            adj[i, p] = adj[p, i] = 1.0
    return adj


def _arccos_safe(a):
    return np.arccos(np.clip(a, -1. + 1e-8, 1. - 1e-8))  # skeleton_utils.py:50-52


def _rot_y(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=F32)


def _rot_x(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=F32)


def axis_aligned_rotation(vec):
    """get_axis_aligned_rotation, core/utils/skeleton_utils.py:544-566 (returns 4x4)."""
    vec_xz = vec[[0, 2]] / np.linalg.norm(vec[[0, 2]])
    theta = _arccos_safe(vec_xz[-1]) * np.sign(vec_xz[0])
    rot_y = _rot_y(theta)
    rotated_y = rot_y[:3, :3] @ vec
    vec_yz = rotated_y[1:3] / np.linalg.norm(rotated_y[1:3])
    psi = _arccos_safe(vec_yz[-1]) * np.sign(vec_yz[0])
    rot_x = _rot_x(psi)
    rot = np.linalg.inv(rot_x @ rot_y)
    return rot.T


def bone_align_transforms(rest_pose):
    """RayCaster.init_bone_align_transforms ('align'), core/raycasters.py:548-591 -> [24,4,4] f32."""
    rest = np.asarray(rest_pose).reshape(J, 3)
    children = [[] for _ in range(J)]
    for i, p in enumerate(JOINT_TREES):
        children[p].append(i)  # the root lists itself -> it has 4 "children" -> identity
    T = np.tile(np.eye(4, dtype=F32), (J, 1, 1))
    for parent, c in enumerate(children):
        if len(c) != 1:
            continue
        d = rest[c[0]] - rest[parent]
        rot = axis_aligned_rotation(d).copy()
        rot[:3, -1] = -0.5 * np.linalg.norm(d) * np.array([0., 0., 1.], dtype=F32)
        T[parent] = rot.astype(F32)  # torch.tensor(float64) assigned into a float32 tensor
    return T


# =============================================================================
# a3 / a4  near / far
# =============================================================================
def near_far_cylinder(rays_o, rays_d, cyl, near, far, chunk=None):
    """get_near_far_in_cylinder, core/utils/ray_utils.py:294-346.

    rays_o, rays_d [R,3]; cyl [R,5]; near, far [R,1].  Rays that miss the cylinder (Q = NaN)
    are back-filled with the nan-mean of their *chunk* (the reference is called once per
    `chunk` rays, core/trainer.py:75-90); `chunk=None` treats the whole input as one chunk.
    """
    R = rays_o.shape[0]
    if chunk is not None and R > chunk:
        outs = [near_far_cylinder(rays_o[i:i + chunk], rays_d[i:i + chunk], cyl[i:i + chunk],
                                  near[i:i + chunk], far[i:i + chunk]) for i in range(0, R, chunk)]
        return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])
    g = [0, 2]
    r_near = (rays_o + rays_d * near)[:, g]
    r_far = (rays_o + rays_d * far)[:, g]
    radius = cyl[:, 2:3]
    center = cyl[:, :2]
    nc = center - r_near
    nf = r_far - r_near
    nf_norm = torch_norm(nf)
    scale = torch_norm(rays_d[:, g])[:, None]
    cross = nc[:, 0] * nf[:, 1] - nc[:, 1] * nf[:, 0]
    dist = (np.abs(cross) / nf_norm)[:, None]
    with np.errstate(invalid='ignore'):
        Q = np.sqrt(radius * radius - dist * dist).astype(F32)
    K = ((nc * nf).sum(-1, dtype=F32) / nf_norm)[:, None].astype(F32)
    with np.errstate(invalid='ignore'):
        mask = (Q < K).astype(F32)
        new_near = (near + mask * (K - Q) / scale).astype(F32)
        new_far = (near + (K + Q) / scale).astype(F32)
    if np.isnan(new_near).any():
        idx = np.isnan(Q[:, 0])
        with np.errstate(all='ignore'):
            avg_near = np.nanmean(new_near) if (~np.isnan(new_near)).any() else np.nan
            avg_far = np.nanmean(new_far) if (~np.isnan(new_far)).any() else np.nan
        new_near[idx, 0] = F32(avg_near) if not np.isnan(avg_near) else near[idx, 0]
        new_far[idx, 0] = F32(avg_far) if not np.isnan(avg_far) else far[idx, 0]
    return new_near, new_far


def cylinder_miss_mask(rays_o, rays_d, cyl, near, far):
    """True where the ray misses the cylinder (Q = NaN before the back-fill)."""
    g = [0, 2]
    r_near = (rays_o + rays_d * near)[:, g]
    nf = (rays_o + rays_d * far)[:, g] - r_near
    nc = cyl[:, :2] - r_near
    dist = np.abs(nc[:, 0] * nf[:, 1] - nc[:, 1] * nf[:, 0]) / np.sqrt((nf * nf).sum(-1))
    return dist > cyl[:, 2]


def near_far_boxes(rays_o, rays_d, skts, align, axis_scale, near, far, bound=1.3, eps=1e-4):
    """GraphCaster.get_near_far + get_ray_box_intersections restated as an fp64 slab test.

    core/raycasters.py:648-707, core/utils/ray_utils.py:383-417.  Per bone: the ray is moved
    into the aligned bone frame, divided by |axis_scale|, intersected with the +-`bound`
    box (6 plane hits, kept when inside the box +eps); a bone counts only if EXACTLY two of
    the six hits are in bounds; the step of a hit is |p - o| / |d| in the unscaled bone
    frame; near/far = min/max over counted bones, else the cylinder values are kept.
    `bound` is always training_bound=1.3 (raycasters.py:671-675).
    """
    R = rays_o.shape[0]
    # torch's batched 3x3 @ 3x1 products sum k = 0, 1, 2 in order with separately rounded mul / add (bit-equal to the reference's
    # rays_ot / rays_dt on 98 304 rows; numpy's matmul is not), i.e. the arithmetic of _affine_unfused
    zero = np.zeros((1, J, 3, 1), F32)
    ot = _affine_unfused(skts[:, :, :3, :], rays_o[:, None, :])
    dt = _affine_unfused(np.concatenate([skts[:, :, :3, :3], np.broadcast_to(zero, skts[:, :, :3, :1].shape)], -1), rays_d[:, None, :])
    ot = _affine_unfused(align[None, :, :3, :], ot)
    dt = _affine_unfused(np.concatenate([align[None, :, :3, :3], zero], -1), dt)
    sc = np.abs(axis_scale)[None].astype(F32)
    o_s = (ot / sc).astype(F32)
    d_s = (dt / sc).astype(F32)
    # bound_range * torch.ones(...) is a FLOAT32 tensor (1.3f = 1.2999999523...) before its .double() (ray_utils.py:394-397)
    bounds = np.stack([-F32(bound) * np.ones(3, F32), F32(bound) * np.ones(3, F32)], 0)[None, None]  # [1,1,2,3]
    with np.errstate(divide='ignore', invalid='ignore'):
        t = (bounds.astype(np.float64) - o_s[:, :, None, :].astype(np.float64)) / d_s[:, :, None, :].astype(np.float64)
        t = t.reshape(R, J, 6, 1)
        p = (t * d_s[:, :, None, :].astype(np.float64) + o_s[:, :, None, :].astype(np.float64)).astype(F32)
        lim = F32(bound + eps)
        p_valid = np.all((p <= lim) & (p >= -lim), axis=-1)  # [R,J,6]
    v_valid = p_valid.sum(-1) == 2
    p_unscaled = p * sc[:, :, None, :]
    diff = p_unscaled - ot[:, :, None, :]
    with np.errstate(invalid='ignore'):
        steps = (torch_norm(diff) / torch_norm(dt)[..., None]).astype(F32)
    big = F32(100000.0)
    s_min = np.where(p_valid, steps, np.inf).min(-1)
    s_max = np.where(p_valid, steps, -np.inf).max(-1)
    v_near = np.where(v_valid, s_min, big).min(-1).astype(F32)
    v_far = np.where(v_valid, s_max, -big).max(-1).astype(F32)
    ray_valid = v_valid.any(-1)
    new_near, new_far = near.copy(), far.copy()
    new_near[ray_valid, 0] = v_near[ray_valid]
    new_far[ray_valid, 0] = v_far[ray_valid]
    return new_near, new_far


# =============================================================================
# a5  sampling along the ray
# =============================================================================
def coarse_z(near, far, S, t_rand=None):
    """sample_from_lineseg (lindisp=False), core/utils/ray_utils.py:206-253.  t_rand [R,S]: the uniforms of the stratified
    branch (perturb > 0, :233-248: one sample per interval between the mid-points of the even depths); None: perturb = 0."""
    t = torch_linspace01(S)
    z = (near * (F32(1.) - t) + far * t).astype(F32)
    if t_rand is None:
        return z
    mids = (F32(.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
    upper = np.concatenate([mids, z[:, -1:]], -1)
    lower = np.concatenate([z[:, :1], mids], -1)
    return (lower + ((upper - lower).astype(F32) * np.asarray(t_rand, dtype=F32)).astype(F32)).astype(F32)


def torch_linspace01(n):
    """torch.linspace(0,1,n) in float32: symmetric evaluation start+i*step / end-(n-1-i)*step."""
    if n == 1:
        return np.zeros(1, F32)
    step = F32(1.0) / F32(n - 1)
    i = np.arange(n)
    lo = (F32(0.) + i.astype(F32) * step).astype(F32)
    hi = (F32(1.) - (n - 1 - i).astype(F32) * step).astype(F32)
    return np.where(i < n // 2, lo, hi).astype(F32)


def sample_points(rays_o, rays_d, z):
    """pts = o + d * z, two roundings (core/raycasters.py:463)."""
    return (rays_o[:, None, :] + (rays_d[:, None, :] * z[:, :, None]).astype(F32)).astype(F32)


def sample_pdf_det(bins, weights, n, u=None):
    """sample_pdf, core/utils/ray_utils.py:159-203 (det=True: u = linspace; else caller's uniforms)."""
    w = (weights + F32(1e-5)).astype(F32)
    pdf = (w / w.sum(-1, keepdims=True, dtype=F32)).astype(F32)
    cdf = np.cumsum(pdf, -1, dtype=F32)
    cdf = np.concatenate([np.zeros_like(cdf[:, :1]), cdf], -1)
    u = np.broadcast_to(torch_linspace01(n), (cdf.shape[0], n)) if u is None else np.asarray(u, dtype=F32)
    inds = np.stack([np.searchsorted(cdf[r], u[r], side='right') for r in range(cdf.shape[0])])
    below = np.maximum(0, inds - 1)
    above = np.minimum(cdf.shape[-1] - 1, inds)
    cg0 = np.take_along_axis(cdf, below, 1)
    cg1 = np.take_along_axis(cdf, above, 1)
    bg0 = np.take_along_axis(bins, below, 1)
    bg1 = np.take_along_axis(bins, above, 1)
    denom = (cg1 - cg0).astype(F32)
    denom = np.where(denom < F32(1e-5), F32(1.), denom)
    t = ((u - cg0) / denom).astype(F32)
    return (bg0 + t * (bg1 - bg0)).astype(F32)


def importance_z(z, weights, n_importance, alpha_base=0.01, u=None):
    """isample_from_lineseg(det=True, is_only=True), core/utils/ray_utils.py:257-291.

    Returns sorted z [R,S+Sf], z_samples [R,Sf], sorted_idxs [R,S+Sf] (stable sort)."""
    mid = (F32(.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
    w_l, w_k, w_u = weights[:, 0:-2], weights[:, 1:-1], weights[:, 2:]
    dw = (F32(0.5) * (np.maximum(w_l, w_k) + np.maximum(w_k, w_u)) + F32(alpha_base)).astype(F32)
    zs = sample_pdf_det(mid, dw, n_importance, u)
    cat = np.concatenate([z, zs], -1)
    idx = np.argsort(cat, -1, kind='stable')
    return np.take_along_axis(cat, idx, -1), zs, idx


# =============================================================================
# a7 / a8 / a12  bone-local transform, in-volume mask, factorised gather
# =============================================================================
def _affine_unfused(M, p):
    """((m0*x + m1*y) + m2*z) + m3 per output row, each op rounded to fp32 (no FMA).

    M [...,3,4] broadcastable against p [...,3]."""
    x, y, z = p[..., 0:1], p[..., 1:2], p[..., 2:3]
    t0 = (M[..., :, 0] * x).astype(F32)
    t1 = (M[..., :, 1] * y).astype(F32)
    t2 = (M[..., :, 2] * z).astype(F32)
    s = (t0 + t1).astype(F32)
    s = (s + t2).astype(F32)
    return (s + M[..., :, 3]).astype(F32)


def bone_local(pts, skts, align):
    """transform_batch_pts + bone alignment (core/encoders.py:288-303, 442-444).

    pts [R,S,3]; skts [R,24,4,4]; align [24,4,4]  ->  pts_t [R,S,24,3] float32."""
    p = pts[:, :, None, :]
    pl = _affine_unfused(skts[:, None, :, :3, :], p)
    return _affine_unfused(align[None, None, :, :3, :], pl)


def in_volume(pts_t, axis_scale):
    """invalid = any_k |x_k| > 1 with x = pts_t/|scale| (gnn_backbone.py:802,808).

    Returns x [R,S,24,3] (true fp32 division) and valid [R,S,24] bool.  Because fp32 division
    is correctly rounded, fl(|p|/s) > 1  <=>  |p| > s, which is the form the HIP kernel tests."""
    sc = np.abs(axis_scale).astype(F32)[None, None]
    x = (pts_t / sc).astype(F32)
    valid = ~(np.abs(x) > F32(1.)).any(-1)
    return x, valid


def window(x):
    """exp(-2 * sum_k x_k^6) (gnn_backbone.py:803-804)."""
    x2 = (x * x).astype(F32)
    x6 = (x2 * x2 * x2).astype(F32)
    with np.errstate(over='ignore', under='ignore'):
        return np.exp((F32(-2.) * x6.sum(-1, dtype=F32)).astype(F32)).astype(F32)


def factorised_gather(volumes, x, pose_of_ray, voxel_feat=5, voxel_res=16):
    """factorize_grid_sample + 'cat' construct (misc.py:331-351, gnn_backbone.py:810-826).

    volumes [G,24,F*res*3] laid out f*(res*3) + r*3 + axis; x [R,S,24,3] -> feat [R,S,24,3F]
    ordered f*3 + axis.  1-D linear interpolation with zero padding, align_corners=False:
    cell coordinate i = ((x+1)*res - 1)/2."""
    R, S = x.shape[:2]
    Fc, res = voxel_feat, voxel_res
    vol = volumes.reshape(-1, J, Fc, res, 3)[pose_of_ray]  # [R,24,F,res,3]
    iy = (((x + F32(1.)) * F32(res) - F32(1.)) / F32(2.)).astype(F32)  # [R,S,24,3]
    y0 = np.floor(iy)
    w1 = (iy - y0).astype(F32)
    w0 = (F32(1.) - w1).astype(F32)
    y0 = y0.astype(np.int64)
    y1 = y0 + 1
    # the column selector lands (to 1 ulp) on integer columns: -2/3, 0, 2/3 -> 0, 1, 2
    out = np.zeros((R, S, J, Fc, 3), dtype=F32)
    rr = np.arange(R)[:, None, None]
    jj = np.arange(J)[None, None, :]
    for k in range(3):
        for (yy, ww) in ((y0[..., k], w0[..., k]), (y1[..., k], w1[..., k])):
            ok = (yy >= 0) & (yy < res)
            yc = np.clip(yy, 0, res - 1)
            v = vol[rr, jj, :, yc, k]  # [R,S,24,F]
            out[..., k] += np.where(ok[..., None], v * ww[..., None], F32(0.))
    return out.reshape(R, S, J, Fc * 3)


# =============================================================================
# a9 / a10 / a11  pose -> rot6d -> PE -> skeleton GNN -> factorised volumes
# =============================================================================
def axis_angle_to_matrix(aa):
    """pytorch3d.transforms.axis_angle_to_matrix (v0.6): via quaternion, Taylor for small angles."""
    aa = aa.astype(F32)
    ang = torch_norm(aa, keepdims=True)
    half = (ang * F32(0.5)).astype(F32)
    small = np.abs(ang) < 1e-6
    with np.errstate(divide='ignore', invalid='ignore'):
        s = np.where(small, F32(0.5) - (ang * ang) / F32(48.), np.sin(half) / ang).astype(F32)
    q = np.concatenate([np.cos(half), aa * s], -1).astype(F32)
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two_s = (F32(2.) / (q * q).sum(-1, dtype=F32)).astype(F32)
    m = np.stack([
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return m.reshape(aa.shape[:-1] + (3, 3)).astype(F32)


def rot6d(bones):
    """axisang_to_rot6d: first two columns, flattened row-major [r00,r01,r10,r11,r20,r21]
    (core/utils/skeleton_utils.py:408-418)."""
    m = axis_angle_to_matrix(bones)
    return m[..., :3, :2].reshape(bones.shape[:-1] + (6,))


def positional_encoding(x, L):
    """Embedder._embed, core/cutoff_embedder.py:62-73: [x, sin(2^0 x), cos(2^0 x), ...]."""
    outs = [x]
    for l in range(L):
        f = F32(2.0 ** l)
        xf = (x * f).astype(F32)
        outs += [np.sin(xf).astype(F32), np.cos(xf).astype(F32)]
    return np.concatenate(outs, -1)


def _adjw(sd, prefix):
    """DenseWGCN.get_adjw: adj_w * adj (gnn_backbone.py:225-247)."""
    return (sd[prefix + '.adj_w'][0] * sd[prefix + '.adj'][0]).astype(F32)


def _plin(x, w, b=None):
    """ParallelLinear.forward einsum('bkl,klj->bkj') (misc.py:174-183)."""
    out = np.einsum('bkl,klj->bkj', x, w).astype(F32)
    return out if b is None else (out + b).astype(F32)


def pose_volumes(sd, bones, multires_graph=5):
    """encode_graph_inputs + graph PE + FactorizeGNN/BodyGNN.forward.

    core/encoders.py:460-473; gnn_backbone.py:683-704 with layers from :651-681.
    bones [G,24,3] -> volumes [G,24,240].  Quirk reproduced: skip_gcn=False makes
    `i == skip_gcn` true at i=0, so the first layer's pre-activation is doubled."""
    w = positional_encoding(rot6d(bones), multires_graph)  # [G,24,66]
    n = w.copy()
    n[:, 0, :] = 0  # mask_root (gnn_backbone.py:687-688)
    keys = sorted({int(k.split('.')[2]) for k in sd if k.startswith('graph_net.layers.')})
    for i in keys:
        p = f'graph_net.layers.{i}'
        if p + '.lin.weight' in sd:  # DensePNGCN
            out = _plin(n, sd[p + '.lin.weight'])
            out = np.matmul(_adjw(sd, p)[None], out).astype(F32)
            out = (out + sd[p + '.bias']).astype(F32)
        else:  # ParallelLinear
            out = _plin(n, sd[p + '.weight'], sd[p + '.bias'])
        if i == 0:
            out = (out + out).astype(F32)
        if i != keys[-1]:
            out = np.maximum(out, F32(0.))
        n = out
    return n


# =============================================================================
# a13 / a14  per-sample bone assignment + blend
# =============================================================================
def assignment_logits(sd, part_feat):
    """MixGNN forward (gnn_backbone.py:567-591,602-629): [M,24,15] -> [M,24]."""
    p0 = 'prob_linears.layers.0'
    y = _plin(part_feat, sd[p0 + '.lin.weight'])
    y = np.matmul(_adjw(sd, p0)[None], y).astype(F32)
    y = np.maximum((y + sd[p0 + '.bias']).astype(F32), F32(0.))
    y = np.maximum(_plin(y, sd['prob_linears.layers.1.weight'], sd['prob_linears.layers.1.bias']), F32(0.))
    a = _plin(y, sd['prob_linears.layers.2.weight'], sd['prob_linears.layers.2.bias'])
    return a[..., 0]


def sigmoid(x):
    with np.errstate(over='ignore'):
        return (F32(1.) / (F32(1.) + np.exp(-x).astype(F32))).astype(F32)


def blend(part_feat, logits, valid):
    """DANBO.sigmoid + blend (danbo.py:299-300,406-415): p=(s(a)*1.002-0.001)*valid."""
    p = ((sigmoid(logits) * F32(1.002) - F32(0.001)) * valid.astype(F32)).astype(F32)
    h = (part_feat * p[..., None]).sum(-2, dtype=F32).astype(F32)
    return p, h


# =============================================================================
# a16 / a17  view features and the density / colour MLP
# =============================================================================
def view_dirs(cfg, rays_d, skts):
    """ray transform + view encoder (core/encoders.py:179-189,495-512,570-578,774-795) -> [R,3]."""
    if cfg['ray_tr_type'] == 'world':
        d = rays_d
    elif cfg['ray_tr_type'] == 'root_local':
        d = (skts[:, 0, :3, :3] @ rays_d[:, :, None])[..., 0].astype(F32)
    else:
        raise NotImplementedError(cfg['ray_tr_type'])
    if cfg['view_type'] == 'relray':
        n = torch_norm(d, keepdims=True)
        d = (d / np.maximum(n, F32(1e-12))).astype(F32)
    return d


def frame_codes(sd, cam_idxs, R):
    """Optcodes.forward in eval (core/networks/embedding.py:17-39): idx<0 -> mean code."""
    codes = sd['framecodes.codes.weight']
    if cam_idxs is None or np.max(cam_idxs) < 0:
        return np.broadcast_to(codes.mean(0, keepdims=True, dtype=F32), (R, codes.shape[1])).astype(F32)
    idx = np.asarray(cam_idxs).reshape(-1).astype(np.int64)
    return codes[idx]


def linear(sd, name, x):
    return (x @ sd[name + '.weight'].T + sd[name + '.bias']).astype(F32)


def mlp(sd, density_in, view_in, D=8, skips=(4,)):
    """NeRF.inference (core/networks/nerf.py:176-209): -> raw [M,4] = (rgb logits, density)."""
    h = density_in
    for i in range(D):
        h = np.maximum(linear(sd, f'pts_linears.{i}', h), F32(0.))
        if i in skips:
            h = np.concatenate([density_in, h], -1)
    alpha = linear(sd, 'alpha_linear', h)
    feat = linear(sd, 'feature_linear', h)
    hv = np.maximum(linear(sd, 'views_linears.0', np.concatenate([feat, view_in], -1)), F32(0.))
    rgb = linear(sd, 'rgb_linear', hv)
    return np.concatenate([rgb, alpha], -1).astype(F32)


# =============================================================================
# a18  alpha compositing
# =============================================================================
def composite(raw, z, rays_d, B=1.0, noise=None):
    """NeRF.raw2outputs with relu density (core/networks/nerf.py:281-347)."""
    d = (z[:, 1:] - z[:, :-1]).astype(F32)
    d = np.concatenate([d, np.full_like(d[:, :1], 1e10)], -1)
    d = (d * torch_norm(rays_d)[:, None]).astype(F32)
    rgb = (sigmoid(raw[..., :3]) * F32(1.002) - F32(0.001)).astype(F32)
    s = (raw[..., 3] / F32(B)).astype(F32)
    if noise is not None:
        s = (s + noise).astype(F32)
    with np.errstate(over='ignore', under='ignore'):
        alpha = (F32(1.) - np.exp(-(np.maximum(s, F32(0.)) * d).astype(F32)).astype(F32)).astype(F32)
    T = np.cumprod(np.concatenate([np.ones_like(alpha[:, :1]), (F32(1.) - alpha + F32(1e-10)).astype(F32)], -1),
                   -1, dtype=F32)[:, :-1]
    w = (alpha * T).astype(F32)
    rgb_map = (w[..., None] * rgb).sum(-2, dtype=F32)
    depth = (w * z).sum(-1, dtype=F32)
    acc = w.sum(-1, dtype=F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        disp = (F32(1.) / np.maximum(F32(1e-10), depth / (acc + F32(1e-10)))).astype(F32)
    disp = np.where(np.isclose(acc, 0.), F32(0.), disp).astype(F32)
    return dict(rgb_map=rgb_map.astype(F32), disp_map=disp, acc_map=np.minimum(acc, F32(1.)),
                weights=w, alpha=alpha)


# =============================================================================
# a21 / a22  A-NeRF cutoff positional encoding
# =============================================================================
def cutoff_pe_dist(v, cutoff, tau, L=7):
    """CutoffEmbedder._embed for joint distances (core/cutoff_embedder.py:151-214) with
    cut_to_dist, cutoff_shift, cutoff_inputs: v [M,24] -> [M,(1+2L)*24] block-major."""
    inp = (cutoff - v).astype(F32)
    sh = (inp * (F32(2.) / cutoff) - F32(1.)).astype(F32)
    w = (F32(1.) - sigmoid((F32(tau) * (v - cutoff)).astype(F32))).astype(F32)
    blocks = [inp]
    for l in range(L):
        f = (sh * F32(2.0 ** l)).astype(F32)
        blocks += [np.sin(f).astype(F32), np.cos(f).astype(F32)]
    e = np.stack(blocks, -2)  # [M,1+2L,24]
    return (e * w[..., None, :]).astype(F32).reshape(v.shape[0], -1), w


def cutoff_pe_view(d, v, cutoff, tau, L=4):
    """view variant (dist_inputs=True, cutoff_inputs=True; core/cutoff_embedder.py:156-166,176-197):
    d [M,72] per-bone unit directions, cutoff weights from the joint distances v [M,24] repeated x3;
    -> [M,(1+2L)*72] block-major [d, sin(2^0 d), cos(2^0 d), ...], EVERY block times the weight."""
    vr = np.repeat(v, 3, -1)
    cr = np.repeat(cutoff, 3, -1)
    w = (F32(1.) - sigmoid((F32(tau) * (vr - cr)).astype(F32))).astype(F32)
    blocks = [d.astype(F32)]
    for l in range(L):
        f = (d * F32(2.0 ** l)).astype(F32)
        blocks += [np.sin(f).astype(F32), np.cos(f).astype(F32)]
    e = np.stack(blocks, -2)  # [M,1+2L,72]
    return (e * w[..., None, :]).astype(F32).reshape(d.shape[0], -1), w


def unit_vectors(x):
    """F.normalize(x, dim=-1, p=2): x / max(|x|, 1e-12)  (VecNormEncoder, core/encoders.py:774-795)."""
    n = torch_norm(x, keepdims=True)
    return (x / np.maximum(n, F32(1e-12))).astype(F32)


def bone_local_rays(rays_d, skts):
    """transform_batch_rays (core/encoders.py:305-317): rotation part of every bone transform -> [R,24,3] (k = 0, 1, 2 summed in
    order, each op rounded: torch's small batched matmul, see near_far_boxes)."""
    M = np.concatenate([skts[:, :, :3, :3], np.zeros_like(skts[:, :, :3, :1])], -1)
    return _affine_unfused(M, rays_d[:, None, :])


class AnerfOracle:
    """Eval-mode forward of A-NeRF (nerf_type=nerf; core/networks/nerf.py:107-122,222-279): joint-distance
    cutoff PE + unit bone-local directions -> density trunk; cutoff-weighted PE of the bone-local ray
    directions + frame code -> colour head.  No skeleton GNN, no culling."""

    def __init__(self, cfg, sd, rest_pose):
        self.cfg = cfg
        self.sd = {k: np.asarray(v, dtype=F32) if np.asarray(v).dtype != np.int64 else v for k, v in sd.items()}
        self.align = bone_align_transforms(rest_pose)

    def forward(self, pts, rays_d, skts, bones=None, cam_idxs=None, n_uniques=1, stages=False):
        cfg, sd = self.cfg, self.sd
        R, S = pts.shape[:2]
        M = R * S
        pts_t = bone_local(pts, skts, self.align)
        v = torch_norm(pts_t)                                                              # RelDist
        r = unit_vectors(pts_t).reshape(R, S, -1)                                          # VecNorm, [R,S,72]
        tau = float(sd['pe_fn.tau'])
        v_pe, w = cutoff_pe_dist(v.reshape(M, J), sd['pe_fn.cutoff_dist'], tau, cfg['multires'])
        dens_in = np.concatenate([v_pe, r.reshape(M, -1)], -1).astype(F32)
        d = unit_vectors(bone_local_rays(rays_d, skts)).reshape(R, -1)                     # [R,72]
        d_pe, _ = cutoff_pe_view(np.repeat(d, S, axis=0), v.reshape(M, J), sd['dirs_pe_fn.cutoff_dist'],
                                 float(sd['dirs_pe_fn.tau']), cfg['multires_views'])
        vin = d_pe
        if cfg['use_framecode']:
            vin = np.concatenate([vin, np.repeat(frame_codes(sd, cam_idxs, R), S, axis=0)], -1)
        raw = mlp(sd, dens_in, vin, cfg['D'], cfg['skips']).reshape(R, S, 4)
        enc = dict(v=v, r=r)
        if stages:
            enc.update(density_inputs=dens_in, view_inputs=vin, view_dirs=d, w=w)
        return raw, enc

    def render(self, ray_batch, skts, bones, cyls, cam_idxs=None, n_uniques=1, N_samples=None, N_importance=None,
               chunk=None, stages=False, near_far=None):
        cfg = self.cfg
        S = N_samples or cfg['N_samples']
        Sf = N_importance or cfg['N_importance']
        rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
        if near_far is None:
            near, far = near_far_cylinder(rays_o, rays_d, cyls, ray_batch[:, 6:7], ray_batch[:, 7:8], chunk)
        else:
            near, far = near_far
        z = coarse_z(near, far, S)
        raw, enc = self.forward(sample_points(rays_o, rays_d, z), rays_d, skts, bones, cam_idxs, n_uniques, stages)
        B = cfg['density_scale']
        out0 = composite(raw, z, rays_d, B)
        z_all, z_fine, order = importance_z(z, out0['weights'], Sf)
        raw_f, _ = self.forward(sample_points(rays_o, rays_d, z_fine), rays_d, skts, bones, cam_idxs, n_uniques)
        raw_all = np.take_along_axis(np.concatenate([raw, raw_f], 1), order[..., None], 1)
        out = composite(raw_all, z_all, rays_d, B)
        ret = dict(rgb_map=out['rgb_map'], disp_map=out['disp_map'], acc_map=out['acc_map'], alpha=out['alpha'],
                   T_i=out['weights'], rgb0=out0['rgb_map'], disp0=out0['disp_map'], acc0=out0['acc_map'],
                   alpha0=out0['alpha'])
        if stages:
            ret.update(near=near, far=far, z_coarse=z, raw_coarse=raw, weights_coarse=out0['weights'], z_fine=z_fine,
                       z_sorted=z_all, sorted_idxs=order, raw_fine=raw_f, raw_sorted=raw_all, enc=enc)
        return ret


# =============================================================================
# a15 / a20  orchestration
# =============================================================================
class DanboOracle:
    """Eval-mode forward of the DANBO configs (FGNNcat + vox_MIXGNN + sigmoid)."""

    def __init__(self, cfg, sd, rest_pose):
        self.cfg = cfg
        self.sd = {k: np.asarray(v, dtype=F32) if np.asarray(v).dtype != np.int64 else v for k, v in sd.items()}
        self.align = bone_align_transforms(rest_pose)
        self.rest_pose = rest_pose

    # ---- network forward on given points: DANBO.forward -> raw, encoded ------------
    def forward(self, pts, rays_d, skts, bones, cam_idxs=None, n_uniques=1, stages=False):
        cfg, sd = self.cfg, self.sd
        R, S = pts.shape[:2]
        skip = R // n_uniques
        pose_of_ray = np.arange(R) // skip
        vols = pose_volumes(sd, bones[::skip], cfg['multires_graph'])
        pts_t = bone_local(pts, skts, self.align)
        x, valid = in_volume(pts_t, sd['graph_net.axis_scale'])
        feat = factorised_gather(vols, x, pose_of_ray, cfg['voxel_feat'], cfg['voxel_res'])
        feat = (feat * window(x)[..., None]).astype(F32)  # attenuate_feat
        M = R * S
        pf = feat.reshape(M, J, -1)
        logits = assignment_logits(sd, pf)
        p, h = blend(pf, logits, valid.reshape(M, J))
        dens_in = positional_encoding(h, cfg['multires_voxel'])
        d = view_dirs(cfg, rays_d, skts)
        vin = positional_encoding(d, cfg['multires_views'])
        if cfg['use_framecode']:
            vin = np.concatenate([vin, frame_codes(sd, cam_idxs, R)], -1)
        vin = np.repeat(vin, S, axis=0)
        raw = mlp(sd, dens_in, vin, cfg['D'], cfg['skips']).reshape(R, S, 4)
        enc = dict(confd=logits.reshape(R, S, J), part_invalid=(~valid).astype(F32), valid=valid)
        if stages:
            enc.update(pts_t=pts_t, part_feat=feat, agg_p=p, h=h, density_inputs=dens_in,
                       view_inputs=vin, volumes=vols)
        return raw, enc

    # ---- RayCaster.render_rays, eval, single_net two-pass (raycasters.py:245-377) ----
    def near_far(self, rays_o, rays_d, cyls, skts, near, far, chunk=None):
        n, f = near_far_cylinder(rays_o, rays_d, cyls, near, far, chunk)
        if self.cfg['use_volume_near_far']:
            n, f = near_far_boxes(rays_o, rays_d, skts, self.align, self.sd['graph_net.axis_scale'], n, f)
        return n, f

    def render(self, ray_batch, skts, bones, cyls, cam_idxs=None, n_uniques=1,
               N_samples=None, N_importance=None, chunk=None, stages=False, near_far=None, draws=None):
        """draws: dict(t_rand [R,S], u_rand [R,Sf], noise_c [R,S], noise_f [R,S+Sf]) -- the random numbers of the training-mode
        branches (perturb > 0: ray_utils.py:240 and :171; raw_noise_std > 0: nerf.py:316, noise already x std x B); None: eval"""
        cfg = self.cfg
        draws = draws or {}
        S = N_samples or cfg['N_samples']
        Sf = N_importance or cfg['N_importance']
        rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
        near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
        if near_far is None:
            near, far = self.near_far(rays_o, rays_d, cyls, skts, near, far, chunk)
        else:  # decoupled parity tests feed the reference's own bounds
            near, far = near_far
        z = coarse_z(near, far, S, draws.get('t_rand'))
        pts = sample_points(rays_o, rays_d, z)
        raw, enc = self.forward(pts, rays_d, skts, bones, cam_idxs, n_uniques, stages)
        B = cfg['density_scale']
        out0 = composite(raw, z, rays_d, B, draws.get('noise_c'))
        z_all, z_fine, order = importance_z(z, out0['weights'], Sf, u=draws.get('u_rand'))
        pts_f = sample_points(rays_o, rays_d, z_fine)
        raw_f, enc_f = self.forward(pts_f, rays_d, skts, bones, cam_idxs, n_uniques, False)
        raw_all = np.take_along_axis(np.concatenate([raw, raw_f], 1), order[..., None], 1)
        out = composite(raw_all, z_all, rays_d, B, draws.get('noise_f'))
        ret = dict(rgb_map=out['rgb_map'], disp_map=out['disp_map'], acc_map=out['acc_map'],
                   alpha=out['alpha'], T_i=out['weights'], rgb0=out0['rgb_map'],
                   disp0=out0['disp_map'], acc0=out0['acc_map'], alpha0=out0['alpha'])
        if stages:
            ret.update(near=near, far=far, z_coarse=z, raw_coarse=raw, weights_coarse=out0['weights'],
                       z_fine=z_fine, z_sorted=z_all, sorted_idxs=order, raw_fine=raw_f,
                       raw_sorted=raw_all, enc=enc)
        return ret


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float('inf') if mse == 0 else -10.0 * math.log10(mse)
