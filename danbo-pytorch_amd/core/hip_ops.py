"""Torch-tensor wrappers around the C ABI (include/danbo_hip.h).

PyTorch is used here for device memory and streams only.  Every function enqueues on
`torch.cuda.current_stream()` and returns without synchronising.  Non-CUDA tensors are
rejected: there is no CPU fallback.
"""
import ctypes

import torch

from . import _hip

J = 24
VOL = 240
FEAT = 15
RAY_FLAT_VMAX = 1e4          # DANBO_RAY_FLAT_VMAX (include/danbo_hip.h)
H_STRIDE = 16
MLP_PACKED_FLOATS = 659456
VIEW_W = 128


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/HIP tensor -- libdanbo_hip has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _call(name, *args):
    _hip.check(getattr(_hip.lib(), name)(*args), name)


# --------------------------------------------------------------------------------------
def pose_volumes(bones, gw, L_graph=5):
    """bones [G,24,3]; gw: dict(w0,adjw0,b0,w1,adjw1,b1,w2,b2,w3,b3) -> volumes [G,24,240]."""
    bones = _f32(bones, "bones")
    G = bones.shape[0]
    W = gw["w1"].shape[-1]
    scratch = torch.empty(3 * G * J * W, device=bones.device, dtype=torch.float32)
    vol = torch.empty(G, J, VOL, device=bones.device, dtype=torch.float32)
    _call("danbo_pose_volumes_fwd", _p(bones), G, L_graph, W,
          _p(gw["w0"]), _p(gw["adjw0"]), _p(gw["b0"]), _p(gw["w1"]), _p(gw["adjw1"]), _p(gw["b1"]),
          _p(gw["w2"]), _p(gw["b2"]), _p(gw["w3"]), _p(gw["b3"]), _p(scratch), _p(vol), _stream())
    return vol


def near_far_cylinder(rays_o, rays_d, cyl, near0=0.0, far0=1.0, chunk=4096, near_in=None, far_in=None):
    """near0/far0: scalar placeholder bounds; near_in/far_in: optional per-ray placeholders [R]."""
    rays_o, rays_d, cyl = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(cyl, "cyl")
    near_in, far_in = _f32(near_in, "near_in"), _f32(far_in, "far_in")
    R, G = rays_o.shape[0], cyl.shape[0]
    nchunk = (R + chunk - 1) // chunk
    scratch = torch.empty(nchunk * 8, device=rays_o.device, dtype=torch.float32)
    near = torch.empty(R, device=rays_o.device, dtype=torch.float32)
    far = torch.empty_like(near)
    _call("danbo_near_far_cylinder", _p(rays_o), _p(rays_d), _p(cyl), R, G, float(near0), float(far0),
          _p(near_in), _p(far_in), int(chunk), _p(scratch), _p(near), _p(far), _stream())
    return near, far


def near_far_boxes(rays_o, rays_d, skts, align, axis_scale, near, far):
    """updates near/far in place"""
    R, G = rays_o.shape[0], skts.shape[0]
    _call("danbo_near_far_boxes", _p(_f32(rays_o, "rays_o")), _p(_f32(rays_d, "rays_d")), _p(_f32(skts, "skts")),
          _p(_f32(align, "align")), _p(_f32(axis_scale, "axis_scale")), R, G, _p(near), _p(far), _stream())
    return near, far


def coarse_samples(near, far, S, t_rand=None):
    R = near.shape[0]
    z = torch.empty(R, S, device=near.device, dtype=torch.float32)
    _call("danbo_coarse_samples", _p(near), _p(far), R, S, _p(_f32(t_rand, "t_rand")), _p(z), _stream())
    return z


class Geometry:
    """Per-call bundle of the geometric inputs shared by K1a / K1b / K2."""

    def __init__(self, rays_o, rays_d, skts, align, axis_scale, z=None, pts=None, ray_mask=None):
        self.rays_o = _f32(rays_o, "rays_o")
        self.rays_d = _f32(rays_d, "rays_d")
        self.skts = _f32(skts, "skts")          # [G,24,4,4]
        self.align = _f32(align, "align")        # [24,4,4]
        self.axis_scale = _f32(axis_scale, "axis_scale")
        self.z = _f32(z, "z")
        self.pts = _f32(pts, "pts")
        assert (self.z is None) != (self.pts is None)
        self.R = self.rays_o.shape[0]
        self.G = self.skts.shape[0]
        self.S = self.z.shape[1] if self.z is not None else self.pts.shape[1]
        self.M = self.R * self.S
        self.device = self.rays_o.device
        # (mask [R] int32, t_lo [R], t_hi [R][, flat [R] int32]) of ray_bone_mask() for these rays, or None: bone_cull's accelerator
        self.ray_mask = ray_mask if self.z is not None else None

    def head(self):
        return (_p(self.rays_o), _p(self.rays_d), _p(self.z), _p(self.pts), self.R, self.S, self.G,
                _p(self.skts), _p(self.align), _p(self.axis_scale))


def bone_cull(geo, compact=True, cnt=None):
    """-> valid_bits [M] (int32 view of uint32), list [M] int32 or None, count [1] int32 or None.
    cnt: a ZEROED [1] int32 the caller provides (render(): one fill for both passes) instead of a fresh one."""
    bits = torch.empty(geo.M, device=geo.device, dtype=torch.int32)
    lst = None
    if compact:
        lst = torch.empty(geo.M, device=geo.device, dtype=torch.int32)
        if cnt is None:
            cnt = torch.zeros(1, device=geo.device, dtype=torch.int32)
        assert cnt.dtype == torch.int32 and cnt.numel() == 1 and cnt.is_contiguous()
    else:
        cnt = None
    rm = tuple(geo.ray_mask) if geo.ray_mask is not None else (None, None, None)
    flat = rm[3] if len(rm) > 3 else None
    _call("danbo_bone_cull", *geo.head(), _p(rm[0]), _p(rm[1]), _p(rm[2]), _p(flat), _p(bits), _p(lst), _p(cnt), _stream())
    return bits, lst, cnt


def ray_bone_mask(rays_o, rays_d, skts, align, axis_scale, t_lo, t_hi, want_flat=False):
    """-> (mask [R] int32, t_lo, t_hi[, flat [R] int32]): bit j of mask[r] clear = no point of ray r between t_lo[r] and t_hi[r]
    can lie inside bone j's volume (conservative slab test, csrc/k_sample.hip:k_ray_bone_mask).  Pass it to
    Geometry(ray_mask=...): bone_cull then skips the rays -- and whole workgroups -- that miss every volume; the in-volume mask
    itself does not depend on it.  want_flat: also the flags of the candidate rays of constants, which bone_cull (given the
    4-tuple) confirms and composite_importance(ray_flat=...) / composite_merged(ray_flat=...) act on (danbo_hip.h)."""
    rays_o, rays_d, skts = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(skts, "skts")
    t_lo, t_hi = _f32(t_lo, "t_lo").reshape(-1), _f32(t_hi, "t_hi").reshape(-1)
    R, G = rays_o.shape[0], skts.shape[0]
    assert t_lo.shape[0] == R and t_hi.shape[0] == R
    mask = torch.empty(R, device=rays_o.device, dtype=torch.int32)
    flat = torch.empty(R, device=rays_o.device, dtype=torch.int32) if want_flat else None
    _call("danbo_ray_bone_mask", _p(rays_o), _p(rays_d), _p(t_lo), _p(t_hi), R, G, _p(skts), _p(_f32(align, "align")),
          _p(_f32(axis_scale, "axis_scale")), _p(mask), _p(flat), _stream())
    return (mask, t_lo, t_hi, flat) if want_flat else (mask, t_lo, t_hi)


def group_rows(bits, lst, cnt):
    """in place: rows of the compacted list whose samples lie inside the same set of bone volumes become neighbours
    (csrc/k_group.hip) -- k_assign16 then evaluates ~1.7 instead of ~2.9 bones per wavefront"""
    _call("danbo_group_rows", _p(bits), _p(lst), _p(cnt), lst.shape[0], _stream())
    return lst


def bone_gather(geo, volumes, lst=None, cnt=None, n=None):
    n = geo.M if n is None else n
    out = torch.empty(n, J, FEAT, device=geo.device, dtype=torch.float32)
    _call("danbo_bone_gather_fwd", *geo.head(), _p(volumes), _p(lst), _p(cnt), n, _p(out), _stream())
    return out


def assign_blend(part_feat, bits, aw, lst=None, cnt=None, n=None, want_confd=False):
    n = part_feat.shape[0] if n is None else n
    h = torch.empty(n, H_STRIDE, device=part_feat.device, dtype=torch.float32)
    confd = torch.empty(n, J, device=part_feat.device, dtype=torch.float32) if want_confd else None
    _call("danbo_assign_blend_fwd", _p(part_feat), _p(bits), _p(lst), _p(cnt), n,
          _p(aw["w0"]), _p(aw["adjw"]), _p(aw["b0"]), _p(aw["w1"]), _p(aw["b1"]), _p(aw["w2"]), _p(aw["b2"]),
          _p(h), _p(confd), _stream())
    return h, confd


def gather_assign_blend(geo, volumes, bits, aw, lst=None, cnt=None, n=None, want_confd=False):
    n = geo.M if n is None else n
    h = torch.empty(n, H_STRIDE, device=geo.device, dtype=torch.float32)
    confd = torch.empty(n, J, device=geo.device, dtype=torch.float32) if want_confd else None
    _call("danbo_gather_assign_blend_fwd", *geo.head(), _p(volumes), _p(bits), _p(lst), _p(cnt), n,
          _p(aw["w0"]), _p(aw["adjw"]), _p(aw["b0"]), _p(aw["w1"]), _p(aw["b1"]), _p(aw["w2"]), _p(aw["b2"]),
          _p(h), _p(confd), _stream())
    return h, confd


ASSIGN16_PACKED_BYTES = 262144


_SMPL_PARENTS = [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
_ADJ_CHECKED = set()


def check_smpl_adjacency(adj):
    """k_assign16 (the fp16-split K2 kernel) walks the SMPL tree's neighbour table, compiled in (csrc/k_assign16.hip a16_nb): a
    checkpoint whose `adj` buffer describes another graph must not be evaluated with it silently.  Checked once per buffer."""
    key = (adj.data_ptr(), adj._version)
    if key in _ADJ_CHECKED:
        return
    want = torch.eye(J)
    for i, p_ in enumerate(_SMPL_PARENTS):
        if i != p_:
            want[i, p_] = want[p_, i] = 1.0
    if not torch.equal((adj.detach().reshape(J, J) != 0).cpu(), want != 0):
        raise NotImplementedError("the assignment net's adjacency is not the SMPL tree: the fast K2 kernel (k_assign16) has that "
                                  "tree compiled in -- use mlp_mode='fp32' (k_assign_blend reads the adjacency it is given)")
    _ADJ_CHECKED.add(key)


def assign16_pack(aw):
    """fp16 hi/lo fragments of the assignment GNN with the adjacency folded into layer 0."""
    packed = torch.empty(ASSIGN16_PACKED_BYTES, device=aw["w0"].device, dtype=torch.uint8)
    _call("danbo_assign16_pack", _p(aw["w0"]), _p(aw["adjw"]), _p(aw["w1"]), _p(packed), _stream())
    return packed


_TICKETS = {}


def _ticket(device):
    """the tile-ticket word of danbo_gather_assign_blend16_fwd: zeroed once, it returns to 0 at the end of every launch; one per
    (device, stream), since launches on different streams may overlap"""
    if torch.cuda.is_current_stream_capturing():
        # memory allocated during a capture belongs to that graph's private pool and dies with the graph: never cache it
        return torch.zeros(1, device=device, dtype=torch.int32)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(1, device=device, dtype=torch.int32)
    return t


def gather_assign_blend16(geo, volumes, bits, aw, packed16, lst=None, cnt=None, n=None, want_confd=False):
    n = geo.M if n is None else n
    h = torch.empty(n, H_STRIDE, device=geo.device, dtype=torch.float32)
    confd = torch.empty(n, J, device=geo.device, dtype=torch.float32) if want_confd else None
    _call("danbo_gather_assign_blend16_fwd", *geo.head(), _p(volumes), _p(bits), _p(lst), _p(cnt), n, _p(packed16),
          _p(aw["b0"]), _p(aw["b1"]), _p(aw["w2"]), _p(aw["b2"]), _p(h), _p(confd), _p(_ticket(geo.device)), _stream())
    return h, confd


def mlp_pack(pts_w, feature_w, views_w):
    """pts_w: 8 tensors; views_w [128, 256+Cv] -> (packed [659456], views_w_ray_t [Cv,128])."""
    dev = feature_w.device
    Cv = views_w.shape[1] - 256
    packed = torch.empty(MLP_PACKED_FLOATS, device=dev, dtype=torch.float32)
    wrt = torch.empty(max(Cv, 1), VIEW_W, device=dev, dtype=torch.float32)
    pts_w = [_f32(w, "pts_w") for w in pts_w]
    _call("danbo_mlp_pack", _ptr_array(pts_w), _p(_f32(feature_w, "feature_w")), _p(_f32(views_w, "views_w")), Cv,
          _p(packed), _p(wrt), _stream())
    return packed, wrt


def view_consts(rays_d, skts, ray_mode, normalise, L_view, framecodes, mean_code, cam_idx, wrt, views_b,
                rgb_w, rgb_b, empty_consts=None, rgb_order=0, code_table=None, ray_list=None, ray_count=None):
    """-> cview [R,128], raw_empty [R,4]; ray_list / ray_count (flat_rays()): only the listed rays' rows are computed"""
    rays_d = _f32(rays_d, "rays_d")
    R, G = rays_d.shape[0], skts.shape[0]
    Cf = 0 if mean_code is None else mean_code.shape[0]
    n_codes = 0 if framecodes is None else framecodes.shape[0]
    cview = torch.empty(R, VIEW_W, device=rays_d.device, dtype=torch.float32)
    raw_empty = torch.empty(R, 4, device=rays_d.device, dtype=torch.float32) if empty_consts is not None else None
    if cam_idx is not None:
        cam_idx = cam_idx.reshape(-1).to(torch.int64).contiguous()
    _call("danbo_view_consts", _p(rays_d), _p(_f32(skts, "skts")), R, G, int(ray_mode), int(normalise), int(L_view),
          _p(framecodes), n_codes, Cf, _p(mean_code), _p(cam_idx), _p(wrt), _p(views_b), _p(rgb_w), _p(rgb_b),
          _p(empty_consts), int(rgb_order), _p(code_table), _p(ray_list), _p(ray_count), _p(cview), _p(raw_empty), _stream())
    return cview, raw_empty


def view_code_table(framecodes, mean_code, L_view, wrt, views_b):
    n_codes, Cf = framecodes.shape
    table = torch.empty(n_codes + 1, VIEW_W, device=framecodes.device, dtype=torch.float32)
    _call("danbo_view_code_table", _p(framecodes), _p(mean_code), n_codes, Cf, int(L_view), _p(wrt), _p(views_b),
          _p(table), _stream())
    return table


def pe_mlp(h, S, packed, pts_b, alpha_w, alpha_b, feature_b, cview, rgb_w, rgb_b, raw_out,
           lst=None, cnt=None, n=None, aux=False):
    n = h.shape[0] if n is None else n
    aux_out = torch.empty(n, VIEW_W + 1, device=h.device, dtype=torch.float32) if aux else None
    _call("danbo_pe_mlp_fwd", _p(h), _p(lst), _p(cnt), n, S, _p(packed), _ptr_array(pts_b), _p(alpha_w), _p(alpha_b),
          _p(feature_b), _p(cview), _p(rgb_w), _p(rgb_b), _p(raw_out), _p(aux_out), _stream())
    return aux_out


MLP16_PACKED_BYTES = 2424832 + 128 + 128 * 256 * 4      # 74 chunks + trailer (winv, wmax, W_fv): include/danbo_hip.h


def mlp16_pack(pts_w, feature_w, feature_b, views_w, views_b, form=16):
    """fp16 hi/lo fragment packing for pe_mlp16 (form 16) / pe_mlp32 (form 32) -> (uint8 buffer [MLP16_PACKED_BYTES], views_b_eff [128])."""
    dev = feature_w.device
    Cv = views_w.shape[1] - 256
    packed = torch.empty(MLP16_PACKED_BYTES, device=dev, dtype=torch.uint8)
    vb = torch.empty(VIEW_W, device=dev, dtype=torch.float32)
    pts_w = [_f32(w, "pts_w") for w in pts_w]
    _call({16: "danbo_mlp16_pack", 32: "danbo_mlp32_pack"}[form], _ptr_array(pts_w), _p(_f32(feature_w, "feature_w")), _p(_f32(feature_b, "feature_b")),
          _p(_f32(views_w, "views_w")), _p(_f32(views_b, "views_b")), Cv, _p(packed), _p(vb), _stream())
    return packed, vb


def pe_mlp16(h, S, packed16, pts_b, alpha_w, alpha_b, cview, rgb_w, rgb_b, raw_out,
             lst=None, cnt=None, n=None, aux=False, form=16):
    n = h.shape[0] if n is None else n
    aux_out = torch.empty(n, VIEW_W + 1, device=h.device, dtype=torch.float32) if aux else None
    _call({16: "danbo_pe_mlp16_fwd", 32: "danbo_pe_mlp32_fwd"}[form], _p(h), _p(lst), _p(cnt), n, S, _p(packed16), _ptr_array(pts_b), _p(alpha_w),
          _p(alpha_b), _p(cview), _p(rgb_w), _p(rgb_b), _p(raw_out), _p(aux_out), _stream())
    return aux_out


def fill_raw(raw_empty, S):
    R = raw_empty.shape[0]
    raw = torch.empty(R, S, 4, device=raw_empty.device, dtype=torch.float32)
    _call("danbo_fill_raw", _p(raw_empty), R, S, _p(raw), _stream())
    return raw


def composite(raw, z, rays_d, B=1.0, noise=None, bits=None, raw_empty=None, flat=None):
    """bits / raw_empty: un-filled raw (samples with in-volume word 0 take raw_empty[ray]: danbo_hip.h);
    flat: flat_rays(want_weights=True)'s result -- only its listed rays are composited, into its buffers"""
    raw, z, rays_d = _f32(raw, "raw"), _f32(z, "z"), _f32(rays_d, "rays_d")
    R, S = z.shape
    dev = raw.device
    if flat is not None:
        o0 = flat["out0"]
        rgb, disp, acc, w, al = o0["rgb_map"], o0["disp_map"], o0["acc_map"], o0["weights"], o0["alpha"]
        assert w is not None and w.shape == (R, S) and al.shape == (R, S)
    else:
        rgb = torch.empty(R, 3, device=dev, dtype=torch.float32)
        disp = torch.empty(R, device=dev, dtype=torch.float32)
        acc = torch.empty(R, device=dev, dtype=torch.float32)
        w = torch.empty(R, S, device=dev, dtype=torch.float32)
        al = torch.empty(R, S, device=dev, dtype=torch.float32)
    if flat is not None or bits is not None:
        _call("danbo_composite_rays_fwd", _p(raw), _p(_f32(raw_empty, "raw_empty")), _p(bits), _p(z), _p(rays_d), R, S, float(B),
              _p(_f32(noise, "noise")), _p(rgb), _p(disp), _p(acc), _p(w), _p(al), _p(flat["ray_list"] if flat else None),
              _p(flat["ray_count"] if flat else None), _stream())
    else:
        _call("danbo_composite_fwd", _p(raw), _p(z), _p(rays_d), R, S, float(B), _p(_f32(noise, "noise")),
              _p(rgb), _p(disp), _p(acc), _p(w), _p(al), _stream())
    return dict(rgb_map=rgb, disp_map=disp, acc_map=acc, weights=w, alpha=al)


def importance_samples(z, weights, Sf, u=None, flat=None):
    """flat: flat_rays()'s result (64 < S <= 256, Sf <= 64) -- only its listed rays are resampled, into its z_fine (the rows of
    the other rays are there; their z_sorted / sorted_idx rows are never made)"""
    R, S = z.shape
    dev = z.device
    zs = torch.empty(R, S + Sf, device=dev, dtype=torch.float32)
    idx = torch.empty(R, S + Sf, device=dev, dtype=torch.int32)
    if flat is not None:
        zf = flat["z_fine"]
        assert zf.shape == (R, Sf)
        _call("danbo_importance_samples_rays", _p(_f32(z, "z")), _p(_f32(weights, "weights")), R, S, Sf, _p(_f32(u, "u")),
              _p(zf), _p(zs), _p(idx), _p(flat["ray_list"]), _p(flat["ray_count"]), _stream())
        return zs, zf, idx
    zf = torch.empty(R, Sf, device=dev, dtype=torch.float32)
    _call("danbo_importance_samples", _p(_f32(z, "z")), _p(_f32(weights, "weights")), R, S, Sf, _p(_f32(u, "u")),
          _p(zf), _p(zs), _p(idx), _stream())
    return zs, zf, idx


def flat_rays(t_lo, ray_flat, S, Sf, want_weights=False, rows_later=False, cnt=None):
    """the rays of constants (danbo_flat_rays, include/danbo_hip.h -- the caller vouches for the model's empty-space density and
    colour): allocates the outputs of BOTH fused composites, writes them for every flagged ray, and lists the other rays ->
    dict(out0=..., out=..., z_fine=..., ray_list=..., ray_count=...) for view_consts(ray_list=...),
    composite_importance(flat=...) and composite_merged(flat=...).
    rows_later: only the list and the per-ray outputs now; the caller runs result["rows"]() (same stream or a later one) before
    anything reads the per-sample rows -- render() puts the view constants of the listed rays in between."""
    t_lo = _f32(t_lo, "t_lo").reshape(-1)
    R, dev = t_lo.shape[0], t_lo.device
    assert ray_flat.dtype == torch.int32 and ray_flat.shape[0] == R
    f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)  # noqa: E731
    out0 = dict(rgb_map=f(R, 3), disp_map=f(R), acc_map=f(R), weights=f(R, S) if want_weights else None, alpha=f(R, S))
    out = dict(rgb_map=f(R, 3), disp_map=f(R), acc_map=f(R), weights=f(R, S + Sf), alpha=f(R, S + Sf))
    zf = f(R, Sf)
    lst = torch.empty(R, device=dev, dtype=torch.int32)
    if cnt is None:          # (render() hands over a zeroed word of the buffer it fills once per frame)
        cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    assert cnt.dtype == torch.int32 and cnt.numel() == 1
    def launch(parts):
        _call("danbo_flat_rays", _p(t_lo), _p(ray_flat), R, int(S), int(Sf), _p(out0["rgb_map"]), _p(out0["disp_map"]),
              _p(out0["acc_map"]), _p(out0["weights"]), _p(out0["alpha"]), _p(zf), _p(out["rgb_map"]), _p(out["disp_map"]),
              _p(out["acc_map"]), _p(out["weights"]), _p(out["alpha"]), _p(lst), _p(cnt), parts, _stream())
    launch(1 if rows_later else 3)
    res = dict(out0=out0, out=out, z_fine=zf, ray_list=lst, ray_count=cnt)
    if rows_later:
        res["rows"] = lambda: launch(2)
    return res


def composite_importance(raw, z, rays_d, Sf, B=1.0, noise=None, u=None, bits=None, raw_empty=None, want_weights=True,
                         flat=None):
    """coarse composite + importance resampling in one launch (S, Sf <= 64) ->
    (out0 dict, z_sorted, z_fine, sorted_idx); bits/raw_empty: un-filled raw (see danbo_hip.h).
    flat: flat_rays()'s result -- only its listed rays are composited, into its buffers (the rows of the other rays are already
    there; their z_sorted / sorted_idx rows are never made)."""
    raw, z, rays_d = _f32(raw, "raw"), _f32(z, "z"), _f32(rays_d, "rays_d")
    R, S = z.shape
    dev = raw.device
    if flat is not None:
        o0, zf = flat["out0"], flat["z_fine"]
        rgb, disp, acc, w, al = o0["rgb_map"], o0["disp_map"], o0["acc_map"], o0["weights"], o0["alpha"]
        assert al.shape == (R, S) and zf.shape == (R, Sf)
    else:
        rgb = torch.empty(R, 3, device=dev, dtype=torch.float32)
        disp = torch.empty(R, device=dev, dtype=torch.float32)
        acc = torch.empty(R, device=dev, dtype=torch.float32)
        w = torch.empty(R, S, device=dev, dtype=torch.float32) if want_weights else None
        al = torch.empty(R, S, device=dev, dtype=torch.float32)
        zf = torch.empty(R, Sf, device=dev, dtype=torch.float32)
    zs = torch.empty(R, S + Sf, device=dev, dtype=torch.float32)
    idx = torch.empty(R, S + Sf, device=dev, dtype=torch.int32)
    _call("danbo_composite_importance_fwd", _p(raw), _p(_f32(raw_empty, "raw_empty")), _p(bits), _p(z), _p(rays_d), R, S,
          int(Sf), float(B), _p(_f32(noise, "noise")), _p(_f32(u, "u")), _p(rgb), _p(disp), _p(acc), _p(w), _p(al), _p(zf),
          _p(zs), _p(idx), _p(flat["ray_list"] if flat else None), _p(flat["ray_count"] if flat else None), _stream())
    return dict(rgb_map=rgb, disp_map=disp, acc_map=acc, weights=w, alpha=al), zs, zf, idx


def composite_merged(raw_a, raw_b, idx, z_sorted, rays_d, B=1.0, noise=None, bits_a=None, bits_b=None, raw_empty=None,
                     want_raw=False, flat=None):
    """final composite reading the coarse / importance raw through the sorted order (no merged copy);
    flat: flat_rays()'s result, as composite_importance"""
    raw_a, raw_b, rays_d = _f32(raw_a, "raw_a"), _f32(raw_b, "raw_b"), _f32(rays_d, "rays_d")
    R, S = raw_a.shape[:2]
    Sf = raw_b.shape[1]
    dev = raw_a.device
    if flat is not None:
        assert not want_raw
        o = flat["out"]
        rgb, disp, acc, w, al = o["rgb_map"], o["disp_map"], o["acc_map"], o["weights"], o["alpha"]
        assert al.shape == (R, S + Sf)
    else:
        rgb = torch.empty(R, 3, device=dev, dtype=torch.float32)
        disp = torch.empty(R, device=dev, dtype=torch.float32)
        acc = torch.empty(R, device=dev, dtype=torch.float32)
        w = torch.empty(R, S + Sf, device=dev, dtype=torch.float32)
        al = torch.empty(R, S + Sf, device=dev, dtype=torch.float32)
    rs = torch.empty(R, S + Sf, 4, device=dev, dtype=torch.float32) if want_raw else None
    _call("danbo_composite_merged_fwd", _p(raw_a), _p(raw_b), _p(_f32(raw_empty, "raw_empty")), _p(bits_a), _p(bits_b),
          _p(idx), _p(_f32(z_sorted, "z_sorted")), _p(rays_d), R, S, Sf, float(B), _p(_f32(noise, "noise")), _p(rgb),
          _p(disp), _p(acc), _p(w), _p(al), _p(rs), _p(flat["ray_list"] if flat else None),
          _p(flat["ray_count"] if flat else None), _stream())
    out = dict(rgb_map=rgb, disp_map=disp, acc_map=acc, weights=w, alpha=al)
    if want_raw:
        out["raw_sorted"] = rs
    return out


def random_draws(state, n_uniform, n_normal, normal_std=1.0):
    """danbo_random_draws (include/danbo_hip.h): -> (uniform [n_uniform] in [0, 1), normal [n_normal] ~ N(0, std^2)), either None
    for a count of 0; state: int64 [3] device tensor (seed, counter, 0) that the kernel advances"""
    assert state.dtype == torch.int64 and state.numel() == 3 and state.is_cuda
    dev = state.device
    u = torch.empty(n_uniform, device=dev, dtype=torch.float32) if n_uniform else None
    nz = torch.empty(n_normal, device=dev, dtype=torch.float32) if n_normal else None
    _call("danbo_random_draws", _p(state), int(n_uniform), _p(u), int(n_normal), float(normal_std), _p(nz), _stream())
    return u, nz


def row_span(t):
    """(tensor to keep alive, rows, words per row, row stride in words) of a tensor whose rows (dim 0) may be strided but are
    contiguous inside (4- or 8-byte elements); anything else is made contiguous first"""
    assert t.element_size() in (4, 8)
    k = t.element_size() // 4
    if t.dim() == 0:
        t = t.reshape(1)
    inner = t[0] if t.shape[0] > 0 else t
    if not (t.shape[0] > 0 and inner.is_contiguous() and t.stride(0) >= 0):
        t = t.contiguous()
        inner = t[0]
    return t, t.shape[0], inner.numel() * k, t.stride(0) * k


def gather_rows(dst, items):
    """danbo_gather_rows: items = [(tensor, first 32-bit word in dst), ...] (<= 12; rows of any stride, see row_span) -> dst,
    a contiguous 4-byte-element buffer, in one launch"""
    assert dst.is_contiguous() and dst.element_size() == 4 and len(items) <= _hip.MAX_ROW_SPANS
    spans = (_hip.DanboRowSpan * len(items))()
    keep = []
    for i, (t, off) in enumerate(items):
        t, rows, words, stride = row_span(t)
        assert off + rows * words <= dst.numel()
        keep.append(t)
        spans[i] = _hip.DanboRowSpan(src=t.data_ptr(), dst_word=int(off), src_row_stride_words=int(stride), rows=int(rows),
                                     row_words=int(words))
    _call("danbo_gather_rows", spans, len(items), _p(dst), _stream())
    return dst


def merge_samples(a, b, idx):
    """a [R,S,C], b [R,Sf,C], idx int32 [R,S+Sf] -> [R,S+Sf,C]"""
    R, S = a.shape[:2]
    Sf = b.shape[1]
    C = a[0, 0].numel()
    out = torch.empty((R, S + Sf) + tuple(a.shape[2:]), device=a.device, dtype=torch.float32)
    _call("danbo_merge_samples", _p(_f32(a, "a")), _p(_f32(b, "b")), _p(idx), R, S, Sf, C, _p(out), _stream())
    return out


# -------------------------------------------------------------------------------------- A-NeRF
def anerf_encode(rays_o, rays_d, skts, align, cutoff, tau, L, row0, nrows, z=None, pts=None, out=None):
    """-> x0 [nrows, (1+2L)*24+72], w [nrows,24] for samples [row0, row0+nrows) of the R x S grid."""
    skts = _f32(skts, "skts")
    G = skts.shape[0]
    if pts is not None:
        pts = _f32(pts, "pts")
        R, S = pts.shape[0], pts.shape[1]
        dev = pts.device
        rays_o = rays_d = z = None
    else:
        rays_o, rays_d, z = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z, "z")
        R, S = z.shape
        dev = z.device
    in_ch = (1 + 2 * L) * J + 3 * J
    if out is None:
        x0 = torch.empty(nrows, in_ch, device=dev, dtype=torch.float32)
        w = torch.empty(nrows, J, device=dev, dtype=torch.float32)
    else:
        x0, w = out[0][:nrows], out[1][:nrows]
    _call("danbo_anerf_encode_fwd", _p(rays_o), _p(rays_d), _p(z), _p(pts), R, S, G, _p(skts), _p(_f32(align, "align")),
          _p(_f32(cutoff, "cutoff")), float(tau), int(L), int(row0), int(nrows), _p(x0), _p(w), _stream())
    return x0, w


ANERF_ENC_FLOATS = 144     # the encoder's compact table: [24][4] + [24][2] floats per sample (csrc/k_anerf.hip)
LINEAR16_ENC_K = 448       # k-slots of the recomputed density inputs (14 k-steps)


def anerf_encode_compact(rays_o, rays_d, skts, align, cutoff, tau, row0, nrows, z=None, pts=None, out=None):
    """-> table [nrows, 144] (per joint (cutoff - distance, shifted distance, cutoff weight, direction x), then the 24 (direction y, z)
    pairs), w [nrows, 24]: what linear16_enc recomputes the 24 (1 + 2 L) + 72 density inputs from (576 instead of 1 728 B per sample)"""
    skts = _f32(skts, "skts")
    G = skts.shape[0]
    if pts is not None:
        pts = _f32(pts, "pts")
        R, S = pts.shape[0], pts.shape[1]
        dev = pts.device
        rays_o = rays_d = z = None
    else:
        rays_o, rays_d, z = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z, "z")
        R, S = z.shape
        dev = z.device
    if out is None:
        table = torch.empty(nrows, ANERF_ENC_FLOATS, device=dev, dtype=torch.float32)
        w = torch.empty(nrows, J, device=dev, dtype=torch.float32)
    else:
        table, w = out[0][:nrows], out[1][:nrows]
    _call("danbo_anerf_encode_compact", _p(rays_o), _p(rays_d), _p(z), _p(pts), R, S, G, _p(skts), _p(_f32(align, "align")),
          _p(_f32(cutoff, "cutoff")), float(tau), int(row0), int(nrows), _p(table), _p(w), _stream())
    return table, w


def anerf_view_pe(rays_d, skts, L):
    rays_d, skts = _f32(rays_d, "rays_d"), _f32(skts, "skts")
    R, G = rays_d.shape[0], skts.shape[0]
    E = torch.empty(R, (1 + 2 * L) * 3 * J, device=rays_d.device, dtype=torch.float32)
    _call("danbo_anerf_view_pe_fwd", _p(rays_d), _p(skts), R, G, int(L), _p(E), _stream())
    return E


def transform_batch(vecs, skt, rot_only=False):
    """danbo_transform_batch_pts: vecs [N, S, 3] (points; rot_only: directions) into the local frame of every joint of skt
    [G, J, 4, 4] with G dividing N (ray r belongs to pose r // (N // G)) -> [N, S, J, 3]"""
    for t, nm in ((vecs, "pts"), (skt, "skt")):
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError(f"transform_batch {nm}: expected a float32 CUDA/HIP tensor -- libdanbo_hip has no CPU fallback")
    N, S = vecs.shape[:2]
    G, J = skt.shape[0], skt.shape[-3]
    if vecs.shape[-1] != 3 or skt.shape[-2:] != (4, 4) or G < 1 or N % G:
        raise ValueError(f"transform_batch: pts {tuple(vecs.shape)}, skt {tuple(skt.shape)}")
    vecs, skt = vecs.contiguous(), skt.contiguous()
    out = torch.empty(N, S, J, 3, device=vecs.device, dtype=torch.float32)
    _call("danbo_transform_batch_pts", _p(vecs), _p(skt), N, S, J, N // G, 1 if rot_only else 0, _p(out), _stream())
    return out


def optcodes(codes, idx, mode):
    """danbo_optcodes_fwd: mode 0 rows codes[idx[:, 0]] (clamped), 1 the mean code per row of idx, 2 lerp(codes[idx[:, 0]],
    codes[idx[:, 1]], idx[:, 2]) -> [N, code_ch]"""
    if not codes.is_cuda or codes.dtype != torch.float32 or not idx.is_cuda:
        raise RuntimeError("optcodes: expected CUDA/HIP tensors -- libdanbo_hip has no CPU fallback")
    idx = idx.reshape(idx.shape[0], -1).to(torch.float32).contiguous()
    codes = codes.contiguous()
    out = torch.empty(idx.shape[0], codes.shape[1], device=codes.device, dtype=torch.float32)
    _call("danbo_optcodes_fwd", _p(codes), codes.shape[0], codes.shape[1], _p(idx), idx.shape[1], idx.shape[0], int(mode), _p(out), _stream())
    return out


def small_matmul(a, b, bias=None, out=None):
    """a [M, K] @ b [K, N] (+ bias [N]) -> float32 [M, N]; a, b: float32 CUDA tensors of ANY strides (slices and .t() views are read
    in place); the sum over k accumulated in float64 and rounded once (danbo_small_matmul: one thread per output -- the
    parameter-sized products of a weight refresh, which were float64 torch matmuls, i.e. library GEMMs, until round 5)"""
    for t, nm in ((a, "a"), (b, "b")):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
            raise RuntimeError(f"small_matmul {nm}: expected a 2-D float32 CUDA/HIP tensor -- libdanbo_hip has no CPU fallback")
    M, K = a.shape
    K2, N = b.shape
    if K != K2:
        raise ValueError(f"small_matmul: {tuple(a.shape)} @ {tuple(b.shape)}")
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    if bias is not None:
        bias = _f32(bias, "bias")
    _call("danbo_small_matmul", _p(a), a.stride(0), a.stride(1), _p(b), b.stride(0), b.stride(1), _p(bias), M, N, K, _p(out), out.stride(0),
          _stream())
    return out


def anerf_view_wj(views_w, col0, L):
    """views_linears.0.weight [VW, ld] -> wj [24, 3 (1 + 2 L), VW]: its view columns regrouped per joint (danbo_anerf_view_wj_pack)"""
    views_w = _f32(views_w, "views_w")
    VW, ld = views_w.shape
    wj = torch.empty(J, 3 * (1 + 2 * L), VW, device=views_w.device, dtype=torch.float32)
    _call("danbo_anerf_view_wj_pack", _p(views_w), ld, int(col0), VW, int(L), _p(wj), _stream())
    return wj


def anerf_view_consts(rays_d, skts, L, wj, out=None):
    """C [24, R, VW]: per-ray, per-joint part of A-NeRF's view layer (danbo_anerf_view_consts_fwd: the cutoff view encoding formed in
    the kernel, one fmaf chain per output)"""
    rays_d, skts, wj = _f32(rays_d, "rays_d"), _f32(skts, "skts"), _f32(wj, "wj")
    R, G, VW = rays_d.shape[0], skts.shape[0], wj.shape[2]
    if wj.shape[0] != J or wj.shape[1] != 3 * (1 + 2 * L):
        raise ValueError(f"wj {tuple(wj.shape)} for L = {L}")
    C = out if out is not None else torch.empty(J, R, VW, device=rays_d.device, dtype=torch.float32)
    _call("danbo_anerf_view_consts_fwd", _p(rays_d), _p(skts), R, G, int(L), _p(wj), VW, _p(C), _stream())
    return C


def anerf_view_consts_bwd(rays_d, skts, L, dC, g_views_w, col0):
    """g_views_w[:, col0 : col0 + 72 (1 + 2 L)] = the adjoint of anerf_view_consts with respect to the view columns (overwritten)"""
    rays_d, skts, dC = _f32(rays_d, "rays_d"), _f32(skts, "skts"), _f32(dC, "dC")
    R, G, VW = rays_d.shape[0], skts.shape[0], dC.shape[2]
    g_views_w = _f32(g_views_w, "g_views_w")
    n = _hip.lib().danbo_anerf_view_consts_bwd_scratch_floats(R, int(L), VW)
    scratch = torch.empty(n, device=rays_d.device, dtype=torch.float32)
    _call("danbo_anerf_view_consts_bwd", _p(rays_d), _p(skts), R, G, int(L), _p(dC), VW, _p(g_views_w), g_views_w.shape[1], int(col0),
          _p(scratch), _stream())
    return g_views_w


def _rows(t, name):
    """fp32 CUDA matrix whose rows may be strided (a column slice of a wider buffer) -> (tensor, row stride in floats)"""
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected a float32 CUDA/HIP tensor -- libdanbo_hip has no CPU fallback")
    if t.dim() == 1:
        t = t.unsqueeze(1)
    if t.stride(1) != 1:
        t = t.contiguous()
    return t, t.stride(0)


def anerf_color(featv, w, C, table, cam_idx, ray0, nrays, S, rgb_w, rgb_b, alpha, raw_out):
    """raw_out [R_total,S,4] rows of rays [ray0, ray0+nrays) are written.  featv [rows,VW] and alpha [rows] may be column
    slices of one wider buffer."""
    VW = featv.shape[1]
    R_total = C.shape[1]
    if cam_idx is not None:
        cam_idx = cam_idx.reshape(-1).to(torch.int64).contiguous()
        if not cam_idx.is_cuda:
            raise RuntimeError("cam_idx: expected a CUDA/HIP tensor")
    featv, ldf = _rows(featv, "featv")
    alpha, lda = _rows(alpha, "alpha")
    _call("danbo_anerf_color_fwd", _p(featv), ldf, _p(_f32(w, "w")), _p(_f32(C, "C")), _p(_f32(table, "table")),
          _p(cam_idx), table.shape[0] - 1, R_total, int(ray0), int(nrays), int(S), VW, _p(_f32(rgb_w, "rgb_w")),
          _p(_f32(rgb_b, "rgb_b")), _p(alpha), lda, _p(raw_out), _stream())
    return raw_out


def linear16_color(h, packed, shape, bias, w, C, table, cam_idx, ray0, S, rgb_w, rgb_b, raw_out):
    """A-NeRF's head layer with the colour head as its epilogue (danbo_linear16_fwd_color): h = FragBuffer [M, K1] (the trunk's last
    activation), (packed, shape) = the (VW + 1)-wide head layer packed with frag_in 1; writes raw_out [R_total, S, 4] for the rays
    [ray0, ray0 + M / S)"""
    N, K1, K2 = shape
    if not isinstance(h, FragBuffer) or h.C != K1 or K2:
        raise ValueError("linear16_color: h must be the FragBuffer the layer was packed for")
    VW, M = N - 1, h.M
    if cam_idx is not None:
        cam_idx = cam_idx.reshape(-1).to(torch.int64).contiguous()
    _call("danbo_linear16_fwd_color", _p(h.data), K1, _p(packed), _p(_f32(bias, "bias")), VW, M, _p(_f32(w, "w")), _p(_f32(C, "C")),
          _p(_f32(table, "table")), _p(cam_idx), table.shape[0] - 1, C.shape[1], int(ray0), int(S), _p(_f32(rgb_w, "rgb_w")),
          _p(_f32(rgb_b, "rgb_b")), _p(raw_out), _stream())
    return raw_out


# -------------------------------------------------------------------------------------- dense layer (fp16-split MFMA)
def linear16_pack(weight, K1=None, transposed=False, frag_in=(False, False)):
    """nn.Linear weight [N, K] (or, `transposed`, a [K, N] matrix used as W^T) -> packed fragment buffer.
    K1: columns that multiply the first input of a two-input (skip) layer; the rest multiply the second.
    frag_in: which of the two inputs arrive in fragment order (FragBuffer) -- the k-slots of that part are packed to match."""
    w = _f32(weight, "weight")
    N, K = (w.shape[1], w.shape[0]) if transposed else w.shape
    K1 = K if K1 is None else int(K1)
    nbytes = _hip.lib().danbo_linear16_packed_bytes(N, K1, K - K1)
    if nbytes < 0:
        raise ValueError(f"linear16: unsupported layer shape N={N}, K={K}")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    sn, sk = (1, w.stride(0)) if transposed else (w.stride(0), 1)
    fr = (1 if frag_in[0] else 0) | (2 if frag_in[1] else 0)
    _call("danbo_linear16_pack_frag", _p(w), sn, sk, N, K1, K - K1, fr, _p(packed), _stream())
    return packed, (N, K1, K - K1)


def linear16_pack_enc(weight, L, frag_second=True):
    """nn.Linear weight [N, 24 (1 + 2 L) + 72 (+ K2)] of a layer whose first inputs are A-NeRF's density inputs -> packed buffer
    for linear16_enc (those inputs are recomputed in the kernel from the encoder's compact table); K2: a second, fragment-order
    input (the skip layer)."""
    w = _f32(weight, "weight")
    N, K = w.shape
    K2 = K - (24 * (1 + 2 * L) + 72)
    if K2 < 0 or K2 % 32:
        raise ValueError(f"linear16_pack_enc: {K} columns do not hold the {24 * (1 + 2 * L) + 72} density inputs (+ a multiple of 32)")
    nbytes = _hip.lib().danbo_linear16_packed_bytes(N, LINEAR16_ENC_K, K2)
    if nbytes < 0:
        raise ValueError(f"linear16: unsupported layer shape N={N}, K={K}")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _call("danbo_linear16_pack_enc", _p(w), w.stride(0), 1, N, int(L), K2, 2 if (K2 > 0 and frag_second) else 0, _p(packed), _stream())
    return packed, (N, LINEAR16_ENC_K, K2)


def linear16_enc(table, packed, shape, L, bias=None, relu=False, x2=None, out=None, count=None):
    """out (FragBuffer) = act([enc(table) | x2] W^T + bias): the first / the skip layer of the A-NeRF trunk on the encoder's compact
    table (anerf_encode_compact); x2: None or a FragBuffer."""
    N, K1, K2 = shape
    M = table.shape[0]
    if K1 != LINEAR16_ENC_K or table.shape[1] != ANERF_ENC_FLOATS or not table.is_contiguous() or not isinstance(out, FragBuffer):
        raise ValueError("linear16_enc: table must be [M, 144] contiguous, out a FragBuffer, the layer packed by linear16_pack_enc")
    if (K2 > 0) != (x2 is not None) or (x2 is not None and (not isinstance(x2, FragBuffer) or x2.M != M or x2.C != K2)):
        raise ValueError("linear16_enc: x2 must be the FragBuffer the layer was packed for")
    if out.M != M or out.C != N:
        raise ValueError("linear16_enc: out shape")
    _call("danbo_linear16_fwd_enc", _p(table), int(L), _p(x2.data) if x2 is not None else None, K2, _p(packed), _p(_f32(bias, "bias")), N,
          1 if relu else 0, _p(out.data), M, _p(count), _stream())
    return out


class FragBuffer:
    """A [rows, C] activation in the fragment order of k_linear16 (include/danbo_hip.h, danbo_linear16_fwd_frag): C % 32 == 0,
    rows padded to the 128-row tile, element (16 g + n, 32 s + 16 h + 4 q + i) at [g][s][h][q][n][i].  Lives between the layers
    of a trunk only; `rows()` un-permutes (tests, debugging)."""

    def __init__(self, rows, C, device, storage=None):
        if C % 32:
            raise ValueError("FragBuffer: the width must be a multiple of 32")
        self.M, self.C = int(rows), int(C)
        n = (self.M + 127) // 128 * 128 * self.C
        if storage is not None and storage.numel() < n:
            raise ValueError("FragBuffer: storage too small")
        self.data = torch.empty(n, device=device, dtype=torch.float32) if storage is None else storage

    def rows(self):
        G = (self.M + 127) // 128 * 8
        t = self.data[:G * 16 * self.C].view(G, self.C // 32, 2, 4, 16, 4).permute(0, 4, 1, 2, 3, 5)
        return t.reshape(G * 16, self.C)[:self.M]

    @staticmethod
    def from_rows(x):
        M, C = x.shape
        fb = FragBuffer(M, C, x.device)
        G = (M + 127) // 128 * 8
        pad = torch.zeros(G * 16, C, device=x.device, dtype=torch.float32)
        pad[:M] = x
        fb.data[:G * 16 * C] = pad.view(G, 16, C // 32, 2, 4, 4).permute(0, 2, 3, 4, 1, 5).reshape(-1)
        return fb


def _aligned_rows(t, name):
    """input of linear16: row stride a multiple of 4 floats covering round_up(K, 4) readable, finite columns, 16-byte aligned
    base.  Anything else (e.g. DANBO's 195-wide first layer) is staged through a zero-padded copy."""
    t, ld = _rows(t, name)
    K = t.shape[1]
    if ld % 4 == 0 and ld >= (K + 3) // 4 * 4 and t.data_ptr() % 16 == 0:
        return t, ld
    buf = torch.zeros(t.shape[0], (K + 3) // 4 * 4, device=t.device, dtype=torch.float32)
    buf[:, :K] = t
    return buf[:, :K], buf.stride(0)


def linear16(x1, packed, shape, bias=None, relu=False, x2=None, out=None, count=None):
    """y = act([x1 | x2] W^T + bias) for the rows of x1 (/ x2); `out` may be a column slice of a wider buffer whose row
    stride is a multiple of 4 floats.  x1 / x2 / out may be FragBuffers (the layer must have been packed with the matching
    `frag_in`); supported combinations: rows -> fragments, fragments -> fragments, [rows | fragments] -> fragments, and
    fragments -> rows for N <= 256."""
    N, K1, K2 = shape
    fr = (1 if isinstance(x1, FragBuffer) else 0) | (2 if isinstance(x2, FragBuffer) else 0) | (4 if isinstance(out, FragBuffer) else 0)
    if fr:
        return _linear16_frag(x1, packed, shape, bias, relu, x2, out, count, fr)
    if x1.shape[1] != K1 or (K2 > 0) != (x2 is not None):
        raise ValueError(f"linear16: inputs do not match the packed layer ({K1} + {K2} columns)")
    x1, ld1 = _aligned_rows(x1, "x1")
    M = x1.shape[0]
    ld2 = 0
    if x2 is not None:
        if tuple(x2.shape) != (M, K2):
            raise ValueError("linear16: x2 shape")
        x2, ld2 = _aligned_rows(x2, "x2")
    if out is None:
        out = torch.empty(M, (N + 3) // 4 * 4, device=x1.device, dtype=torch.float32)[:, :N]
    y, ldy = _rows(out, "out")
    if y.data_ptr() != out.data_ptr() or tuple(y.shape) != (M, N) or ldy % 4 or y.data_ptr() % 16:
        raise ValueError("linear16: out must be a [M, N] float32 matrix, unit column stride, rows 16-byte aligned")
    _call("danbo_linear16_fwd", _p(x1), ld1, K1, _p(x2), ld2, K2, _p(packed), _p(_f32(bias, "bias")), N, 1 if relu else 0,
          _p(y), ldy, M, _p(count), _stream())
    return out


def _linear16_frag(x1, packed, shape, bias, relu, x2, out, count, fr):
    N, K1, K2 = shape
    if fr not in (1, 4, 5, 6) or (fr == 1 and N > 256):
        raise ValueError("linear16: unsupported combination of fragment-order inputs / output")
    if (K2 > 0) != (x2 is not None):
        raise ValueError(f"linear16: inputs do not match the packed layer ({K1} + {K2} columns)")

    def operand(x, K, name):
        if isinstance(x, FragBuffer):
            if x.C != K:
                raise ValueError(f"linear16: {name} has {x.C} columns, the packed layer wants {K}")
            return x.data, 0, x.M
        if x.shape[1] != K:
            raise ValueError(f"linear16: {name} has {x.shape[1]} columns, the packed layer wants {K}")
        t, ld = _aligned_rows(x, name)
        return t, ld, t.shape[0]
    a1, ld1, M = operand(x1, K1, "x1")
    a2, ld2 = None, 0
    if x2 is not None:
        a2, ld2, M2 = operand(x2, K2, "x2")
        if M2 != M:
            raise ValueError("linear16: x2 rows")
    if isinstance(out, FragBuffer):
        if out.C != N or out.M != M:
            raise ValueError("linear16: fragment-order out does not match [rows, N]")
        y, ldy = out.data, 0
    else:
        if out is None:
            out = torch.empty(M, (N + 3) // 4 * 4, device=a1.device, dtype=torch.float32)[:, :N]
        y, ldy = _rows(out, "out")
        if y.data_ptr() != out.data_ptr() or tuple(y.shape) != (M, N) or ldy % 4 or y.data_ptr() % 16:
            raise ValueError("linear16: out must be a [M, N] float32 matrix, unit column stride, rows 16-byte aligned")
    _call("danbo_linear16_fwd_frag", _p(a1), ld1, K1, _p(a2), ld2, K2, _p(packed), _p(_f32(bias, "bias")), N, 1 if relu else 0,
          _p(y), ldy, M, _p(count), fr, _stream())
    return out
