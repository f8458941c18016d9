"""ctypes binding of libdanbo_hip.so (C ABI declared in include/danbo_hip.h).

The library is the only compute backend of this package: there is NO PyTorch/CPU fallback.
`lib()` raises if the shared object is missing, and every wrapper in hip_ops raises if a
tensor is not a CUDA(HIP) tensor.
"""
import ctypes
import os
from ctypes import c_long, c_char_p, c_float, c_int, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DANBO_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libdanbo_hip.so")

P = c_void_p  # device pointer
I = c_int
F = c_float

# name -> argtypes, exactly as declared in include/danbo_hip.h
SIGNATURES = {
    "danbo_abi_version": [],
    "danbo_device_info": [POINTER(c_int), POINTER(c_int), c_char_p, I],
    "danbo_pose_volumes_fwd": [P, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "danbo_near_far_cylinder": [P, P, P, I, I, F, F, P, P, I, P, P, P, P],
    "danbo_near_far_boxes": [P, P, P, P, P, I, I, P, P, P],
    "danbo_coarse_samples": [P, P, I, I, P, P, P],
    "danbo_bone_cull": [P, P, P, P, I, I, I, P, P, P, P, P, P, P],
    "danbo_bone_gather_fwd": [P, P, P, P, I, I, I, P, P, P, P, P, P, I, P, P],
    "danbo_assign_blend_fwd": [P, P, P, P, I, P, P, P, P, P, P, P, P, P, P],
    "danbo_gather_assign_blend_fwd": [P, P, P, P, I, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P],
    "danbo_assign16_pack": [P, P, P, P, P],
    "danbo_gather_assign_blend16_fwd": [P, P, P, P, I, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P],
    "danbo_mlp_pack": [POINTER(c_void_p), P, P, I, P, P, P],
    "danbo_view_consts": [P, P, I, I, I, I, I, P, I, I, P, P, P, P, P, P, P, I, P, P, P, P],
    "danbo_view_code_table": [P, P, I, I, I, P, P, P, P],
    "danbo_mlp16_pack": [POINTER(c_void_p), P, P, P, P, I, P, P, P],
    "danbo_pe_mlp16_fwd": [P, P, P, I, I, P, POINTER(c_void_p), P, P, P, P, P, P, P, P],
    "danbo_pe_mlp_fwd": [P, P, P, I, I, P, POINTER(c_void_p), P, P, P, P, P, P, P, P, P],
    "danbo_fill_raw": [P, I, I, P, P],
    "danbo_composite_fwd": [P, P, P, I, I, F, P, P, P, P, P, P, P],
    "danbo_composite_bwd": [P, P, P, I, I, F, P, P, P, P, P],
    "danbo_bone_gather_bwd": [P, P, P, P, I, I, I, P, P, P, P, P, I, P, P, P, P],
    "danbo_importance_samples": [P, P, I, I, I, P, P, P, P, P],
    "danbo_merge_samples": [P, P, P, I, I, I, I, P, P],
    "danbo_composite_importance_fwd": [P, P, P, P, P, I, I, I, F, P, P, P, P, P, P, P, P, P, P, P],
    "danbo_composite_merged_fwd": [P, P, P, P, P, P, P, P, I, I, I, F, P, P, P, P, P, P, P, P],
    "danbo_anerf_encode_fwd": [P, P, P, P, I, I, I, P, P, P, F, I, c_long, I, P, P, P],
    "danbo_anerf_view_pe_fwd": [P, P, I, I, I, P, P],
    "danbo_anerf_color_fwd": [P, I, P, P, P, P, I, I, I, I, I, I, P, P, P, I, P, P],
    "danbo_linear16_set_trace": [P],
    "danbo_linear16_packed_bytes": [I, I, I],
    "danbo_linear16_pack": [P, c_long, c_long, I, I, I, P, P],
    "danbo_linear16_fwd": [P, I, I, P, I, I, P, P, I, I, P, I, I, P, P],
    "danbo_render_frame_workspace": [I, I, I, I, I, I],
    "danbo_render_frame": [P, P, I, I, P, P, c_size_t, P],
}
RESTYPES = {"danbo_render_frame_workspace": c_size_t}   # everything else returns int (0 = ok)


class DanboModel(ctypes.Structure):
    """mirror of `struct DanboModel` in include/danbo_hip.h"""
    _fields_ = ([(n, P) for n in ("g_w0", "g_adjw0", "g_b0", "g_w1", "g_adjw1", "g_b1", "g_w2", "g_b2", "g_w3", "g_b3")]
                + [("L_graph", I), ("graph_width", I), ("align", P), ("axis_scale", P), ("assign16", P)]
                + [(n, P) for n in ("a_b0", "a_b1", "a_w2", "a_b2")] + [("mlp16", P), ("pts_b", P * 8)]
                + [(n, P) for n in ("alpha_w", "alpha_b", "rgb_w", "rgb_b", "views_w_ray_t", "views_b_eff", "framecodes",
                                    "mean_code", "code_table", "empty_consts")]
                + [(n, I) for n in ("n_codes", "code_size", "L_view", "ray_mode", "normalise")]
                + [("density_scale", F), ("use_volume_near_far", I)])


class DanboRays(ctypes.Structure):
    _fields_ = ([(n, P) for n in ("rays_o", "rays_d", "skts", "bones", "cyls", "cam_idx", "near_in", "far_in")]
                + [("R", I), ("G", I), ("chunk", I)])


class DanboFrameOut(ctypes.Structure):
    _fields_ = [(n, P) for n in ("rgb_map", "disp_map", "acc_map", "alpha", "weights", "rgb0", "disp0", "acc0", "alpha0")]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C danbo-pytorch_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU / PyTorch fallback.")
        import torch  # noqa: F401  (loads torch's bundled libamdhip64 first so both share one HIP runtime)
        l = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, c_int)
        _lib = l
    return _lib


class HipError(RuntimeError):
    pass


def check(code, name):
    if code != 0:
        raise HipError(f"{name} failed with code {code}" + (" (invalid argument)" if code == -22 else ""))
