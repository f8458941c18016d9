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
    "danbo_bone_cull": [P, P, P, P, I, I, I, P, P, P, P, P, P, P, P, P, P, P],
    "danbo_ray_bone_mask": [P, P, P, P, I, I, P, P, P, P, P, P],
    "danbo_bone_gather_fwd": [P, P, P, P, I, I, I, P, P, P, P, P, P, I, P, P],
    "danbo_assign_blend_fwd": [P, P, P, P, I, P, P, P, P, P, P, P, P, P, P],
    "danbo_gather_assign_blend_fwd": [P, P, P, P, I, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P],
    "danbo_assign16_pack": [P, P, P, P, P],
    "danbo_gather_assign_blend16_fwd": [P, P, P, P, I, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P],
    "danbo_mlp_pack": [POINTER(c_void_p), P, P, I, P, P, P],
    "danbo_view_consts": [P, P, I, I, I, I, I, P, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P],
    "danbo_view_code_table": [P, P, I, I, I, P, P, P, P],
    "danbo_mlp16_pack": [POINTER(c_void_p), P, P, P, P, I, P, P, P],
    "danbo_pe_mlp16_fwd": [P, P, P, I, I, P, POINTER(c_void_p), P, P, P, P, P, P, P, P],
    "danbo_transform_batch_pts": [P, P, c_long, I, I, I, I, P, P],
    "danbo_optcodes_fwd": [P, I, I, P, I, c_long, I, P, P],
    "danbo_mlp32_pack": [POINTER(c_void_p), P, P, P, P, I, P, P, P],
    "danbo_pe_mlp32_fwd": [P, P, P, I, I, P, POINTER(c_void_p), P, P, P, P, P, P, P, P],
    "danbo_pe_mlp_fwd": [P, P, P, I, I, P, POINTER(c_void_p), P, P, P, P, P, P, P, P, P],
    "danbo_fill_raw": [P, I, I, P, P],
    "danbo_composite_rays_fwd": [P, P, P, P, P, I, I, F, P, P, P, P, P, P, P, P, P],
    "danbo_importance_samples_rays": [P, P, I, I, I, P, P, P, P, P, P, P],
    "danbo_composite_fwd": [P, P, P, I, I, F, P, P, P, P, P, P, P],
    "danbo_composite_bwd": [P, P, P, I, I, F, P, P, P, P, P],
    "danbo_bone_gather_bwd": [P, P, P, P, I, I, I, P, P, P, P, P, I, P, P, P, P],
    "danbo_importance_samples": [P, P, I, I, I, P, P, P, P, P],
    "danbo_merge_samples": [P, P, P, I, I, I, I, P, P],
    "danbo_composite_importance_fwd": [P, P, P, P, P, I, I, I, F, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "danbo_composite_merged_fwd": [P, P, P, P, P, P, P, P, I, I, I, F, P, P, P, P, P, P, P, P, P, P],
    "danbo_flat_rays": [P, P, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, I, P],
    "danbo_anerf_encode_fwd": [P, P, P, P, I, I, I, P, P, P, F, I, c_long, I, P, P, P],
    "danbo_anerf_encode_compact": [P, P, P, P, I, I, I, P, P, P, F, c_long, I, P, P, P],
    "danbo_anerf_view_pe_fwd": [P, P, I, I, I, P, P],
    "danbo_anerf_color_fwd": [P, I, P, P, P, P, I, I, I, I, I, I, P, P, P, I, P, P],
    "danbo_linear16_set_trace": [P],
    "danbo_linear16_packed_bytes": [I, I, I],
    "danbo_linear16_pack": [P, c_long, c_long, I, I, I, P, P],
    "danbo_linear16_fwd": [P, I, I, P, I, I, P, P, I, I, P, I, I, P, P],
    "danbo_linear16_pack_frag": [P, c_long, c_long, I, I, I, I, P, P],
    "danbo_linear16_fwd_frag": [P, I, I, P, I, I, P, P, I, I, P, I, I, P, I, P],
    "danbo_linear16_pack_enc": [P, c_long, c_long, I, I, I, I, P, P],
    "danbo_linear16_fwd_enc": [P, I, P, I, P, P, I, I, P, I, P, P],
    "danbo_linear16_fwd_color": [P, I, P, P, I, I, P, P, P, P, I, I, I, I, P, P, P, P],
    "danbo_render_frame_workspace": [I, I, I, I, I, I],
    "danbo_render_frame": [P, P, I, I, P, P, c_size_t, P],
    # ---- training step
    "danbo_composite_bwd_lazy": [P, P, P, P, P, I, I, F, P, P, P, P, P],
    "danbo_dw16_scratch_floats": [P, I, I],
    "danbo_dw16": [P, I, I, P, I, P, P],
    "danbo_gather_assign_blend16_train": [P, P, P, I, I, I, P, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P],
    "danbo_train_view_inputs": [P, P, I, I, I, I, I, P, I, I, P, P, I, P],
    "danbo_train_loss_grad": [P, P, P, P, P, P, I, I, I, F, F, P, P, P, P, P, P],
    "danbo_train_draw_unmerge": [P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P, P],
    "danbo_train_mid": [P, P, P, P, P, P, I, I, I, I, I, F, F, F] + [P] * 25,
    "danbo_train_bone_lists": [P, P, P, P, I, I, P, P, P],
    "danbo_assign_blend_bwd": [P, P],
    "danbo_pose_volumes_bwd": [P, I, I, I] + [P] * 24,
    "danbo_adam_step": [P, P, P, P, c_long, F, F, F, F, F, F, F, P],
    "danbo_random_draws": [P, c_long, P, c_long, F, P, P],
    "danbo_gather_rows": [P, I, P, P],
    "danbo_trunk_pack": [P, P],
    "danbo_trunk_fwd": [P, P, I, P],
    "danbo_trunk_bwd": [P, P, P],
    "danbo_trunk_pe_column": [I],
    "danbo_train_cview": [P, I, I, P, P, I, P, P],
    "danbo_train_view_grads": [P, P, P, I, I, P, I, I, P, I, P, P, P, P, P],
    "danbo_train_head_chain": [P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P, P],
    "danbo_train_workspace": [P, I, I, I, I, I],
    "danbo_train_step": [P, P, P, P, c_size_t, P],
    "danbo_train_step_phase": [P, P, P, P, c_size_t, I, P],
    "danbo_train_workspace_view": [P, I, I, I, I, I, P, P],
    "danbo_group_rows": [P, P, P, I, P],
    # ---- A-NeRF on own kernels end to end (ABI 8)
    "danbo_small_matmul": [P, c_long, c_long, P, c_long, c_long, P, I, I, I, P, c_long, P],
    "danbo_anerf_view_wj_pack": [P, I, I, I, I, P, P],
    "danbo_anerf_view_consts_fwd": [P, P, I, I, I, P, I, P, P],
    "danbo_anerf_view_consts_bwd_scratch_floats": [I, I, I],
    "danbo_anerf_view_consts_bwd": [P, P, I, I, I, P, I, P, I, I, P, P],
    "danbo_anerf_color_train_fwd": [P, I, P, P, P, I, I, I, I, P, P, P, I, P, P, P],
    "danbo_anerf_color_bwd_part_floats": [I, I],
    "danbo_anerf_color_bwd": [P, P, P, I, I, I, I, P, P, P, P, P, I, P, P, I, P, P],
    "danbo_anerf_rgb_reduce": [P, c_long, I, P, P, P],
    "danbo_anerf_ray_table": [P, I, I, I, P, P, I, P, I, I, P, P],
    "danbo_anerf_code_grads": [P, P, I, I, I, P, I, P, I, I, P, P, P, P, P],
    "danbo_anerf_relu_mask": [P, P, c_long, P, P, P, P, P, P],
    "danbo_anerf_unmerge": [P, P, P, I, I, I, P, P, P],
    "danbo_anerf_encode_fwd_dtau": [P, P, P, P, I, I, I, P, P, P, P, I, c_long, I, P, P, P],
    "danbo_anerf_train_workspace": [P, I, I, I, I, I],
    "danbo_anerf_train_step": [P, P, P, P, c_size_t, P],
    "danbo_anerf_train_workspace_view": [P, I, I, I, I, I, P, P],
    "danbo_assign16_set_trace": [P],
}
# everything else returns int (0 = ok)
RESTYPES = {"danbo_render_frame_workspace": c_size_t, "danbo_train_workspace": c_size_t,
            "danbo_dw16_scratch_floats": c_long, "danbo_anerf_train_workspace": c_size_t,
            "danbo_anerf_view_consts_bwd_scratch_floats": c_long, "danbo_anerf_color_bwd_part_floats": c_long}


class DanboModel(ctypes.Structure):
    """mirror of `struct DanboModel` in include/danbo_hip.h"""
    _fields_ = ([(n, P) for n in ("g_w0", "g_adjw0", "g_b0", "g_w1", "g_adjw1", "g_b1", "g_w2", "g_b2", "g_w3", "g_b3")]
                + [("L_graph", I), ("graph_width", I), ("align", P), ("axis_scale", P), ("assign16", P)]
                + [(n, P) for n in ("a_b0", "a_b1", "a_w2", "a_b2")] + [("mlp16", P), ("pts_b", P * 8)]
                + [(n, P) for n in ("alpha_w", "alpha_b", "rgb_w", "rgb_b", "views_w_ray_t", "views_b_eff", "framecodes",
                                    "mean_code", "code_table", "empty_consts")]
                + [(n, I) for n in ("n_codes", "code_size", "L_view", "ray_mode", "normalise")]
                + [("density_scale", F), ("use_volume_near_far", I), ("flat_rays_ok", I)])


MAX_ROW_SPANS = 12


class DanboRowSpan(ctypes.Structure):
    _fields_ = [("src", P), ("dst_word", c_long), ("src_row_stride_words", c_long), ("rows", I), ("row_words", I)]


class DanboRays(ctypes.Structure):
    _fields_ = ([(n, P) for n in ("rays_o", "rays_d", "skts", "bones", "cyls", "cam_idx", "near_in", "far_in")]
                + [("R", I), ("G", I), ("chunk", I)])


class DanboFrameOut(ctypes.Structure):
    _fields_ = [(n, P) for n in ("rgb_map", "disp_map", "acc_map", "alpha", "weights", "rgb0", "disp0", "acc0", "alpha0")]



# ---- training step (include/danbo_hip.h: enum DanboTrainTensor and the Danbo{LinearEx,PackDesc,DwLayer,AssignBwd,Train*} structs)
TRAIN_TENSORS = (
    ["graph_net.layers.0.lin.weight", "graph_net.layers.0.adj_w", "graph_net.layers.0.bias", "graph_net.layers.1.lin.weight",
     "graph_net.layers.1.adj_w", "graph_net.layers.1.bias", "graph_net.layers.2.weight", "graph_net.layers.2.bias",
     "graph_net.layers.3.weight", "graph_net.layers.3.bias", "graph_net.axis_scale",
     "prob_linears.layers.0.lin.weight", "prob_linears.layers.0.adj_w", "prob_linears.layers.0.bias", "prob_linears.layers.1.weight",
     "prob_linears.layers.1.bias", "prob_linears.layers.2.weight", "prob_linears.layers.2.bias"]
    + [f"pts_linears.{i}.weight" for i in range(8)] + [f"pts_linears.{i}.bias" for i in range(8)]
    + ["alpha_linear.weight", "alpha_linear.bias", "feature_linear.weight", "feature_linear.bias", "views_linears.0.weight",
       "views_linears.0.bias", "rgb_linear.weight", "rgb_linear.bias", "framecodes.codes.weight"])
N_TRAIN_TENSORS = len(TRAIN_TENSORS)   # DANBO_T_COUNT


class DanboDwLayer(ctypes.Structure):
    _fields_ = ([(n, P) for n in ("dy", "x1", "x2", "dy_maxabs", "gw", "gw2", "gb", "gb2")]
                + [(n, I) for n in ("ldy", "ld1", "ld2", "N", "K1", "K2", "split_n", "frag", "gw_ld", "gw_col0", "x1_pe")])


class DanboAssignBwd(ctypes.Structure):
    _fields_ = ([(n, P) for n in ("rays_o", "rays_d", "z_c", "z_f", "skts", "align", "axis_scale", "volumes")]
                + [(n, I) for n in ("R", "S", "Sf", "G", "rows_cap")]
                + [(n, P) for n in ("row_sample", "row_ray", "cnt", "lists", "cntb", "h_rows", "d_h", "label_c", "label_f", "bits_c",
                                    "bits_f", "w0", "adj_w", "adj", "b0", "w1", "b1", "w2", "b2", "g_w0", "g_adj_w", "g_b0", "g_w1",
                                    "g_b1", "g_w2", "g_b2", "g_vol", "g_scale")]
                + [("c_ss", F), ("loss", P), ("d_p", P)])


class DanboTrunkWeights(ctypes.Structure):
    _fields_ = ([("pts_w", P * 8), ("pts_b", P * 8)]
                + [(n, P) for n in ("alpha_w", "alpha_b", "feature_w", "feature_b", "views_w", "views_b", "rgb_w", "rgb_b")]
                + [("view_ch", I)] + [(n, P) for n in ("packed", "wfv", "b_eff", "wmax", "winv")])


class DanboTrunkRows(ctypes.Structure):
    _fields_ = ([(n, P) for n in ("cnt", "row_sample", "h_rows", "cview")] + [(n, I) for n in ("R", "S", "Sf", "rows_cap")]
                + [("rows_pad", c_long)]
                + [(n, P) for n in ("y", "pe", "relu", "hv", "hv_bits", "raw_rows", "raw_c", "raw_f", "raw_empty", "row_ray",
                                    "d_raw_c", "d_raw_f", "d_raw_rows", "dz", "dpre_v", "d_alpha4", "d_h", "maxabs")])


class DanboTrainModel(ctypes.Structure):
    _fields_ = ([("p", P * N_TRAIN_TENSORS), ("g", P * N_TRAIN_TENSORS), ("g_flat", P), ("n_flat", c_long)]
                + [(n, P) for n in ("g_adj0", "g_adj1", "a_adj", "align", "init_scale")]
                + [(n, I) for n in ("L_graph", "graph_width", "L_view", "L_voxel", "ray_mode", "normalise", "n_codes", "code_size",
                                    "view_ch", "use_volume_near_far", "loss_mse", "use_background")]
                + [(n, F) for n in ("density_scale", "rgb_loss_coef", "coarse_weight", "soft_softmax_coef", "vol_scale_penalty")])


class DanboTrainBatch(ctypes.Structure):
    _fields_ = ([(n, P) for n in ("rays_o", "rays_d", "skts", "bones", "cyls", "near_in", "far_in", "cam_idx", "target", "bgs",
                                  "t_rand", "u_rand", "noise_c", "noise_f")]
                + [(n, I) for n in ("R", "G", "S", "Sf", "chunk")]
                + [(n, P) for n in ("rng_state", "rng_uniform", "rng_normal")]
                + [("n_uniform", ctypes.c_longlong), ("n_normal", ctypes.c_longlong), ("normal_std", F)])


class DanboTrainOut(ctypes.Structure):
    _fields_ = [(n, P) for n in ("rgb_map", "disp_map", "acc_map", "alpha", "weights", "rgb0", "disp0", "acc0", "alpha0", "loss",
                                 "counts")]


ANERF_MAX_D = 8


class DanboAnerfTrainModel(ctypes.Structure):
    """mirror of `struct DanboAnerfTrainModel` in include/danbo_hip.h"""
    _fields_ = ([(n, I) for n in ("D", "W", "VW", "skip", "L", "L_view", "n_codes", "code_size")]
                + [("pts_w", P * ANERF_MAX_D), ("pts_b", P * ANERF_MAX_D)]
                + [(n, P) for n in ("alpha_w", "alpha_b", "feature_w", "feature_b", "views_w", "views_b", "rgb_w", "rgb_b", "codes")]
                + [("g_pts_w", P * ANERF_MAX_D), ("g_pts_b", P * ANERF_MAX_D)]
                + [(n, P) for n in ("g_alpha_w", "g_alpha_b", "g_feature_w", "g_feature_b", "g_views_w", "g_views_b", "g_rgb_w", "g_rgb_b",
                                    "g_codes", "g_flat")]
                + [("n_flat", c_long)] + [(n, P) for n in ("align", "cutoff", "tau")]
                + [("loss_mse", I), ("use_background", I)] + [(n, F) for n in ("density_scale", "rgb_loss_coef", "coarse_weight")])


class DanboTrainView(ctypes.Structure):
    _fields_ = [(n, P) for n in ("z_coarse", "z_fine", "z_sorted", "order", "bits_coarse", "bits_fine")]


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
