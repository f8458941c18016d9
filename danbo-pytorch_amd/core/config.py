"""Command-line / config-file flags of the render path.

Same flag names, types and defaults as the reference's `run_nerf.config_parser`
(run_nerf.py:186-572) for every flag the hot path reads; config files
use the reference's `key = value` format (configs/*.txt), booleans are store_true flags and
lists are written `[a, b]`.  Flags of out-of-scope subsystems (datasets, pose optimisation,
logging cadence) are accepted and ignored so the reference's config files parse unchanged.
"""
import argparse

_FLAGS = [
    # name, type, default        (type bool => store_true)
    ("expname", str, None), ("basedir", str, "./logs/"), ("datadir", str, "./data"),
    ("netdepth", int, 8), ("netwidth", int, 256), ("netwidth_view", int, None),
    ("netdepth_fine", int, 8), ("netwidth_fine", int, 256),
    ("N_rand", int, 4096), ("lrate", float, 5e-4), ("lrate_decay", int, 250), ("lrate_decay_rate", float, 0.1),
    ("decay_unit", int, 1000), ("weight_decay", float, None), ("single_net", bool, False),
    ("align_bones", str, "align"), ("coarse_weight", float, 1.0), ("chunk", int, 65536), ("netchunk", int, 65536),
    ("no_reload", bool, False), ("ft_path", str, None), ("finetune", bool, False), ("finetune_light", bool, False),
    ("loss_fn", str, "MSE"), ("rgb_loss_coef", float, 1.0), ("density_scale", float, 1.0),
    ("N_samples", int, 64), ("N_importance", int, 0), ("perturb", float, 1.0), ("use_viewdirs", bool, False),
    ("i_embed", int, 0), ("multires", int, 10), ("multires_views", int, 4), ("multires_bones", int, 0),
    ("raw_noise_std", float, 0.0), ("ray_noise_std", float, 0.0), ("render_factor", int, 0),
    ("nerf_type", str, "nerf"), ("density_type", str, "relu"), ("lindisp", bool, False),
    ("gnn_concat", bool, False), ("adj_self_one", bool, False), ("gnn_backbone", str, "PoolPNGCN"),
    ("node_W", int, 32), ("gcn_D", int, 4), ("gcn_fc_D", int, 1), ("gcn_sep_bias", bool, False),
    ("no_adj", bool, False), ("init_adj_w", float, 0.05), ("aggregate_dim", int, None),
    ("attenuate_feat", bool, False), ("attenuate_invalid", bool, False), ("agg_type", str, "softmax"),
    ("soft_softmax_loss_coef", float, 0.01), ("opt_vol_scale", bool, False), ("vol_cal_scale", bool, False),
    ("vol_scale_penalty", float, 0.01), ("multires_graph", int, 5), ("multires_voxel", int, 5),
    ("voxel_res", int, 4), ("voxel_feat", int, 4), ("align_corners", bool, False), ("graph_input_type", str, "quat"),
    ("agg_backbone", str, "mlp"), ("agg_W", int, 16), ("agg_D", int, 3), ("mask_root", bool, False),
    ("mask_vol_prob", bool, False), ("use_volume_near_far", bool, False), ("detach_agg_grad", bool, False),
    ("opt_framecode", bool, False), ("n_framecodes", int, None), ("framecode_size", int, 16),
    ("opt_posecode", bool, False), ("white_bkgd", bool, False), ("ext_scale", float, 0.001),
    ("use_background", bool, False), ("kp_dist_type", str, "reldist"), ("view_type", str, "relray"),
    ("bone_type", str, "reldir"), ("pts_tr_type", str, "local"), ("ray_tr_type", str, "local"),
    ("use_cutoff", bool, False), ("normalize_cutoff", bool, False), ("cutoff_mm", float, 500),
    ("cutoff_inputs", bool, False), ("cut_to_dist", bool, False), ("cutoff_shift", bool, False),
    ("cutoff_viewdir", bool, False), ("opt_cutoff", bool, False), ("cutoff_step", int, 250),
    ("cutoff_rate", float, 10.0), ("cutoff_bones", bool, False), ("freq_schedule", bool, False),
    ("init_freq", float, 0.0), ("freq_schedule_step", int, 0), ("N_sample_images", int, 8),
    ("n_iters", int, 150000), ("i_weights", int, 10000), ("i_testset", int, 50000), ("debug", bool, False),
    ("input_coords", bool, False), ("cat_coords", bool, False), ("cat_all", bool, False),
    ("i_print", int, 100), ("dataset_type", str, "synthetic"),
    # this build's array sources (core/load_data.py): the reference's HDF5 dataset types are not readable here
    ("syn_poses", int, 8), ("syn_cams", int, 4), ("syn_res", int, 64), ("syn_seed", int, 0), ("syn_rest_scale", float, 0.48),
]


def config_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--config", type=str, default=None, help="config file path (key = value lines)")
    for name, tp, default in _FLAGS:
        if tp is bool:
            p.add_argument(f"--{name}", action="store_true")
        else:
            p.add_argument(f"--{name}", type=tp, default=default)
    return p


def config_file_to_argv(path):
    argv = []
    for line in open(path):
        line = line.split("#")[0].strip()
        if "=" not in line:
            continue
        k, v = [s.strip() for s in line.split("=", 1)]
        if v in ("True", "true"):
            argv.append(f"--{k}")
        elif v in ("False", "false", ""):
            continue
        elif v.startswith("[") and v.endswith("]"):
            argv += [f"--{k}"] + [s.strip() for s in v[1:-1].split(",") if s.strip()]
        else:
            argv += [f"--{k}", v]
    return argv


def parse_args(argv=None, config=None):
    """argv flags override the config file; unknown flags (out-of-scope subsystems) are ignored."""
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    pre, _ = config_parser().parse_known_args(argv)
    config = config or pre.config
    full = (config_file_to_argv(config) if config else []) + argv
    args, _ = config_parser().parse_known_args(full)
    return args


def txt_to_argstring(path, ignore_config=False):
    """`args.txt` written by the trainer (`key = value` per line, every argparse attribute, Python reprs) -> argv
    (reference core/utils/evaluation_helpers.py:221-255).  `None` values are dropped, `True` becomes a bare flag,
    `False` nothing, lists -- real ones or the bracketed strings of config files -- one token per element."""
    import ast
    argv = []
    with open(path, 'r') as f:
        for line in f:
            parts = line.strip().split(' = ')
            if len(parts) != 2:          # no ' = ' (or one inside the value): not a key/value line
                continue
            key, text = parts
            try:
                value = ast.literal_eval(text)
            except (ValueError, SyntaxError):
                value = text
            if value is None or (key == 'config' and ignore_config):
                continue
            if isinstance(value, bool):
                if value:
                    argv.append(f'--{key}')
                continue
            argv.append(f'--{key}')
            if isinstance(value, list):
                argv.extend(str(v) for v in value)
            elif isinstance(value, str) and value[:1] == '[' and value[-1:] == ']':
                argv.extend(t.strip() for t in value[1:-1].split(','))
            else:
                argv.append(text)
    return argv
