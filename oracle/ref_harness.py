"""Import harness for the *reference* DANBO-pytorch implementation (TEST INFRASTRUCTURE ONLY).

This file only works inside the build container, where the read-only reference tree is
mounted at /root/reference.  It is used by `oracle/gen_golden.py` to (a) generate the
golden vectors committed under `tests/golden/` and (b) validate the numpy/C restatement in
`oracle/`.  Nothing here is imported by the product package, by `-m gpu` tests, by
`__graft_entry__.smoke()` or by `bench.py`; `/root/reference` does not exist on the GPU box.

How the reference is made importable on CPU:
  * `cv2` is stubbed (only `cv2.MARKER_CROSS` is touched at import, skeleton_utils.py:1642)
  * `pytorch3d.transforms.rotation_conversions` is stubbed with a restatement of the
    published pytorch3d 0.6 algorithm (axis-angle -> quaternion -> matrix, Taylor branch for
    |theta| < 1e-6).  pytorch3d is an un-vendored third-party dependency of the reference
    (README.md:31-33); the reference holds no test pinning it, so that boundary is
    "parity unpinned" by the reference itself; we cross-check against scipy in the tests.
  * `config_parser` is lifted out of run_nerf.py:186-572 with `ast` at run time and run
    against an argparse shim for `configargparse` (nothing is copied into this repo).
"""
import ast
import argparse
import os
import sys
import types

REF_ROOT = os.environ.get("DANBO_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "core", "networks"))


# ----------------------------------------------------------------------------- stubs
def _p3d_axis_angle_to_quaternion(axis_angle):
    import torch
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    eps = 1e-6
    small = angles.abs() < eps
    s = torch.empty_like(angles)
    s[~small] = torch.sin(half[~small]) / angles[~small]
    s[small] = 0.5 - (angles[small] * angles[small]) / 48
    return torch.cat([torch.cos(half), axis_angle * s], dim=-1)


def _p3d_quaternion_to_matrix(q):
    import torch
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def _p3d_axis_angle_to_matrix(axis_angle):
    return _p3d_quaternion_to_matrix(_p3d_axis_angle_to_quaternion(axis_angle))


def _p3d_matrix_to_axis_angle(m):
    import torch
    from scipy.spatial.transform import Rotation
    shp = m.shape[:-2]
    rv = Rotation.from_matrix(m.reshape(-1, 3, 3).detach().cpu().numpy()).as_rotvec()
    return torch.tensor(rv, dtype=m.dtype).reshape(*shp, 3)


def install_stubs():
    """Install sys.modules stubs and put the reference root on sys.path."""
    sys.dont_write_bytecode = True  # reference dir is read-only
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.MARKER_CROSS = 0
        sys.modules["cv2"] = cv2
    if "pytorch3d" not in sys.modules:
        p3d = types.ModuleType("pytorch3d")
        tr = types.ModuleType("pytorch3d.transforms")
        rc = types.ModuleType("pytorch3d.transforms.rotation_conversions")
        rc.axis_angle_to_matrix = _p3d_axis_angle_to_matrix
        rc.axis_angle_to_quaternion = _p3d_axis_angle_to_quaternion
        rc.matrix_to_axis_angle = _p3d_matrix_to_axis_angle
        rc.quaternion_to_matrix = _p3d_quaternion_to_matrix
        tr.rotation_conversions = rc
        p3d.transforms = tr
        sys.modules["pytorch3d"] = p3d
        sys.modules["pytorch3d.transforms"] = tr
        sys.modules["pytorch3d.transforms.rotation_conversions"] = rc
    # our own drop-in package also calls itself `core`; make sure the reference wins here
    for k in [k for k in sys.modules if k == "core" or k.startswith("core.")]:
        mod = sys.modules[k]
        f = getattr(mod, "__file__", "") or ""
        if not f.startswith(REF_ROOT):
            del sys.modules[k]
    if REF_ROOT in sys.path:
        sys.path.remove(REF_ROOT)
    sys.path.insert(0, REF_ROOT)


# ----------------------------------------------------------------------------- config
class _ShimParser(argparse.ArgumentParser):
    def add_argument(self, *a, **kw):
        kw.pop("is_config_file", None)
        return super().add_argument(*a, **kw)


def _lift_config_parser():
    src = open(os.path.join(REF_ROOT, "run_nerf.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "config_parser"][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    shim = types.ModuleType("configargparse")
    shim.ArgumentParser = _ShimParser
    had = sys.modules.get("configargparse")
    sys.modules["configargparse"] = shim
    ns = {}
    try:
        exec(compile(mod, "<run_nerf.config_parser>", "exec"), ns)
        parser = ns["config_parser"]()
    finally:
        if had is None:
            del sys.modules["configargparse"]
        else:
            sys.modules["configargparse"] = had
    return parser


def lift_functions(rel_path, names, namespace):
    """Compile the named top-level functions of a reference script (whose module-level imports need packages this
    image lacks: h5py, deepdish, imageio, tensorboard ...) into `namespace` and return them.  The source is read
    from /root/reference at run time; nothing is copied into this repo."""
    tree = ast.parse(open(os.path.join(REF_ROOT, rel_path)).read())
    found = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    mod = ast.Module(body=[found[n] for n in names], type_ignores=[])
    exec(compile(mod, f"<{rel_path}>", "exec"), namespace)
    return [namespace[n] for n in names]


def config_file_to_argv(path):
    """`key = value` lines -> argv; booleans are store_true flags; lists are `[a, b]`."""
    argv = []
    for line in open(path):
        line = line.split("#")[0].strip()
        if not line or "=" not in line:
            continue
        k, v = [s.strip() for s in line.split("=", 1)]
        if v in ("True", "true"):
            argv.append(f"--{k}")
        elif v in ("False", "false"):
            continue
        elif v.startswith("[") and v.endswith("]"):
            argv.append(f"--{k}")
            argv += [s.strip() for s in v[1:-1].split(",") if s.strip()]
        else:
            argv += [f"--{k}", v]
    return argv


def parse_reference_config(config_rel, extra_argv=()):
    """Parse e.g. 'configs/h36m_zju/danbo_base.txt' with the reference's own parser."""
    parser = _lift_config_parser()
    argv = config_file_to_argv(os.path.join(REF_ROOT, config_rel)) + list(extra_argv)
    args, _ = parser.parse_known_args(argv)
    return args


# ----------------------------------------------------------------------------- model
def build_reference_caster(args, rest_pose, n_views, tmpdir):
    """create_raycaster (raycasters.py:17-143) with the work-arounds of SURVEY §8c item 4."""
    import numpy as np
    import torch
    install_stubs()
    import core.raycasters as rc
    from core.utils.skeleton_utils import SMPLSkeleton

    os.makedirs(os.path.join(tmpdir, "oracle_exp"), exist_ok=True)
    args.basedir = tmpdir
    args.expname = "oracle_exp"
    args.no_reload = True
    data_attrs = {
        "skel_type": SMPLSkeleton,
        "near": 0.0,
        "far": 100.0,
        "n_views": n_views,
        "rest_pose": np.asarray(rest_pose, dtype=np.float64),  # (i) float64 required
        "hwf": (64, 64, 80.0),
    }
    if args.nerf_type == "nerf":
        # (ii) create_raycaster passes kwargs NeRF.__init__ rejects -> strip them
        orig = rc.create_nerf

        def _create(a, kw, da):
            kw = dict(kw)
            kw.pop("mask_vol_prob", None)
            kw.pop("agg_type", None)
            return orig(a, kw, da)

        rc.create_nerf = _create
        try:
            out = rc.create_raycaster(args, data_attrs)
        finally:
            rc.create_nerf = orig
    else:
        out = rc.create_raycaster(args, data_attrs)
    render_kwargs_train, render_kwargs_test = out[0], out[1]
    caster = render_kwargs_test["ray_caster"]
    return caster, render_kwargs_train, render_kwargs_test
