"""GPU tests of the entry points: `run_nerf.render_path` against the reference's own `render_path` output (golden
render_path.npz, oracle/gen_golden.py), and a train -> checkpoint -> run_render round trip on the synthetic data source."""
import json
import os

import numpy as np
import pytest
import torch

import danbo_oracle as o
from helpers import ROOT, golden, max_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(x, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(x), dtype=dtype, device=DEV)


def _surreal_caster(g):
    from core.config import parse_args
    from core.raycasters import create_raycaster
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import SMPLSkeleton
    args = parse_args(["--no_reload"], config=os.path.join(ROOT, "danbo-pytorch_amd", "configs", "surreal", "danbo_fast.txt"))
    cfg = syn.model_config("danbo_surreal")
    rest = syn.rest_pose(cfg["rest_scale"])
    da = dict(skel_type=SMPLSkeleton, near=0., far=100., n_views=4, rest_pose=rest, hwf=(40, 32, 50.))
    _, te, *_ = create_raycaster(args, da, device=DEV)
    sd = syn.make_state_dict(cfg, int(g["weight_seed"]), 4, rest)
    te["ray_caster"].network.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    te["ray_caster"].eval()
    return te


@pytest.mark.parametrize("tag", ["bg", "white"])
def test_render_path_matches_reference(tag):
    import run_nerf
    g = golden("render_path")
    kw = dict(_surreal_caster(g), N_samples=int(g["N_samples"]), N_importance=int(g["N_importance"]))
    extra = dict(bg_imgs=g["bg_imgs"], bg_indices=g["bg_indices"]) if tag == "bg" else dict(white_bkgd=True)
    rgbs, disps, accs, idxs, boxes = run_nerf.render_path(T(g["cams"]), (40, 32, float(g["focal"])), 4096, kw, kp=T(g["kps"]),
                                                          skts=T(g["skts"]), bones=T(g["bones"]), ret_acc=True, ext_scale=0.001,
                                                          **extra)
    assert np.array_equal(np.array([[b[0], b[1]] for b in boxes]), g["boxes"])
    assert [len(i) for i in idxs] == list(g["n_valid"])
    assert rgbs.shape == (4, 40, 32, 3) and disps.shape == accs.shape == (4, 40, 32, 1)
    assert max_err(accs, g[f"{tag}_accs"]) < 5e-5 and max_err(rgbs, g[f"{tag}_rgbs"]) < 5e-5       # measured 1.1e-5
    assert o.psnr(rgbs, g[f"{tag}_rgbs"]) > 90.0
    d = g[f"{tag}_disps"]
    assert np.max(np.abs(disps - d) / np.maximum(np.abs(d), 1.0)) < 2e-3
    # outside the boxes the image is exactly the background
    (tl, br) = boxes[0]
    out = np.ones((40, 32), bool)
    out[tl[1]:br[1], tl[0]:br[0]] = False
    assert np.array_equal(rgbs[0][out], g[f"{tag}_rgbs"][0][out])


def test_train_checkpoint_render_round_trip(tmp_path):
    """run_nerf.train on the synthetic source (teacher-rendered targets) writes args.txt + a reference-layout checkpoint;
    run_render loads both and renders a bullet-time sequence, the validation frames (scored) and a density grid."""
    import run_nerf
    import run_render
    cfg = os.path.join(ROOT, "danbo-pytorch_amd", "configs", "surreal", "danbo_fast.txt")
    common = ["--config", cfg, "--basedir", str(tmp_path), "--expname", "demo", "--syn_poses", "2", "--syn_cams", "2",
              "--syn_res", "32", "--syn_rest_scale", "0.714", "--N_rand", "512", "--N_sample_images", "4", "--i_print", "10",
              "--i_weights", "20", "--i_testset", "20", "--render_factor", "0"]
    trainer = run_nerf.train(common + ["--n_iters", "20"])
    log = tmp_path / "demo"
    assert (log / "args.txt").exists() and (log / "config.txt").exists() and (log / "000020.tar").exists()
    ckpt = torch.load(log / "000020.tar", map_location="cpu")
    assert ckpt["global_step"] == 19 and "network_fn_state_dict" in ckpt and "optimizer_state_dict" in ckpt
    scal = [json.loads(l) for l in open(log / "scalars.jsonl")]
    assert any("Val/psnr" in s for s in scal) and any("Stats/psnr" in s for s in scal)
    assert float(open(str(log / "demo_val_000020_") + "psnr.txt").read()) > 5.0
    # resume: picks the checkpoint up (optimizer included) and continues from its step; like the reference the loop restarts
    # at the saved iteration itself (global_step 19 -> i = 20, 21, 22), so the optimizer has taken 20 + 3 steps
    trainer2 = run_nerf.train(common + ["--n_iters", "22"])
    st = trainer2.optimizer.state[trainer2.optimizer.param_groups[0]["params"][0]]["step"]
    assert int(st) == 23
    # the fused step's random stream is part of the checkpoint: the resumed run continued it -- it did not restart at
    # counter 0 -- with the seed the first run drew from
    if trainer.engine is not None and "danbo_rng_state" in ckpt:
        saved = ckpt["danbo_rng_state"]
        assert saved["counter"] > 0 and trainer2.engine.rng_state_dict()["seed"] == saved["seed"]
        assert trainer2.engine.rng_state_dict()["counter"] > saved["counter"]
    base = ["--nerf_args", str(log / "args.txt"), "--ckptpath", str(log / "000020.tar"), "--dataset", "synthetic", "--entry", "val",
            "--outputdir", str(tmp_path / "out")]
    rgbs, accs, boxes, _ = run_render.run_render(base + ["--render_type", "bullet", "--n_bullet", "3", "--selected_idxs", "0", "3",
                                                         "--runname", "bt", "--render_res", "48", "48", "--white_bkgd"])
    assert rgbs.shape == (6, 48, 48, 3) and accs.shape == (6, 48, 48, 1) and len(boxes) == 6
    assert np.isfinite(rgbs).all() and accs.max() > 0.05
    assert np.load(tmp_path / "out" / "bt" / "image.npy").shape == (6, 48, 48, 3)
    # pose interpolation between two frames: n_step blends per interval + the last pose, all from the first camera
    rgbs, accs, boxes, _ = run_render.run_render(base + ["--render_type", "interpolate", "--n_step", "3", "--selected_idxs", "0", "2",
                                                         "--runname", "ip", "--render_res", "32", "32", "--no_save"])
    assert rgbs.shape == (4, 32, 32, 3) and np.isfinite(rgbs).all() and not (tmp_path / "out" / "ip" / "image.npy").exists()
    assert np.abs(rgbs[0] - rgbs[-1]).max() > 1e-3          # the pose really changes along the sequence
    rgbs, _, _, _ = run_render.run_render(base + ["--render_type", "pose_rotate", "--n_bullet", "6", "--selected_idxs", "2", "--runname", "pr",
                                                  "--render_res", "32", "32", "--no_save"])
    assert rgbs.shape == (6, 32, 32, 3) and np.isfinite(rgbs).all() and np.abs(rgbs[0] - rgbs[1]).max() > 1e-3
    rgbs, _, boxes, _ = run_render.run_render(base + ["--render_type", "bubble", "--n_step", "3", "--selected_idxs", "1", "--runname", "bb",
                                                      "--render_res", "32", "32", "--no_save"])
    assert rgbs.shape == (3, 32, 32, 3) and np.isfinite(rgbs).all()
    assert np.abs(rgbs[0] - rgbs[2]).max() < 1e-5 and np.abs(rgbs[0] - rgbs[1]).max() > 1e-3   # closed curve: last = first camera
    rgbs, _, _, scores = run_render.run_render(base + ["--render_type", "val", "--runname", "val", "--render_res", "32", "32", "--eval"])
    assert rgbs.shape[0] == 4 and len(scores["psnr"]) == 4 and np.isfinite(scores["psnr"]).all()
    assert (tmp_path / "out" / "val" / "score_final.txt").exists()
    # the trained weights really are the ones rendered: same frame through the trainer's caster
    w0 = trainer.render_kwargs_train["ray_caster"].network.state_dict()["pts_linears.0.weight"]
    assert torch.equal(ckpt["network_fn_state_dict"]["pts_linears.0.weight"].cpu(), w0.cpu())
    run_render.run_render(base + ["--render_type", "selected", "--selected_idxs", "1", "--runname", "mesh", "--render_mesh",
                                  "--mesh_res", "15", "--mesh_radius", "1.2"])
    sig = np.load(tmp_path / "out" / "mesh" / "meshes" / "000_sigma.npy")
    assert sig.shape == (16, 16, 16) and sig.min() >= 0 and sig.max() > 0
