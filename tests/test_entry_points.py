"""Host logic of the entry points (run_nerf.py / run_render.py) against goldens generated from the reference
(oracle/gen_golden.py: gen_sequences) -- CPU only; the GPU side is tests/test_gpu_entry_points.py."""
import numpy as np
import pytest
import torch

from helpers import golden


@pytest.fixture(scope="module")
def seq():
    return golden("sequences")


def test_generate_bullet_time_axes(seq):
    from core.load_data import generate_bullet_time
    for ax in "xyz":
        np.testing.assert_allclose(generate_bullet_time(seq["c2ws"][1], 5, ax), seq[f"ring_{ax}"], atol=1e-6)
    with pytest.raises(NotImplementedError):
        generate_bullet_time(seq["c2ws"][1], 5, "w")


NAMES = ("kps", "skts", "c2ws", "cam_idxs", "focals", "bones", "centers")


@pytest.mark.parametrize("tag,kw", [("bt", {}), ("bt_nokp", dict(center_kps=False)),
                                    ("bt_raw", dict(center_kps=False, center_cam=False, undo_rot=True))])
def test_load_bullettime_matches_reference(seq, tag, kw):
    import run_render
    out = run_render.load_bullettime(seq["kps"].copy(), seq["bones"].copy(), seq["c2ws"].copy(), seq["focals"].copy(), seq["rest"],
                                     seq["sel"], n_bullet=3, centers=seq["centers"].copy(), **kw)
    for n, v in zip(NAMES, out):
        np.testing.assert_allclose(v, seq[f"{tag}_{n}"], atol=2e-6, err_msg=n)
    assert out[0].shape == (9, 24, 3) and out[2].shape == (9, 4, 4)


@pytest.mark.parametrize("tag,kw", [("ip", {}), ("ip_c", dict(center_cam=True)), ("ip_k", dict(center_kps=True))])
def test_load_interpolate_matches_reference(seq, tag, kw):
    import run_render
    out = run_render.load_interpolate(seq["kps"].copy(), seq["bones"].copy(), seq["c2ws"].copy(), seq["focals"].copy(), seq["rest"],
                                      seq["sel"], n_step=4, **kw)
    for n, v in zip(NAMES[:5], out):
        np.testing.assert_allclose(v, seq[f"{tag}_{n}"], atol=2e-6, err_msg=n)
    assert len(out[5]) == len(out[0]) == 9          # 2 intervals x 4 steps + the last pose


def test_load_selected_matches_reference(seq):
    import run_render
    out = run_render.load_selected(seq["kps"].copy(), seq["bones"].copy(), seq["c2ws"].copy(), seq["focals"].copy(), seq["rest"],
                                   seq["sel"], centers=seq["centers"].copy())
    for n, v in zip(NAMES, out):
        np.testing.assert_allclose(v, seq[f"sel_{n}"], atol=2e-6, err_msg=n)


def test_load_bubble_matches_reference(seq):
    import run_render
    out = run_render.load_bubble(seq["kps"].copy(), seq["bones"].copy(), seq["c2ws"].copy(), seq["focals"].copy(), seq["rest"],
                                 seq["sel"], centers=seq["centers"].copy(), n_step=4)
    for n, v in zip(NAMES, out):
        np.testing.assert_allclose(v, seq[f"bb_{n}"], atol=2e-6, err_msg=n)
    assert out[2].shape == (12, 4, 4) and out[5].shape == (3, 24, 3)        # bones: once per pose, as in the reference


def test_load_pose_rotate_matches_reference(seq):
    """root-rotation sweep; the reference converts through pytorch3d (restated in oracle/ref_harness.py), here scipy: rotation
    vectors agree as ROTATIONS (a turn by pi has two axis-angle forms), so the matrices derived from them are compared"""
    import run_render
    from scipy.spatial.transform import Rotation
    out = run_render.load_pose_rotate(seq["kps"].copy(), seq["bones"].copy(), seq["c2ws"].copy(), seq["focals"].copy(), seq["rest"],
                                      np.array([2]), n_bullet=9)
    names = ("kps", "skts", "bones", "c2ws", "cam_idxs", "focals")
    for n, v in zip(names, out):
        if n == "bones":
            a = Rotation.from_rotvec(v.reshape(-1, 3).astype(np.float64)).as_matrix()
            b = Rotation.from_rotvec(seq["pr_bones"].reshape(-1, 3).astype(np.float64)).as_matrix()
            np.testing.assert_allclose(a, b, atol=2e-6)
        else:
            np.testing.assert_allclose(v, seq[f"pr_{n}"], atol=5e-6, err_msg=n)
    assert out[0].shape == (9, 24, 3)
    with pytest.raises(ValueError):
        run_render.load_pose_rotate(seq["kps"], seq["bones"], seq["c2ws"], seq["focals"], seq["rest"], np.array([0, 1]))


def test_sequence_loaders_leave_inputs_untouched(seq):
    import run_render
    kps, bones, c2ws = seq["kps"].copy(), seq["bones"].copy(), seq["c2ws"].copy()
    run_render.load_bullettime(kps, bones, c2ws, seq["focals"], seq["rest"], seq["sel"], n_bullet=2, undo_rot=True)
    run_render.load_interpolate(kps, bones, c2ws, seq["focals"], seq["rest"], seq["sel"], n_step=2, center_cam=True, undo_rot=True)
    assert np.array_equal(kps, seq["kps"]) and np.array_equal(bones, seq["bones"]) and np.array_equal(c2ws, seq["c2ws"])


# ---------------------------------------------------------------------------------------------------- metrics
def test_psnr_ssim_identities():
    from core.utils.evaluation_helpers import evaluate_metric, ssim_map
    rng = np.random.default_rng(0)
    gt = rng.uniform(size=(2, 24, 20, 3)).astype(np.float32)
    x = torch.tensor(gt).permute(0, 3, 1, 2)
    assert torch.allclose(ssim_map(x, x), torch.ones_like(x), atol=1e-5)          # SSIM(x, x) = 1 everywhere
    noisy = np.clip(gt + rng.normal(0, 0.1, gt.shape).astype(np.float32), 0, 1)
    s = ssim_map(torch.tensor(noisy).permute(0, 3, 1, 2), x)
    assert s.shape == x.shape and float(s.mean()) < 0.99 and float(s.max()) <= 1.0 + 1e-5
    m = evaluate_metric(noisy, gt)
    want = np.mean([-10 * np.log10(np.mean((a - b) ** 2)) for a, b in zip(noisy, gt)])
    assert abs(m["psnr"] - want) < 1e-4 and m["psnr_fg"] is None
    # constant offset: PSNR known in closed form
    m = evaluate_metric(np.clip(gt * 0 + 0.5, 0, 1), gt * 0 + 0.6)
    assert abs(m["psnr"] - 20.0) < 1e-3


def test_ssim_map_known_answer():
    """ssim_map against an INDEPENDENT float64 scipy evaluation of the published algorithm on committed images
    (oracle/gen_ssim_golden.py -> tests/golden/ssim_known_answer.npz): the zero-padded image-sized map, and -- its interior -- the
    un-padded map of upstream pytorch-msssim, computed here a second way (valid convolutions)."""
    import torch.nn.functional as F
    from core.utils.evaluation_helpers import ssim_map, _gauss
    from helpers import golden
    g = golden("ssim_known_answer")
    x, y = torch.tensor(g["pred"]), torch.tensor(g["gt"])
    m = ssim_map(x, y).numpy()
    assert m.shape == g["map_same"].shape and np.abs(m - g["map_same"]).max() < 2e-5
    assert np.abs(m[:, :, 5:-5, 5:-5] - g["map_valid"]).max() < 2e-5
    assert float(np.abs(m[0, :, :4, :] - 1.0).max()) < 1e-5                 # the error-free rows, a window away from the others
    # the interior again through VALID convolutions (no padding anywhere in the computation)
    w = _gauss().double()
    blur = lambda t: F.conv2d(F.conv2d(t, w.view(1, 1, -1, 1).expand(3, 1, -1, 1), groups=3), w.view(1, 1, 1, -1).expand(3, 1, 1, -1), groups=3)  # noqa: E731
    xd, yd = x.double(), y.double()
    mu1, mu2 = blur(xd), blur(yd)
    s1, s2, s12 = blur(xd * xd) - mu1 * mu1, blur(yd * yd) - mu2 * mu2, blur(xd * yd) - mu1 * mu2
    v = (2 * mu1 * mu2 + 1e-4) / (mu1 * mu1 + mu2 * mu2 + 1e-4) * (2 * s12 + 9e-4) / (s1 + s2 + 9e-4)
    assert np.abs(v.numpy() - g["map_valid"]).max() < 5e-6            # (the product's window is built in float32)


def test_evaluate_metric_masks(tmp_path):
    from core.utils.evaluation_helpers import evaluate_metric
    rng = np.random.default_rng(1)
    gt = rng.uniform(size=(3, 16, 16, 3)).astype(np.float32)
    pred = gt.copy()
    pred[:, :8] += 0.1                                     # error only in the top half
    fg = np.zeros((3, 16, 16, 1), np.float32)
    fg[:2, 8:] = 1                                         # foreground = error-free bottom half; image 2 has no person
    m = evaluate_metric(pred, gt, gt_masks=fg)
    assert m["psnr"] == m["psnr_fg"] == 0.0                # zero error -> inf -> 0 (reference's convention)
    fg[:2] = 0
    fg[:2, :8] = 1
    m = evaluate_metric(pred, gt, gt_masks=fg)
    assert abs(m["psnr_fg"] - 20.0) < 1e-3                 # mse 0.01 inside the mask
    valid = [torch.arange(0, 256), torch.arange(0, 128), torch.arange(0, 256)]
    base = str(tmp_path / "val_")
    m = evaluate_metric(pred, gt, gt_masks=fg, valid_idxs=valid, eval_both=True, vid_base=base)
    # image 0: box = whole image -> mse 0.005; image 1: box = top half -> mse 0.01; image 2 dropped (empty mask)
    want = np.mean([-10 * np.log10(0.005), -10 * np.log10(0.01)])
    assert abs(m["psnr"] - want) < 1e-3 and abs(m["psnr_fg"] - 20.0) < 1e-3
    assert float(open(base + "psnr.txt").read()) == pytest.approx(want, abs=1e-3)
    assert float(open(base + "psnr_fg.txt").read()) == pytest.approx(20.0, abs=1e-3)


def test_evaluate_in_boxes():
    from core.utils.evaluation_helpers import evaluate_in_boxes
    gt = np.full((2, 12, 10, 3), 0.5, np.float32)
    rgb = gt.copy()
    rgb[0, 2:6, 3:7] += 0.1
    mask = np.zeros((2, 12, 10, 1), np.float32)
    mask[0, 2:6, 3:7] = 1                                   # frame 1: empty mask inside its box -> skipped
    boxes = [((3, 2), (7, 6)), ((0, 0), (4, 4))]
    s = evaluate_in_boxes(rgb, None, boxes, gt, mask)
    assert len(s["psnr"]) == 1 and abs(s["psnr"][0] - 20.0) < 1e-3 and abs(s["fg_psnr"][0] - 20.0) < 1e-3
    s = evaluate_in_boxes(rgb, None, boxes, gt)
    assert len(s["psnr"]) == 2 and s["fg_psnr"] == []


# ---------------------------------------------------------------------------------------------------- data layer
def _toy_dataset(n=6, H=16, W=20):
    from core.load_data import PoseImageDataset, synthetic_arrays
    arr = synthetic_arrays(n_poses=3, n_cams=2, H=H, W=W, pose_seed=1)
    rng = np.random.default_rng(0)
    imgs = rng.uniform(size=(n, H, W, 3)).astype(np.float32)
    fgs = np.zeros((n, H, W, 1), np.float32)
    fgs[:, 4:12, 6:14] = 1
    return PoseImageDataset(imgs, fgs, np.ones((1, H, W, 3), np.float32), np.zeros(n, np.int64), arr["c2ws"], arr["focals"],
                            arr["kp3d"], arr["bones"], arr["skts"], arr["rest_pose"]), arr


def test_dataset_batch_layout_and_rank_sharding():
    from core.utils.ray_utils import get_rays
    ds, arr = _toy_dataset()
    b = ds.sample_batch(N_images=4, N_rand=64)
    assert b["N_uniques"] == 4 and b["rays_o"].shape == (64, 3) and b["skts"].shape == (64, 24, 4, 4)
    assert b["cyls"].shape == (64, 5) and b["cam_idxs"].dtype == torch.int64
    # pose tensors are constant within each image's block of 16 rays (what N_uniques promises the network)
    blocks = b["bones"].reshape(4, 16, 24, 3)
    assert torch.equal(blocks, blocks[:, :1].expand_as(blocks))
    # rays are the pinhole rays of the sampled pixels, targets the image values there, pixels inside the grown mask
    i = int(b["kp_idx"][0])
    ro, rd = get_rays(16, 20, float(ds.focals[i]), torch.tensor(ds.c2ws[i]))
    rd = rd.reshape(-1, 3)
    d = (rd[None] - b["rays_d"][:16, None]).abs().sum(-1)
    pix = d.argmin(1)
    assert float(d.min(1).values.max()) < 1e-5
    assert torch.allclose(torch.tensor(ds.imgs[i].reshape(-1, 3))[pix], b["target_s"][:16])
    assert set(pix.tolist()) <= set(ds.sampling_idxs[i].tolist())
    # two ranks drawing from equally seeded generators split the same batch by whole images
    ds0, _ = _toy_dataset()
    ds1, _ = _toy_dataset()
    full, _ = _toy_dataset()
    f = full.sample_batch(4, 64)
    h0, h1 = ds0.sample_batch(4, 64, rank=0, world=2), ds1.sample_batch(4, 64, rank=1, world=2)
    assert h0["N_uniques"] == h1["N_uniques"] == 2
    for k in ("rays_d", "target_s", "kp3d", "cam_idxs"):
        assert torch.equal(torch.cat([h0[k], h1[k]]), f[k])


def test_dataset_meta_and_render_data_keys():
    ds, _ = _toy_dataset()
    meta, rd = ds.get_meta(), ds.get_render_data()
    for k in ("hwf", "center", "c2ws", "near", "far", "n_views", "skel_type", "rest_pose", "kp3d", "skts", "bones"):
        assert k in meta
    assert meta["rest_pose"].dtype == np.float64 and meta["n_views"] == 6
    for k in ("imgs", "fgs", "bgs", "bg_idxs", "cam_idxs", "c2ws", "hwf", "center", "kp3d", "skts", "bones"):
        assert k in rd
    assert rd["imgs"].shape == (6, 16, 20, 3)


def test_filter_state_dict_rules():
    from core.raycasters import filter_state_dict
    model = {"a.weight": torch.zeros(3, 4), "framecodes.codes.weight": torch.zeros(5, 2), "pe_fn.tau": torch.tensor(20.),
             "pe_fn.cutoff_dist": torch.full((24,), 0.5), "b.weight": torch.zeros(2, 2), "c.weight": torch.zeros(1)}
    ckpt = {"a.weight": torch.ones(3, 4), "framecodes.codes.weight": torch.tensor([[1., 3.], [3., 5.]]), "b.weight": torch.ones(3, 3),
            "extra": torch.ones(1)}
    out = filter_state_dict(model, ckpt)
    assert torch.equal(out["a.weight"], torch.ones(3, 4))
    assert torch.equal(out["framecodes.codes.weight"], torch.tensor([[2., 4.]]).repeat(5, 1))      # mean code in every row
    assert float(out["pe_fn.tau"]) == 1000. and torch.equal(out["pe_fn.cutoff_dist"], model["pe_fn.cutoff_dist"])
    assert "b.weight" not in out and "c.weight" not in out and "extra" not in out


def test_render_and_batchify_rays_shapes():
    from core.trainer import batchify_rays, render
    calls = []

    def caster(rays, kp_batch=None, flag=None):
        calls.append((rays.shape[0], kp_batch.shape[0], flag))
        return {"rgb_map": rays[:, 3:6] * 2, "acc_map": rays[:, 0]}
    rays = torch.arange(10 * 11, dtype=torch.float32).reshape(10, 11)
    out = batchify_rays(rays, chunk=4, ray_caster=caster, kp_batch=torch.zeros(10, 24, 3), flag="x")
    assert calls == [(4, 4, "x"), (4, 4, "x"), (2, 2, "x")] and torch.equal(out["rgb_map"], rays[:, 3:6] * 2)
    ro, rd = torch.zeros(2, 3, 3), torch.ones(2, 3, 3) * 2
    out = render(4, 4, 10., chunk=5, rays=(ro, rd), use_viewdirs=True, ray_caster=caster, kp_batch=torch.zeros(6, 24, 3))
    assert out["rgb_map"].shape == (2, 3, 3) and out["acc_map"].shape == (2, 3)
