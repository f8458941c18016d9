"""Image metrics and `args.txt` parsing for the entry points (reference: core/utils/evaluation_helpers.py:221-385 and
run_render.py:1178-1263).

PSNR is the reference's formula.  SSIM: the reference imports a pinned fork of a third-party package
(`pytorch-msssim @ git+https://github.com/LemonATsu/pytorch-msssim.git@f77a2446`, requirements.txt:7) that returns the
per-pixel SSIM *map* -- the package is not in this image, so `ssim_map` restates the published algorithm (Wang et al. 2004 as
implemented by pytorch-msssim: 11-tap Gaussian window, sigma 1.5, K1 = 0.01, K2 = 0.03, separable filtering) with zero padding
so the map has the image's size.  **SSIM parity is unpinned** (no reference output to compare with); PSNR needs no third-party code.
What the reference's call sites fix about the fork (4-D image-sized map, default window, unit data range) and what they leave open
(the padding mode: only the 5-pixel border of the map depends on it) is written out in oracle/gen_ssim_golden.py, whose
known-answer vector pins this function's arithmetic (tests/test_host_logic.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..config import txt_to_argstring  # noqa: F401  (reference location of this helper)


def to8b(x):
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def _gauss(size=11, sigma=1.5):
    x = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _blur(x, g):
    C, p = x.shape[1], g.numel() // 2
    x = F.conv2d(x, g.view(1, 1, -1, 1).expand(C, 1, -1, 1), padding=(p, 0), groups=C)
    return F.conv2d(x, g.view(1, 1, 1, -1).expand(C, 1, 1, -1), padding=(0, p), groups=C)


def ssim_map(x, y, data_range=1.0, win_size=11, win_sigma=1.5, K=(0.01, 0.03)):
    """x, y [N,C,H,W] -> per-pixel, per-channel SSIM [N,C,H,W]"""
    g = _gauss(win_size, win_sigma).to(x.device, x.dtype)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _blur(x, g), _blur(y, g)
    s1, s2, s12 = _blur(x * x, g) - mu1 * mu1, _blur(y * y, g) - mu2 * mu2, _blur(x * y, g) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    return (2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1) * cs


def _masked_scores(sqr_diff, ssim, mask):
    """per-image PSNR / SSIM over the pixels of a [N,H,W,1] mask (x3 channels), inf -> 0, mean over images"""
    n = len(sqr_diff)
    denom = np.maximum(mask.reshape(n, -1).sum(-1) * 3., 1.)
    with np.errstate(divide='ignore'):
        psnr = -10. * np.log10((sqr_diff * mask).reshape(n, -1).sum(-1) / denom)
    ssim = (ssim * mask).reshape(n, -1).sum(-1) / denom
    psnr[psnr == np.inf] = 0.
    return float(psnr.mean()), float(ssim.mean())


def evaluate_metric(rgbs, gt_imgs, disps=None, gt_masks=None, valid_idxs=None, poses=None, kps=None, hwf=None, centers=None,
                    ext_scale=None, vid_base=None, eval_postfix="", eval_both=False, white_bkgd=False, render_factor=0, **_):
    """Validation scores of `render_testset` (reference evaluation_helpers.py:257-385): whole image when there is neither a
    foreground mask nor a valid-ray mask; foreground only when `gt_masks` is given; with `eval_both` the headline numbers are
    taken inside the cylinder boxes (`valid_idxs`) and the foreground ones are reported next to them.  Score files are appended
    under `vid_base` like the reference (videos are not written: no imageio in this image)."""
    rgbs, gt_imgs = np.asarray(rgbs, dtype=np.float32), np.asarray(gt_imgs, dtype=np.float32)
    H, W = gt_imgs.shape[1:3]
    valid_masks = None
    if eval_both:
        if valid_idxs is None or render_factor != 0:
            from .ray_utils import kp_to_valid_rays
            _, valid_idxs, _, _ = kp_to_valid_rays(poses, *hwf, centers=centers, kps=kps, ext_scale=ext_scale)
        valid_masks = np.zeros((len(valid_idxs), H * W, 1), dtype=np.float32)
        for i, idx in enumerate(valid_idxs):
            valid_masks[i, np.asarray(idx)] = 1
        valid_masks = valid_masks.reshape(-1, H, W, 1)
    if gt_masks is not None:
        keep = np.where(gt_masks.reshape(len(gt_masks), -1).sum(-1) > 0)[0]      # images with a person in them
        rgbs, gt_imgs, gt_masks = rgbs[keep], gt_imgs[keep], gt_masks[keep]
        valid_masks = valid_masks[keep] if valid_masks is not None else None
    th_rgbs = torch.tensor(rgbs).permute(0, 3, 1, 2)
    if render_factor > 0:
        th_rgbs = F.interpolate(th_rgbs, size=gt_imgs.shape[1:3], mode='bilinear', align_corners=False)
        rgbs = th_rgbs.permute(0, 2, 3, 1).numpy()
    ssim = ssim_map(th_rgbs, torch.tensor(gt_imgs).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
    sqr_diff = np.square(gt_imgs - rgbs)
    fg_psnr = fg_ssim = None
    if gt_masks is not None:
        fg_psnr, fg_ssim = _masked_scores(sqr_diff, ssim, gt_masks[..., :1])
    if valid_masks is None and gt_masks is None:
        psnr, ssim_v = _masked_scores(sqr_diff, ssim, np.ones_like(sqr_diff[..., :1]))
    elif valid_masks is not None:
        psnr, ssim_v = _masked_scores(sqr_diff, ssim, valid_masks)
    else:
        psnr, ssim_v = fg_psnr, fg_ssim
    if vid_base is not None:
        scores = [("psnr", psnr), ("ssim", ssim_v)]
        if valid_masks is not None and gt_masks is not None:
            scores += [("psnr_fg", fg_psnr), ("ssim_fg", fg_ssim)]
        for name, v in scores:
            base, _, fg = name.partition("_")
            with open(vid_base + f"{base}{eval_postfix}{'_fg' if fg else ''}.txt", "a") as f:
                f.write(f"{v}\n")
    return {"psnr": psnr, "ssim": ssim_v, "psnr_fg": fg_psnr, "ssim_fg": fg_ssim}


def evaluate_in_boxes(rgbs, accs, bboxes, gt_imgs, gt_masks=None, bg_imgs=None, bg_indices=None):
    """Per-frame scores inside each frame's 2-D cylinder box (reference run_render.py:1178-1263): box PSNR / SSIM and, with
    masks, foreground PSNR / SSIM; frames whose cropped mask is empty are skipped.  -> dict of lists"""
    out = {'psnr': [], 'ssim': [], 'fg_psnr': [], 'fg_ssim': []}
    for i, (rgb, (tl, br), gt) in enumerate(zip(rgbs, bboxes, gt_imgs)):
        gt = np.asarray(gt, dtype=np.float32).reshape(rgb.shape)
        mask = None
        if gt_masks is not None:
            mask = np.asarray(gt_masks[i], dtype=np.float32).reshape(*rgb.shape[:2], -1)[..., :1]
            if bg_imgs is not None:
                gt = gt * mask + (1. - mask) * bg_imgs[bg_indices[i]]
            mask = mask[tl[1]:br[1], tl[0]:br[0]]
            if mask.sum() < 1:
                continue
        g, r = gt[tl[1]:br[1], tl[0]:br[0]], np.asarray(rgb, dtype=np.float32)[tl[1]:br[1], tl[0]:br[0]]
        se = np.square(g - r)
        s = ssim_map(torch.tensor(r[None]).permute(0, 3, 1, 2), torch.tensor(g[None]).permute(0, 3, 1, 2))
        s = s.permute(0, 2, 3, 1).numpy()[0]
        out['psnr'].append(float(-10. * np.log10(se.mean())))
        out['ssim'].append(float(s.mean()))
        if mask is not None:
            denom = mask.sum() * 3.
            out['fg_psnr'].append(float(-10. * np.log10((se * mask).sum() / denom)))
            out['fg_ssim'].append(float((s * mask).sum() / denom))
    return out
