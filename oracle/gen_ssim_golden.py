"""Known-answer vector for core/utils/evaluation_helpers.ssim_map (TEST INFRASTRUCTURE; run in the build container:
    python oracle/gen_ssim_golden.py   ->  tests/golden/ssim_known_answer.npz).

The reference scores SSIM with `SSIM(size_average=False)` of a pinned FORK of pytorch-msssim (requirements.txt:7,
core/utils/evaluation_helpers.py:9,307-321) that is not in this image and cannot be fetched, so the reference's own SSIM output
cannot be generated here.  What CAN be pinned is the algorithm that package publishes (Wang et al. 2004: 11-tap Gaussian window,
sigma 1.5, K1 = 0.01, K2 = 0.03, local moments by separable filtering): this script evaluates it INDEPENDENTLY of the product --
float64, scipy.ndimage.correlate1d, no torch -- on seeded images, and commits inputs + outputs:
  * `map_same`: the per-pixel map with zero padding (the image-sized map the reference multiplies with its H x W masks);
  * `map_valid`: the un-padded ("valid") map of upstream pytorch-msssim, = the interior of `map_same`.
What the reference's CALL SITES (core/utils/evaluation_helpers.py:307-353, run_render.py:1178-1263) fix about the fork, and what this
vector therefore assumes:
  * constructor `SSIM(size_average=False)` and nothing else: every other argument is the fork's default.  Upstream's defaults at the
    fork point are win_size = 11, win_sigma = 1.5, K = (0.01, 0.03), channel = 3 -- assumed unchanged;
  * the result is used as `th_ssim.permute(0, 2, 3, 1)` and multiplied with [N, H, W, 1] masks (:321, :331, :342): it is a 4-D
    per-pixel, per-channel map OF THE IMAGE'S SIZE.  Upstream returns a per-image scalar from an un-padded ("valid") filter, so
    the fork both skips the spatial mean and pads; WHICH padding (zeros / reflect / replicate) the call sites cannot tell.
    `map_same` assumes zeros (F.conv2d's `padding=`), and `map_valid` -- the interior, 5 pixels from every border -- does not
    depend on that choice;
  * the images handed over are float RGB in [0, 1] (:308-315).  Upstream's `data_range` default is 255; with it C1, C2 would be
    6.5 / 58.5 on unit-range images and every score ~1, which the scores the reference's README / paper report (0.8 - 0.9 range)
    rule out: `data_range = 1.0` assumed.
The fork's padding mode and data_range default therefore remain unverified (the repository at the pinned commit is not reachable
from here): SSIM parity against the reference stays "unpinned" in that sense; the arithmetic of the
published algorithm, and the interior of the map whatever the padding, is pinned by this vector."""
import os

import numpy as np
from scipy.ndimage import correlate1d

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ssim_known_answer.npz")


def gauss(size=11, sigma=1.5):
    x = np.arange(size, dtype=np.float64) - size // 2
    g = np.exp(-(x ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def blur_same(x, g):          # x [N,C,H,W]; zero padding
    return correlate1d(correlate1d(x, g, axis=2, mode="constant", cval=0.0), g, axis=3, mode="constant", cval=0.0)


def ssim(x, y, data_range=1.0, K=(0.01, 0.03)):
    g = gauss()
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = blur_same(x, g), blur_same(y, g)
    s1, s2, s12 = blur_same(x * x, g) - mu1 * mu1, blur_same(y * y, g) - mu2 * mu2, blur_same(x * y, g) - mu1 * mu2
    return (2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1) * (2 * s12 + C2) / (s1 + s2 + C2)


def main():
    rng = np.random.default_rng(2024)
    gt = rng.uniform(size=(2, 3, 28, 24))
    gt[1] = np.clip(np.linspace(0, 1, 24)[None, None, :] * np.linspace(0.2, 1, 28)[None, :, None] + 0.05 * rng.normal(size=(3, 28, 24)), 0, 1)
    pred = np.clip(gt + 0.08 * rng.normal(size=gt.shape), 0, 1)
    pred[0, :, :10] = gt[0, :, :10]                      # an error-free region: SSIM = 1 away from its border
    x, y = pred.astype(np.float32), gt.astype(np.float32)
    m = ssim(x.astype(np.float64), y.astype(np.float64))
    np.savez_compressed(OUT, pred=x, gt=y, map_same=m, map_valid=m[:, :, 5:-5, 5:-5], data_range=1.0, win_size=11, win_sigma=1.5,
                        K=np.array([0.01, 0.03]))
    print("wrote", OUT, "mean SSIM", m.mean(), "min", m.min())


if __name__ == "__main__":
    main()
