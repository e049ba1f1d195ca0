"""Third-party pins for the label synthesis (/root/reference/multitasking_utils.py:6-35,
preprocess_save_patches_ISPRS.py:223-228).  cv2 is in no interpreter of this image, but the build container's second
interpreter has scikit-image 0.18 + scipy 1.7 - libraries this repository's labels.py does not run on:

    /opt/conda/bin/python3.9 tests/golden/make_labels_thirdparty.py        ->  tests/golden/labels_thirdparty.npz

Written: the INPUTS (binary masks, sparse edge maps, RGB patches) and what the third-party routines make of them -
  * dist_*  : scipy.ndimage.distance_transform_edt (scipy 1.7.1, exact Euclidean transform) + min-max to [0, 1]
              = cv2.distanceTransform(DIST_L2, DIST_MASK_PRECISE) + cv2.normalize(NORM_MINMAX)
  * dil3_*, dil5_* : skimage.morphology.dilation with the 3x3 / 5x5 cross = cv2.dilate(getStructuringElement(MORPH_CROSS, k))
  * hsv_*   : skimage.color.rgb2hsv (real-valued) quantised to OpenCV's 8-bit ranges H = round(180 h) mod 180, S, V = round(255 .)
tests/test_labels.py::test_third_party_pins requires labels.py to match exactly (distance to float32 rounding, dilation
bit for bit) and within one unit (HSV: OpenCV's 12-bit fixed-point tables round differently from the real formula).
"""
import os
import sys

import numpy as np
import scipy
import skimage
from scipy import ndimage
from skimage import color, morphology

HERE = os.path.dirname(os.path.abspath(__file__))


def masks(n=64):
    yy, xx = np.mgrid[:n, :n]
    rng = np.random.default_rng(7)
    out = {
        "ellipse": ((yy - 30) ** 2 / 18.0 ** 2 + (xx - 28) ** 2 / 11.0 ** 2) <= 1.0,
        "band": np.abs(0.6 * xx - yy + 10) <= 5,
        "cut_by_border": ((yy + 6) ** 2 + (xx - 60) ** 2) <= 20 ** 2,
        "checker": ((yy // 8) + (xx // 8)) % 2 == 0,
        "blobs": ndimage.binary_opening(rng.random((n, n)) > 0.45, iterations=2),
        "thin_line": (yy == 20) | (xx == 41),
        "one_pixel": (yy == 33) & (xx == 17),
    }
    return {k: v.astype(np.uint8) for k, v in out.items()}


def main():
    rng = np.random.default_rng(11)
    out = {"versions": np.array([f"python {sys.version.split()[0]}", f"numpy {np.__version__}", f"scipy {scipy.__version__}",
                                 f"scikit-image {skimage.__version__}"])}
    cross3 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    cross5 = np.zeros((5, 5), np.uint8); cross5[2, :] = 1; cross5[:, 2] = 1
    for name, m in masks().items():
        out["mask_" + name] = m
        d = ndimage.distance_transform_edt(m).astype(np.float32)
        lo, hi = float(d.min()), float(d.max())
        out["dist_" + name] = ((d - lo) / (hi - lo)).astype(np.float32) if hi > lo else np.zeros_like(d)
        # an edge-like sparse map of the same shape: the mask's inner contour plus salt noise
        edge = (m & ~ndimage.binary_erosion(m).astype(np.uint8)) | (rng.random(m.shape) > 0.985).astype(np.uint8)
        edge = (edge * 255).astype(np.uint8)
        out["edge_" + name] = edge
        out["dil3_" + name] = morphology.dilation(edge, cross3)
        out["dil5_" + name] = morphology.dilation(edge, cross5)
    rgb = rng.integers(0, 256, size=(48, 48, 3)).astype(np.uint8)
    rgb[:4] = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]], np.uint8)[:, None, :]      # pure colours and grey rows
    rgb[4] = np.arange(48, dtype=np.uint8)[:, None] * 5
    out["rgb"] = rgb
    hsv = color.rgb2hsv(rgb)
    q = np.stack([np.rint(hsv[..., 0] * 180.0) % 180, np.rint(hsv[..., 1] * 255.0), np.rint(hsv[..., 2] * 255.0)], axis=-1)
    out["hsv_u8"] = q.astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "labels_thirdparty.npz"), **out)
    print("wrote labels_thirdparty.npz:", ", ".join(out["versions"]))


if __name__ == "__main__":
    main()
