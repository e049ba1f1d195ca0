"""Multitask label synthesis without OpenCV: the boundary, distance and colour targets the reference
derives from a one-hot segmentation patch and the RGB patch (multitasking_utils.py:6-35,
preprocess_save_patches_ISPRS.py:206-228).

The arithmetic lives in OpenCV (un-vendored, un-pinned; cv2 is not installed here), so these are restatements of
OpenCV's published algorithms - `cv2.Canny` (3x3 Sobel, L1 magnitude, 4-direction non-maximum suppression with the
integer tan(22.5 deg) test, hysteresis), `cv2.dilate` with a 3x3 cross, `cv2.distanceTransform(DIST_L2, DIST_MASK_PRECISE)`
+ `cv2.normalize(NORM_MINMAX)`, and the 8-bit `COLOR_RGB2HSV` fixed-point tables.  PARITY UNPINNED: there is no cv2 in
this image to check them against; `tests/test_labels.py` pins them to hand-derived answers only.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def _sobel3(img: np.ndarray):
    """3x3 Sobel dx, dy (int32) with replicated borders, as cv2.Sobel(..., ksize=3, BORDER_REPLICATE) inside Canny."""
    p = np.pad(img.astype(np.int32), 1, mode="edge")
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return dx, dy


def canny_u8(img: np.ndarray, low: float, high: float) -> np.ndarray:
    """cv2.Canny(img_u8, low, high) with apertureSize 3 and L2gradient False -> uint8 {0, 255}."""
    low, high = (high, low) if low > high else (low, high)
    low, high = int(np.floor(low)), int(np.floor(high))
    dx, dy = _sobel3(img)
    mag = np.abs(dx) + np.abs(dy)
    H, W = mag.shape
    mp = np.pad(mag, 1, mode="constant")                       # the magnitude buffer is zero outside the image
    m = mp[1:-1, 1:-1]
    ax, ay = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22x = ax * 13573                                         # tan(22.5 deg) * 2^15
    tg67x = tg22x + (ax << 16)
    left, right = mp[1:-1, :-2], mp[1:-1, 2:]
    up, down = mp[:-2, 1:-1], mp[2:, 1:-1]
    s_neg = (dx ^ dy) < 0                                      # gradient along the anti-diagonal
    d_prev = np.where(s_neg, mp[:-2, 2:], mp[:-2, :-2])        # mag[row-1][j-s], s = -1 if (dx ^ dy) < 0 else 1
    d_next = np.where(s_neg, mp[2:, :-2], mp[2:, 2:])          # mag[row+1][j+s]
    horiz = ay < tg22x
    vert = ay > tg67x
    is_max = np.where(horiz, (m > left) & (m >= right), np.where(vert, (m > up) & (m >= down), (m > d_prev) & (m > d_next)))
    cand = (m > low) & is_max
    strong = cand & (m > high)
    # hysteresis: candidates 8-connected to a strong pixel
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), bool))
    keep = np.zeros(n + 1, bool)
    keep[np.unique(lab[strong])] = True
    keep[0] = False
    return np.where(keep[lab], 255, 0).astype(np.uint8)


def dilate_cross(img: np.ndarray, ksize=(3, 3)) -> np.ndarray:
    """cv2.dilate(img, getStructuringElement(MORPH_CROSS, ksize), iterations=1): ksize = (width, height); the element is the
    centre row plus the centre column (anchor at the centre), the outside of the image is ignored (max over the inside)."""
    kw, kh = int(ksize[0]), int(ksize[1])
    ax, ay = kw // 2, kh // 2
    H, W = img.shape
    p = np.pad(img, ((ay, kh - 1 - ay), (ax, kw - 1 - ax)), mode="constant", constant_values=0)
    views = [p[ay:ay + H, j:j + W] for j in range(kw)] + [p[i:i + H, ax:ax + W] for i in range(kh)]
    return np.maximum.reduce(views)


def dilate_cross3(img: np.ndarray) -> np.ndarray:
    return dilate_cross(img, (3, 3))


def get_boundary_label(label: np.ndarray, kernel_size=(3, 3)) -> np.ndarray:
    """multitasking_utils.py:6-23: per class channel Canny(0, 1) of the {0,1} mask, dilated by a `kernel_size` cross (the
    reference's callers use the default 3x3), scaled to [0, 1]."""
    tl = label.astype(np.uint8)
    out = np.empty(label.shape, np.float32)
    for c in range(label.shape[2]):
        out[:, :, c] = dilate_cross(canny_u8(tl[:, :, c], 0, 1), kernel_size).astype(np.float32) / 255.0
    return out


def get_distance_label(label: np.ndarray) -> np.ndarray:
    """multitasking_utils.py:26-35: exact Euclidean distance of every class pixel to the nearest non-class pixel, min-max
    normalised to [0, 1] per channel (cv2.normalize maps a constant image - a class that is absent or fills the patch - to 0)."""
    out = np.empty(label.shape, np.float32)
    for c in range(label.shape[2]):
        patch = label[:, :, c].astype(np.uint8)
        if patch.all() or not patch.any():
            out[:, :, c] = 0.0
            continue
        dist = ndimage.distance_transform_edt(patch).astype(np.float32)
        lo, hi = float(dist.min()), float(dist.max())
        out[:, :, c] = (dist - lo) / (hi - lo) if hi > lo else 0.0
    return out


_SDIV = np.zeros(256, np.int64)
_HDIV = np.zeros(256, np.int64)
_i = np.arange(1, 256)
_SDIV[1:] = np.rint((255 << 12) / (1.0 * _i)).astype(np.int64)
_HDIV[1:] = np.rint((180 << 12) / (6.0 * _i)).astype(np.int64)


def rgb_to_hsv_u8(rgb: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(uint8 RGB, COLOR_RGB2HSV): H in [0, 179], S and V in [0, 255] (OpenCV's 12-bit fixed-point tables)."""
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    diff = v - np.minimum(np.minimum(r, g), b)
    vr, vg = v == r, v == g
    s = (diff * _SDIV[v] + (1 << 11)) >> 12
    h = np.where(vr, g - b, np.where(vg, (b - r) + 2 * diff, (r - g) + 4 * diff))
    h = (h * _HDIV[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


def color_label(rgb_u8: np.ndarray, norm_type: int = 1) -> np.ndarray:
    """preprocess_save_patches_ISPRS.py:223-228 + normalize_hsv (:89-97, including the norm_type 2 precedence quirk)."""
    hsv = rgb_to_hsv_u8(rgb_u8).astype(np.float32)
    if norm_type == 1:
        hsv /= np.array([179.0, 255.0, 255.0], np.float32)
    elif norm_type == 2:
        hsv /= np.array([89.5 - 1.0, 127.5 - 1.0, 127.5 - 1.0], np.float32)
    else:
        raise NotImplementedError("norm_type 3 (StandardScaler) is unfinished in the reference as well")
    return hsv


def multitask_labels(seg_onehot: np.ndarray, rgb_u8: np.ndarray, norm_type: int = 1):
    """The four float32 targets the reference writes for one patch (labels/{seg,bound,dist,color})."""
    seg = seg_onehot.astype(np.float32)
    return {"seg": seg, "bound": get_boundary_label(seg), "dist": get_distance_label(seg), "color": color_label(rgb_u8, norm_type)}
