"""TEST INFRASTRUCTURE ONLY (never imported by the product path).  Pixel-loop restatements of the OpenCV routines the reference's
label synthesis calls (multitasking_utils.py:6-35, preprocess_save_patches_ISPRS.py:206-228): a second, independently written
statement of the same published algorithms, used to cross-check the vectorised `resunet_a_mltsk_keras_amd/labels.py` on
non-trivial shapes.  PARITY UNPINNED against cv2 itself (cv2 is in neither interpreter of this image); what these loops pin is that
the two restatements - written from the algorithm descriptions along different routes - agree everywhere.

canny_loops follows cv::Canny's scalar path (imgproc/src/canny.cpp) as published: 3x3 Sobel with replicated borders, L1
magnitude into a buffer with a zero frame, a pixel is a candidate if mag > low and it is a local maximum along its gradient
direction chosen by the integer tan(22.5 deg) tests (TG22 = 13573 = tan * 2^15): |dy| * 2^15 < |dx| * TG22 -> compare left
(strict) / right (>=); |dy| * 2^15 > |dx| * (TG22 + 2^16) -> compare up (strict) / down (>=); else the diagonal pair picked by
the sign of dx ^ dy (both strict); candidates above `high` seed a stack-based 8-neighbour flood that keeps candidates reached."""
import numpy as np


def sobel_loops(img):
    H, W = img.shape
    dx = np.zeros((H, W), np.int64)
    dy = np.zeros((H, W), np.int64)

    def px(y, x):
        return int(img[min(max(y, 0), H - 1), min(max(x, 0), W - 1)])
    for y in range(H):
        for x in range(W):
            dx[y, x] = (px(y - 1, x + 1) + 2 * px(y, x + 1) + px(y + 1, x + 1)) - (px(y - 1, x - 1) + 2 * px(y, x - 1) + px(y + 1, x - 1))
            dy[y, x] = (px(y + 1, x - 1) + 2 * px(y + 1, x) + px(y + 1, x + 1)) - (px(y - 1, x - 1) + 2 * px(y - 1, x) + px(y - 1, x + 1))
    return dx, dy


def canny_loops(img, low, high):
    if low > high:
        low, high = high, low
    low, high = int(np.floor(low)), int(np.floor(high))
    H, W = img.shape
    dx, dy = sobel_loops(img)
    mag = np.zeros((H + 2, W + 2), np.int64)
    mag[1:-1, 1:-1] = np.abs(dx) + np.abs(dy)
    state = np.zeros((H, W), np.uint8)                    # 0: not an edge, 1: candidate, 2: edge
    stack = []
    TG22 = 13573
    for y in range(H):
        for x in range(W):
            m = mag[y + 1, x + 1]
            if m <= low:
                continue
            xs, ys = int(dx[y, x]), int(dy[y, x])
            ax, ay = abs(xs), abs(ys) << 15
            tg22x = ax * TG22
            if ay < tg22x:
                ok = m > mag[y + 1, x] and m >= mag[y + 1, x + 2]
            else:
                tg67x = tg22x + (ax << 16)
                if ay > tg67x:
                    ok = m > mag[y, x + 1] and m >= mag[y + 2, x + 1]
                else:
                    s = -1 if (xs ^ ys) < 0 else 1
                    ok = m > mag[y, x + 1 - s] and m > mag[y + 2, x + 1 + s]
            if ok:
                if m > high:
                    state[y, x] = 2
                    stack.append((y, x))
                else:
                    state[y, x] = 1
    while stack:
        y, x = stack.pop()
        for yy in (y - 1, y, y + 1):
            for xx in (x - 1, x, x + 1):
                if 0 <= yy < H and 0 <= xx < W and state[yy, xx] == 1:
                    state[yy, xx] = 2
                    stack.append((yy, xx))
    return np.where(state == 2, 255, 0).astype(np.uint8)


def edt_brute(mask):
    """cv2.distanceTransform(mask, DIST_L2, DIST_MASK_PRECISE): distance of every non-zero pixel to the nearest zero pixel."""
    H, W = mask.shape
    zy, zx = np.nonzero(mask == 0)
    out = np.zeros((H, W), np.float64)
    if len(zy) == 0:
        return out
    for y, x in zip(*np.nonzero(mask)):
        out[y, x] = np.sqrt(((zy - y) ** 2 + (zx - x) ** 2).min())
    return out


def hsv_float(rgb):
    """The real-valued RGB -> HSV of OpenCV's documentation (V = max, S = (V - min) / V, H by the sextant formulas, H / 2 for 8 bits),
    before rounding: the 8-bit fixed-point tables of labels.rgb_to_hsv_u8 must land within one unit of it."""
    r, g, b = (rgb[..., k].astype(np.float64) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    diff = v - mn
    s = np.where(v > 0, 255.0 * diff / np.where(v > 0, v, 1), 0.0)
    safe = np.where(diff > 0, diff, 1)
    h = np.where(v == r, 60 * (g - b) / safe, np.where(v == g, 120 + 60 * (b - r) / safe, 240 + 60 * (r - g) / safe))
    h = np.where(diff > 0, h, 0.0)
    h = np.where(h < 0, h + 360, h) / 2
    return np.stack([h, s, v], axis=-1)
