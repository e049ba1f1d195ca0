"""Second, independent CPU restatement: plain numpy loops over the Keras layer definitions.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Written without torch so that it shares
no code with oracle/resuneta_ref.py; tests require the two to agree.  All tensors are
float NHWC like the reference (train_ISPRS.py:73-92); arithmetic is float64 inside.
"""
from __future__ import annotations

import numpy as np


def conv2d_nhwc(x, kernel_hwio, bias=None, stride=1, dilation=1, padding="valid"):
    """KL.Conv2D (model2.py:19-24,37,101-111).  'same' => zero pad dilation*(k//2) per side."""
    x = np.asarray(x, np.float64)
    k = np.asarray(kernel_hwio, np.float64)
    B, H, W, Cin = x.shape
    kh, kw, _, Cout = k.shape
    ph = dilation * (kh // 2) if padding == "same" else 0
    pw = dilation * (kw // 2) if padding == "same" else 0
    xp = np.zeros((B, H + 2 * ph, W + 2 * pw, Cin))
    xp[:, ph:ph + H, pw:pw + W] = x
    Ho = (H + 2 * ph - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pw - dilation * (kw - 1) - 1) // stride + 1
    y = np.zeros((B, Ho, Wo, Cout))
    for i in range(kh):
        for j in range(kw):
            win = xp[:, i * dilation: i * dilation + (Ho - 1) * stride + 1: stride,
                     j * dilation: j * dilation + (Wo - 1) * stride + 1: stride, :]
            y += win @ k[i, j]
    if bias is not None:
        y += np.asarray(bias, np.float64)
    return y


def batchnorm_train(x, gamma, beta, eps=1e-3):
    """KL.BatchNormalization in training mode: batch mean / biased variance over N,H,W."""
    x = np.asarray(x, np.float64)
    mean = x.mean(axis=(0, 1, 2))
    var = x.var(axis=(0, 1, 2))
    return (x - mean) / np.sqrt(var + eps) * gamma + beta, mean, var


def batchnorm_infer(x, gamma, beta, moving_mean, moving_var, eps=1e-3):
    return (np.asarray(x, np.float64) - moving_mean) / np.sqrt(moving_var + eps) * gamma + beta


def maxpool(x, k):
    """KL.MaxPooling2D(pool_size=k, strides=k), valid (model2.py:47-52)."""
    B, H, W, C = x.shape
    Ho, Wo = H // k, W // k
    y = np.empty((B, Ho, Wo, C), x.dtype)
    for i in range(Ho):
        for j in range(Wo):
            y[:, i, j] = x[:, i * k:(i + 1) * k, j * k:(j + 1) * k].reshape(B, k * k, C).max(axis=1)
    return y


def upsample_nearest(x, k):
    """KL.UpSampling2D(size=k), nearest (model2.py:55-60,91)."""
    return np.repeat(np.repeat(x, k, axis=1), k, axis=2)


def softmax(z):
    z = np.asarray(z, np.float64)
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-np.asarray(z, np.float64)))


def tanimoto_loss(label, pred):
    """multitasking_utils.py:38-68, scalar loops, NHWC."""
    label = np.asarray(label, np.float64)
    pred = np.asarray(pred, np.float64)
    B, H, W, C = label.shape
    vli = np.zeros(C)
    for c in range(C):
        vli[c] = sum(label[n, :, :, c].sum() for n in range(B)) / B
    with np.errstate(divide="ignore"):
        wli = 1.0 / (vli * vli)
    finite = [w for w in wli if not np.isinf(w)]
    mx = max(finite) if finite else 0.0
    wli = np.array([mx if np.isinf(w) else w for w in wli])
    out = np.zeros(B)
    for n in range(B):
        num = den = 0.0
        for c in range(C):
            sp = (pred[n, :, :, c] * label[n, :, :, c]).sum()
            ss = (pred[n, :, :, c] ** 2 + label[n, :, :, c] ** 2).sum()
            num += wli[c] * sp
            den += wli[c] * (ss - sp)
        out[n] = (num + 1e-5) / (den + 1e-5)
    return out


def tanimoto_dual_loss(label, pred):
    """multitasking_utils.py:71-85 (swapped arguments in the first call)."""
    l1 = tanimoto_loss(pred, label)
    l2 = tanimoto_loss(1.0 - np.asarray(label, np.float64), 1.0 - np.asarray(pred, np.float64))
    return 1.0 - 0.5 * (l1 + l2)


def weighted_cce(weights, y_true, y_pred, eps=1e-7):
    """utils.py:481-490 — returns per-pixel loss (B,H,W)."""
    p = np.asarray(y_pred, np.float64)
    p = p / p.sum(axis=-1, keepdims=True)
    p = np.clip(p, eps, 1 - eps)
    return -(np.asarray(y_true, np.float64) * np.log(p) * np.asarray(weights, np.float64)).sum(axis=-1)


def adam_step(theta, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    """Keras Adam (train_ISPRS.py:405): epsilon outside the bias-corrected sqrt."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v


def sgd_step(theta, g, vel, lr, momentum=0.8):
    """Keras SGD(momentum) (train_ISPRS.py:407), non-Nesterov."""
    vel = momentum * vel - lr * g
    return theta + vel, vel
