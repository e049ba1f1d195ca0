"""Seeded synthetic patches in the reference's on-disk format (the dataset layout is
preprocess_save_patches_ISPRS.py:178-228: float32 NHWC, labels one-hot / [0,1])."""
import numpy as np


def make_batch(batch, patch, channels, num_classes, multitasking=True, seed=1234, block=32):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 1.0, (batch, patch, patch, channels)).astype(np.float32)
    nb = max(patch // block, 1)
    ids = rng.integers(0, num_classes, size=(batch, nb, nb))
    k = min(num_classes, nb)
    ids[:, 0, :k] = np.arange(k)                       # every class present when the grid allows
    cls = np.repeat(np.repeat(ids, patch // nb, axis=1), patch // nb, axis=2)
    seg = np.eye(num_classes, dtype=np.float32)[cls]
    if not multitasking:
        return x, seg
    edge = np.zeros((batch, patch, patch), bool)
    edge[:, 1:, :] |= cls[:, 1:, :] != cls[:, :-1, :]
    edge[:, :-1, :] |= cls[:, 1:, :] != cls[:, :-1, :]
    edge[:, :, 1:] |= cls[:, :, 1:] != cls[:, :, :-1]
    edge[:, :, :-1] |= cls[:, :, 1:] != cls[:, :, :-1]
    bound = seg * edge[..., None].astype(np.float32)
    dist = (seg * rng.uniform(0.0, 1.0, (batch, patch, patch, 1))).astype(np.float32)
    color = rng.uniform(0.0, 1.0, (batch, patch, patch, 3)).astype(np.float32)
    return x, {"seg": seg, "bound": bound, "dist": dist, "color": color}
