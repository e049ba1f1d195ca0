"""Debug aid: cfg5 (128 x 128 x 7, 2 classes, single task, batch 32) forward in bf16 against fp32 storage with the same weights; where the
logits differ most, and which tuning keys change that.  usage: python tools/cfg5_debug.py [key=value ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402
from resunet_a_mltsk_keras_amd.engine import Engine, LossSpec, ModelConfig  # noqa: E402
from resunet_a_mltsk_keras_amd.synthetic import make_batch  # noqa: E402


def logits(dtype, x, y, tune):
    lib = L.lib()
    eng = Engine(ModelConfig(input_shape=(128, 128, 7), num_classes=2, multitasking=False, depth=6), dtype=dtype, seed=0, split_k=True)
    eng.compile(LossSpec(kind={"seg": L.LOSS_TANIMOTO}, weight={"seg": 1.0}, optimizer="adam", lr=1e-3))
    if tune:
        lib.set_tuning(**tune)
    eng.forward_backward(x, y)
    torch.cuda.synchronize()
    z = eng.logits(True, x.shape[0])["seg"].copy()
    del eng
    torch.cuda.empty_cache()
    return z


def main():
    torch.cuda.set_device(0)
    x, y = make_batch(32, 128, 7, 2, False, seed=555)
    ref = logits("f32", x, y, {})
    variants = [{}] + [dict([kv.split("=")[0], int(kv.split("=")[1])] for kv in a.split(",")) for a in sys.argv[1:]]
    for tune in variants:
        z = logits("bf16", x, y, tune)
        d = np.abs(z.astype(np.float64) - ref)
        idx = np.unravel_index(np.argmax(d), d.shape)
        bad = (d.max(axis=-1) > 0.2 * np.abs(ref).max())
        print(tune, "max |bf16 - f32| / max |f32| = %.4f at %s; pixels off by > 20 %%: %d of %d; images touched: %s" %
              (d.max() / np.abs(ref).max(), idx, int(bad.sum()), bad.size, sorted(set(np.nonzero(bad)[0].tolist()))[:12]), flush=True)
        if bad.sum():
            n = idx[0]
            rows = np.nonzero(bad[n].any(axis=1))[0]; cols = np.nonzero(bad[n].any(axis=0))[0]
            print("   image %d: rows %d..%d, cols %d..%d" % (n, rows.min(), rows.max(), cols.min(), cols.max()), flush=True)
        L.lib().set_tuning(**{k: dflt(k) for k in tune})


_d = {}


def dflt(k):
    return _d[k]


if __name__ == "__main__":
    for a in sys.argv[1:]:
        for kv in a.split(","):
            _d[kv.split("=")[0]] = L.lib().get_tuning(kv.split("=")[0])
    main()
