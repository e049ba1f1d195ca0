"""Full-size golden fixtures: tests/golden/fullsize_<cfg>.npz (run in the dev container; the files are committed).

TEST INFRASTRUCTURE.  One training-mode forward of the CPU oracle (oracle/resuneta_ref.py, the restatement of
ResUnet_a/model2.py:14-193 + multitasking_utils.py:38-85 + utils.py:466-491) per BASELINE configuration at the configuration's
own size and batch: the losses train_on_batch reports and, per head, a strided sample of the pre-activation logits with their
scale and two checksums.  The GPU tests compare the HIP path with these instead of running the oracle on the GPU box's host
cores (20-60 s per configuration, most of the GPU suite's time); tests/test_oracle_kat.py re-derives one of them live in the CPU
suite, tests/test_model_gpu.py keeps one live full-size oracle step.  Parity stays unpinned by the reference (it holds no
fixtures): these files pin the oracle, not the reference.

    python oracle/make_golden_fullsize.py [cfg1 cfg2 ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import resuneta_ref as ref  # noqa: E402
from resunet_a_mltsk_keras_amd.synthetic import make_batch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
NSAMPLE = 20000

# name: (input shape, classes, multitask, loss, batch, data seed, class weights, depth)   - the tests' configurations (tests/test_model_gpu.py)
CONFIGS = {
    "cfg1": ((256, 256, 3), 6, False, "weighted_cross_entropy", 4, 1234, [1.0] * 6, 6),
    "cfg2": ((256, 256, 6), 6, False, "tanimoto", 8, 4321, None, 6),       # BASELINE's batch 8 (rounds 1 - 4: batch 2)
    "cfg3": ((256, 256, 6), 6, True, "tanimoto", 8, 1234, None, 6),
    "cfg4": ((512, 512, 6), 6, True, "tanimoto", 4, 777, None, 7),
    "cfg5": ((128, 128, 7), 2, False, "tanimoto", 32, 555, None, 6),
}
PARAM_SEED = 3


def sample_stride(n):
    return max(1, n // NSAMPLE) | 1                      # odd: walks through every channel position


def oracle_forward(name):
    shape, C, mt, loss, B, seed, cw, depth = CONFIGS[name]
    lw = {"seg": 1.0, "bound": 1.0, "dist": 1.0, "color": 1.0}
    rcfg = ref.RefConfig(input_shape=shape, num_classes=C, multitasking=mt, depth=depth)
    params, order = ref.init_params(rcfg, PARAM_SEED)
    tr = ref.RefTrainer(rcfg, params, order, ref.CompileSpec(loss=loss, class_weights=cw, loss_weights=lw, optimizer="adam", lr=1e-3))
    x, y = make_batch(B, shape[0], shape[2], C, mt, seed=seed)
    exp = tr.losses_on_batch(x, y)
    taps = {k: v for k, v in tr.last_taps.items() if k.endswith("logits")}
    return exp, taps


def digest(exp, taps):
    out = {"losses": np.asarray(exp, np.float64)}
    for k, v in taps.items():
        flat = np.ascontiguousarray(v, np.float32).ravel()
        st = sample_stride(flat.size)
        out[k + "_n"] = np.int64(flat.size)
        out[k + "_stride"] = np.int64(st)
        out[k + "_sample"] = flat[::st].copy()
        out[k + "_maxabs"] = np.float64(np.abs(flat).max())
        out[k + "_sum"] = np.float64(flat.astype(np.float64).sum())
        out[k + "_sumabs"] = np.float64(np.abs(flat.astype(np.float64)).sum())
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in (sys.argv[1:] or list(CONFIGS)):
        t0 = time.time()
        exp, taps = oracle_forward(name)
        np.savez_compressed(os.path.join(OUT, "fullsize_%s.npz" % name), **digest(exp, taps))
        print(name, "losses", np.round(exp[:5], 6), "heads", sorted(taps), "%.1f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
