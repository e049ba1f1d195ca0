"""Generates tests/golden/*.npz from the CPU oracle (run in the dev container; the files are committed).

The reference cannot run here (TensorFlow is not installed) and has no fixtures of its own, so these
goldens pin the ORACLE (against regressions) and give the GPU tests oracle-free expected values:
  tiny_multitask.npz : reference-width (32) multitask model on 64x64x6 patches (the smallest size the d6 graph accepts), batch 2, Tanimoto-dual on all
                       heads: seeded inputs, total/per-head losses and logits of one training-mode forward,
                       the five largest-magnitude parameter gradients' checksums, eval-mode predictions.
  tanimoto_kat.npz   : hand-computed Tanimoto-dual cases (the KATs of tests/test_oracle_kat.py as data).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import naive_ops as nv  # noqa: E402
from oracle import resuneta_ref as ref  # noqa: E402
from resunet_a_mltsk_keras_amd.synthetic import make_batch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    cfg = ref.RefConfig(input_shape=(64, 64, 6), num_classes=6, multitasking=True)
    params, order = ref.init_params(cfg, 7)
    lw = {"seg": 1.0, "bound": 0.5, "dist": 2.0, "color": 1.5}
    tr = ref.RefTrainer(cfg, {k: v.clone() for k, v in params.items()}, order, ref.CompileSpec(loss="tanimoto", loss_weights=lw, lr=1e-3))
    x, y = make_batch(2, 64, 6, 6, True, seed=99, block=16)
    pred_eval = ref.forward(cfg, params, x, training=False)
    res = tr.train_on_batch(x, y)
    gsum = {k: float(np.abs(tr.last_grads[k].numpy()).sum()) for k in order if k.endswith("kernel")}
    top = sorted(gsum, key=gsum.get)[-5:]
    np.savez_compressed(
        os.path.join(OUT, "tiny_multitask.npz"), seed=7, x=x, **{"y_" + k: v for k, v in y.items()},
        losses=np.array(res[:5]), metrics=np.array(res[5:]), **{"logits_" + k[:-7]: v for k, v in tr.last_taps.items()},
        **{"pred_eval_" + k: v for k, v in pred_eval.items()}, grad_names=np.array(top), grad_abs_sums=np.array([gsum[k] for k in top]),
        loss_weights=np.array([lw[h] for h in ref.HEADS]))
    # Tanimoto KATs as data
    yk = np.zeros((1, 2, 2, 2), np.float32); yk[0, :, :, 0] = [[1, 1], [1, 0]]; yk[0, :, :, 1] = [[0, 0], [0, 1]]
    pk = np.zeros_like(yk); pk[0, :, :, 0] = [[1, 1], [0, 0]]; pk[0, :, :, 1] = [[0, 0], [1, 1]]
    rng = np.random.default_rng(5)
    yr = np.eye(4, dtype=np.float32)[rng.integers(0, 3, (3, 6, 5))]          # class 3 absent: inf-weight path
    pr = rng.uniform(0.02, 0.98, yr.shape).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "tanimoto_kat.npz"), y_swap=yk, p_swap=pk, l_swap=nv.tanimoto_dual_loss(yk, pk),
                        y_rand=yr, p_rand=pr, l_rand=nv.tanimoto_dual_loss(yr, pr))
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
