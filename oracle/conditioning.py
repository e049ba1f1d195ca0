"""Test infrastructure (oracle side only): how far apart are two float32 evaluations of the reference graph's gradient?

The ResUnet-a graph normalises with batch statistics; channels that are (nearly) constant over a batch - dead after a
ReLU - come out of BatchNorm as rounding noise times 1/sqrt(eps) = 31.6, so the sign of those pre-activations, and with it
the ReLU mask of the backward pass, depends on the summation ORDER of the statistics.  This script measures the effect on
the oracle alone: the fp32 oracle, and the fp32 oracle fed the same batch with the samples in reverse order (another
summation order, same mathematics), against the float64 oracle.  Measured in this container (median / max over the
parameter tensors of  max|g - g64| / max|g64|):

    64x64   B=2 depth 4 seed 11 : 1.1e-06 / 1.0e-05     reversed batch 4.5e-04 / 3.0e-03
    64x64   B=2 depth 4 seed 12 : 4.8e-04 / 5.8e-03     reversed batch 4.8e-04 / 5.8e-03
    64x64   B=3 depth 4 seed 23 : 1.5e-03 / 2.2e-02     reversed batch 8.2e-04 / 2.2e-02
    64x64   B=2 depth 6 seed 11 : 2.6e-06 / 7.9e-05     (tests/test_model_gpu.py::test_tiny_multitask_fp32_two_steps)
    128x128 B=2 depth 6 seed 11 : 2.2e-03 / 1.3e-02
    256x256 B=2 depth 6 (cfg3)  : 1.5e-03 / 3.8e-01     reversed batch 1.3e-03 / 3.8e-01

i.e. one and the same fp32 implementation lands anywhere between 1e-6 and 2e-3 (median) of the exact gradient depending on
the order it adds in.  tests/test_model_gpu.py::check_step therefore anchors on float64 and asks the HIP gradient to be
within 2x the worse of the two oracle orders plus twice that intrinsic spread (5e-3), never a tensor off by 0.5 of its scale.
Usage: python oracle/conditioning.py
"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import resuneta_ref as ref
from resunet_a_mltsk_keras_amd.synthetic import make_batch

def grads(cfg, params, order, x, y, dt, lw):
    p = {k: v.detach().clone().to(dt) for k, v in params.items()}
    tr = ref.RefTrainer(cfg, p, order, ref.CompileSpec(loss="tanimoto", loss_weights=lw, lr=1e-3))
    npdt = np.float64 if dt == torch.float64 else np.float32
    tr.train_on_batch(x.astype(npdt), {k: v.astype(npdt) for k, v in y.items()})
    return {k: tr.last_grads[k].double().numpy() for k in order}

cases = [(64, 3, 4, 2, 4, 11, 16), (64, 3, 4, 2, 4, 12, 16), (64, 3, 4, 3, 4, 23, 16), (256, 6, 6, 2, 6, 1234, 32)]
for (size, ch, C, B, depth, seed, block) in cases:
    t0 = time.time()
    lw = {"seg": 1.0, "bound": 0.7, "dist": 1.3, "color": 0.5}
    cfg = ref.RefConfig(input_shape=(size, size, ch), num_classes=C, multitasking=True, width=32, depth=depth)
    params, order = ref.init_params(cfg, 3)
    x, y = make_batch(B, size, ch, C, True, seed=seed, block=block)
    e = grads(cfg, params, order, x, y, torch.float64, lw)
    a = grads(cfg, params, order, x, y, torch.float32, lw)
    xr = x[::-1].copy(); yr = {k: v[::-1].copy() for k, v in y.items()}
    b = grads(cfg, params, order, xr, yr, torch.float32, lw)       # same batch, samples in reverse order: other summation order
    gmax = max(np.abs(v).max() for v in e.values())
    ra, rb = [], []
    for k in order:
        if np.abs(e[k]).max() < 1e-5 * gmax: continue
        s = np.abs(e[k]).max()
        ra.append(np.abs(a[k] - e[k]).max() / s); rb.append(np.abs(b[k] - e[k]).max() / s)
    print(f"size {size} B {B} depth {depth} seed {seed}: fp32 vs f64 median/max {np.median(ra):.2e}/{np.max(ra):.2e}; reversed-batch fp32 {np.median(rb):.2e}/{np.max(rb):.2e}  ({time.time()-t0:.0f}s)", flush=True)
