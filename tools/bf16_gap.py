"""Where does the bf16-vs-fp32 gap of the 200-step training test come from?  

tests/test_model_gpu.py::test_cfg3_bf16_training_tracks_fp32_over_200_steps_on_varying_batches measures the bf16 run 4.6 % above the fp32
run at the end.  This tool repeats that run (cfg3, batch 8, Adam 1e-3, eight rotating batches, same initial weights) in variants that
separate the candidates:

  f32            fp32 storage (the yardstick)
  f32_eps        fp32 storage, initial weights multiplied by (1 + 1e-6 * N(0,1))        -> how far two fp32 runs drift apart by themselves
  f32_w0bf16     fp32 storage, INITIAL weights rounded to bf16 once                      -> a one-off perturbation of bf16 size
  f32_wbf16      fp32 activations, the convolutions' weight copies rounded to bf16 every step (eager step, hook behind the weight refresh)
  bf16           bf16 activations + bf16 weight copies (the benchmarked path)
  bf16_nofuse    the same with RUA_FUSE_BN=0 (BatchNorm materialised instead of applied on load: other rounding points)

Prints, per variant, the mean total loss over the last 20 steps, its relative gap to f32, and the largest per-step gap.
    python tools/bf16_gap.py [steps]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(dtype, weights, batches, steps, hook_round=False, env=None):
    from test_model_gpu import hip_engine
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        e = hip_engine((256, 256, 6), 6, True, "tanimoto", dtype, weights)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if hook_round:
        e.use_graph = False
        orig = e._prep_weights

        def prep(s):
            dirty = e.weights_dirty
            orig(s)
            if dirty:
                e.Wf.copy_(e.Wf.bfloat16().float())
                e.Wd.copy_(e.Wd.bfloat16().float())
        e._prep_weights = prep
    tr = np.array([e.train_step(*batches[i % len(batches)])[0] for i in range(steps)])
    del e
    torch.cuda.empty_cache()
    return tr


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    from test_model_gpu import golden_step
    from resunet_a_mltsk_keras_amd.synthetic import make_batch
    batches = [make_batch(8, 256, 6, 6, True, seed=9000 + i) for i in range(8)]
    w = golden_step("cfg3")["params"]
    rng = np.random.default_rng(7)
    w_eps = {k: (v * (1.0 + 1e-6 * rng.standard_normal(v.shape))).astype(np.float32) for k, v in w.items()}
    w_bf = {k: torch.from_numpy(v).bfloat16().float().numpy() if k.endswith("/kernel") else v for k, v in w.items()}
    runs = {}
    runs["f32"] = run("f32", w, batches, steps)
    runs["f32_eps"] = run("f32", w_eps, batches, steps)
    runs["f32_w0bf16"] = run("f32", w_bf, batches, steps)
    runs["f32_wbf16"] = run("f32", w, batches, steps, hook_round=True)
    runs["bf16"] = run("bf16", w, batches, steps)
    runs["bf16_nofuse"] = run("bf16", w, batches, steps, env={"RUA_FUSE_BN": "0"})
    a = runs["f32"]
    print("variant        first    last20   gap_end  worst_step_gap  (gap = relative to f32)")
    for k, b in runs.items():
        ge = (b[-20:].mean() - a[-20:].mean()) / a[-20:].mean()
        gs = float(np.max(np.abs(b - a) / a))
        print("%-13s %7.4f  %7.4f  %+8.2e  %8.2e" % (k, b[0], b[-20:].mean(), ge, gs), flush=True)
    for k in ("f32", "bf16"):
        print(k, "every 20th step:", np.round(runs[k][::20], 4))


if __name__ == "__main__":
    main()
