"""Whole-path parity on the GPU: the recorded HIP plan (through librua_hip.so) against the CPU oracle on
the same seeded inputs and the same weights.  fp32 storage: loss and per-head logits within 1e-3
relative (the north-star tolerance), parameter gradients within 2e-3 of each tensor's scale."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import resuneta_ref as ref  # noqa: E402
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402
from resunet_a_mltsk_keras_amd.engine import Engine, LossSpec, ModelConfig  # noqa: E402
from resunet_a_mltsk_keras_amd.synthetic import make_batch  # noqa: E402

KIND = {"tanimoto": L.LOSS_TANIMOTO}


def make_pair(shape, C, mt, width, loss, optimizer="adam", dtype="f32", seed=3, cw=None, lw=None, split_k=False,
              variant="model2", depth=6):
    """split_k=False by default: on 64x64 inputs the bottleneck BatchNorms see 2-8 samples per channel and amplify the
    atomic-order noise of split-K convolutions ~1e4x, which would drown the 1e-3 comparisons (kernels themselves are
    checked with split-K in test_kernels_gpu.py; the full-size and bf16 tests below run with it)."""
    rcfg = ref.RefConfig(input_shape=shape, num_classes=C, multitasking=mt, width=width, variant=variant, depth=depth)
    params, order = ref.init_params(rcfg, seed)
    lw = lw or {"seg": 1.0, "bound": 0.7, "dist": 1.3, "color": 0.5}
    rspec = ref.CompileSpec(loss=loss, class_weights=cw, loss_weights=lw, optimizer=optimizer, lr=1e-3)
    trainer = ref.RefTrainer(rcfg, {k: v.clone() for k, v in params.items()}, order, rspec)
    eng = Engine(ModelConfig(input_shape=shape, num_classes=C, multitasking=mt, width=width, variant=variant, depth=depth),
                 dtype=dtype, seed=0, split_k=split_k)
    if loss == "tanimoto":
        kind = {h: L.LOSS_TANIMOTO for h in ref.HEADS}
    elif loss == "weighted_cross_entropy":
        kind = {"seg": L.LOSS_WCE, "bound": L.LOSS_BCE_LOGITS, "dist": L.LOSS_MSE, "color": L.LOSS_MSE}
    else:
        kind = {"seg": L.LOSS_CE_LOGITS, "bound": L.LOSS_BCE_LOGITS, "dist": L.LOSS_MSE, "color": L.LOSS_MSE}
    eng.compile(LossSpec(kind=kind, weight=lw, class_weights=cw, optimizer=optimizer, lr=1e-3))
    eng.set_weights({k: v.numpy() for k, v in params.items()})
    return trainer, eng


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def exact_grads(trainer, x, y, second_order=False):
    """The same oracle step in float64 (call BEFORE the fp32 oracle step, it clones the current weights).  second_order: also - key
    "_spread" - the gradient of a second float32 evaluation ORDER of the oracle (same batch, samples reversed): the fp32
    oracle's own distance to the exact gradient depends on the order it sums in (oracle/conditioning.py).  That third oracle
    step (8 - 20 s of host time on a 42 M-parameter graph) is taken where the bound is meant to be tight - the full-size gradient
    test, the Tanimoto / Adam step of the small graph; elsewhere check_step uses the documented upper range of that spread."""
    f64 = lambda a: a.astype(np.float64)
    rev = lambda a: np.ascontiguousarray(a[::-1])
    t64 = ref.RefTrainer(trainer.cfg, {k: v.detach().double() for k, v in trainer.params.items()}, trainer.order, trainer.spec)
    t64.train_on_batch(f64(x), {k: f64(v) for k, v in y.items()} if isinstance(y, dict) else f64(y))
    out = {k: t64.last_grads[k].numpy() for k in trainer.order}
    if second_order:
        t32 = ref.RefTrainer(trainer.cfg, {k: v.detach().clone() for k, v in trainer.params.items()}, trainer.order, trainer.spec)
        t32.train_on_batch(rev(x), {k: rev(v) for k, v in y.items()} if isinstance(y, dict) else rev(y))
        out["_spread"] = {k: t32.last_grads[k].numpy() for k in trainer.order}
    return out


def check_step(trainer, eng, x, y, mt, tol_loss, tol_logit, tol_grad, tol_w, check_grads=True, exact=None):
    exp = trainer.train_on_batch(x, y)
    B = x.shape[0]
    g = eng.forward_backward(x, y)
    torch.cuda.synchronize()
    got = eng._results(g)
    # loss values (total and per head) and metrics
    n_loss = 5 if mt else 1
    for i in range(n_loss):
        assert abs(got[i] - exp[i]) <= tol_loss * max(1.0, abs(exp[i])), (i, got[i], exp[i])
    assert abs(got[n_loss] - exp[n_loss]) < 5e-3                      # accuracy
    tot = sum(exp[n_loss + 1:])
    assert sum(got[n_loss + 1:]) == tot                                 # TP+FP+TN+FN = all elements
    for a, b in zip(got[n_loss + 1:], exp[n_loss + 1:]):
        assert abs(a - b) <= 2e-3 * tot
    # logits
    lg = eng.logits(True, B)
    for h, z in lg.items():
        key = (h + "_logits") if mt else "logits"
        assert rel(z, trainer.last_taps[key]) < tol_logit, (h, rel(z, trainer.last_taps[key]))
    # gradients, tensor by tensor.  Biases (and BN betas) that feed a 1x1 conv + training-mode BN, or a BN
    # directly, have an exactly-zero true gradient: both sides then hold ~1e-10 rounding noise.
    if check_grads and exact is not None:
        # Anchor: the same step evaluated in float64.  Two fp32 evaluations of this graph differ from each other by more
        # than rounding wherever a ReLU pre-activation sits within ~1e-7 of zero (batch-normalised values; the two sides
        # sum the statistics in different orders) and, in the model.py graph (no identity path: gradients pass through
        # chains of cancelling BatchNorm backward passes), by conditioning: the fp32 ORACLE itself is then ~1e-2 away
        # from the exact gradient.  Measured on the oracle alone (oracle/conditioning.py): the SAME fp32 implementation is
        # between 1e-6 and 2e-3 (median over the tensors) of the float64 gradient depending on the order it sums in, at
        # every problem size including full-size cfg3.  Bar for every graph variant: the HIP gradient is as close to the
        # float64 gradient as fp32 evaluations of the oracle are - median within 2x the worse of two oracle summation
        # orders + 5e-3 (twice that intrinsic spread), the 90th percentile within 10x that, no tensor off by 0.5 of its scale
        # (a wrong kernel is O(1) on whole tensors).
        grads = eng.grads_keras()
        gmax = max(float(np.abs(exact[k]).max()) for k in trainer.order)
        r_hip, r_ref, r_alt = [], [], []
        for k in trainer.order:
            e = exact[k]
            if np.abs(e).max() < 1e-5 * gmax:
                assert np.abs(grads[k]).max() < 1e-4 * gmax, (k, float(np.abs(grads[k]).max()))
                continue
            r_hip.append(float(np.abs(grads[k] - e).max() / np.abs(e).max()))
            r_ref.append(float(np.abs(trainer.last_grads[k].numpy() - e).max() / np.abs(e).max()))
            if "_spread" in exact:
                r_alt.append(float(np.abs(exact["_spread"][k] - e).max() / np.abs(e).max()))
        r_hip, r_ref = np.array(r_hip), np.array(r_ref)
        # without the second fp32 order of the oracle: the upper range of the spread measured over sizes and seeds (oracle/conditioning.py: 1e-6 .. 2e-3)
        r_alt = np.array(r_alt) if r_alt else np.array([2e-3])
        spread = max(float(np.median(r_ref)), float(np.median(r_alt)))
        print("gradient distance to float64 (median / max): HIP %.2e / %.2e, oracle fp32 %.2e / %.2e, oracle fp32 reversed batch %.2e / %.2e"
              % (np.median(r_hip), r_hip.max(), np.median(r_ref), r_ref.max(), np.median(r_alt), r_alt.max()))
        bound = 2.0 * spread + 5e-3
        assert np.median(r_hip) <= bound, (float(np.median(r_hip)), spread)
        assert np.quantile(r_hip, 0.9) <= 10.0 * bound, (float(np.quantile(r_hip, 0.9)), spread)
        assert r_hip.max() < 0.5, float(r_hip.max())
        del tol_grad                                           # (kept in the signature: the bound now comes from the oracle's own spread)
    elif check_grads:
        raise AssertionError("check_step: gradient checks are anchored on the float64 oracle - pass exact=exact_grads(...)")
    eng.optimizer_step(1.0)
    torch.cuda.synchronize()
    w = eng.get_weights()
    # Adam's first steps move every element by ~lr*sign(g): elements whose gradient is rounding noise can take
    # either sign, so compare the mean displacement tightly and bound the worst element by one sign flip.
    lr = trainer.spec.lr
    for k in trainer.params:
        e = trainer.params[k].detach().numpy()
        d = np.abs(w[k] - e)
        assert d.mean() <= tol_w * np.abs(e).mean() + 5e-5, (k, float(d.mean()))
        assert d.max() <= tol_w * np.abs(e).max() + 2.5 * lr, (k, float(d.max()))
    return got, exp


@pytest.mark.parametrize("loss,opt", [("tanimoto", "adam"), ("weighted_cross_entropy", "sgd"), ("cross_entropy", "adam")])
def test_tiny_multitask_fp32_two_steps(loss, opt):
    shape, C = (64, 64, 3), 4
    cw = [1.0, 2.0, 3.0, 4.0] if loss == "weighted_cross_entropy" else None
    # the loss / optimizer variants run on the four-level graph (8x8 bottleneck, 2.7 M parameters): what they vary is outside the levels
    trainer, eng = make_pair(shape, C, True, 32, loss, opt, cw=cw, depth=6 if loss == "tanimoto" else 4)
    for step in range(2):
        x, y = make_batch(2, 64, 3, C, True, seed=11 + step, block=16)
        ex = exact_grads(trainer, x, y, second_order=(loss == "tanimoto")) if step == 0 else None
        check_step(trainer, eng, x, y, True, 1e-3, 1e-3, 5e-3, 2e-3, check_grads=(step == 0), exact=ex)
        # the second step starts from the oracle's weights: after a ReLU flip (see check_step) Adam moves the affected
        # elements by +-lr the other way, and the 8-sample BatchNorms turn that into ~3e-3 on the logits
        eng.set_weights({k: v.detach().numpy() for k, v in trainer.params.items()})
    # inference path: moving statistics, nothing updated
    x, y = make_batch(2, 64, 3, C, True, seed=31, block=16)
    exp = trainer.test_on_batch(x, y)
    got = eng.test_step(x, y)
    for i in range(5):
        assert abs(got[i] - exp[i]) <= 2e-3 * max(1.0, abs(exp[i])), (i, got[i], exp[i])
    pred = eng.predict(x)
    rp = ref.forward(trainer.cfg, trainer.params, x, training=False)
    for h in pred:
        assert np.abs(pred[h] - rp[h]).max() < 5e-3        # after two noisy Adam steps (see check_step)


def test_tiny_singletask_fp32_128():
    """width-128 input enables the third PSP branch (model2.py:49-50)."""
    shape, C = (128, 128, 7), 2
    trainer, eng = make_pair(shape, C, False, 32, "tanimoto")
    x, y = make_batch(2, 128, 7, C, False, seed=5)
    check_step(trainer, eng, x, y, False, 1e-3, 1e-3, 8e-3, 2e-3, exact=exact_grads(trainer, x, y))


def test_full_width_block_bf16_close_to_oracle():
    """bf16 storage on the reference width (32): loss within 3e-2, logits within 6e-2 of their scale."""
    shape, C = (64, 64, 6), 6
    trainer, eng = make_pair(shape, C, True, 32, "tanimoto", dtype="bf16", split_k=True)
    x, y = make_batch(2, 64, 6, C, True, seed=7, block=16)
    exp = trainer.train_on_batch(x, y)
    g = eng.forward_backward(x, y)
    torch.cuda.synchronize()
    got = eng._results(g)
    for i in range(5):
        assert abs(got[i] - exp[i]) <= 3e-2 * max(1.0, abs(exp[i])), (i, got[i], exp[i])
    for h, z in eng.logits(True, 2).items():
        assert rel(z, trainer.last_taps[h + "_logits"]) < 8e-2, h


_ORACLE_CACHE = {}


def oracle_step(name, shape, C, mt, loss, B, seed, cw=None, depth=6):
    """One oracle train step per full-size configuration, computed once per test session (20-60 s of host time each) and
    shared by the tests that compare different HIP storage types against it."""
    if name not in _ORACLE_CACHE:
        lw = {"seg": 1.0, "bound": 1.0, "dist": 1.0, "color": 1.0}
        rcfg = ref.RefConfig(input_shape=shape, num_classes=C, multitasking=mt, depth=depth)
        params, order = ref.init_params(rcfg, 3)
        tr = ref.RefTrainer(rcfg, {k: v.clone() for k, v in params.items()}, order,
                            ref.CompileSpec(loss=loss, class_weights=cw, loss_weights=lw, optimizer="adam", lr=1e-3))
        x, y = make_batch(B, shape[0], shape[2], C, mt, seed=seed)
        exp = tr.train_on_batch(x, y)
        taps = {k: v for k, v in tr.last_taps.items() if k.endswith("logits")}
        _ORACLE_CACHE[name] = dict(params={k: v.numpy().copy() for k, v in params.items()}, x=x, y=y, exp=exp, taps=taps)
    return _ORACLE_CACHE[name]


def golden_step(name):
    """The same step from the committed fixture tests/golden/fullsize_<name>.npz (oracle/make_golden_fullsize.py: one training-mode
    forward of the CPU oracle at the configuration's own size and batch; the CPU suite re-derives fixtures live,
    tests/test_oracle_kat.py::test_fullsize_goldens_match_the_live_oracle): losses, and per head a strided sample of the logits with
    their scale and two checksums.  Weights and batch are regenerated from their seeds (cheap); the oracle itself - 20-60 s per
    configuration with its backward pass on the GPU box's host cores, 150 s of the round-3 GPU suite - does not run."""
    import os
    from oracle import make_golden_fullsize as mg
    if ("g", name) not in _ORACLE_CACHE:
        shape, C, mt, loss, B, seed, cw, depth = mg.CONFIGS[name]
        rcfg = ref.RefConfig(input_shape=shape, num_classes=C, multitasking=mt, depth=depth)
        params, _ = ref.init_params(rcfg, mg.PARAM_SEED)
        x, y = make_batch(B, shape[0], shape[2], C, mt, seed=seed)
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_%s.npz" % name))
        _ORACLE_CACHE[("g", name)] = dict(params={k: v.numpy().copy() for k, v in params.items()}, x=x, y=y, exp=list(z["losses"]),
                                          digest={k: z[k] for k in z.files}, cfg=mg.CONFIGS[name])
    return _ORACLE_CACHE[("g", name)]


def hip_engine(shape, C, mt, loss, dtype, weights, cw=None, depth=6):
    eng = Engine(ModelConfig(input_shape=shape, num_classes=C, multitasking=mt, depth=depth), dtype=dtype, seed=0, split_k=True)
    heads = ref.HEADS if mt else ["seg"]
    if loss == "tanimoto":
        kind = {h: L.LOSS_TANIMOTO for h in heads}
    else:
        kind = {"seg": L.LOSS_WCE, "bound": L.LOSS_BCE_LOGITS, "dist": L.LOSS_MSE, "color": L.LOSS_MSE}
    eng.compile(LossSpec(kind={h: kind[h] for h in heads}, weight={h: 1.0 for h in heads}, class_weights=cw, optimizer="adam", lr=1e-3))
    eng.set_weights(weights)
    return eng


def compare_full_size(o, eng, mt, tol_loss, tol_logit):
    g = eng.forward_backward(o["x"], o["y"])
    torch.cuda.synchronize()
    got, exp = eng._results(g), o["exp"]
    worst = {}
    for i in range(5 if mt else 1):
        worst["loss%d" % i] = abs(got[i] - exp[i]) / max(1.0, abs(exp[i]))
        assert worst["loss%d" % i] <= tol_loss, (i, got[i], exp[i])
    for h, z in eng.logits(True, o["x"].shape[0]).items():
        key = (h + "_logits") if mt else "logits"
        if "taps" in o:
            worst[h] = rel(z, o["taps"][key])
        else:                                                # fixture: the strided sample against the logits' scale, and the sum as a checksum of the rest
            dg = o["digest"]
            flat = np.ascontiguousarray(z, np.float32).ravel()
            assert flat.size == int(dg[key + "_n"]), (h, flat.size)
            worst[h] = float(np.abs(flat[::int(dg[key + "_stride"])] - dg[key + "_sample"]).max() / float(dg[key + "_maxabs"]))
            csum = abs(float(flat.astype(np.float64).sum()) - float(dg[key + "_sum"])) / float(dg[key + "_sumabs"])
            assert csum < tol_logit * 0.1, (h, "checksum", csum)
        assert worst[h] < tol_logit, (h, worst[h])
    print("full-size parity (%s):" % eng.dtype, {k: float("%.3g" % v) for k, v in worst.items()})
    return got


CFG3 = ("cfg3", (256, 256, 6), 6, True, "tanimoto", 2, 1234)


def test_cfg3_full_size_fp32_loss_and_logits():
    """BASELINE config 3 at full size (256x256x6, 6 classes, multitask Tanimoto, reference width), batch 2:
    loss and per-head logits within the north-star 1e-3 relative tolerance of the CPU oracle."""
    o = oracle_step(*CFG3)                                   # the one LIVE full-size oracle step of the suite (batch 2, with its backward)
    eng = hip_engine((256, 256, 6), 6, True, "tanimoto", "f32", o["params"])
    compare_full_size(o, eng, True, 1e-3, 1e-3)
    assert eng.count_params() == 42736869
    del eng
    torch.cuda.empty_cache()
    # ... and BASELINE's own batch 8 against the committed fixture (oracle forward at batch 8, re-derived live by the CPU suite)
    o8 = golden_step("cfg3")
    assert o8["x"].shape[0] == 8
    compare_full_size(o8, hip_engine((256, 256, 6), 6, True, "tanimoto", "f32", o8["params"]), True, 1e-3, 1e-3)


def test_cfg3_full_size_bf16_bound_and_trajectory():
    """The BENCHMARKED storage type at the benchmarked size AND batch.  (a) bf16 activations / bf16 weight copies
    (fp32 master weights, statistics, losses) against the oracle's batch-8 fixture (tests/golden/fullsize_cfg3.npz): loss within
    2e-3, per-head logits within 5e-2 of their scale (measured: 2.4e-5 and 2.8e-2).  (b) batch 8 (the bench's), ten Adam steps from the same weights,
    bf16 against fp32 storage on the HIP path: the loss trajectories stay within 2e-3 of each other at every step
    (measured: 1.3e-4) and both fall."""
    o = golden_step("cfg3")
    eng = hip_engine((256, 256, 6), 6, True, "tanimoto", "bf16", o["params"])
    compare_full_size(o, eng, True, 2e-3, 5e-2)
    x, y = make_batch(8, 256, 6, 6, True, seed=1234)
    traj = {}
    for dtype in ("f32", "bf16"):
        e = hip_engine((256, 256, 6), 6, True, "tanimoto", dtype, o["params"])
        traj[dtype] = np.array([e.train_step(x, y)[:5] for _ in range(10)])
        del e
        torch.cuda.empty_cache()
    a, b = traj["f32"], traj["bf16"]
    dev = np.abs(a - b) / np.maximum(1.0, np.abs(a))
    print("cfg3 B=8 ten-step trajectory, total loss f32:", np.round(a[:, 0], 4), "bf16:", np.round(b[:, 0], 4), "max dev", float(dev.max()))
    assert np.all(np.isfinite(b)) and dev.max() < 2e-3, (a[:, 0], b[:, 0])
    assert a[-1, 0] < a[0, 0] and b[-1, 0] < b[0, 0]


def test_cfg2_full_size_single_task_fp32_and_bf16():
    """BASELINE config 2: ResUnet-a d6 single-task segmentation head (model2.py:144-147), 256x256x6, 6 classes: loss and
    logits within 1e-3 of the oracle in fp32 storage, within the bf16 bounds of the cfg3 test in bf16 (the configuration's
    own dtype)."""
    o = golden_step("cfg2")
    assert o["x"].shape[0] == 8                              # BASELINE's batch
    eng = hip_engine((256, 256, 6), 6, False, "tanimoto", "f32", o["params"])
    compare_full_size(o, eng, False, 1e-3, 1e-3)
    assert eng.count_params() == 42690134
    del eng
    compare_full_size(o, hip_engine((256, 256, 6), 6, False, "tanimoto", "bf16", o["params"]), False, 2e-3, 5e-2)


def test_cfg1_hip_fp32_vs_the_cpu_baseline_step():
    """BASELINE config 1 (256x256x3, 6 classes, single task, bs 4, weighted CE with unit weights, Adam 1e-3) - the very
    step bench.py's cpu_baseline leg times on the host cores ("parity check on the same run") - on the HIP fp32
    path: loss and logits within 1e-3, and the metrics the reference prints (train_ISPRS.py:458-461)."""
    o = golden_step("cfg1")
    eng = hip_engine((256, 256, 3), 6, False, "weighted_cross_entropy", "f32", o["params"], cw=[1.0] * 6)
    got = compare_full_size(o, eng, False, 1e-3, 1e-3)
    exp = o["exp"]
    assert abs(got[1] - exp[1]) < 2e-3                                  # accuracy
    assert sum(got[2:]) == sum(exp[2:]) == 4 * 256 * 256 * 6            # TP+FP+TN+FN = every one-hot element
    assert all(abs(a - b) <= 1e-3 * sum(exp[2:]) for a, b in zip(got[2:], exp[2:]))


def test_graph_replay_equals_eager_launches():
    """The captured-HIP-graph step (single GPU fast path) must reproduce the eager launch sequence up to
    atomic-order noise: same losses and same weights after three SGD steps (SGD is linear in the gradient, so
    rounding noise is not sign-amplified the way Adam's first steps do)."""
    shape, C = (64, 64, 6), 6
    outs = []
    for use_graph in (False, True):
        _, eng = make_pair(shape, C, True, 32, "tanimoto", "sgd", dtype="bf16", seed=5, split_k=False)
        eng.use_graph = use_graph
        losses = []
        for step in range(3):
            x, y = make_batch(2, 64, 6, C, True, seed=40 + step, block=16)
            losses.append(eng.train_step(x, y)[:5])
        torch.cuda.synchronize()
        outs.append((np.array(losses), eng.P.cpu().numpy().copy(), eng.t))
    assert outs[0][2] == outs[1][2] == 3
    assert np.allclose(outs[0][0], outs[1][0], rtol=2e-3, atol=2e-4), (outs[0][0], outs[1][0])
    d = np.abs(outs[0][1] - outs[1][1])
    moved = np.abs(outs[0][1]).mean()
    assert d.mean() < 1e-4 * moved and d.max() < 1e-3, (float(d.mean()), float(d.max()), float(moved))


@pytest.mark.parametrize("mt,size,ch", [(True, 64, 3), (False, 128, 7)])
def test_model_py_graph_variant_fp32(mt, size, ch):
    """ResUnet_a/model.py graph (model.py:14-171): no skip term in ResBlock, PSP = pool -> conv -> upsample without BN,
    1x1 conv before UpSampling2D, combine without BN, no ReLU after PSP.  Same bar as the model2 tests."""
    shape, C = (size, size, ch), 4
    trainer, eng = make_pair(shape, C, mt, 32, "tanimoto", "adam", variant="model")
    assert eng.count_params() == ref.count_params(trainer.params)
    for step in range(2):
        x, y = make_batch(2, size, ch, C, mt, seed=17 + step, block=16)
        ex = exact_grads(trainer, x, y) if step == 0 else None
        check_step(trainer, eng, x, y, mt, 1e-3, 1e-3, 8e-3, 2e-3, check_grads=(step == 0), exact=ex)
        # Adam moves noise-level gradient entries by +-lr either way, and this graph amplifies such weight differences
        # ~10x into the logits: continue from the oracle's weights (check_step has just compared the updated ones)
        eng.set_weights({k: v.detach().numpy() for k, v in trainer.params.items()})
    x, y = make_batch(2, size, ch, C, mt, seed=33, block=16)
    exp = trainer.test_on_batch(x, y)
    got = eng.test_step(x, y)
    for i in range(5 if mt else 1):
        assert abs(got[i] - exp[i]) <= 2e-3 * max(1.0, abs(exp[i])), (i, got[i], exp[i])


def test_odd_batch_partial_tiles_fp32():
    """B = 3 on 64x64: 12288 pixels at the top level, 3 at the 1x1-pooled PSP branch - none of the tile sizes (32, 64, 128,
    256 rows) divides every level, so the ragged-tile paths of every kernel run (and batch statistics over 3 samples)."""
    shape, C = (64, 64, 3), 4
    trainer, eng = make_pair(shape, C, True, 32, "weighted_cross_entropy", "sgd", cw=[1.0, 2.0, 0.5, 1.5])
    x, y = make_batch(3, 64, 3, C, True, seed=23, block=16)
    check_step(trainer, eng, x, y, True, 1e-3, 1e-3, 5e-3, 2e-3, exact=exact_grads(trainer, x, y))


def test_odd_batch_bf16_graph_steps():
    """Same ragged shapes through the bf16 kernels (conv_dmap / conv_halo are bf16-only) and the whole-step HIP graph:
    three Adam steps must track the oracle's loss within bf16 storage error and stay finite."""
    shape, C = (128, 128, 6), 6
    trainer, eng = make_pair(shape, C, True, 32, "tanimoto", "adam", dtype="bf16", split_k=True)
    for step in range(3):
        x, y = make_batch(3, 128, 6, C, True, seed=50 + step)
        exp = trainer.train_on_batch(x, y)
        got = eng.train_step(x, y)
        assert np.all(np.isfinite(got))
        assert abs(got[0] - exp[0]) <= 5e-2 * max(1.0, abs(exp[0])), (step, got[0], exp[0])


def test_normalise_on_load_resblocks_match_materialised_batchnorm():
    """The top-level ResBlocks run with BatchNorm + ReLU applied on load by the conv / weight-gradient kernels (conv_strip,
    wgrad_taps: no normalised copy of the conv input in HBM, model2.py:17-24) - against the same engine with the copies
    materialised by rua_bn_fwd, and both against fp32 storage on the HIP path.  Losses and logits of the two bf16 variants agree
    like any two bf16 evaluations of this graph (2e-3 / 5e-2 of the logit scale: each is ~2.5e-2 from the oracle).  Gradients
    at batch 1 are noisy in bf16 (median distance to the fp32 gradient ~0.17 of a tensor's scale for EITHER variant: the
    bottleneck BatchNorms see 64 samples): the bar is that normalising on load is no further from the fp32 gradient than
    materialising is."""
    shape, C = (256, 256, 6), 6
    x, y = make_batch(1, 256, 6, C, True, seed=99)
    res = {}
    for tag, fuse, dtype in (("fused", True, "bf16"), ("copies", False, "bf16"), ("f32", False, "f32")):
        eng = Engine(ModelConfig(input_shape=shape, num_classes=C, multitasking=True), dtype=dtype, seed=4, split_k=True)
        eng.fuse_bn = fuse
        eng.compile(LossSpec(kind={h: L.LOSS_TANIMOTO for h in ref.HEADS}, weight={h: 1.0 for h in ref.HEADS}))
        g = eng.forward_backward(x, y)
        torch.cuda.synchronize()
        names = [c[1] for c in g.fwd.calls]
        res[tag] = (eng._results(g), eng.logits(True, 1), eng.grads_keras(), names.count("rua_bn_fwd"), len(names))
        del eng, g
        torch.cuda.empty_cache()
    (la, za, ga, bna, na), (lb, zb, gb, bnb, nb), (lf, zf, gf, _, _) = res["fused"], res["copies"], res["f32"]
    assert bna < bnb and na <= nb                              # coefficient-only launches replace the apply passes
    for i in range(5):
        assert abs(la[i] - lb[i]) <= 2e-3 * max(1.0, abs(lb[i])), (i, la[i], lb[i])
        assert abs(la[i] - lf[i]) <= 2e-3 * max(1.0, abs(lf[i])), (i, la[i], lf[i])
    for h in za:
        assert rel(za[h], zb[h]) < 5e-2 and rel(za[h], zf[h]) < 5e-2, (h, rel(za[h], zb[h]), rel(za[h], zf[h]))
    gmax = max(float(np.abs(v).max()) for v in gf.values())
    keys = [k for k in gf if k.endswith(("/kernel", "/gamma"))]          # well-conditioned sums (biases / betas cancel to ~0)
    dist = lambda g: np.array([float(np.abs(g[k] - gf[k]).max() / max(np.abs(gf[k]).max(), 1e-3 * gmax)) for k in keys])
    da, db = dist(ga), dist(gb)
    print("bf16 gradient distance to the fp32 gradient (median / max): on load %.3f / %.3f, copies %.3f / %.3f" % (np.median(da), da.max(), np.median(db), db.max()))
    assert np.median(da) <= 1.25 * np.median(db) + 1e-2 and da.max() <= 1.25 * db.max() + 5e-2, (np.median(da), np.median(db), da.max(), db.max())
    assert all(np.isfinite(ga[k]).all() for k in ga)


def test_inference_with_normalise_on_load_matches_materialised_and_oracle():
    """Evaluation mode (moving statistics, test_ISPRS.py:28 / test_on_batch) through the normalise-on-load ResBlocks: the
    coefficient-only BatchNorm launch takes the moving statistics, conv_strip applies them on load.  bf16 predictions with and
    without the fused blocks, and the oracle's, agree within the bf16 bounds; batch 1 is what `predict(batch_size=1)` uses."""
    shape, C = (256, 256, 6), 6
    rcfg = ref.RefConfig(input_shape=shape, num_classes=C, multitasking=True)
    params, order = ref.init_params(rcfg, 9)
    rng = np.random.default_rng(1)
    for k in params:                                           # non-trivial moving statistics
        if k.endswith("moving_mean"):
            params[k] = torch.from_numpy(rng.normal(0, 0.2, params[k].shape).astype(np.float32))
        if k.endswith("moving_variance"):
            params[k] = torch.from_numpy(rng.uniform(0.5, 1.5, params[k].shape).astype(np.float32))
    x, _ = make_batch(1, 256, 6, C, True, seed=5)
    exp = ref.forward(rcfg, params, x, training=False)
    outs = []
    for fuse in (True, False):
        eng = Engine(ModelConfig(input_shape=shape, num_classes=C, multitasking=True), dtype="bf16", seed=0)
        eng.fuse_bn = fuse
        eng.compile(LossSpec(kind={h: L.LOSS_TANIMOTO for h in ref.HEADS}, weight={h: 1.0 for h in ref.HEADS}))
        eng.set_weights({k: v.numpy() for k, v in params.items()})
        outs.append(eng.predict(x))
        names = [c[1] for c in eng.graph(1, False).fwd.calls]
        assert ("rua_bn_apply" not in names)
        del eng
        torch.cuda.empty_cache()
    for h in outs[0]:
        assert np.abs(outs[0][h] - outs[1][h]).max() < 3e-2, h            # probabilities in [0, 1]
        assert np.abs(outs[0][h] - np.asarray(exp[h])).max() < 3e-2, h


# ---- the other BASELINE configurations in their OWN form: dtype bf16, their own per-GPU batch ---------------
def test_cfg5_own_form_bf16_batch32():
    """BASELINE config 5 as it is benchmarked (amazon_py/main_tcc.py:37,131 patch size / bands; model2.py:144-147 single-task head):
    128x128x7 patches, 2 classes, bf16 storage, batch 32 - 128-pixel strips, the 3-branch PSPPooling (model2.py:49-52) and the 4x4
    bottleneck, none of which cfg3 reaches.  Loss and logits against the CPU oracle on the same step, same bounds as cfg3's bf16
    test (loss 2e-3, logits 5e-2 of their scale)."""
    o = golden_step("cfg5")
    eng = hip_engine((128, 128, 7), 2, False, "tanimoto", "bf16", o["params"])
    compare_full_size(o, eng, False, 2e-3, 5e-2)
    assert eng.count_params() == 42163914
    del eng
    torch.cuda.empty_cache()
    compare_full_size(o, hip_engine((128, 128, 7), 2, False, "tanimoto", "f32", o["params"]), False, 1e-3, 1e-3)


def test_cfg4_d7_512_bf16_and_fp32_at_its_own_batch4():
    """BASELINE config 4 (the d7 extrapolation, SURVEY A15: not in the reference, restatement-vs-kernel) in its own storage type and at
    its OWN batch: 512x512x6, multitask, bf16, batch 4 per GPU.  (Round 3 ran batch 2: the float32 oracle step held ~12 GB of autograd
    state there.  Loss and logits need no autograd: the fixture is a training-mode forward of the oracle under torch.no_grad at batch 4,
    oracle/make_golden_fullsize.py.)  Same bf16 bounds as cfg3 (loss 2e-3, logits 5e-2), fp32 storage within 1e-3."""
    o = golden_step("cfg4")
    eng = hip_engine((512, 512, 6), 6, True, "tanimoto", "bf16", o["params"], depth=7)
    compare_full_size(o, eng, True, 2e-3, 5e-2)
    assert eng.count_params() > 150e6
    del eng
    torch.cuda.empty_cache()
    # fp32 storage on the same oracle step (encoder stage 7 - 1x1 s2 -> 2048, ResBlock(2048,[1]) -, PSPPooling(2048) on the 8x8 bottleneck of a
    # 512x512 patch, one more decoder stage): loss and per-head logits within 1e-3
    compare_full_size(o, hip_engine((512, 512, 6), 6, True, "tanimoto", "f32", o["params"], depth=7), True, 1e-3, 1e-3)


def test_cfg3_full_size_gradients_fp32_vs_float64_oracle_and_bf16_at_batch8():
    """Parameter gradients at the BENCHMARKED size (256x256x6 multitask Tanimoto, every kernel on its full-size dispatch: conv_strip /
    conv_band at 256 x 256, split-K at the bottom levels).  (a) fp32 storage, batch 2, against the oracle evaluated in float64 with
    the criterion of check_step (median distance within 2x the spread of two fp32 summation orders of the oracle + 5e-3, 90th
    percentile within 10x, no tensor off by half its scale).  (b) bf16 against fp32 storage on the HIP path at batch 8 (the bench's):
    per-tensor distance of the kernel / gamma gradients relative to the tensor's scale - the number to read beside the 0.17 median
    that the batch-1 test (test_normalise_on_load_resblocks_match_materialised_batchnorm) records; bounds from the measurement."""
    lw = {"seg": 1.0, "bound": 1.0, "dist": 1.0, "color": 1.0}
    trainer, eng = make_pair((256, 256, 6), 6, True, 32, "tanimoto", lw=lw, split_k=True)
    x, y = make_batch(2, 256, 6, 6, True, seed=1234)
    check_step(trainer, eng, x, y, True, 1e-3, 1e-3, 0.0, 2e-3, exact=exact_grads(trainer, x, y, second_order=True))
    weights = {k: v.detach().numpy().copy() for k, v in trainer.params.items()}
    del trainer, eng
    torch.cuda.empty_cache()
    x, y = make_batch(8, 256, 6, 6, True, seed=1234)
    grads = {}
    for dtype in ("f32", "bf16"):
        e = hip_engine((256, 256, 6), 6, True, "tanimoto", dtype, weights)
        e.forward_backward(x, y)
        torch.cuda.synchronize()
        grads[dtype] = e.grads_keras()
        del e
        torch.cuda.empty_cache()
    gf, gb = grads["f32"], grads["bf16"]
    gmax = max(float(np.abs(v).max()) for v in gf.values())
    keys = [k for k in gf if k.endswith(("/kernel", "/gamma"))]
    dist = np.array([float(np.abs(gb[k] - gf[k]).max() / max(np.abs(gf[k]).max(), 1e-3 * gmax)) for k in keys])
    cos = np.array([float((gb[k].ravel() @ gf[k].ravel()) / (np.linalg.norm(gb[k]) * np.linalg.norm(gf[k]) + 1e-30)) for k in keys])
    print("cfg3 B=8 bf16 vs fp32 gradient: distance median %.3f / p90 %.3f / max %.3f of the tensor's scale; cosine median %.4f / min %.4f"
          % (np.median(dist), np.quantile(dist, 0.9), dist.max(), np.median(cos), cos.min()))
    for i in np.argsort(-dist)[:6]:
        print("   farthest: %-28s distance %.3f cosine %.4f |g| max %.3g (largest of the model %.3g)" % (keys[i], dist[i], cos[i], np.abs(gf[keys[i]]).max(), gmax))
    for i in np.argsort(cos)[:6]:
        print("   least aligned: %-28s cosine %.4f distance %.3f |g| max %.3g" % (keys[i], cos[i], dist[i], np.abs(gf[keys[i]]).max()))
    assert all(np.isfinite(gb[k]).all() for k in gb)
    assert np.median(dist) < BF16_GRAD_MEDIAN_B8 and dist.max() < BF16_GRAD_MAX_B8 and np.median(cos) > BF16_GRAD_COS_B8 and cos.min() > BF16_GRAD_COSMIN_B8


def test_first_writer_overwrite_step_equals_the_accumulating_backward():
    """The gradient arena's two conventions: forward_backward() on its own ADDS to G; inside a whole step (train_step, the
    captured graph) the weight-gradient kernels with one producer per tensor STORE over the zeroed arena (rua_wgrad_desc.overwrite_dev: the
    single-K-slice epilogues, the slab / block-partial reductions, wgrad_img).  Same batch, same weights, both conventions on a zeroed
    arena: every gradient tensor agrees to fp32 summation noise (the deterministic kernels bit for bit) - at cfg3's full size, where every
    kernel family takes its full-size dispatch.  And a whole step that FOLLOWS an unapplied forward_backward() accumulates on top of it."""
    o = golden_step("cfg3")
    x, y = o["x"][:2], {k: v[:2] for k, v in o["y"].items()}
    eng = hip_engine((256, 256, 6), 6, True, "tanimoto", "bf16", o["params"])
    n = eng.params.n
    eng.forward_backward(x, y)
    torch.cuda.synchronize()
    g_acc = eng.G[:n].clone()
    assert eng._g_pending and eng._ow == 0
    eng.G.zero_(); eng._g_pending = False
    eng.forward_backward(x, y, _whole_step=True)
    torch.cuda.synchronize()
    assert eng._ow == 1
    g_ow = eng.G[:n].clone()
    scale = float(g_acc.abs().max())
    diff = float((g_ow - g_acc).abs().max())
    same = float((g_ow == g_acc).float().mean())
    print("overwrite vs accumulate on a zero arena: max |diff| %.3g of scale %.3g, %.2f %% of the elements bit-identical" % (diff, scale, 100 * same))
    assert diff <= 1e-5 * scale and same > 0.9
    # unapplied gradients in the arena: the next whole step must not overwrite them
    eng._g_pending = True
    eng.forward_backward(x, y, _whole_step=True)
    torch.cuda.synchronize()
    assert eng._ow == 0
    twice = eng.G[:n]
    assert float((twice - 2 * g_acc).abs().max()) <= 2e-5 * scale


BF16_GRAD_COSMIN_B8 = 0.85                                                      # 1 - cosine within twice the measured 0.094 of the least aligned tensor
BF16_GRAD_MEDIAN_B8, BF16_GRAD_MAX_B8, BF16_GRAD_COS_B8 = 0.10, 0.6, 0.95     # measured: median 0.053 / p90 0.198 / max 0.321 of a tensor's scale, cosine median 0.976 / min 0.906 (batch 1: median 0.17)


def test_cfg3_bf16_training_tracks_fp32_over_200_steps_on_varying_batches():
    """Training quality of the BENCHMARKED storage type (VERDICT r3 next#5, r4 next#5c): 200 Adam steps of cfg3 (256x256x6 multitask,
    Tanimoto-dual on all heads, batch 8, lr 1e-3) over eight DIFFERENT synthetic batches in rotation, bf16 storage against fp32 storage on
    the HIP path from the same initial weights - and a SECOND fp32 run whose initial weights differ by 1e-6 relative noise.  Round 5's
    diagnosis (tools/bf16_gap.py): this training problem is chaotic at the percent level - two fp32 runs 1e-6 apart end 1.4 %
    apart (worst step 4.8 %), fp32 with bf16-rounded weight copies -0.2 %, fp32 from bf16-rounded initial weights -0.7 %, bf16 +0.4 % (round
    4's run: +4.6 %), bf16 with materialised BatchNorm -3.0 %: the bf16 run lands on either side of the fp32 run, inside the spread fp32
    shows against itself - trajectory divergence, not a bias of the storage type.  The fp32 pair's own gap is printed beside the bf16
    run's (a single draw of a quantity that scatters over 0.1 - 5 %, so it is context, not the bound); the bounds are absolute:
    BF16_TRAIN_GAP at the end, BF16_TRAIN_STEP_GAP at the worst step; all runs fall to about 60 % of the first loss."""
    batches = [make_batch(8, 256, 6, 6, True, seed=9000 + i) for i in range(8)]
    o = golden_step("cfg3")
    rng = np.random.default_rng(7)
    w_eps = {k: (v * (1.0 + 1e-6 * rng.standard_normal(v.shape))).astype(np.float32) for k, v in o["params"].items()}
    traj = {}
    for name, dtype, w in (("f32", "f32", o["params"]), ("f32_eps", "f32", w_eps), ("bf16", "bf16", o["params"])):
        e = hip_engine((256, 256, 6), 6, True, "tanimoto", dtype, w)
        traj[name] = np.array([e.train_step(*batches[i % len(batches)])[:5] for i in range(200)])
        del e
        torch.cuda.empty_cache()
    a, a2, b = traj["f32"][:, 0], traj["f32_eps"][:, 0], traj["bf16"][:, 0]
    end = lambda v: abs(v[-20:].mean() - a[-20:].mean()) / a[-20:].mean()
    step = lambda v: float(np.max(np.abs(v - a) / a))
    print("cfg3 B=8, 200 steps on 8 rotating batches: total loss f32 %.4f -> %.4f, bf16 %.4f -> %.4f; mean of the last 20 steps: f32 %.4f, f32 + 1e-6 %.4f, "
          "bf16 %.4f; relative gap at the end: fp32 pair %.2e, bf16 %.2e; largest per-step gap: fp32 pair %.2e, bf16 %.2e"
          % (a[0], a[-1], b[0], b[-1], a[-20:].mean(), a2[-20:].mean(), b[-20:].mean(), end(a2), end(b), step(a2), step(b)))
    assert all(np.all(np.isfinite(v)) for v in traj.values())
    assert a[-20:].mean() < 0.8 * a[:8].mean() and b[-20:].mean() < 0.8 * b[:8].mean()       # both fall (measured: to about 60 %)
    assert end(b) < BF16_TRAIN_GAP and step(b) < BF16_TRAIN_STEP_GAP, (end(b), step(b), end(a2), step(a2))


BF16_TRAIN_GAP, BF16_TRAIN_STEP_GAP = 0.09, 0.18     # absolute caps = 2x / 1.7x the largest gaps ever measured (round 4: 4.6e-2 at the end, 1.05e-1 at the worst step; round 5, six builds: 0.4 - 4.5e-2 / 4.1 - 9.3e-2, with either sign - and the fp32 pair 0.3 - 2.5e-2 / 3.9 - 7.0e-2: a draw of a chaotic quantity, so the cap leaves room)
