"""Known-answer tests that pin the CPU oracle (SURVEY.md §8c list) and require the two
independent restatements (torch functional vs numpy loops) to agree."""
import numpy as np
import pytest
import torch

from oracle import naive_ops as nv
from oracle import resuneta_ref as ref

rng = np.random.default_rng(7)


def nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).permute(0, 3, 1, 2)


# (1) Tanimoto dual ------------------------------------------------------------------
def onehot(B, H, W, C, seed=0):
    ids = np.random.default_rng(seed).integers(0, C, size=(B, H, W))
    return np.eye(C, dtype=np.float32)[ids]


def test_tanimoto_perfect_prediction_is_zero():
    y = onehot(2, 8, 8, 3)
    assert np.allclose(nv.tanimoto_dual_loss(y, y), 0.0, atol=1e-6)
    assert torch.allclose(ref.tanimoto_dual_loss(nchw(y), nchw(y)), torch.zeros(2), atol=1e-6)


def test_tanimoto_inverted_prediction_is_about_one():
    y = onehot(2, 8, 8, 3)
    l = nv.tanimoto_dual_loss(y, 1 - y)
    assert np.all(l > 0.999)


def test_tanimoto_absent_class_gets_max_finite_weight():
    y = onehot(2, 4, 4, 3)
    y[..., 2] = 0            # class 2 absent everywhere -> Vli=0 -> inf -> max finite weight
    y[..., 0] = 1 - y[..., 1]
    p = rng.uniform(0.05, 0.95, y.shape).astype(np.float32)
    a = nv.tanimoto_loss(y, p)
    b = ref.tanimoto_loss(nchw(y), nchw(p)).numpy()
    assert np.allclose(a, b, rtol=1e-5)
    # hand computation
    v = y.sum(axis=(1, 2)).mean(axis=0)
    w = np.array([1 / v[0] ** 2, 1 / v[1] ** 2, 0])
    w[2] = w[:2].max()
    sp = (p * y).sum(axis=(1, 2))
    ss = (p ** 2 + y ** 2).sum(axis=(1, 2))
    exp = ((w * sp).sum(-1) + 1e-5) / ((w * (ss - sp)).sum(-1) + 1e-5)
    assert np.allclose(a, exp, rtol=1e-6)


def test_tanimoto_all_weights_inf_gives_one():
    z = np.zeros((2, 4, 4, 3), np.float32)
    assert np.allclose(nv.tanimoto_loss(z, z), 1.0)
    assert np.allclose(ref.tanimoto_loss(nchw(z), nchw(z)).numpy(), 1.0)


def test_tanimoto_swapped_argument_weights_come_from_pred():
    # 1x2x2x2 example: label volumes (3,1), pred volumes (2,2) -> first-term weights 1/4,1/4
    y = np.zeros((1, 2, 2, 2), np.float32)
    y[0, :, :, 0] = [[1, 1], [1, 0]]
    y[0, :, :, 1] = [[0, 0], [0, 1]]
    p = np.zeros_like(y)
    p[0, :, :, 0] = [[1, 1], [0, 0]]
    p[0, :, :, 1] = [[0, 0], [1, 1]]
    w1 = np.array([1 / 4, 1 / 4])                       # from pred volumes
    sp = np.array([2.0, 1.0]); ss = np.array([5.0, 3.0])
    l1 = ((w1 * sp).sum() + 1e-5) / ((w1 * (ss - sp)).sum() + 1e-5)
    yc, pc = 1 - y, 1 - p                               # complement volumes of label: (1,3)
    w2 = np.array([1.0, 1 / 9])
    sp2 = (pc * yc).sum(axis=(1, 2))[0]; ss2 = (pc ** 2 + yc ** 2).sum(axis=(1, 2))[0]
    l2 = ((w2 * sp2).sum() + 1e-5) / ((w2 * (ss2 - sp2)).sum() + 1e-5)
    exp = 1 - 0.5 * (l1 + l2)
    assert np.allclose(nv.tanimoto_dual_loss(y, p), exp, rtol=1e-6)
    assert np.allclose(ref.tanimoto_dual_loss(nchw(y), nchw(p)).numpy(), exp, rtol=1e-5)
    # would differ if the weights came from the label volumes (9x different ratio)
    wl = np.array([1 / 9, 1.0])
    wrong = ((wl * sp).sum() + 1e-5) / ((wl * (ss - sp)).sum() + 1e-5)
    assert abs(wrong - l1) > 1e-2


# (3) weighted CE --------------------------------------------------------------------
def test_weighted_ce_uniform_and_clip():
    C = 5
    w = np.array([4.3, 2.9, 3.9, 5.6, 374.0])
    y = onehot(1, 2, 2, C, seed=3)
    p = np.full_like(y, 1.0 / C)
    out = nv.weighted_cce(w, y, p)
    assert np.allclose(out, (y * w).sum(-1) * np.log(C))
    p0 = y[..., ::-1].copy()                            # prob 0 on the true class (where different)
    p0 = np.where(y == 1, 0.0, 1.0 / (C - 1)).astype(np.float32)
    out0 = nv.weighted_cce(w, y, p0)
    assert np.allclose(out0, -(y * w).sum(-1) * np.log(1e-7))
    t = ref.weighted_cce(w)(nchw(y), nchw(p0)).numpy()
    assert np.allclose(t, out0, rtol=1e-5)


# (4) dilated conv -------------------------------------------------------------------
@pytest.mark.parametrize("d", [1, 3, 15])
def test_dilated_conv_delta_lands_at_plus_minus_d(d):
    H = 2 * d + 3
    x = np.zeros((1, H, H, 1), np.float32)
    x[0, d + 1, d + 1, 0] = 1
    k = np.arange(1, 10, dtype=np.float32).reshape(3, 3, 1, 1)
    y = nv.conv2d_nhwc(x, k, dilation=d, padding="same")[0, :, :, 0]
    for i in range(3):
        for j in range(3):
            # correlation: y[p] = sum k[i,j] x[p + (i-1)d, (j-1)d]  => tap (i,j) shows at centre-(i-1)d
            assert y[d + 1 - (i - 1) * d, d + 1 - (j - 1) * d] == k[i, j, 0, 0]
    assert np.count_nonzero(y) == 9


def test_dilated_conv_border_counts_d15_on_16():
    x = np.ones((1, 16, 16, 1), np.float32)
    k = np.ones((3, 3, 1, 1), np.float32)
    y = nv.conv2d_nhwc(x, k, dilation=15, padding="same")[0, :, :, 0]
    # rows 1..14 see only the centre row; row 0 sees rows {0,15}; same for columns
    assert y[5, 5] == 1 and y[0, 5] == 2 and y[0, 0] == 4 and y[15, 15] == 4 and y[15, 7] == 2
    t = torch.nn.functional.conv2d(torch.ones(1, 1, 16, 16), torch.ones(1, 1, 3, 3), padding=15, dilation=15)
    assert np.array_equal(t[0, 0].numpy(), y)


def test_conv_restatements_agree():
    x = rng.standard_normal((2, 9, 11, 5)).astype(np.float32)
    k = rng.standard_normal((3, 3, 5, 4)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    for d in (1, 3):
        a = nv.conv2d_nhwc(x, k, b, dilation=d, padding="same")
        t = torch.nn.functional.conv2d(nchw(x), torch.from_numpy(k).permute(3, 2, 0, 1), torch.from_numpy(b),
                                       padding=d, dilation=d).permute(0, 2, 3, 1).numpy()
        assert np.allclose(a, t, atol=1e-4)


# (5) stride-2 1x1 samples even indices -------------------------------------------------
def test_stride2_1x1_samples_even_pixels():
    x = np.arange(36, dtype=np.float32).reshape(1, 6, 6, 1)
    y = nv.conv2d_nhwc(x, np.ones((1, 1, 1, 1), np.float32), stride=2)
    assert np.array_equal(y[0, :, :, 0], x[0, ::2, ::2, 0])


# (6) PSP identities ---------------------------------------------------------------------
def test_pool_upsample_ramp_and_commutation():
    x = np.arange(64, dtype=np.float32).reshape(1, 8, 8, 1)
    p = nv.maxpool(x, 2)
    assert np.array_equal(p[0, :, :, 0], x[0, 1::2, 1::2, 0])
    u = nv.upsample_nearest(p, 2)
    assert u.shape == x.shape and u[0, 0, 0, 0] == u[0, 1, 1, 0] == 9
    z = rng.standard_normal((1, 4, 4, 3)).astype(np.float32)
    k = rng.standard_normal((1, 1, 3, 2)).astype(np.float32)
    a = nv.conv2d_nhwc(nv.upsample_nearest(z, 2), k)
    b = nv.upsample_nearest(nv.conv2d_nhwc(z, k), 2)
    assert np.allclose(a, b)
    # BN batch statistics invariant under replication
    _, m1, v1 = nv.batchnorm_train(z, 1, 0)
    _, m2, v2 = nv.batchnorm_train(nv.upsample_nearest(z, 4), 1, 0)
    assert np.allclose(m1, m2) and np.allclose(v1, v2)


# (7) fresh BN in inference ----------------------------------------------------------------
def test_fresh_bn_inference_scale():
    x = rng.standard_normal((1, 3, 3, 2))
    y = nv.batchnorm_infer(x, 1.0, 0.0, 0.0, 1.0)
    assert np.allclose(y, x / np.sqrt(1 + 1e-3))


# (8) parameter counts -------------------------------------------------------------------
@pytest.mark.parametrize("shape,C,mt,count", [
    ((256, 256, 6), 6, True, 42736869),
    ((256, 256, 6), 6, False, 42690134),
    ((128, 128, 7), 2, False, 42163914),
])
def test_param_counts(shape, C, mt, count):
    cfg = ref.RefConfig(input_shape=shape, num_classes=C, multitasking=mt)
    params, _ = ref.init_params(cfg, 0)
    assert ref.count_params(params) == count


def test_forward_macs_cfg3():
    cfg = ref.RefConfig(input_shape=(256, 256, 6), num_classes=6, multitasking=True)
    assert abs(ref.forward_macs(cfg) / 1e9 - 42.07) < 0.01


# (9) optimizers ----------------------------------------------------------------------------
def test_adam_first_step_keras_epsilon_placement():
    th, m, v = nv.adam_step(1.0, 0.5, 0.0, 0.0, 1, 1e-3)
    # m=0.05, v=2.5e-4; lr_t = 1e-3*sqrt(1e-3)/0.1 ; step = lr_t*0.05/(sqrt(2.5e-4)+1e-7)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.isclose(th, 1.0 - lr_t * 0.05 / (np.sqrt(2.5e-4) + 1e-7))


def test_sgd_momentum():
    th, vel = nv.sgd_step(1.0, 2.0, 0.0, 0.1)
    th, vel = nv.sgd_step(th, 2.0, vel, 0.1)
    assert np.isclose(vel, 0.8 * -0.2 - 0.2) and np.isclose(th, 1 - 0.2 - 0.36)


# cross-restatement: tiny full model, forward through numpy layer by layer -------------------
def test_tiny_model_forward_matches_numpy_composition():
    cfg = ref.RefConfig(input_shape=(64, 64, 3), num_classes=4, multitasking=False, width=4)
    params, _ = ref.init_params(cfg, 1)
    x = rng.uniform(0, 1, (2, 64, 64, 3)).astype(np.float32)
    out_t, taps = ref.forward(cfg, params, x, training=True, want_taps=True)
    P = {k: v.numpy() for k, v in params.items()}

    class Cnt:
        conv = 0
        bn = 0

    def conv(t, k=None, stride=1, dil=1, padding="valid", name=None):
        if name is None:
            name = "conv2d" if Cnt.conv == 0 else f"conv2d_{Cnt.conv}"
            Cnt.conv += 1
        return nv.conv2d_nhwc(t, P[name + "/kernel"], P[name + "/bias"], stride, dil, padding)

    def bn(t):
        name = "batch_normalization" if Cnt.bn == 0 else f"batch_normalization_{Cnt.bn}"
        Cnt.bn += 1
        return nv.batchnorm_train(t, P[name + "/gamma"], P[name + "/beta"])[0]

    relu = lambda t: np.maximum(t, 0)

    def resblock(t, dils):
        out = t
        for d in dils:
            y = conv(relu(bn(t)), dil=d, padding="same")
            y = conv(relu(bn(y)), dil=d, padding="same")
            out = out + y
        return out

    def psp(t):
        ks = [1, 2]                                        # width 64 < 128
        br = [bn(conv(nv.upsample_nearest(nv.maxpool(t, k), k) if k > 1 else t)) for k in ks]
        return bn(conv(np.concatenate(br + [t], axis=-1)))

    lv = cfg.levels()
    t = conv(x)
    c1 = t
    skips = []
    for i, (nf, dils) in enumerate(lv):
        if i > 0:
            t = conv(t, stride=2)
        t = resblock(t, dils)
        skips.append(t)
    t = relu(psp(t))
    for i in range(len(lv) - 2, -1, -1):
        t = bn(conv(nv.upsample_nearest(t, 2)))
        t = bn(conv(np.concatenate([relu(t), skips[i]], axis=-1)))
        t = resblock(t, lv[i][1])
    t = bn(conv(np.concatenate([relu(t), c1], axis=-1)))
    t = relu(psp(t))
    z = conv(t)
    assert np.allclose(z, taps["logits"], atol=2e-3, rtol=2e-3)
    assert np.allclose(nv.softmax(z), out_t, atol=1e-3)


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
def test_fullsize_goldens_match_the_live_oracle(name):
    """tests/golden/fullsize_<cfg>.npz (what the GPU tests compare the HIP path with at full size) against the oracle run HERE: the same
    training-mode forward, the same digest - every BASELINE configuration at its own size and batch (cfg2 / cfg3 at batch 8 since round 5)."""
    import os
    from oracle import make_golden_fullsize as mg
    exp, taps = mg.oracle_forward(name)
    live = mg.digest(exp, taps)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_%s.npz" % name))
    assert sorted(gold.files) == sorted(live)
    for k in gold.files:
        a, b = np.asarray(gold[k], np.float64), np.asarray(live[k], np.float64)
        scale = max(1.0, float(np.abs(a).max()))
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-5 * scale, (k, float(np.abs(a - b).max()), scale)
