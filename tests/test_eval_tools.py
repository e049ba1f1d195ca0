"""Host functions of the evaluation script (reference test_ISPRS.py:39-210): tiling, mosaic, label maps."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("eval_isprs", os.path.join(ROOT, "test_ISPRS.py"))
ev = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ev)


def test_tiles_and_mosaic_round_trip_row_major():
    rng = np.random.default_rng(0)
    ref = rng.integers(0, 5, size=(70, 100)).astype(np.uint8)            # does not divide by the patch size
    p = ev.extract_patches_test(ref, 32)
    assert p.shape == (2 * 3, 32, 32)
    assert np.array_equal(p[4], ref[32:64, 32:64])                       # row-major: patch 4 = row 1, column 1
    back = ev.pred_recostruction(32, p, ref, img_type=1)
    assert back.shape == ref.shape
    assert np.array_equal(back[:64, :96], ref[:64, :96]) and not back[64:].any() and not back[:, 96:].any()
    img = rng.random((70, 100, 3)).astype(np.float32)
    q = ev.extract_patches_train(img, 32)
    assert q.shape == (6, 32, 32, 3) and np.allclose(q[5], img[32:64, 64:96])
    assert np.allclose(ev.pred_recostruction(32, q, ref, img_type=2)[:64, :96], img[:64, :96])


def test_label_maps_and_normalisation():
    colours = [eval(k) for k in ev.LABEL_DICT]
    cls = np.random.default_rng(1).integers(0, 5, size=(9, 7))
    rgb = np.array(colours, np.uint8)[cls]
    b = ev.binarize_matrix(rgb, ev.LABEL_DICT)
    assert b.dtype == np.uint8 and np.array_equal(b, cls)
    assert np.array_equal(ev.convert_preds2rgb(b.astype(np.float64), ev.LABEL_DICT), rgb)
    rgb[0, 0] = (1, 2, 3)
    with pytest.raises(KeyError):
        ev.binarize_matrix(rgb, ev.LABEL_DICT)
    x = np.full((2, 2, 3), 255, np.float32)
    assert np.allclose(ev.normalize_rgb(x.copy(), 1), 1.0)
    assert np.allclose(ev.normalize_rgb(x.copy(), 2), 255 / 126.5)        # the reference's precedence quirk, kept on purpose


def test_metrics_match_hand_count():
    t = np.array([0, 0, 1, 1, 2, 2, 2, 1])
    p = np.array([0, 1, 1, 1, 2, 0, 2, 1])
    acc, f1, rec, prec = ev.compute_metrics_hw(t, p)
    assert acc == pytest.approx(75.0)
    assert rec == pytest.approx([50.0, 100.0, 100 * 2 / 3])
    assert prec == pytest.approx([50.0, 75.0, 100.0])
    assert f1[1] == pytest.approx(100 * 2 * 0.75 / 1.75)
