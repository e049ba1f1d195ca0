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


@pytest.mark.parametrize("run", [0, 1])
def test_metric_functions_reproduce_the_numbers_the_reference_printed(run):
    """The only numbers the reference holds for this repository's paths: two confusion matrices of the ISPRS test tile and the
    accuracy / F1 / recall / precision utils.compute_metrics printed from the same label vectors
    (/root/reference/infos_training_train_on_batch.txt:77-88, 104-116; utils.py:52-57 = test_ISPRS.py:39-45).  Label vectors
    with exactly that confusion matrix must give exactly those metrics through this repository's compute_metrics_hw."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_confusion.json")))["runs"][run]
    cm = np.asarray(gold["confusion"], np.int64)
    n = cm.shape[0]
    true = np.repeat(np.repeat(np.arange(n), n), cm.reshape(-1))          # cell (i, j) -> cm[i, j] pixels of class i predicted j
    pred = np.repeat(np.tile(np.arange(n), n), cm.reshape(-1))
    from sklearn.metrics import confusion_matrix
    assert np.array_equal(confusion_matrix(true, pred), cm)
    acc, f1, rec, prec = ev.compute_metrics_hw(true, pred)
    assert acc == pytest.approx(gold["accuracy"], rel=1e-12)
    for got, name in ((f1, "f1score"), (rec, "recall"), (prec, "precision")):
        assert np.allclose(got, gold[name], rtol=0, atol=1e-6), (name, got, gold[name])     # printed with 8 decimals
