"""Host logic of the asynchronous patch loader: same batches as the reference's serial loop
(train_ISPRS.py:102-141: n // batch_size batches in the given order, last partial batch dropped, float32)."""
import os

import numpy as np
import pytest

from resunet_a_mltsk_keras_amd.loader import PrefetchLoader


def make_dataset(root, n=11, ps=8, cin=3, ncls=4):
    heads = {"seg": ncls, "bound": ncls, "dist": ncls, "color": 3}
    os.makedirs(os.path.join(root, "train"))
    for h in heads:
        os.makedirs(os.path.join(root, "labels", h))
    rng = np.random.default_rng(0)
    xs, ys = [], {h: [] for h in heads}
    for i in range(n):
        name = f"patch_{i}.npy"
        p = os.path.join(root, "train", name)
        np.save(p, rng.random((ps, ps, cin)).astype(np.float32))
        xs.append(p)
        for h, c in heads.items():
            q = os.path.join(root, "labels", h, name)
            np.save(q, (rng.random((ps, ps, c)) * (i + 1)).astype(np.float64 if h == "dist" else np.float32))   # dist: cast path
            ys[h].append(q)
    return xs, ys


def serial_batches(xs, ys, order, B):
    for k in range(len(order) // B):
        idx = order[k * B:(k + 1) * B]
        yield (np.stack([np.load(xs[i]) for i in idx]), {h: np.stack([np.load(v[i]).astype(np.float32) for i in idx]) for h, v in ys.items()})


@pytest.mark.parametrize("depth,workers", [(1, 1), (2, 4), (5, 3)])
def test_prefetch_loader_equals_serial_loop(tmp_path, depth, workers):
    xs, ys = make_dataset(str(tmp_path))
    order = list(np.random.default_rng(3).permutation(len(xs)))
    ld = PrefetchLoader(xs, ys, 4, order=order, depth=depth, workers=workers, pin=False)
    assert len(ld) == 2                                        # 11 // 4: the partial batch is dropped like in the reference
    for epoch in range(2):                                     # reusable for several passes
        got = [(x.numpy().copy(), {h: t.numpy().copy() for h, t in y.items()}) for x, y in ld]
        exp = list(serial_batches(xs, ys, order, 4))
        assert len(got) == len(exp) == 2
        for (gx, gy), (ex, ey) in zip(got, exp):
            assert gx.dtype == np.float32 and np.array_equal(gx, ex)
            for h in ey:
                assert gy[h].dtype == np.float32 and np.array_equal(gy[h], ey[h])
        order = order[::-1]
        ld.set_order(order)


def test_prefetch_loader_slot_stays_valid_until_next_batch(tmp_path):
    """The batch handed out must not be overwritten while the caller still uses it (the step uploads it asynchronously)."""
    xs, ys = make_dataset(str(tmp_path), n=16)
    ld = PrefetchLoader(xs, ys, 2, depth=1, workers=2, pin=False)
    exp = list(serial_batches(xs, ys, list(range(16)), 2))
    it = iter(ld)
    x0, y0 = next(it)
    import time
    time.sleep(0.2)                                            # give the producer every chance to run ahead
    assert np.array_equal(x0.numpy(), exp[0][0]) and np.array_equal(y0["color"].numpy(), exp[0][1]["color"])
    x1, _ = next(it)
    assert np.array_equal(x1.numpy(), exp[1][0])
    it.close()                                                 # abandoning a pass stops the producer
    assert sum(1 for _ in ld) == 8                             # and the loader is usable again


def test_prefetch_loader_reports_missing_file(tmp_path):
    xs, ys = make_dataset(str(tmp_path), n=8)
    os.remove(ys["bound"][5])
    ld = PrefetchLoader(xs, ys, 4, pin=False)
    with pytest.raises(FileNotFoundError):
        for _ in ld:
            pass
    with pytest.raises(ValueError):
        PrefetchLoader(xs, {"seg": ys["seg"][:-1]}, 4, pin=False)


def test_prefetch_loader_shards_the_global_batch(tmp_path):
    """Data parallel: `batch_size` is the global batch; rank r reads samples [r*B/world, (r+1)*B/world) of every global
    batch (contiguous split, keras_api.Model._local_batch) and nothing else."""
    xs, ys = make_dataset(str(tmp_path), n=13)
    order = list(np.random.default_rng(5).permutation(len(xs)))
    exp = list(serial_batches(xs, ys, order, 4))
    opened = []
    real_load = np.load

    def spy(path, *a, **k):
        opened.append(str(path))
        return real_load(path, *a, **k)

    shards = []
    for r in range(2):
        ld = PrefetchLoader(xs, ys, 4, order=order, rank=r, world=2, pin=False, workers=1)
        assert len(ld) == 3 and ld.local_B == 2
        opened.clear()
        import unittest.mock as mock
        with mock.patch.object(np, "load", spy):
            shards.append([(x.numpy().copy(), {h: t.numpy().copy() for h, t in y.items()}) for x, y in ld])
        mine = {xs[i] for k in range(3) for i in order[k * 4 + r * 2:k * 4 + r * 2 + 2]}
        assert {p for p in opened if os.sep + "train" + os.sep in p} == mine          # no other rank's files were read
    for k, (ex, ey) in enumerate(exp):
        assert np.array_equal(np.concatenate([shards[0][k][0], shards[1][k][0]]), ex)
        for h in ey:
            assert np.array_equal(np.concatenate([shards[0][k][1][h], shards[1][k][1][h]]), ey[h])
    with pytest.raises(ValueError):
        PrefetchLoader(xs, ys, 5, rank=0, world=2, pin=False)
