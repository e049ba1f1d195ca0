"""CPU tests (no GPU): parameter layout vs the oracle, Keras<->flat layout round trip, CLI surface, dataset
pairing, the C-ABI library exporting every symbol include/rua_hip.h declares, and loud failure without a GPU."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import resuneta_ref as ref
from resunet_a_mltsk_keras_amd import _lib as L
from resunet_a_mltsk_keras_amd.engine import Engine, ModelConfig, stats_replicas
from resunet_a_mltsk_keras_amd.synthetic import make_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kw,count", [
    (dict(input_shape=(256, 256, 6), num_classes=6, multitasking=True), 42736869),
    (dict(input_shape=(256, 256, 6), num_classes=6, multitasking=False), 42690134),
    (dict(input_shape=(128, 128, 7), num_classes=2, multitasking=False), 42163914),
])
def test_param_layout_counts_and_keras_names(kw, count):
    ps = Engine.param_layout(ModelConfig(**kw))
    assert ps.count() == count
    params, order = ref.init_params(ref.RefConfig(**kw), 0)
    names = []
    for e in ps.entries:
        if e["name"] not in names:
            names.append(e["name"])
    assert names == order                                   # same layers, same Keras creation order
    assert all(e["off"] % 16 == 0 for e in ps.entries)      # 64-byte aligned slices of the flat buffer


def test_keras_layout_round_trip_and_concat_split():
    kw = dict(input_shape=(64, 64, 3), num_classes=4, multitasking=True)
    ps = Engine.param_layout(ModelConfig(**kw))
    params, _ = ref.init_params(ref.RefConfig(**kw), 3)
    kd = {k: v.numpy() for k, v in params.items()}
    P, S = ps.from_keras(kd)
    back = ps.to_keras(P, S)
    assert set(back) == set(kd) and all(np.array_equal(back[k], kd[k]) for k in kd)
    # a concatenating 1x1 conv is stored as one [tap][Cout][Cin_seg] block per source
    multi = [c for c in ps.convs.values() if len(c["segs"]) > 1]
    assert multi, "PSP / combine convs must be segmented"
    c = multi[0]
    e0 = next(e for e in ps.entries if e["off"] == c["segs"][0]["off"])
    k = kd[c["name"] + "/kernel"]
    blk = P[e0["off"]:e0["off"] + e0["size"]].reshape(e0["taps"], e0["cout"], e0["cin"])
    assert np.array_equal(blk[0], k[0, 0, :e0["cin"], :].T)


def test_engine_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.RuaError, match="no CPU fallback"):
        Engine(ModelConfig(input_shape=(64, 64, 3), num_classes=4))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rua_hip.h")).read()
    declared = set(re.findall(r"\b(rua_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rua_conv_seg", "rua_conv_desc", "rua_wgrad_desc", "rua_wprep_item"}
    assert len(declared) >= 35
    dll = ctypes.CDLL(L.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(dll, s)]
    assert not missing, missing
    assert set(L.EXPORTED_SYMBOLS) == declared, set(L.EXPORTED_SYMBOLS) ^ declared
    lib = L.lib()
    assert lib.raw("rua_version")() >= 200
    assert lib.raw("rua_stats_replicas")(5000) == stats_replicas(5000) == 32
    assert lib.raw("rua_stats_replicas")(10) == stats_replicas(10) == 2
    assert lib.raw("rua_stats_replicas")(8) == stats_replicas(8) == 1


def test_tuning_switches_are_the_only_global_state_and_no_environment_reads():
    """SURVEY 8b / VERDICT r1: the launchers keep no hidden state and never read the environment - heuristics change only
    through rua_set_tuning().  (No launch happens here: safe without a GPU.)"""
    lib = L.lib()
    keys = []
    while lib.dll.rua_tuning_key(len(keys)):
        keys.append(lib.dll.rua_tuning_key(len(keys)).decode())
    assert {"conv_halo", "conv_dmap", "wgrad_dmap", "dmap_target", "bn_grid", "conv_strip"} <= set(keys)
    assert lib.get_tuning("conv_halo") == 1 and lib.get_tuning("dmap_target") == 0          # 0 = derived from the CU count
    lib.set_tuning(conv_halo=0, conv_pw_minm=1 << 40)
    assert lib.get_tuning("conv_halo") == 0 and lib.get_tuning("conv_pw_minm") == 1 << 40
    lib.set_tuning(conv_halo=1, conv_pw_minm=65536)
    assert lib.dll.rua_set_tuning(b"no_such_key", 1) == -1 and b"unknown key" in lib.dll.rua_last_error()
    for f in os.listdir(os.path.join(ROOT, "resunet_a_mltsk_keras_amd", "csrc")):
        assert "getenv" not in open(os.path.join(ROOT, "resunet_a_mltsk_keras_amd", "csrc", f)).read(), f
    # the library exports no collective of its own (one data-parallel path: dist.py over torch.distributed / RCCL)
    assert not hasattr(lib.dll, "rua_allreduce_bucket") and not hasattr(lib.dll, "rua_comm_init")


def test_c_abi_struct_sizes_match_header():
    # rua_conv_seg: 2 pointers + 6 int32 ; rua_wprep_item: 2 int64 + 4 int32
    assert ctypes.sizeof(L.ConvSeg) == 40 and ctypes.sizeof(L.WprepItem) == 32
    assert L.ConvDesc.seg.offset == 0 and L.ConvDesc.nseg.offset == 240
    assert ctypes.sizeof(L.WgradDesc) % 8 == 0


def test_argument_validation_without_launch():
    """Precondition violations are rejected on the host before any launch (safe without a GPU)."""
    lib = L.lib()
    d = L.ConvDesc()
    assert lib.raw("rua_conv_fwd")(ctypes.byref(d), None) == -1
    assert b"nseg" in lib.dll.rua_last_error()
    d.nseg, d.dtype, d.Cout = 1, L.RUA_BF16, 12
    assert lib.raw("rua_conv_fwd")(ctypes.byref(d), None) == -1
    assert b"multiple of 8" in lib.dll.rua_last_error()
    assert lib.raw("rua_tanimoto_sums")(None, None, 1, 1, 9, None, None) == -1


def test_cli_surface_matches_reference_flags():
    sys.path.insert(0, ROOT)
    import train_ISPRS as cli
    a = cli.build_parser().parse_args([])
    ref_defaults = dict(resunet_a=False, multitasking=False, gpu_parallel=False, results_path="./results/results_run1",
                        checkpoint_path=None, dataset_path="./DATASETS/patch_size=256_stride=32", batch_size=4, learning_rate=1e-3,
                        loss="weighted_cross_entropy", optimizer="adam", num_classes=5, epochs=500, patch_size=256,
                        bound_weight=1.0, dist_weight=1.0, color_weight=1.0)
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k
    b = cli.build_parser().parse_args("-rp r -cp c -dp d -bs 8 -lr 0.01 -optm sgd -ps 128 --loss tanimoto --multitasking yes".split())
    assert (b.results_path, b.checkpoint_path, b.dataset_path, b.batch_size, b.optimizer, b.patch_size, b.multitasking) == \
        ("r", "c", "d", 8, "sgd", 128, True)
    assert cli.str2bool("T") is True and cli.str2bool("0") is False
    with pytest.raises(Exception):
        cli.str2bool("maybe")
    assert abs(cli.compute_mcc(50, 40, 5, 5) - (50 * 40 - 25) / np.sqrt(55 * 55 * 45 * 45)) < 1e-12


def test_dataset_listing_pairs_by_name(tmp_path):
    import train_ISPRS as cli
    x, y = make_batch(5, 16, 3, 4, True, seed=0, block=8)
    for h in ["seg", "bound", "dist", "color"]:
        os.makedirs(tmp_path / "labels" / h)
    os.makedirs(tmp_path / "train")
    for i in [3, 0, 4, 1, 2]:                                # write in scrambled order
        np.save(tmp_path / "train" / f"patch_{i}.npy", x[i])
        for h in y:
            np.save(tmp_path / "labels" / h / f"patch_{i}.npy", y[h][i])
    xs, ys = cli.list_dataset(str(tmp_path), True)
    assert [os.path.basename(p) for p in xs] == [os.path.basename(p) for p in ys["color"]]
    x_tr, y_tr, x_va, y_va = cli.split_dataset(xs, ys)
    assert len(x_tr) == 4 and len(x_va) == 1
    assert all(os.path.basename(a) == os.path.basename(b) for a, b in zip(x_tr, y_tr["dist"]))
    os.remove(tmp_path / "labels" / "bound" / "patch_2.npy")
    with pytest.raises(FileNotFoundError):
        cli.list_dataset(str(tmp_path), True)


def test_reference_config_values():
    from ResUnet_a.config import UnetConfig
    c = UnetConfig()
    assert (c.CLASSES_NUM, c.IMAGE_W, c.IMAGE_H, c.IMAGE_C, c.EPOCHS, c.batch_size) == (5, 512, 512, 3, 5000, 8)
    assert np.array_equal(c.MEAN, np.array([82, 92, 88], dtype=float))


def test_synthetic_batch_format():
    x, y = make_batch(3, 64, 6, 6, True, seed=1, block=16)
    assert x.dtype == np.float32 and x.shape == (3, 64, 64, 6) and 0 <= x.min() and x.max() < 1
    assert np.array_equal(y["seg"].sum(-1), np.ones((3, 64, 64), np.float32))
    assert set(np.unique(y["bound"])) <= {0.0, 1.0} and y["color"].shape[-1] == 3
    assert (y["seg"].sum(axis=(0, 1, 2)) > 0).sum() >= 4          # classes present


def test_goldens_pin_the_oracle():
    g = np.load(os.path.join(ROOT, "tests", "golden", "tiny_multitask.npz"))
    cfg = ref.RefConfig(input_shape=(64, 64, 6), num_classes=6, multitasking=True)
    params, order = ref.init_params(cfg, int(g["seed"]))
    lw = dict(zip(ref.HEADS, g["loss_weights"].tolist()))
    tr = ref.RefTrainer(cfg, params, order, ref.CompileSpec(loss="tanimoto", loss_weights=lw, lr=1e-3))
    y = {h: g["y_" + h] for h in ref.HEADS}
    res = tr.train_on_batch(g["x"], y)
    assert np.allclose(res[:5], g["losses"], rtol=2e-5, atol=1e-6)
    for h in ref.HEADS:
        assert np.allclose(tr.last_taps[h + "_logits"], g["logits_" + h], rtol=1e-4, atol=2e-5)
    for name, s in zip(g["grad_names"], g["grad_abs_sums"]):
        assert abs(float(np.abs(tr.last_grads[str(name)].numpy()).sum()) - s) < 1e-3 * s
    k = np.load(os.path.join(ROOT, "tests", "golden", "tanimoto_kat.npz"))
    nchw = lambda a: torch.from_numpy(a).permute(0, 3, 1, 2)
    assert np.allclose(ref.tanimoto_dual_loss(nchw(k["y_swap"]), nchw(k["p_swap"])).numpy(), k["l_swap"], rtol=1e-5)
    assert np.allclose(ref.tanimoto_dual_loss(nchw(k["y_rand"]), nchw(k["p_rand"])).numpy(), k["l_rand"], rtol=1e-5)


def test_keras_variable_names_are_matched_by_order_within_a_layer_type():
    """keras_api.canonical_keras_names: `conv2d_85/kernel:0` of a process that built other models first."""
    from resunet_a_mltsk_keras_amd.keras_api import canonical_keras_names
    want = {"conv2d/kernel": np.zeros((1, 1, 3, 8), np.float32), "conv2d/bias": np.zeros(8, np.float32),
            "conv2d_1/kernel": np.zeros((3, 3, 8, 8), np.float32), "batch_normalization/gamma": np.zeros(8, np.float32),
            "batch_normalization/moving_variance": np.ones(8, np.float32)}
    given = {"conv2d_84/kernel:0": np.full((1, 1, 3, 8), 1.0), "conv2d_84/bias:0": np.full(8, 2.0),
             "conv2d_85/kernel:0": np.full((3, 3, 8, 8), 3.0), "batch_normalization_9/gamma:0": np.full(8, 4.0),
             "batch_normalization_9/moving_variance:0": np.full(8, 5.0)}
    out = canonical_keras_names(given, want)
    assert set(out) == set(want) and out["conv2d_1/kernel"][0, 0, 0, 0] == 3.0 and out["batch_normalization/gamma"][0] == 4.0
    assert all(v.dtype == np.float32 for v in out.values())
    bad = dict(given); bad["conv2d_85/kernel:0"] = np.zeros((3, 3, 8, 4))
    with pytest.raises(ValueError, match="conv2d_85/kernel"):
        canonical_keras_names(bad, want)
    short = dict(given); del short["conv2d_84/bias:0"]
    with pytest.raises(ValueError, match="conv2d/bias"):
        canonical_keras_names(short, want)
    extra = dict(given); extra["dense/kernel:0"] = np.zeros(3)
    with pytest.raises(ValueError, match="dense/kernel"):
        canonical_keras_names(extra, want)
