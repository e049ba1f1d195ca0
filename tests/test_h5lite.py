"""HDF5 interop: the pure-Python reader / writer of Keras weight files against libhdf5.
  * tests/golden/keras_tiny.h5 was written by h5py in Keras' layout (tests/golden/make_keras_h5.py) - read it here;
  * files written here are read by h5py itself when an interpreter that has it exists (/opt/conda/bin/python3.9 in the
    build container; skipped elsewhere);
  * host side of the checkpoint format (no GPU): Keras-layout groups, parameter-layout matching, errors."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from resunet_a_mltsk_keras_amd import h5lite as h

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
H5PY_PYTHON = "/opt/conda/bin/python3.9"


def _generator():
    src = open(os.path.join(GOLD, "make_keras_h5.py")).read().replace("import h5py", "h5py = None")
    ns = {"__file__": os.path.join(GOLD, "make_keras_h5.py"), "__name__": "make_keras_h5"}
    exec(compile(src, "make_keras_h5", "exec"), ns)
    return ns


def test_reads_a_file_written_by_libhdf5_in_keras_layout():
    mk = _generator()
    g = h.read_h5(os.path.join(GOLD, "keras_tiny.h5"))
    assert g.attrs["backend"] in (b"tensorflow", "tensorflow") and "Functional" in str(g.attrs["model_config"])
    assert g.attrs["note_vlen"] == "variable-length UTF-8 string é"                        # global-heap (variable-length) string
    w = h.keras_weights_from_group(g)                                                      # descends into model_weights by itself
    exp = [(wn, shape) for _, ws in mk["LAYERS"] for wn, shape in ws]
    assert list(w) == [wn for wn, _ in exp]                                                # file order = Keras layer order
    for wn, shape in exp:
        assert w[wn].dtype == np.float32 and np.array_equal(w[wn], mk["fill"](wn, shape)), wn
    assert int(g["optimizer_weights/Adam/iter:0"]) == 12
    assert np.array_equal(g["optimizer_weights/Adam/conv2d/kernel/m:0"], mk["fill"]("m", (1, 1, 3, 32)))
    assert np.array_equal(g["extra/chunked"], mk["fill"]("chunked", (10, 7)))             # unfiltered chunked layout
    assert g["extra/f64"].dtype == np.float64


def _sample_tree():
    rng = np.random.default_rng(0)
    w = {}
    for i in range(150):                                      # > 128 links: more than one symbol-table node per group
        n = "conv2d" if i == 0 else f"conv2d_{i}"
        w[n + "/kernel:0"] = rng.standard_normal((3, 3, 4, 8)).astype(np.float32)
        w[n + "/bias:0"] = rng.standard_normal(8).astype(np.float32)
    root = h.Group()
    root.children["model_weights"] = h.keras_group_from_weights(w)
    root.attrs["rua_checkpoint"] = json.dumps({"format": "rua-checkpoint-2"}).encode()
    root.attrs["scalar"] = np.float32(1.5)
    root.attrs["vec"] = np.arange(5, dtype=np.int64)
    root.require_group("optimizer_weights").set("rua/iterations:0", np.array(7, np.int64))
    root["optimizer_weights"].set("rua/m:0", rng.standard_normal(1000).astype(np.float32))
    return root, w


def test_round_trip_and_errors(tmp_path):
    root, w = _sample_tree()
    p = str(tmp_path / "a.h5")
    h.write_h5(p, root)
    assert h.is_hdf5(p) and not h.is_hdf5(__file__)
    back = h.read_h5(p)
    w2 = h.keras_weights_from_group(back)
    assert list(w2) == list(w) and all(np.array_equal(w2[k], w[k]) for k in w)
    assert back.attrs["scalar"] == np.float32(1.5) and list(back.attrs["vec"]) == [0, 1, 2, 3, 4]
    assert int(back["optimizer_weights/rua/iterations:0"]) == 7 and "nope" not in back
    with pytest.raises(h.H5Error, match="not an HDF5 file"):
        h._Reader(b"PK\x03\x04" + b"\0" * 200)
    with pytest.raises(h.H5Error):
        h.read_h5(_truncate(p, tmp_path))
    with pytest.raises(h.H5Error, match="layer_names"):
        h.keras_weights_from_group(h.Group())


def _truncate(p, tmp_path):
    q = str(tmp_path / "short.h5")
    open(q, "wb").write(open(p, "rb").read()[:4000])
    return q


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py in this image")
def test_files_written_here_are_read_by_h5py(tmp_path):
    root, w = _sample_tree()
    p = str(tmp_path / "mine.h5")
    h.write_h5(p, root)
    code = (
        "import h5py, numpy as np, sys, json\n"
        "f = h5py.File(sys.argv[1], 'r')\n"
        "g = f['model_weights']\n"
        "names = [n.decode() for n in g.attrs['layer_names']]\n"
        "tot = 0.0; cnt = 0\n"
        "for n in names:\n"
        "    for wn in g[n].attrs['weight_names']:\n"
        "        a = g[n][wn.decode()][()]; tot += float(np.abs(a.astype(np.float64)).sum()); cnt += a.size\n"
        "print(json.dumps(dict(layers=len(names), first=names[:2], tot=tot, cnt=cnt, t=int(f['optimizer_weights/rua/iterations:0'][()]),\n"
        "                      meta=f.attrs['rua_checkpoint'].decode() if isinstance(f.attrs['rua_checkpoint'], bytes) else str(f.attrs['rua_checkpoint']),\n"
        "                      scalar=float(f.attrs['scalar']), kshape=list(g['conv2d_7']['conv2d_7/kernel:0'].shape))))\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    out = subprocess.run([H5PY_PYTHON, "-c", code, p], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["layers"] == 150 and r["first"] == ["conv2d", "conv2d_1"] and r["t"] == 7 and r["scalar"] == 1.5
    assert r["kshape"] == [3, 3, 4, 8] and r["cnt"] == sum(v.size for v in w.values())
    assert abs(r["tot"] - sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values())) < 1e-6 * r["tot"]
    assert json.loads(r["meta"])["format"] == "rua-checkpoint-2"


def test_keras_file_configuration_is_read_off_the_variables():
    """A file Keras wrote carries no metadata of ours: graph variant, width, depth, classes and heads come from the
    variable shapes (host logic only: the parameter layout needs no device)."""
    from resunet_a_mltsk_keras_amd.engine import Engine, ModelConfig
    from resunet_a_mltsk_keras_amd.keras_api import _cfg_from_keras_file
    for variant, mt, ncls in (("model2", True, 5), ("model", False, 3)):
        cfg = ModelConfig((64, 64, 6), ncls, mt, variant)
        ps = Engine.param_layout(cfg)
        P, S = ps.init_host(1)
        w = {k + ":0": v for k, v in ps.to_keras(P, S).items()}
        shifted = {}                                          # Keras numbers layers per process: shift every index
        for k, v in w.items():
            layer, var = k.split("/")
            base, _, idx = layer.rpartition("_")
            if base and idx.isdigit():
                layer = f"{base}_{int(idx) + 11}"
            elif layer in ("conv2d", "batch_normalization"):
                layer = f"{layer}_11"
            shifted[f"{layer}/{var}"] = v
        got = _cfg_from_keras_file(h.Group(), shifted, (64, 64, 6))
        assert (got.variant, got.multitasking, got.num_classes, got.width, got.depth, got.input_shape) == (variant, mt, ncls, 32, 6, (64, 64, 6))
        with pytest.raises(ValueError, match="patch size"):
            _cfg_from_keras_file(h.Group(), shifted, None)
        root = h.Group({"model_config": json.dumps({"config": {"layers": [{"class_name": "InputLayer", "config": {"batch_input_shape": [None, 64, 64, 6]}}]}}).encode()})
        assert _cfg_from_keras_file(root, shifted, None).input_shape == (64, 64, 6)
        with pytest.raises(ValueError, match="fit neither"):                      # the PSP branch count depends on the input width (model2.py:49-53)
            _cfg_from_keras_file(h.Group(), shifted, (256, 256, 6))


# -- Keras' layer order (a topological load_weights on the TensorFlow side zips layer_names with model.layers) ------
def test_keras_graph_names_equal_the_parameter_store_names_in_creation_order():
    from resunet_a_mltsk_keras_amd import keras_graph
    from resunet_a_mltsk_keras_amd.engine import Engine, ModelConfig
    for shape, mt, variant, depth in (((256, 256, 6), True, "model2", 6), ((128, 128, 7), False, "model2", 6), ((64, 64, 3), True, "model", 6),
                                      ((512, 512, 6), True, "model2", 7), ((256, 256, 3), False, "model", 6)):
        ps = Engine.param_layout(ModelConfig(shape, 6, mt, variant, 32, depth))
        mine = []
        for e in ps.entries:
            if e["layer"] not in mine:
                mine.append(e["layer"])
        kg = keras_graph.build(shape[1], mt, variant, depth)
        assert kg.names_in_creation_order(weighted_only=True) == mine, (shape, mt, variant, depth)
        order = kg.names_in_model_order(weighted_only=True)
        assert sorted(order) == sorted(mine) and order != mine          # same layers, Keras lists them by depth


def test_keras_model_layers_order_of_a_resblock_is_by_depth_then_first_reach():
    """Hand-derived from functional.py::_map_graph_network on model2.py:15-34 with two branches: depth decreases along a
    branch, both branches share depths, and inside one depth the layer reached first by the depth-first walk from the Add
    (inputs in list order [x, branch0, branch1]) comes first."""
    from resunet_a_mltsk_keras_amd.keras_graph import KerasGraph
    g = KerasGraph()
    x = g.add("conv", [g.add("input", [])])
    outs = [x]
    for _ in range(2):
        t = g.add("bn", [x]); t = g.add("act", [t]); t = g.add("conv", [t]); t = g.add("bn", [t]); t = g.add("act", [t]); t = g.add("conv", [t])
        outs.append(t)
    g.outputs = {"out": g.add("add", outs)}
    assert g.names_in_model_order() == [
        "input_1", "conv2d",
        "batch_normalization", "batch_normalization_2", "activation", "activation_2", "conv2d_1", "conv2d_3",
        "batch_normalization_1", "batch_normalization_3", "activation_1", "activation_3", "conv2d_2", "conv2d_4", "add"]


def test_exported_layer_names_follow_keras_depth_order_and_round_trip(tmp_path):
    from resunet_a_mltsk_keras_amd import keras_graph
    from resunet_a_mltsk_keras_amd.engine import Engine, ModelConfig
    cfg = ModelConfig((64, 64, 3), 4, True, "model2", 32, 6)
    ps = Engine.param_layout(cfg)
    rng = np.random.default_rng(3)                                   # the layout is what is tested: small stand-in arrays under the real names
    w = {e["name"]: rng.random(3).astype(np.float32) for e in ps.entries}
    w.update({s_["name"]: rng.random(2).astype(np.float32) for s_ in ps.state})
    order = keras_graph.weighted_layer_order(64, True, "model2", 6)
    path = str(tmp_path / "w.h5")
    h.write_h5(path, h.keras_group_from_weights(w, order))
    root = h.read_h5(path)
    names = [n.decode() if isinstance(n, bytes) else str(n) for n in root.attrs["layer_names"]]
    assert names == order
    # the four first BatchNorms of the top ResBlock come before any of its convolutions (they share a depth)
    assert names[:6] == ["conv2d", "batch_normalization", "batch_normalization_2", "batch_normalization_4", "batch_normalization_6", "conv2d_1"]
    back = h.keras_weights_from_group(root)
    assert set(back) == {k + ":0" for k in w} and all(np.array_equal(back[k + ":0"], v) for k, v in w.items())
    with pytest.raises(h.H5Error):
        h.keras_group_from_weights(w, order[:-1])


def test_attribute_over_64k_is_refused_by_name_not_read_as_empty():
    """tests/golden/dense_attr.h5 (make_dense_attr_h5.py): libhdf5 stores an attribute over 64 KiB only in the new file format;
    the reader must name what it cannot read instead of returning a root without attributes; garbage raises H5Error too."""
    with pytest.raises(h.H5Error, match="superblock version|dense attribute"):
        h.read_h5(os.path.join(GOLD, "dense_attr.h5"))
    assert h.is_hdf5(os.path.join(GOLD, "dense_attr.h5"))
