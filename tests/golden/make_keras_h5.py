#!/opt/conda/bin/python3.9
"""Generates tests/golden/keras_tiny.h5 with h5py (libhdf5) in exactly the layout Keras 2.x `model.save()` /
`save_weights()` writes (keras/saving/hdf5_format.py: save_weights_to_hdf5_group, save_optimizer_weights_to_hdf5_group):
root attributes keras_version / backend / model_config / training_config, group `model_weights` with the `layer_names`
attribute (weightless layers included), one group per layer with `weight_names` and the variables under their full names
(`conv2d/kernel:0` => nested group `conv2d`), group `optimizer_weights`.  The runtime image has no h5py; the build container
has one for Python 3.9 under /opt/conda.  Values are closed-form (fill(name, shape)) so the test needs no second copy.

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py
"""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def fill(name, shape):
    n = int(np.prod(shape))
    seed = sum(ord(c) for c in name) % 97
    return ((np.arange(n, dtype=np.float64) * 0.37 + seed) % 5.0 - 2.5).astype(np.float32).reshape(shape)


LAYERS = [
    ("input_1", []),
    ("conv2d", [("conv2d/kernel:0", (1, 1, 3, 32)), ("conv2d/bias:0", (32,))]),
    ("batch_normalization", [("batch_normalization/gamma:0", (32,)), ("batch_normalization/beta:0", (32,)),
                             ("batch_normalization/moving_mean:0", (32,)), ("batch_normalization/moving_variance:0", (32,))]),
    ("activation", []),
    ("conv2d_1", [("conv2d_1/kernel:0", (3, 3, 32, 32)), ("conv2d_1/bias:0", (32,))]),
    ("seg3", [("seg3/kernel:0", (1, 1, 32, 4)), ("seg3/bias:0", (4,))]),
]


def main():
    path = os.path.join(HERE, "keras_tiny.h5")
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = b"2.4.0"
        f.attrs["backend"] = b"tensorflow"
        f.attrs["model_config"] = json.dumps({"class_name": "Functional", "config": {"name": "model"}}).encode("utf8")
        f.attrs["training_config"] = json.dumps({"optimizer_config": {"class_name": "Adam", "config": {"learning_rate": 1e-3}}}).encode("utf8")
        f.attrs["note_vlen"] = "variable-length UTF-8 string é"           # what newer h5py / Keras write for str values
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = [n.encode("utf8") for n, _ in LAYERS]
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = b"2.4.0"
        for name, ws in LAYERS:
            lg = g.create_group(name)
            lg.attrs["weight_names"] = [w.encode("utf8") for w, _ in ws]
            for w, shape in ws:
                d = lg.create_dataset(w, shape, dtype="float32")
                d[...] = fill(w, shape)
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = [b"Adam/iter:0", b"Adam/conv2d/kernel/m:0"]
        it = og.create_dataset("Adam/iter:0", (), dtype="int64")
        it[()] = 12
        og.create_dataset("Adam/conv2d/kernel/m:0", data=fill("m", (1, 1, 3, 32)))
        # not Keras, but legal HDF5 a converter may produce: a chunked (uncompressed) and a compact dataset
        f.create_dataset("extra/chunked", data=fill("chunked", (10, 7)), chunks=(4, 3))
        f.create_dataset("extra/f64", data=fill("f64", (5,)).astype(np.float64))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
