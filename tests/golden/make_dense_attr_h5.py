"""tests/golden/dense_attr.h5: an HDF5 file whose root carries an attribute over 64 KiB (a stand-in for a `model_config` grown
past 64 KiB).  Measured with libhdf5 1.12 / h5py 3.3 while writing this: with the default or 'earliest' format bounds -
what Keras' `model.save` uses, and the only object-header format h5lite reads - libhdf5 REFUSES such an attribute ("object header
message is too large": Keras' own save fails there); dense attribute storage (Attribute Info 0x0015 + fractal heap) only exists with
libver='latest', whose version-2/3 superblock h5lite already rejects by name.  This file is that case; tests/test_h5lite.py requires
the H5Error.      /opt/conda/bin/python3.9 tests/golden/make_dense_attr_h5.py"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for bounds in ("earliest", None, ("earliest", "latest")):
    try:
        with h5py.File(os.path.join(HERE, "dense_attr.h5"), "w", libver=bounds) as f:
            f.attrs["model_config"] = np.bytes_(b"x" * 70000)
        raise SystemExit(f"libver={bounds!r} accepted a 70 kB attribute: the statement in the docstring no longer holds")
    except RuntimeError as exc:
        print(f"libver={bounds!r}: {exc}")
with h5py.File(os.path.join(HERE, "dense_attr.h5"), "w", libver="latest") as f:
    f.attrs["keras_version"] = b"2.4.0"
    f.attrs["model_config"] = np.bytes_(b'{"class_name": "Functional", "pad": "' + b"x" * 70000 + b'"}')
    g = f.create_group("model_weights")
    g.attrs["layer_names"] = np.array([b"conv2d"])
    c = g.create_group("conv2d")
    c.attrs["weight_names"] = np.array([b"conv2d/kernel:0"])
    c.create_dataset("conv2d/kernel:0", data=np.zeros((1, 1, 3, 4), np.float32))
print("wrote dense_attr.h5")
