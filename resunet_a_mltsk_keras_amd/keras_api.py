"""Drop-in Python surface of the reference's training path, backed by the HIP engine.

Mirrors exactly what the reference's callers touch:
  * `Resunet_a(input_shape, num_classes, args, inputs=None).model`      ResUnet_a/model.py:6-12, model2.py:6-12
  * Keras-Model duck type: summary / compile / train_on_batch / test_on_batch / predict / save / output_names /
    optimizer.lr                                                          train_ISPRS.py:95,131,148,167,186,292,445-461,478-480
  * `Adam(lr=, beta_1=)`, `SGD(lr=, momentum=)`, `K.get_value/set_value`, `load_model`          utils.py imports, train_ISPRS.py:404-407,474
  * loss factories `Tanimoto_dual_loss()` and `weighted_categorical_crossentropy(weights)`     multitasking_utils.py:71-85, utils.py:466-491
x / y cross this boundary as host numpy float32 NHWC arrays (dict of arrays for the multitask heads), like in
the reference.  Everything numerical runs in librua_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from .engine import HEADS, Engine, LossSpec, ModelConfig


# ---- optimizers / backend shims -------------------------------------------------------------------
class _Var:
    """Stands in for a Keras backend variable (optimizer.lr)."""

    def __init__(self, v):
        self.value = float(v)

    def numpy(self):
        return self.value

    def __float__(self):
        return self.value


class Adam:
    def __init__(self, lr=None, learning_rate=None, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **_):
        v = learning_rate if learning_rate is not None else (lr if lr is not None else 1e-3)
        self.lr = _Var(v)
        self.learning_rate = self.lr
        self.beta_1, self.beta_2, self.epsilon, self.kind = beta_1, beta_2, epsilon, "adam"
        if abs(epsilon - 1e-7) > 1e-12:
            raise ValueError("the HIP Adam kernel is built with Keras' default epsilon 1e-7")


class SGD:
    def __init__(self, lr=None, learning_rate=None, momentum=0.0, nesterov=False, **_):
        v = learning_rate if learning_rate is not None else (lr if lr is not None else 1e-2)
        self.lr = _Var(v)
        self.learning_rate = self.lr
        self.momentum, self.kind = momentum, "sgd"
        if nesterov:
            raise ValueError("Nesterov momentum is not part of the reference path")


class _Backend:
    """`K` as train_ISPRS.py:478-480 uses it."""

    @staticmethod
    def get_value(v):
        return float(v.value) if isinstance(v, _Var) else v

    @staticmethod
    def set_value(v, x):
        v.value = float(x)

    @staticmethod
    def epsilon():
        return 1e-7


K = _Backend()


# ---- losses ------------------------------------------------------------------------------------------
def _dev_f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


class TanimotoDualLoss:
    """Returned by Tanimoto_dual_loss(): a tag `compile` recognises; calling it evaluates the (B,) loss vector
    of multitasking_utils.py:71-85 on the GPU (rua_tanimoto_sums / rua_tanimoto_finalize)."""
    kind = L.LOSS_TANIMOTO
    __name__ = "loss"

    def __call__(self, label, pred):
        label, pred = np.asarray(label, np.float32), np.asarray(pred, np.float32)
        B, Cc = label.shape[0], label.shape[-1]
        HW = int(np.prod(label.shape[1:-1]))
        lib = L.lib()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p, y = _dev_f32(pred), _dev_f32(label)
        sums = torch.zeros(B * Cc * 6, dtype=torch.float64, device="cuda")
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        per = torch.zeros(B, dtype=torch.float32, device="cuda")
        lib.call("rua_tanimoto_sums", p.data_ptr(), y.data_ptr(), B, HW, Cc, sums.data_ptr(), s)
        lib.call("rua_tanimoto_finalize", sums.data_ptr(), B, HW, Cc, 1.0, out.data_ptr(), None, per.data_ptr(), s)
        return per.cpu().numpy()


def Tanimoto_dual_loss():
    return TanimotoDualLoss()


def Tanimoto_loss(label, pred):
    """multitasking_utils.py:38-68: the (B,) Tanimoto ratio with class weights 1/V^2 from the volumes of `label`
    (inf -> largest finite weight), on the GPU: the moments kernel of the training path with the arguments in the
    reference's order, finished by rua_tanimoto_ratio."""
    label, pred = np.asarray(label, np.float32), np.asarray(pred, np.float32)
    if label.shape != pred.shape:
        raise ValueError(f"label {label.shape} and pred {pred.shape} differ")
    B, Cc = label.shape[0], label.shape[-1]
    HW = int(np.prod(label.shape[1:-1]))
    lib = L.lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a, b = _dev_f32(label), _dev_f32(pred)
    sums = torch.zeros(B * Cc * 6, dtype=torch.float64, device="cuda")
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    per = torch.zeros(B, dtype=torch.float32, device="cuda")
    lib.call("rua_tanimoto_sums", a.data_ptr(), b.data_ptr(), B, HW, Cc, sums.data_ptr(), s)      # p := label, y := pred
    lib.call("rua_tanimoto_ratio", sums.data_ptr(), B, Cc, out.data_ptr(), per.data_ptr(), s)
    return per.cpu().numpy()


class WeightedCategoricalCrossentropy:
    """Returned by weighted_categorical_crossentropy(weights) (utils.py:466-491); calling it gives the (B,H,W) map."""
    kind = L.LOSS_WCE
    __name__ = "loss"

    def __init__(self, weights):
        self.weights = [float(w) for w in np.asarray(weights).reshape(-1)]

    def __call__(self, y_true, y_pred):
        y_true, y_pred = np.asarray(y_true, np.float32), np.asarray(y_pred, np.float32)
        Cc = y_true.shape[-1]
        if Cc != len(self.weights):
            raise ValueError(f"{len(self.weights)} class weights for {Cc} classes")
        M = int(np.prod(y_true.shape[:-1]))
        lib = L.lib()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p, y, w = _dev_f32(y_pred), _dev_f32(y_true), _dev_f32(self.weights + [0.0] * 8)
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        per = torch.zeros(M, dtype=torch.float32, device="cuda")
        lib.call("rua_pixel_loss", L.LOSS_WCE, p.data_ptr(), None, y.data_ptr(), w.data_ptr(), M, Cc, out.data_ptr(), per.data_ptr(), s)
        return per.cpu().numpy().reshape(y_true.shape[:-1])


def weighted_categorical_crossentropy(weights):
    return WeightedCategoricalCrossentropy(weights)


class CategoricalCrossentropy:      # tf.keras.losses.* as train_ISPRS.py:414-416,427-428 instantiates them
    kind = L.LOSS_CE_LOGITS


class BinaryCrossentropy:
    kind = L.LOSS_BCE_LOGITS


class MeanSquaredError:
    kind = L.LOSS_MSE


_LOSS_NAMES = {"categorical_crossentropy": L.LOSS_CE_LOGITS, "binary_crossentropy": L.LOSS_BCE_LOGITS,
               "mse": L.LOSS_MSE, "mean_squared_error": L.LOSS_MSE, "tanimoto": L.LOSS_TANIMOTO}


def _loss_kind(loss, head):
    if isinstance(loss, str):
        if loss not in _LOSS_NAMES:
            raise ValueError(f"unknown loss '{loss}'")
        return _LOSS_NAMES[loss], None
    kind = getattr(loss, "kind", None)
    if kind is None:
        raise TypeError(f"loss for output '{head}' must come from this package's factories (got {type(loss).__name__}); "
                        "arbitrary Python loss callables cannot run on the HIP path")
    if kind == L.LOSS_CE_LOGITS and head in ("bound", "color"):
        raise ValueError("categorical cross-entropy needs a softmax head")
    if kind == L.LOSS_BCE_LOGITS and head in ("seg", "dist"):
        raise ValueError("binary cross-entropy needs a sigmoid head")
    return kind, getattr(loss, "weights", None)


# ---- the model ---------------------------------------------------------------------------------------
class Model:
    """Keras-Model duck type over the recorded HIP plan."""

    def __init__(self, cfg: ModelConfig, dtype: str = "bf16", seed: int = 0):
        self.cfg = cfg
        self.engine = Engine(cfg, dtype=dtype, seed=seed)
        self.optimizer = None
        self.output_names = list(HEADS) if cfg.multitasking else ["softmax"]
        self.metrics_names: List[str] = []
        self._compiled = False

    # -- Keras surface ---------------------------------------------------------------------------------
    def summary(self, print_fn=print):
        ps = self.engine.params
        print_fn(f'Model: "resunet_a_{self.cfg.variant}"  (HIP/gfx950 plan, activations {self.engine.dtype})')
        print_fn("_" * 78)
        print_fn(f"{'Layer (Keras name)':38s}{'Param shape':26s}{'Param #':>12s}")
        print_fn("=" * 78)
        shown = {}
        for e in ps.entries:
            if e["kind"] == "kernel":
                k = int(round(e["taps"] ** 0.5))
                shown.setdefault(e["name"], [(k, k, e["cin_total"], e["cout"]), 0])
                shown[e["name"]][1] += e["size"]
            else:
                shown[e["name"]] = [(e["size"],), e["size"]]
        for name, (shape, n) in shown.items():
            print_fn(f"{name:38s}{str(shape):26s}{n:12d}")
        train = sum(e["size"] for e in ps.entries)
        total = ps.count()
        print_fn("=" * 78)
        print_fn(f"Total params: {total:,}\nTrainable params: {train:,}\nNon-trainable params: {total - train:,}")

    def count_params(self):
        return self.engine.count_params()

    def compile(self, optimizer=None, loss=None, loss_weights=None, metrics=None, **_):
        if optimizer is None or isinstance(optimizer, str):
            optimizer = {"adam": Adam, "sgd": SGD}[optimizer or "adam"]()
        self.optimizer = optimizer
        heads = HEADS if self.cfg.multitasking else ["seg"]
        kinds, cw = {}, None
        for h in heads:
            l = loss[h] if isinstance(loss, dict) else loss
            kinds[h], w = _loss_kind(l, h)
            if w is not None:
                if len(w) != self.cfg.num_classes:
                    raise ValueError(f"{len(w)} class weights for {self.cfg.num_classes} classes")
                cw = w
        weights = {h: 1.0 for h in heads}
        if loss_weights:
            weights.update({h: float(loss_weights[h]) for h in heads if h in loss_weights})
        spec = LossSpec(kind=kinds, weight=weights, class_weights=cw, optimizer=optimizer.kind, lr=optimizer.lr.value,
                        beta_1=getattr(optimizer, "beta_1", 0.9), beta_2=getattr(optimizer, "beta_2", 0.999),
                        momentum=getattr(optimizer, "momentum", 0.0))
        self.engine.compile(spec)
        self._finish_compile()

    def _finish_compile(self, restored_optimizer_state: bool = False):
        """Shared tail of compile() and load_model(compile=True): attach data parallel when the process is one rank of a
        torch.distributed job (MirroredStrategy scope of train_ISPRS.py:347,432), name the returned metrics."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .dist import DataParallel
            dp = DataParallel(self.engine)                   # broadcasts rank 0's weights and BN state
            if restored_optimizer_state:
                dp.broadcast_optimizer_state()               # ... and rank 0's Adam moments / step count on resume
        m = ["accuracy", "true_positives", "false_positives", "true_negatives", "false_negatives"]
        if self.cfg.multitasking:
            self.metrics_names = ["loss"] + [h + "_loss" for h in HEADS] + ["seg_" + x for x in m]
        else:
            self.metrics_names = ["loss"] + m
        self._compiled = True

    def _sync_lr(self):
        self.engine.loss.lr = self.optimizer.lr.value          # K.set_value(model.optimizer.lr, ...) takes effect

    def _local_batch(self, x, y, local_shard=False):
        """Under DP `-bs` is the GLOBAL batch (train_ISPRS.py:314,347): each rank takes its contiguous shard.
        local_shard=True: the caller already holds only this rank's shard (loader.PrefetchLoader(rank=, world=))."""
        w = self.engine.world
        if w == 1 or local_shard:
            return x, y
        import torch.distributed as dist
        r, B = dist.get_rank(), x.shape[0]
        if B % w:
            raise ValueError(f"global batch {B} not divisible by {w} replicas")
        sl = slice(r * (B // w), (r + 1) * (B // w))
        return x[sl], ({k: v[sl] for k, v in y.items()} if isinstance(y, dict) else (None if y is None else y[sl]))

    def train_on_batch(self, x, y=None, return_dict=False, local_shard=False, **_):
        assert self._compiled, "compile() first"
        self._sync_lr()
        x, y = self._local_batch(x, y, local_shard)
        res = self.engine.train_step(x, y)
        return dict(zip(self.metrics_names, res)) if return_dict else res

    def test_on_batch(self, x, y=None, return_dict=False, local_shard=False, **_):
        assert self._compiled, "compile() first"
        x, y = self._local_batch(x, y, local_shard)
        res = self.engine.test_step(x, y)
        return dict(zip(self.metrics_names, res)) if return_dict else res

    def predict(self, x, batch_size=1, **_):
        x = np.asarray(x, np.float32)
        if self.engine.loss is None:                      # load_model(..., compile=False) then predict (test_ISPRS.py:278,28)
            self.engine.compile(LossSpec(kind={h: L.LOSS_TANIMOTO for h in HEADS}, weight={h: 1.0 for h in HEADS}))
        outs = [self.engine.predict(x[i:i + batch_size]) for i in range(0, x.shape[0], batch_size)]
        if self.cfg.multitasking:
            return {h: np.concatenate([o[h] for o in outs], axis=0) for h in HEADS}
        return np.concatenate(outs, axis=0)

    # -- checkpoints (train_ISPRS.py:292,474-480): real HDF5 in the Keras layout, written / read by h5lite (no h5py needed) --
    def save(self, path):
        """`model.save('best_model.h5')` (train_ISPRS.py:292): an HDF5 file whose `model_weights` group is exactly what Keras'
        save_weights writes (layer_names / weight_names attributes, HWIO kernels, BN gamma, beta, moving_mean,
        moving_variance) - so Keras' `load_weights` reads it - plus this package's own metadata (root attribute
        `rua_checkpoint`: graph configuration, compile arguments, step count) and optimizer moments (`optimizer_weights`)."""
        from . import h5lite
        e = self.engine
        meta = dict(format="rua-checkpoint-2", cfg=dict(input_shape=list(self.cfg.input_shape), num_classes=self.cfg.num_classes,
                                                        multitasking=self.cfg.multitasking, variant=self.cfg.variant, width=self.cfg.width,
                                                        depth=self.cfg.depth), dtype=e.dtype, t=e.t)
        root = h5lite.Group()
        if e.loss is not None:
            sp = e.loss
            meta["loss"] = dict(kind=sp.kind, weight=sp.weight, class_weights=sp.class_weights, optimizer=sp.optimizer, lr=sp.lr,
                                beta_1=sp.beta_1, beta_2=sp.beta_2, momentum=sp.momentum)
            og = root.require_group("optimizer_weights")
            og.set("rua/m:0", e.M1.cpu().numpy()); og.set("rua/v:0", e.V1.cpu().numpy())
            og.set("rua/iterations:0", np.array(e.t, np.int64))
            og.attrs["weight_names"] = np.array([b"rua/iterations:0", b"rua/m:0", b"rua/v:0"])
        root.attrs["keras_version"] = b"2.4.0"
        root.attrs["backend"] = b"tensorflow"
        root.attrs["rua_checkpoint"] = json.dumps(meta).encode("utf8")
        root.children["model_weights"] = h5lite.keras_group_from_weights(e.get_weights(), self._keras_layer_order())
        h5lite.write_h5(path, root)

    def _keras_layer_order(self):
        """The weighted layers in the order of Keras' `model.layers` (graph depth, keras_graph.py): what a topological
        `load_weights` on the TensorFlow side zips the file's `layer_names` with."""
        from . import keras_graph
        c = self.cfg
        return keras_graph.weighted_layer_order(c.input_shape[1], c.multitasking, c.variant, c.depth)

    def get_weights_dict(self) -> Dict[str, np.ndarray]:
        return self.engine.get_weights()

    def set_weights_dict(self, w: Dict[str, np.ndarray]):
        self.engine.set_weights(w)

    # -- weight exchange with the reference: Keras' own HDF5 layout (`model.save_weights('w.h5')` on the
    #    TensorFlow side, or the `model_weights` group of a full `model.save`), kernels HWIO, BN gamma / beta / moving_mean /
    #    moving_variance, layers matched by order within a type (canonical_keras_names).  `.npz` (Keras variable name ->
    #    array) stays as a second carrier.
    def save_weights(self, path):
        if str(path).endswith(".npz"):
            np.savez(path, **{k + ":0": v for k, v in self.engine.get_weights().items()})
            return
        from . import h5lite
        h5lite.write_h5(path, h5lite.keras_group_from_weights(self.engine.get_weights(), self._keras_layer_order()))

    def load_weights(self, path):
        if str(path).endswith(".npz"):
            with np.load(path) as z:
                given = {k: z[k] for k in z.files}
        else:
            from . import h5lite
            given = h5lite.keras_weights_from_group(h5lite.read_h5(path))
        self.engine.set_weights(canonical_keras_names(given, self.engine.get_weights()))


def canonical_keras_names(given: Dict[str, np.ndarray], want: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Maps a dict of Keras variables onto this model's names.  Keras numbers layers per process (`conv2d_85/kernel:0`
    if other models were built before), so layers are matched by their ORDER within a layer type, not by the absolute
    index; a `:0` suffix is dropped.  Any missing / extra variable or shape mismatch is an error naming the variable."""
    def split(name):
        layer, _, var = name.split(":")[0].rpartition("/")
        base, _, idx = layer.rpartition("_")
        return (base, int(idx), var) if base and idx.isdigit() else (layer, 0, var)

    def ranked(names):
        seen: Dict[str, list] = {}
        for n in names:
            b, i, _ = split(n)
            if i not in seen.setdefault(b, []):
                seen[b].append(i)
        rank = {b: {i: r for r, i in enumerate(sorted(v))} for b, v in seen.items()}
        return {(split(n)[0], rank[split(n)[0]][split(n)[1]], split(n)[2]): n for n in names}

    g, w = ranked(given), ranked(want)
    missing = [w[k] for k in w if k not in g]
    extra = [g[k] for k in g if k not in w]
    if missing or extra:
        raise ValueError(f"weights do not match the model: missing {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                         f"unexpected {extra[:4]}{'...' if len(extra) > 4 else ''}")
    out = {}
    for k, name in w.items():
        a = np.asarray(given[g[k]], np.float32)
        if a.shape != want[name].shape:
            raise ValueError(f"{g[k]}: shape {a.shape}, the model's {name} is {want[name].shape}")
        out[name] = a
    return out


def _cfg_from_keras_file(root, weights, input_shape):
    """Graph configuration of a file Keras itself wrote (no `rua_checkpoint` attribute): channels / width / classes / heads /
    depth from the variable shapes (file order = Keras layer order), the graph variant (model2.py or model.py) by which
    parameter layout the variables fit, the patch size from `model_config` (InputLayer batch_input_shape) or the caller."""
    kernels = [(k, v) for k, v in weights.items() if k.split(":")[0].endswith("/kernel") and v.ndim == 4]
    if not kernels:
        raise ValueError("no convolution kernels in the file")
    stem = kernels[0][1]
    cin, width = int(stem.shape[2]), int(stem.shape[3])
    names = {k.split(":")[0] for k in weights}
    multitask = any(n.startswith(("seg3", "color")) for n in names)             # the multitask heads carry explicit layer names (model2.py:153-188)
    ncls = int(next(v for k, v in kernels if k.startswith("seg3")).shape[-1]) if multitask else int(kernels[-1][1].shape[-1])
    depth = 7 if max(int(v.shape[-1]) for _, v in kernels) >= width * 64 else 6
    if input_shape is None:
        cfgs = root.attrs.get("model_config")
        try:
            mc = json.loads(cfgs.decode("utf8") if isinstance(cfgs, bytes) else cfgs)
            shp = next(l["config"]["batch_input_shape"] for l in mc["config"]["layers"] if l["class_name"] == "InputLayer")
            input_shape = (int(shp[1]), int(shp[2]), cin)
        except Exception:
            raise ValueError("this Keras file does not say the patch size (no usable model_config): pass load_model(path, input_shape=(H, W, C))") from None
    err = None
    for variant in ("model2", "model"):
        cfg = ModelConfig(tuple(input_shape), ncls, multitask, variant, width, depth)
        ps = Engine.param_layout(cfg)
        want = ps.to_keras(np.zeros(ps.n, np.float32), np.zeros(ps.ns, np.float32))
        try:
            canonical_keras_names(weights, want)
            return cfg
        except ValueError as exc:
            err = exc
    raise ValueError(f"the variables fit neither ResUnet_a/model2.py nor ResUnet_a/model.py ({err})")


def load_model(path, compile=True, custom_objects=None, dtype=None, input_shape=None, **_):
    """`load_model('best_model.h5')` (train_ISPRS.py:474, test_ISPRS.py:278).  Reads HDF5 only - nothing is unpickled, so a
    checkpoint cannot run code.  A file written by Model.save() restores graph, compile arguments and optimizer state; a file
    written by Keras itself (`model.save` / `save_weights` of the reference) is loaded as weights into a model whose
    configuration is read off the variable shapes (compile it yourself afterwards)."""
    from . import h5lite
    if not h5lite.is_hdf5(path):
        raise ValueError(f"{path} is not an HDF5 file (checkpoints of this package and of Keras are HDF5; nothing else is read)")
    root = h5lite.read_h5(path)
    weights = h5lite.keras_weights_from_group(root)
    meta_raw = root.attrs.get("rua_checkpoint")
    if meta_raw is None:
        m = Model(_cfg_from_keras_file(root, weights, input_shape), dtype=dtype or "bf16")
        m.engine.set_weights(canonical_keras_names(weights, m.engine.get_weights()))
        return m
    meta = json.loads(meta_raw.decode("utf8") if isinstance(meta_raw, bytes) else meta_raw)
    if meta.get("format") != "rua-checkpoint-2":
        raise ValueError(f"{path}: unknown checkpoint format {meta.get('format')!r}")
    c = meta["cfg"]
    cfg = ModelConfig(tuple(c["input_shape"]), c["num_classes"], c["multitasking"], c["variant"], c["width"], c["depth"])
    m = Model(cfg, dtype=dtype or meta["dtype"])
    m.engine.set_weights(canonical_keras_names(weights, m.engine.get_weights()))
    if compile and "loss" in meta:
        lo = meta["loss"]
        opt = Adam(lr=lo["lr"], beta_1=lo["beta_1"], beta_2=lo["beta_2"]) if lo["optimizer"] == "adam" else SGD(lr=lo["lr"], momentum=lo["momentum"])
        m.optimizer = opt
        m.engine.compile(LossSpec(kind={k: int(v) for k, v in lo["kind"].items()}, weight=lo["weight"], class_weights=lo["class_weights"],
                                  optimizer=lo["optimizer"], lr=lo["lr"], beta_1=lo["beta_1"], beta_2=lo["beta_2"], momentum=lo["momentum"]))
        og = root["optimizer_weights"]
        m.engine.M1.copy_(torch.from_numpy(np.ascontiguousarray(og["rua/m:0"]))); m.engine.V1.copy_(torch.from_numpy(np.ascontiguousarray(og["rua/v:0"])))
        m.engine.t = int(og["rua/iterations:0"])
        m._finish_compile(restored_optimizer_state=True)
    return m


class Resunet_a(object):
    """Same constructor as the reference (ResUnet_a/model2.py:6-12): builds `.model`."""
    variant = "model2"

    def __init__(self, input_shape, num_classes, args, inputs=None):
        self.num_classes = num_classes
        self.img_height, self.img_width, self.img_channel = input_shape
        self.args = args
        self.inputs = inputs
        self.model = self.build_model_ResUneta()

    def build_model_ResUneta(self):
        cfg = ModelConfig(input_shape=(self.img_height, self.img_width, self.img_channel), num_classes=self.num_classes,
                          multitasking=bool(getattr(self.args, "multitasking", False)), variant=self.variant,
                          width=int(getattr(self.args, "width", 32)), depth=int(getattr(self.args, "depth", 6)))
        return Model(cfg, dtype=getattr(self.args, "dtype", "bf16"), seed=int(getattr(self.args, "seed", 0)))


class Resunet_a_v1(Resunet_a):
    """ResUnet_a/model.py graph (no skip in the ResBlock sum, no BN on the 1x1 convs, conv-then-upsample)."""
    variant = "model"
