"""Keras' view of the ResUnet-a graph: which layers exist, how they are named and in which ORDER `model.layers` lists them.

Why this exists (checkpoint interop): Keras' topological `model.load_weights(path)` zips the HDF5 file's
`layer_names` with `model.layers`, and a functional model orders `model.layers` by graph DEPTH, not by creation: all four
dilation branches of a ResBlock (/root/reference/ResUnet_a/model2.py:15-34) sit at the same depths, so the k-th weighted layer
in depth order is the first BatchNorm of branch 0, then of branch 1, ... - not branch 0's (bn, conv, bn, conv) first.  A file
whose `layer_names` follow the creation order therefore fails on the TensorFlow side (a BatchNorm's 4 arrays against a conv's
2).  `weighted_layer_order()` reproduces Keras' ordering so that `h5lite.keras_group_from_weights` can emit it.

The ordering is the one of `tensorflow/python/keras/engine/functional.py::_map_graph_network` (TF 2.2's network.py has the same
code), restated for a graph in which every layer is called once:

  1. depth-first walk from the model's outputs (a dict of outputs is flattened in sorted key order: bound, color, dist, seg);
     a layer gets its `layer_index` when it is first reached (pre-order), its inputs are then visited in the order they were
     passed to the layer, and it is appended to `nodes_in_decreasing_depth` when its inputs are done (post-order);
  2. in the reverse of that post-order: depth(layer) = max over its consumers of depth(consumer) + 1, outputs start at 0;
  3. `model.layers` = layers grouped by decreasing depth, within one depth sorted by `layer_index`.

Layer names are Keras' automatic ones (`conv2d`, `conv2d_1`, ..., `batch_normalization_7`, `activation_3`, `add`, ...: one counter
per layer class, in creation order, explicit `name=` arguments take no number) for a model that is the first one built in its
process; `engine.ParamStore` numbers its weighted layers the same way, which `tests/test_h5lite.py` checks.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

_BASE = {"input": "input", "conv": "conv2d", "bn": "batch_normalization", "act": "activation", "add": "add", "cat": "concatenate",
         "pool": "max_pooling2d", "up": "up_sampling2d", "pad": "zero_padding2d"}


class KerasGraph:
    """Topology only: layers[i] = (name, kind, [input layer ids])."""

    def __init__(self):
        self.layers: List[Tuple[str, str, List[int]]] = []
        self.counts: Dict[str, int] = {}
        self.outputs: Dict[str, int] = {}

    def add(self, kind: str, inputs: List[int], name: Optional[str] = None) -> int:
        if name is None:
            n = self.counts.get(kind, 0)
            self.counts[kind] = n + 1
            if kind == "input":
                name = f"input_{n + 1}"
            else:
                name = _BASE[kind] if n == 0 else f"{_BASE[kind]}_{n}"
        self.layers.append((name, kind, list(inputs)))
        return len(self.layers) - 1

    # -- Keras' ordering ---------------------------------------------------------------------------------
    def layer_order(self) -> List[int]:
        outs = [self.outputs[k] for k in sorted(self.outputs)]          # nest.flatten of a dict: sorted keys
        index: Dict[int, int] = {}
        post: List[int] = []
        done = set()
        for o in outs:                                                   # iterative form of _build_map_helper
            stack = [(o, 0)]
            while stack:
                node, k = stack.pop()
                if k == 0:
                    if node in done:
                        continue
                    if node not in index:
                        index[node] = len(index)
                ins = self.layers[node][2]
                if k < len(ins):
                    stack.append((node, k + 1))
                    if ins[k] not in done:
                        stack.append((ins[k], 0))
                else:
                    if node not in done:
                        done.add(node)
                        post.append(node)
        depth: Dict[int, int] = {}
        for node in reversed(post):
            d = depth.setdefault(node, 0)
            for p in self.layers[node][2]:
                depth[p] = max(d + 1, depth.get(p, 0))
        by_depth: Dict[int, List[int]] = {}
        for node, d in depth.items():
            by_depth.setdefault(d, []).append(node)
        order: List[int] = []
        for d in sorted(by_depth, reverse=True):
            order += sorted(by_depth[d], key=lambda n: index[n])
        return order

    def names_in_model_order(self, weighted_only: bool = False) -> List[str]:
        return [self.layers[i][0] for i in self.layer_order() if not weighted_only or self.layers[i][1] in ("conv", "bn")]

    def names_in_creation_order(self, weighted_only: bool = False) -> List[str]:
        return [n for n, k, _ in self.layers if not weighted_only or k in ("conv", "bn")]


def build(input_width: int, multitasking: bool, variant: str = "model2", depth: int = 6) -> KerasGraph:
    """The layer graph of ResUnet_a/model2.py:14-193 (variant 'model2') or ResUnet_a/model.py:14-171 ('model'); depth 7 is the
    extrapolation of engine.ModelConfig (one more encoder / decoder stage of the same pattern)."""
    g = KerasGraph()
    v2 = variant == "model2"
    dil = [[1, 3, 15, 31], [1, 3, 15, 31], [1, 3, 15], [1, 3, 15], [1], [1], [1]][:depth]

    def resblock(x, dils):
        outs = [x] if v2 else []
        for _ in dils:
            t = g.add("bn", [x]); t = g.add("act", [t]); t = g.add("conv", [t])
            t = g.add("bn", [t]); t = g.add("act", [t]); t = g.add("conv", [t])
            outs.append(t)
        if v2 or len(dils) > 1:
            return g.add("add", outs)
        return outs[0]

    def psp(x):
        ks = [1, 2] + ([4] if input_width >= 128 else []) + ([8] if input_width >= 256 else [])
        br = [g.add("pool", [x]) for _ in ks]
        if v2:                                                   # pool -> upsample -> conv + BN  (model2.py:47-68)
            br = [g.add("up", [b]) for b in br]
            br = [g.add("bn", [g.add("conv", [b])]) for b in br]
        else:                                                    # pool -> conv -> upsample       (model.py:40-57)
            br = [g.add("conv", [b]) for b in br]
            br = [g.add("up", [b]) for b in br]
        c = g.add("cat", br + [x])
        c = g.add("conv", [c])
        return g.add("bn", [c]) if v2 else c

    def combine(a, skip):
        t = g.add("act", [a]); t = g.add("cat", [t, skip]); t = g.add("conv", [t])
        return g.add("bn", [t]) if v2 else t

    def upsampling(x):
        if v2:                                                   # model2.py:89-94
            t = g.add("up", [x]); t = g.add("conv", [t])
            return g.add("bn", [t])
        t = g.add("conv", [x])                                   # model.py:93-94
        return g.add("up", [t])

    inp = g.add("input", [])
    c1 = x = g.add("conv", [inp])
    skips = []
    for i, dils in enumerate(dil):
        if i > 0:
            x = g.add("conv", [x])
        x = resblock(x, dils)
        skips.append(x)
    x = psp(x)
    if v2:
        x = g.add("act", [x])
    for i in range(depth - 2, -1, -1):
        x = upsampling(x)
        x = combine(x, skips[i])
        x = resblock(x, dil[i])
    x_comb = combine(x, c1)
    x_psp = psp(x_comb)
    if v2:
        x_psp = g.add("act", [x_psp])
    if not multitasking:
        t = g.add("conv", [x_psp])
        g.outputs = {"out": g.add("act", [t])}
        return g
    t = g.add("pad", [x_psp]); t = g.add("conv", [t], "seg1"); t = g.add("pad", [t]); t = g.add("conv", [t], "seg2")
    t = g.add("conv", [t], "seg3")
    seg = g.add("act", [t], "seg")
    t = g.add("pad", [x_psp]); t = g.add("conv", [t]); t = g.add("conv", [t])
    bound = g.add("act", [t], "bound")
    t = g.add("pad", [x_comb]); t = g.add("conv", [t]); t = g.add("pad", [t]); t = g.add("conv", [t]); t = g.add("conv", [t])
    dist = g.add("act", [t], "dist")
    color = g.add("conv", [x_comb], "color")
    g.outputs = {"seg": seg, "bound": bound, "dist": dist, "color": color}
    return g


def weighted_layer_order(input_width: int, multitasking: bool, variant: str = "model2", depth: int = 6) -> List[str]:
    """Names of the layers that carry weights, in the order Keras' `model.layers` lists them."""
    return build(input_width, multitasking, variant, depth).names_in_model_order(weighted_only=True)
