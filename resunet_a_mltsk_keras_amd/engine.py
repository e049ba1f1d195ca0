"""Static-plan execution engine for the ResUnet-a training path on MI355X.

The network is a fixed graph on fixed shapes, so instead of tracing or autograd the engine
RECORDS, once per (batch, mode), the exact list of librua_hip.so launches for forward,
backward and the optimizer, with every buffer pre-allocated; a step replays that list on one
HIP stream (optionally as a captured HIP graph).  PyTorch is used for device memory, streams
and torch.distributed only.  Reference call sites replaced: ResUnet_a/model2.py:14-193
(graph), train_ISPRS.py:131,148,167,186 (train_on_batch / test_on_batch), :404-407 (optimizers).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L

BN_EPS, BN_MOMENTUM, KERAS_EPS = 1e-3, 0.99, 1e-7
HEADS = ["seg", "bound", "dist", "color"]
SUMS_REPLICAS = 1          # copies of a head's Tanimoto moments the blocks of rua_head_fwd_loss_rep spread their atomics over (measured: 8 copies take ~1 us off
                           # a head's forward and put ~8 us on the one-block finalisation that folds them - one copy it is)


@dataclass
class ModelConfig:
    input_shape: Tuple[int, int, int] = (256, 256, 3)
    num_classes: int = 5
    multitasking: bool = False
    variant: str = "model2"
    width: int = 32
    depth: int = 6

    def levels(self):
        dil = [[1, 3, 15, 31], [1, 3, 15, 31], [1, 3, 15], [1, 3, 15], [1], [1], [1]]
        return [(self.width * (2 ** i), dil[i]) for i in range(self.depth)]


@dataclass
class LossSpec:
    """What model.compile(...) received (train_ISPRS.py:411-461)."""
    kind: Dict[str, int] = field(default_factory=dict)       # head -> L.LOSS_*
    weight: Dict[str, float] = field(default_factory=dict)   # head -> loss weight
    class_weights: Optional[List[float]] = None
    optimizer: str = "adam"
    lr: float = 1e-3
    beta_1: float = 0.9
    beta_2: float = 0.999
    momentum: float = 0.8


# ---------------------------------------------------------------------------------------
class ParamStore:
    """All trainable parameters in ONE flat fp32 buffer (Keras creation order), so the optimizer
    is one launch and gradient all-reduce buckets are contiguous slices.  BN moving statistics
    live in a second flat buffer.  Conv kernels are stored [tap][Cout][Cin_segment]."""

    def __init__(self):
        self.entries: List[dict] = []      # trainable entries in order
        self.state: List[dict] = []        # moving_mean / moving_variance
        self.n = 0
        self.ns = 0
        self.nw = 0                        # elements in the activation-dtype weight copies
        self.convs: Dict[str, dict] = {}
        self.bns: Dict[str, dict] = {}
        self.n_conv = 0
        self.n_bn = 0

    def _add(self, name, size, **kw):
        off = self.n
        self.entries.append(dict(name=name, off=off, size=size, **kw))
        self.n += (size + 15) // 16 * 16          # 64-byte aligned slices
        return off

    def conv(self, cins: List[int], cout: int, taps: int, name: Optional[str] = None, mfma: bool = True):
        if name is None:
            name = "conv2d" if self.n_conv == 0 else f"conv2d_{self.n_conv}"
            self.n_conv += 1
        segs = []
        c0 = 0
        for ci in cins:
            off = self._add(name + "/kernel", taps * cout * ci, kind="kernel", layer=name, taps=taps, cout=cout, cin=ci, cin_off=c0,
                            cin_total=sum(cins))
            dst = -1
            if mfma:
                dst = off                                  # the activation-dtype copies share the master's index space: the optimizer writes the forward
                self.nw = self.n                           # copy element for element (rua_adam_step_w), rua_weight_prep_dgrad builds the other from it
            segs.append(dict(off=off, dst=dst, C=ci))
            c0 += ci
        boff = self._add(name + "/bias", cout, kind="bias", layer=name)
        rec = dict(name=name, segs=segs, bias=boff, taps=taps, cout=cout, mfma=mfma)
        self.convs[name] = rec
        return rec

    def bn(self, c: int):
        name = "batch_normalization" if self.n_bn == 0 else f"batch_normalization_{self.n_bn}"
        self.n_bn += 1
        g = self._add(name + "/gamma", c, kind="gamma", layer=name)
        b = self._add(name + "/beta", c, kind="beta", layer=name)
        mm = self.ns
        self.state.append(dict(name=name + "/moving_mean", off=mm, size=c, init=0.0))
        self.ns += (c + 15) // 16 * 16
        mv = self.ns
        self.state.append(dict(name=name + "/moving_variance", off=mv, size=c, init=1.0))
        self.ns += (c + 15) // 16 * 16
        rec = dict(name=name, gamma=g, beta=b, mm=mm, mv=mv, C=c)
        self.bns[name] = rec
        return rec

    # -- host <-> device ------------------------------------------------------------------
    def init_host(self, seed: int) -> Tuple[np.ndarray, np.ndarray]:
        rng = np.random.default_rng(seed)
        P = np.zeros(self.n, np.float32)
        S = np.zeros(self.ns, np.float32)
        limits = {}
        for e in self.entries:
            if e["kind"] == "kernel":
                fan = e["taps"] * e["cin_total"] + e["taps"] * e["cout"]
                lim = math.sqrt(6.0 / fan)
                P[e["off"]:e["off"] + e["size"]] = rng.uniform(-lim, lim, e["size"]).astype(np.float32)
            elif e["kind"] == "gamma":
                P[e["off"]:e["off"] + e["size"]] = 1.0
        for s in self.state:
            S[s["off"]:s["off"] + s["size"]] = s["init"]
        return P, S

    def from_keras(self, weights: Dict[str, np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
        """Keras-layout dict (kernel HWIO, bias, gamma, beta, moving_*) -> flat host buffers."""
        P = np.zeros(self.n, np.float32)
        S = np.zeros(self.ns, np.float32)
        for e in self.entries:
            w = np.asarray(weights[e["name"]], np.float32)
            if e["kind"] == "kernel":
                kh, kw, cin, cout = w.shape
                assert kh * kw == e["taps"] and cout == e["cout"] and cin == e["cin_total"], (e["name"], w.shape)
                seg = w[:, :, e["cin_off"]:e["cin_off"] + e["cin"], :]                      # (kh,kw,ci,co)
                P[e["off"]:e["off"] + e["size"]] = seg.reshape(kh * kw, e["cin"], cout).transpose(0, 2, 1).reshape(-1)
            else:
                assert w.size == e["size"], (e["name"], w.shape)
                P[e["off"]:e["off"] + e["size"]] = w.reshape(-1)
        for s in self.state:
            S[s["off"]:s["off"] + s["size"]] = np.asarray(weights[s["name"]], np.float32).reshape(-1)
        return P, S

    def to_keras(self, P: np.ndarray, S: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        out: Dict[str, np.ndarray] = {}
        for e in self.entries:
            v = P[e["off"]:e["off"] + e["size"]]
            if e["kind"] == "kernel":
                k = int(round(math.sqrt(e["taps"])))
                seg = v.reshape(e["taps"], e["cout"], e["cin"]).transpose(0, 2, 1).reshape(k, k, e["cin"], e["cout"])
                if e["name"] not in out:
                    out[e["name"]] = np.zeros((k, k, e["cin_total"], e["cout"]), np.float32)
                out[e["name"]][:, :, e["cin_off"]:e["cin_off"] + e["cin"], :] = seg
            else:
                out[e["name"]] = v.copy()
        if S is not None:
            for s in self.state:
                out[s["name"]] = S[s["off"]:s["off"] + s["size"]].copy()
        return out

    def count(self) -> int:
        return sum(e["size"] for e in self.entries) + sum(s["size"] for s in self.state)


# ---------------------------------------------------------------------------------------
class Ten:
    """A device activation tensor [N][H][W][C] (storage owned by a torch tensor)."""
    __slots__ = ("N", "H", "W", "C", "t", "ptr", "grad", "gw", "stats", "f32", "relu_out", "masked_w", "plain_w", "bias_offs", "bias_done", "bn_src")

    def __init__(self, t: torch.Tensor, N, H, W, C, f32=False):
        self.t, self.N, self.H, self.W, self.C, self.f32 = t, N, H, W, C, f32
        self.ptr = t.data_ptr()
        self.grad: Optional["Ten"] = None
        self.gw = False
        self.stats: Optional[int] = None
        self.relu_out = False      # output of a fused ReLU: gradient writers that know how apply the (t > 0) mask themselves
        self.masked_w = 0          # gradient writers that applied it / that did not
        self.plain_w = 0
        self.bias_offs = None      # bias of the convolution that produced this tensor, if its gradient (the per-channel sum of this tensor's
        self.bias_done = False     # gradient) may be taken by the kernel that writes the gradient: done = it was
        self.bn_src = None         # bn_node record of the ReLU-less BatchNorm that produced this tensor: the kernel that writes its whole gradient may take the
                                   # BatchNorm backward's statistics (sum g, sum g * this tensor) on the way

    @property
    def M(self):
        return self.N * self.H * self.W


def stats_replicas(blocks: int) -> int:
    """Mirror of rua_stats_replicas(): replicas of a statistics buffer fed by `blocks` workgroups."""
    r = 1
    while r < 32 and r * 8 < blocks:
        r *= 2
    return r


def conv_stats_replicas(blocks: int) -> int:
    """Replicas for statistics produced by a convolution epilogue: its workgroups finish spread over the kernel's run
    time, so ~64 adds per address do not queue up (same-address fp64 atomics serialise at ~180 ns) - and every replica
    costs each consumer block of the fused BatchNorm kernels a dependent round trip per 4 replicas in its prologue."""
    r = 1
    while r < 32 and r * 64 < blocks:
        r *= 2
    return r


class Stat:
    """fp64 statistics buffer [R][2][C] in the per-step arena."""
    __slots__ = ("ptr", "R")

    def __init__(self, ptr, R):
        self.ptr, self.R = ptr, R


class _Dummy:
    """Stand-in for a device allocation during the dry (parameter-enumeration) pass."""

    def __init__(self, n=0):
        self._n = n

    def data_ptr(self):
        return 0

    def numel(self):
        return self._n

    def element_size(self):
        return 4

    def stride(self, i):
        return 0


class Plan:
    """Recorded launches.  Entries carry a lane: lane 0 is the caller's stream, lanes 1..3 are engine-owned side
    streams used for the independent branches of a ResBlock (fork / join markers become event dependencies, which a
    HIP-graph capture turns into parallel graph branches)."""
    FORK, JOIN = "fork", "join"

    def __init__(self, dry=False):
        self.calls: List[tuple] = []
        self.keep: List[object] = []
        self.dry = dry
        self.lane = 0
        self.open_fork = 0
        self.scope: Optional[str] = None     # composite the next launches belong to (per-block timing in bench.py)
        self.scopes: List[Optional[str]] = []  # scope of calls[i]

    def add(self, name: str, *args):
        if not self.dry:
            self.calls.append((L.lib().raw(name), name, args, self.lane))
            self.scopes.append(self.scope)

    def fork(self, n: int):
        if not self.dry and n > 1:
            self.calls.append((None, Plan.FORK, n, 0))
            self.scopes.append(self.scope)
        self.open_fork = n

    def join(self, n: int):
        self.lane = 0
        if not self.dry and n > 1:
            self.calls.append((None, Plan.JOIN, n, 0))
            self.scopes.append(self.scope)
        self.open_fork = 0

    def set_lane(self, lane: int):
        self.lane = lane

    def safe_hook_index(self, idx: int) -> int:
        """First call index >= idx after which no side lane is running (where a 'gradients so far are final' hook may fire)."""
        depth, out = 0, None
        for i, c in enumerate(self.calls):
            if c[1] == Plan.FORK:
                depth += 1
            elif c[1] == Plan.JOIN:
                depth -= 1
            if i >= idx and depth == 0:
                return i
        return len(self.calls) - 1

    def run(self, stream_ptr: int, hooks=None, side=None, first=0, last=None):
        """Replay calls [first, last).  side: torch streams for lanes 1..; None => everything on the caller's stream.
        hooks: {call index -> python callable run right after that launch} (gradient buckets)."""
        lib = L.lib()
        ptrs = [C.c_void_p(stream_ptr)]
        main = None
        if side is not None:
            main = torch.cuda.current_stream()
            ptrs += [C.c_void_p(st.cuda_stream) for st in side]
        calls = self.calls if (first == 0 and last is None) else self.calls[first:last]
        for i, (fn, name, args, lane) in enumerate(calls, start=first):
            if fn is None:
                if side is not None:
                    if name == Plan.FORK:
                        ev = torch.cuda.Event()
                        ev.record(main)
                        for l in range(1, args):
                            side[l - 1].wait_event(ev)
                    else:
                        for l in range(1, args):
                            ev = torch.cuda.Event()
                            ev.record(side[l - 1])
                            main.wait_event(ev)
            else:
                rc = fn(*args, ptrs[lane if side is not None else 0])
                if rc != 0:
                    lib.check(rc, name)
            if hooks:
                h = hooks.get(i)
                if h is not None:
                    h()


class Coef:
    """Per-channel fp32 vectors of one BN application (scale, shift, mean, rstd, A, B, C)."""

    def __init__(self, g: "Graph", C: int):
        t = g.alloc((7, (C + 15) // 16 * 16), torch.float32, zero=True)
        self.t = t
        p, st = t.data_ptr(), t.stride(0) * 4
        self.scale, self.shift, self.mean, self.rstd, self.A, self.B, self.Cc = [p + i * st for i in range(7)]


class _BackSteps(list):
    """The backward closures of a graph, in creation order.  A closure appended while a composite tag is set records its launches
    under that tag (unless it names a scope itself, as the ResBlocks do)."""

    def __init__(self, g):
        super().__init__()
        self.g = g

    def append(self, fn):
        tag = self.g.cur_tag
        if tag is None:
            return super().append(fn)

        def run():
            Bp = self.g.bwd
            prev = Bp.scope
            Bp.scope = tag
            fn()
            Bp.scope = prev
        super().append(run)


class Graph:
    """One recorded instance of the network for a fixed batch size and mode."""

    def __init__(self, eng: "Engine", batch: int, training: bool, dry: bool = False):
        self.e = eng
        self.dry = dry
        self.cfg = eng.cfg
        self.B = batch
        self.training = training
        self.dev = eng.dev
        self.dt = eng.dt
        self.tdt = torch.bfloat16 if eng.dt == L.RUA_BF16 else torch.float32
        self.vec = 8 if eng.dt == L.RUA_BF16 else 4
        self.fwd = Plan(dry)
        self.bwd = Plan(dry)
        self.loss_plan = Plan(dry)
        self.back_steps: List = _BackSteps(self)
        self.cur_tag: Optional[str] = None      # composite being recorded (stem, down, PSP, combine, heads): per-composite timing in bench.py
        self.grad_touch: Dict[int, int] = {}
        self.cur_lane = 0
        self.block_tag = ""
        self.pending: List[tuple] = []          # deferred weight-gradient reductions (record, dW offset)
        self.pending_bytes = 0                  # ... and the bytes of partials they will read (Engine.flush_bytes)
        self.pending_bucket = 0
        self.allocs: List[torch.Tensor] = []
        self.stats_used = 16                 # the first 16 doubles of the arena are the loss / metric scalars
        self.act_bytes = 0
        self._build()

    # -- allocation -------------------------------------------------------------------------
    def alloc(self, shape, dtype, zero=False):
        if self.dry:
            return _Dummy(int(np.prod(shape)))
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)
        self.act_bytes += t.numel() * t.element_size()
        self.allocs.append(t)          # the plan holds raw pointers: the graph owns every buffer for its lifetime
        return t

    def new(self, N, H, W, C, f32=False) -> Ten:
        return Ten(self.alloc((N, H, W, C), torch.float32 if f32 else self.tdt), N, H, W, C, f32)

    def like(self, x: Ten) -> Ten:
        return self.new(x.N, x.H, x.W, x.C, x.f32)

    def salloc(self, n: int) -> int:
        """n doubles from the per-step statistics arena (zeroed at the start of every step)."""
        off = self.stats_used
        self.stats_used += (n + 7) // 8 * 8
        if self.dry:
            return 0
        assert self.stats_used <= self.e.stats_arena.numel(), "statistics arena exhausted"
        return self.e.stats_arena.data_ptr() + off * 8

    def stat(self, C: int, blocks: int, burst: bool = False) -> Stat:
        """burst: the producer's workgroups all finish together (rua_col_stats*); else a convolution epilogue."""
        R = stats_replicas(blocks) if burst else conv_stats_replicas(blocks)
        return Stat(self.salloc(R * 2 * C), R)

    def gacc(self, x: Ten, masked: bool = False) -> Tuple[Ten, int]:
        """Gradient buffer of x and whether the next writer must accumulate.  masked: this writer applies x's ReLU mask to
        what it writes (the mask is idempotent on a masked sum, so it commutes with accumulation)."""
        assert not x.bias_done, "a gradient whose per-channel sums were already taken (bias gradient) gets another writer"
        if x.grad is None:
            x.grad = self.like(x)
        acc = 1 if x.gw else 0
        x.gw = True
        if masked:
            x.masked_w += 1
        else:
            x.plain_w += 1
        return x.grad, acc

    # -- primitive recorders ------------------------------------------------------------------
    def P(self, off):   # pointer into the flat fp32 parameter buffer
        return 0 if self.dry else self.e.P.data_ptr() + off * 4

    def G(self, off):
        if self.dry:
            return 0
        self.grad_touch[off] = len(self.bwd.calls)          # index of the launch about to be recorded
        return self.e.G.data_ptr() + off * 4

    def S(self, off):
        return 0 if self.dry else self.e.S.data_ptr() + off * 4

    def Wf(self, dst):
        return 0 if self.dry else self.e.Wf.data_ptr() + dst * self.e.esize

    def Wd(self, dst):
        return 0 if self.dry else self.e.Wd.data_ptr() + dst * self.e.esize

    def stat_blocks(self, x: Ten) -> int:
        """Workgroups rua_col_stats / rua_col_stats2 launch for x (>= 8 pieces per thread, at most 1024)."""
        return max(1, min(1024, x.M * (x.C // self.vec) // 2048))

    def col_stats(self, plan: Plan, x: Ten) -> Stat:
        s = self.stat(x.C, self.stat_blocks(x), burst=True)
        plan.add("rua_col_stats", x.ptr, x.M, x.C, s.ptr, s.R, self.dt)
        return s

    def bn_finalize(self, plan: Plan, stats: Optional[Stat], count, bn, bessel=None) -> Coef:
        c = Coef(self, bn["C"])
        plan.keep.append(c)
        plan.add("rua_bn_finalize", stats.ptr if self.training else None, stats.R if self.training else 1, float(count), float(bessel or count),
                 self.P(bn["gamma"]), self.P(bn["beta"]), self.S(bn["mm"]), self.S(bn["mv"]), BN_MOMENTUM, BN_EPS,
                 1 if self.training else 0, c.scale, c.shift, c.mean, c.rstd, bn["C"])
        return c

    def bn_apply(self, plan: Plan, x: Ten, coefs: List[Coef], relu: bool) -> List[Ten]:
        outs = [self.like(x) for _ in coefs]
        sc = L.ptr_array([c.scale for c in coefs]); sh = L.ptr_array([c.shift for c in coefs]); ou = L.ptr_array([o.ptr for o in outs])
        plan.keep += [sc, sh, ou]
        plan.add("rua_bn_apply", x.ptr, len(coefs), sc, sh, 1 if relu else 0, ou, x.M, x.C, self.dt)
        return outs

    def bn_fwd(self, plan: Plan, x: Ten, bns: List[dict], relu: bool, stats: Optional[Stat], count, bessel=None, defer: Optional[List] = None,
               out_stats: bool = False):
        """[relu](BN_b(x)) for every branch b in ONE launch: coefficients from the statistics in the kernel prologue,
        published to `Coef` buffers (block 0) for the ReLU masks / backward; moving statistics updated in training."""
        outs = [self.like(x) for _ in bns]
        coefs = [Coef(self, bn["C"]) for bn in bns]
        d = L.BnFwdDesc()
        d.x, d.M, d.C, d.dtype, d.nb, d.relu = x.ptr, x.M, x.C, self.dt, len(bns), 1 if relu else 0
        d.training = 1 if self.training else 0
        if self.training:
            d.stats, d.replicas = stats.ptr, stats.R
        d.count, d.bessel_n, d.momentum, d.eps = float(count), float(bessel or count), BN_MOMENTUM, BN_EPS
        for i, (bn, c, o) in enumerate(zip(bns, coefs, outs)):
            b = d.br[i]
            b.gamma, b.beta, b.moving_mean, b.moving_var = self.P(bn["gamma"]), self.P(bn["beta"]), self.S(bn["mm"]), self.S(bn["mv"])
            b.scale, b.shift, b.mean, b.rstd, b.out = c.scale, c.shift, c.mean, c.rstd, o.ptr
            if out_stats and self.training and not relu and not self.dry and self.e.bn_out_stats:
                # the output's own statistics come out of the coefficients (sum = M beta, sum of squares = M (beta^2 + gamma^2 var / (var + eps))):
                # the BatchNorms of the ResBlock this tensor feeds need no rua_col_stats pass over it
                o.stats = Stat(self.salloc(2 * x.C), 1)
                b.out_stats = o.stats.ptr
        plan.keep += coefs
        if defer is not None:                                # the caller issues several as one rua_bn_fwd_group
            defer.append(d)
            return outs, coefs
        plan.keep.append(d)
        plan.add("rua_bn_fwd", C.byref(d))
        return outs, coefs

    def issue_bn_fwd(self, plan: Plan, ds: List):
        """The deferred rua_bn_fwd descriptors `ds` as one rua_bn_fwd_group launch."""
        if len(ds) == 1:
            plan.keep.append(ds[0])
            plan.add("rua_bn_fwd", C.byref(ds[0]))
        elif ds:
            arr = (L.BnFwdDesc * len(ds))()
            for i, d in enumerate(ds):
                C.memmove(C.byref(arr, i * C.sizeof(L.BnFwdDesc)), C.byref(d), C.sizeof(L.BnFwdDesc))
            plan.keep.append(arr)
            plan.add("rua_bn_fwd_group", arr, len(ds))

    def issue_bn_bwd(self, plan: Plan, ds: List):
        if len(ds) == 1:
            plan.keep.append(ds[0])
            plan.add("rua_bn_bwd", C.byref(ds[0]))
        elif ds:
            arr = (L.BnBwdDesc * len(ds))()
            for i, d in enumerate(ds):
                C.memmove(C.byref(arr, i * C.sizeof(L.BnBwdDesc)), C.byref(d), C.sizeof(L.BnBwdDesc))
            plan.keep.append(arr)
            plan.add("rua_bn_bwd_group", arr, len(ds))

    def bn_fwd_group(self, plan: Plan, items: List[tuple], relu: bool, count):
        """[relu](BN(x_i)) for several tensors of equal shape, each with its own BatchNorm and statistics, as ONE launch: items = (x, bn, stats);
        returns ([out_i], [coef_i])."""
        ds, outs, coefs = [], [], []
        for x, bn, st in items:
            o, c = self.bn_fwd(plan, x, [bn], relu, st, count, defer=ds)
            outs.append(o[0]); coefs.append(c[0])
        self.issue_bn_fwd(plan, ds)
        return outs, coefs

    def bn_coefs(self, plan: Plan, like: Ten, bns: List[dict], stats: List[Optional[Stat]], count, bessel=None) -> List[Coef]:
        """Coefficients (scale, shift, mean, rstd; moving statistics updated in training) of several BatchNorms in ONE
        one-block launch, nothing applied: the consumers normalise on load.  stats[i]: branch i's own statistics."""
        coefs = [Coef(self, bn["C"]) for bn in bns]
        d = L.BnFwdDesc()
        d.x, d.M, d.C, d.dtype, d.nb, d.relu = None, like.M, like.C, self.dt, len(bns), 1
        d.training = 1 if self.training else 0
        d.count, d.bessel_n, d.momentum, d.eps = float(count), float(bessel or count), BN_MOMENTUM, BN_EPS
        for i, (bn, c, st) in enumerate(zip(bns, coefs, stats)):
            b = d.br[i]
            b.gamma, b.beta, b.moving_mean, b.moving_var = self.P(bn["gamma"]), self.P(bn["beta"]), self.S(bn["mm"]), self.S(bn["mv"])
            b.scale, b.shift, b.mean, b.rstd, b.out = c.scale, c.shift, c.mean, c.rstd, None
            if self.training:
                b.stats, b.replicas = st.ptr, st.R
        plan.keep += [d] + coefs
        plan.add("rua_bn_fwd", C.byref(d))
        return coefs

    def bn_bwd(self, plan: Plan, gs: List[Ten], coefs: List[Coef], bns: List[dict], stats2: List[Stat], x: Ten, out: Ten,
               accumulate: int, count, dskip: Optional[Ten] = None, masked=False, skip_bias: Optional[List[int]] = None, defer: Optional[List] = None,
               dx_bias: bool = False, stats2_out: Optional[List[bool]] = None):
        """dx (=|+=) [dskip] + sum_b BN-backward_b(g_b) in ONE launch; dgamma/dbeta added by block 0.
        skip_bias: bias offsets whose gradient is the per-channel sum of dskip - accumulated by this launch while it reads
        dskip anyway (instead of a col_stats pass over the same tensor), converted by one rua_stats_to_f32."""
        d = L.BnBwdDesc()
        d.x, d.dx, d.M, d.C, d.dtype, d.nb = x.ptr, out.ptr, x.M, x.C, self.dt, len(gs)
        d.dskip = dskip.ptr if dskip is not None else None
        d.masked, d.accumulate, d.count = 1 if masked else 0, accumulate, float(count)
        for i, (g, c, bn, s2) in enumerate(zip(gs, coefs, bns, stats2)):
            b = d.br[i]
            b.g, b.stats2, b.replicas = g.ptr, s2.ptr, s2.R
            b.stats2_out = 1 if (stats2_out and stats2_out[i]) else 0
            b.gamma, b.mean, b.rstd, b.scale, b.shift = self.P(bn["gamma"]), c.mean, c.rstd, c.scale, c.shift
            b.dgamma, b.dbeta = self.G(bn["gamma"]), self.G(bn["beta"])
        st = None
        cg = x.C // self.vec
        if skip_bias and dskip is not None and not (cg <= 256 and 256 % cg == 0):
            self.bias_grad(plan, dskip, skip_bias)            # (d7 in fp32: 2048 channels = 512 pieces per pixel: its own pass)
            skip_bias = None
        if skip_bias and dskip is not None:
            blocks = max(1, min(1024, x.M * (x.C // self.vec) // 256))
            st = self.stat(x.C, blocks, burst=True)
            d.skip_stats, d.skip_replicas = st.ptr, st.R
        if defer is not None:                                # the caller issues several as one rua_bn_bwd_group
            assert st is None
            defer.append(d)
            return
        sx = None
        if (dx_bias and x.bn_src is not None and not x.bn_src["fused"] and not accumulate and not self.dry and self.e.bn_dx_bias and cg <= 256 and 256 % cg == 0):
            # x is the output of a ReLU-less BatchNorm (the combine BatchNorm in front of a decoder ResBlock, model2.py:86) and `out` its complete gradient g:
            # the sums of g and of g * x taken here are the statistics that BatchNorm's backward needs (it converts x back to its own input): no rua_col_stats2 pass
            blocks = max(1, min(1024, x.M * (x.C // self.vec) // 256))
            s2 = self.stat(x.C, blocks, burst=True)
            d.dx_stats, d.dx_replicas = s2.ptr, s2.R
            x.bn_src["s2"], x.bn_src["s2_out"] = s2, True
            x.bias_done = True                               # (no further writer of this gradient)
        elif (dx_bias and x.bias_offs and not accumulate and not self.dry and self.e.bn_dx_bias and cg <= 256 and 256 % cg == 0):
            # x is the output of a convolution with a bias and `out` its complete gradient: the per-channel sums of what this launch
            # writes ARE that bias gradient (no rua_col_stats pass over the gradient: the stride-2 convs in front of the encoder ResBlocks)
            blocks = max(1, min(1024, x.M * (x.C // self.vec) // 256))
            sx = self.stat(x.C, blocks, burst=True)
            d.dx_stats, d.dx_replicas = sx.ptr, sx.R
            x.bias_done = True
        plan.keep.append(d)
        plan.add("rua_bn_bwd", C.byref(d))
        if st is not None:
            self.stats_to_grads(plan, st, x.C, skip_bias)
        if sx is not None:
            self.stats_to_grads(plan, sx, x.C, x.bias_offs)

    def bn_bwd_group(self, plan: Plan, items: List[tuple]):
        """The one-branch BatchNorm backwards of a ResBlock's dilation branches (their second BatchNorms: own gradient, own input, own
        output) as ONE launch: items = (g, coef, bn, stats2, x, out, count)."""
        ds: List = []
        for g, c, bn, s2, x, out, cnt in items:
            self.bn_bwd(plan, [g], [c], [bn], [s2], x, out, 0, cnt, defer=ds)
        self.issue_bn_bwd(plan, ds)

    def conv(self, plan: Plan, segs, layer_segs, cout, bias_ptr, out: Ten, stride=1, residual: Optional[Ten] = None,
             out_relu=False, stats=None, bias_more=(), in_bn: Optional["Coef"] = None, accumulate: int = 0, in_fold=None):
        """segs: [(Ten, up_shift, dil, taps)], layer_segs: [param seg dict] (same order).  in_bn: the (single) source is
        read as relu(scale * x + shift) - BatchNorm + ReLU applied by the kernel as the tile lands (normalise on load)."""
        d = self.conv_desc(segs, layer_segs, cout, bias_ptr, out, stride, residual, out_relu, stats, bias_more, in_bn, accumulate, in_fold)
        plan.keep.append(d)
        plan.add("rua_conv_fwd", C.byref(d))

    def bn_fold(self, plan: Plan, bn: dict, stats: Stat, count, bessel=None):
        """(Coef, rua_bn_fold) of a training-mode BatchNorm whose coefficients the consuming convolution derives in its own
        prologue from `stats` (no coefficient launch); the Coef buffers are filled by that convolution."""
        c = Coef(self, bn["C"])
        f = L.BnFold()
        f.stats, f.replicas = stats.ptr, stats.R
        f.count, f.bessel_n, f.eps, f.momentum = float(count), float(bessel or count), BN_EPS, BN_MOMENTUM
        f.gamma, f.beta, f.moving_mean, f.moving_var = self.P(bn["gamma"]), self.P(bn["beta"]), self.S(bn["mm"]), self.S(bn["mv"])
        f.scale, f.shift, f.mean, f.rstd = c.scale, c.shift, c.mean, c.rstd
        plan.keep += [c, f]
        return c, f

    def conv_desc(self, segs, layer_segs, cout, bias_ptr, out: Ten, stride=1, residual=None, out_relu=False, stats=None, bias_more=(),
                  in_bn=None, accumulate=0, in_fold=None):
        d = L.ConvDesc()
        d.nseg = len(segs)
        for i, ((t, up, dil, taps), ps) in enumerate(zip(segs, layer_segs)):
            assert t.C == ps["C"], (t.C, ps)
            s = d.seg[i]
            s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = t.ptr, self.Wf(ps["dst"]), t.C, t.H, t.W, up, dil, taps
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = out.N, out.H, out.W, cout, stride, self.dt
        d.bias = bias_ptr
        for i, b in enumerate(bias_more):                  # added after `bias` in this order (<= 3)
            d.bias_more[i] = b
        if residual is not None:
            d.aux, d.aux_mode = residual.ptr, 1
        d.out_relu = 1 if out_relu else 0
        d.y, d.out_stride, d.OH, d.OW = out.ptr, 1, out.H, out.W
        if stats is not None:
            d.stats, d.stats_mode, d.stats_replicas = stats.ptr, 1, stats.R
        d.accumulate = accumulate
        if in_fold is not None:                             # the kernel derives (and publishes) the coefficients itself
            d.in_fold, d.in_relu = C.addressof(in_fold), 1
        elif in_bn is not None:
            d.in_scale, d.in_shift, d.in_relu = in_bn.scale, in_bn.shift, 1
        self._ws(d)
        return d

    def fused_input_ok(self, x: Ten, nf: int, dil: int) -> bool:
        """Does a 3x3 conv x -> nf channels at this dilation run on a kernel that normalises on load (and its weight
        gradient on the all-taps kernel, which does too)?  Asked of the library, not guessed (rua_conv_fused_input_ok)."""
        if self.dry or self.dt != L.RUA_BF16 or not self.e.fuse_bn:
            return False
        d = L.ConvDesc()
        d.nseg = 1
        sg = d.seg[0]
        sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = x.ptr, x.ptr, x.C, x.H, x.W, 0, dil, 9
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = x.N, x.H, x.W, nf, 1, self.dt
        d.y, d.out_stride, d.OH, d.OW = x.ptr, 1, x.H, x.W
        w = L.WgradDesc()
        w.a, w.C, w.Hs, w.Ws, w.dy, w.Cout, w.H, w.W = x.ptr, x.C, x.H, x.W, x.ptr, nf, x.H, x.W
        w.N, w.stride, w.dil, w.taps, w.dtype = x.N, 1, dil, 9, self.dt
        sc = self.e.scratches[0]
        w.workspace, w.workspace_bytes = sc.data_ptr(), sc.numel() * 4
        lib = L.lib()
        return lib.raw("rua_conv_fused_input_ok")(C.byref(d)) == 1 and lib.raw("rua_wgrad_kind")(C.byref(w)) == 1

    def _ws(self, d):
        """Shared split-K scratch (launches are serialised on one stream, so one buffer serves every conv)."""
        if not self.dry and self.e.split_k and d.N * d.H * d.W * d.Cout * 4 <= self.e.workspace.numel() * 4:
            w = self.e.workspaces[self.cur_lane]
            d.workspace, d.workspace_bytes = w.data_ptr(), w.numel() * 4

    def dgrad(self, plan: Plan, dy: Ten, wd_ptr, cin: int, dil: int, taps: int, out: Ten, accumulate: int,
              mask: Optional[Tuple[Ten, Optional[int], Optional[int]]] = None, stats2: Optional[int] = None,
              stat_aux: Optional[Ten] = None, out_stride: int = 1):
        """out (=|+=) conv(dy, W^T flipped) [* relu-mask(aux)], optional sum g / sum g*aux statistics."""
        d = self.dgrad_desc(dy, wd_ptr, cin, dil, taps, out, accumulate, mask, stats2, stat_aux, out_stride)
        plan.keep.append(d)
        plan.add("rua_conv_fwd", C.byref(d))

    def conv_group(self, plan: Plan, descs: List):
        """Independent convolutions (the dilation branches of a ResBlock) in one call: members on the same kernel share ONE grid."""
        if len(descs) == 1:
            plan.keep.append(descs[0])
            plan.add("rua_conv_fwd", C.byref(descs[0]))
            return
        if len(descs) > L.RUA_MAX_BRANCH:                      # (the five sources of a PSPPooling fuse conv: the library takes RUA_MAX_BRANCH members per call)
            self.conv_group(plan, descs[:L.RUA_MAX_BRANCH])
            self.conv_group(plan, descs[L.RUA_MAX_BRANCH:])
            return
        arr = (L.ConvDesc * len(descs))()
        for i, dsc in enumerate(descs):
            C.memmove(C.byref(arr, i * C.sizeof(L.ConvDesc)), C.byref(dsc), C.sizeof(L.ConvDesc))
        plan.keep.append(arr)
        plan.add("rua_conv_fwd_group", arr, len(descs))

    def conv_sum(self, plan: Plan, descs: List):
        """Convolutions into ONE output, summed (member 0 writes, the others accumulate): rua_conv_fwd_sum."""
        if len(descs) == 1:
            plan.keep.append(descs[0])
            plan.add("rua_conv_fwd", C.byref(descs[0]))
            return
        arr = (L.ConvDesc * len(descs))()
        for i, dsc in enumerate(descs):
            C.memmove(C.byref(arr, i * C.sizeof(L.ConvDesc)), C.byref(dsc), C.sizeof(L.ConvDesc))
        plan.keep.append(arr)
        plan.add("rua_conv_fwd_sum", arr, len(descs))

    def dgrad_desc(self, dy: Ten, wd_ptr, cin: int, dil: int, taps: int, out: Ten, accumulate: int, mask=None, stats2=None,
                   stat_aux=None, out_stride: int = 1):
        d = L.ConvDesc()
        d.nseg = 1
        s = d.seg[0]
        s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = dy.ptr, wd_ptr, dy.C, dy.H, dy.W, 0, dil, taps
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = dy.N, dy.H, dy.W, cin, 1, self.dt
        d.accumulate = accumulate
        d.y, d.out_stride, d.OH, d.OW = out.ptr, out_stride, out.H, out.W
        if mask is not None:
            d.aux, d.aux_mode, d.mscale, d.mshift = mask[0].ptr, 2, mask[1], mask[2]
        elif stat_aux is not None:
            d.aux, d.aux_mode = stat_aux.ptr, 3
        if stats2 is not None:
            d.stats, d.stats_mode, d.stats_replicas = stats2.ptr, 2, stats2.R
        self._ws(d)
        return d

    def wgrad_desc(self, plan: Plan, a: Ten, dy: Ten, dw_off: int, stride: int, dil: int, taps: int, in_bn: Optional["Coef"] = None,
                   may_flush: bool = True, group: int = 0, defer_ok: bool = True):
        d = L.WgradDesc()
        d.group_members = group                            # members of the rua_conv_wgrad_group call this descriptor belongs to
        d.a, d.C, d.Hs, d.Ws = a.ptr, a.C, a.H, a.W
        if in_bn is not None:                               # a is read as relu(scale * a + shift) (normalise on load)
            d.in_scale, d.in_shift, d.in_relu = in_bn.scale, in_bn.shift, 1
        d.dy, d.Cout, d.H, d.W = dy.ptr, dy.C, dy.H, dy.W
        d.N, d.stride, d.dil, d.taps, d.dtype = dy.N, stride, dil, taps, self.dt
        if not self.dry and self.e.wgrad_overwrite:
            d.overwrite_dev = self.e.ow_flag.data_ptr()    # whole steps store dW instead of adding to the zeroed arena (Engine._set_overwrite)
        defer = (not self.dry) and self.e.defer_reduce and plan is self.bwd and defer_ok
        if defer and may_flush:                             # before G(): a flush is a launch of its own and must not count as this one
            bucket = self.e.dist.bucket_of(dw_off) if self.e.dist is not None else 0
            if self.pending and (bucket != self.pending_bucket or self.pending_bytes > self.e.flush_bytes):
                self.flush_wgrad(plan)
            self.pending_bucket = bucket
        d.dw = self.G(dw_off)
        if not self.dry:
            sc = self.e.scratches[self.cur_lane]
            d.workspace, d.workspace_bytes = sc.data_ptr(), sc.numel() * 4
            if defer:
                self._defer_wgrad(plan, d, dw_off)
        return d

    def wgrad(self, plan: Plan, a: Ten, dy: Ten, dw_off: int, stride: int, dil: int, taps: int, in_bn: Optional["Coef"] = None):
        d = self.wgrad_desc(plan, a, dy, dw_off, stride, dil, taps, in_bn)
        plan.keep.append(d)
        plan.add("rua_conv_wgrad", C.byref(d))

    def wgrad_group(self, plan: Plan, specs: List[tuple]):
        """Independent weight gradients (the dilation branches of a ResBlock) in one call: members on the same kernel share ONE
        grid.  specs: (a, dy, dw_off, stride, dil, taps, in_bn).  Under data parallel the group counts as one launch of the
        bucket of its first member (a pending-reduction flush can only come in front of the whole group)."""
        if len(specs) == 1 or self.dry:
            for sp in specs:
                self.wgrad(plan, *sp)
            return
        if len(specs) > L.RUA_MAX_WGRAD_GROUP:
            self.wgrad_group(plan, specs[:L.RUA_MAX_WGRAD_GROUP])
            self.wgrad_group(plan, specs[L.RUA_MAX_WGRAD_GROUP:])
            return
        descs = [self.wgrad_desc(plan, *sp, may_flush=(i == 0), group=len(specs)) for i, sp in enumerate(specs)]
        arr = (L.WgradDesc * len(descs))()
        for i, dsc in enumerate(descs):
            C.memmove(C.byref(arr, i * C.sizeof(L.WgradDesc)), C.byref(dsc), C.sizeof(L.WgradDesc))
        plan.keep.append(arr)
        plan.add("rua_conv_wgrad_group", arr, len(descs))

    def wgrad_now_or_later(self, plan: Plan, specs: List[tuple]) -> List[tuple]:
        """The weight gradients of a ResBlock's SECOND convolutions: issued here - or returned, to ride in ONE group with the first convolutions' (whose
        dy exists three launches later).  Twice the members per grid = half the blocks per member: half the block partials to write and to reduce
        (levels 3 - 4: 24 MB per member) and half the ring fills per row of work.  Engine.merge_wgrad holds the channel counts that do."""
        if self.dry or len(specs) < 2 or 2 * len(specs) > L.RUA_MAX_WGRAD_GROUP or specs[0][0].C not in self.e.merge_wgrad:
            self.wgrad_group(plan, specs)
            return []
        return specs

    def wgrad_pw_group(self, plan: Plan, specs: List[tuple]):
        """The narrow 1x1 weight gradients of one composite (the sources of a concatenating conv, the branch convs of a PSPPooling: independent, 2 - 15 us
        apiece and mostly launch ramp + drain) as ONE rua_conv_wgrad_group call.  The library runs them as one grid when every member is a wgrad_pw launch with
        replicas / tickets of its own - so each member gets a private (zeroed) workspace here; anything else falls back to one launch each, as before."""
        lib = L.lib()
        if self.dry or len(specs) < 2 or self.dt != L.RUA_BF16 or not self.e.group_wgrad_pw or len(specs) > L.RUA_MAX_BRANCH:
            if len(specs) > L.RUA_MAX_BRANCH and not self.dry and self.dt == L.RUA_BF16 and self.e.group_wgrad_pw:
                self.wgrad_pw_group(plan, specs[:L.RUA_MAX_BRANCH])
                self.wgrad_pw_group(plan, specs[L.RUA_MAX_BRANCH:])
                return
            for sp in specs:
                self.wgrad(plan, *sp)
            return
        descs = [self.wgrad_desc(plan, *sp, may_flush=(i == 0)) for i, sp in enumerate(specs)]
        if not all(lib.raw("rua_wgrad_kind")(C.byref(d)) == 3 for d in descs):
            for d in descs:                                    # (descriptors are already recorded for the deferred reductions: launch exactly these)
                plan.keep.append(d)
                plan.add("rua_conv_wgrad", C.byref(d))
            return
        for d in descs:
            if d.defer:                                        # block partials, summed by the next rua_wgrad_reduce_batch: _defer_wgrad gave it a private workspace
                continue
            nbytes = int(lib.raw("rua_wgrad_workspace_bytes")(C.byref(d)))
            ws = self.alloc(((nbytes + 3) // 4,), torch.float32, zero=True)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        arr = (L.WgradDesc * len(descs))()
        for i, dsc in enumerate(descs):
            C.memmove(C.byref(arr, i * C.sizeof(L.WgradDesc)), C.byref(dsc), C.sizeof(L.WgradDesc))
        plan.keep.append(arr)
        plan.add("rua_conv_wgrad_group", arr, len(descs))

    # -- deferred weight-gradient reductions: the partial sums of many weight gradients (all-taps block partials, K-slice slabs)
    #    are added into dW by ONE batched launch instead of one small launch each (80 of them per cfg3 step).  Under data
    #    parallel the batch is flushed whenever the next weight gradient belongs to another all-reduce bucket, so a bucket's
    #    gradients are final - and its all-reduce starts - as early as before.
    def _defer_wgrad(self, plan: Plan, d, dw_off: int):
        lib = L.lib()
        rec = L.WgradPending()
        d.defer = 1
        lib.call("rua_wgrad_plan", C.byref(d), C.byref(rec))           # with the shared scratch: which kind of partials, if any
        if rec.kind == 0:
            d.defer = 0
            return
        # exactly the partials this call writes (all-taps: one per CU and output-channel half; slabs: one dW per K slice) + the tail
        nbytes = (self.e.cu_count * 9 * 32 * rec.CC * 4 if rec.kind == 1 else rec.parts * rec.n * 4) + (264 << 10)
        ws = self.alloc(((nbytes + 3) // 4,), torch.float32, zero=True)   # private until the flush (zeroed: the tail convention)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        lib.call("rua_wgrad_plan", C.byref(d), C.byref(rec))           # the record for the private workspace
        assert rec.kind != 0
        self.pending.append((rec, dw_off))
        self.pending_bytes += nbytes

    def flush_wgrad(self, plan: Plan):
        self.pending_bytes = 0
        if not self.pending:
            return
        recs = (L.WgradPending * len(self.pending))()
        blocks = 0
        for i, (r, off) in enumerate(self.pending):
            r.block_begin = blocks
            blocks += r.blocks
            C.memmove(C.byref(recs, i * C.sizeof(L.WgradPending)), C.byref(r), C.sizeof(L.WgradPending))
            self.grad_touch[off] = len(plan.calls)                     # this launch makes the gradient final
        table = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).to(self.dev)
        self.allocs.append(table)
        plan.add("rua_wgrad_reduce_batch", table.data_ptr(), len(self.pending), blocks)
        self.pending = []

    def bias_grad(self, plan: Plan, dy: Ten, bias_offs: List[int]):
        s = self.col_stats(plan, dy)
        self.stats_to_grads(plan, s, dy.C, bias_offs)

    def stats_to_grads(self, plan: Plan, s: Stat, Cc: int, offs: List[int]):
        """G[off .. off + C) += per-channel sums of `s` for every offset: its own small launch, or - deferred - records of the next
        batched reduction launch (see _defer_wgrad)."""
        if self.dry or not (self.e.defer_reduce and plan is self.bwd):
            dst = L.ptr_array([self.G(o) for o in offs])
            plan.keep.append(dst)
            plan.add("rua_stats_to_f32", s.ptr, s.R, Cc, dst, len(offs))
            return
        for o in offs:
            bucket = self.e.dist.bucket_of(o) if self.e.dist is not None else 0
            if self.pending and bucket != self.pending_bucket:
                self.flush_wgrad(plan)
            self.pending_bucket = bucket
            rec = L.WgradPending()
            rec.kind, rec.parts, rec.n, rec.partials, rec.dw, rec.blocks = 3, s.R, Cc, s.ptr, self.G(o), (Cc + 255) // 256
            self.pending.append((rec, o))

    def bn_bwd_finalize(self, plan: Plan, stats2, count, bn, coef: Coef):
        plan.add("rua_bn_bwd_finalize", stats2.ptr, stats2.R, float(count), self.P(bn["gamma"]), coef.mean, coef.rstd,
                 self.G(bn["gamma"]), self.G(bn["beta"]), coef.A, coef.B, coef.Cc, bn["C"])

    def bn_bwd_apply(self, plan: Plan, gs: List[Ten], coefs: List[Coef], x: Ten, out: Ten, accumulate: int,
                     dskip: Optional[Ten] = None, masked=False):
        g = L.ptr_array([t.ptr for t in gs]); A = L.ptr_array([c.A for c in coefs]); Bc = L.ptr_array([c.B for c in coefs])
        Cc = L.ptr_array([c.Cc for c in coefs]); ms = L.ptr_array([c.scale for c in coefs]); mt = L.ptr_array([c.shift for c in coefs])
        plan.keep += [g, A, Bc, Cc, ms, mt]
        plan.add("rua_bn_bwd_apply", len(gs), g, A, Bc, Cc, ms, mt, 1 if masked else 0, x.ptr,
                 dskip.ptr if dskip is not None else None, out.ptr, accumulate, x.M, x.C, self.dt)

    # -- layers: created on the first graph build, replayed (same order) on later builds ----------
    def Lconv(self, cins, cout, taps, name=None, mfma=True):
        return self.e.layer("conv", cins, cout, taps, name, mfma)

    def Lbn(self, c):
        return self.e.layer("bn", c)

    # -- composites -------------------------------------------------------------------------------
    def resblock(self, x: Ten, nf: int, dils: List[int]) -> Ten:
        """model2.py:15-34.  BN1 statistics are shared by all branches (same input)."""
        tr, F = self.training, self.fwd
        v2 = self.cfg.variant == "model2"
        scope = f"{self.block_tag}:ResBlock({nf},{dils})@{x.H}x{x.W}"
        F.scope = scope
        lay = [(self.Lbn(nf), self.Lconv([nf], nf, 9), self.Lbn(nf), self.Lconv([nf], nf, 9)) for _ in dils]
        cnt = x.M
        if tr and x.stats is None:
            x.stats = self.col_stats(F, x)
        if all(self.fused_input_ok(x, nf, d) for d in dils) or self.group_fused_ok(x, nf, dils):
            return self.resblock_fused(x, nf, dils, lay, scope)
        a1, coef1 = self.bn_fwd(F, x, [l[0] for l in lay], True, x.stats, cnt)
        y1 = [self.like(x) for _ in dils]
        st1 = [self.stat(nf, (cnt + 127) // 128) if tr else None for _ in dils]
        # the branches are independent until the final sum: their first convs go out as one grouped call
        self.conv_group(F, [self.conv_desc([(a, 0, d, 9)], l[1]["segs"], nf, self.P(l[1]["bias"]), y, stats=st)
                            for d, l, a, y, st in zip(dils, lay, a1, y1, st1)])
        out = self.like(x)
        # Second stage.  Where the library sums the branches on chip with every BatchNorm applied on load (rua_conv_fwd_sum ->
        # conv_band64 at the C = 64 level) no normalised copy of a first-conv output is written: ONE launch instead of a
        # rua_bn_fwd per branch + the concatenated conv; the weight gradients then normalise y1_b on load as well.
        sum2 = self.second_stage_sum_ok(x, y1, nf, dils, lay, out)
        if sum2:
            nb = len(dils)
            if tr and self.e.fold_bn:
                cf2 = [self.bn_fold(F, l[2], st, cnt) for l, st in zip(lay, st1)]
                coef2, fold2 = [c for c, _ in cf2], [f for _, f in cf2]
            else:
                coef2, fold2 = self.bn_coefs(F, x, [l[2] for l in lay], st1, cnt), [None] * nb
            a2 = [None] * nb
            self.conv_sum(F, [self.conv_desc([(y, 0, d, 9)], [l[3]["segs"][0]], nf, self.P(l[3]["bias"]), out, in_bn=c2, in_fold=f2,
                                             residual=x if (v2 and bi == 0) else None, accumulate=1 if bi > 0 else 0)
                              for bi, (d, l, y, c2, f2) in enumerate(zip(dils, lay, y1, coef2, fold2))])
        else:
            a2, coef2 = self.bn_fwd_group(F, [(y, l[2], st) for l, y, st in zip(lay, y1, st1)], True, cnt)
            biases = [self.P(l[3]["bias"]) for l in lay]        # the concatenated conv's bias = sum of the branches' biases
            self.conv(F, [(a, 0, d, 9) for a, d in zip(a2, dils)], [l[3]["segs"][0] for l in lay], nf, biases[0], out,
                      residual=x if v2 else None, bias_more=biases[1:])
        F.scope = None
        if not tr:
            return out

        def back():
            Bp = self.bwd
            Bp.scope = scope
            dO = out.grad
            if not v2:
                self.bias_grad(Bp, dO, [l[3]["bias"] for l in lay])      # (model2: summed by the final bn_bwd, which reads dO as the skip gradient)
            if sum2:                                           # y1_b normalised on load by the weight gradient too
                w2 = [(y, dO, l[3]["segs"][0]["off"], 1, d, 9, c2) for d, l, y, c2 in zip(dils, lay, y1, coef2)]
            else:
                w2 = [(a_2, dO, l[3]["segs"][0]["off"], 1, d, 9, None) for d, l, a_2 in zip(dils, lay, a2)]
            w2 = self.wgrad_now_or_later(Bp, w2)
            g2s = [self.like(x) for _ in dils]
            s2s = [self.stat(nf, (cnt + 127) // 128) for _ in dils]
            self.conv_group(Bp, [self.dgrad_desc(dO, self.Wd(l[3]["segs"][0]["dst"]), nf, d, 9, g2, 0, mask=(y, c2.scale, c2.shift), stats2=s2)
                                 for d, l, y, c2, g2, s2 in zip(dils, lay, y1, coef2, g2s, s2s)])
            dy1s = [self.like(x) for _ in dils]
            self.bn_bwd_group(Bp, [(g2, c2, l[2], s2, y, dy1, cnt) for l, y, c2, g2, s2, dy1 in zip(lay, y1, coef2, g2s, s2s, dy1s)])
            # no bias gradient launch: the output of a BN backward sums to zero per channel, so d b1 == 0 exactly
            self.wgrad_group(Bp, w2 + [(a_1, dy1, l[1]["segs"][0]["off"], 1, d, 9, None) for d, l, a_1, dy1 in zip(dils, lay, a1, dy1s)])
            g1s = g2s                                          # g2 is dead after its bn_bwd: reuse the storage
            s1s = [self.stat(nf, (cnt + 127) // 128) for _ in dils]
            self.conv_group(Bp, [self.dgrad_desc(dy1, self.Wd(l[1]["segs"][0]["dst"]), nf, d, 9, g1, 0, mask=(x, c1.scale, c1.shift), stats2=s1)
                                 for d, l, dy1, c1, g1, s1 in zip(dils, lay, dy1s, coef1, g1s, s1s)])
            gx, acc = self.gacc(x)
            self.bn_bwd(Bp, g1s, coef1, [l[0] for l in lay], s1s, x, gx, acc, cnt, dskip=dO if v2 else None,
                        skip_bias=[l[3]["bias"] for l in lay] if v2 else None, dx_bias=True)
            Bp.scope = None
        self.back_steps.append(back)
        return out

    def group_fused_ok(self, x: Ten, nf: int, dils: List[int]) -> bool:
        """The C = 64 level: no single convolution normalises on load there, but rua_conv_fwd_group runs the branches' first convs
        (and their data gradients) as ONE conv_band64m launch that does, rua_conv_fwd_sum the second convs (conv_band64), and the
        all-taps weight gradient normalises on load as well - then the whole ResBlock takes the normalise-on-load path."""
        if self.dry or self.dt != L.RUA_BF16 or not self.e.fuse_bn or len(dils) < 2:
            return False
        lib = L.lib()
        arr = (L.ConvDesc * len(dils))()
        dummy = L.BnFold()
        outs = [self.e.scratches[0].data_ptr() + 4096 * (i + 1) for i in range(len(dils))]       # distinct, never dereferenced
        for bi, d in enumerate(dils):
            q = arr[bi]
            q.nseg = 1
            sg = q.seg[0]
            sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = x.ptr, x.ptr, x.C, x.H, x.W, 0, d, 9
            q.N, q.H, q.W, q.Cout, q.stride, q.dtype = x.N, x.H, x.W, nf, 1, self.dt
            q.y, q.out_stride, q.OH, q.OW = outs[bi], 1, x.H, x.W
            q.in_fold, q.in_relu = C.addressof(dummy), 1
        if lib.raw("rua_conv_group_band_ok")(arr, len(dils)) != 1:
            return False
        for bi in range(len(dils)):                          # ... and the second convs as one summed launch
            arr[bi].y = outs[0]
            arr[bi].accumulate = 1 if bi > 0 else 0
        if lib.raw("rua_conv_sum_kernel")(arr, len(dils)) == 0:
            return False
        w = L.WgradDesc()
        w.a, w.C, w.Hs, w.Ws, w.dy, w.Cout, w.H, w.W = x.ptr, x.C, x.H, x.W, x.ptr, nf, x.H, x.W
        w.N, w.stride, w.taps, w.dtype = x.N, 1, 9, self.dt
        sc = self.e.scratches[0]
        w.workspace, w.workspace_bytes = sc.data_ptr(), sc.numel() * 4
        for d in dils:
            w.dil = d
            if lib.raw("rua_wgrad_kind")(C.byref(w)) != 1:
                return False
        return True

    def second_stage_sum_ok(self, x: Ten, y1: List[Ten], nf: int, dils: List[int], lay, out: Ten) -> bool:
        """Does the library run `out = x + sum_b conv(relu(BN2_b(y1_b)))` as ONE launch with the BatchNorms applied on load
        (rua_conv_sum_kernel), and the matching weight gradient with y1_b normalised on load (the all-taps kernel)?  Asked of the
        library, not guessed."""
        if self.dry or self.dt != L.RUA_BF16 or not self.e.fuse_bn or len(dils) < 1:
            return False
        arr = (L.ConvDesc * len(dils))()
        dummy = L.BnFold()
        for bi, (d, y) in enumerate(zip(dils, y1)):
            q = arr[bi]
            q.nseg = 1
            sg = q.seg[0]
            sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = y.ptr, y.ptr, y.C, y.H, y.W, 0, d, 9
            q.N, q.H, q.W, q.Cout, q.stride, q.dtype = x.N, x.H, x.W, nf, 1, self.dt
            q.y, q.out_stride, q.OH, q.OW = out.ptr, 1, x.H, x.W
            q.accumulate = 1 if bi > 0 else 0
            q.in_fold, q.in_relu = C.addressof(dummy), 1
        lib = L.lib()
        if lib.raw("rua_conv_sum_kernel")(arr, len(dils)) == 0:
            return False
        w = L.WgradDesc()
        w.a, w.C, w.Hs, w.Ws, w.dy, w.Cout, w.H, w.W = x.ptr, x.C, x.H, x.W, x.ptr, nf, x.H, x.W
        w.N, w.stride, w.taps, w.dtype = x.N, 1, 9, self.dt
        sc = self.e.scratches[0]
        w.workspace, w.workspace_bytes = sc.data_ptr(), sc.numel() * 4
        for d in dils:
            w.dil = d
            if lib.raw("rua_wgrad_kind")(C.byref(w)) != 1:
                return False
        return True

    def resblock_fused(self, x: Ten, nf: int, dils: List[int], lay, scope: str) -> Ten:
        """The same ResBlock (model2.py:15-34) with every BatchNorm + ReLU applied ON LOAD by the consuming kernel: no
        normalised copy of x or of a first-conv output is ever written.  Forward per branch: conv(relu(BN1_b(x))) -> y1_b
        (statistics in the epilogue) ; one coefficient launch for all BN2_b ; out = x + sum_b conv(relu(BN2_b(y1_b))) as one
        residual launch and accumulating launches (each reads its y1_b once).  Backward: the weight gradients read x / y1_b
        and normalise on load too; masks and BN-backward sums come out of the data-gradient epilogues as before."""
        tr, F = self.training, self.fwd
        v2 = self.cfg.variant == "model2"
        cnt = x.M
        nb = len(dils)
        fold = tr and self.e.fold_bn
        if fold:                                             # training: the convs derive the BN coefficients in their prologue
            cf1 = [self.bn_fold(F, l[0], x.stats, cnt) for l in lay]
            coef1, fold1 = [c for c, _ in cf1], [f for _, f in cf1]
        else:
            coef1, fold1 = self.bn_coefs(F, x, [l[0] for l in lay], [x.stats] * nb, cnt), [None] * nb
        y1 = [self.like(x) for _ in dils]
        st1 = [self.stat(nf, (cnt + 127) // 128) if tr else None for _ in dils]
        self.conv_group(F, [self.conv_desc([(x, 0, d, 9)], l[1]["segs"], nf, self.P(l[1]["bias"]), y, stats=st, in_bn=c1, in_fold=f1)
                            for d, l, c1, f1, y, st in zip(dils, lay, coef1, fold1, y1, st1)])       # all four dilations: ONE grid
        if fold:
            cf2 = [self.bn_fold(F, l[2], st, cnt) for l, st in zip(lay, st1)]
            coef2, fold2 = [c for c, _ in cf2], [f for _, f in cf2]
        else:
            coef2, fold2 = self.bn_coefs(F, x, [l[2] for l in lay], st1, cnt), [None] * nb
        out = self.like(x)
        # out = x + sum of the branches' second convs: ONE call, the sum kept on chip where the library has the kernel for it
        # (rua_conv_fwd_sum -> conv_band32: out written once); otherwise the members run one by one, member 0 with the residual
        self.conv_sum(F, [self.conv_desc([(y, 0, d, 9)], [l[3]["segs"][0]], nf, self.P(l[3]["bias"]), out, in_bn=c2, in_fold=f2,
                                         residual=x if (v2 and bi == 0) else None, accumulate=1 if bi > 0 else 0)
                          for bi, (d, l, y, c2, f2) in enumerate(zip(dils, lay, y1, coef2, fold2))])
        F.scope = None
        if not tr:
            return out

        def back():
            Bp = self.bwd
            Bp.scope = scope
            dO = out.grad
            if not v2:
                self.bias_grad(Bp, dO, [l[3]["bias"] for l in lay])
            w2 = self.wgrad_now_or_later(Bp, [(y, dO, l[3]["segs"][0]["off"], 1, d, 9, c2) for d, l, y, c2 in zip(dils, lay, y1, coef2)])
            g2s = [self.like(x) for _ in dils]
            s2s = [self.stat(nf, (cnt + 127) // 128) for _ in dils]
            self.conv_group(Bp, [self.dgrad_desc(dO, self.Wd(l[3]["segs"][0]["dst"]), nf, d, 9, g2, 0, mask=(y, c2.scale, c2.shift), stats2=s2)
                                 for d, l, y, c2, g2, s2 in zip(dils, lay, y1, coef2, g2s, s2s)])
            dy1s = [self.like(x) for _ in dils]
            self.bn_bwd_group(Bp, [(g2, c2, l[2], s2, y, dy1, cnt) for l, y, c2, g2, s2, dy1 in zip(lay, y1, coef2, g2s, s2s, dy1s)])
            self.wgrad_group(Bp, w2 + [(x, dy1, l[1]["segs"][0]["off"], 1, d, 9, c1) for d, l, c1, dy1 in zip(dils, lay, coef1, dy1s)])
            g1s = g2s
            s1s = [self.stat(nf, (cnt + 127) // 128) for _ in dils]
            self.conv_group(Bp, [self.dgrad_desc(dy1, self.Wd(l[1]["segs"][0]["dst"]), nf, d, 9, g1, 0, mask=(x, c1.scale, c1.shift), stats2=s1)
                                 for d, l, dy1, c1, g1, s1 in zip(dils, lay, dy1s, coef1, g1s, s1s)])
            gx, acc = self.gacc(x)
            self.bn_bwd(Bp, g1s, coef1, [l[0] for l in lay], s1s, x, gx, acc, cnt, dskip=dO if v2 else None,
                        skip_bias=[l[3]["bias"] for l in lay] if v2 else None, dx_bias=True)
            Bp.scope = None
        self.back_steps.append(back)
        return out

    def down(self, x: Ten, nf: int) -> Ten:
        """Conv2D(nf,(1,1),strides=(2,2)) model2.py:103-111: samples pixels 0,2,4,..; no BN, no activation."""
        F, tr = self.fwd, self.training
        lay = self.Lconv([x.C], nf, 1)
        y = self.new(x.N, x.H // 2, x.W // 2, nf)
        st = self.stat(nf, (y.M + 127) // 128) if tr else None
        self.conv(F, [(x, 0, 1, 1)], lay["segs"], nf, self.P(lay["bias"]), y, stride=2, stats=st)
        y.stats = st
        y.bias_offs = [lay["bias"]]                          # the ResBlock behind it writes y's whole gradient: it may take the bias gradient on the way
        if tr:
            def back():
                Bp = self.bwd
                dy = y.grad
                if not y.bias_done:
                    self.bias_grad(Bp, dy, [lay["bias"]])
                self.wgrad(Bp, x, dy, lay["segs"][0]["off"], 2, 1, 1)
                gx, acc = self.gacc(x)
                if not acc:
                    Bp.add("rua_fill_zero", gx.ptr, gx.t.numel() * gx.t.element_size())
                self.dgrad(Bp, dy, self.Wd(lay["segs"][0]["dst"]), x.C, 1, 1, gx, 1, out_stride=2)
            self.back_steps.append(back)
        return y

    def bn_node(self, x: Ten, bn, relu: bool, count=None, bessel=None, stats=None, defer: Optional[List] = None):
        """y = [relu](BN(x)) materialised; returns (y, coef, backward(fused: bool)).
        If the single consumer's dgrad wrote g (masked) and statistics itself, `fused` skips both."""
        F, tr = self.fwd, self.training
        cnt = count or x.M
        if tr and stats is None:
            stats = x.stats if x.stats is not None else self.col_stats(F, x)
        ys, coefs = self.bn_fwd(F, x, [bn], relu, stats, cnt, bessel, defer=defer, out_stats=True)     # defer: the caller issues several as one group launch
        y, coef = ys[0], coefs[0]
        node = dict(x=x, y=y, coef=coef, bn=bn, relu=relu, cnt=cnt, s2=None, fused=False, s2_out=False)
        if not relu and tr:
            y.bn_src = node

        def back(defer=None):
            Bp = self.bwd
            g = y.grad
            if not node["fused"] and not node["s2_out"]:
                node["s2"] = self.stat(x.C, self.stat_blocks(x), burst=True)
                Bp.add("rua_col_stats2", g.ptr, x.ptr, coef.scale, coef.shift, 1 if relu else 0, x.M, x.C, node["s2"].ptr, node["s2"].R, self.dt)
            gx, acc = self.gacc(x)
            self.bn_bwd(Bp, [g], [coef], [bn], [node["s2"]], x, gx, acc, cnt, masked=(relu and not node["fused"]), defer=defer,
                        stats2_out=[node["s2_out"]])
        node["back"] = back
        return y, node

    def fuse_target(self, node):
        """dgrad epilogue arguments that make `node`'s backward fused (single consumer only)."""
        node["fused"] = True
        node["s2"] = self.stat(node["x"].C, (node["x"].M + 127) // 128)
        g, _ = self.gacc(node["y"])
        if node["relu"]:
            return dict(out=g, mask=(node["x"], node["coef"].scale, node["coef"].shift), stats2=node["s2"])
        return dict(out=g, stat_aux=node["x"], stats2=node["s2"])

    def conv1x1_multi(self, segs, cout, out_hw, want_stats=True, defer: Optional[List] = None):
        """1x1 conv over concatenated sources [(Ten, up_shift)] -> raw output (with statistics).  defer: the descriptor is appended to this list
        instead of being launched (the caller issues independent convolutions as one group: Graph.conv_group)."""
        F, tr = self.fwd, self.training
        lay = self.Lconv([t.C for t, _ in segs], cout, 1)
        y = self.new(self.B, out_hw[0], out_hw[1], cout)
        st = self.stat(cout, (y.M + 127) // 128) if (tr and want_stats) else None
        if defer is not None:
            defer.append(self.conv_desc([(t, up, 1, 1) for t, up in segs], lay["segs"], cout, self.P(lay["bias"]), y, stats=st))
        else:
            self.conv(F, [(t, up, 1, 1) for t, up in segs], lay["segs"], cout, self.P(lay["bias"]), y, stats=st)
        y.stats = st
        return y, lay

    def conv1x1_multi_back(self, Bp, segs, lay, y: Ten, targets, bias: bool = False):
        """Backward of conv1x1_multi.  targets[i] = dict(out, mask/stat_aux/stats2) for a fused BN source,
        or None for a plain accumulate into the source's gradient.  In the model2 graph every conv1x1_multi is
        followed by a training-mode BatchNorm, so its bias gradient (the per-channel sum of a BN backward output) is
        exactly zero and no launch is spent on it; the model.py graph has no BN there and asks for it (bias=True)."""
        dy = y.grad
        if bias:
            self.bias_grad(Bp, dy, [lay["bias"]])
        pooled = {0: dy}
        ups = sorted({up for _, up in segs})
        if ups == [0, 1, 2, 3] and dy.H % 8 == 0 and dy.W % 8 == 0 and self.e.pool_pyramid:      # the PSP fuse conv: 2 / 4 / 8 in one pass over dy
            for up in (1, 2, 3):
                pooled[up] = self.new(dy.N, dy.H >> up, dy.W >> up, dy.C)
            Bp.add("rua_sumpool_pyramid", dy.ptr, pooled[1].ptr, pooled[2].ptr, pooled[3].ptr, dy.N, dy.H, dy.W, dy.C, self.dt)
        for (t, up), seg, tg in zip(segs, lay["segs"], targets):
            if up not in pooled:
                k = 1 << up
                pd = self.new(dy.N, dy.H // k, dy.W // k, dy.C)
                Bp.add("rua_sumpool", dy.ptr, pd.ptr, dy.N, dy.H, dy.W, dy.C, k, self.dt)
                pooled[up] = pd
        grouped = len(segs) > 1 and not self.dry and self.e.group_1x1 and self.dt == L.RUA_BF16
        ddescs = []
        # the per-source weight gradients: one grid where every member is a wgrad_pw launch (round 5), else one launch each (as a generic group these
        # unequal, tiny weight gradients measured slower - 18 vs 12 us)
        self.wgrad_pw_group(Bp, [(t, pooled[up], seg["off"], 1, 1, 1) for (t, up), seg in zip(segs, lay["segs"])])
        for (t, up), seg, tg in zip(segs, lay["segs"], targets):
            d = pooled[up]
            if tg is None:
                gx, acc = self.gacc(t)
                dd = self.dgrad_desc(d, self.Wd(seg["dst"]), t.C, 1, 1, gx, acc)
            else:
                dd = self.dgrad_desc(d, self.Wd(seg["dst"]), t.C, 1, 1, tg["out"], 0, tg.get("mask"), tg.get("stats2"), tg.get("stat_aux"))
            if grouped:
                ddescs.append(dd)
            else:
                Bp.keep.append(dd)
                Bp.add("rua_conv_fwd", C.byref(dd))
        if grouped:
            # the sources of a concatenating 1x1 conv have their own gradients: the data gradients of all sources as one group (members on the same
            # kernel share a grid: the pooled PSPPooling branches are ~10 us of latency apiece)
            self.conv_group(Bp, ddescs)

    def psp(self, x: Ten, nf: int) -> Ten:
        """PSPPooling + the ReLU the caller applies (model2.py:41-79,116,142): max-pool k -> [nearest up] ->
        1x1 conv + BN per branch, concat with the input, 1x1 conv + BN, ReLU.  The pooled branches are
        convolved and normalised at pooled resolution (conv and BN statistics commute with replication)
        and the upsample is folded into the fuse conv's read."""
        F, tr = self.fwd, self.training
        w_in = self.cfg.input_shape[1]
        ks = [1, 2] + ([4] if w_in >= 128 else []) + ([8] if w_in >= 256 else [])
        pooled, idxs = [], []
        pyramid = ks == [1, 2, 4, 8] and x.H % 8 == 0 and x.W % 8 == 0 and self.e.pool_pyramid     # one pass for the three poolings
        for k in ks:
            if k == 1:
                pooled.append(x); idxs.append(None)
            else:
                assert x.H % k == 0 and x.W % k == 0, f"PSP pool {k} does not divide {x.H}x{x.W}"
                p = self.new(x.N, x.H // k, x.W // k, x.C)
                idx = self.alloc((p.t.numel(),), torch.uint8)
                F.keep.append(idx)
                if not pyramid:
                    F.add("rua_maxpool_fwd", x.ptr, p.ptr, idx.data_ptr(), x.N, x.H, x.W, x.C, k, self.dt)
                pooled.append(p); idxs.append(idx)
        if pyramid:                                            # x is read once (k = 2); the 4- and 8-windows come from the level below
            F.add("rua_maxpool_fwd", x.ptr, pooled[1].ptr, idxs[1].data_ptr(), x.N, x.H, x.W, x.C, 2, self.dt)
            F.add("rua_maxpool_derive", pooled[1].ptr, idxs[1].data_ptr(), pooled[2].ptr, idxs[2].data_ptr(), x.N, x.H // 2, x.W // 2, x.C, 2, self.dt)
            F.add("rua_maxpool_derive", pooled[2].ptr, idxs[2].data_ptr(), pooled[3].ptr, idxs[3].data_ptr(), x.N, x.H // 4, x.W // 4, x.C, 4, self.dt)
        # Keras creates the branch Conv2DN layers (conv, bn) in order, then the fuse Conv2DN
        br, bds = [], []
        group_bn = not self.dry and self.dt == L.RUA_BF16      # the branches' BatchNorms (equal channels, unequal pixel counts) as ONE launch
        zs = []
        bdescs = [] if (not self.dry and self.e.group_1x1 and self.dt == L.RUA_BF16) else None       # the branch convolutions as ONE group
        for k, p in zip(ks, pooled):
            z, lay = self.conv1x1_multi([(p, 0)], nf // 4, (p.H, p.W), defer=bdescs)
            bn = self.Lbn(nf // 4)
            zs.append((k, p, z, lay, bn))
            if not group_bn:
                zb, node = self.bn_node(z, bn, False, count=z.M, bessel=z.M * k * k, stats=z.stats)
                br.append((k, p, z, lay, zb, node))
        if bdescs:                                             # (only with group_bn: the BatchNorms follow as one launch)
            self.conv_group(F, bdescs)
        if group_bn:
            for k, p, z, lay, bn in zs:
                zb, node = self.bn_node(z, bn, False, count=z.M, bessel=z.M * k * k, stats=z.stats, defer=bds)
                br.append((k, p, z, lay, zb, node))
            self.issue_bn_fwd(F, bds)
        segs = [(b[4], int(math.log2(b[0]))) for b in br] + [(x, 0)]
        zf, layf = self.conv1x1_multi(segs, nf, (x.H, x.W))
        bnf = self.Lbn(nf)
        out, nodef = self.bn_node(zf, bnf, True, stats=zf.stats)
        if tr:
            def back():
                Bp = self.bwd
                nodef["back"]()                                    # out.grad -> zf.grad (unfused: several consumers)
                targets = [self.fuse_target(b[5]) for b in br] + [None]
                self.conv1x1_multi_back(Bp, segs, layf, zf, targets)
                bbs = [] if (group_bn and all(b_[5]["fused"] for b_ in br)) else None
                if bbs is not None:                                # every zb.grad is there: the four BatchNorm backwards as one launch
                    for b_ in br:
                        b_[5]["back"](defer=bbs)
                    self.issue_bn_bwd(Bp, bbs)
                if bbs is not None and pyramid and bdescs is not None:
                    # the branch convolutions' data gradients as one group (independent: own source gradients)
                    self.wgrad_pw_group(Bp, [(p, z.grad, lay["segs"][0]["off"], 1, 1, 1) for (k, p, z, lay, zb, node) in br])
                    dd = []
                    for (k, p, z, lay, zb, node) in br:
                        gx, acc = self.gacc(p)
                        dd.append(self.dgrad_desc(z.grad, self.Wd(lay["segs"][0]["dst"]), p.C, 1, 1, gx, acc))
                    self.conv_group(Bp, dd)
                else:
                    for (k, p, z, lay, zb, node), idx in zip(br, idxs):
                        if bbs is None:
                            node["back"]()                             # zb.grad -> z.grad
                        self.conv1x1_multi_back(Bp, [(p, 0)], lay, z, [None])   # -> p.grad (p is x for k == 1)
                        if k > 1 and not pyramid:
                            gx, acc = self.gacc(x)
                            Bp.add("rua_maxpool_bwd", p.grad.ptr, idx.data_ptr(), gx.ptr, acc, x.N, x.H, x.W, x.C, k, self.dt)
                if pyramid:                                        # the three pooled gradients scattered in ONE pass over x.grad
                    gx, acc = self.gacc(x)
                    dys = L.ptr_array([b_[1].grad.ptr for b_ in br[1:]]); ixs = L.ptr_array([i.data_ptr() for i in idxs[1:]])
                    kk = (C.c_int32 * 3)(2, 4, 8)
                    Bp.keep += [dys, ixs, kk]
                    Bp.add("rua_maxpool_bwd_multi", 3, dys, ixs, kk, gx.ptr, acc, x.N, x.H, x.W, x.C, self.dt)
            self.back_steps.append(back)
        return out

    def up_combine(self, x: Ten, skip: Ten, nf: int) -> Ten:
        """UpSampling(x, nf/2) then combine(., skip, nf)  (model2.py:81-94): nearest x2 -> 1x1 conv -> BN, then
        ReLU || skip -> 1x1 conv -> BN.  The 1x1 conv and its BN run at LOW resolution (exact: conv and BN
        batch statistics commute with nearest replication) and the x2 is folded into the combine conv's read."""
        F, tr = self.fwd, self.training
        z, lay_u = self.conv1x1_multi([(x, 0)], nf // 2, (x.H, x.W))
        bn_u = self.Lbn(nf // 2)
        a, node_u = self.bn_node(z, bn_u, True, count=z.M, bessel=z.M * 4, stats=z.stats)
        segs = [(a, 1), (skip, 0)]
        zc, lay_c = self.conv1x1_multi(segs, nf, (skip.H, skip.W))
        bn_c = self.Lbn(nf)
        t, node_c = self.bn_node(zc, bn_c, False, stats=zc.stats)
        if tr:
            def back():
                Bp = self.bwd
                node_c["back"]()
                self.conv1x1_multi_back(Bp, segs, lay_c, zc, [self.fuse_target(node_u), None])
                node_u["back"]()
                self.conv1x1_multi_back(Bp, [(x, 0)], lay_u, z, [None])
            self.back_steps.append(back)
        return t

    def final_combine(self, x: Ten, c1: Ten, nf: int) -> Ten:
        """x_comb = combine(x, c1, 32)  (model2.py:140): ReLU(x) || c1 -> 1x1 conv -> BN."""
        F, tr = self.fwd, self.training
        r = self.like(x)
        F.add("rua_relu", x.ptr, r.ptr, x.t.numel(), self.dt)
        segs = [(r, 0), (c1, 0)]
        zc, lay = self.conv1x1_multi(segs, nf, (x.H, x.W))
        bn = self.Lbn(nf)
        t, node = self.bn_node(zc, bn, False, stats=zc.stats)
        if tr:
            def back():
                Bp = self.bwd
                node["back"]()
                gx, acc = self.gacc(x)
                assert acc == 0
                self.conv1x1_multi_back(Bp, segs, lay, zc, [dict(out=gx, mask=(x, None, None)), None])
            self.back_steps.append(back)
        return t

    # -- ResUnet_a/model.py graph: no BN around the 1x1 convs, no ReLU after PSP, no skip term in ResBlock ---------
    def psp_v1(self, x: Ten, nf: int) -> Ten:
        """PSPPooling of model.py:35-64: max-pool k -> 1x1 conv (bias, no BN) -> nearest up k; concat with the input;
        1x1 conv.  The convs already run at pooled resolution in the reference; the upsample is folded into the
        fuse conv's read."""
        F, tr = self.fwd, self.training
        w_in = self.cfg.input_shape[1]
        ks = [1, 2] + ([4] if w_in >= 128 else []) + ([8] if w_in >= 256 else [])
        br = []
        for k in ks:
            p, idx = x, None
            if k > 1:
                assert x.H % k == 0 and x.W % k == 0, f"PSP pool {k} does not divide {x.H}x{x.W}"
                p = self.new(x.N, x.H // k, x.W // k, x.C)
                idx = self.alloc((p.t.numel(),), torch.uint8)
                F.keep.append(idx)
                F.add("rua_maxpool_fwd", x.ptr, p.ptr, idx.data_ptr(), x.N, x.H, x.W, x.C, k, self.dt)
            br.append([k, p, idx])
        for b in br:                                             # Keras creates the branch convs after all the pools
            z, lay = self.conv1x1_multi([(b[1], 0)], nf // 4, (b[1].H, b[1].W), want_stats=False)
            b += [z, lay]
        segs = [(b[3], int(math.log2(b[0]))) for b in br] + [(x, 0)]
        zf, layf = self.conv1x1_multi(segs, nf, (x.H, x.W), want_stats=False)
        if tr:
            def back():
                Bp = self.bwd
                self.conv1x1_multi_back(Bp, segs, layf, zf, [None] * len(segs), bias=True)
                for k, p, idx, z, lay in br:
                    self.conv1x1_multi_back(Bp, [(p, 0)], lay, z, [None], bias=True)     # -> p.grad (p is x for k == 1)
                    if k > 1:
                        gx, acc = self.gacc(x)
                        Bp.add("rua_maxpool_bwd", p.grad.ptr, idx.data_ptr(), gx.ptr, acc, x.N, x.H, x.W, x.C, k, self.dt)
            self.back_steps.append(back)
        return zf

    def relu_cat_conv_v1(self, z: Ten, up: int, skip: Ten, nf: int, want_stats: bool) -> Ten:
        """combine() of model.py:66-70: ReLU(z) [nearest x2^up, folded into the read] || skip -> 1x1 conv (bias, no BN)."""
        F, tr = self.fwd, self.training
        r = self.like(z)
        F.add("rua_relu", z.ptr, r.ptr, z.t.numel(), self.dt)
        segs = [(r, up), (skip, 0)]
        zc, lay = self.conv1x1_multi(segs, nf, (skip.H, skip.W), want_stats=want_stats)
        if tr:
            def back():
                gz, acc = self.gacc(z)
                assert acc == 0
                self.conv1x1_multi_back(self.bwd, segs, lay, zc, [dict(out=gz, mask=(z, None, None)), None], bias=True)
            self.back_steps.append(back)
        return zc

    def up_combine_v1(self, x: Ten, skip: Ten, nf: int) -> Ten:
        """model.py:93-95: Conv2D(nf,(1,1)) at low resolution -> UpSampling2D -> combine(., skip, nf)."""
        z, lay_u = self.conv1x1_multi([(x, 0)], nf, (x.H, x.W), want_stats=False)
        if self.training:
            self.back_steps.append(lambda: self.conv1x1_multi_back(self.bwd, [(x, 0)], lay_u, z, [None], bias=True))
        return self.relu_cat_conv_v1(z, 1, skip, nf, want_stats=True)

    def conv3x3_relu(self, x: Ten, nf: int, name=None):
        """ZeroPadding2D(1)+Conv2D(32,(3,3),relu,valid) of the heads (model2.py:153-158) == same-pad 3x3 + ReLU.
        The ReLU's backward mask is applied by whoever writes y.grad (the next conv's data gradient or the head's), so no
        separate masking pass over dy runs; it is launched only if some writer could not."""
        F, tr = self.fwd, self.training
        lay = self.Lconv([x.C], nf, 9, name=name)
        y = self.new(x.N, x.H, x.W, nf)
        y.relu_out = True
        self.conv(F, [(x, 0, 1, 9)], lay["segs"], nf, self.P(lay["bias"]), y, out_relu=True)
        y.bias_offs = [lay["bias"]]                          # the single consumer writes y's whole (masked) gradient: it may take the bias gradient on the way
        if tr:
            def back():
                Bp = self.bwd
                dy = y.grad
                assert not (y.masked_w and y.plain_w), "mixed masked / unmasked writers of a ReLU output's gradient"
                if y.plain_w:
                    assert not y.bias_done
                    Bp.add("rua_relu_mask", dy.ptr, y.ptr, y.t.numel(), self.dt)
                if not y.bias_done:
                    self.bias_grad(Bp, dy, [lay["bias"]])
                self.wgrad(Bp, x, dy, lay["segs"][0]["off"], 1, 1, 9)
                gx, acc = self.gacc(x, masked=x.relu_out)
                sx = None
                if x.bias_offs and x.relu_out and not acc and not self.dry and self.e.bn_dx_bias:
                    # x is the ReLU'ed output of a conv with a bias and this data gradient its only consumer: the epilogue's per-channel sums of the
                    # masked gradient are that bias gradient (no rua_col_stats pass)
                    sx = self.stat(x.C, (x.M + 127) // 128)
                self.dgrad(Bp, dy, self.Wd(lay["segs"][0]["dst"]), x.C, 1, 9, gx, acc, mask=(x, None, None) if x.relu_out else None, stats2=sx)
                if sx is not None:
                    self.stats_to_grads(Bp, sx, x.C, x.bias_offs)
                    x.bias_done = True
            self.back_steps.append(back)
        return y

    def head(self, x: Ten, cout: int, act: int, hname: str, name=None, lay=None):
        """Conv2D(C,(1,1)) + softmax/sigmoid, its loss and the gradient w.r.t. the logits."""
        F, tr = self.fwd, self.training
        lay = lay if lay is not None else self.Lconv([x.C], cout, 1, name=name, mfma=False)
        z = self.new(x.N, x.H, x.W, cout, f32=True)
        p = self.new(x.N, x.H, x.W, cout, f32=True)
        y = self.new(x.N, x.H, x.W, cout, f32=True)           # label buffer (host uploads into it)
        h = dict(name=hname, x=x, z=z, p=p, y=y, lay=lay, act=act, C=cout, slot=len(self.heads))
        sp = self.e.loss
        tani = sp is not None and sp.kind.get(hname) == L.LOSS_TANIMOTO
        if sp is not None and not self.dry and self.e.fuse_head_loss and (tani or hname == "seg"):
            # the Tanimoto moments (and the 'seg' head's accuracy / confusion counts) in the head's own epilogue: p is not read again
            if tani:
                # SUMS_REPLICAS copies of sums[B][C][6] (+ the slot rua_tanimoto_finalize_rep folds them into): a block of the head's forward adds
                # into one copy, so the ~64 blocks of a sample do not queue up on the same 36 fp64 addresses at the end of the launch
                h["sums_rep"] = SUMS_REPLICAS
                h["sums"] = self.salloc((SUMS_REPLICAS + 1) * x.N * cout * 6)
            if hname == "seg":
                h["metrics_done"] = True
            F.add("rua_head_fwd_loss_rep", x.ptr, self.P(lay["segs"][0]["off"]), self.P(lay["bias"]), z.ptr, p.ptr, y.ptr, h.get("sums"),
                  h.get("sums_rep", 1), self.e.scalars_ptr + 8 * 8 if hname == "seg" else None, x.N, x.H * x.W, x.C, cout, act, self.dt)
        else:
            F.add("rua_head_fwd", x.ptr, self.P(lay["segs"][0]["off"]), self.P(lay["bias"]), z.ptr, p.ptr, x.M, x.C, cout, act, self.dt)
        self.heads.append(h)
        return h

    def heads_grouped(self, x_psp: Ten, x_comb: Ten, w0: int, Cc: int):
        """The multitask heads (model2.py:148-191) with their five 3x3 + ReLU convolutions issued as GROUPS across the heads - forward {seg1, bound1,
        dist1} and {seg2, dist2}, backward the weight gradients and the data gradients of {seg2, bound1, dist2}, then of {seg1, dist1} - instead of
        one launch per convolution: members on the same kernel form share one grid of persistent blocks (conv_strip32s_g / wgrad_rows32_g), which
        runs a 9.66 GFLOP member in ~14 us where a launch of its own takes 17 - 22.  Layers are created in the order of the head-by-head path
        (parameter layout and Keras names unchanged); same kernels, same arithmetic per convolution."""
        F, tr = self.fwd, self.training
        lay = {}
        lay["seg1"] = self.Lconv([x_psp.C], w0, 9, name="seg1"); lay["seg2"] = self.Lconv([w0], w0, 9, name="seg2")
        lay["seg3"] = self.Lconv([w0], Cc, 1, name="seg3", mfma=False)
        lay["bound1"] = self.Lconv([x_psp.C], w0, 9); lay["bound3"] = self.Lconv([w0], Cc, 1, mfma=False)
        lay["dist1"] = self.Lconv([x_comb.C], w0, 9); lay["dist2"] = self.Lconv([w0], w0, 9); lay["dist3"] = self.Lconv([w0], Cc, 1, mfma=False)
        lay["color"] = self.Lconv([x_comb.C], 3, 1, name="color", mfma=False)

        def out_of(x: Ten, ly) -> Ten:
            y = self.new(x.N, x.H, x.W, w0)
            y.relu_out = True
            y.bias_offs = [ly["bias"]]
            return y

        def fdesc(x: Ten, ly, y: Ten):
            return self.conv_desc([(x, 0, 1, 9)], ly["segs"], w0, self.P(ly["bias"]), y, out_relu=True)
        s1, b1, d1 = out_of(x_psp, lay["seg1"]), out_of(x_psp, lay["bound1"]), out_of(x_comb, lay["dist1"])
        self.cur_tag = F.scope = "heads_conv"
        self.conv_group(F, [fdesc(x_psp, lay["seg1"], s1), fdesc(x_psp, lay["bound1"], b1), fdesc(x_comb, lay["dist1"], d1)])
        s2, d2 = out_of(s1, lay["seg2"]), out_of(d1, lay["dist2"])
        self.conv_group(F, [fdesc(s1, lay["seg2"], s2), fdesc(d1, lay["dist2"], d2)])
        self.cur_tag = F.scope = None
        if tr:
            def phase(convs, carried=(), later=False):      # [(x, y, layer)]: weight gradients, then data gradients, of independent convolutions
                Bp = self.bwd
                for x, y, ly in convs:
                    assert not (y.masked_w and y.plain_w), "mixed masked / unmasked writers of a ReLU output's gradient"
                    if y.plain_w:
                        assert not y.bias_done
                        Bp.add("rua_relu_mask", y.grad.ptr, y.ptr, y.t.numel(), self.dt)
                    if not y.bias_done:
                        self.bias_grad(Bp, y.grad, [ly["bias"]])
                wg = list(carried) + [(x, y.grad, ly["segs"][0]["off"], 1, 1, 9, None) for x, y, ly in convs]
                if not later:                               # (later: these ride in the next phase's group - as the two groups of a ResBlock do, wgrad_now_or_later)
                    self.wgrad_group(Bp, wg)
                descs, sums = [], []
                for x, y, ly in convs:
                    gx, acc = self.gacc(x, masked=x.relu_out)
                    sx = None
                    if x.bias_offs and x.relu_out and not acc and not self.dry and self.e.bn_dx_bias:
                        sx = self.stat(x.C, (x.M + 127) // 128)     # the epilogue's sums of the masked gradient = the bias gradient of the conv behind x
                        sums.append((sx, x))
                    descs.append(self.dgrad_desc(y.grad, self.Wd(ly["segs"][0]["dst"]), x.C, 1, 9, gx, acc, mask=(x, None, None) if x.relu_out else None, stats2=sx))
                self.conv_group(Bp, descs)
                for sx, x in sums:
                    self.stats_to_grads(Bp, sx, x.C, x.bias_offs)
                    x.bias_done = True
                return wg if later else []

            def back():
                Bp = self.bwd
                Bp.scope = "heads_conv"
                merge = (not self.dry) and 1 in self.e.merge_wgrad            # (1: the heads)
                wg = phase([(s1, s2, lay["seg2"]), (x_psp, b1, lay["bound1"]), (d1, d2, lay["dist2"])], later=merge)
                phase([(x_psp, s1, lay["seg1"]), (x_comb, d1, lay["dist1"])], carried=wg)
                Bp.scope = None
            self.back_steps.append(back)
        self.tagged("head_seg", self.head, s2, Cc, L.ACT_SOFTMAX, "seg", "seg3", lay["seg3"])
        self.tagged("head_bound", self.head, b1, Cc, L.ACT_SIGMOID, "bound", None, lay["bound3"])
        self.tagged("head_dist", self.head, d2, Cc, L.ACT_SOFTMAX, "dist", None, lay["dist3"])
        self.tagged("head_color", self.head, x_comb, 3, L.ACT_SIGMOID, "color", "color", lay["color"])

    def head_loss(self, h):
        """Loss value (fp64 scalar slot) and, in training, d(total)/d(logits) -> head weights and x.grad."""
        sp, LP = self.e.loss, self.loss_plan
        kind, wgt = sp.kind[h["name"]], sp.weight[h["name"]]
        B, HW, Cc = self.B, h["x"].H * h["x"].W, h["C"]
        M = B * HW
        slot = self.e.scalars_ptr + h["slot"] * 8
        coef = None
        if kind == L.LOSS_TANIMOTO:
            coef = self.alloc((B * Cc * 3,), torch.float32, zero=True)
            LP.keep.append(coef)
            sums = h.get("sums")
            if sums is None:                                # (the head's forward did not take the moments itself)
                sums = self.salloc(B * Cc * 6)
                LP.add("rua_tanimoto_sums", h["p"].ptr, h["y"].ptr, B, HW, Cc, sums)
            if self.multi_head:                             # one launch for all heads (_record_losses)
                th = L.TaniHead()
                th.sums, th.replicas, th.B, th.C, th.grad_scale, th.loss_out, th.coef, th.per_sample = sums, h.get("sums_rep", 1), B, Cc, wgt / B, slot, coef.data_ptr(), None
                self._tani_multi.append(th)
            else:
                LP.add("rua_tanimoto_finalize_rep", sums, h.get("sums_rep", 1), B, HW, Cc, wgt / B, slot, coef.data_ptr(), None)
            h["norm"] = 1.0
        else:
            LP.add("rua_pixel_loss", kind, h["p"].ptr, h["z"].ptr, h["y"].ptr, self.e.class_w_ptr, M, Cc, slot, None)
            h["norm"] = 1.0 / M
        if not self.training:
            return

        dz = self.new(B, h["x"].H, h["x"].W, Cc, f32=True)
        gs = wgt / B if kind == L.LOSS_TANIMOTO else wgt / M
        if self.multi_head:
            dh = L.DzHead()
            dh.kind, dh.act, dh.p, dh.y, dh.coef, dh.class_w = kind, h["act"], h["p"].ptr, h["y"].ptr, coef.data_ptr() if coef is not None else None, self.e.class_w_ptr
            dh.grad_scale, dh.B, dh.HW, dh.C, dh.dz = gs, B, HW, Cc, dz.ptr
            self._dz_multi.append(dh)

        def back():
            Bp = self.bwd
            if self.multi_head:
                if not self._dz_emitted:                    # the first backward step of the loss section: d(loss)/d(logits) of ALL heads in one launch
                    arr = (L.DzHead * len(self._dz_multi))(*self._dz_multi)
                    Bp.keep.append(arr)
                    Bp.add("rua_head_dz_multi", arr, len(self._dz_multi))
                    self._dz_emitted = True
            else:
                Bp.add("rua_head_dz", kind, h["act"], h["p"].ptr, h["y"].ptr, coef.data_ptr() if coef is not None else None,
                       self.e.class_w_ptr, gs, B, HW, Cc, dz.ptr)
            hx = h["x"]
            gx, acc = self.gacc(hx, masked=hx.relu_out)
            lay = h["lay"]
            dxsum = None
            if hx.bias_offs and hx.relu_out and not acc and not self.dry and self.e.bn_dx_bias:
                dxsum = self.G(hx.bias_offs[0])            # the head's backward sums the masked gradient it writes: the bias gradient of the conv behind hx
                hx.bias_done = True
            Bp.add("rua_head_bwd_sums", hx.ptr, dz.ptr, self.P(lay["segs"][0]["off"]), gx.ptr, acc, self.G(lay["segs"][0]["off"]),
                   self.G(lay["bias"]), dxsum, self.e.scratch.data_ptr(), self.e.scratch.numel() * 4, M, hx.C, Cc, self.dt,
                   1 if hx.relu_out else 0)
        self.back_steps.append(back)

    # -- whole network ---------------------------------------------------------------------------------
    def tagged(self, tag: str, fn, *args):
        """Record fn(*args) - forward launches now, its backward closures later - under the composite name `tag`."""
        self.cur_tag = tag
        prev, self.fwd.scope = self.fwd.scope, tag
        out = fn(*args)
        self.fwd.scope = prev
        self.cur_tag = None
        return out

    def _build(self):
        cfg, F, tr = self.cfg, self.fwd, self.training
        if cfg.variant not in ("model2", "model"):
            raise ValueError(f"unknown graph variant {cfg.variant!r} (model2 = ResUnet_a/model2.py, model = ResUnet_a/model.py)")
        v2 = cfg.variant == "model2"
        H, W, Cin = cfg.input_shape
        lv = cfg.levels()
        w0 = lv[0][0]
        self.heads: List[dict] = []
        self.x_in = self.new(self.B, H, W, Cin, f32=True)
        stem = self.Lconv([Cin], w0, 1, mfma=False)
        c1 = self.new(self.B, H, W, w0)
        self.cur_tag, F.scope = "stem", "stem"
        # bf16 storage, <= 7 bands: the stem's weight gradient on the matrix pipe - the forward also writes the input as bf16 hi | lo | 1 (xpack), the backward is
        # the 1x1 weight gradient of dy against it + rua_stem_bwd_fold (include/rua_hip.h)
        pack = tr and not self.dry and self.e.stem_mfma and self.e.dtype == "bf16" and Cin <= 7 and w0 % 16 == 0 and c1.M >= 2048
        xpack = self.new(self.B, H, W, 16) if pack else None
        if tr and getattr(self.e, "stem_stats", True):
            # the statistics the first BatchNorm of the encoder needs, from the kernel that writes the tensor (no rua_col_stats pass over it)
            c1.stats = self.stat(w0, 2 * 256, burst=True)
        if pack:
            st = getattr(c1, "stats", None)
            F.add("rua_stem_fwd_pack", self.x_in.ptr, self.P(stem["segs"][0]["off"]), self.P(stem["bias"]), c1.ptr, c1.M, Cin, w0, self.dt,
                  st.ptr if st is not None else None, st.R if st is not None else 1, xpack.ptr)
        elif tr and getattr(self.e, "stem_stats", True):
            F.add("rua_stem_fwd_stats", self.x_in.ptr, self.P(stem["segs"][0]["off"]), self.P(stem["bias"]), c1.ptr, c1.M, Cin, w0, self.dt, c1.stats.ptr, c1.stats.R)
        else:
            F.add("rua_stem_fwd", self.x_in.ptr, self.P(stem["segs"][0]["off"]), self.P(stem["bias"]), c1.ptr, c1.M, Cin, w0, self.dt)
        if tr:
            def stem_back():
                Bp = self.bwd
                if pack:
                    tmp = self.alloc((w0 * 16,), torch.float32, zero=True)          # [Cout][16]: zero before every launch (the fold leaves it so)
                    d = self.wgrad_desc(Bp, xpack, c1.grad, stem["segs"][0]["off"], 1, 1, 1, defer_ok=False)      # the fold reads tmp right behind the call: its partials are summed by the call itself
                    d.dw = tmp.data_ptr()
                    Bp.keep.append(d)
                    Bp.add("rua_conv_wgrad", C.byref(d))
                    Bp.add("rua_stem_bwd_fold", tmp.data_ptr(), self.G(stem["segs"][0]["off"]), self.G(stem["bias"]), Cin, w0)
                else:
                    Bp.add("rua_stem_bwd", self.x_in.ptr, c1.grad.ptr, self.G(stem["segs"][0]["off"]), self.G(stem["bias"]),
                           c1.M, Cin, w0, self.dt)
            self.back_steps.append(stem_back)
        self.cur_tag, F.scope = None, None
        x = c1
        skips = []
        for i, (nf, dils) in enumerate(lv):
            if i > 0:
                x = self.tagged(f"down{i + 1}", self.down, x, nf)
            self.block_tag = f"enc{i + 1}"
            x = self.resblock(x, nf, dils)
            skips.append(x)
        x = self.tagged("psp_mid", self.psp if v2 else self.psp_v1, x, lv[-1][0])
        for i in range(len(lv) - 2, -1, -1):
            nf, dils = lv[i]
            x = self.tagged(f"up{i + 1}", self.up_combine if v2 else self.up_combine_v1, x, skips[i], nf)
            self.block_tag = f"dec{i + 1}"
            x = self.resblock(x, nf, dils)
        x_comb = self.tagged("combine_top", self.final_combine, x, c1, w0) if v2 else self.tagged("combine_top", self.relu_cat_conv_v1, x, 0, c1, w0, False)
        x_psp = self.tagged("psp_top", self.psp if v2 else self.psp_v1, x_comb, w0)
        Cc = cfg.num_classes
        if not cfg.multitasking:
            self.tagged("head_seg", self.head, x_psp, Cc, L.ACT_SOFTMAX, "seg")
        elif not self.dry and self.e.group_heads and self.e.dtype == "bf16":
            self.heads_grouped(x_psp, x_comb, w0, Cc)
        else:
            def seg_head():
                s = self.conv3x3_relu(x_psp, w0, "seg1")
                s = self.conv3x3_relu(s, w0, "seg2")
                self.head(s, Cc, L.ACT_SOFTMAX, "seg", "seg3")

            def dist_head():
                d = self.conv3x3_relu(x_comb, w0)
                d = self.conv3x3_relu(d, w0)
                self.head(d, Cc, L.ACT_SOFTMAX, "dist")
            self.tagged("head_seg", seg_head)
            self.tagged("head_bound", lambda: self.head(self.conv3x3_relu(x_psp, w0), Cc, L.ACT_SIGMOID, "bound"))
            self.tagged("head_dist", dist_head)
            self.tagged("head_color", self.head, x_comb, 3, L.ACT_SIGMOID, "color", "color")
        self.outputs = {h["name"]: h for h in self.heads}
        if self.e.loss is not None and not self.dry:
            self._record_losses()

    def _record_losses(self):
        """Loss / metric launches and (training) the whole backward plan, in reverse creation order."""
        # multitask: the heads' Tanimoto finalisations and their d(loss)/d(logits) as one launch each (rua_tanimoto_finalize_multi, rua_head_dz_multi)
        self.multi_head = len(self.heads) > 1 and self.e.multi_head
        self._tani_multi, self._dz_multi, self._dz_emitted = [], [], False
        for h in self.heads:
            self.cur_tag = self.loss_plan.scope = "loss_" + h["name"]
            self.head_loss(h)
        if self._tani_multi:
            self.loss_plan.scope = "loss_finalize"
            arr = (L.TaniHead * len(self._tani_multi))(*self._tani_multi)
            self.loss_plan.keep.append(arr)
            self.loss_plan.add("rua_tanimoto_finalize_multi", arr, len(self._tani_multi))
        self.cur_tag = self.loss_plan.scope = None
        seg = self.outputs["seg"]
        if not seg.get("metrics_done"):
            self.loss_plan.add("rua_seg_metrics", seg["p"].ptr, seg["y"].ptr, seg["x"].M, seg["C"], self.e.scalars_ptr + 8 * 8)
        if self.training:
            for step in reversed(self.back_steps):
                step()
            self.back_steps = []
            self.flush_wgrad(self.bwd)


# ---------------------------------------------------------------------------------------
class Engine:
    """Owns parameters, optimizer state and the recorded graphs; one instance per process (= per GPU)."""

    def __init__(self, cfg: ModelConfig, dtype: str = "bf16", device: Optional[torch.device] = None, seed: int = 0,
                 split_k: bool = True, _layout_only: bool = False):
        assert dtype in ("bf16", "f32")
        self.cfg = cfg
        self.dt = L.RUA_BF16 if dtype == "bf16" else L.RUA_F32
        self.dtype = dtype
        self.esize = 2 if dtype == "bf16" else 4
        if cfg.width % 32 != 0:
            raise ValueError(f"width={cfg.width}: the first-stage width must be a multiple of 32 (PSP branches are width/4 "
                             "channels and the MFMA epilogue stores 8-channel pieces)")
        self.dev = None
        self.fuse_bn = os.environ.get("RUA_FUSE_BN", "1") != "0"     # normalise-on-load ResBlocks where the library offers it
        self.fuse_head_loss = os.environ.get("RUA_FUSE_HEAD_LOSS", "1") != "0"   # Tanimoto moments / seg metrics in the heads' forward epilogue
        self.bn_out_stats = os.environ.get("RUA_BN_OUT_STATS", "1") != "0"       # statistics of a ReLU-less BatchNorm's output from its coefficients
        self.bn_dx_bias = os.environ.get("RUA_BN_DX_BIAS", "1") != "0"           # bias gradients of the stride-2 convs from the sums of the BatchNorm backward that writes their output's gradient
        self.pool_pyramid = os.environ.get("RUA_POOL_PYRAMID", "1") != "0"   # PSPPooling: the 2 / 4 / 8 poolings (and their adjoints) in single passes
        self.fold_bn = os.environ.get("RUA_FOLD_BN", "1") != "0"     # ... and their coefficient launches folded into the convs' prologues
        self.defer_reduce = os.environ.get("RUA_DEFER_REDUCE", "1") != "0"     # weight-gradient partials summed by batched launches
        # ... flushed whenever more than this many bytes of partials are pending: the K-slice slabs of levels 3 - 5 are then still in the 256 MB
        # Infinity Cache when the reduction reads them (one reduction at the end of the backward read all 0.6 GB from HBM)
        self.flush_bytes = int(float(os.environ.get("RUA_FLUSH_MB", "1e9")) * (1 << 20))
        self.cu_count = 256
        self.split_k = split_k       # False: bit-reproducible convolutions (no fp32-atomic K slices); parity tests on tiny
                                     # inputs use it because a BatchNorm over 2 samples amplifies atomic-order noise ~1e4x
        self.params = ParamStore()
        self.layers: List[dict] = []
        self.cursor = 0
        self.built = False
        self.loss: Optional[LossSpec] = None
        self.graphs: Dict[Tuple[int, bool], Graph] = {}
        self.class_w_ptr = None
        self.scalars_ptr = 0
        Graph(self, 1, True, dry=True)                      # enumerate layers / parameters (no device needed)
        self.built = True
        ps = self.params
        if _layout_only:
            return
        L.lib()                                             # raises if librua_hip.so is missing
        if not torch.cuda.is_available():
            raise L.RuaError("no HIP device visible: the ResUnet-a training path runs on MI355X only (there is no CPU fallback)")
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        cu = C.c_int32(0)
        if L.lib().raw("rua_device_info")(C.byref(cu), None, None, 0) == 0 and cu.value > 0:
            self.cu_count = cu.value
        z = lambda n, dt=torch.float32: torch.zeros(max(n, 16), dtype=dt, device=self.dev)
        self.P, self.G, self.M1, self.V1, self.S = z(ps.n), z(ps.n + ps.ns + 16), z(ps.n), z(ps.n), z(ps.ns)      # G: room behind the last gradient for the BatchNorm state's ride in the first all-reduce bucket (dist.py)
        tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.Wf, self.Wd = z(ps.n, tdt), z(ps.n, tdt)
        # first-writer overwrite of the gradient arena: in a whole step every weight gradient has ONE producer and the optimizer left the arena zero, so the kernels
        # that end in a read-modify-write of dW store instead (a device flag, include/rua_hip.h: forward_backward() called on its own still ACCUMULATES)
        self.wgrad_overwrite = os.environ.get("RUA_WGRAD_OVERWRITE", "1") != "0"
        self.ow_flag = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self._ow = 0
        self._g_pending = False                             # G holds gradients no optimizer step has consumed (forward_backward() on its own)
        self.opt_wcopy = dtype == "bf16" and os.environ.get("RUA_OPT_WCOPY", "1") != "0"   # the optimizer writes the forward-layout bf16 copy itself
        self.wf_fresh = False                               # the forward copy holds bf16(P): written by the optimizer (or a full rua_weight_prep) since P last changed
        items = np.zeros(0, dtype=[("src", "<i8"), ("dst", "<i8"), ("taps", "<i4"), ("cout", "<i4"), ("c", "<i4"), ("pad", "<i4")])
        rows, mx = [], 1
        for lay in self.layers:
            if "segs" in lay and lay["mfma"]:
                for sg in lay["segs"]:
                    rows.append((sg["off"], sg["dst"], lay["taps"], lay["cout"], sg["C"], 0))
                    mx = max(mx, lay["taps"] * lay["cout"] * sg["C"])
        items = np.array(rows, dtype=items.dtype)
        self.wprep_items = torch.from_numpy(np.frombuffer(items.tobytes(), dtype=np.uint8).copy()).to(self.dev)
        self.wprep_n, self.wprep_max = len(rows), mx
        bm = []                                              # block map of rua_weight_prep_dgrad: (item, first tile) per block
        for i, (_, _, taps, cout, ci, _) in enumerate(rows):
            for b in range(L.lib().raw("rua_wprep_blocks")(taps, cout, ci)):
                bm += [i, 8 * b]
        self.wprep_map = torch.tensor(bm if bm else [0, 0], dtype=torch.int32, device=self.dev)
        self.wprep_map_n = len(bm) // 2
        self.stats_arena = torch.zeros(1 << 22, dtype=torch.float64, device=self.dev)
        # one split-K slab workspace and one weight-gradient partial scratch per lane: branches run concurrently
        self.workspaces = [torch.zeros(8 << 20, dtype=torch.float32, device=self.dev) for _ in range(4)]
        self.scratches = [torch.zeros((32 if i == 0 else 16) << 20, dtype=torch.float32, device=self.dev) for i in range(4)]   # lane 0: room for the fp32 K-slice slabs of the deterministic weight gradients (128 MB)
        self.workspace, self.scratch = self.workspaces[0], self.scratches[0]
        self.side_streams = [torch.cuda.Stream(device=self.dev) for _ in range(3)]
        self.use_lanes = False      # measured: no gain (13.5 vs 14.0 ms/step), the step is bound by shared memory-side resources
        self.lr_dev = torch.zeros(16, dtype=torch.float32, device=self.dev)               # step-dependent optimizer scalars
        self.lr_state = torch.zeros(2, dtype=torch.float64, device=self.dev)              # [steps taken, base learning rate]
        self._t_dev, self._lr_base_dev = -1, None                                         # what lr_state holds (host shadow)
        self._dp_fence = torch.zeros(16, dtype=torch.float32, device=self.dev)            # see _graph_step_dp
        self.use_graph = os.environ.get("RUA_USE_GRAPH", "1") != "0"                      # 0 (experiments): the single-GPU step as eager launches too
        self.group_1x1 = os.environ.get("RUA_GROUP_1X1", "1") != "0"       # PSPPooling's branch convolutions and the per-source gradients of concatenating 1x1 convolutions as groups
        self.group_heads = os.environ.get("RUA_GROUP_HEADS", "1") != "0"   # the heads' 3x3 convolutions (and their gradients) grouped across the heads
        self.multi_head = os.environ.get("RUA_MULTI_HEAD", "1") != "0"     # the heads' loss finalisation / d(loss)/d(logits) as one launch each
        self.merge_wgrad = {int(v) for v in os.environ.get("RUA_MERGE_WGRAD", "1,32,64,128,256").split(",") if v}     # channel counts whose ResBlocks issue both weight-gradient groups as one (Graph.wgrad_now_or_later)
        self.group_wgrad_pw = os.environ.get("RUA_GROUP_WGRAD_PW", "1") != "0"   # the narrow 1x1 weight gradients of a composite as one grid (Graph.wgrad_pw_group)
        self.stem_mfma = os.environ.get("RUA_STEM_MFMA", "1") != "0"       # bf16: the stem's weight gradient through rua_stem_fwd_pack / rua_conv_wgrad / rua_stem_bwd_fold
        self.stem_stats = os.environ.get("RUA_STEM_STATS", "1") != "0"     # rua_stem_fwd_stats instead of a rua_col_stats pass over the stem's output
        self._captured: Dict[int, object] = {}
        self._captured_eval: Dict[int, object] = {}        # batch -> graph of the inference forward (False: capture failed)
        self._eval_seen = set()
        self._captured_dp: Dict[int, list] = {}
        # Data-parallel step: eager launches by default (2.4 ms of host time per 10.4 ms step; measured as fast as the
        # HIP-graph pieces and immune to the graph-launch / stream-event hazard described in _graph_step_dp); RUA_DP_GRAPH=1
        # selects the pieces.
        self.dp_graph = os.environ.get("RUA_DP_GRAPH", "0") == "1"
        self.dp_fence = os.environ.get("RUA_DP_FENCE", "0") == "1"      # experiments only (tools/dp_graph_check.py): an eager kernel behind every piece replay
        self.scalars_ptr = self.stats_arena.data_ptr()
        self.t = 0
        self.weights_dirty = True
        self.world = 1
        self.dist = None
        P, S = ps.init_host(seed)
        self.P.copy_(torch.from_numpy(P)); self.S.copy_(torch.from_numpy(S))

    @classmethod
    def param_layout(cls, cfg: ModelConfig, dtype: str = "f32") -> ParamStore:
        """Parameter table (names, shapes, flat offsets) without touching a device."""
        return cls(cfg, dtype, _layout_only=True).params

    # -- layer registry ------------------------------------------------------------------------------
    def layer(self, kind, *args):
        if not self.built:
            rec = self.params.conv(*args) if kind == "conv" else self.params.bn(*args)
            self.layers.append(rec)
            return rec
        rec = self.layers[self.cursor]
        self.cursor += 1
        return rec

    def graph(self, batch: int, training: bool) -> Graph:
        key = (batch, training)
        if key not in self.graphs:
            assert self.loss is not None, "compile() first"
            self.cursor = 0
            self.graphs[key] = Graph(self, batch, training)
            assert self.cursor == len(self.layers)
        return self.graphs[key]

    def compile(self, spec: LossSpec):
        if self.dev is not None and (self.graphs or self._captured or self._captured_eval or self._captured_dp):
            # a re-compile (Keras fine-tune pattern, or predict()'s auto-compile followed by compile): the captured HIP graphs
            # hold raw pointers into the old plan's buffers and bake in the old loss / optimizer - drop them with the plan
            torch.cuda.synchronize()
        self._captured, self._captured_eval, self._captured_dp = {}, {}, {}
        self._eval_seen = set()
        self._opt_graph = None
        if self.loss is not None and (self.loss.optimizer != spec.optimizer):
            self._t_dev, self._lr_base_dev = -1, None          # device-side step counter / base rate are pushed again
        self.loss = spec
        self.graphs = {}
        if spec.class_weights is not None:
            self.class_w = torch.tensor(list(spec.class_weights) + [0.0] * 8, dtype=torch.float32, device=self.dev)
            self.class_w_ptr = self.class_w.data_ptr()
        else:
            self.class_w_ptr = None

    def drop_plans(self):
        """Forget every recorded plan and captured graph (they are rebuilt on the next step): a launch heuristic that is read when a plan
        is RECORDED - the CU count behind the weight gradients' partial counts - has changed (rua_set_tuning(cu_reserve))."""
        if self.dev is not None and (self.graphs or self._captured or self._captured_eval or self._captured_dp):
            torch.cuda.synchronize()
        self._captured, self._captured_eval, self._captured_dp = {}, {}, {}
        self._eval_seen = set()
        self._opt_graph = None
        self.graphs = {}
        cu = C.c_int32(0)
        if self.dev is not None and L.lib().raw("rua_device_info")(C.byref(cu), None, None, 0) == 0 and cu.value > 0:
            self.cu_count = cu.value

    # -- weights ---------------------------------------------------------------------------------------
    def set_weights(self, keras_dict: Dict[str, np.ndarray]):
        P, S = self.params.from_keras(keras_dict)
        self.P.copy_(torch.from_numpy(P)); self.S.copy_(torch.from_numpy(S))
        self.params_changed()

    def params_changed(self):
        """P was written from outside the optimizer (set_weights, a checkpoint, a broadcast): both weight copies are rebuilt from it."""
        self.weights_dirty = True
        self.wf_fresh = False

    def get_weights(self) -> Dict[str, np.ndarray]:
        return self.params.to_keras(self.P.cpu().numpy(), self.S.cpu().numpy())

    def grads_keras(self) -> Dict[str, np.ndarray]:
        return self.params.to_keras(self.G.cpu().numpy())

    def count_params(self) -> int:
        return self.params.count()

    # -- execution ---------------------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _prep_weights(self, s):
        if self.weights_dirty:
            if self.opt_wcopy and self.wf_fresh:            # the optimizer left the forward copy: only the data-gradient layout is built, from it
                L.lib().call("rua_weight_prep_dgrad", self.Wf.data_ptr(), self.Wd.data_ptr(), self.wprep_items.data_ptr(), self.wprep_n, self.wprep_max,
                             self.wprep_map.data_ptr() if self.wprep_map_n else None, self.wprep_map_n, self.dt, C.c_void_p(s))
            else:
                L.lib().call("rua_weight_prep", self.P.data_ptr(), self.Wf.data_ptr(), self.Wd.data_ptr(), self.wprep_items.data_ptr(),
                             self.wprep_n, self.wprep_max, self.dt, C.c_void_p(s))
                self.wf_fresh = True
            self.weights_dirty = False

    def _ensure_forward_copy(self):
        """Before a captured step is replayed (its weight refresh is the data-gradient-only one): if P changed from outside, rebuild both copies now."""
        if self.opt_wcopy and not self.wf_fresh:
            self.weights_dirty = True
            self._prep_weights(self._stream())

    def _upload(self, g: Graph, x, y):
        def put(dst: torch.Tensor, src):
            if src is None:
                return
            if isinstance(src, np.ndarray):
                src = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32))
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"expected shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            # pageable sources may be temporaries: blocking copy.  Pinned tensors (loader.PrefetchLoader slots, which stay
            # alive and untouched until the step has returned its metrics) go up asynchronously on the compute stream.
            dst.copy_(src, non_blocking=bool(src.is_pinned()) if isinstance(src, torch.Tensor) else False)
        put(g.x_in.t, x)
        if y is not None:
            if isinstance(y, dict):
                for h in g.heads:
                    put(h["y"].t, y[h["name"]])
            else:
                put(g.heads[0]["y"].t, y)

    def _zero_arena(self, g: Graph, s):
        L.lib().call("rua_fill_zero", self.stats_arena.data_ptr(), g.stats_used * 8, C.c_void_p(s))

    def _results(self, g: Graph):
        """Metric list of the step in the reference's order.  Under data parallel the scalars are summed over the replicas
        first (one all-reduce of 16 doubles), so every rank returns the SAME numbers, aggregated the way
        MirroredStrategy reports them (train_ISPRS.py:347,432): losses = mean of the replicas' means, accuracy over all
        replicas' pixels, TP/FP/TN/FN summed.  Every rank must therefore fetch results in the same steps."""
        sc, w = self.stats_arena[:16], 1
        if self.dist is not None and self.world > 1:
            sc, w = self.dist.reduce_scalars(sc), self.world
        sc = sc.cpu().numpy()
        per = [float(sc[h["slot"]] * h["norm"] / w) for h in g.heads]
        total = sum(self.loss.weight[h["name"]] * v for h, v in zip(g.heads, per))
        M = g.outputs["seg"]["x"].M * w
        mets = [float(sc[8] / M), float(sc[9]), float(sc[10]), float(sc[11]), float(sc[12])]
        if self.cfg.multitasking:
            return [total] + per + mets
        return [total] + mets

    def _set_overwrite(self, v: int):
        if self._ow != v:
            self.ow_flag.fill_(v)
            self._ow = v

    def forward_backward(self, x=None, y=None, _whole_step: bool = False):
        """forward + losses + backward on the current stream; gradients are ADDED to self.G (several calls before one optimizer_step accumulate, e.g. the
        replicas of a data-parallel step played one after the other); _whole_step (train_step): the arena is zero and this is the step's only backward."""
        # first-writer overwrite needs a ZERO gradient arena: after a standalone forward_backward() (which accumulates) the arena holds
        # unapplied gradients, and the next whole step must accumulate on top of them too (mixing the two calls keeps its old meaning)
        self._set_overwrite(1 if (_whole_step and self.wgrad_overwrite and not self._g_pending) else 0)
        if not _whole_step:
            self._g_pending = True
        B = x.shape[0] if x is not None else self._last_B
        self._last_B = B
        g = self.graph(B, True)
        s = self._stream()
        self._upload(g, x, y)
        self._zero_arena(g, s)
        self._prep_weights(s)
        side = self.side_streams if self.use_lanes else None
        g.fwd.run(s, side=side)
        g.loss_plan.run(s)
        hooks = None
        if self.dist is not None:
            self.dist.start_state_reduce(self)
            self.dist.reducer.begin()
            if self.dist.overlap:
                hooks = self._bucket_hooks(g)
        g.bwd.run(s, hooks, side=side)
        return g

    def _bucket_marks(self, g: Graph) -> Dict[int, List[int]]:
        """{backward launch index -> gradient buckets that are complete after that launch}."""
        if getattr(g, "_marks", None) is None:
            last = [-1] * len(self.dist.buckets)
            for off, idx in g.grad_touch.items():
                b = self.dist.bucket_of(off)
                last[b] = max(last[b], idx)
            by_idx: Dict[int, List[int]] = {}
            for b, idx in enumerate(last):
                if idx >= 0:
                    by_idx.setdefault(g.bwd.safe_hook_index(idx), []).append(b)    # never while side lanes are running
            g._marks = by_idx
        return g._marks

    def _bucket_hooks(self, g: Graph):
        """{backward launch index -> fire the all-reduce of every gradient bucket that is complete after it}."""
        if getattr(g, "_hooks", None) is None:
            red = self.dist.reducer
            g._hooks = {idx: (lambda bs=bs: [red.ready(b) for b in bs]) for idx, bs in self._bucket_marks(g).items()}
        return g._hooks

    def _set_lr(self):
        """Advance the step counter.  The step-dependent learning rate is produced on the device by the optimizer launch
        (`rua_lr_step`), so a captured step replays with no host-side scalar update in between; the host only pushes the
        base rate / the counter when they were changed from outside (K.set_value(optimizer.lr), a loaded checkpoint)."""
        sp = self.loss
        if self._t_dev != self.t or self._lr_base_dev != sp.lr:
            self.lr_state.copy_(torch.tensor([float(self.t), float(sp.lr)], dtype=torch.float64))
            self._t_dev, self._lr_base_dev = self.t, sp.lr
        self.t += 1
        self._t_dev += 1                                    # the optimizer launch of this step advances the device counter

    def _launch_optimizer(self, grad_scale, s):
        sp = self.loss
        L.lib().call("rua_lr_step", self.lr_state.data_ptr(), self.lr_dev.data_ptr(), 1 if sp.optimizer == "adam" else 0,
                     float(sp.beta_1), float(sp.beta_2), C.c_void_p(s))
        wc = self.Wf.data_ptr() if self.opt_wcopy else None    # bf16 storage: the updated weights leave as the forward-layout copy in the same pass
        if sp.optimizer == "adam":
            L.lib().call("rua_adam_step_w", self.P.data_ptr(), self.G.data_ptr(), self.M1.data_ptr(), self.V1.data_ptr(), self.params.n,
                         0.0, self.lr_dev.data_ptr(), sp.beta_1, sp.beta_2, KERAS_EPS, grad_scale, 1, wc, C.c_void_p(s))
        else:
            L.lib().call("rua_sgd_step_w", self.P.data_ptr(), self.G.data_ptr(), self.M1.data_ptr(), self.params.n, 0.0,
                         self.lr_dev.data_ptr(), sp.momentum, grad_scale, 1, wc, C.c_void_p(s))

    def optimizer_step(self, grad_scale: float = 1.0):
        self._set_lr()
        self._launch_optimizer(grad_scale, self._stream())
        self.weights_dirty = True
        self._g_pending = False                             # the optimizer zeroed the arena as it consumed it

    def _graph_step(self, x, y):
        """Single-GPU fast path: the whole step (arena zeroing, weight refresh, forward, losses, backward, optimizer)
        captured once into a HIP graph and replayed; only the input upload and the lr scalar stay outside."""
        B = x.shape[0] if x is not None else self._last_B
        self._last_B = B
        g = self.graph(B, True)
        self._upload(g, x, y)
        self._set_lr()
        self._set_overwrite(1 if (self.wgrad_overwrite and not self._g_pending) else 0)
        self._g_pending = False                             # the step's optimizer launch zeroes the arena
        cap = self._captured.get(B)
        if cap is None:
            # warm-up run outside capture (sets kernel attributes, pays first-launch costs), then capture
            s = self._stream()
            side = self.side_streams if self.use_lanes else None
            self._zero_arena(g, s); self.weights_dirty = True; self._prep_weights(s)
            g.fwd.run(s, side=side); g.loss_plan.run(s); g.bwd.run(s, side=side)
            self._launch_optimizer(1.0, s)
            torch.cuda.synchronize()
            cap = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cap):
                s = self._stream()
                self._zero_arena(g, s); self.weights_dirty = True; self._prep_weights(s)
                g.fwd.run(s, side=side); g.loss_plan.run(s); g.bwd.run(s, side=side)
                self._launch_optimizer(1.0, s)
            self._captured[B] = cap
            self.weights_dirty = True
            # the warm-up consumed this step's update; the capture itself launched nothing
            return g
        self._ensure_forward_copy()
        cap.replay()
        self.weights_dirty = True
        return g

    def count_step_dispatches(self, batch: int) -> Optional[int]:
        """Kernel nodes of the whole training step (arena fill, weight refresh, forward, losses, backward, optimizer) captured
        into a throw-away HIP graph: the number of dispatches a step costs, counted rather than read off a profile.  The step must
        have run at least once (kernel attributes set); the capture launches nothing."""
        try:
            g = self.graph(batch, True)
            torch.cuda.synchronize()
            cap = torch.cuda.CUDAGraph(keep_graph=True)
            dirty, fresh = self.weights_dirty, self.wf_fresh
            with torch.cuda.graph(cap):
                s = self._stream()
                self._zero_arena(g, s); self.weights_dirty = True; self._prep_weights(s)
                g.fwd.run(s); g.loss_plan.run(s); g.bwd.run(s)
                self._launch_optimizer(1.0, s)
            self.weights_dirty, self.wf_fresh = dirty, fresh
            k = C.c_int32(0)
            L.lib().call("rua_graph_kernel_nodes", C.c_void_p(cap.raw_cuda_graph()), C.byref(k), None)
            del cap
            return int(k.value)
        except Exception:                                   # diagnostics only
            torch.cuda.synchronize()
            return None

    def _graph_step_dp(self, x, y):
        """Data-parallel fast path: the step is cut at the launches after which a gradient bucket is complete; every
        piece is its own HIP graph, the bucket all-reduces are issued eagerly between the replays on the reducer's side
        stream (RCCL stays outside the captures), so they overlap the next pieces exactly like in the eager path."""
        B = x.shape[0] if x is not None else self._last_B
        self._last_B = B
        g = self.graph(B, True)
        self._upload(g, x, y)
        red = self.dist.reducer
        pieces = self._captured_dp.get(B)
        if pieces is None:
            # one eager step (first-launch costs, kernel attributes, RCCL channel set-up), then capture the pieces
            self.forward_backward(None, None, _whole_step=True)
            self.dist.reduce_gradients(self)
            self.optimizer_step(1.0 / self.world)
            torch.cuda.synchronize()
            marks = self._bucket_marks(g) if self.dist.overlap else {}
            cuts = sorted(marks)
            nb = len(g.bwd.calls)
            pieces, first = [], 0
            try:
                for ci, idx in enumerate(cuts + [nb - 1]):
                    if ci == len(cuts) and first >= nb:
                        break
                    cap = torch.cuda.CUDAGraph()
                    # thread_local: the process group's watchdog thread may query events while we capture
                    with torch.cuda.graph(cap, capture_error_mode="thread_local"):
                        s = self._stream()
                        if first == 0:
                            self._zero_arena(g, s); self.weights_dirty = True; self._prep_weights(s)
                            g.fwd.run(s); g.loss_plan.run(s)
                        g.bwd.run(s, first=first, last=idx + 1)
                    pieces.append((cap, marks.get(idx, []) if ci < len(cuts) else []))
                    first = idx + 1
                opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(opt, capture_error_mode="thread_local"):
                    self._launch_optimizer(1.0 / self.world, self._stream())
                ok = 1
            except RuntimeError as exc:
                # Capture is an optimisation: the eager launch sequence issues the same kernels and the same collectives
                # (BN state first, then the buckets in the same order - see the replay below).  Say so loudly and carry on.
                import sys
                print(f"[resunet_a] HIP-graph capture of the data-parallel step failed ({exc}); running it eagerly", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                ok = 0
            # the ranks switch TOGETHER: the fallback is decided by a MIN over the ranks' capture results
            if self.dist.all_ranks_ok(ok) == 0:
                self.dp_graph = False
                self.weights_dirty = True
                return g
            self._captured_dp[B] = pieces
            self._opt_graph = opt
            self.weights_dirty = True
            return g
        self._set_lr()
        self._set_overwrite(1 if (self.wgrad_overwrite and not self._g_pending) else 0)
        self._g_pending = False
        self._ensure_forward_copy()
        red.begin()
        for pi, (cap, buckets) in enumerate(pieces):
            cap.replay()
            # (History, DESIGN.md section 6: with hipMemsetAsync captured as memset NODES - the statistics-arena fill at the head of the
            # first piece, the gradient fills of the stride-2 convolutions - these back-to-back piece replays corrupted the step on
            # some boxes of the pool, deterministically per process and NOT as an ordering problem: a device synchronise between the
            # replays did not help, an eager kernel launch between them did, and so does what is in place now - rua_fill_zero is an
            # ordinary kernel, no graph holds a memset node.  tools/dp_graph_check.py reproduces all of it; RUA_DP_FENCE=1 brings the
            # eager kernel back for experiments.)
            if self.dp_fence:
                self._dp_fence.zero_()
            elif os.environ.get("RUA_DP_SYNC") == "1":          # experiment (tools/dp_graph_check.py): a device synchronise instead
                torch.cuda.synchronize()
            if pi == 0:
                # same collective order as the eager path (forward_backward): BN moving statistics first - they are final
                # once the forward (inside the first piece) has run - then the gradient buckets
                self.dist.start_state_reduce(self)
            for b in buckets:
                red.ready(b)
        self.dist.reduce_gradients(self)
        self._opt_graph.replay()
        self.weights_dirty = True
        return g

    def train_step(self, x=None, y=None, fetch: bool = True):
        """One Keras train_on_batch (train_ISPRS.py:131,148): returns the metric list in the reference's order."""
        if self.use_graph and self.dist is None:
            g = self._graph_step(x, y)
            return self._results(g) if fetch else None
        if self.use_graph and self.dp_graph and not self.dist.host_staged and not self.use_lanes:
            g = self._graph_step_dp(x, y)
            return self._results(g) if fetch else None
        g = self.forward_backward(x, y, _whole_step=True)
        if self.dist is not None:
            self.dist.reduce_gradients(self)
        self.optimizer_step(1.0 / self.world)
        return self._results(g) if fetch else None

    def test_step(self, x, y):
        """Keras test_on_batch (train_ISPRS.py:167,186): BN uses moving statistics, nothing is updated."""
        g = self.graph(x.shape[0], False)
        s = self._stream()
        self._upload(g, x, y)
        self._zero_arena(g, s)
        self._prep_weights(s)
        g.fwd.run(s)
        g.loss_plan.run(s)
        return self._results(g)

    def predict(self, x):
        """Inference forward (moving BN statistics).  The forward launch list of a batch size is captured into a HIP graph
        on its second use (the reference's evaluation calls predict(batch_size=1) per patch, test_ISPRS.py:28: ~250
        launches of host time per patch otherwise)."""
        B = x.shape[0]
        g = self.graph(B, False)
        s = self._stream()
        self._upload(g, x, None)
        self._prep_weights(s)
        cap = self._captured_eval.get(B) if self.use_graph else None
        if cap is None or self.use_lanes:
            g.fwd.run(s)
            if self.use_graph and not self.use_lanes and B in self._eval_seen:
                try:
                    torch.cuda.synchronize()
                    cap = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(cap):
                        g.fwd.run(self._stream())
                    self._captured_eval[B] = cap           # capture launched nothing: this call's results are the eager run's
                except RuntimeError as exc:
                    import sys
                    print(f"[resunet_a] HIP-graph capture of the inference forward failed ({exc}); staying eager", file=sys.stderr, flush=True)
                    torch.cuda.synchronize()
                    self._captured_eval[B] = False
            self._eval_seen.add(B)
        elif cap is False:
            g.fwd.run(s)
        else:
            cap.replay()
        outs = {h["name"]: h["p"].t.cpu().numpy() for h in g.heads}
        return outs if self.cfg.multitasking else outs["seg"]

    def logits(self, training: bool, batch: int) -> Dict[str, np.ndarray]:
        g = self.graph(batch, training)
        return {h["name"]: h["z"].t.cpu().numpy() for h in g.heads}
