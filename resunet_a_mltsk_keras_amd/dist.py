"""Data-parallel training over RCCL/xGMI: one process per GPU, patches sharded by sample.

Semantics follow tf.distribute.MirroredStrategy as the reference uses it (train_ISPRS.py:347,432):
every replica runs the step on its local shard with LOCAL BatchNorm batch statistics and LOCAL
Tanimoto class volumes, gradients are summed and divided by the number of replicas, BN moving
statistics are mean-aggregated.  The only data-path collective is the gradient all-reduce: the flat
fp32 gradient buffer is cut into contiguous buckets; a bucket's asynchronous all-reduce is issued (it runs
on the process group's own stream) as soon as the backward plan has issued the last launch that writes
into it, so the all-reduce of the deep (parameter-heavy) stages overlaps the backward of the shallow
(FLOP-heavy) ones.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def make_buckets(entries: List[Tuple[int, int]], total: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """Contiguous [begin, end) slices of the flat buffer, walking parameters from the END of the buffer
    (the order backward produces them), each at least bucket_elems long except the last two (the front of the buffer)."""
    bounds = sorted(set([0, total] + [o for o, _ in entries]))
    buckets, end = [], total
    for b in reversed(bounds[:-1]):
        if end - b >= bucket_elems or b == 0:
            buckets.append((b, end))
            end = b
    # The front bucket is complete only when backward ends, so its all-reduce is the one nothing can hide: keep it small
    # (the first parameters of the buffer = the top encoder levels, ~1/16 of a bucket) and give the rest its own bucket,
    # which completes a couple of levels earlier.
    tail = max(bucket_elems // 16, 1)
    b0, e0 = buckets[-1]
    if e0 - b0 > 2 * tail:
        cut = next((b for b in bounds if b >= tail and b < e0), None)
        if cut is not None and cut > b0:
            buckets[-1:] = [(cut, e0), (b0, cut)]
    return buckets


class GradReducer:
    """Bucketed all-reduce (sum) of a flat tensor.  Device-agnostic so that the bucketing / readiness logic is
    testable with gloo on CPU; on GPUs the collectives run on `side` and are fenced with events."""

    def __init__(self, flat: torch.Tensor, buckets: List[Tuple[int, int]], group=None, use_side_stream: bool = True,
                 host_staged: bool = False):
        self.flat, self.buckets, self.group = flat, buckets, group
        self.cuda = flat.is_cuda
        self.host_staged = host_staged           # gloo backend with device tensors: stage through pinned host memory
        self.side = torch.cuda.Stream(device=flat.device) if (self.cuda and use_side_stream and not host_staged) else None
        self.works: List = []
        self.fired = [False] * len(buckets)
        self.measure = False                     # bench.py: bracket the wait for the collectives with events on the launch stream
        self._exposed: List[Tuple[object, object]] = []
        self._holes: List[Tuple[object, object]] = []    # measure: event pairs around every collective's ISSUE on the launch stream
        self.n_coll = 0                          # collectives issued since begin()
        self.n_coll_last = 0                     # ... of the last finished step

    def exposed_ms(self) -> Optional[float]:
        """Mean time per step the LAUNCH stream stood waiting for gradient all-reduces (what the backward did not hide), from the
        event pairs recorded while `measure` was set; None if nothing was measured.  Call after a device synchronise."""
        if not self._exposed:
            return None
        ms = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return float(sum(ms) / len(ms))

    def hole_us(self) -> Optional[float]:
        """Mean time per step between the two launch-stream events that bracket the ISSUE of each gradient all-reduce, summed over the step's
        collectives (what a collective costs the launch stream even when its transfer is hidden: DESIGN section 6); None if nothing was measured."""
        if not self._holes or not self._steps_measured:
            return None
        us = sum(a.elapsed_time(b) for a, b in self._holes) * 1e3 / self._steps_measured
        self._holes, self._steps_measured = [], 0
        return float(us)

    _steps_measured = 0

    def begin(self):
        self.works = []
        self.fired = [False] * len(self.buckets)
        self.n_coll = 0

    def ready(self, i: int):
        """Bucket i is final on the current stream: launch its all-reduce (async)."""
        if self.fired[i]:
            return
        self.fired[i] = True
        a, b = self.buckets[i]
        view = self.flat[a:b]
        if self.host_staged:
            h = view.cpu()
            dist.all_reduce(h, group=self.group)
            view.copy_(h)
            return
        self.n_coll += 1
        pair = None
        if self.measure and self.cuda:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record(torch.cuda.current_stream(self.flat.device))
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                self.works.append(dist.all_reduce(view, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(view, group=self.group, async_op=True))
        if pair is not None:
            pair[1].record(torch.cuda.current_stream(self.flat.device))
            self._holes.append(pair)

    def finish(self):
        """Every bucket reduced and visible to the current stream."""
        for i in range(len(self.buckets)):
            self.ready(i)
        ev = None
        if self.measure and self.cuda and not self.host_staged:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(self.flat.device))
        self.n_coll_last = self.n_coll
        if self.measure:
            self._steps_measured += 1
        for w in self.works:
            w.wait()
        if self.side is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.side)
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(self.flat.device))
            self._exposed.append(ev)
        self.works = []


class DataParallel:
    """Attaches to an Engine: broadcasts rank-0 weights, reduces gradients during backward, averages BN state."""

    def __init__(self, eng, bucket_mb: float = 50.0, group=None, overlap: bool = True):      # 50 MB: four collectives a step at cfg3 (51 / 73 / 36 / 3 MB); 25 MB: six - each costs ~25 us of launch-stream time
        assert dist.is_initialized()
        self.eng, self.group = eng, group
        self.world = dist.get_world_size(group)
        self.overlap = overlap
        eng.world = self.world
        eng.dist = self
        backend = dist.get_backend(group)
        self.host_staged = backend == "gloo" and eng.P.is_cuda
        entries = [(e["off"], e["size"]) for e in eng.params.entries]
        self.buckets = make_buckets(entries, eng.params.n, int(bucket_mb * (1 << 20) / 4))
        # No side stream of our own: an async all-reduce already runs on the process group's internal stream (which waits for
        # the launch stream at the call and is joined again by work.wait()), so issuing it from the launch stream at the
        # bucket's hook overlaps it with the rest of the backward.  A second stream of ours in between cost 0.3 ms per step on
        # one GPU (10.42 vs 10.14 ms: event record + wait + stream switch per bucket) for nothing.
        side = os.environ.get("RUA_DP_SIDE_STREAM", "0") == "1"
        # The BatchNorm moving statistics (~50 k floats, mean-aggregated like MirroredStrategy's mirrored variables) ride in the FIRST gradient bucket:
        # the gradient arena has room for a copy of them behind the last parameter, the bucket that ends there (the deepest layers: complete first) is
        # extended over it - one collective less per step (each costs the launch stream ~20 - 26 us even when its transfer is hidden).
        n, ns = eng.params.n, eng.params.ns
        self.fold_state = os.environ.get("RUA_DP_FOLD_STATE", "1") != "0" and eng.G.numel() >= n + ns and ns > 0
        if self.fold_state:
            self.buckets[0] = (self.buckets[0][0], n + ns)
        self.reducer = GradReducer(eng.G[:n + ns] if self.fold_state else eng.G[:n], self.buckets, group, use_side_stream=overlap and side, host_staged=self.host_staged)
        self._bcast(eng.P); self._bcast(eng.S)
        eng.params_changed()
        # Room for RCCL's kernels: several launchers size their grid at exactly one block per CU; with a bucket's all-reduce in flight
        # its workgroups hold some CUs and the last blocks of such a grid would wait for a second round.  Measured COST on one GPU
        # (nothing to make room for there): 8 CUs +0.3 %, 16 +1.2 %, 32 +2.0 % of the step; the benefit needs N > 1 to show - so the
        # default is 0 (the configuration every GPU test runs) until an N > 1 box has measured it; RUA_DP_CU_RESERVE opts in.
        # The CU count is read at plan time (partial counts recorded in the deferred reductions) AND at launch time: a change drops
        # every recorded plan and captured graph of the engine, so no plan outlives the count it was recorded with.
        self.cu_reserve = int(os.environ.get("RUA_DP_CU_RESERVE", "0")) if (self.world > 1 and eng.P.is_cuda) else 0
        if eng.P.is_cuda:
            from . import _lib as L
            if L.lib().get_tuning("cu_reserve") != self.cu_reserve:
                L.lib().set_tuning(cu_reserve=self.cu_reserve)
                eng.drop_plans()

    def _bcast(self, t):
        if self.host_staged:
            h = t.cpu(); dist.broadcast(h, 0, group=self.group); t.copy_(h)
        else:
            dist.broadcast(t, 0, group=self.group)

    def broadcast_optimizer_state(self):
        """Rank 0's optimizer moments and step count become everyone's (a checkpoint restored on every rank, or only on
        rank 0: train_ISPRS.py:474-480 resume)."""
        eng = self.eng
        self._bcast(eng.M1); self._bcast(eng.V1)
        t = torch.tensor([float(eng.t)], dtype=torch.float64)
        if self.host_staged or not eng.P.is_cuda:
            dist.broadcast(t, 0, group=self.group)
        else:
            t = t.to(eng.P.device); dist.broadcast(t, 0, group=self.group)
        eng.t = int(t.item())
        eng._t_dev = -1                                       # the device-side counter is pushed again at the next step

    def reduce_scalars(self, sc: torch.Tensor) -> torch.Tensor:
        """Sum over the replicas of the step's loss / metric scalars ("all-reduce of ~10 floats"); returns a
        new tensor, the arena itself stays local."""
        if self.host_staged:
            h = sc.cpu()
            dist.all_reduce(h, group=self.group)
            return h
        t = sc.clone()
        dist.all_reduce(t, group=self.group)
        return t

    def all_ranks_ok(self, ok: int) -> int:
        """MIN over the ranks of a 0/1 flag (collective: every rank calls it at the same point)."""
        t = torch.tensor([int(ok)], dtype=torch.int32)
        if not self.host_staged and self.eng.P.is_cuda:
            t = t.to(self.eng.P.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def bucket_of(self, off: int) -> int:
        for i, (a, b) in enumerate(self.buckets):
            if a <= off < b:
                return i
        raise ValueError(off)

    def start_state_reduce(self, eng):
        """BN moving statistics are final once the forward has been issued: their (small) all-reduce starts here and hides
        behind the backward instead of standing between the last gradient bucket and the optimizer."""
        self._s_work = None
        if self.fold_state:                                   # a copy behind the last gradient: reduced with the first bucket
            n, ns = eng.params.n, eng.params.ns
            eng.G[n:n + ns].copy_(eng.S[:ns])
        elif self.overlap and not self.host_staged:
            self._s_work = dist.all_reduce(eng.S, group=self.group, async_op=True)

    def reduce_gradients(self, eng):
        """Called after the backward plan (buckets already fired by the plan's markers when overlap is on)."""
        self.reducer.finish()
        S = eng.S
        work = getattr(self, "_s_work", None)
        self._s_work = None
        self.collectives_last = self.reducer.n_coll_last + (0 if self.fold_state else 1)
        if self.fold_state:
            n, ns = eng.params.n, eng.params.ns
            S[:ns].copy_(eng.G[n:n + ns])
            S[:ns].div_(self.world)
        elif self.host_staged:
            h = S.cpu(); dist.all_reduce(h, group=self.group); S.copy_(h / self.world)
        elif work is not None:
            work.wait()
            S.div_(self.world)
        else:
            dist.all_reduce(S, group=self.group)
            S.div_(self.world)


def comm_ranks(group=None, device=None) -> int:
    """How many ranks really take part in the group's collectives - counted BY a collective (a sum of ones), not read from the
    environment: the first line a multi-GPU bench record should be able to show."""
    t = torch.ones(1, dtype=torch.float32, device=device if device is not None else "cpu")
    dist.all_reduce(t, group=group)
    return int(round(float(t.item())))


def gather_floats(v: float, group=None, device=None) -> List[float]:
    """v of every rank, in rank order (all_gather)."""
    world = dist.get_world_size(group)
    t = torch.tensor([float(v)], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [float(o.item()) for o in out]


def dp_report(dp: "DataParallel", ms_per_step: float, device=None) -> dict:
    """The data-parallel block of bench.py's JSON line (collective: every rank calls it): ranks counted by the communicator,
    bucket sizes, the exposed all-reduce time, the ranks' own step times."""
    per_rank = gather_floats(ms_per_step, dp.group, device)
    exposed = dp.reducer.exposed_ms()
    hole = dp.reducer.hole_us() if hasattr(dp.reducer, "hole_us") else None
    exp_all = gather_floats(-1.0 if exposed is None else exposed, dp.group, device)
    return {
        "rccl_ranks": comm_ranks(dp.group, device),
        "backend": dist.get_backend(dp.group),
        "bucket_mb": [round((b - a) * 4 / 2**20, 2) for a, b in dp.buckets],
        "allreduce_exposed_ms": None if exposed is None else round(max(exp_all), 3),
        "allreduce_exposed_ms_per_rank": None if exposed is None else [round(v, 3) for v in exp_all],
        "ms_per_step_min": round(min(per_rank), 3), "ms_per_step_max": round(max(per_rank), 3),
        "ms_per_step_per_rank": [round(v, 3) for v in per_rank],
        "cu_reserve": dp.cu_reserve, "overlap": bool(dp.overlap),
        # collectives the step issues (gradient buckets [+ the BatchNorm state when it does not ride in the first bucket]; the 16 result scalars only when
        # results are fetched) and what their ISSUE costs the launch stream (event pairs around each all-reduce call), per step
        "collectives_per_step": getattr(dp, "collectives_last", None),
        "state_in_first_bucket": bool(getattr(dp, "fold_state", False)),
        "launch_stream_hole_us": None if hole is None else round(hole, 1),
    }
