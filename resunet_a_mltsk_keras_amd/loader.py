"""Asynchronous patch loader for the reference's on-disk format.

The reference reads 5*B `.npy` files serially inside the training loop before every step
(train_ISPRS.py:115-141); at a GPU step of ~11 ms that loop, not the network, sets the pace.  This loader
keeps the same semantics - the caller's list of paths, paired BY NAME upstream, `n // batch_size` batches per
pass, the last partial batch dropped (train_ISPRS.py:102,154) - and moves the file reads to worker threads
that fill a small ring of preallocated (pinned, when a GPU is present) host buffers ahead of the consumer.
Buffers are float32 NHWC like the files, so the upload into the engine is a plain async copy.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


class PrefetchLoader:
    """Iterates `len(x_paths) // batch_size` batches `(x, y)`; x: float32 tensor [B,H,W,C], y: {head: tensor}.

    order:   sample indices of this pass (default 0..n-1); a new pass with another order: `loader.set_order(...)`.
    depth:   batches read ahead (ring slots = depth + 1: the slot handed to the caller is reused only after the
             caller has asked for the NEXT batch, i.e. after its training step has returned).
    workers: threads reading files (np.load releases the GIL while it reads).
    rank, world: data-parallel shard.  `batch_size` stays the GLOBAL batch (train_ISPRS.py:314 under MirroredStrategy);
             rank r reads and yields only samples [r*B/world, (r+1)*B/world) of every global batch - the contiguous
             split Keras makes - so N ranks read each file once between them instead of N times.
    """

    def __init__(self, x_paths: Sequence[str], y_paths: Dict[str, Sequence[str]], batch_size: int,
                 order: Optional[Sequence[int]] = None, depth: int = 2, workers: int = 4, pin: Optional[bool] = None,
                 rank: int = 0, world: int = 1):
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        if batch_size % world:
            raise ValueError(f"global batch {batch_size} not divisible by {world} replicas")
        self.rank, self.world, self.local_B = int(rank), int(world), int(batch_size) // int(world)
        for h, lst in y_paths.items():
            if len(lst) != len(x_paths):
                raise ValueError(f"label list '{h}' has {len(lst)} entries for {len(x_paths)} patches")
        self.x_paths, self.y_paths, self.B = list(x_paths), {h: list(v) for h, v in y_paths.items()}, int(batch_size)
        self.depth, self.workers = max(1, int(depth)), max(1, int(workers))
        self.pin = torch.cuda.is_available() if pin is None else bool(pin)
        self.order = list(range(len(self.x_paths))) if order is None else [int(i) for i in order]
        self._slots: List[Tuple[torch.Tensor, Dict[str, torch.Tensor]]] = []
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()
        if self.x_paths:
            x0 = np.load(self.x_paths[0])
            shapes = {h: np.load(v[0]).shape for h, v in self.y_paths.items()}
            mk = lambda shp: torch.empty((self.local_B,) + tuple(shp), dtype=torch.float32, pin_memory=self.pin)
            self._slots = [(mk(x0.shape), {h: mk(s) for h, s in shapes.items()}) for _ in range(self.depth + 1)]

    def __len__(self) -> int:
        return len(self.order) // self.B

    def set_order(self, order: Sequence[int]) -> None:
        self.order = [int(i) for i in order]

    # -- producer -------------------------------------------------------------------------------------------
    def _fill(self, slot: int, idx: Sequence[int], pool: ThreadPoolExecutor) -> None:
        xb, yb = self._slots[slot]
        xn, yn = xb.numpy(), {h: t.numpy() for h, t in yb.items()}

        def one(b: int, i: int) -> None:
            xn[b] = np.load(self.x_paths[i])
            for h, arr in yn.items():
                arr[b] = np.load(self.y_paths[h][i])          # assignment casts to float32 like .astype(np.float32)

        for f in [pool.submit(one, b, i) for b, i in enumerate(idx)]:
            f.result()                                        # re-raises a worker's exception here

    def _produce(self, ready: "queue.Queue", free: "queue.Queue") -> None:
        try:
            with ThreadPoolExecutor(self.workers) as pool:
                for k in range(len(self)):
                    slot = free.get()
                    if slot is None or self._stop.is_set():
                        return
                    first = k * self.B + self.rank * self.local_B
                    self._fill(slot, self.order[first:first + self.local_B], pool)
                    ready.put(slot)
            ready.put(None)
        except BaseException as exc:                          # hand the error to the consumer instead of dying silently
            ready.put(exc)

    # -- consumer -------------------------------------------------------------------------------------------
    def __iter__(self) -> Iterator[Tuple[torch.Tensor, Dict[str, torch.Tensor]]]:
        if self._thread is not None:
            raise RuntimeError("PrefetchLoader: one pass at a time")
        ready: "queue.Queue" = queue.Queue()
        free: "queue.Queue" = queue.Queue()
        for s in range(self.depth):                           # the last slot enters circulation when the first is handed out
            free.put(s)
        spare = self.depth
        self._stop.clear()
        self._thread = threading.Thread(target=self._produce, args=(ready, free), daemon=True)
        self._thread.start()
        held = None
        try:
            while True:
                item = ready.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                if held is not None:
                    free.put(held)                            # the caller is done with the previous batch
                elif spare is not None:
                    free.put(spare); spare = None
                held = item
                yield self._slots[item]
        finally:
            self._stop.set()
            free.put(None)
            self._thread.join()
            self._thread = None
