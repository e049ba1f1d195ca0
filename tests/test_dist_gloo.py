"""world_size-2 CPU (gloo) tests of the data-parallel pieces: bucket construction, the bucketed gradient
reducer fired out of order, and MirroredStrategy's loss/gradient scaling convention."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from resunet_a_mltsk_keras_amd.dist import GradReducer, make_buckets
from resunet_a_mltsk_keras_amd.engine import Engine, ModelConfig


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_buckets_cover_flat_buffer_back_to_front():
    ps = Engine.param_layout(ModelConfig(input_shape=(256, 256, 6), num_classes=6, multitasking=True))
    entries = [(e["off"], e["size"]) for e in ps.entries]
    b = make_buckets(entries, ps.n, int(25 * (1 << 20) / 4))
    assert b[0][1] == ps.n and b[-1][0] == 0
    assert all(b[i][0] == b[i + 1][1] for i in range(len(b) - 1))           # contiguous, descending
    assert all(e - s >= 25 * (1 << 20) // 4 for s, e in b[:-2])              # all but the split front of the buffer
    assert b[-1][1] - b[-1][0] <= 2 * (25 * (1 << 20) // 4) // 16 + max(sz for _, sz in entries if _ < b[-1][1])   # small last-completing bucket
    starts = {o for o, _ in entries}
    assert all(s in starts or s == 0 for s, _ in b)                         # cut only at parameter boundaries
    assert 5 <= len(b) <= 8                                                 # 171 MB of fp32 gradients / 25 MB


def _worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    flat = torch.randn(n)
    mine = flat.clone()
    buckets = make_buckets([(o, 0) for o in range(0, n, 97)], n, 1000)
    red = GradReducer(flat, buckets, use_side_stream=False)
    red.begin()
    for i in [2, 0, 2, len(buckets) - 1]:                                    # out of order, repeated: fires once each
        red.ready(i)
    red.finish()                                                              # fires the rest, waits for all
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    exp = sum(gathered)
    ok = torch.allclose(flat, exp, atol=1e-6)
    # MirroredStrategy convention: per-replica mean loss, gradients summed then divided by the replica count
    g_local = torch.full((4,), float(rank + 1))
    dist.all_reduce(g_local)
    ok = ok and torch.allclose(g_local / world, torch.full((4,), (1 + 2) / 2.0))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_bucketed_reducer_two_ranks_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, 5003, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
