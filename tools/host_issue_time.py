"""Host time the EAGER step takes to issue (the path every rank of a data-parallel job runs): one rank, one-rank RCCL group, cfg3 bf16 bs 8.
Prints per step: host time from the first launch call to the return of the last one (no synchronisation inside), and the GPU time of the same steps
(device synchronise at both ends of the block).  If the first is well below the second the GPU is never waiting for the host, whatever issues the launches."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch.distributed as dist
    args = argparse.Namespace(force_dp=True, bucket_mb=25.0, no_overlap=False)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29578", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    eng, x, y, B, depth = bench.build_engine("cfg3", "bf16", 0, 0, 1, args)
    eng.train_step(x, y, fetch=False)
    for _ in range(5):
        eng.train_step(None, None, fetch=False)
    torch.cuda.synchronize()
    n = 30
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        eng.train_step(None, None, fetch=False)
        host.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    g = eng.graph(B, True)
    nl = len(g.fwd.calls) + len(g.loss_plan.calls) + len(g.bwd.calls)
    host.sort()
    print(f"{nl} plan launches per step; host issue time per step: median {host[n // 2] * 1e3:.3f} ms, min {host[0] * 1e3:.3f}, max {host[-1] * 1e3:.3f}")
    print(f"{n} steps: issue loop {1e3 * (t1 - t0) / n:.3f} ms/step, with the final synchronise {1e3 * (t2 - t0) / n:.3f} ms/step "
          f"(the loop itself blocks once the launch queue is full, so the first figure is an upper bound of the host's own time)")
    # the host's own time: the same step with the GPU idle at every launch (synchronise before each step, time only the issue)
    own = []
    for _ in range(10):
        torch.cuda.synchronize()
        a = time.perf_counter()
        eng.train_step(None, None, fetch=False)
        own.append(time.perf_counter() - a)
    own.sort()
    print(f"host issue time of one step into an empty queue: median {own[5] * 1e3:.3f} ms ({own[5] * 1e6 / nl:.2f} us per launch)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
