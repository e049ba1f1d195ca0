"""Does an async all-reduce on a one-rank RCCL group block the HOST until the launch stream has drained?  (tools/ab_dp.sh found ~26 us of launch-stream idle time per
collective in the data-parallel step with nothing running on any other queue.)  Queues ~50 ms of GPU work, then times the host side of dist.all_reduce(async_op=True)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
g = torch.zeros(8 << 20, device=dev); a = torch.randn(4096, 4096, device=dev)
dist.all_reduce(g); torch.cuda.synchronize()
for label, n in (("idle GPU", 0), ("~50 ms queued", 40), ("~50 ms queued (again)", 40)):
    for _ in range(n):
        b = a @ a
    t0 = time.perf_counter(); w = dist.all_reduce(g, async_op=True); t1 = time.perf_counter()
    e = torch.cuda.Event(); e.record(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{label:24s}: all_reduce(async) host call {1e6*(t1-t0):9.1f} us, event record {1e6*(t2-t1):7.1f} us, drain {1e3*(t3-t2):7.2f} ms", flush=True)
    w.wait()
dist.destroy_process_group()
