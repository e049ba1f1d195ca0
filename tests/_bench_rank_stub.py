"""Stand-in for a bench.py rank (tests/test_bench_launch.py): started by torch.distributed.run through bench.py's
self-launch branch; rank 0 prints one JSON line with what it was given, every rank exits with STUB_RC."""
import json
import os
import sys

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if rank == 0:
    print("noise that is not the result line", flush=True)
    print(json.dumps({"stub": True, "world": world, "argv": sys.argv[1:], "master": os.environ.get("MASTER_ADDR")}), flush=True)
sys.exit(int(os.environ.get("STUB_RC", "0")))
