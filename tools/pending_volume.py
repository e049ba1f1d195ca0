import os, sys, collections
sys.path.insert(0, os.getcwd())
import torch
from resunet_a_mltsk_keras_amd import engine as E
orig = E.Graph.flush_wgrad
tot = collections.Counter(); cnt = collections.Counter()
def fl(self, plan):
    for r, off in self.pending:
        if r.kind == 1: b = self.e.cu_count * 9 * 32 * r.CC * 4
        elif r.kind == 3: b = r.parts * r.n * 8
        else: b = r.parts * r.n * 4
        key = (r.kind, r.parts, r.n, r.CC)
        tot[key] += b; cnt[key] += 1
    return orig(self, plan)
E.Graph.flush_wgrad = fl
import bench
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--blocks", "1"]
try:
    bench.main()
except SystemExit:
    pass
s = 0
for k, b in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(k, cnt[k], f"{b/1e6:.1f} MB"); s += b
print("total", s / 1e6, "MB")
