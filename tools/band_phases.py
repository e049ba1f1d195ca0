"""Wall-clock stamps (100 MHz) of three blocks of conv_band32 (debug build -DRUA_BAND_TS, RUA_LIB_PATH=<that build>): entry, tables / BatchNorm fold,
ring filled, end of each (branch, kernel row) phase, ring drained, epilogue stores issued, stores acknowledged.  8 x 256 x 256 x 32, four branches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BB_ONLY", "sum")
os.environ.setdefault("BB_REPS", "5")
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402
import bench_conv_band  # noqa: E402

bench_conv_band.main()
torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
ts = (C.c_ulonglong * 128)()
raw.rua_band_debug_ts(ts)
names = ["entry", "tables + fold", "ring filled, row 0 normalised"] + [f"phase {p} (branch {p // 3}, kernel row {p % 3})" for p in range(12)] + ["-", "ring drained", "epilogue stores issued", "stores acknowledged"]
t0 = min(ts[b * 32] for b in range(3))
for b, blk in enumerate(("block 0", "block 100", "last block")):
    v = list(ts[b * 32:b * 32 + 19])
    if v[18] == 0:
        v[18] = v[17]
    print(f"{blk}: entered {(v[0] - t0) * 10} ns after the first of the three; total {(v[18] - v[0]) * 10} ns")
    for i in range(1, 19):
        if i == 15:
            continue
        prev = v[i - 1] if i != 16 else v[14]
        print(f"   {names[i]:44s} +{(v[i] - prev) * 10:6d} ns")

if ts[96]:
    print("one stage of block 100 (phase 4, row 3), shader clocks after the barrier.  first half (wave 0): row DMAs issued | MFMAs issued | fragment reads issued | "
          "row s + 3 landed | in-place reads + arithmetic + writes issued | LDS drained | next barrier passed;  second half (wave NW/2): row DMAs issued | row s + 3 landed | "
          "in-place pass issued | MFMAs issued | fragment reads issued | LDS drained | next barrier passed")
    for h in range(2):
        v = list(ts[96 + 8 * h:96 + 8 * h + 8])
        print("   half", h, [int(x - v[0]) for x in v[1:]])
