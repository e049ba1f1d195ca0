"""Phase timestamps of ONE block of conv_dmap (debug build -DRUA_DMAP_DBG_TS, RUA_LIB_PATH=<that build>): kernel entry, K loop entry / exit, tile
in LDS, epilogue loads issued, arithmetic + stores done, end.  8 x 64 x 64 x 128, 3x3."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda", 0)
N, HW, Cc = 8, 64, 128
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16)
aux = torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn((9, Cc, Cc), generator=g) / 34).to(dev).to(torch.bfloat16)
y = torch.zeros((N, HW, HW, Cc), device=dev, dtype=torch.bfloat16)
bias = torch.randn(Cc, device=dev)
sc, sh = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.3
stats = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
raw = C.CDLL(L.LIB_PATH)
for kind in ("bias + statistics", "mask + statistics 2"):
    d = L.ConvDesc()
    d.nseg = 1
    sg = d.seg[0]
    sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = x.data_ptr(), w.data_ptr(), Cc, HW, HW, 0, 3, 9
    d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, HW, HW, Cc, 1, L.RUA_BF16
    d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, HW, HW
    d.stats, d.stats_mode, d.stats_replicas = stats.data_ptr(), 1, 32
    if kind.startswith("bias"):
        d.bias = bias.data_ptr()
    else:
        d.aux, d.aux_mode, d.mscale, d.mshift, d.stats_mode = aux.data_ptr(), 2, sc.data_ptr(), sh.data_ptr(), 2
    for _ in range(5):
        lib.call("rua_conv_fwd", C.byref(d), s)
    torch.cuda.synchronize()
    ts = (C.c_ulonglong * 32)()
    raw.rua_debug_ts(ts)
    c, wc = list(ts[:7]), list(ts[16:23])
    names = ["entry", "K loop entry", "K loop exit", "tile in LDS", "epilogue loads issued", "arithmetic, stores issued", "end (statistics)"]
    print(kind, f"(shader clock / 100 MHz wall clock over the kernel: {(c[6] - c[0]) / max(1, wc[6] - wc[0]) * 100:.0f} MHz)")
    for i in range(1, 7):
        print(f"  {names[i]:28s} +{(wc[i] - wc[i - 1]) * 10:6d} ns   ({c[i] - c[i - 1]} clocks)")
