"""Where does conv_strip32s differ from conv_strip32?  Runs one 3x3 convolution through both (tuning key strip_stag) and prints
the pattern of the differing elements (rows, columns, channels).  python tools/strip_debug.py [N H W dil mode]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402


def main():
    a = sys.argv[1:]
    N, H, W, dil = (int(v) for v in a[:4]) if len(a) >= 4 else (1, 256, 256, 1)
    modes = a[4:] if len(a) > 4 else ["plain", "bn", "bn_stats", "mask_stats2", "relu"]
    lib = L.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    Cc = 32
    x = torch.randn((N, H, W, Cc), generator=g).to(dev).to(torch.bfloat16)
    aux = torch.randn((N, H, W, Cc), generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn((9, Cc, Cc), generator=g) / 17).to(dev).to(torch.bfloat16)
    sc = torch.rand(Cc, device=dev) + 0.5
    sh = torch.randn(Cc, device=dev) * 0.3
    bias = torch.randn(Cc, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for mode in modes:
        outs, sts = [], []
        for stag in (0, 1):
            y = torch.full((N, H, W, Cc), 7.0, device=dev, dtype=torch.bfloat16)
            stats = torch.zeros(8 * 2 * Cc, dtype=torch.float64, device=dev)
            d = L.ConvDesc()
            d.nseg = 1
            sg = d.seg[0]
            sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = x.data_ptr(), w.data_ptr(), Cc, H, W, 0, dil, 9
            d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cc, 1, L.RUA_BF16
            d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
            d.stats, d.stats_replicas = stats.data_ptr(), 8
            d.bias = bias.data_ptr()
            if mode in ("bn", "bn_stats"):
                d.in_scale, d.in_shift, d.in_relu = sc.data_ptr(), sh.data_ptr(), 1
                d.stats_mode = 1 if mode == "bn_stats" else 0
            elif mode == "mask_stats2":
                d.aux, d.aux_mode, d.mscale, d.mshift, d.stats_mode = aux.data_ptr(), 2, sc.data_ptr(), sh.data_ptr(), 2
            elif mode == "relu":
                d.out_relu = 1
            lib.set_tuning(strip_stag=stag)
            lib.call("rua_conv_fwd", C.byref(d), s)
            torch.cuda.synchronize()
            outs.append(y.float().cpu().numpy())
            sts.append(stats.cpu().numpy().reshape(8, 2 * Cc).sum(0))
        a0, a1 = outs
        bad = ~np.isclose(a0, a1, rtol=0, atol=0)
        print(f"mode {mode}: {bad.sum()} of {bad.size} elements differ; max |diff| {np.nanmax(np.abs(a0 - a1)):.3g}; nan {np.isnan(a1).sum()}", flush=True)
        if bad.any():
            n, h, w_, c = np.nonzero(bad)
            print("  images", np.unique(n)[:8], "rows", np.unique(h)[:40], "...", len(np.unique(h)))
            print("  cols", np.unique(w_)[:40], "...", len(np.unique(w_)))
            print("  channels", np.unique(c))
            hh = np.bincount(h, minlength=H)
            print("  bad per row (first 40 rows):", hh[:40])
            ww = np.bincount(w_, minlength=W)
            print("  bad per 32-column block:", ww.reshape(-1, 32).sum(1))
        ds = np.abs(sts[0] - sts[1]).max() / (np.abs(sts[0]).max() + 1e-30)
        print(f"  statistics: relative difference {ds:.3g}")
    lib.set_tuning(strip_stag=1)


if __name__ == "__main__":
    main()
