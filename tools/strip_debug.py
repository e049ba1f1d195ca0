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
    modes = a[4:] if len(a) > 4 else ["plain", "bn", "bn_stats", "mask_stats2", "relu", "mask", "acc", "bn_acc", "mask_stats2_nocoef"]
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
            if mode not in ("mask_stats2", "mask", "acc", "mask_stats2_nocoef"):      # (a data gradient has no bias: the forms conv_strip32s serves)
                d.bias = bias.data_ptr()
            if mode in ("bn", "bn_stats"):
                d.in_scale, d.in_shift, d.in_relu = sc.data_ptr(), sh.data_ptr(), 1
                d.stats_mode = 1 if mode == "bn_stats" else 0
            elif mode == "mask_stats2":
                d.aux, d.aux_mode, d.mscale, d.mshift, d.stats_mode = aux.data_ptr(), 2, sc.data_ptr(), sh.data_ptr(), 2
            elif mode == "relu":
                d.out_relu = 1
            elif mode == "mask":                            # the heads' data gradients: mask = the ReLU output itself
                d.aux, d.aux_mode = aux.data_ptr(), 2
            elif mode == "mask_stats2_nocoef":
                d.aux, d.aux_mode, d.stats_mode = aux.data_ptr(), 2, 2
            elif mode == "acc":
                d.accumulate = 1
            elif mode == "bn_acc":
                d.accumulate = 1
                d.in_scale, d.in_shift, d.in_relu = sc.data_ptr(), sh.data_ptr(), 1
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


def group_check():
    """the d6 block's first convs (in_fold, statistics) and data gradients (mask, sums) as GROUPED launches: conv_strip32s against conv_strip32"""
    lib = L.lib()
    dev = torch.device("cuda", 0)
    N, H, W, Cc = [int(v) for v in os.environ.get("SD_SHAPE", "8,256,256").split(",")] + [32]
    dils = [1, 3, 15, 31]
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((N, H, W, Cc), generator=g).to(dev).to(torch.bfloat16)
    auxs = [torch.randn((N, H, W, Cc), generator=g).to(dev).to(torch.bfloat16) for _ in dils]
    ws = [(torch.randn((9, Cc, Cc), generator=g) / 17).to(dev).to(torch.bfloat16) for _ in dils]
    bias = torch.randn(Cc, device=dev)
    gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.3
    sc, sh = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.3
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    R, M = 32, N * H * W
    for kind in ("first", "dgrad"):
        res = []
        for stag in (0, 1):
            lib.set_tuning(strip_stag=stag)
            keep = []
            st = torch.zeros(R * 2 * Cc, dtype=torch.float64, device=dev)
            lib.call("rua_col_stats", x.data_ptr(), M, Cc, st.data_ptr(), R, L.RUA_BF16, s)
            arr = (L.ConvDesc * len(dils))()
            ys, stats, cos = [], [], []
            for b, dil in enumerate(dils):
                y = torch.full((N, H, W, Cc), 3.0, device=dev, dtype=torch.bfloat16)
                so = torch.zeros(R * 2 * Cc, dtype=torch.float64, device=dev)
                d = L.ConvDesc()
                d.nseg = 1
                sg = d.seg[0]
                sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = x.data_ptr(), ws[b].data_ptr(), Cc, H, W, 0, dil, 9
                d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cc, 1, L.RUA_BF16
                d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
                d.stats, d.stats_replicas = so.data_ptr(), R
                if kind == "first":
                    co = torch.zeros(4, Cc, dtype=torch.float32, device=dev)
                    mm, mv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
                    f = L.BnFold()
                    f.stats, f.replicas, f.count, f.bessel_n, f.eps, f.momentum = st.data_ptr(), R, float(M), float(M), 1e-3, 0.99
                    f.gamma, f.beta, f.moving_mean, f.moving_var = gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr()
                    f.scale, f.shift, f.mean, f.rstd = (co[i].data_ptr() for i in range(4))
                    keep.extend([co, mm, mv, f])
                    cos.append((co, mm, mv))
                    d.bias, d.in_fold, d.in_relu, d.stats_mode = bias.data_ptr(), C.addressof(f), 1, 1
                else:
                    d.aux, d.aux_mode, d.mscale, d.mshift, d.stats_mode = auxs[b].data_ptr(), 2, sc.data_ptr(), sh.data_ptr(), 2
                C.memmove(C.byref(arr, b * C.sizeof(L.ConvDesc)), C.byref(d), C.sizeof(L.ConvDesc))
                ys.append(y); stats.append(so)
            for _ in range(int(os.environ.get("SD_REPS", "1"))):
                for so in stats:
                    so.zero_()
                lib.call("rua_conv_fwd_group", arr, len(dils), s)
            torch.cuda.synchronize()
            res.append(([y.float().cpu().numpy() for y in ys], [so.cpu().numpy().reshape(R, 2 * Cc).sum(0) for so in stats],
                        [[t.cpu().numpy() for t in c] for c in cos]))
        for b, dil in enumerate(dils):
            a0, a1 = res[0][0][b], res[1][0][b]
            bad = a0 != a1
            s0, s1 = res[0][1][b], res[1][1][b]
            line = f"{kind} d={dil}: {bad.sum()} of {bad.size} outputs differ; statistics max rel diff {np.abs(s0 - s1).max() / (np.abs(s0).max() + 1e-30):.3g}"
            if kind == "first":
                c0, c1 = res[0][2][b], res[1][2][b]
                line += "; published coefficients / moving statistics max diff " + " ".join(f"{np.abs(u - v).max():.2g}" for u, v in zip(c0, c1))
            print(line, flush=True)
            if bad.any():
                n, h, w_, c = np.nonzero(bad)
                print("   images", np.unique(n), "rows", np.unique(h)[:24], "cols", np.unique(w_)[:16], "channels", np.unique(c)[:16])
    lib.set_tuning(strip_stag=1)


if __name__ == "__main__":
    if os.environ.get("SD_GROUP"):
        group_check()
    else:
        main()
