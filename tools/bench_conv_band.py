"""Micro-benchmark of the d6 residual atrous block's forward convolutions through the C ABI (model2.py:15-34 at 8 x 256 x 256 x 32):
the grouped first convolutions (conv_strip32_g: four dilations, BatchNorm + ReLU on load, statistics) and the summed second
convolutions - rua_conv_fwd_sum as ONE conv_band32 launch against the four accumulating conv_strip launches.
Usage: python tools/bench_conv_band.py      (under rocprofv3: `rocprofv3 ... -- python3 tools/bench_conv_band.py`)
Environment: BB_SHAPE=N,H,W  BB_C=32|64  BB_DILS=1,3,15,31  BB_REPS=50  BB_ONLY=first|sum|each  (C = 64: conv_band64 only)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    dev = torch.device("cuda", 0)
    N, H, W = (int(v) for v in os.environ.get("BB_SHAPE", "8,256,256").split(","))
    dils = [int(v) for v in os.environ.get("BB_DILS", "1,3,15,31").split(",")]
    reps = int(os.environ.get("BB_REPS", "50"))
    only = os.environ.get("BB_ONLY")
    Cc, nb, M = int(os.environ.get("BB_C", "32")), len(dils), N * H * W
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((N, H, W, Cc), generator=g).to(dev).to(torch.bfloat16)
    y1 = [torch.randn((N, H, W, Cc), generator=g).to(dev).to(torch.bfloat16) for _ in range(nb)]
    w = [(torch.randn((9, Cc, Cc), generator=g) / 17).to(dev).to(torch.bfloat16) for _ in range(2 * nb)]
    out = torch.zeros((N, H, W, Cc), device=dev, dtype=torch.bfloat16)
    bias = torch.randn(Cc, device=dev)
    gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.3
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    R = 32
    keep = []

    def fold_of(t):
        st = torch.zeros(R * 2 * Cc, dtype=torch.float64, device=dev)
        lib.call("rua_col_stats", t.data_ptr(), M, Cc, st.data_ptr(), R, L.RUA_BF16, s)
        co = torch.zeros(4, Cc, dtype=torch.float32, device=dev)
        mm, mv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        f = L.BnFold()
        f.stats, f.replicas, f.count, f.bessel_n, f.eps, f.momentum = st.data_ptr(), R, float(M), float(M), 1e-3, 0.99
        f.gamma, f.beta, f.moving_mean, f.moving_var = gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr()
        f.scale, f.shift, f.mean, f.rstd = (co[i].data_ptr() for i in range(4))
        keep.extend([st, co, mm, mv, f])
        return f

    def desc(src, wt, dst, dil, fold):
        d = L.ConvDesc()
        d.nseg = 1
        sg = d.seg[0]
        sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = src.data_ptr(), wt.data_ptr(), Cc, H, W, 0, dil, 9
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cc, 1, L.RUA_BF16
        d.y, d.out_stride, d.OH, d.OW, d.bias = dst.data_ptr(), 1, H, W, bias.data_ptr()
        d.in_fold, d.in_relu = C.addressof(fold), 1
        return d

    fx = fold_of(x)
    first = (L.ConvDesc * nb)()
    stats1 = [torch.zeros(R * 2 * Cc, dtype=torch.float64, device=dev) for _ in range(nb)]
    for b in range(nb):
        d = desc(x, w[b], y1[b], dils[b], fx)
        d.stats, d.stats_mode, d.stats_replicas = stats1[b].data_ptr(), 1, R
        C.memmove(C.byref(first, b * C.sizeof(L.ConvDesc)), C.byref(d), C.sizeof(L.ConvDesc))
    second = (L.ConvDesc * nb)()
    for b in range(nb):
        d = desc(y1[b], w[nb + b], out, dils[b], fold_of(y1[b]))
        d.accumulate = 1 if b else 0
        if b == 0:
            d.aux, d.aux_mode = x.data_ptr(), 1
        C.memmove(C.byref(second, b * C.sizeof(L.ConvDesc)), C.byref(d), C.sizeof(L.ConvDesc))
    flops = 2.0 * M * Cc * Cc * 9 * nb
    mb = M * Cc * 2 / 1e6

    def timed(name, fn, passes):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{name:46s} {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s  {passes} tensor passes = {passes * mb / us / 1e3 * 1e3:6.0f} GB/s", flush=True)

    print(f"{N}x{H}x{W}x{Cc}, dilations {dils}: {flops / 1e9:.1f} GFLOP per stage, {mb:.1f} MB per tensor")
    if only in (None, "first") and Cc == 32:
        for dbg in [int(v) for v in os.environ.get("BB_DBG", "0").split(",")]:
            lib.set_tuning(band_dbg=dbg)                  # conv_strip32s ablations: 4 no row DMAs in the loop, 8 no stores, 16 no MFMAs
            timed(f"first convs, grouped (conv_strip32s_g) dbg={dbg}", lambda: lib.call("rua_conv_fwd_group", first, nb, s), 2 * nb)
        lib.set_tuning(band_dbg=0)
    if only in (None, "sum"):
        lib.set_tuning(conv_band=1, band_stag=int(os.environ.get("BB_STAG", "1")))      # 0: conv_band32, 4 / 6: conv_band32s with that many ring slots
        for dbg in [int(v) for v in os.environ.get("BB_DBG", "0").split(",")]:
            lib.set_tuning(band_dbg=dbg)
            timed(f"second convs, rua_conv_fwd_sum (conv_band32) dbg={dbg}", lambda: lib.call("rua_conv_fwd_sum", second, nb, s), nb + 2)
        lib.set_tuning(band_dbg=0)
        assert lib.raw("rua_conv_sum_last_kernel")() == (1 if Cc == 32 else 2)
    if only in (None, "each") and Cc == 32:
        lib.set_tuning(conv_band=0)
        timed("second convs, member by member (conv_strip32)", lambda: lib.call("rua_conv_fwd_sum", second, nb, s), 3 * nb)
        lib.set_tuning(conv_band=1)


if __name__ == "__main__":
    main()
