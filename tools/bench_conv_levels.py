"""Micro-benchmark of the 3x3 convolutions of levels 3 - 6 (conv_dmap: 8 x 64 x 64 x 128 ... 8 x 8 x 8 x 1024) through the C ABI:
one convolution (bias + statistics) per launch, timed back to back (operands warm in L2 / Infinity Cache) and behind a 512 MB
sweep (operands cold, as inside the step), and the second-stage sum of a ResBlock as ONE multi-segment launch against the
accumulate chain.  Usage: python tools/bench_conv_levels.py   (BL_REPS, BL_LEVELS=3,4 ...)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402

LEVELS = {3: (64, 128, [1, 3, 15]), 4: (32, 256, [1, 3, 15]), 5: (16, 512, [1]), 6: (8, 1024, [1])}


def main():
    lib = L.lib()
    dev = torch.device("cuda", 0)
    reps = int(os.environ.get("BL_REPS", "30"))
    levels = [int(v) for v in os.environ.get("BL_LEVELS", "3,4,5,6").split(",")]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sweep = torch.zeros(128 << 20, dtype=torch.float32, device=dev)            # 512 MB: larger than the Infinity Cache
    ws = torch.zeros(16 << 20, dtype=torch.float32, device=dev)
    N = 8
    g = torch.Generator(device="cpu").manual_seed(0)
    for lv in levels:
        HW, Cc, dils = LEVELS[lv]
        xs = [torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16) for _ in dils]
        wts = [(torch.randn((9, Cc, Cc), generator=g) / (3 * Cc ** 0.5)).to(dev).to(torch.bfloat16) for _ in dils]
        y = torch.zeros((N, HW, HW, Cc), device=dev, dtype=torch.bfloat16)
        bias = torch.randn(Cc, device=dev)
        stats = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
        flops = 2.0 * N * HW * HW * Cc * Cc * 9

        def desc(members, accumulate=False, with_stats=True):
            d = L.ConvDesc()
            d.nseg = len(members)
            for i, b in enumerate(members):
                sg = d.seg[i]
                sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = xs[b].data_ptr(), wts[b].data_ptr(), Cc, HW, HW, 0, dils[b], 9
            d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, HW, HW, Cc, 1, L.RUA_BF16
            d.y, d.out_stride, d.OH, d.OW, d.bias = y.data_ptr(), 1, HW, HW, bias.data_ptr()
            d.accumulate = 1 if accumulate else 0
            if with_stats:
                d.stats, d.stats_mode, d.stats_replicas = stats.data_ptr(), 1, 32
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
            return d

        def timed(ds, cold):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
            for r in range(reps + 3):
                if cold:
                    sweep.add_(1.0)
                if r >= 3:
                    e[2 * (r - 3)].record()
                for d in ds:
                    lib.call("rua_conv_fwd", C.byref(d), s)
                if r >= 3:
                    e[2 * (r - 3) + 1].record()
            torch.cuda.synchronize()
            t = sorted(e[2 * i].elapsed_time(e[2 * i + 1]) * 1e3 for i in range(reps))
            return t[len(t) // 2]

        print(f"level {lv}: 8 x {HW} x {HW} x {Cc}, {flops / 1e9:.2f} GFLOP per conv (event pair overhead included, ~5 us)")
        for b, dil in enumerate(dils):
            d = desc([b])
            kid, ks = lib.raw("rua_conv_kernel_id")(C.byref(d)), 0
            tw, tc = timed([d], False), timed([d], True)
            ks = lib.raw("rua_conv_last_ksplit")()
            print(f"  d={dil:2d} single        kernel {kid} ksplit {ks:2d} | warm {tw:6.1f} us {flops / tw / 1e6:5.0f} TF/s | cold {tc:6.1f} us {flops / tc / 1e6:5.0f} TF/s")
        if len(dils) > 1:
            ys = [torch.zeros((N, HW, HW, Cc), device=dev, dtype=torch.bfloat16) for _ in dils]
            sts = [torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev) for _ in dils]
            aux = torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16)
            sc, sh = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.3
            for kind in ("first convs (bias, statistics)", "data gradients (mask, statistics 2)"):
                arr = (L.ConvDesc * len(dils))()
                for b in range(len(dils)):
                    d = desc([0 if kind.startswith("first") else b])
                    d.seg[0].w, d.seg[0].dil = wts[b].data_ptr(), dils[b]
                    d.y, d.stats = ys[b].data_ptr(), sts[b].data_ptr()
                    if not kind.startswith("first"):
                        d.bias = None
                        d.aux, d.aux_mode, d.mscale, d.mshift, d.stats_mode = aux.data_ptr(), 2, sc.data_ptr(), sh.data_ptr(), 2
                    C.memmove(C.byref(arr, b * C.sizeof(L.ConvDesc)), C.byref(d), C.sizeof(L.ConvDesc))
                res = []
                for cold in (False, True):
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
                    for r in range(reps + 3):
                        if cold:
                            sweep.add_(1.0)
                        if r >= 3:
                            e[2 * (r - 3)].record()
                        lib.call("rua_conv_fwd_group", arr, len(dils), s)
                        if r >= 3:
                            e[2 * (r - 3) + 1].record()
                    torch.cuda.synchronize()
                    t = sorted(e[2 * i].elapsed_time(e[2 * i + 1]) * 1e3 for i in range(reps))
                    res.append(t[len(t) // 2])
                f = flops * len(dils)
                print(f"  group of {len(dils)}: {kind:36s} grids {lib.raw('rua_conv_group_last_grids')()} chain {lib.raw('rua_conv_group_last_chain')()} | warm {res[0]:6.1f} us {f / res[0] / 1e6:5.0f} TF/s | cold {res[1]:6.1f} us {f / res[1] / 1e6:5.0f} TF/s")
            chain = [desc([b], accumulate=b > 0, with_stats=b == len(dils) - 1) for b in range(len(dils))]
            one = desc(list(range(len(dils))))
            for name, ds in (("accumulate chain", chain), ("one launch, K x %d" % len(dils), [one])):
                tw, tc = timed(ds, False), timed(ds, True)
                f = flops * len(dils)
                print(f"  sum of {len(dils)}: {name:18s} | warm {tw:6.1f} us {f / tw / 1e6:5.0f} TF/s | cold {tc:6.1f} us {f / tc / 1e6:5.0f} TF/s")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
