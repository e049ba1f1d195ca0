"""Micro-benchmark of the grouped 3x3 convolutions of the level-3 ResBlock (8 x 64 x 64 x 128, dilations 1 / 3 / 15) through the C ABI:
rua_conv_fwd_group as ONE conv_band128m launch (round 5) against the grouped conv_dmap grid it replaces, in the two forms the engine issues -
"first" (shared input, bias, statistics sum v / sum v^2) and "dgrad" (own inputs, ReLU mask from an aux tensor, statistics sum g / sum g * aux) -
timed back to back (operands warm) and behind a 512 MB sweep (cold, as inside the step), alternating rounds in one process.
Usage: python tools/bench_band128.py   (BB_REPS=30)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    dev = torch.device("cuda", 0)
    reps = int(os.environ.get("BB_REPS", "30"))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sweep = torch.zeros(128 << 20, dtype=torch.float32, device=dev)
    ws = torch.zeros(16 << 20, dtype=torch.float32, device=dev)
    N, HW, Cc, dils = {"3": (8, 64, 128, [1, 3, 15]), "2": (8, 128, 64, [1, 3, 15, 31]), "4": (8, 32, 256, [1, 3, 15])}[os.environ.get("BB_LEVEL", "3")]    # the level of the network
    g = torch.Generator(device="cpu").manual_seed(0)
    nb = len(dils)
    xs = [torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16) for _ in dils]
    aux = [torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16) for _ in dils]
    wts = [(torch.randn((9, Cc, Cc), generator=g) / (3 * Cc ** 0.5)).to(dev).to(torch.bfloat16) for _ in dils]
    ys = [torch.zeros((N, HW, HW, Cc), device=dev, dtype=torch.bfloat16) for _ in dils]
    bias = [torch.randn(Cc, device=dev) for _ in dils]
    msc = [torch.rand(Cc, device=dev) + 0.5 for _ in dils]
    msh = [0.3 * torch.randn(Cc, device=dev) for _ in dils]
    stats = [torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev) for _ in dils]
    flops = nb * 2.0 * N * HW * HW * Cc * Cc * 9

    def group(kind):
        arr = (L.ConvDesc * nb)()
        for b in range(nb):
            d = arr[b]
            d.nseg = 1
            sg = d.seg[0]
            src = xs[0] if kind == "first" else xs[b]
            sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = src.data_ptr(), wts[b].data_ptr(), Cc, HW, HW, 0, dils[b], 9
            d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, HW, HW, Cc, 1, L.RUA_BF16
            d.y, d.out_stride, d.OH, d.OW = ys[b].data_ptr(), 1, HW, HW
            d.stats, d.stats_replicas = stats[b].data_ptr(), 32
            if kind == "first":
                d.bias, d.stats_mode = bias[b].data_ptr(), 1
            else:
                d.aux, d.aux_mode, d.mscale, d.mshift, d.stats_mode = aux[b].data_ptr(), 2, msc[b].data_ptr(), msh[b].data_ptr(), 2
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        return arr

    def timed(arr, cold):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
        for r in range(reps + 3):
            if cold:
                sweep.add_(1.0)
            if r >= 3:
                e[2 * (r - 3)].record()
            lib.call("rua_conv_fwd_group", arr, nb, s)
            if r >= 3:
                e[2 * (r - 3) + 1].record()
        torch.cuda.synchronize()
        t = sorted(e[2 * i].elapsed_time(e[2 * i + 1]) * 1e3 for i in range(reps))
        return t[len(t) // 2], t[0]

    variants = [int(v) for v in os.environ.get("BB_VARIANTS", "7").split(",")]      # values of the tuning key conv_band128m to time against 0 (3: the kernel at both levels; level 2: 1 = conv_band64m)
    for kind in ("first", "dgrad"):
        arr = group(kind)
        outs = {}
        for v in variants + [0]:
            lib.set_tuning(conv_band128m=v)
            for st in stats:
                st.zero_()
            lib.call("rua_conv_fwd_group", arr, nb, s)
            torch.cuda.synchronize()
            outs[v] = ([y.float().clone() for y in ys], [st.view(32, -1).sum(0).clone() for st in stats], lib.raw("rua_conv_group_last_band")())
        for v in variants:
            for b in range(nb):
                dy = (outs[v][0][b] - outs[0][0][b]).abs().max().item() / outs[0][0][b].abs().max().item()
                ds = ((outs[v][1][b] - outs[0][1][b]).abs().max() / outs[0][1][b].abs().max()).item()
                print(f"{kind} member {b} (d = {dils[b]}): band128 form {v} vs conv_dmap: output {dy:.2e} of scale, statistics {ds:.2e}   (band flags {outs[v][2]} / {outs[0][2]})")
        for rnd in range(2):
            for v, name in [(v, "conv_band128m=%d" % v) for v in variants] + [(0, "conv_band128m=0  ")]:
                lib.set_tuning(conv_band128m=v)
                for cold in (False, True):
                    med, best = timed(arr, cold)
                    print(f"{kind:6s} {name:16s} {'cold' if cold else 'warm'}: median {med:6.1f} us, best {best:6.1f} us  = {flops / med / 1e6:6.1f} TFLOP/s", flush=True)
    lib.set_tuning(conv_band128m=13)


if __name__ == "__main__":
    main()
