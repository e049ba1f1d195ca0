"""Micro-benchmark of the grouped 3x3 weight gradients of the level-3 / level-4 ResBlocks (8 x 64 x 64 x 128 and 8 x 32 x 32 x 256, dilations 1 / 3 / 15) through the C ABI:
rua_conv_wgrad_group with the reductions deferred (as in the step), for values of the tuning key wgrad_rows - 31: wgrad_rows128 (level 3) / wgrad_dmap (level 4), the
round-4 kernels; 63: + wgrad_rowsx<1> at level 4; 127: wgrad_rowsx at both - timed back to back (warm) and behind a 512 MB sweep (cold, as inside the step), and the
results of the forms against each other.
Usage: python tools/bench_wgrad_rows.py   (BW_REPS=30, BW_LEVELS=3,4, BW_VARIANTS=31,127)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    dev = torch.device("cuda", 0)
    reps = int(os.environ.get("BW_REPS", "30"))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sweep = torch.zeros(128 << 20, dtype=torch.float32, device=dev)
    variants = [int(v) for v in os.environ.get("BW_VARIANTS", "31,127").split(",")]
    default = lib.get_tuning("wgrad_rows")
    for level in os.environ.get("BW_LEVELS", "3,4").split(","):
        N, HW, Cc, dils = {"1": (8, 256, 32, [1, 3, 15, 31]), "2": (8, 128, 64, [1, 3, 15, 31]), "3": (8, 64, 128, [1, 3, 15]), "4": (8, 32, 256, [1, 3, 15])}[level]
        if os.environ.get("BW_DILS"):                        # (what the short chains of the large dilations cost: BW_DILS=1,1,1,1 against the default)
            dils = [int(v) for v in os.environ["BW_DILS"].split(",")]
        g = torch.Generator(device="cpu").manual_seed(0)
        nb = len(dils)
        a = torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16)
        dys = [torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16) for _ in dils]
        dws = [torch.zeros(9 * Cc * Cc, dtype=torch.float32, device=dev) for _ in dils]
        flops = nb * 2.0 * N * HW * HW * Cc * Cc * 9
        keep = []

        def group():
            arr = (L.WgradDesc * nb)()
            for b in range(nb):
                d = arr[b]
                d.a, d.C, d.Hs, d.Ws, d.dy, d.Cout, d.H, d.W = a.data_ptr(), Cc, HW, HW, dys[b].data_ptr(), Cc, HW, HW
                d.N, d.stride, d.dil, d.taps, d.dtype = N, 1, dils[b], 9, L.RUA_BF16
                d.group_members = nb
                d.dw = dws[b].data_ptr()
                ws = torch.zeros(lib.raw("rua_wgrad_workspace_bytes")(C.byref(d)) // 4 + 16, dtype=torch.float32, device=dev)
                keep.append(ws)
                d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
            return arr

        def run_full(arr):                                   # with the reductions, for the comparison of results
            for b in range(nb):
                arr[b].defer = 0
                dws[b].zero_()
            lib.call("rua_conv_wgrad_group", arr, nb, s)
            torch.cuda.synchronize()
            return [w.clone() for w in dws], lib.raw("rua_wgrad_group_last_grids")()

        def timed(arr, cold):
            for b in range(nb):
                arr[b].defer = 1
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
            for r in range(reps + 3):
                if cold:
                    sweep.add_(1.0)
                if r >= 3:
                    e[2 * (r - 3)].record()
                lib.call("rua_conv_wgrad_group", arr, nb, s)
                if r >= 3:
                    e[2 * (r - 3) + 1].record()
            torch.cuda.synchronize()
            t = sorted(e[2 * i].elapsed_time(e[2 * i + 1]) * 1e3 for i in range(reps))
            return t[len(t) // 2], t[0]

        outs = {}
        for v in variants:
            lib.set_tuning(wgrad_rows=v)
            arr = group()
            kinds = [lib.raw("rua_wgrad_kind")(C.byref(arr[b])) for b in range(nb)]
            outs[v] = run_full(arr)
            print(f"level {level} wgrad_rows={v}: kinds {kinds}, grids {outs[v][1]}")
        for v in variants[1:]:
            for b in range(nb):
                df = (outs[v][0][b] - outs[variants[0]][0][b]).abs().max().item() / outs[variants[0]][0][b].abs().max().item()
                print(f"level {level} member {b} (d = {dils[b]}): wgrad_rows={v} vs {variants[0]}: {df:.2e} of scale")
        for rnd in range(2):
            for v in variants:
                lib.set_tuning(wgrad_rows=v)
                arr = group()
                for cold in (False, True):
                    med, best = timed(arr, cold)
                    print(f"level {level} wgrad_rows={v:3d} {'cold' if cold else 'warm'}: median {med:6.1f} us, best {best:6.1f} us  = {flops / med / 1e6:6.1f} TFLOP/s", flush=True)
    lib.set_tuning(wgrad_rows=default)


if __name__ == "__main__":
    main()
