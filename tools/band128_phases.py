"""In-kernel section times of conv_band128m (diagnostic build -DRUA_B128_STAMPS; wave 0 of every block sums s_memrealtime intervals):
  RUA_BUILD_FLAGS=-DRUA_B128_STAMPS RUA_BUILD_OUT=scratch/stamps.so python -m resunet_a_mltsk_keras_amd.build --force
  RUA_LIB_PATH=$PWD/scratch/stamps.so python tools/band128_phases.py [first|dgrad]
Sections: 0 prologue, 1 phase top (wait for the weights + barrier), 2 weight fragments -> registers + barrier + weight DMA issue, 3 fragment reads + MFMAs of
the stages, 4 wait for the next stage's rows (+ in-place BatchNorm), 5 barrier + row DMA issue of stage 1, 6 member epilogues, 7 tail."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "first"
    form = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # value of the tuning key conv_band128m (experiment builds with more than one form)
    lib = L.lib()
    dev = torch.device("cuda", 0)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N, HW, Cc, dils = {"3": (8, 64, 128, [1, 3, 15]), "4": (8, 32, 256, [1, 3, 15]), "2": (8, 128, 64, [1, 3, 15, 31])}[os.environ.get("BP_LEVEL", "3")]      # BP_LEVEL=4: conv_band128m<256,32,128,32>
    nb = len(dils)
    g = torch.Generator(device="cpu").manual_seed(0)
    xs = [torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16) for _ in dils]
    aux = [torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16) for _ in dils]
    wts = [(torch.randn((9, Cc, Cc), generator=g) / (3 * Cc ** 0.5)).to(dev).to(torch.bfloat16) for _ in dils]
    ys = [torch.zeros((N, HW, HW, Cc), device=dev, dtype=torch.bfloat16) for _ in dils]
    bias = [torch.randn(Cc, device=dev) for _ in dils]
    msc = [torch.rand(Cc, device=dev) + 0.5 for _ in dils]
    msh = [0.3 * torch.randn(Cc, device=dev) for _ in dils]
    stats = [torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev) for _ in dils]
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
    njobs = N * (HW // 4) * 2 if Cc == 128 else (N * (HW // 8) * 8 if Cc == 256 else N * (HW // 4))
    stamps = torch.zeros(njobs * 8 + njobs * 8 * 2 * 4 * 2, dtype=torch.int64, device=dev)
    lib.set_tuning(dbg_ptr=stamps.data_ptr(), conv_band128m=form)
    for _ in range(20):
        lib.call("rua_conv_fwd_group", arr, nb, s)
    torch.cuda.synchronize()
    t = stamps[:njobs * 8].view(njobs, 8).double().cpu()
    ev = stamps[njobs * 8:].view(njobs, 8, 2, 4, 2).double().cpu()
    lib.set_tuning(dbg_ptr=0, conv_band128m=13)
    names = ["prologue", "phase top wait+barrier", "weights->regs", "stage MFMAs", "wait next rows", "stage-1 barrier", "epilogues", "tail"]
    print(f"{kind}: s_memrealtime ticks (100 MHz): median over {njobs} blocks; us = ticks / 100")
    tot = 0.0
    for i, n in enumerate(names):
        med = t[:, i].median().item()
        tot += med
        print(f"  {n:24s} {med / 100.0:7.2f} us   (min {t[:, i].min().item() / 100.0:6.2f}, max {t[:, i].max().item() / 100.0:6.2f})")
    print(f"  sum of medians           {tot / 100.0:7.2f} us")
    # timeline of the 8 waves of one block through phase 4 (us since the block's first wave left the stage-0 barrier) and the shader clock
    for job in (0, njobs // 2 + 3):
        e = ev[job]
        t0 = e[:, 0, 0, 0].min()
        print(f"  block {job}: wave: stage 0 [barrier left, DMAs issued, MFMAs done, rows waited] stage 1 [...]  (us)")
        for w in range(8):
            row = " ".join("%6.2f" % ((e[w, sp, k, 0] - t0) / 100.0) for sp in range(2) for k in range(4))
            ghz = (e[w, 1, 3, 1] - e[w, 0, 0, 1]) / max(e[w, 1, 3, 0] - e[w, 0, 0, 0], 1.0) / 10.0
            print(f"    wave {w}: {row}   shader clock {ghz:5.2f} GHz")


if __name__ == "__main__":
    main()
