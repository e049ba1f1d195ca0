"""Micro-benchmark of the memory-bound tail kernels of the step through the C ABI, at cfg3's shapes (8 x 256 x 256): the heads
(model2.py:144-191: rua_head_fwd / rua_head_fwd_loss / rua_head_dz / rua_head_bwd_sums), the stem (model2.py:101: rua_stem_fwd /
rua_stem_fwd_stats / rua_stem_bwd).  Every row: microseconds per launch (HIP events around REPS back-to-back launches) and the
algorithmic bytes / that time.  Usage: python tools/bench_tail.py   (BT_REPS=50)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    dev = torch.device("cuda", 0)
    B, H, W = 8, 256, 256
    HW, M = H * W, B * H * W
    reps = int(os.environ.get("BT_REPS", "50"))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(0)
    dt = L.RUA_BF16

    def timed(name, nbytes, fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{name:58s} {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s", flush=True)

    x32 = torch.randn((M, 32), generator=g).to(dev).to(torch.bfloat16)
    for Cout, act, an in ((6, L.ACT_SOFTMAX, "softmax"), (6, L.ACT_SIGMOID, "sigmoid"), (3, L.ACT_SIGMOID, "sigmoid")):
        w = (torch.randn((Cout, 32), generator=g) / 6).to(dev)
        b = torch.randn(Cout, generator=g).to(dev)
        z = torch.zeros((M, Cout), device=dev)
        p = torch.zeros((M, Cout), device=dev)
        y = torch.rand((M, Cout), generator=g).to(dev)
        sums = torch.zeros(B * Cout * 6, dtype=torch.float64, device=dev)
        met = torch.zeros(8, dtype=torch.float64, device=dev)
        fb = M * (64 + 4 * Cout * 2)
        timed(f"rua_head_fwd            32->{Cout} {an}", fb, lambda: lib.call("rua_head_fwd", x32.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), p.data_ptr(), M, 32, Cout, act, dt, s))
        fb = M * (64 + 4 * Cout * 3)
        timed(f"rua_head_fwd_loss sums  32->{Cout} {an}", fb, lambda: lib.call("rua_head_fwd_loss", x32.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), p.data_ptr(), y.data_ptr(), sums.data_ptr(), None, B, HW, 32, Cout, act, dt, s))
        timed(f"rua_head_fwd_loss sums+metrics 32->{Cout} {an}", fb, lambda: lib.call("rua_head_fwd_loss", x32.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), p.data_ptr(), y.data_ptr(), sums.data_ptr(), met.data_ptr(), B, HW, 32, Cout, act, dt, s))
        sr = torch.zeros(9 * B * Cout * 6, dtype=torch.float64, device=dev)
        timed(f"rua_head_fwd_loss_rep sums x 8 copies + metrics 32->{Cout} {an}", fb, lambda: lib.call("rua_head_fwd_loss_rep", x32.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), p.data_ptr(), y.data_ptr(), sr.data_ptr(), 8, met.data_ptr(), B, HW, 32, Cout, act, dt, s))
        timed(f"rua_head_fwd_loss_rep sums x 8 copies 32->{Cout} {an}", fb, lambda: lib.call("rua_head_fwd_loss_rep", x32.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), p.data_ptr(), y.data_ptr(), sr.data_ptr(), 8, None, B, HW, 32, Cout, act, dt, s))
        coef = torch.rand(B * Cout * 3, generator=g).to(dev)
        dz = torch.zeros((M, Cout), device=dev)
        timed(f"rua_head_dz tanimoto    {Cout} {an}", M * 4 * Cout * 3, lambda: lib.call("rua_head_dz", L.LOSS_TANIMOTO, act, p.data_ptr(), y.data_ptr(), coef.data_ptr(), None, 0.125, B, HW, Cout, dz.data_ptr(), s))
        dx = torch.zeros((M, 32), device=dev, dtype=torch.bfloat16)
        dw, db, ds = torch.zeros(Cout * 32, device=dev), torch.zeros(Cout, device=dev), torch.zeros(32, device=dev)
        scratch = torch.zeros(8 << 20, device=dev)
        bb = M * (64 * 2 + 4 * Cout)
        for mask in (0, 1):
            timed(f"rua_head_bwd_sums partials mask={mask} 32->{Cout}", bb, lambda: lib.call("rua_head_bwd_sums", x32.data_ptr(), dz.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, dw.data_ptr(), db.data_ptr(), ds.data_ptr(), scratch.data_ptr(), scratch.numel() * 4, M, 32, Cout, dt, mask, s))
        timed(f"rua_head_bwd_sums partials no dx 32->{Cout}", M * (64 + 4 * Cout), lambda: lib.call("rua_head_bwd_sums", x32.data_ptr(), dz.data_ptr(), w.data_ptr(), None, 0, dw.data_ptr(), db.data_ptr(), None, scratch.data_ptr(), scratch.numel() * 4, M, 32, Cout, dt, 0, s))
        for hb in (256, 1024):
            lib.set_tuning(head_blocks=hb)
            timed(f"rua_head_bwd_sums partials mask=1 32->{Cout} ({hb} blocks)", bb, lambda: lib.call("rua_head_bwd_sums", x32.data_ptr(), dz.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, dw.data_ptr(), db.data_ptr(), ds.data_ptr(), scratch.data_ptr(), scratch.numel() * 4, M, 32, Cout, dt, 1, s))
        lib.set_tuning(head_blocks=0)
        timed(f"rua_head_bwd_sums atomics  mask=1 32->{Cout}", bb, lambda: lib.call("rua_head_bwd_sums", x32.data_ptr(), dz.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, dw.data_ptr(), db.data_ptr(), ds.data_ptr(), None, 0, M, 32, Cout, dt, 1, s))
    # stem
    Cin = 6
    xin = torch.rand((M, Cin), generator=g).to(dev)
    w = (torch.randn((32, Cin), generator=g) / 3).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    yo = torch.zeros((M, 32), device=dev, dtype=torch.bfloat16)
    R = 32
    st = torch.zeros(R * 2 * 32, dtype=torch.float64, device=dev)
    sb = M * (4 * Cin + 64)
    timed("rua_stem_fwd            6->32", sb, lambda: lib.call("rua_stem_fwd", xin.data_ptr(), w.data_ptr(), b.data_ptr(), yo.data_ptr(), M, Cin, 32, dt, s))
    timed("rua_stem_fwd_stats      6->32", sb, lambda: lib.call("rua_stem_fwd_stats", xin.data_ptr(), w.data_ptr(), b.data_ptr(), yo.data_ptr(), M, Cin, 32, dt, st.data_ptr(), R, s))
    dw, db = torch.zeros(32 * Cin, device=dev), torch.zeros(32, device=dev)
    timed("rua_stem_bwd            6->32", sb, lambda: lib.call("rua_stem_bwd", xin.data_ptr(), yo.data_ptr(), dw.data_ptr(), db.data_ptr(), M, Cin, 32, dt, s))
    lib.set_tuning(stem_reg=0, head_fwd3=0)
    Cout, act = 6, L.ACT_SOFTMAX
    w6 = (torch.randn((Cout, 32), generator=g) / 6).to(dev)
    b6 = torch.randn(Cout, generator=g).to(dev)
    z = torch.zeros((M, Cout), device=dev); p = torch.zeros((M, Cout), device=dev)
    y = torch.rand((M, Cout), generator=g).to(dev)
    sums = torch.zeros(B * Cout * 6, dtype=torch.float64, device=dev)
    met = torch.zeros(8, dtype=torch.float64, device=dev)
    timed("rua_head_fwd            32->6 softmax (head_fwd2 form)", M * (64 + 48), lambda: lib.call("rua_head_fwd", x32.data_ptr(), w6.data_ptr(), b6.data_ptr(), z.data_ptr(), p.data_ptr(), M, 32, Cout, act, dt, s))
    timed("rua_head_fwd_loss sums+metrics 32->6 softmax (head_fwd2 form)", M * (64 + 72), lambda: lib.call("rua_head_fwd_loss", x32.data_ptr(), w6.data_ptr(), b6.data_ptr(), z.data_ptr(), p.data_ptr(), y.data_ptr(), sums.data_ptr(), met.data_ptr(), B, HW, 32, Cout, act, dt, s))
    sr = torch.zeros(9 * B * Cout * 6, dtype=torch.float64, device=dev)
    for bpc in (1, 3, 4, 8):
        lib.set_tuning(head_fwd3=1, head_fwd3_bpc=bpc)
        timed(f"rua_head_fwd_loss_rep x 8 copies 32->6 softmax (head_fwd3, {bpc} blocks per CU)", M * (64 + 72), lambda: lib.call("rua_head_fwd_loss_rep", x32.data_ptr(), w6.data_ptr(), b6.data_ptr(), z.data_ptr(), p.data_ptr(), y.data_ptr(), sr.data_ptr(), 8, None, B, HW, 32, Cout, act, dt, s))
    timed("rua_stem_fwd_stats      6->32 (LDS-weights form)", sb, lambda: lib.call("rua_stem_fwd_stats", xin.data_ptr(), w.data_ptr(), b.data_ptr(), yo.data_ptr(), M, Cin, 32, dt, st.data_ptr(), R, s))


if __name__ == "__main__":
    main()
