"""Micro-benchmark of the deepest levels' 3x3 weight gradient through the C ABI (rua_conv_wgrad: wgrad_img / wgrad_imgs / the generic tiled kernel by RUA_TUNE_WGRAD_ROWS):
microseconds per launch, HIP events around REPS back-to-back launches.  Usage: python tools/bench_wgrad_img.py   (BT_REPS=50; RUA_LIB_PATH=<experiment build>)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    dev = torch.device("cuda", 0)
    reps = int(os.environ.get("BT_REPS", "50"))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for (N, H, Cs, Cout) in ((8, 16, 512, 512), (8, 8, 1024, 1024)):
        a = torch.randn((N, H, H, Cs), device=dev).to(torch.bfloat16)
        dy = torch.randn((N, H, H, Cout), device=dev).to(torch.bfloat16)
        dw = torch.zeros((9, Cout, Cs), device=dev)
        ws = torch.empty((64 << 20) // 4, dtype=torch.float32, device=dev)
        d = L.WgradDesc()
        d.a, d.C, d.Hs, d.Ws = a.data_ptr(), Cs, H, H
        d.dy, d.Cout, d.H, d.W = dy.data_ptr(), Cout, H, H
        d.N, d.stride, d.dil, d.taps, d.dtype = N, 1, 1, 9, L.RUA_BF16
        d.dw, d.workspace, d.workspace_bytes = dw.data_ptr(), ws.data_ptr(), ws.numel() * 4
        flag = torch.ones(4, dtype=torch.int32, device=dev)
        d.overwrite_dev = flag.data_ptr()                         # as in a whole step: dW is stored, not added to
        d.defer = 1                                               # the launch alone (K-slice slabs stay unreduced)
        for cold in (0, 1):
            for _ in range(3):
                lib.call("rua_conv_wgrad", C.byref(d), s)
            torch.cuda.synchronize()
            tot = 0.0
            n = reps if not cold else 10
            if cold:
                for _ in range(n):
                    flush.fill_(1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); lib.call("rua_conv_wgrad", C.byref(d), s); e1.record()
                    torch.cuda.synchronize()
                    tot += e0.elapsed_time(e1) * 1e3
                us = tot / n
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    lib.call("rua_conv_wgrad", C.byref(d), s)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / n
            print(f"rua_conv_wgrad {N}x{H}x{H} {Cs}->{Cout} {'cold (512 MB flushed, single launches incl. event pair)' if cold else 'back to back'}: {us:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
