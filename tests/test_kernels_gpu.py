"""Kernel-level parity through the C ABI (librua_hip.so) against plain PyTorch-CPU fp32 references.
fp32 storage must match to ~1e-5 relative (exact-fp32 MFMA, different summation order);
bf16 storage is compared with the bf16-rounded inputs, tolerance 2e-2 of the output scale."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from resunet_a_mltsk_keras_amd import _lib as L  # noqa: E402


def dev():
    return torch.device("cuda", 0)


def tdt(dt):
    return torch.bfloat16 if dt == L.RUA_BF16 else torch.float32


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_dev(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev()).to(tdt(dt)).contiguous()


def rel_err(got, exp):
    got = np.asarray(got, np.float64); exp = np.asarray(exp, np.float64)
    return float(np.abs(got - exp).max() / (np.abs(exp).max() + 1e-12))


def tol(dt):
    return 2e-2 if dt == L.RUA_BF16 else 2e-5


def rnd(dt, a):
    """what the device sees after storage rounding"""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(tdt(dt)).float()


def ref_conv_nhwc(x, w_tco_ci, bias, dil, taps, stride=1, up=0):
    """x NHWC float tensor, w [taps][Cout][Cin] -> NHWC"""
    xt = x.permute(0, 3, 1, 2)
    if up:
        k = 1 << up
        xt = xt.repeat_interleave(k, 2).repeat_interleave(k, 3)
    k = 3 if taps == 9 else 1
    Cout, Cin = w_tco_ci.shape[1], w_tco_ci.shape[2]
    w = w_tco_ci.reshape(k, k, Cout, Cin).permute(2, 3, 0, 1)
    y = F.conv2d(xt.double(), w.double(), None if bias is None else bias.double(), stride=stride, padding=dil * (k // 2), dilation=dil)
    return y.permute(0, 2, 3, 1)


CONV_CASES = [
    # N, H, W, [(C, up, dil, taps)], Cout, stride
    (2, 16, 16, [(32, 0, 1, 9)], 32, 1),
    (1, 20, 12, [(32, 0, 3, 9)], 64, 1),
    (2, 16, 16, [(16, 0, 15, 9)], 32, 1),
    (1, 40, 40, [(64, 0, 31, 9)], 64, 1),
    (2, 8, 8, [(128, 0, 1, 9)], 256, 1),
    (2, 16, 16, [(32, 0, 1, 1)], 64, 2),
    (2, 16, 16, [(16, 1, 1, 1), (32, 0, 1, 1)], 32, 1),
    (1, 16, 16, [(8, 0, 1, 1), (8, 1, 1, 1), (8, 2, 1, 1), (8, 3, 1, 1), (32, 0, 1, 1)], 32, 1),
    (1, 16, 16, [(32, 0, 1, 9), (32, 0, 3, 9), (32, 0, 15, 9)], 32, 1),
    (3, 8, 8, [(40, 0, 1, 9)], 8, 1),
    (8, 32, 32, [(128, 0, 3, 9)], 256, 1),     # bf16: conv_dmap with 64-row tiles (128 tiles of 128 x 128 would leave half the CUs idle)
    # bf16: conv_pw (narrow 1x1 convs, >= 65536 pixels: per-wave streaming, no LDS in the loop)
    (1, 256, 256, [(16, 1, 1, 1), (32, 0, 1, 1)], 32, 1),
    (1, 256, 256, [(8, 0, 1, 1), (8, 1, 1, 1), (8, 2, 1, 1), (8, 3, 1, 1), (32, 0, 1, 1)], 32, 1),
    (2, 256, 128, [(32, 0, 1, 1)], 8, 1),
    (1, 320, 224, [(64, 0, 1, 1)], 24, 1),     # not a power of two: dense segments only
    # bf16: conv_small (1x1 convs over <= 4096 pixels: a block = 32 pixels x 64 channels over the whole K, fragments straight from global memory)
    (8, 8, 8, [(1024, 0, 1, 1)], 256, 1),
    (8, 8, 8, [(256, 0, 1, 1), (256, 1, 1, 1), (256, 2, 1, 1), (256, 3, 1, 1), (1024, 0, 1, 1)], 1024, 1),      # the PSPPooling fuse conv at the bottleneck
    (8, 8, 8, [(512, 0, 1, 1)], 1024, 2),      # stride 2 (model2.py:103-111)
    (3, 5, 7, [(48, 0, 1, 1), (16, 0, 1, 1)], 40, 1),      # ragged: 105 pixels, 40 channels, K = 4 k-steps over two sources (a wave without work)
    (8, 16, 16, [(256, 1, 1, 1), (512, 0, 1, 1)], 512, 1),  # upsampled + skip source (model2.py:81-94)
]


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(case, dt):
    N, H, W, segs, Cout, stride = case
    rng = np.random.default_rng(1)
    lib = L.lib()
    d = L.ConvDesc()
    d.nseg = len(segs)
    keep = []
    exp = 0
    for i, (Cs, up, dil, taps) in enumerate(segs):
        Hs, Ws = (H * stride) >> up, (W * stride) >> up
        x = rng.standard_normal((N, Hs, Ws, Cs)).astype(np.float32)
        w = (rng.standard_normal((taps, Cout, Cs)) / np.sqrt(taps * Cs)).astype(np.float32)
        xd, wd = to_dev(x, dt), to_dev(w, dt)
        keep += [xd, wd]
        s = d.seg[i]
        s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd.data_ptr(), wd.data_ptr(), Cs, Hs, Ws, up, dil, taps
        exp = exp + ref_conv_nhwc(rnd(dt, x), rnd(dt, w), None, dil, taps, stride, up)
    bias = rng.standard_normal(Cout).astype(np.float32)
    more = [rng.standard_normal(Cout).astype(np.float32) for _ in range(2)]          # bias_more: added after `bias`, in order
    md = [torch.from_numpy(m).to(dev()) for m in more]
    for i, m in enumerate(md):
        d.bias_more[i] = m.data_ptr()
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    bd, rd = torch.from_numpy(bias).to(dev()), to_dev(res, dt)
    bias = (bias + more[0]) + more[1]
    y = torch.empty((N, H, W, Cout), dtype=tdt(dt), device=dev())
    stats = torch.zeros(2 * 2 * Cout, dtype=torch.float64, device=dev())
    d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cout, stride, dt
    d.bias, d.aux, d.aux_mode = bd.data_ptr(), rd.data_ptr(), 1
    d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
    d.stats, d.stats_mode, d.stats_replicas = stats.data_ptr(), 1, 2
    if dt == L.RUA_BF16 and case[:3] == (8, 32, 32):
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 2 and lib.raw("rua_conv_tile_bm")(C.byref(d)) == 64
    if dt == L.RUA_BF16 and N * H * W >= 65536 and all(t == 1 for _, _, _, t in segs):
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 4
    if dt == L.RUA_BF16 and N * H * W <= 4096 and Cout >= 32 and all(t == 1 and c % 16 == 0 for c, _, _, t in segs):
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 6
    lib.call("rua_conv_fwd", C.byref(d), stream())
    torch.cuda.synchronize()
    exp = (exp + torch.from_numpy(bias).double() + rnd(dt, res).double()).numpy()
    got = y.float().cpu().numpy()
    assert rel_err(got, exp) < tol(dt)
    st = stats.cpu().numpy().reshape(2, 2 * Cout).sum(0)
    assert rel_err(st[:Cout], exp.sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4
    assert rel_err(st[Cout:], (exp ** 2).sum(axis=(0, 1, 2))) < 5 * tol(dt)


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_conv_split_k_matches_single_pass(dt):
    """Small output grid + long K (the 8x8x1024 level): with a workspace the launcher splits K over blocks
    (one fp32 slab per K slice + summing finisher); result, statistics and mask must match the torch reference, and a
    second launch must reproduce the first bit for bit (no atomics)."""
    rng = np.random.default_rng(12)
    lib = L.lib()
    N, H, W, Cs, Cout = 2, 8, 8, 256, 128
    x = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
    w = (rng.standard_normal((9, Cout, Cs)) / np.sqrt(9 * Cs)).astype(np.float32)
    aux = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    xd, wd, ad, bd = to_dev(x, dt), to_dev(w, dt), to_dev(aux, dt), torch.from_numpy(bias).to(dev())
    y = torch.empty((N, H, W, Cout), dtype=tdt(dt), device=dev())
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev())
    ws = torch.full((16 * N * H * W * Cout + 1024,), float("nan"), dtype=torch.float32, device=dev())   # slab contents on entry are irrelevant
    ws[-1024:] = 0                                            # ... except the ticket counters in the last 4 KiB: zero once
    d = L.ConvDesc()
    d.nseg = 1
    s = d.seg[0]
    s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd.data_ptr(), wd.data_ptr(), Cs, H, W, 0, 1, 9
    d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cout, 1, dt
    d.bias, d.aux, d.aux_mode = bd.data_ptr(), ad.data_ptr(), 2
    d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
    d.stats, d.stats_mode, d.stats_replicas = stats.data_ptr(), 2, 1
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    assert lib.raw("rua_conv_workspace_bytes")(C.byref(d)) * 16 + 4096 == ws.numel() * 4
    lib.call("rua_conv_fwd", C.byref(d), stream())
    torch.cuda.synchronize()
    y1 = y.clone()
    stats.zero_()
    lib.call("rua_conv_fwd", C.byref(d), stream())
    torch.cuda.synchronize()
    assert torch.equal(y1.view(torch.uint8), y.view(torch.uint8))
    assert int(ws[-1024:].view(torch.int32).abs().max()) == 0          # counters (RUA_DMAP_FUSED_FINISH=1 only) are back to zero
    a = rnd(dt, aux).double().numpy()
    exp = (ref_conv_nhwc(rnd(dt, x), rnd(dt, w), torch.from_numpy(bias), 1, 9).numpy()) * (a > 0)
    assert rel_err(y.float().cpu().numpy(), exp) < tol(dt)
    st = stats.cpu().numpy()
    assert rel_err(st[:Cout], exp.sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4
    assert rel_err(st[Cout:], (exp * a).sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_conv_mask_accumulate_strided_out(dt):
    """dgrad-style epilogues: ReLU mask from aux with scale/shift, statistics sum g / sum g*aux,
    accumulate into a strided (stride-2 scatter) output."""
    rng = np.random.default_rng(2)
    lib = L.lib()
    N, H, W, Cs, Cout = 2, 8, 8, 32, 32
    x = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
    w = (rng.standard_normal((9, Cout, Cs)) / 17).astype(np.float32)
    aux = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    ms = rng.standard_normal(Cout).astype(np.float32)
    mt = rng.standard_normal(Cout).astype(np.float32)
    xd, wd, ad = to_dev(x, dt), to_dev(w, dt), to_dev(aux, dt)
    msd, mtd = torch.from_numpy(ms).to(dev()), torch.from_numpy(mt).to(dev())
    y = torch.empty((N, H, W, Cout), dtype=tdt(dt), device=dev())
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev())
    d = L.ConvDesc()
    d.nseg = 1
    s = d.seg[0]
    s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd.data_ptr(), wd.data_ptr(), Cs, H, W, 0, 3, 9
    d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cout, 1, dt
    d.aux, d.aux_mode, d.mscale, d.mshift = ad.data_ptr(), 2, msd.data_ptr(), mtd.data_ptr()
    d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
    d.stats, d.stats_mode = stats.data_ptr(), 2
    lib.call("rua_conv_fwd", C.byref(d), stream())
    torch.cuda.synchronize()
    conv = ref_conv_nhwc(rnd(dt, x), rnd(dt, w), None, 3, 9).numpy()
    a = rnd(dt, aux).numpy().astype(np.float64)
    mask = (np.float32(ms) * a.astype(np.float32) + np.float32(mt)) > 0
    exp = conv * mask
    assert rel_err(y.float().cpu().numpy(), exp) < tol(dt)
    st = stats.cpu().numpy()
    assert rel_err(st[:Cout], exp.sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4
    assert rel_err(st[Cout:], (exp * a).sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4
    # strided accumulate: 1x1 conv of a half-resolution tensor scattered to even pixels of a full-res buffer
    x2 = rng.standard_normal((N, H // 2, W // 2, Cs)).astype(np.float32)
    w1 = (rng.standard_normal((1, Cout, Cs)) / 6).astype(np.float32)
    base = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    x2d, w1d, yd = to_dev(x2, dt), to_dev(w1, dt), to_dev(base, dt)
    d2 = L.ConvDesc()
    d2.nseg = 1
    s = d2.seg[0]
    s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = x2d.data_ptr(), w1d.data_ptr(), Cs, H // 2, W // 2, 0, 1, 1
    d2.N, d2.H, d2.W, d2.Cout, d2.stride, d2.dtype = N, H // 2, W // 2, Cout, 1, dt
    d2.accumulate = 1
    d2.y, d2.out_stride, d2.OH, d2.OW = yd.data_ptr(), 2, H, W
    lib.call("rua_conv_fwd", C.byref(d2), stream())
    torch.cuda.synchronize()
    exp2 = rnd(dt, base).double().numpy()
    exp2[:, ::2, ::2, :] += ref_conv_nhwc(rnd(dt, x2), rnd(dt, w1), None, 1, 1).numpy()
    assert rel_err(yd.float().cpu().numpy(), exp2) < tol(dt)


WGRAD_CASES = [
    # N, Hs, Ws, C, Cout, stride, dil, taps
    (2, 16, 16, 32, 32, 1, 1, 9),
    (1, 24, 20, 64, 32, 1, 3, 9),
    (2, 16, 16, 16, 48, 1, 15, 9),
    (2, 16, 16, 32, 64, 2, 1, 1),
    (3, 8, 8, 128, 72, 1, 1, 1),
    (1, 12, 12, 8, 8, 1, 1, 9),
    # wide levels (bf16: wgrad_dmap, 128 x 128 tiles, LDS-DMA stages; power-of-two maps)
    (2, 16, 16, 128, 128, 1, 1, 9),
    (1, 32, 32, 128, 256, 1, 3, 9),
    (1, 16, 16, 256, 128, 1, 15, 9),
    (2, 8, 8, 256, 128, 1, 1, 1),
    (3, 4, 4, 128, 128, 1, 1, 9),
    (4, 32, 32, 256, 256, 1, 3, 9),      # 36 tiles, 64 stages: the shape class the dispatch gives to wgrad_dmap
    (8, 32, 32, 256, 128, 1, 15, 9),
    (4, 64, 64, 128, 128, 1, 3, 9),      # 9 tiles, 256 stages: the 64x64x128 level's class
]


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(case, dt):
    N, Hs, Ws, Cs, Cout, stride, dil, taps = case
    H, W = Hs // stride, Ws // stride
    rng = np.random.default_rng(3)
    a = rng.standard_normal((N, Hs, Ws, Cs)).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    ad, dyd = to_dev(a, dt), to_dev(dy, dt)
    dw = torch.zeros((taps, Cout, Cs), dtype=torch.float32, device=dev())
    d = L.WgradDesc()
    d.a, d.C, d.Hs, d.Ws = ad.data_ptr(), Cs, Hs, Ws
    d.dy, d.Cout, d.H, d.W = dyd.data_ptr(), Cout, H, W
    d.N, d.stride, d.dil, d.taps, d.dtype = N, stride, dil, taps, dt
    d.dw = dw.data_ptr()
    if dt == L.RUA_BF16 and case[:5] == (4, 32, 32, 256, 256):
        assert L.lib().raw("rua_wgrad_kind")(C.byref(d)) == 2
    nine = dt == L.RUA_BF16 and case[:5] == (4, 64, 64, 128, 128)
    if nine:                                                  # 9 tiles: wgrad_kernel by default (faster inside the step); still served by wgrad_dmap on request
        assert L.lib().raw("rua_wgrad_kind")(C.byref(d)) == 0
        L.lib().set_tuning(wgd_mintiles=9)
        assert L.lib().raw("rua_wgrad_kind")(C.byref(d)) == 2
    try:
        L.lib().call("rua_conv_wgrad", C.byref(d), stream())
    finally:
        if nine:
            L.lib().set_tuning(wgd_mintiles=10)
    torch.cuda.synchronize()
    w = torch.zeros((taps, Cout, Cs), dtype=torch.float64, requires_grad=True)
    y = ref_conv_nhwc(rnd(dt, a).double(), w, None, dil, taps, stride)
    y.backward(rnd(dt, dy).double())
    assert rel_err(dw.cpu().numpy(), w.grad.numpy()) < tol(dt)
    # with a workspace the K slices go through fp32 slabs summed in a fixed order: the same gradient, bit-reproducible and
    # ADDED to what dW holds (the atomic K split was order-dependent)
    ws = torch.empty((32 << 20) // 4, dtype=torch.float32, device=dev())
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if L.lib().raw("rua_wgrad_kind")(C.byref(d)) in (0, 2):
        outs = []
        for rep in range(2):
            dw.fill_(1.0)
            ws.uniform_(-1e3, 1e3)                               # slabs need no initialisation
            L.lib().call("rua_conv_wgrad", C.byref(d), stream())
            torch.cuda.synchronize()
            outs.append(dw.cpu().numpy().copy())
        assert np.array_equal(outs[0], outs[1])
        assert rel_err(outs[0] - 1.0, w.grad.numpy()) < tol(dt)


@pytest.mark.parametrize("case", [(2, 64, 64, 32, 1), (1, 128, 64, 32, 3), (2, 64, 128, 32, 15), (1, 64, 64, 32, 31),
                                  (2, 40, 256, 32, 31), (1, 256, 256, 32, 15), (3, 7, 128, 32, 3), (8, 16, 256, 32, 1),      # full-width rows at C = 32: wgrad_rows32
                                  (2, 64, 64, 64, 1), (1, 64, 128, 64, 15), (1, 128, 64, 64, 31), (3, 64, 64, 64, 3),
                                  (2, 128, 128, 64, 31), (1, 128, 128, 64, 1), (3, 40, 128, 64, 3),      # C = 64 on 128-pixel rows: wgrad_rows64
                                  (2, 64, 64, 128, 1), (1, 64, 64, 128, 15), (3, 40, 64, 128, 3), (8, 64, 64, 128, 31)])  # C = 128 on 64-pixel rows: wgrad_rows128
@pytest.mark.parametrize("rows", [127, 255])
def test_wgrad_all_taps_kernel(case, rows):
    """Top-level weight gradient (C = Cout in {32, 64}, W % 64 == 0, bf16): all nine taps from one LDS halo,
    deterministic partial reduction; must add into dW and match autograd.  rows = the tuning key wgrad_rows: bit 7 deals the rows of wgrad_rows32 / wgrad_rows64
    (full-width rows at C = 32 / 64) as one stream of slots instead of (chain, segment) jobs."""
    L.lib().set_tuning(wgrad_rows=rows)
    try:
        _wgrad_all_taps(case)
    finally:
        L.lib().set_tuning(wgrad_rows=WGRAD_ROWS_DEFAULT)


def _wgrad_all_taps(case):
    N, H, W, Cc, dil = case
    dt = L.RUA_BF16
    rng = np.random.default_rng(13)
    a = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    ad, dyd = to_dev(a, dt), to_dev(dy, dt)
    base = rng.standard_normal((9, Cc, Cc)).astype(np.float32)
    dw = torch.from_numpy(base).to(dev())
    ws = torch.empty(256 * 9 * 32 * Cc, dtype=torch.float32, device=dev())
    d = L.WgradDesc()
    d.a, d.C, d.Hs, d.Ws = ad.data_ptr(), Cc, H, W
    d.dy, d.Cout, d.H, d.W = dyd.data_ptr(), Cc, H, W
    d.N, d.stride, d.dil, d.taps, d.dtype = N, 1, dil, 9, dt
    d.dw, d.workspace, d.workspace_bytes = dw.data_ptr(), ws.data_ptr(), ws.numel() * 4
    assert L.lib().raw("rua_wgrad_kind")(C.byref(d)) == 1
    outs = []
    for rep in range(2):
        dw.copy_(torch.from_numpy(base))
        L.lib().call("rua_conv_wgrad", C.byref(d), stream())
        torch.cuda.synchronize()
        outs.append(dw.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])                    # deterministic (no atomics)
    w = torch.zeros((9, Cc, Cc), dtype=torch.float64, requires_grad=True)
    y = ref_conv_nhwc(rnd(dt, a).double(), w, None, dil, 9, 1)
    y.backward(rnd(dt, dy).double())
    assert rel_err(outs[0] - base, w.grad.numpy()) < 2e-3


WGRAD_ROWS_DEFAULT = 127                                       # csrc/common.h: Tuning::wgrad_rows


@pytest.mark.parametrize("case", [(2, 64, 64, 128, 1), (1, 64, 64, 128, 15), (3, 40, 64, 128, 3), (8, 64, 64, 128, 31), (1, 7, 64, 128, 3),     # C = 128 on 64-pixel rows
                                  (8, 32, 32, 256, 1), (8, 32, 32, 256, 3), (8, 32, 32, 256, 15), (2, 32, 32, 256, 15), (4, 20, 32, 256, 16), (6, 48, 32, 256, 7),
                                  (2, 3, 32, 256, 1)])                                                                                            # C = 256 on 32-pixel rows, image pairs
def test_wgrad_row_stream_kernel(case):
    """wgrad_rowsx (round 5): the whole-row weight gradient with the rows of all images and dilation chains dealt to the blocks as ONE stream of slots (a chain's rows +
    one separator row of zeros), at C = 128 on 64-pixel rows and - two images per stage, a 128-channel half of the input per block - at C = 256 on 32-pixel rows
    (the level-4 ResBlock; was wgrad_dmap).  Deterministic block partials, adds into dW, matches autograd; short images and dilations up to the zero gap of a slot."""
    N, H, W, Cc, dil = case
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(113)
    a = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    ad, dyd = to_dev(a, dt), to_dev(dy, dt)
    base = rng.standard_normal((9, Cc, Cc)).astype(np.float32)
    dw = torch.from_numpy(base).to(dev())
    d = L.WgradDesc()
    d.a, d.C, d.Hs, d.Ws = ad.data_ptr(), Cc, H, W
    d.dy, d.Cout, d.H, d.W = dyd.data_ptr(), Cc, H, W
    d.N, d.stride, d.dil, d.taps, d.dtype = N, 1, dil, 9, dt
    d.dw = dw.data_ptr()
    lib.set_tuning(wgrad_rows=127)
    try:
        ws = torch.empty(lib.raw("rua_wgrad_workspace_bytes")(C.byref(d)) // 4, dtype=torch.float32, device=dev())
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        assert lib.raw("rua_wgrad_kind")(C.byref(d)) == 1
        outs = []
        for rep in range(2):
            dw.copy_(torch.from_numpy(base))
            ws.uniform_(-1e3, 1e3)                              # partials need no initialisation
            lib.call("rua_conv_wgrad", C.byref(d), stream())
            torch.cuda.synchronize()
            outs.append(dw.cpu().numpy().copy())
    finally:
        lib.set_tuning(wgrad_rows=WGRAD_ROWS_DEFAULT)
    assert np.array_equal(outs[0], outs[1])                    # deterministic (no atomics)
    w = torch.zeros((9, Cc, Cc), dtype=torch.float64, requires_grad=True)
    y = ref_conv_nhwc(rnd(dt, a).double(), w, None, dil, 9, 1)
    y.backward(rnd(dt, dy).double())
    assert rel_err(outs[0] - base, w.grad.numpy()) < 2e-3


@pytest.mark.parametrize("case", [(8, 8, 1024, 1024), (8, 16, 512, 512), (2, 16, 128, 64), (16, 8, 64, 128), (8, 8, 64, 64), (24, 8, 128, 128), (4, 16, 64, 192),
                                  (4, 16, 512, 256), (6, 16, 256, 512), (10, 16, 384, 352), (16, 8, 512, 256), (24, 8, 256, 544)])      # 16 x 16, > 512 pixels, >= 128 tiles of 32 x 32: wgrad_imgs (streamed chunks, no K slices)
def test_wgrad_whole_image_kernel(case):
    """The deepest levels' 3x3 weight gradient (8 x 8 and 16 x 16 maps, dilation 1, bf16: wgrad_img): whole images resident in LDS, all nine taps per block,
    512-pixel chunks as K slices through slabs - or, at 16 x 16 with enough 32 x 32 tiles to fill the chip, streamed through an LDS ring by one block (wgrad_imgs);
    adds into dW (or stores, under the first-writer flag), deterministic, matches autograd."""
    N, H, Cs, Cout = case
    dt = L.RUA_BF16
    rng = np.random.default_rng(17)
    a = rng.standard_normal((N, H, H, Cs)).astype(np.float32)
    dy = rng.standard_normal((N, H, H, Cout)).astype(np.float32)
    ad, dyd = to_dev(a, dt), to_dev(dy, dt)
    base = rng.standard_normal((9, Cout, Cs)).astype(np.float32)
    dw = torch.from_numpy(base).to(dev())
    ws = torch.empty((64 << 20) // 4, dtype=torch.float32, device=dev())
    flag = torch.zeros(4, dtype=torch.int32, device=dev())
    d = L.WgradDesc()
    d.a, d.C, d.Hs, d.Ws = ad.data_ptr(), Cs, H, H
    d.dy, d.Cout, d.H, d.W = dyd.data_ptr(), Cout, H, H
    d.N, d.stride, d.dil, d.taps, d.dtype = N, 1, 1, 9, dt
    d.dw, d.workspace, d.workspace_bytes = dw.data_ptr(), ws.data_ptr(), ws.numel() * 4
    d.overwrite_dev = flag.data_ptr()
    assert L.lib().raw("rua_wgrad_kind")(C.byref(d)) == 0
    streamed = N * H * H > 512 and (Cs // 32) * (Cout // 32) >= 128
    assert L.lib().raw("rua_wgrad_img_kind")(C.byref(d)) == (2 if streamed else 1)
    w = torch.zeros((9, Cout, Cs), dtype=torch.float64, requires_grad=True)
    y = ref_conv_nhwc(rnd(dt, a).double(), w, None, 1, 9, 1)
    y.backward(rnd(dt, dy).double())
    outs = []
    for rep in range(2):
        dw.copy_(torch.from_numpy(base))
        ws.uniform_(-1e3, 1e3)
        L.lib().call("rua_conv_wgrad", C.byref(d), stream())
        torch.cuda.synchronize()
        outs.append(dw.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])
    assert rel_err(outs[0] - base, w.grad.numpy()) < 2e-3
    if N * H * H == 512 or streamed:                           # no K slices: the block stores dW itself when the flag is up
        flag.fill_(1)
        L.lib().call("rua_conv_wgrad", C.byref(d), stream())
        torch.cuda.synchronize()
        assert rel_err(dw.cpu().numpy(), w.grad.numpy()) < 2e-3


PW_CASES = [
    # N, Hs, Ws, C, Cout, stride
    (2, 64, 64, 32, 32, 1), (1, 128, 128, 32, 8, 1), (2, 64, 64, 8, 32, 1), (2, 64, 64, 64, 64, 1), (2, 64, 64, 32, 64, 2),
    (1, 64, 64, 16, 64, 1), (3, 40, 24, 32, 32, 1), (2, 64, 64, 48, 24, 1), (4, 64, 64, 64, 16, 2), (8, 64, 64, 64, 40, 1),
    (1, 64, 32, 8, 8, 1),
]


@pytest.mark.parametrize("mode", [7, 1])
@pytest.mark.parametrize("case", PW_CASES)
def test_wgrad_pointwise_kernel(case, mode):
    """Narrow 1x1 weight gradients (C, Cout <= 64, bf16: wgrad_pw): per-wave streaming; the block sums leave as partials in front of the workspace's tail and are
    summed in a fixed order (tuning key wgrad_pw: bit 1, round 5 - for deferred reductions, the engine's case; bit 2: for calls that reduce right away too, as here) - or, = 1, through replica accumulators and a ticket in the tail of the workspace, which must be
    zero before the first call and is left zero by every call."""
    L.lib().set_tuning(wgrad_pw=mode)
    try:
        _wgrad_pointwise(case, mode)
    finally:
        L.lib().set_tuning(wgrad_pw=3)


def _wgrad_pointwise(case, mode):
    N, Hs, Ws, Cs, Cout, stride = case
    H, W = Hs // stride, Ws // stride
    dt = L.RUA_BF16
    rng = np.random.default_rng(17)
    a = rng.standard_normal((N, Hs, Ws, Cs)).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    ad, dyd = to_dev(a, dt), to_dev(dy, dt)
    base = rng.standard_normal((1, Cout, Cs)).astype(np.float32)
    dw = torch.from_numpy(base).to(dev())
    d = L.WgradDesc()
    d.a, d.C, d.Hs, d.Ws = ad.data_ptr(), Cs, Hs, Ws
    d.dy, d.Cout, d.H, d.W = dyd.data_ptr(), Cout, H, W
    d.N, d.stride, d.dil, d.taps, d.dtype = N, stride, 1, 1, dt
    d.dw = dw.data_ptr()
    assert L.lib().raw("rua_wgrad_kind")(C.byref(d)) == 0       # no workspace: the generic kernel
    nbytes = L.lib().raw("rua_wgrad_workspace_bytes")(C.byref(d))
    ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev())
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    assert L.lib().raw("rua_wgrad_kind")(C.byref(d)) == 3
    w = torch.zeros((1, Cout, Cs), dtype=torch.float64, requires_grad=True)
    y = ref_conv_nhwc(rnd(dt, a).double(), w, None, 1, 1, stride)
    y.backward(rnd(dt, dy).double())
    outs = []
    for rep in range(2):                                        # the second call runs on what the first left behind
        dw.copy_(torch.from_numpy(base))
        L.lib().call("rua_conv_wgrad", C.byref(d), stream())
        torch.cuda.synchronize()
        outs.append(dw.cpu().numpy().copy())
        assert rel_err(outs[-1] - base, w.grad.numpy()) < 2e-3
        assert float(ws[-(16 * 64 * 64 + 2048):].abs().max()) == 0.0
    if mode == 7:
        assert np.array_equal(outs[0], outs[1])                # block partials, fixed order: bit-reproducible


def test_conv_pointwise_group_members_share_a_grid():
    """rua_conv_fwd_group over narrow 1x1 convolutions of UNEQUAL size (round 5: the branch convolutions of the top-level PSPPooling, model2.py:47-68, and the per-source
    data gradients of its fuse conv): the conv_pw members (>= 65 536 pixels) run as ONE conv_pw_g grid, the smaller ones on their own kernels - every member bit for bit
    what a launch of its own gives (statistics: fp64 atomics of the same per-block sums)."""
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(31)
    shapes = [(8, 256, 256), (8, 128, 128), (8, 64, 64), (8, 32, 32)]
    Cs, Cout = 32, 8
    keep, descs = [], []
    for N, H, W in shapes:
        x = to_dev(rng.standard_normal((N, H, W, Cs)).astype(np.float32), dt)
        w = to_dev((rng.standard_normal((1, Cout, Cs)) / np.sqrt(Cs)).astype(np.float32), dt)
        bias = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32)).to(dev())
        y = torch.zeros((N, H, W, Cout), dtype=torch.bfloat16, device=dev())
        stats = torch.zeros(8 * 2 * Cout, dtype=torch.float64, device=dev())
        d = L.ConvDesc()
        d.nseg = 1
        sg = d.seg[0]
        sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = x.data_ptr(), w.data_ptr(), Cs, H, W, 0, 1, 1
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cout, 1, dt
        d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
        d.bias = bias.data_ptr()
        d.stats, d.stats_mode, d.stats_replicas = stats.data_ptr(), 1, 8
        keep.append((x, w, bias, y, stats)); descs.append(d)
    npw = sum(1 for d in descs if lib.raw("rua_conv_kernel_id")(C.byref(d)) == 4)
    assert npw == 2                                            # the 256 x 256 and 128 x 128 members
    sep = []
    for d, kp in zip(descs, keep):
        lib.call("rua_conv_fwd", C.byref(d), stream())
        torch.cuda.synchronize()
        sep.append((kp[3].clone(), kp[4].clone()))
        kp[3].zero_(); kp[4].zero_()
    arr = (L.ConvDesc * len(descs))()
    for i, d in enumerate(descs):
        C.memmove(C.byref(arr, i * C.sizeof(L.ConvDesc)), C.byref(d), C.sizeof(L.ConvDesc))
    lib.call("rua_conv_fwd_group", arr, len(descs), stream())
    torch.cuda.synchronize()
    assert lib.raw("rua_conv_group_last_grids")() <= len(descs) - (npw - 1)
    for (yy, st), kp in zip(sep, keep):
        assert torch.equal(kp[3], yy)
        assert np.allclose(kp[4].cpu().numpy().reshape(8, -1).sum(0), st.cpu().numpy().reshape(8, -1).sum(0), rtol=1e-12, atol=1e-9)


def test_wgrad_pointwise_group_is_one_grid():
    """rua_conv_wgrad_group over narrow 1x1 weight gradients with workspaces of their own (round 5: the per-source weight gradients of a concatenating
    1x1 conv, the branch convs of a PSPPooling - Graph.wgrad_pw_group): members of UNEQUAL size and channel counts run as one wgrad_pw_g grid per
    (NCO, NCI) form, bit for bit what the members give one by one apart from the replica atomics' order, against the fp64 gradient; replicas and
    tickets are left zero; members sharing a workspace fall back to one launch each."""
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(23)
    members = [(8, 64, 64, 32, 32), (8, 32, 32, 8, 32), (2, 64, 32, 16, 32), (8, 16, 16, 8, 32)]      # N, H, W, C, Cout: four grids of 64 / 4 / 2 / 1 blocks ... one form (1, 1)
    keep, descs, refs, bases, dws, wss = [], [], [], [], [], []
    for N, H, W, Cs, Cout in members:
        a = rng.standard_normal((N, H, W, Cs)).astype(np.float32); dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
        ad, dyd = to_dev(a, dt), to_dev(dy, dt)
        base = rng.standard_normal((1, Cout, Cs)).astype(np.float32)
        dw = torch.from_numpy(base).to(dev())
        d = L.WgradDesc()
        d.a, d.C, d.Hs, d.Ws = ad.data_ptr(), Cs, H, W
        d.dy, d.Cout, d.H, d.W = dyd.data_ptr(), Cout, H, W
        d.N, d.stride, d.dil, d.taps, d.dtype = N, 1, 1, 1, dt
        d.dw = dw.data_ptr()
        nbytes = lib.raw("rua_wgrad_workspace_bytes")(C.byref(d))
        ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev())
        d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
        assert lib.raw("rua_wgrad_kind")(C.byref(d)) == 3
        w = torch.zeros((1, Cout, Cs), dtype=torch.float64, requires_grad=True)
        y = ref_conv_nhwc(rnd(dt, a).double(), w, None, 1, 1, 1)
        y.backward(rnd(dt, dy).double())
        keep += [ad, dyd]; descs.append(d); refs.append(w.grad.numpy()); bases.append(base); dws.append(dw); wss.append(ws)
    arr = (L.WgradDesc * len(descs))()
    for i, d in enumerate(descs):
        C.memmove(C.byref(arr, i * C.sizeof(L.WgradDesc)), C.byref(d), C.sizeof(L.WgradDesc))
    for rep in range(2):                                        # the second call runs on what the first left behind
        for dw, base in zip(dws, bases):
            dw.copy_(torch.from_numpy(base))
        lib.call("rua_conv_wgrad_group", arr, len(descs), stream())
        assert lib.raw("rua_wgrad_group_last_grids")() == 1
        torch.cuda.synchronize()
        for dw, base, ref, ws in zip(dws, bases, refs, wss):
            assert rel_err(dw.cpu().numpy() - base, ref) < 2e-3
            assert float(ws[-(16 * 64 * 64 + 2048):].abs().max()) == 0.0
    # two members on ONE workspace (shared replicas): the library must not put them into one grid
    arr[1].workspace, arr[1].workspace_bytes = arr[0].workspace, arr[0].workspace_bytes
    for dw, base in zip(dws, bases):
        dw.copy_(torch.from_numpy(base))
    lib.call("rua_conv_wgrad_group", arr, len(descs), stream())
    assert lib.raw("rua_wgrad_group_last_grids")() == len(descs)
    torch.cuda.synchronize()
    for dw, base, ref in zip(dws, bases, refs):
        assert rel_err(dw.cpu().numpy() - base, ref) < 2e-3


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_weight_prep_layouts(dt):
    rng = np.random.default_rng(4)
    specs = [(9, 32, 16), (1, 8, 64), (9, 8, 8)]
    offs, total = [], 0
    for t, co, ci in specs:
        offs.append(total); total += (t * co * ci + 15) // 16 * 16
    master = rng.standard_normal(total).astype(np.float32)
    items = np.array([(o, o, t, co, ci, 0) for o, (t, co, ci) in zip(offs, specs)],
                     dtype=[("src", "<i8"), ("dst", "<i8"), ("taps", "<i4"), ("cout", "<i4"), ("c", "<i4"), ("pad", "<i4")])
    it = torch.from_numpy(np.frombuffer(items.tobytes(), dtype=np.uint8).copy()).to(dev())
    m = torch.from_numpy(master).to(dev())
    wf = torch.zeros(total, dtype=tdt(dt), device=dev()); wd = torch.zeros(total, dtype=tdt(dt), device=dev())
    L.lib().call("rua_weight_prep", m.data_ptr(), wf.data_ptr(), wd.data_ptr(), it.data_ptr(), len(specs), 9 * 32 * 16, dt, stream())
    torch.cuda.synchronize()
    for o, (t, co, ci) in zip(offs, specs):
        src = rnd(dt, master[o:o + t * co * ci]).numpy().reshape(t, co, ci)
        assert np.array_equal(wf[o:o + t * co * ci].float().cpu().numpy().reshape(t, co, ci), src)
        exp = src[::-1].transpose(0, 2, 1)
        assert np.array_equal(wd[o:o + t * co * ci].float().cpu().numpy().reshape(t, ci, co), exp)
    if dt == L.RUA_BF16:
        # rua_weight_prep_dgrad: the data-gradient layout alone, from the forward copy (the optimizer writes that one itself): same bytes, on the
        # register-transpose path (C, Cout multiples of 8; tiles that end inside a 64 x 64 block) and on the LDS path (C = 12)
        specs2 = [(9, 32, 16), (1, 8, 64), (9, 8, 8), (9, 128, 256), (1, 72, 200), (9, 16, 12), (1, 24, 4)]
        offs2, total2 = [], 0
        for t, co, ci in specs2:
            offs2.append(total2); total2 += (t * co * ci + 15) // 16 * 16
        master2 = rng.standard_normal(total2).astype(np.float32)
        items2 = np.array([(o, o, t, co, ci, 0) for o, (t, co, ci) in zip(offs2, specs2)], dtype=items.dtype)
        it2 = torch.from_numpy(np.frombuffer(items2.tobytes(), dtype=np.uint8).copy()).to(dev())
        m2 = torch.from_numpy(master2).to(dev())
        wf2 = torch.zeros(total2, dtype=tdt(dt), device=dev()); wd2 = torch.zeros_like(wf2); wd3 = torch.zeros_like(wf2)
        mx = max(t * co * ci for t, co, ci in specs2)
        L.lib().call("rua_weight_prep", m2.data_ptr(), wf2.data_ptr(), wd2.data_ptr(), it2.data_ptr(), len(specs2), mx, dt, stream())
        L.lib().call("rua_weight_prep_dgrad", wf2.data_ptr(), wd3.data_ptr(), it2.data_ptr(), len(specs2), mx, None, 0, dt, stream())
        bmap = []                                                             # ... and with a block map: the grid follows the tensors' sizes
        for i, (t, co, ci) in enumerate(specs2):
            nbk = L.lib().raw("rua_wprep_blocks")(t, co, ci)
            assert nbk == -(-(t * -(-co // 64) * -(-ci // 64)) // 8)
            for b in range(nbk):
                bmap += [i, 8 * b]
        bm = torch.tensor(bmap, dtype=torch.int32, device=dev())
        wd4 = torch.zeros_like(wf2)
        L.lib().call("rua_weight_prep_dgrad", wf2.data_ptr(), wd4.data_ptr(), it2.data_ptr(), len(specs2), mx, bm.data_ptr(), len(bmap) // 2, dt, stream())
        torch.cuda.synchronize()
        assert torch.equal(wd2.view(torch.int16), wd3.view(torch.int16)) and wd3.float().abs().sum().item() > 0
        assert torch.equal(wd2.view(torch.int16), wd4.view(torch.int16))
        assert L.lib().raw("rua_weight_prep_dgrad")(wf2.data_ptr(), wd3.data_ptr(), it2.data_ptr(), len(specs2), mx, None, 0, L.RUA_F32, None) != 0


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_dgrad_equals_autograd(dt):
    """conv with prepared dgrad weights == d/dx of the forward conv (3x3 dilated and 1x1)."""
    rng = np.random.default_rng(5)
    lib = L.lib()
    for (taps, dil) in [(9, 1), (9, 3), (1, 1)]:
        N, H, W, Cin, Cout = 2, 12, 12, 16, 32
        w = (rng.standard_normal((taps, Cout, Cin)) / 8).astype(np.float32)
        dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
        n = taps * Cout * Cin
        items = np.array([(0, 0, taps, Cout, Cin, 0)], dtype=[("src", "<i8"), ("dst", "<i8"), ("taps", "<i4"), ("cout", "<i4"), ("c", "<i4"), ("pad", "<i4")])
        it = torch.from_numpy(np.frombuffer(items.tobytes(), dtype=np.uint8).copy()).to(dev())
        m = torch.from_numpy(w.reshape(-1)).to(dev())
        wf = torch.zeros(n, dtype=tdt(dt), device=dev()); wd = torch.zeros(n, dtype=tdt(dt), device=dev())
        lib.call("rua_weight_prep", m.data_ptr(), wf.data_ptr(), wd.data_ptr(), it.data_ptr(), 1, n, dt, stream())
        dyd = to_dev(dy, dt)
        dx = torch.empty((N, H, W, Cin), dtype=tdt(dt), device=dev())
        d = L.ConvDesc()
        d.nseg = 1
        s = d.seg[0]
        s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = dyd.data_ptr(), wd.data_ptr(), Cout, H, W, 0, dil, taps
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cin, 1, dt
        d.y, d.out_stride, d.OH, d.OW = dx.data_ptr(), 1, H, W
        lib.call("rua_conv_fwd", C.byref(d), stream())
        torch.cuda.synchronize()
        x = torch.zeros((N, H, W, Cin), dtype=torch.float64, requires_grad=True)
        y = ref_conv_nhwc(x, rnd(dt, w).double(), None, dil, taps)
        y.backward(rnd(dt, dy).double())
        assert rel_err(dx.float().cpu().numpy(), x.grad.numpy()) < tol(dt)


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_bn_stats_apply_backward(dt):
    rng = np.random.default_rng(6)
    lib = L.lib()
    N, H, W, Cc = 2, 10, 6, 32
    M = N * H * W
    x = (rng.standard_normal((M, Cc)) * 2 + 0.5).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, Cc).astype(np.float32); beta = rng.standard_normal(Cc).astype(np.float32)
    xd = to_dev(x, dt)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    gd, bd = f(gamma), f(beta)
    mm, mv = f(np.zeros(Cc)), f(np.ones(Cc))
    stats = torch.zeros(2 * Cc, dtype=torch.float64, device=dev())
    coef = torch.zeros(7, Cc, dtype=torch.float32, device=dev())
    cp = [coef[i].data_ptr() for i in range(7)]
    R = 4
    stats = torch.zeros(R * 2 * Cc, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats", xd.data_ptr(), M, Cc, stats.data_ptr(), R, dt, stream())
    lib.call("rua_bn_finalize", stats.data_ptr(), R, float(M), float(M), gd.data_ptr(), bd.data_ptr(), mm.data_ptr(), mv.data_ptr(),
             0.99, 1e-3, 1, cp[0], cp[1], cp[2], cp[3], Cc, stream())
    out = torch.empty((M, Cc), dtype=tdt(dt), device=dev())
    sc, sh, ou = L.ptr_array([cp[0]]), L.ptr_array([cp[1]]), L.ptr_array([out.data_ptr()])
    lib.call("rua_bn_apply", xd.data_ptr(), 1, sc, sh, 1, ou, M, Cc, dt, stream())
    torch.cuda.synchronize()
    xr = rnd(dt, x).double().requires_grad_(True)
    mean = xr.mean(0); var = xr.var(0, unbiased=False)
    yr = torch.relu((xr - mean) / torch.sqrt(var + 1e-3) * torch.from_numpy(gamma).double() + torch.from_numpy(beta).double())
    assert rel_err(out.float().cpu().numpy(), yr.detach().numpy()) < tol(dt)
    assert np.allclose(mm.cpu().numpy(), 0.01 * mean.detach().numpy(), rtol=1e-4, atol=1e-6)
    assert np.allclose(mv.cpu().numpy(), 0.99 + 0.01 * var.detach().numpy() * M / (M - 1), rtol=1e-4)
    # backward
    g = rng.standard_normal((M, Cc)).astype(np.float32)
    gdv = to_dev(g, dt)
    yr.backward(rnd(dt, g).double())
    st2 = torch.zeros(2 * 2 * Cc, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats2", gdv.data_ptr(), xd.data_ptr(), cp[0], cp[1], 1, M, Cc, st2.data_ptr(), 2, dt, stream())
    dgam = torch.zeros(Cc, device=dev()); dbet = torch.zeros(Cc, device=dev())
    lib.call("rua_bn_bwd_finalize", st2.data_ptr(), 2, float(M), gd.data_ptr(), cp[2], cp[3], dgam.data_ptr(), dbet.data_ptr(),
             cp[4], cp[5], cp[6], Cc, stream())
    dx = torch.empty((M, Cc), dtype=tdt(dt), device=dev())
    ga, A, B_, C_, ms, mt = (L.ptr_array([gdv.data_ptr()]), L.ptr_array([cp[4]]), L.ptr_array([cp[5]]), L.ptr_array([cp[6]]),
                             L.ptr_array([cp[0]]), L.ptr_array([cp[1]]))
    lib.call("rua_bn_bwd_apply", 1, ga, A, B_, C_, ms, mt, 1, xd.data_ptr(), None, dx.data_ptr(), 0, M, Cc, dt, stream())
    torch.cuda.synchronize()
    assert rel_err(dx.float().cpu().numpy(), xr.grad.numpy()) < tol(dt)


@pytest.mark.parametrize("Cc", [32, 2048])
@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_bn_fused_forward_backward_two_branches(dt, Cc):
    """rua_bn_fwd / rua_bn_bwd: finalize folded into the launch, two branches sharing one input (the ResBlock case),
    skip gradient added, moving statistics and parameter gradients updated by block 0.  C = 2048: the backward's coefficient
    table of (5 nb + 2) C floats is 96 KB, beyond the default 64 KB of dynamic LDS."""
    rng = np.random.default_rng(16)
    lib = L.lib()
    M, R = 640, 4
    x = (rng.standard_normal((M, Cc)) * 1.5 + 0.3).astype(np.float32)
    xd = to_dev(x, dt)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    gam = [rng.uniform(0.5, 1.5, Cc).astype(np.float32) for _ in range(2)]
    bet = [rng.standard_normal(Cc).astype(np.float32) for _ in range(2)]
    gd, bd = [f(g) for g in gam], [f(b) for b in bet]
    mm, mv = [f(np.zeros(Cc)) for _ in range(2)], [f(np.ones(Cc)) for _ in range(2)]
    coef = [torch.zeros(4, Cc, device=dev()) for _ in range(2)]
    outs = [torch.empty((M, Cc), dtype=tdt(dt), device=dev()) for _ in range(2)]
    stats = torch.zeros(R * 2 * Cc, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats", xd.data_ptr(), M, Cc, stats.data_ptr(), R, dt, stream())
    d = L.BnFwdDesc()
    d.x, d.M, d.C, d.dtype, d.nb, d.relu, d.training, d.replicas = xd.data_ptr(), M, Cc, dt, 2, 1, 1, R
    d.stats, d.count, d.bessel_n, d.momentum, d.eps = stats.data_ptr(), float(M), float(M), 0.99, 1e-3
    for b in range(2):
        br = d.br[b]
        br.gamma, br.beta, br.moving_mean, br.moving_var = gd[b].data_ptr(), bd[b].data_ptr(), mm[b].data_ptr(), mv[b].data_ptr()
        br.scale, br.shift, br.mean, br.rstd = [coef[b][i].data_ptr() for i in range(4)]
        br.out = outs[b].data_ptr()
    lib.call("rua_bn_fwd", C.byref(d), stream())
    torch.cuda.synchronize()
    xr = rnd(dt, x).double().requires_grad_(True)
    mean, var = xr.mean(0), xr.var(0, unbiased=False)
    ys = [torch.relu((xr - mean) / torch.sqrt(var + 1e-3) * torch.from_numpy(gam[b]).double() + torch.from_numpy(bet[b]).double()) for b in range(2)]
    for b in range(2):
        assert rel_err(outs[b].float().cpu().numpy(), ys[b].detach().numpy()) < tol(dt)
        assert np.allclose(mm[b].cpu().numpy(), 0.01 * mean.detach().numpy(), rtol=1e-4, atol=1e-6)
        assert np.allclose(mv[b].cpu().numpy(), 0.99 + 0.01 * var.detach().numpy() * M / (M - 1), rtol=1e-4)
    g = [rng.standard_normal((M, Cc)).astype(np.float32) for _ in range(2)]
    skip = rng.standard_normal((M, Cc)).astype(np.float32)
    gdv, skd = [to_dev(a, dt) for a in g], to_dev(skip, dt)
    (ys[0] * rnd(dt, g[0]).double() + ys[1] * rnd(dt, g[1]).double()).sum().backward()
    st2 = [torch.zeros(2 * 2 * Cc, dtype=torch.float64, device=dev()) for _ in range(2)]
    for b in range(2):
        lib.call("rua_col_stats2", gdv[b].data_ptr(), xd.data_ptr(), coef[b][0].data_ptr(), coef[b][1].data_ptr(), 1, M, Cc, st2[b].data_ptr(), 2, dt, stream())
    dgam, dbet = [torch.zeros(Cc, device=dev()) for _ in range(2)], [torch.zeros(Cc, device=dev()) for _ in range(2)]
    dx = torch.empty((M, Cc), dtype=tdt(dt), device=dev())
    e = L.BnBwdDesc()
    e.x, e.dskip, e.dx, e.M, e.C, e.dtype, e.nb, e.masked, e.accumulate, e.count = xd.data_ptr(), skd.data_ptr(), dx.data_ptr(), M, Cc, dt, 2, 1, 0, float(M)
    for b in range(2):
        br = e.br[b]
        br.g, br.stats2, br.replicas, br.gamma = gdv[b].data_ptr(), st2[b].data_ptr(), 2, gd[b].data_ptr()
        br.scale, br.shift, br.mean, br.rstd = [coef[b][i].data_ptr() for i in range(4)]
        br.dgamma, br.dbeta = dgam[b].data_ptr(), dbet[b].data_ptr()
    lib.call("rua_bn_bwd", C.byref(e), stream())
    torch.cuda.synchronize()
    assert rel_err(dx.float().cpu().numpy(), xr.grad.numpy() + rnd(dt, skip).numpy()) < tol(dt)
    for b in range(2):
        xhat = ((xr - mean) / torch.sqrt(var + 1e-3)).detach().numpy()
        gm = rnd(dt, g[b]).numpy() * (ys[b].detach().numpy() > 0)
        assert rel_err(dgam[b].cpu().numpy(), (gm * xhat).sum(0)) < 5 * tol(dt) + 1e-4
        assert rel_err(dbet[b].cpu().numpy(), gm.sum(0)) < 5 * tol(dt) + 1e-4


@pytest.mark.parametrize("Cc", [6, 3, 2, 5])
def test_tanimoto_sums_match_numpy(Cc):
    """rua_tanimoto_sums: sums[n][c] = {sum p, sum (1-l), sum p*l, sum p^2+l^2, sum (1-p)(1-l), sum (1-p)^2+(1-l)^2}
    (multitasking_utils.py:38-85); C in {6, 3, 2} take the 16-byte-load kernel, 5 the scalar one."""
    rng = np.random.default_rng(31)
    B, HW = 3, 72 * 60
    p = rng.uniform(0, 1, (B, HW, Cc)).astype(np.float32)
    l = (rng.uniform(0, 1, (B, HW, Cc)) > 0.7).astype(np.float32)
    pd, ld = torch.from_numpy(p).to(dev()), torch.from_numpy(l).to(dev())
    sums = torch.zeros(B * Cc * 6, dtype=torch.float64, device=dev())
    L.lib().call("rua_tanimoto_sums", pd.data_ptr(), ld.data_ptr(), B, HW, Cc, sums.data_ptr(), stream())
    torch.cuda.synchronize()
    P, Y = p.astype(np.float64), l.astype(np.float64)
    want = np.stack([P.sum(1), (1 - Y).sum(1), (P * Y).sum(1), (P * P + Y * Y).sum(1), ((1 - P) * (1 - Y)).sum(1),
                     ((1 - P) ** 2 + (1 - Y) ** 2).sum(1)], axis=-1)            # [B][C][6]
    got = sums.cpu().numpy().reshape(B, Cc, 6)
    assert np.allclose(got, want, rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_pooling_family(dt):
    rng = np.random.default_rng(7)
    lib = L.lib()
    N, H, W, Cc = 2, 48, 48, 16
    x = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    xd = to_dev(x, dt)
    xr = rnd(dt, x)
    for k in (2, 4, 8, 3, 1):                                # 2 / 4 / 8: compile-time windows; 3, 1: the any-k kernel
        y = torch.empty((N, H // k, W // k, Cc), dtype=tdt(dt), device=dev())
        idx = torch.empty(y.numel(), dtype=torch.uint8, device=dev())
        lib.call("rua_maxpool_fwd", xd.data_ptr(), y.data_ptr(), idx.data_ptr(), N, H, W, Cc, k, dt, stream())
        xt = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
        yt = F.max_pool2d(xt, k, k)
        torch.cuda.synchronize()
        assert np.array_equal(y.float().cpu().numpy(), yt.detach().permute(0, 2, 3, 1).numpy())
        g = rng.standard_normal((N, H // k, W // k, Cc)).astype(np.float32)
        gd = to_dev(g, dt)
        dx = torch.empty((N, H, W, Cc), dtype=tdt(dt), device=dev())
        lib.call("rua_maxpool_bwd", gd.data_ptr(), idx.data_ptr(), dx.data_ptr(), 0, N, H, W, Cc, k, dt, stream())
        yt.backward(rnd(dt, g).permute(0, 3, 1, 2))
        torch.cuda.synchronize()
        assert np.array_equal(dx.float().cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy())
        sp = torch.empty((N, H // k, W // k, Cc), dtype=tdt(dt), device=dev())
        lib.call("rua_sumpool", xd.data_ptr(), sp.data_ptr(), N, H, W, Cc, k, dt, stream())
        torch.cuda.synchronize()
        exp = F.avg_pool2d(xr.permute(0, 3, 1, 2).double(), k, k).permute(0, 2, 3, 1).numpy() * k * k
        assert rel_err(sp.float().cpu().numpy(), exp) < tol(dt)


def test_stem_and_head_kernels():
    rng = np.random.default_rng(8)
    lib = L.lib()
    for dt in (L.RUA_F32, L.RUA_BF16):
        M, Cin, Cout = 500, 6, 32
        x = rng.standard_normal((M, Cin)).astype(np.float32)
        w = rng.standard_normal((Cout, Cin)).astype(np.float32); b = rng.standard_normal(Cout).astype(np.float32)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
        xd, wd, bd = f(x), f(w), f(b)
        y = torch.empty((M, Cout), dtype=tdt(dt), device=dev())
        lib.call("rua_stem_fwd", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, Cin, Cout, dt, stream())
        torch.cuda.synchronize()
        assert rel_err(y.float().cpu().numpy(), x @ w.T + b) < tol(dt)
        # rua_stem_fwd_stats: the same output bit for bit + the statistics rua_col_stats takes from it (of the values as stored), at a size with
        # several blocks, a ragged tail and more blocks than replicas
        for Ms, R in ((M, 4), (70001, 8)):
            xs = f(rng.standard_normal((Ms, Cin)))
            y0 = torch.empty((Ms, Cout), dtype=tdt(dt), device=dev()); y1 = torch.empty_like(y0)
            st = torch.zeros(R * 2 * Cout, dtype=torch.float64, device=dev()); ref = torch.zeros_like(st)
            lib.call("rua_stem_fwd", xs.data_ptr(), wd.data_ptr(), bd.data_ptr(), y0.data_ptr(), Ms, Cin, Cout, dt, stream())
            lib.call("rua_stem_fwd_stats", xs.data_ptr(), wd.data_ptr(), bd.data_ptr(), y1.data_ptr(), Ms, Cin, Cout, dt, st.data_ptr(), R, stream())
            lib.call("rua_col_stats", y0.data_ptr(), Ms, Cout, ref.data_ptr(), R, dt, stream())
            torch.cuda.synchronize()
            assert torch.equal(y0, y1)
            got, want = st.view(R, 2 * Cout).sum(0).cpu().numpy(), ref.view(R, 2 * Cout).sum(0).cpu().numpy()
            exact = np.concatenate([y0.double().sum(0).cpu().numpy(), (y0.double() ** 2).sum(0).cpu().numpy()])
            assert np.abs(got - exact).max() <= 2e-6 * Ms + np.abs(want - exact).max(), (np.abs(got - exact).max(), np.abs(want - exact).max())
        dy = rng.standard_normal((M, Cout)).astype(np.float32)
        dyd = to_dev(dy, dt)
        dw = torch.zeros((Cout, Cin), device=dev()); db = torch.zeros(Cout, device=dev())
        lib.call("rua_stem_bwd", xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), M, Cin, Cout, dt, stream())
        torch.cuda.synchronize()
        dyr = rnd(dt, dy).numpy()
        assert rel_err(dw.cpu().numpy(), dyr.T @ x) < 1e-4
        assert rel_err(db.cpu().numpy(), dyr.sum(0)) < 1e-4
        if dt == L.RUA_BF16:
            # the stem's weight gradient on the matrix pipe: rua_stem_fwd_pack (same y and statistics as rua_stem_fwd_stats + the input as bf16 hi | lo | 1),
            # the 1x1 weight gradient of dy against it, rua_stem_bwd_fold - against the exact sums (the hi + lo split carries ~16 mantissa bits of x)
            for Ms, Ci3 in ((4096, 6), (70016, 3), (8192, 7)):
                xs_h = rng.standard_normal((Ms, Ci3)).astype(np.float32)
                w3 = f(rng.standard_normal((Cout, Ci3)))
                xs = f(xs_h)
                y0 = torch.empty((Ms, Cout), dtype=tdt(dt), device=dev()); y1 = torch.empty_like(y0)
                R = 8
                st0 = torch.zeros(R * 2 * Cout, dtype=torch.float64, device=dev()); st1 = torch.zeros_like(st0)
                xp = torch.zeros((Ms, 16), dtype=torch.bfloat16, device=dev())
                lib.call("rua_stem_fwd_stats", xs.data_ptr(), w3.data_ptr(), bd.data_ptr(), y0.data_ptr(), Ms, Ci3, Cout, dt, st0.data_ptr(), R, stream())
                lib.call("rua_stem_fwd_pack", xs.data_ptr(), w3.data_ptr(), bd.data_ptr(), y1.data_ptr(), Ms, Ci3, Cout, dt, st1.data_ptr(), R, xp.data_ptr(), stream())
                torch.cuda.synchronize()
                assert torch.equal(y0, y1) and torch.equal(st0, st1)
                xpf = xp.float().cpu().numpy()
                assert np.abs(xpf[:, :Ci3] + xpf[:, 8:8 + Ci3] - xs_h).max() < 1e-4 and (xpf[:, 15] == 1).all() and (xpf[:, Ci3:8] == 0).all()
                dys = rng.standard_normal((Ms, Cout)).astype(np.float32)
                dysd = to_dev(dys, dt)
                tmp = torch.zeros((Cout, 16), device=dev())
                ws = torch.zeros(16 << 20, device=dev())
                d = L.WgradDesc()
                d.a, d.C, d.Hs, d.Ws = xp.data_ptr(), 16, 1, Ms
                d.dy, d.Cout, d.H, d.W = dysd.data_ptr(), Cout, 1, Ms
                d.N, d.stride, d.dil, d.taps, d.dtype = 1, 1, 1, 1, dt
                d.dw, d.workspace, d.workspace_bytes = tmp.data_ptr(), ws.data_ptr(), ws.numel() * 4
                assert lib.raw("rua_wgrad_kind")(C.byref(d)) == 3                  # wgrad_pw
                dw3 = torch.zeros((Cout, Ci3), device=dev()); db3 = torch.zeros(Cout, device=dev())
                for _ in range(2):                                                # twice: the fold leaves tmp zero, the gradients accumulate
                    lib.call("rua_conv_wgrad", C.byref(d), stream())
                    lib.call("rua_stem_bwd_fold", tmp.data_ptr(), dw3.data_ptr(), db3.data_ptr(), Ci3, Cout, stream())
                torch.cuda.synchronize()
                dyr3 = rnd(dt, dys).double().numpy()
                assert rel_err(dw3.cpu().numpy(), 2 * dyr3.T @ xs_h.astype(np.float64)) < 1e-4
                assert rel_err(db3.cpu().numpy(), 2 * dyr3.sum(0)) < 1e-4
                assert (tmp == 0).all()
        # head
        Ci, Co = 32, 6
        hx = rng.standard_normal((M, Ci)).astype(np.float32)
        hw = (rng.standard_normal((Co, Ci)) / 4).astype(np.float32); hb = rng.standard_normal(Co).astype(np.float32)
        hxd, hwd, hbd = to_dev(hx, dt), f(hw), f(hb)
        z = torch.empty((M, Co), device=dev()); p = torch.empty((M, Co), device=dev())
        for act in (L.ACT_SOFTMAX, L.ACT_SIGMOID):
            lib.call("rua_head_fwd", hxd.data_ptr(), hwd.data_ptr(), hbd.data_ptr(), z.data_ptr(), p.data_ptr(), M, Ci, Co, act, dt, stream())
            torch.cuda.synchronize()
            zr = rnd(dt, hx).double().numpy() @ hw.T.astype(np.float64) + hb
            assert rel_err(z.cpu().numpy(), zr) < 1e-5
            pr = torch.softmax(torch.from_numpy(zr), 1).numpy() if act == L.ACT_SOFTMAX else 1 / (1 + np.exp(-zr))
            assert rel_err(p.cpu().numpy(), pr) < 1e-5
        dz = rng.standard_normal((M, Co)).astype(np.float32)
        dzd = f(dz)
        dx = torch.empty((M, Ci), dtype=tdt(dt), device=dev())
        dwh = torch.zeros((Co, Ci), device=dev()); dbh = torch.zeros(Co, device=dev())
        scr = torch.empty(1024 * (Co * Ci + Co), device=dev())
        for scratch, mask_dx in ((None, 0), (scr, 0), (scr, 1)):   # fp32-atomic path, deterministic partial path, fused ReLU mask
            dwh.zero_(); dbh.zero_()
            lib.call("rua_head_bwd", hxd.data_ptr(), dzd.data_ptr(), hwd.data_ptr(), dx.data_ptr(), 0, dwh.data_ptr(), dbh.data_ptr(),
                     None if scratch is None else scratch.data_ptr(), 0 if scratch is None else scratch.numel() * 4, M, Ci, Co, dt,
                     mask_dx, stream())
            torch.cuda.synchronize()
            assert rel_err(dx.float().cpu().numpy(), (dz @ hw) * ((rnd(dt, hx).numpy() > 0) if mask_dx else 1.0)) < tol(dt)
            assert rel_err(dwh.cpu().numpy(), dz.T @ rnd(dt, hx).numpy()) < 1e-4
            assert rel_err(dbh.cpu().numpy(), dz.sum(0)) < 1e-4


def test_losses_against_oracle():
    """Tanimoto dual / weighted CE / CE / BCE / MSE values and logits-gradients vs torch autograd of the oracle."""
    from oracle import resuneta_ref as ref
    rng = np.random.default_rng(9)
    lib = L.lib()
    B, H, W, Cc = 3, 12, 10, 5
    HW = H * W
    M = B * HW
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    ids = rng.integers(0, Cc - 1, size=(B, H, W))            # last class absent everywhere -> inf weight path
    y1h = np.eye(Cc, dtype=np.float32)[ids]
    ysoft = rng.uniform(0, 1, (B, H, W, Cc)).astype(np.float32)
    cw = np.array([4.3, 2.9, 3.9, 5.6, 37.0], np.float32)
    cases = [(L.LOSS_TANIMOTO, L.ACT_SOFTMAX, y1h), (L.LOSS_TANIMOTO, L.ACT_SIGMOID, ysoft), (L.LOSS_WCE, L.ACT_SOFTMAX, y1h),
             (L.LOSS_CE_LOGITS, L.ACT_SOFTMAX, y1h), (L.LOSS_BCE_LOGITS, L.ACT_SIGMOID, y1h), (L.LOSS_MSE, L.ACT_SOFTMAX, ysoft),
             (L.LOSS_MSE, L.ACT_SIGMOID, ysoft)]
    for kind, act, y in cases:
        z = (rng.standard_normal((B, H, W, Cc)) * 2).astype(np.float32)
        zt = torch.from_numpy(z).permute(0, 3, 1, 2).double().requires_grad_(True)
        yt = torch.from_numpy(y).permute(0, 3, 1, 2).double()
        pt = torch.softmax(zt, 1) if act == L.ACT_SOFTMAX else torch.sigmoid(zt)
        wgt = 0.7
        if kind == L.LOSS_TANIMOTO:
            lt = ref.tanimoto_dual_loss(yt, pt).mean()
        elif kind == L.LOSS_WCE:
            lt = (-(yt * torch.log(torch.clamp(pt / pt.sum(1, keepdim=True), 1e-7, 1 - 1e-7)) * torch.from_numpy(cw).double()[None, :, None, None]).sum(1)).mean()
        elif kind == L.LOSS_CE_LOGITS:
            lt = ref.categorical_ce_logits(yt, zt).mean()
        elif kind == L.LOSS_BCE_LOGITS:
            lt = ref.binary_ce_logits(yt, zt).mean()
        else:
            lt = ref.mse(yt, pt).mean()
        (wgt * lt).backward()
        gz = zt.grad.permute(0, 2, 3, 1).numpy()
        pd, yd, zd = f(pt.detach().permute(0, 2, 3, 1).numpy()), f(y), f(z)
        scal = torch.zeros(16, dtype=torch.float64, device=dev())
        dz = torch.empty((B, H, W, Cc), device=dev())
        cwd = f(cw)
        coef = torch.zeros(B * Cc * 3, device=dev())
        if kind == L.LOSS_TANIMOTO:
            sums = torch.zeros(B * Cc * 6, dtype=torch.float64, device=dev())
            lib.call("rua_tanimoto_sums", pd.data_ptr(), yd.data_ptr(), B, HW, Cc, sums.data_ptr(), stream())
            per = torch.zeros(B, device=dev())
            lib.call("rua_tanimoto_finalize", sums.data_ptr(), B, HW, Cc, wgt / B, scal.data_ptr(), coef.data_ptr(), per.data_ptr(), stream())
            gs, norm = wgt / B, 1.0
            torch.cuda.synchronize()
            assert np.allclose(per.cpu().numpy(), ref.tanimoto_dual_loss(yt, pt).detach().numpy(), rtol=1e-5, atol=1e-6)
        else:
            lib.call("rua_pixel_loss", kind, pd.data_ptr(), zd.data_ptr(), yd.data_ptr(), cwd.data_ptr(), M, Cc, scal.data_ptr(), None, stream())
            gs, norm = wgt / M, 1.0 / M
        lib.call("rua_head_dz", kind, act, pd.data_ptr(), yd.data_ptr(), coef.data_ptr(), cwd.data_ptr(), gs, B, HW, Cc, dz.data_ptr(), stream())
        torch.cuda.synchronize()
        assert abs(float(scal[0]) * norm - float(lt.detach())) < 1e-5 * max(1.0, abs(float(lt))), (kind, act)
        assert rel_err(dz.cpu().numpy(), gz) < 2e-4, (kind, act)


def test_metrics_and_optimizers():
    from oracle import naive_ops as nv
    rng = np.random.default_rng(10)
    lib = L.lib()
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    for M, Cc in ((777, 6), (70000, 6), (4096, 2), (4096, 5)):   # odd M / C = 5: scalar kernel; even M with C in {6, 2}: 16-byte loads
        p = rng.uniform(0, 1, (M, Cc)).astype(np.float32); p /= p.sum(1, keepdims=True)
        y = np.eye(Cc, dtype=np.float32)[rng.integers(0, Cc, M)]
        out = torch.zeros(5, dtype=torch.float64, device=dev())
        pd, yd = f(p), f(y)
        lib.call("rua_seg_metrics", pd.data_ptr(), yd.data_ptr(), M, Cc, out.data_ptr(), stream())
        torch.cuda.synchronize()
        t, q = y > 0.5, p > 0.5
        exp = [(p.argmax(1) == y.argmax(1)).sum(), (t & q).sum(), (~t & q).sum(), (~t & ~q).sum(), (t & ~q).sum()]
        assert np.array_equal(out.cpu().numpy(), np.array(exp, np.float64)), (M, Cc)
    n = 1000
    th = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
    thd, gd, md, vd = f(th), f(g), f(np.zeros(n)), f(np.zeros(n))
    e_th, e_m, e_v = th.astype(np.float64), np.zeros(n), np.zeros(n)
    for step in (1, 2):
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        gd.copy_(f(g))
        lib.call("rua_adam_step", thd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, lr_t, None, 0.9, 0.999, 1e-7, 0.5, 1, stream())
        e_th, e_m, e_v = nv.adam_step(e_th, 0.5 * g.astype(np.float64), e_m, e_v, step, 1e-3)
    torch.cuda.synchronize()
    assert np.allclose(thd.cpu().numpy(), e_th, rtol=1e-5, atol=1e-6)
    assert float(gd.abs().max()) == 0.0
    thd, gd, vd = f(th), f(g), f(np.zeros(n))
    lib.call("rua_sgd_step", thd.data_ptr(), gd.data_ptr(), vd.data_ptr(), n, 0.1, None, 0.8, 1.0, 0, stream())
    torch.cuda.synchronize()
    e_th, _ = nv.sgd_step(th.astype(np.float64), g.astype(np.float64), np.zeros(n), 0.1)
    assert np.allclose(thd.cpu().numpy(), e_th, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("H,W,dil,Cs", [(256, 256, 1, 32), (256, 256, 3, 32), (256, 256, 15, 32), (256, 256, 31, 32), (272, 248, 3, 32),
                                        (248, 272, 15, 32), (256, 256, 1, 64), (256, 256, 3, 64), (264, 256, 15, 64), (256, 256, 31, 64)])
@pytest.mark.parametrize("mode", ["residual_stats", "mask_accumulate_stats2"])
def test_conv_halo_lattice_tiles(H, W, dil, Cs, mode):
    """The kernel of the two top levels (C = Cout in {32, 64}, bf16): input + halo resident in LDS, dilation by lattice decomposition.
    Every dilation of the reference's ResBlocks, ragged maps (lattice tiles that overhang the image, residue classes of
    unequal size), and both epilogue families (forward: bias + residual + sum/sum^2; data gradient: ReLU mask from
    aux*scale+shift, accumulate into y, sum g / sum g*aux)."""
    dt = L.RUA_BF16
    rng = np.random.default_rng(H + dil)
    lib = L.lib()
    N, Cout = 1, Cs
    x = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
    w = (rng.standard_normal((9, Cout, Cs)) / np.sqrt(9 * Cs)).astype(np.float32)
    aux = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    y0 = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    sc = (0.5 + rng.random(Cout)).astype(np.float32); sh = (0.3 * rng.standard_normal(Cout)).astype(np.float32)
    xd, wd, ad, bd = to_dev(x, dt), to_dev(w, dt), to_dev(aux, dt), torch.from_numpy(bias).to(dev())
    scd, shd = torch.from_numpy(sc).to(dev()), torch.from_numpy(sh).to(dev())
    y = to_dev(y0, dt)
    R = 8
    stats = torch.zeros(R * 2 * Cout, dtype=torch.float64, device=dev())
    d = L.ConvDesc()
    d.nseg = 1
    s = d.seg[0]
    s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd.data_ptr(), wd.data_ptr(), Cs, H, W, 0, dil, 9
    d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cout, 1, dt
    d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
    d.stats, d.stats_replicas = stats.data_ptr(), R
    conv = ref_conv_nhwc(rnd(dt, x), rnd(dt, w), None, dil, 9).numpy()
    a = rnd(dt, aux).double().numpy()
    if mode == "residual_stats":
        d.bias, d.aux, d.aux_mode, d.stats_mode = bd.data_ptr(), ad.data_ptr(), 1, 1
        exp = conv + bias.astype(np.float64) + a
        s2 = (exp ** 2).sum(axis=(0, 1, 2))
    else:
        d.aux, d.aux_mode, d.mscale, d.mshift, d.accumulate, d.stats_mode = ad.data_ptr(), 2, scd.data_ptr(), shd.data_ptr(), 1, 2
        exp = (conv + rnd(dt, y0).double().numpy()) * ((a * sc + sh) > 0)
        s2 = (exp * a).sum(axis=(0, 1, 2))
    # (C = 64: conv_halo serves every dilation on request - rua_set_tuning("halo64_maxd", 31) - and none by default)
    # C = 32 with a row that splits into 128 / 256-pixel strips and one epilogue stream goes to conv_strip (its own test below)
    strip = Cs == 32 and W % 128 == 0 and mode == "residual_stats"
    if Cs == 32:
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == (5 if strip else 3)
    if strip:
        lib.set_tuning(conv_strip=0)                             # this test is about conv_halo: route the shape back to it
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 3
    if Cs == 64:                                                 # conv_halo<64> is off by default (the grouped conv_igemm grid takes d = 1 too)
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) != 3
        lib.set_tuning(halo64_maxd=31)
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 3
    try:
        lib.call("rua_conv_fwd", C.byref(d), stream())
    finally:
        lib.set_tuning(conv_strip=1, halo64_maxd=0)
    torch.cuda.synchronize()
    got = y.float().cpu().numpy()
    assert rel_err(got, exp) < tol(dt)
    st = stats.cpu().numpy().reshape(R, 2 * Cout).sum(0)
    assert rel_err(st[:Cout], exp.sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4
    assert rel_err(st[Cout:], s2) < 5 * tol(dt) + 1e-4


@pytest.mark.parametrize("kern", ["img2", "img"])
@pytest.mark.parametrize("mode", ["residual_stats", "mask_acc_stats2", "plain"])
@pytest.mark.parametrize("case", [(8, 8, 1024, 1024), (8, 16, 512, 512), (8, 16, 256, 512), (16, 8, 512, 256), (8, 16, 512, 256), (4, 8, 2048, 128)])
def test_conv_whole_image_kernel(case, mode, kern):
    """conv_img2 (round 5, the default at the 8 x 8 / 16 x 16 levels: whole images resident, 64-channel output slices, 128-channel input chunks as K slices, weights of a
    kernel row as coalesced swizzled rows -> registers, conv_splitk_finish for the epilogue; the last case: two chunks per block) and conv_img (round 4, opt-in, tuning key conv_img): the 3x3 convolutions of the 8 x 8 / 16 x 16 levels with a whole image resident in LDS (one image x 32 output channels per block, weights streamed from
    L2 into registers; 16 x 16 x 512 and wider: two channel halves as K slices + the finisher).  Both epilogue families and the plain form, against PyTorch."""
    N, H, Cs, Cout = case
    dt = L.RUA_BF16
    rng = np.random.default_rng(N + Cs)
    lib = L.lib()
    x = rng.standard_normal((N, H, H, Cs)).astype(np.float32)
    w = (rng.standard_normal((9, Cout, Cs)) / np.sqrt(9 * Cs)).astype(np.float32)
    aux = rng.standard_normal((N, H, H, Cout)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    y0 = rng.standard_normal((N, H, H, Cout)).astype(np.float32)
    sc = (0.5 + rng.random(Cout)).astype(np.float32); sh = (0.3 * rng.standard_normal(Cout)).astype(np.float32)
    xd, wd, ad, bd = to_dev(x, dt), to_dev(w, dt), to_dev(aux, dt), torch.from_numpy(bias).to(dev())
    scd, shd = torch.from_numpy(sc).to(dev()), torch.from_numpy(sh).to(dev())
    y = to_dev(y0, dt)
    R = 4
    stats = torch.zeros(R * 2 * Cout, dtype=torch.float64, device=dev())
    if kern == "img" and case == (4, 8, 2048, 128):
        pytest.skip("not a conv_img shape")
    ws = torch.zeros(((8 if kern == "img2" else 2) * N * H * H * Cout * 4 + 8192) // 4, dtype=torch.float32, device=dev())
    d = L.ConvDesc()
    d.nseg = 1
    s = d.seg[0]
    s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd.data_ptr(), wd.data_ptr(), Cs, H, H, 0, 1, 9
    d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, H, Cout, 1, dt
    d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, H
    d.stats, d.stats_replicas = stats.data_ptr(), R
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    conv = ref_conv_nhwc(rnd(dt, x), rnd(dt, w), None, 1, 9).numpy()
    a = rnd(dt, aux).double().numpy()
    s2 = None
    if mode == "residual_stats":
        d.bias, d.aux, d.aux_mode, d.stats_mode = bd.data_ptr(), ad.data_ptr(), 1, 1
        exp = conv + bias.astype(np.float64) + a
        s2 = (exp ** 2).sum(axis=(0, 1, 2))
    elif mode == "mask_acc_stats2":
        d.aux, d.aux_mode, d.mscale, d.mshift, d.accumulate, d.stats_mode = ad.data_ptr(), 2, scd.data_ptr(), shd.data_ptr(), 1, 2
        exp = (conv + rnd(dt, y0).double().numpy()) * ((a * sc + sh) > 0)
        s2 = (exp * a).sum(axis=(0, 1, 2))
    else:
        d.stats, d.stats_mode = None, 0
        exp = conv
    assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 8     # the default: conv_img2
    lib.set_tuning(conv_img2=0)
    try:
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 2     # conv_dmap + split K (conv_img: level with it in the step, opt-in by tuning key conv_img)
        lib.set_tuning(conv_img=1 if kern == "img" else 0, conv_img2=1 if kern == "img2" else 0)
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == (7 if kern == "img" else 8)
        lib.call("rua_conv_fwd", C.byref(d), stream())
        torch.cuda.synchronize()
        if kern == "img":
            assert lib.raw("rua_conv_last_ksplit")() == (1 if H * H * Cs * 2 <= 131072 else 2)
        else:
            assert lib.raw("rua_conv_last_ksplit")() >= 2
    finally:
        lib.set_tuning(conv_img=0, conv_img2=0)
    got = y.float().cpu().numpy()
    assert rel_err(got, exp) < tol(dt)
    if s2 is not None:
        st = stats.cpu().numpy().reshape(R, 2 * Cout).sum(0)
        assert rel_err(st[:Cout], exp.sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4
        assert rel_err(st[Cout:], s2) < 5 * tol(dt) + 1e-4
    y2 = to_dev(y0, dt)                                          # the same call through conv_dmap + split K: the two paths agree to storage rounding
    d.y = y2.data_ptr()
    if d.stats_mode:
        stats.zero_()
    try:
        lib.call("rua_conv_fwd", C.byref(d), stream())
        torch.cuda.synchronize()
    finally:
        lib.set_tuning(conv_img2=1)
    assert rel_err(y2.float().cpu().numpy(), got) < tol(dt)


STRIP_CASES = [
    # N, H, W, dil  (C = Cout = 32; W % 256 == 0: 8-wave blocks; W % 128 == 0: 4-wave blocks; ragged H: residue classes of unequal size)
    (1, 256, 256, 1), (1, 256, 256, 3), (1, 256, 256, 15), (1, 256, 256, 31), (4, 128, 128, 3), (4, 128, 128, 31),
    (1, 200, 512, 15), (2, 190, 384, 1), (1, 257, 256, 3),
]


@pytest.mark.parametrize("mode", ["bn_plain_stats", "bn_residual", "mask_stats2", "bn_accumulate_relu", "plain"])
@pytest.mark.parametrize("N,H,W,dil", STRIP_CASES)
def test_conv_strip_streaming_kernel(N, H, W, dil, mode):
    """conv_strip (C = Cout = 32, bf16): rows streamed through an LDS ring by LDS-DMA with counted waits, BatchNorm + ReLU of
    the input applied as a row lands (zero padding must stay zero: model2.py:17-24 pads AFTER the activation), transposed
    MFMA product with the epilogue in registers, the epilogue's per-pixel tensor (residual / ReLU-mask source / old output)
    through a second ring.  Every dilation of the reference's ResBlocks, one- and two-strip rows, ragged heights, chains cut
    into segments, and every epilogue family, against a float64 convolution of the bf16-rounded operands."""
    dt = L.RUA_BF16
    rng = np.random.default_rng(1000 * dil + H + len(mode))
    lib = L.lib()
    Cs = Cout = 32
    x = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
    w = (rng.standard_normal((9, Cout, Cs)) / np.sqrt(9 * Cs)).astype(np.float32)
    aux = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    y0 = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    isc = (0.5 + rng.random(Cs)).astype(np.float32); ish = (0.4 * rng.standard_normal(Cs)).astype(np.float32)
    msc = (0.5 + rng.random(Cout)).astype(np.float32); msh = (0.3 * rng.standard_normal(Cout)).astype(np.float32)
    xd, wd, ad, bd = to_dev(x, dt), to_dev(w, dt), to_dev(aux, dt), torch.from_numpy(bias).to(dev())
    iscd, ishd, mscd, mshd = (torch.from_numpy(a).to(dev()) for a in (isc, ish, msc, msh))
    y = to_dev(y0, dt)
    R = 8
    stats = torch.zeros(R * 2 * Cout, dtype=torch.float64, device=dev())
    d = L.ConvDesc()
    d.nseg = 1
    s = d.seg[0]
    s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd.data_ptr(), wd.data_ptr(), Cs, H, W, 0, dil, 9
    d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cout, 1, dt
    d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
    d.stats, d.stats_replicas = stats.data_ptr(), R
    xin = rnd(dt, x)
    if mode.startswith("bn"):
        d.in_scale, d.in_shift, d.in_relu = iscd.data_ptr(), ishd.data_ptr(), 1
        xin = torch.relu(xin * torch.from_numpy(isc) + torch.from_numpy(ish)).to(torch.bfloat16).float()       # what lands in LDS
    conv = ref_conv_nhwc(xin, rnd(dt, w), None, dil, 9).numpy()
    a = rnd(dt, aux).double().numpy()
    s2 = None
    if mode == "bn_plain_stats":
        d.bias, d.stats_mode = bd.data_ptr(), 1
        exp = conv + bias.astype(np.float64)
        s2 = (exp ** 2).sum(axis=(0, 1, 2))
    elif mode == "bn_residual":
        d.bias, d.aux, d.aux_mode, d.stats_mode = bd.data_ptr(), ad.data_ptr(), 1, 1
        exp = conv + bias.astype(np.float64) + a
        s2 = (exp ** 2).sum(axis=(0, 1, 2))
    elif mode == "mask_stats2":
        d.aux, d.aux_mode, d.mscale, d.mshift, d.stats_mode = ad.data_ptr(), 2, mscd.data_ptr(), mshd.data_ptr(), 2
        exp = conv * ((a * msc + msh) > 0)
        s2 = (exp * a).sum(axis=(0, 1, 2))
    elif mode == "bn_accumulate_relu":
        d.bias, d.accumulate, d.out_relu = bd.data_ptr(), 1, 1
        exp = np.maximum(conv + bias.astype(np.float64) + rnd(dt, y0).double().numpy(), 0.0)
    else:
        exp = conv
    assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 5 and lib.raw("rua_conv_fused_input_ok")(C.byref(d)) == 1
    lib.call("rua_conv_fwd", C.byref(d), stream())
    torch.cuda.synchronize()
    got = y.float().cpu().numpy()
    assert rel_err(got, exp) < tol(dt), rel_err(got, exp)
    if s2 is not None:
        st = stats.cpu().numpy().reshape(R, 2 * Cout).sum(0)
        assert rel_err(st[:Cout], exp.sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4
        assert rel_err(st[Cout:], s2) < 5 * tol(dt) + 1e-4
    # a shape no fused-input kernel serves must refuse the request instead of ignoring it
    d.Cout = 64
    assert lib.raw("rua_conv_fused_input_ok")(C.byref(d)) == 0


@pytest.mark.parametrize("Cs,H,W,dil", [(32, 256, 256, 1), (32, 256, 256, 31), (64, 128, 128, 3), (32, 136, 192, 15), (64, 128, 128, 31), (64, 72, 128, 1)])
def test_wgrad_all_taps_normalise_on_load(Cs, H, W, dil):
    """rua_conv_wgrad with in_scale / in_shift / in_relu (all-taps kernel of the two top levels): the conv input is
    BatchNorm'ed + ReLU'ed as it enters LDS, so the weight gradient equals the one taken against the materialised
    activation relu(scale * a + shift) - with the zero padding left at zero."""
    dt = L.RUA_BF16
    rng = np.random.default_rng(Cs + dil)
    lib = L.lib()
    N = 2
    a = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
    sc = (0.5 + rng.random(Cs)).astype(np.float32); sh = (0.4 * rng.standard_normal(Cs)).astype(np.float32)
    ad, dyd = to_dev(a, dt), to_dev(dy, dt)
    scd, shd = torch.from_numpy(sc).to(dev()), torch.from_numpy(sh).to(dev())
    act = torch.relu(rnd(dt, a) * torch.from_numpy(sc) + torch.from_numpy(sh)).to(torch.bfloat16).float()
    actd = act.to(dev()).to(torch.bfloat16).contiguous()
    d = L.WgradDesc()
    d.C, d.Hs, d.Ws, d.dy, d.Cout, d.H, d.W = Cs, H, W, dyd.data_ptr(), Cs, H, W
    d.N, d.stride, d.dil, d.taps, d.dtype = N, 1, dil, 9, dt
    nb = lib.raw("rua_wgrad_workspace_bytes")(C.byref(d))
    ws = torch.zeros(max(nb // 4, 16) + (264 << 8), dtype=torch.float32, device=dev())
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    assert lib.raw("rua_wgrad_kind")(C.byref(d)) == 1
    outs = []
    for fused in (True, False):
        dw = torch.zeros(9 * Cs * Cs, dtype=torch.float32, device=dev())
        d.dw = dw.data_ptr()
        if fused:
            d.a, d.in_scale, d.in_shift, d.in_relu = ad.data_ptr(), scd.data_ptr(), shd.data_ptr(), 1
        else:
            d.a, d.in_scale, d.in_shift, d.in_relu = actd.data_ptr(), None, None, 0
        lib.call("rua_conv_wgrad", C.byref(d), stream())
        torch.cuda.synchronize()
        outs.append(dw.cpu().numpy())
    # the same operands up to the rounding of one fused multiply-add (kernel) vs multiply then add (host) before the bf16 cast
    assert rel_err(outs[0], outs[1]) < 2e-3
    at = act.permute(0, 3, 1, 2).double()
    g = torch.nn.grad.conv2d_weight(at, (Cs, Cs, 3, 3), rnd(dt, dy).permute(0, 3, 1, 2).double(), padding=dil, dilation=dil)
    exp = g.permute(2, 3, 0, 1).reshape(9, Cs, Cs).numpy()
    assert rel_err(outs[0].reshape(9, Cs, Cs), exp) < tol(dt)


def test_wgrad_deferred_batched_reduction_is_bit_identical():
    """rua_conv_wgrad(defer=1) leaves its partial sums (all-taps block partials / K-slice slabs) in a private workspace;
    rua_wgrad_reduce_batch adds the partials of SEVERAL weight gradients into their dW in one launch - same arithmetic, same
    order as the per-call reductions, so bit-identical to them."""
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(77)
    cases = [(2, 64, 64, 32, 32, 1, 3, 9), (1, 128, 64, 64, 64, 1, 15, 9), (5, 32, 32, 256, 256, 1, 3, 9), (2, 16, 16, 128, 256, 1, 1, 1),
             (2, 64, 64, 32, 8, 1, 1, 1), (4, 32, 32, 256, 256, 1, 3, 9)]    # all-taps x2, wgrad_dmap (an odd number of images: not wgrad_rowsx), generic with K split, wgrad_pw (nothing pending), wgrad_rowsx<1>
    descs, keep, imm = [], [], []
    for (N, H, W, Cs, Cout, stride, dil, taps) in cases:
        a = to_dev(rng.standard_normal((N, H, W, Cs)).astype(np.float32), dt)
        dy = to_dev(rng.standard_normal((N, H, W, Cout)).astype(np.float32), dt)
        base = torch.from_numpy(rng.standard_normal((taps, Cout, Cs)).astype(np.float32)).to(dev())
        d = L.WgradDesc()
        d.a, d.C, d.Hs, d.Ws, d.dy, d.Cout, d.H, d.W = a.data_ptr(), Cs, H, W, dy.data_ptr(), Cout, H, W
        d.N, d.stride, d.dil, d.taps, d.dtype = N, stride, dil, taps, dt
        ws = torch.zeros(lib.raw("rua_wgrad_workspace_bytes")(C.byref(d)) // 4 + 16, dtype=torch.float32, device=dev())
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        dw = base.clone()
        d.dw = dw.data_ptr()
        lib.call("rua_conv_wgrad", C.byref(d), stream())                      # immediate
        torch.cuda.synchronize()
        imm.append(dw.cpu().numpy().copy())
        dw.copy_(base)
        descs.append(d); keep += [a, dy, ws, dw, base]
    recs, kinds = [], []
    for d in descs:                                                           # deferred
        d.defer = 1
        r = L.WgradPending()
        lib.call("rua_wgrad_plan", C.byref(d), C.byref(r))
        kinds.append(r.kind)
        lib.call("rua_conv_wgrad", C.byref(d), stream())
        if r.kind:
            recs.append(r)
    assert kinds == [1, 1, 2, 2, 2, 1]                                       # (wgrad_pw leaves block partials since round 5: bit-reproducible like the rest)
    table = (L.WgradPending * len(recs))()
    blocks = 0
    for i, r in enumerate(recs):
        r.block_begin = blocks
        blocks += r.blocks
        C.memmove(C.byref(table, i * C.sizeof(L.WgradPending)), C.byref(r), C.sizeof(L.WgradPending))
    tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev())
    lib.call("rua_wgrad_reduce_batch", tdev.data_ptr(), len(recs), blocks, stream())
    torch.cuda.synchronize()
    for i, d in enumerate(descs):
        got = keep[5 * i + 3].cpu().numpy()
        assert kinds[i] != 0
        if i == 4:                                                            # wgrad_pw: the immediate call went through replica atomics, the deferred one through block partials
            assert np.allclose(got, imm[i], rtol=1e-5, atol=1e-4), i
        else:
            assert np.array_equal(got, imm[i]), i


@pytest.mark.parametrize("shape", [(8, 64, 64, 128, [1, 3, 15]), (2, 256, 256, 32, [1, 3, 15, 31]), (4, 128, 128, 64, [3, 15, 31]), (8, 16, 16, 512, [1, 3]),
                                   (6, 128, 128, 32, [1, 3, 15, 31]), (8, 32, 32, 256, [1, 3, 15]), (3, 24, 40, 128, [1, 15]),
                                   (32, 64, 64, 64, [1, 3, 15, 31]), (5, 64, 64, 64, [1, 15])])     # C = 64 off the 128-pixel rows: conv_igemm_g
def test_conv_group_equals_separate_launches(shape):
    # (C = 128 / 256 groups run on conv_band128m since round 5 - test_conv_group_band64_multi compares that kernel with these members one by one;
    # this test keeps its subject, the grouped grids of the members' own kernels: conv_band128m off; and conv_img2 - which a conv on its own takes at
    # 16 x 16 x 512, d = 1, and a member of a group does not - off, so that "one by one" means the same kernel)
    L.lib().set_tuning(conv_band128m=0, conv_img2=0)
    try:
        _group_equals_separate(shape)
    finally:
        L.lib().set_tuning(conv_band128m=BAND128M_DEFAULT, conv_img2=1)


def _group_equals_separate(shape):
    """rua_conv_fwd_group: the dilation branches of a ResBlock in one call.  Members on the same kernel (conv_dmap at the
    64x64x128 level, conv_strip at 256x256x32) share ONE grid, the 128x128x64 level runs as ONE conv_band64m launch; members the launcher cannot group
    (split-K at 16x16x512) run one by one.  Either way the results are bit-identical to separate rua_conv_fwd calls."""
    N, H, W, Cs, dils = shape
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(Cs)
    x = to_dev(rng.standard_normal((N, H, W, Cs)).astype(np.float32), dt)
    aux = to_dev(rng.standard_normal((N, H, W, Cs)).astype(np.float32), dt)
    sc = torch.from_numpy((0.5 + rng.random(Cs)).astype(np.float32)).to(dev()); sh = torch.from_numpy((0.3 * rng.standard_normal(Cs)).astype(np.float32)).to(dev())
    ws = torch.zeros(8 << 20, dtype=torch.float32, device=dev())
    ws2 = torch.zeros(8 << 20, dtype=torch.float32, device=dev())
    keep, descs = [], []
    for dil in dils:
        w = to_dev((rng.standard_normal((9, Cs, Cs)) / np.sqrt(9 * Cs)).astype(np.float32), dt)
        bias = torch.from_numpy(rng.standard_normal(Cs).astype(np.float32)).to(dev())
        y = torch.zeros((N, H, W, Cs), dtype=torch.bfloat16, device=dev())
        stats = torch.zeros(32 * 2 * Cs, dtype=torch.float64, device=dev())
        d = L.ConvDesc()
        d.nseg = 1
        s = d.seg[0]
        s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = x.data_ptr(), w.data_ptr(), Cs, H, W, 0, dil, 9
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cs, 1, dt
        d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
        if Cs != 32:                                        # (C = 32: the form the engine issues - a data gradient has no bias - and conv_strip32s serves)
            d.bias = bias.data_ptr()
        d.aux, d.aux_mode, d.mscale, d.mshift = aux.data_ptr(), 2, sc.data_ptr(), sh.data_ptr()
        d.stats, d.stats_mode, d.stats_replicas = stats.data_ptr(), 2, 32
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        keep += [w, bias, y, stats]
        descs.append(d)
    sep = []
    for d in descs:
        lib.call("rua_conv_fwd", C.byref(d), stream())
    split = lib.raw("rua_conv_last_ksplit")() > 1               # K-split members are launched one by one
    torch.cuda.synchronize()
    for i in range(len(dils)):
        sep.append((keep[4 * i + 2].clone(), keep[4 * i + 3].clone()))
        keep[4 * i + 2].zero_(); keep[4 * i + 3].zero_()
    arr = (L.ConvDesc * len(descs))()
    for i, d in enumerate(descs):
        C.memmove(C.byref(arr, i * C.sizeof(L.ConvDesc)), C.byref(d), C.sizeof(L.ConvDesc))
    lib.call("rua_conv_fwd_group", arr, len(descs), stream())
    # 16x16x512: split-K members launch one by one; members of unequal job counts are renumbered each over their own jobs.  (128-pixel
    # strips, W = 128, on conv_strip32: d = 31 needs more LDS than two blocks per CU allow and did not share a grid with d = 1, 3, 15;
    # conv_strip32s lays its slots out for the largest dilation, so full-width strips of every dilation share ONE grid)
    assert lib.raw("rua_conv_group_last_grids")() == (len(dils) if split else 1)
    torch.cuda.synchronize()
    assert lib.raw("rua_conv_group_last_chain")() == 0
    band = lib.raw("rua_conv_group_last_band")() == 1          # 128x128x64: the group is ONE conv_band64m launch - another kernel than the
    assert band == (Cs == 64 and W % 128 == 0)                 # members' own (conv_igemm), another fp32 summation order
    for i in range(len(dils)):
        if band:
            assert rel_err(keep[4 * i + 2].float().cpu().numpy(), sep[i][0].float().cpu().numpy()) < tol(dt), (i, dils[i])
            assert np.allclose(keep[4 * i + 3].cpu().numpy().reshape(32, -1).sum(0), sep[i][1].cpu().numpy().reshape(32, -1).sum(0), rtol=1e-3, atol=1e-2)
            continue
        assert torch.equal(keep[4 * i + 2], sep[i][0]), (i, dils[i])
        # (outputs bit-identical; the statistics are fp32 per-lane partial sums over a block's rows folded in fp64: the members of a
        # grouped conv_strip launch share one round of blocks - longer chain segments than a launch on its own - so the partial sums
        # are taken over other row sets and differ in their last fp32 bits)
        # conv_strip32s: a block's piece of the group's rows is four times a lone launch's: bound = 2e-7 of the sum of the terms' magnitudes
        assert np.allclose(keep[4 * i + 3].cpu().numpy().reshape(32, -1).sum(0), sep[i][1].cpu().numpy().reshape(32, -1).sum(0), rtol=2e-6,
                           atol=2e-7 * N * H * W)
    if Cs >= 128 and not split:
        # tuning key dmap_chain (off by default: measured no gain): conv_dmap members with the same tiles walk back to back through ONE
        # grid, a block running every member over its pixel tile with the DMA ring kept alive across the epilogues - bit-identical
        for i in range(len(dils)):
            keep[4 * i + 2].zero_(); keep[4 * i + 3].zero_()
        lib.set_tuning(dmap_chain=3, dmap_spread=0)             # (the chain is built on the burst-issue form of the kernel)
        try:
            lib.call("rua_conv_fwd_group", arr, len(descs), stream())
            assert lib.raw("rua_conv_group_last_chain")() == len(dils) and lib.raw("rua_conv_group_last_grids")() == 1
            torch.cuda.synchronize()
        finally:
            lib.set_tuning(dmap_chain=0, dmap_spread=1)
        for i in range(len(dils)):
            assert torch.equal(keep[4 * i + 2], sep[i][0]), (i, dils[i])
            assert np.array_equal(keep[4 * i + 3].cpu().numpy().reshape(32, -1).sum(0), sep[i][1].cpu().numpy().reshape(32, -1).sum(0))


@pytest.mark.parametrize("shape", [(2, 256, 256, 32, [1, 3, 15, 31], True), (4, 128, 128, 64, [1, 3, 15, 31], False), (8, 32, 32, 256, [1, 3, 15], False),
                                   (8, 64, 64, 128, [1, 3, 15], False), (4, 32, 32, 64, [1, 3, 15], False),
                                   (8, 32, 32, 256, [1, 3, 15, 15, 3, 1], False), (2, 128, 128, 64, [1, 3, 15, 31, 31, 15, 3, 1], False)])   # both convolutions of every branch: up to RUA_MAX_WGRAD_GROUP members
def test_wgrad_group_equals_separate_launches(shape):
    """rua_conv_wgrad_group: the weight gradients of the dilation branches of a ResBlock in one call (wgrad_taps<32> with BatchNorm
    on load, wgrad_taps<64>, wgrad_dmap, wgrad_kernel with K slabs): ONE grid, results bit-identical to separate calls - both with
    the reductions run by the call itself and deferred to rua_wgrad_reduce_batch.  Members that share workspace run one by one."""
    N, H, W, Cs, dils, bn = shape
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(Cs + 1)
    a = to_dev(rng.standard_normal((N, H, W, Cs)).astype(np.float32), dt)
    sc = torch.from_numpy((0.5 + rng.random(Cs)).astype(np.float32)).to(dev()); sh = torch.from_numpy((0.3 * rng.standard_normal(Cs)).astype(np.float32)).to(dev())
    keep, descs = [], []
    for dil in dils:
        dy = to_dev(rng.standard_normal((N, H, W, Cs)).astype(np.float32), dt)
        d = L.WgradDesc()
        d.a, d.C, d.Hs, d.Ws, d.dy, d.Cout, d.H, d.W = a.data_ptr(), Cs, H, W, dy.data_ptr(), Cs, H, W
        d.N, d.stride, d.dil, d.taps, d.dtype = N, 1, dil, 9, dt
        if bn:
            d.in_scale, d.in_shift, d.in_relu = sc.data_ptr(), sh.data_ptr(), 1
        ws = torch.zeros(lib.raw("rua_wgrad_workspace_bytes")(C.byref(d)) // 4 + 16, dtype=torch.float32, device=dev())
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        dw = torch.zeros(9 * Cs * Cs, dtype=torch.float32, device=dev())
        d.dw = dw.data_ptr()
        keep += [dy, ws, dw]
        descs.append(d)
    sep = []
    for i, d in enumerate(descs):
        lib.call("rua_conv_wgrad", C.byref(d), stream())
        torch.cuda.synchronize()
        sep.append(keep[3 * i + 2].clone())
        keep[3 * i + 2].zero_()

    def as_array():
        arr = (L.WgradDesc * len(descs))()
        for i, d in enumerate(descs):
            C.memmove(C.byref(arr, i * C.sizeof(L.WgradDesc)), C.byref(d), C.sizeof(L.WgradDesc))
        return arr
    old = lib.get_tuning("wgrad_group")
    lib.set_tuning(wgrad_group=15)                          # every kernel family (wgrad_dmap is grouped only on request: slower in the step)
    lib.call("rua_conv_wgrad_group", as_array(), len(descs), stream())
    assert lib.raw("rua_wgrad_group_last_grids")() == 1
    torch.cuda.synchronize()
    for i in range(len(dils)):
        assert torch.equal(keep[3 * i + 2], sep[i]), (i, dils[i])
        keep[3 * i + 2].zero_()
    lib.set_tuning(wgrad_group=0)                           # no grouping: one launch per member
    lib.call("rua_conv_wgrad_group", as_array(), len(descs), stream())
    assert lib.raw("rua_wgrad_group_last_grids")() == len(descs)
    torch.cuda.synchronize()
    for i in range(len(dils)):
        assert torch.equal(keep[3 * i + 2], sep[i]), (i, dils[i])
        keep[3 * i + 2].zero_()
    lib.set_tuning(wgrad_group=old)
    # deferred reductions: the group leaves the partial sums, one batched launch adds them
    recs = []
    for d in descs:
        d.defer = 1
        r = L.WgradPending()
        lib.call("rua_wgrad_plan", C.byref(d), C.byref(r))
        assert r.kind != 0
        recs.append(r)
    lib.call("rua_conv_wgrad_group", as_array(), len(descs), stream())
    table = (L.WgradPending * len(recs))()
    blocks = 0
    for i, r in enumerate(recs):
        r.block_begin = blocks
        blocks += r.blocks
        C.memmove(C.byref(table, i * C.sizeof(L.WgradPending)), C.byref(r), C.sizeof(L.WgradPending))
    tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev())
    lib.call("rua_wgrad_reduce_batch", tdev.data_ptr(), len(recs), blocks, stream())
    torch.cuda.synchronize()
    for i in range(len(dils)):
        assert torch.equal(keep[3 * i + 2], sep[i]), ("deferred", i, dils[i])
        keep[3 * i + 2].zero_()
    # group_members: the all-taps members share ONE round of blocks (1 / n of the block partials each, longer chain segments) - same
    # gradient up to the order the fp32 partial sums are taken in, and the deferred record must name the smaller number of partials
    for mode in ("own reduction", "deferred"):
        for d in descs:
            d.group_members, d.defer = len(descs), (1 if mode == "deferred" else 0)
        recs = []
        for d in descs:
            r = L.WgradPending()
            lib.call("rua_wgrad_plan", C.byref(d), C.byref(r))
            recs.append(r)
        lib.call("rua_conv_wgrad_group", as_array(), len(descs), stream())
        if mode == "deferred":
            table = (L.WgradPending * len(recs))()
            blocks = 0
            for i, r in enumerate(recs):
                r.block_begin = blocks
                blocks += r.blocks
                C.memmove(C.byref(table, i * C.sizeof(L.WgradPending)), C.byref(r), C.sizeof(L.WgradPending))
            tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev())
            lib.call("rua_wgrad_reduce_batch", tdev.data_ptr(), len(recs), blocks, stream())
        torch.cuda.synchronize()
        for i in range(len(dils)):
            got, exp = keep[3 * i + 2].cpu().numpy(), sep[i].cpu().numpy()
            assert np.abs(got - exp).max() <= 2e-5 * np.abs(exp).max() + 1e-6, (mode, i, dils[i], float(np.abs(got - exp).max()))
            keep[3 * i + 2].zero_()
    for d in descs:
        d.group_members = 0
    # shared workspace: not groupable, still correct
    for d in descs:
        d.defer = 0
        d.workspace, d.workspace_bytes = descs[0].workspace, descs[0].workspace_bytes
    lib.call("rua_conv_wgrad_group", as_array(), len(descs), stream())
    assert lib.raw("rua_wgrad_group_last_grids")() == len(descs)
    torch.cuda.synchronize()
    for i in range(len(dils)):
        assert torch.equal(keep[3 * i + 2], sep[i]), ("shared", i, dils[i])


@pytest.mark.parametrize("N,H,W,dil,R", [(2, 256, 256, 1, 32), (1, 256, 256, 31, 8), (4, 128, 128, 3, 1), (1, 200, 512, 15, 16)])
def test_conv_strip_folded_batchnorm_coefficients(N, H, W, dil, R):
    """rua_conv_desc.in_fold: the convolution derives the training-mode BatchNorm coefficients of its input from the replicated
    fp64 statistics in its own prologue (no coefficient launch), publishes scale / shift / mean / rstd and updates the moving
    statistics once.  Against rua_bn_fwd's coefficient launch + the in_scale path: published vectors and moving statistics agree
    to fp32 rounding of a differently ordered fp64 sum, the conv output to bf16 rounding of those coefficients."""
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(H + dil)
    Cs = 32
    x = (1.5 * rng.standard_normal((N, H, W, Cs)) + 0.3).astype(np.float32)
    xd = to_dev(x, dt)
    w = to_dev((rng.standard_normal((9, Cs, Cs)) / np.sqrt(9 * Cs)).astype(np.float32), dt)
    bias = torch.from_numpy(rng.standard_normal(Cs).astype(np.float32)).to(dev())
    gamma = torch.from_numpy((0.5 + rng.random(Cs)).astype(np.float32)).to(dev())
    beta = torch.from_numpy((0.3 * rng.standard_normal(Cs)).astype(np.float32)).to(dev())
    M = N * H * W
    stats = torch.zeros(R * 2 * Cs, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats", xd.data_ptr(), M, Cs, stats.data_ptr(), R, dt, stream())
    res = []
    for fold in (False, True):
        mm = torch.full((Cs,), 0.25, dtype=torch.float32, device=dev()); mv = torch.full((Cs,), 2.0, dtype=torch.float32, device=dev())
        co = torch.zeros(4, Cs, dtype=torch.float32, device=dev())
        y = torch.zeros((N, H, W, Cs), dtype=torch.bfloat16, device=dev())
        d = L.ConvDesc()
        d.nseg = 1
        s = d.seg[0]
        s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd.data_ptr(), w.data_ptr(), Cs, H, W, 0, dil, 9
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cs, 1, dt
        d.y, d.out_stride, d.OH, d.OW, d.bias = y.data_ptr(), 1, H, W, bias.data_ptr()
        f = L.BnFold()
        if fold:
            f.stats, f.replicas, f.count, f.bessel_n, f.eps, f.momentum = stats.data_ptr(), R, float(M), float(M), 1e-3, 0.99
            f.gamma, f.beta, f.moving_mean, f.moving_var = gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr()
            f.scale, f.shift, f.mean, f.rstd = (co[i].data_ptr() for i in range(4))
            d.in_fold, d.in_relu = C.addressof(f), 1
        else:
            b = L.BnFwdDesc()
            b.x, b.M, b.C, b.dtype, b.nb, b.relu, b.training = None, M, Cs, dt, 1, 1, 1
            b.stats, b.replicas, b.count, b.bessel_n, b.momentum, b.eps = stats.data_ptr(), R, float(M), float(M), 0.99, 1e-3
            br = b.br[0]
            br.gamma, br.beta, br.moving_mean, br.moving_var = gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr()
            br.scale, br.shift, br.mean, br.rstd, br.out = (co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(), co[3].data_ptr(), None)
            lib.call("rua_bn_fwd", C.byref(b), stream())
            d.in_scale, d.in_shift, d.in_relu = co[0].data_ptr(), co[1].data_ptr(), 1
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 5
        lib.call("rua_conv_fwd", C.byref(d), stream())
        torch.cuda.synchronize()
        res.append((y.float().cpu().numpy(), co.cpu().numpy(), mm.cpu().numpy(), mv.cpu().numpy()))
    (y0, c0, mm0, mv0), (y1, c1, mm1, mv1) = res
    assert np.allclose(c1, c0, rtol=2e-6, atol=1e-7)
    assert np.allclose(mm1, mm0, rtol=1e-6) and np.allclose(mv1, mv0, rtol=1e-6)
    xr = rnd(dt, x).double().numpy()
    assert np.allclose(c1[2], xr.mean(axis=(0, 1, 2)), rtol=1e-4, atol=1e-5)       # published mean = the batch mean of the bf16 input
    assert np.allclose(mm1, 0.25 * 0.99 + 0.01 * xr.mean(axis=(0, 1, 2)), rtol=1e-4, atol=1e-6)
    assert rel_err(y1, y0) < 2e-3                                                 # a last-bit coefficient difference moves a few bf16 roundings
    # in_fold on a shape no normalise-on-load kernel serves is refused
    d.Cout = 64
    assert lib.raw("rua_conv_fwd")(C.byref(d), None) != 0


@pytest.mark.parametrize("shape", [(2, 40, 56, 16), (2, 32, 64, 32)])      # the second: every extent a power of two (rua_maxpool_bwd_multi's shift-and-mask form)
@pytest.mark.parametrize("dt", [L.RUA_BF16, L.RUA_F32])
def test_pooling_pyramid_passes_equal_the_separate_launches(dt, shape):
    """PSPPooling's 2 / 4 / 8 pyramid (model2.py:47-60): rua_maxpool_derive gives the values AND the argmax bytes of a direct
    rua_maxpool_fwd with twice the window from the level below (many ties: the input is quantised, the first maximum in row-major
    order must win),
    rua_maxpool_bwd_multi the sum of three rua_maxpool_bwd scatters (plain and accumulating), rua_sumpool_pyramid the three
    window sums."""
    rng = np.random.default_rng(21)
    lib = L.lib()
    N, H, W, Cc = shape
    x = np.round(rng.standard_normal((N, H, W, Cc)) * 2).astype(np.float32) / 2          # ties
    xd = to_dev(x, dt)
    ys, ids, gs = {}, {}, {}
    for k in (2, 4, 8):
        ys[k] = torch.empty((N, H // k, W // k, Cc), dtype=tdt(dt), device=dev())
        ids[k] = torch.empty(ys[k].numel(), dtype=torch.uint8, device=dev())
        lib.call("rua_maxpool_fwd", xd.data_ptr(), ys[k].data_ptr(), ids[k].data_ptr(), N, H, W, Cc, k, dt, stream())
        gs[k] = to_dev(rng.standard_normal((N, H // k, W // k, Cc)).astype(np.float32), dt)
    py = {k: torch.zeros_like(ys[k]) for k in ys}
    pi = {k: torch.full_like(ids[k], 255) for k in ids}
    lib.call("rua_maxpool_derive", ys[2].data_ptr(), ids[2].data_ptr(), py[4].data_ptr(), pi[4].data_ptr(), N, H // 2, W // 2, Cc, 2, dt, stream())
    lib.call("rua_maxpool_derive", py[4].data_ptr(), pi[4].data_ptr(), py[8].data_ptr(), pi[8].data_ptr(), N, H // 4, W // 4, Cc, 4, dt, stream())
    torch.cuda.synchronize()
    for k in (4, 8):
        assert torch.equal(py[k], ys[k]) and torch.equal(pi[k], ids[k]), k
    for acc in (0, 1):
        base = to_dev(rng.standard_normal((N, H, W, Cc)).astype(np.float32), dt)
        ref = base.clone()
        for j, k in enumerate((2, 4, 8)):
            lib.call("rua_maxpool_bwd", gs[k].data_ptr(), ids[k].data_ptr(), ref.data_ptr(), 1 if (acc or j) else 0, N, H, W, Cc, k, dt, stream())
        got = base.clone()
        dys = L.ptr_array([gs[k].data_ptr() for k in (2, 4, 8)]); ixs = L.ptr_array([ids[k].data_ptr() for k in (2, 4, 8)])
        kk = (C.c_int32 * 3)(2, 4, 8)
        lib.call("rua_maxpool_bwd_multi", 3, dys, ixs, kk, got.data_ptr(), acc, N, H, W, Cc, dt, stream())
        torch.cuda.synchronize()
        # one rounding after the three additions here, one per launch there
        assert rel_err(got.float().cpu().numpy(), ref.float().cpu().numpy()) < (1e-2 if dt == L.RUA_BF16 else 1e-6)
    sp = {k: torch.empty_like(ys[k]) for k in ys}
    lib.call("rua_sumpool_pyramid", xd.data_ptr(), sp[2].data_ptr(), sp[4].data_ptr(), sp[8].data_ptr(), N, H, W, Cc, dt, stream())
    torch.cuda.synchronize()
    xr = rnd(dt, x)
    for k in (2, 4, 8):
        exp = F.avg_pool2d(xr.permute(0, 3, 1, 2).double(), k, k).permute(0, 2, 3, 1).numpy() * k * k
        assert rel_err(sp[k].float().cpu().numpy(), exp) < tol(dt), k
    # shapes the pyramid cannot serve are refused
    assert lib.raw("rua_sumpool_pyramid")(xd.data_ptr(), sp[2].data_ptr(), sp[4].data_ptr(), sp[8].data_ptr(), N, 36, W, Cc, dt, None) != 0


@pytest.mark.parametrize("dt,act,Cout", [(L.RUA_BF16, L.ACT_SOFTMAX, 6), (L.RUA_BF16, L.ACT_SIGMOID, 3), (L.RUA_F32, L.ACT_SOFTMAX, 6), (L.RUA_BF16, L.ACT_SIGMOID, 6)])
def test_head_forward_with_loss_moments_equals_the_separate_passes(dt, act, Cout):
    """rua_head_fwd_loss = rua_head_fwd + rua_tanimoto_sums + rua_seg_metrics in one pass: logits and probabilities bit-identical,
    the fp64 moments / counts equal up to the order of the fp32 partial sums."""
    rng = np.random.default_rng(31 + Cout)
    lib = L.lib()
    B, Hh, Ww, Cin = 3, 40, 52, 32                                            # HW = 2080: blocks that end inside a sample
    HW, M = Hh * Ww, B * Hh * Ww
    x = to_dev(rng.standard_normal((M, Cin)).astype(np.float32), dt)
    w = torch.from_numpy((rng.standard_normal((Cout, Cin)) / 4).astype(np.float32)).to(dev())
    b = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32)).to(dev())
    lab = np.eye(Cout, dtype=np.float32)[rng.integers(0, Cout, size=M)]
    if act == L.ACT_SIGMOID:
        lab = (rng.random((M, Cout)) > 0.6).astype(np.float32)
    y = torch.from_numpy(lab).to(dev())
    z0 = torch.empty((M, Cout), device=dev()); p0 = torch.empty((M, Cout), device=dev())
    lib.call("rua_head_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), z0.data_ptr(), p0.data_ptr(), M, Cin, Cout, act, dt, stream())
    s0 = torch.zeros(B * Cout * 6, dtype=torch.float64, device=dev()); m0 = torch.zeros(5, dtype=torch.float64, device=dev())
    lib.call("rua_tanimoto_sums", p0.data_ptr(), y.data_ptr(), B, HW, Cout, s0.data_ptr(), stream())
    lib.call("rua_seg_metrics", p0.data_ptr(), y.data_ptr(), M, Cout, m0.data_ptr(), stream())
    z1 = torch.empty_like(z0); p1 = torch.empty_like(p0)
    s1 = torch.zeros_like(s0); m1 = torch.zeros_like(m0)
    lib.call("rua_head_fwd_loss", x.data_ptr(), w.data_ptr(), b.data_ptr(), z1.data_ptr(), p1.data_ptr(), y.data_ptr(), s1.data_ptr(), m1.data_ptr(),
             B, HW, Cin, Cout, act, dt, stream())
    torch.cuda.synchronize()
    assert torch.equal(z1, z0) and torch.equal(p1, p0)
    assert np.allclose(s1.cpu().numpy(), s0.cpu().numpy(), rtol=1e-5, atol=1e-3)
    assert np.array_equal(m1.cpu().numpy(), m0.cpu().numpy())                 # counts: exact
    assert m1[0].item() > 0 and abs(m1[1:].sum().item() - M * Cout) < 0.5     # TP + FP + TN + FN = every (pixel, class)
    # moments only / counts only
    s2 = torch.zeros_like(s0)
    lib.call("rua_head_fwd_loss", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, p1.data_ptr(), y.data_ptr(), s2.data_ptr(), None, B, HW, Cin, Cout, act, dt, stream())
    torch.cuda.synchronize()
    assert torch.equal(s2, s1) or np.allclose(s2.cpu().numpy(), s1.cpu().numpy(), rtol=1e-6, atol=1e-4)
    assert lib.raw("rua_head_fwd_loss")(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, p1.data_ptr(), y.data_ptr(), None, None, B, HW, Cin, Cout, act, dt, None) != 0
    # the moments spread over R copies (rua_head_fwd_loss_rep) + the fold in rua_tanimoto_finalize_rep: same loss, coefficients and per-sample values
    R = 8
    sr = torch.zeros((R + 1) * B * Cout * 6, dtype=torch.float64, device=dev())
    lib.call("rua_head_fwd_loss_rep", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, p1.data_ptr(), y.data_ptr(), sr.data_ptr(), R, None, B, HW, Cin, Cout, act, dt, stream())
    outs = []
    for sums, rep in ((s1, 1), (sr, R), (sr, R)):                             # twice: the fold is idempotent
        lo = torch.zeros(1, dtype=torch.float64, device=dev()); co = torch.zeros(B * Cout * 3, device=dev()); pe = torch.zeros(B, device=dev())
        lib.call("rua_tanimoto_finalize_rep", sums.data_ptr(), rep, B, HW, Cout, 0.25, lo.data_ptr(), co.data_ptr(), pe.data_ptr(), stream())
        torch.cuda.synchronize()
        outs.append((lo.item(), co.cpu().numpy(), pe.cpu().numpy()))
    folded = sr.view(R + 1, -1)
    assert np.allclose(folded[:R].sum(0).cpu().numpy(), s1.cpu().numpy(), rtol=1e-5, atol=1e-3)
    if dt == L.RUA_BF16:
        assert (folded[1:R].abs().sum(1) > 0).all()                           # every copy was used
    for o in outs[1:]:
        assert abs(o[0] - outs[0][0]) < 1e-6 and np.allclose(o[1], outs[0][1], rtol=1e-4, atol=1e-7) and np.allclose(o[2], outs[0][2], rtol=1e-5)
    assert outs[1][0] == outs[2][0] and np.array_equal(outs[1][1], outs[2][1])


def test_multi_head_loss_launches_equal_the_per_head_ones():
    """rua_tanimoto_finalize_multi / rua_head_dz_multi (the four heads of the multitask model in one launch each, train_ISPRS.py:417-421) against
    rua_tanimoto_finalize / rua_head_dz head by head: bit for bit."""
    rng = np.random.default_rng(77)
    lib = L.lib()
    B, HW = 3, 40 * 52
    M = B * HW
    heads = [(6, L.ACT_SOFTMAX, 1.0), (6, L.ACT_SIGMOID, 0.5), (6, L.ACT_SOFTMAX, 2.0), (3, L.ACT_SIGMOID, 1.0)]
    keep, th, dh, ref = [], [], [], []
    for Cc, act, wgt in heads:
        p = torch.from_numpy(rng.random((M, Cc)).astype(np.float32)).to(dev())
        if act == L.ACT_SOFTMAX:
            p = p / p.sum(1, keepdim=True)
        y = torch.from_numpy((rng.random((M, Cc)) > 0.6).astype(np.float32)).to(dev())
        sums = torch.zeros(B * Cc * 6, dtype=torch.float64, device=dev())
        lib.call("rua_tanimoto_sums", p.data_ptr(), y.data_ptr(), B, HW, Cc, sums.data_ptr(), stream())
        lo0, co0, dz0 = torch.zeros(1, dtype=torch.float64, device=dev()), torch.zeros(B * Cc * 3, device=dev()), torch.zeros((M, Cc), device=dev())
        lo1, co1, dz1 = torch.zeros_like(lo0), torch.zeros_like(co0), torch.zeros_like(dz0)
        lib.call("rua_tanimoto_finalize", sums.data_ptr(), B, HW, Cc, wgt / B, lo0.data_ptr(), co0.data_ptr(), None, stream())
        lib.call("rua_head_dz", L.LOSS_TANIMOTO, act, p.data_ptr(), y.data_ptr(), co0.data_ptr(), None, wgt / B, B, HW, Cc, dz0.data_ptr(), stream())
        t = L.TaniHead(); t.sums, t.replicas, t.B, t.C, t.grad_scale, t.loss_out, t.coef, t.per_sample = sums.data_ptr(), 1, B, Cc, wgt / B, lo1.data_ptr(), co1.data_ptr(), None
        d = L.DzHead(); d.kind, d.act, d.p, d.y, d.coef, d.class_w, d.grad_scale, d.B, d.HW, d.C, d.dz = L.LOSS_TANIMOTO, act, p.data_ptr(), y.data_ptr(), co1.data_ptr(), None, wgt / B, B, HW, Cc, dz1.data_ptr()
        th.append(t); dh.append(d); keep += [p, y, sums]; ref.append((lo0, co0, dz0, lo1, co1, dz1))
    ta = (L.TaniHead * len(th))(*th); da = (L.DzHead * len(dh))(*dh)
    lib.call("rua_tanimoto_finalize_multi", ta, len(th), stream())
    lib.call("rua_head_dz_multi", da, len(dh), stream())
    torch.cuda.synchronize()
    for lo0, co0, dz0, lo1, co1, dz1 in ref:
        assert torch.equal(lo0, lo1) and torch.equal(co0, co1) and torch.equal(dz0, dz1) and dz0.abs().sum().item() > 0
    assert lib.raw("rua_tanimoto_finalize_multi")(ta, 0, None) != 0 and lib.raw("rua_head_dz_multi")(da, 9, None) != 0


BAND_CASES = [
    # N, H, W, dilations, normalise-on-load kind, residual
    (2, 256, 256, [1, 3, 15, 31], "fold", True),      # the d6 residual atrous block's second convs (8-wave blocks, 256-pixel strips)
    (1, 256, 256, [1, 3, 15, 31], "scale", True),     # evaluation mode: coefficients given
    (4, 128, 128, [1, 3, 15, 31], "fold", True),      # cfg5's rows: 4-wave blocks
    (1, 512, 512, [1, 3, 15, 31], "fold", False),     # two strips per row (the halo columns are real data), no residual (model.py graph)
    (1, 256, 256, [31, 32], "none", True),            # two members, the largest dilation the halo holds, plain reads
    (1, 264, 384, [3, 15, 1], "scale", True),         # H, W no powers of two: 33 bands, three 128-pixel strips
    # C = Cout = 64 (conv_band64: 8 waves = 4 pixel tiles x 2 output-channel halves, 4-row bands, weights staged per kernel row)
    (2, 128, 128, [1, 3, 15, 31], "fold", True, 64),  # the level-2 ResBlock's second convs
    (1, 128, 128, [1, 3, 15, 31], "scale", True, 64),
    (1, 256, 256, [1, 3, 15, 31], "fold", False, 64), # two strips per row, no residual
    (1, 132, 384, [32, 1], "none", True, 64),         # three strips, 33 bands, plain reads: comparable with the member-by-member path
    (1, 128, 128, [3], "fold", True, 64),             # one member
]


@pytest.mark.parametrize("case", [(8, 64, 64, 128, [1, 3, 15], True), (2, 64, 64, 128, [1, 15], False), (8, 32, 32, 256, [1, 3, 15], True), (4, 32, 32, 256, [15, 3], True),
                                  (1, 64, 128, 128, [3, 1, 15, 31], True)])
def test_conv_segments_sum_on_band128(case):
    """rua_conv_fwd with one 3x3 segment per dilation branch (model2.py:26-31 at levels 3 - 4: out = x + sum_b (bias_b + conv_b(a2_b)), the inputs materialised BatchNorm
    outputs) as ONE conv_band128m launch with the accumulators kept over the members (tuning key conv_band128m, bit 3; rua_conv_kernel_id = 9) - against float64 on the
    bf16-rounded operands and against the K-concatenated conv_dmap form it replaces."""
    N, H, W, Cs, dils, res = case
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(H + Cs + len(dils))
    nb = len(dils)
    xs = [rng.standard_normal((N, H, W, Cs)).astype(np.float32) for _ in dils]
    ws = [(rng.standard_normal((9, Cs, Cs)) / np.sqrt(9 * Cs * nb)).astype(np.float32) for _ in dils]
    biases = [rng.standard_normal(Cs).astype(np.float32) for _ in dils]
    aux = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
    xd = [to_dev(a, dt) for a in xs]; wd = [to_dev(a, dt) for a in ws]
    bd = [torch.from_numpy(a).to(dev()) for a in biases]
    ad = to_dev(aux, dt)
    wsb = torch.zeros(8 << 20, dtype=torch.float32, device=dev())
    outs = {}
    try:
        for form in (13, 5):
            lib.set_tuning(conv_band128m=form)
            y = torch.full((N, H, W, Cs), 7.0, dtype=torch.bfloat16, device=dev())
            d = L.ConvDesc()
            d.nseg = nb
            for b in range(nb):
                sg = d.seg[b]
                sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = xd[b].data_ptr(), wd[b].data_ptr(), Cs, H, W, 0, dils[b], 9
            d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cs, 1, dt
            d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
            d.bias = bd[0].data_ptr()
            for b in range(1, nb):
                d.bias_more[b - 1] = bd[b].data_ptr()
            if res:
                d.aux, d.aux_mode = ad.data_ptr(), 1
            d.workspace, d.workspace_bytes = wsb.data_ptr(), wsb.numel() * 4
            assert (lib.raw("rua_conv_kernel_id")(C.byref(d)) == 9) == (form == 13)
            lib.call("rua_conv_fwd", C.byref(d), stream())
            torch.cuda.synchronize()
            outs[form] = y.float().cpu().numpy()
    finally:
        lib.set_tuning(conv_band128m=BAND128M_DEFAULT)
    exp = sum(ref_conv_nhwc(rnd(dt, xs[b]), rnd(dt, ws[b]), None, dils[b], 9).numpy() + biases[b].astype(np.float64) for b in range(nb))
    if res:
        exp = exp + rnd(dt, aux).numpy()
    assert rel_err(outs[13], exp) < tol(dt)
    assert rel_err(outs[13], outs[5]) < tol(dt)


@pytest.mark.parametrize("case", BAND_CASES)
def test_conv_sum_band_kernel(case):
    """rua_conv_fwd_sum -> conv_band32 (model2.py:26-31: out = x_input + sum of the branches' second convolutions, every branch
    reading relu(BatchNorm(y1_b)) normalised on load): one launch, accumulators of an 8-row band in registers over all branches,
    the output written once.  Against (a) a float64 evaluation on the bf16-rounded operands (each branch's normalised input rounded
    to bf16 as it is in LDS, zero padding AFTER the activation) and (b) the same members through separate accumulating launches;
    folded BatchNorms: published coefficients and moving statistics against the coefficient launch."""
    N, H, W, dils, kind, res = case[:6]
    Cs = case[6] if len(case) > 6 else 32
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(H + W + len(dils) + sum(dils) + Cs)
    nb, M = len(dils), N * H * W
    xs = [(1.2 * rng.standard_normal((N, H, W, Cs)) + 0.2 * b).astype(np.float32) for b in range(nb)]
    ws = [(rng.standard_normal((9, Cs, Cs)) / np.sqrt(9 * Cs * nb)).astype(np.float32) for _ in range(nb)]
    biases = [rng.standard_normal(Cs).astype(np.float32) for _ in range(nb)]
    aux = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
    gam = [(0.5 + rng.random(Cs)).astype(np.float32) for _ in range(nb)]
    bet = [(0.3 * rng.standard_normal(Cs)).astype(np.float32) for _ in range(nb)]
    xd = [to_dev(a, dt) for a in xs]; wd = [to_dev(a, dt) for a in ws]
    bd = [torch.from_numpy(a).to(dev()) for a in biases]
    ad = to_dev(aux, dt)
    gd = [torch.from_numpy(a).to(dev()) for a in gam]; btd = [torch.from_numpy(a).to(dev()) for a in bet]
    R = 8
    stats = []
    for b in range(nb):
        st = torch.zeros(R * 2 * Cs, dtype=torch.float64, device=dev())
        lib.call("rua_col_stats", xd[b].data_ptr(), M, Cs, st.data_ptr(), R, dt, stream())
        stats.append(st)
    # the coefficient launch as the reference for the folded path (and the source of in_scale / in_shift for the given-coefficient path)
    coef_ref, mm_ref, mv_ref = [], [], []
    for b in range(nb):
        mm = torch.full((Cs,), 0.25, dtype=torch.float32, device=dev()); mv = torch.full((Cs,), 2.0, dtype=torch.float32, device=dev())
        co = torch.zeros(4, Cs, dtype=torch.float32, device=dev())
        q = L.BnFwdDesc()
        q.x, q.M, q.C, q.dtype, q.nb, q.relu, q.training = None, M, Cs, dt, 1, 1, 1
        q.stats, q.replicas, q.count, q.bessel_n, q.momentum, q.eps = stats[b].data_ptr(), R, float(M), float(M), 0.99, 1e-3
        br = q.br[0]
        br.gamma, br.beta, br.moving_mean, br.moving_var = gd[b].data_ptr(), btd[b].data_ptr(), mm.data_ptr(), mv.data_ptr()
        br.scale, br.shift, br.mean, br.rstd, br.out = (co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(), co[3].data_ptr(), None)
        lib.call("rua_bn_fwd", C.byref(q), stream())
        coef_ref.append(co); mm_ref.append(mm); mv_ref.append(mv)
    torch.cuda.synchronize()

    def run(entry, use_kind):
        y = torch.full((N, H, W, Cs), 7.0, dtype=torch.bfloat16, device=dev())          # garbage the first member must overwrite
        arr = (L.ConvDesc * nb)()
        keep = []
        pub = []
        for b in range(nb):
            d = arr[b]
            d.nseg = 1
            s = d.seg[0]
            s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd[b].data_ptr(), wd[b].data_ptr(), Cs, H, W, 0, dils[b], 9
            d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cs, 1, dt
            d.y, d.out_stride, d.OH, d.OW, d.bias = y.data_ptr(), 1, H, W, bd[b].data_ptr()
            d.accumulate = 1 if b > 0 else 0
            if b == 0 and res:
                d.aux, d.aux_mode = ad.data_ptr(), 1
            if use_kind == "fold":
                mm = torch.full((Cs,), 0.25, dtype=torch.float32, device=dev()); mv = torch.full((Cs,), 2.0, dtype=torch.float32, device=dev())
                co = torch.zeros(4, Cs, dtype=torch.float32, device=dev())
                f = L.BnFold()
                f.stats, f.replicas, f.count, f.bessel_n, f.eps, f.momentum = stats[b].data_ptr(), R, float(M), float(M), 1e-3, 0.99
                f.gamma, f.beta, f.moving_mean, f.moving_var = gd[b].data_ptr(), btd[b].data_ptr(), mm.data_ptr(), mv.data_ptr()
                f.scale, f.shift, f.mean, f.rstd = (co[i].data_ptr() for i in range(4))
                d.in_fold, d.in_relu = C.addressof(f), 1
                keep.append(f); pub.append((co, mm, mv))
            elif use_kind == "scale":
                d.in_scale, d.in_shift, d.in_relu = coef_ref[b][0].data_ptr(), coef_ref[b][1].data_ptr(), 1
        run.arr = arr
        if entry == "sum":
            lib.call("rua_conv_fwd_sum", arr, nb, stream())
            ran = lib.raw("rua_conv_sum_last_kernel")()
        else:
            for b in range(nb):
                lib.call("rua_conv_fwd", C.byref(arr[b]), stream())
            ran = 0
        torch.cuda.synchronize()
        return y.float().cpu().numpy(), ran, pub

    got, ran, pub = run("sum", kind)
    assert ran == (1 if Cs == 32 else 2)                       # the band kernel (conv_band32 / conv_band64), not the member-by-member path
    assert lib.raw("rua_conv_sum_kernel")(run.arr, nb) == ran
    has_sep = Cs == 32 or kind == "none"                       # (no single-conv kernel normalises on load at C = 64: nothing to compare with there)
    sep = run("each", kind)[0] if has_sep else None
    exp = rnd(dt, aux).double().numpy() if res else 0.0
    for b in range(nb):
        xin = rnd(dt, xs[b])
        if kind != "none":
            sc, sh = coef_ref[b][0].cpu(), coef_ref[b][1].cpu()
            xin = torch.relu(xin * sc + sh).to(torch.bfloat16).float()
        exp = exp + ref_conv_nhwc(xin, rnd(dt, ws[b]), None, dils[b], 9).numpy() + biases[b].astype(np.float64)
    assert rel_err(got, exp) < 1e-2, rel_err(got, exp)          # one bf16 rounding of the output (the separate launches round it nb times)
    if has_sep:
        assert rel_err(sep, exp) < tol(dt)
        assert rel_err(got, sep) < tol(dt)
    if kind == "fold":
        for b in range(nb):
            co, mm, mv = pub[b]
            assert np.allclose(co.cpu().numpy(), coef_ref[b].cpu().numpy(), rtol=2e-6, atol=1e-7)
            assert np.allclose(mm.cpu().numpy(), mm_ref[b].cpu().numpy(), rtol=1e-6) and np.allclose(mv.cpu().numpy(), mv_ref[b].cpu().numpy(), rtol=1e-6)
    # with the tuning switch off, or a member the kernel does not take, the members run one by one - same numbers
    if has_sep:
        lib.set_tuning(conv_band=0, conv_band64=0)
        try:
            off, ran_off, _ = run("sum", kind)
        finally:
            lib.set_tuning(conv_band=1, conv_band64=1)
        assert ran_off == 0 and np.array_equal(off, sep)


@pytest.mark.parametrize("offset,nbytes", [(0, 1 << 20), (4, 1000), (3, 37), (16, 16), (8, 5), (0, 3 * (1 << 20) + 7), (1, 15)])
def test_fill_zero_is_a_kernel_and_exact_at_the_edges(offset, nbytes):
    """rua_fill_zero runs as an ordinary kernel (tuning key fill_kernel, default 1) so that captured HIP graphs hold no memset nodes
    (memset nodes corrupted the piecewise data-parallel step on this stack; tools/dp_graph_check.py).  Exactly
    [p, p + bytes) is cleared for any alignment / length; the bytes around it stay; the hipMemsetAsync path (fill_kernel=0) agrees."""
    lib = L.lib()
    assert lib.get_tuning("fill_kernel") == 1
    for mode in (1, 0):
        lib.set_tuning(fill_kernel=mode)
        try:
            buf = torch.full((offset + nbytes + 64,), 0xAB, dtype=torch.uint8, device=dev())
            lib.call("rua_fill_zero", buf.data_ptr() + offset, nbytes, stream())
            torch.cuda.synchronize()
            h = buf.cpu().numpy()
            assert (h[offset:offset + nbytes] == 0).all() and (h[:offset] == 0xAB).all() and (h[offset + nbytes:] == 0xAB).all(), mode
        finally:
            lib.set_tuning(fill_kernel=1)
    lib.call("rua_fill_zero", buf.data_ptr(), 0, stream())           # zero bytes: nothing launched, no error


BAND128M_DEFAULT = 13                                 # RuaTuning::conv_band128m (csrc/common.h)
BAND64M_CASES = [
    # N, H, W, dilations, kind: "first" = the branches' first convs (shared input, per-branch BatchNorm on load, bias, statistics sum v / sum v^2),
    #                           "dgrad" = their data gradients (own inputs, ReLU mask from an aux tensor, statistics sum g / sum g * aux)
    (2, 128, 128, [1, 3, 15, 31], "first"),
    (2, 128, 128, [1, 3, 15, 31], "dgrad"),
    (1, 256, 256, [3, 31], "first"),                  # two strips per row, two members
    (1, 132, 128, [1, 15, 32], "dgrad"),              # 33 bands, three members
    (1, 128, 128, [1, 3, 15, 31], "first_scale"),     # evaluation mode: coefficients given, no statistics
    # C = Cout = 128 on 64-pixel rows -> conv_band128m (round 5: the level-3 ResBlock, model2.py:105-106,128-131): two-row stages, 64-channel output slices,
    # tap columns outside the row read a zero pixel (d = 15 on 64 pixels: 15 of 64 columns per side)
    (8, 64, 64, [1, 3, 15], "first_plain", 128),      # the engine's form at this level: materialised BatchNorm input, bias, statistics
    (8, 64, 64, [1, 3, 15], "dgrad", 128),
    (2, 64, 64, [1, 3, 15], "first", 128),            # coefficients given (in_scale / in_shift) + ReLU on load, statistics
    (1, 36, 64, [3, 31, 15, 1], "dgrad", 128),        # 9 bands, four members, a dilation wider than half the row
    (3, 8, 64, [15, 3], "first_scale", 128),          # two bands: every row of the d = 15 member's outer kernel rows is padding
    (1, 128, 128, [1, 3, 15], "dgrad", 128),          # one row per stage, two-row bands (cfg4's level 3)
    (1, 64, 64, [1, 3, 15, 31], "first", 64),         # C = 64 on 64-pixel rows (cfg5's level 2): four rows per stage, eight-row bands, in_fold
    # C = Cout = 256 on 32-pixel rows (the level-4 ResBlock): 128 of the 256 input channels per phase, 32-channel output slices, the k-steps of a phase split
    # over two waves whose accumulators meet in the member epilogue; d = 15 on 32 x 32: most of the outer kernel rows and columns are padding
    (8, 32, 32, [1, 3, 15], "first_plain", 256),
    (8, 32, 32, [1, 3, 15], "dgrad", 256),
    (2, 32, 32, [3, 15], "first", 256),
    (2, 16, 32, [15, 1, 3, 5], "dgrad", 256),         # two bands per image, four members
]


@pytest.mark.parametrize("case", BAND64M_CASES)
def test_conv_group_band64_multi(case):
    try:
        _band_multi(case)
    finally:
        L.lib().set_tuning(conv_band128m=BAND128M_DEFAULT, conv_band64m=1)      # the defaults, whatever happened


def _band_multi(case):
    """rua_conv_fwd_group at C = Cout = 64 -> conv_band64m: the independent 3x3 convolutions of a level-2 ResBlock (model2.py:17-24 first
    convs of every dilation branch; their data gradients) as ONE row-streaming launch, every member with its own normalise-on-load,
    bias, ReLU mask and statistics.  Against float64 convolutions of the bf16-rounded operands (and, for the data-gradient form,
    against the members run one by one on the implicit-GEMM kernel: outputs to bf16 rounding, statistics to fp32 partial-sum order)."""
    N, H, W, dils, kind = case[:5]
    Cs = case[5] if len(case) > 5 else 64
    L.lib().set_tuning(conv_band128m=7)                       # every form of the kernel (bits: 1 C = 128, 2 C = 64, 4 C = 256), whatever the defaults are
    # C = 64 runs on conv_band128m's two-tiles-per-wave form since round 5 (tuning key conv_band128m bit 1; bit 1 off: conv_band64m, which the second
    # pass below also exercises), C = 128 on its one-tile form (bit 0)
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(H + W + len(dils) + len(kind))
    nb, M = len(dils), N * H * W
    first = kind.startswith("first")
    xs = [(1.1 * rng.standard_normal((N, H, W, Cs)) + 0.1).astype(np.float32) for _ in range(1 if first else nb)]
    ws = [(rng.standard_normal((9, Cs, Cs)) / np.sqrt(9 * Cs)).astype(np.float32) for _ in range(nb)]
    biases = [rng.standard_normal(Cs).astype(np.float32) for _ in range(nb)]
    auxs = [rng.standard_normal((N, H, W, Cs)).astype(np.float32) for _ in range(nb)]
    gam = [(0.5 + rng.random(Cs)).astype(np.float32) for _ in range(nb)]
    bet = [(0.3 * rng.standard_normal(Cs)).astype(np.float32) for _ in range(nb)]
    msc = [(0.5 + rng.random(Cs)).astype(np.float32) for _ in range(nb)]
    msh = [(0.3 * rng.standard_normal(Cs)).astype(np.float32) for _ in range(nb)]
    xd = [to_dev(a, dt) for a in xs]; wd = [to_dev(a, dt) for a in ws]; ad = [to_dev(a, dt) for a in auxs]
    dv = lambda a: torch.from_numpy(a).to(dev())
    bd, gd, btd, mscd, mshd = ([dv(a) for a in lst] for lst in (biases, gam, bet, msc, msh))
    R = 8
    st_in = torch.zeros(R * 2 * Cs, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats", xd[0].data_ptr(), M, Cs, st_in.data_ptr(), R, dt, stream())
    # the per-branch BatchNorm coefficients of the shared input, as the coefficient launch makes them (batch statistics, eps 1e-3)
    x0 = rnd(dt, xs[0]).double().numpy().reshape(M, Cs)
    mean, var = x0.mean(0), x0.var(0)
    coef_ref = []
    for b in range(nb):
        sc = gam[b] / np.sqrt(var + 1e-3)
        coef_ref.append(torch.from_numpy(np.stack([sc, bet[b] - mean * sc]).astype(np.float32)).to(dev()))
    torch.cuda.synchronize()

    def build(with_norm):
        ys = [torch.full((N, H, W, Cs), 3.0, dtype=torch.bfloat16, device=dev()) for _ in range(nb)]
        sts = [torch.zeros(R * 2 * Cs, dtype=torch.float64, device=dev()) for _ in range(nb)]
        arr = (L.ConvDesc * nb)()
        keep = []
        for b in range(nb):
            d = arr[b]
            d.nseg = 1
            s = d.seg[0]
            src = xd[0] if first else xd[b]
            s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = src.data_ptr(), wd[b].data_ptr(), Cs, H, W, 0, dils[b], 9
            d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cs, 1, dt
            d.y, d.out_stride, d.OH, d.OW = ys[b].data_ptr(), 1, H, W
            if first:
                d.bias = bd[b].data_ptr()
                if kind in ("first", "first_plain"):
                    d.stats, d.stats_mode, d.stats_replicas = sts[b].data_ptr(), 1, R
                    if with_norm and kind == "first" and Cs != 64:      # (no in_fold at this level: the coefficient launch makes them)
                        d.in_scale, d.in_shift, d.in_relu = coef_ref[b][0].data_ptr(), coef_ref[b][1].data_ptr(), 1
                    elif with_norm and kind == "first":
                        mm = torch.zeros(Cs, device=dev()); mv = torch.ones(Cs, device=dev()); co = torch.zeros(4, Cs, dtype=torch.float32, device=dev())
                        f = L.BnFold()
                        f.stats, f.replicas, f.count, f.bessel_n, f.eps, f.momentum = st_in.data_ptr(), R, float(M), float(M), 1e-3, 0.99
                        f.gamma, f.beta, f.moving_mean, f.moving_var = gd[b].data_ptr(), btd[b].data_ptr(), mm.data_ptr(), mv.data_ptr()
                        f.scale, f.shift, f.mean, f.rstd = (co[i].data_ptr() for i in range(4))
                        d.in_fold, d.in_relu = C.addressof(f), 1
                        keep += [f, mm, mv, co]
                elif with_norm and kind == "first_scale":
                    d.in_scale, d.in_shift, d.in_relu = coef_ref[b][0].data_ptr(), coef_ref[b][1].data_ptr(), 1
            else:
                d.aux, d.aux_mode, d.mscale, d.mshift = ad[b].data_ptr(), 2, mscd[b].data_ptr(), mshd[b].data_ptr()
                d.stats, d.stats_mode, d.stats_replicas = sts[b].data_ptr(), 2, R
        return arr, ys, sts, keep

    arr, ys, sts, keep = build(True)
    assert lib.raw("rua_conv_group_band_ok")(arr, nb) == 1
    lib.call("rua_conv_fwd_group", arr, nb, stream())
    assert lib.raw("rua_conv_group_last_band")() == 2 and lib.raw("rua_conv_group_last_grids")() == 1
    torch.cuda.synchronize()
    if kind == "first" and Cs == 64:                          # every member published its coefficients and moved its moving statistics once
        for b in range(nb):
            f, mm, mv, co = keep[4 * b:4 * b + 4]
            assert np.allclose(co[:2].cpu().numpy(), coef_ref[b].cpu().numpy(), rtol=1e-4, atol=1e-5), b
            assert np.allclose(mm.cpu().numpy(), 0.01 * mean, rtol=1e-4, atol=1e-6), b
            assert np.allclose(mv.cpu().numpy(), 0.99 + 0.01 * var * M / (M - 1), rtol=1e-4), b
    for b in range(nb):
        xin = rnd(dt, xs[0] if first else xs[b])
        if first and kind != "first_plain":
            sc, sh = coef_ref[b][0].cpu(), coef_ref[b][1].cpu()
            xin = torch.relu(xin * sc + sh).to(torch.bfloat16).float()
        exp = ref_conv_nhwc(xin, rnd(dt, ws[b]), None, dils[b], 9).numpy()
        a = rnd(dt, auxs[b]).double().numpy()
        if first:
            exp = exp + biases[b].astype(np.float64)
            s2 = (exp ** 2).sum(axis=(0, 1, 2))
        else:
            exp = exp * ((a * msc[b] + msh[b]) > 0)
            s2 = (exp * a).sum(axis=(0, 1, 2))
        got = ys[b].float().cpu().numpy()
        assert rel_err(got, exp) < tol(dt), (b, rel_err(got, exp))
        if kind != "first_scale":
            stv = sts[b].cpu().numpy().reshape(R, 2 * Cs).sum(0)
            assert rel_err(stv[:Cs], exp.sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4, b
            assert rel_err(stv[Cs:], s2) < 5 * tol(dt) + 1e-4, b
    if not first:                                              # the same members through the implicit-GEMM kernel, one grid or one by one
        for keys, flag in (([("conv_band128m", 0)], 1 if Cs == 64 else 0), ([("conv_band128m", 0), ("conv_band64m", 0)], 0)):
            lib.set_tuning(**dict(keys))
            try:
                arr2, ys2, sts2, _ = build(True)
                lib.call("rua_conv_fwd_group", arr2, nb, stream())
                assert lib.raw("rua_conv_group_last_band")() == flag
                torch.cuda.synchronize()
            finally:
                lib.set_tuning(conv_band128m=7, conv_band64m=1)
            for b in range(nb):
                assert rel_err(ys[b].float().cpu().numpy(), ys2[b].float().cpu().numpy()) < tol(dt)
                assert np.allclose(sts[b].cpu().numpy().reshape(R, -1).sum(0), sts2[b].cpu().numpy().reshape(R, -1).sum(0), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("N,H,W,Cs,Cout,dil", [(8, 64, 64, 128, 128, 3), (4, 32, 32, 256, 256, 15), (8, 16, 16, 512, 512, 1), (2, 40, 24, 192, 128, 1)])
def test_conv_dmap_issue_forms_are_bit_identical(N, H, W, Cs, Cout, dil):
    """conv_dmap's three ways of issuing a stage's DMA instructions (tuning key dmap_spread: 0 one burst behind the stage barrier, 1 a
    quarter per k-step between the MFMAs with four stage buffers - the default -, 2 / 6 by DMA waves of their own beside the MFMA waves,
    512-thread blocks) stage the same bytes and multiply in the same order: outputs and statistics are bit for bit the same, with and
    without a K split, and form 0 is the one the fp64 comparisons of test_conv_fwd were written against."""
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(Cs + dil)
    x = to_dev(rng.standard_normal((N, H, W, Cs)).astype(np.float32), dt)
    aux = to_dev(rng.standard_normal((N, H, W, Cout)).astype(np.float32), dt)
    w = to_dev((rng.standard_normal((9, Cout, Cs)) / np.sqrt(9 * Cs)).astype(np.float32), dt)
    bias = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32)).to(dev())
    sc = torch.from_numpy((0.5 + rng.random(Cout)).astype(np.float32)).to(dev()); sh = torch.from_numpy((0.3 * rng.standard_normal(Cout)).astype(np.float32)).to(dev())
    ws = torch.zeros(8 << 20, dtype=torch.float32, device=dev())
    res = {}
    try:
        lib.set_tuning(conv_img2=0)                          # (16 x 16 x 512, d = 1 is conv_img2's by default since round 5: the subject here is conv_dmap)
        for form in (0, 1, 2, 6):
            lib.set_tuning(dmap_spread=form)
            outs = []
            for kind in ("bias", "mask"):
                y = torch.zeros((N, H, W, Cout), dtype=torch.bfloat16, device=dev())
                stats = torch.zeros(8 * 2 * Cout, dtype=torch.float64, device=dev())
                d = L.ConvDesc()
                d.nseg = 1
                s = d.seg[0]
                s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = x.data_ptr(), w.data_ptr(), Cs, H, W, 0, dil, 9
                d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cout, 1, dt
                d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
                d.stats, d.stats_replicas = stats.data_ptr(), 8
                if kind == "bias":
                    d.bias, d.stats_mode = bias.data_ptr(), 1
                else:
                    d.aux, d.aux_mode, d.mscale, d.mshift, d.stats_mode = aux.data_ptr(), 2, sc.data_ptr(), sh.data_ptr(), 2
                d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
                assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 2
                lib.call("rua_conv_fwd", C.byref(d), stream())
                torch.cuda.synchronize()
                outs.append((y, stats.cpu().numpy().reshape(8, -1).sum(0)))
            res[form] = outs
    finally:
        lib.set_tuning(dmap_spread=1, conv_img2=1)
    exp = ref_conv_nhwc(rnd(dt, x.float().cpu().numpy()), rnd(dt, w.float().cpu().numpy()), None, dil, 9).numpy() + bias.cpu().numpy().astype(np.float64)
    assert rel_err(res[0][0][0].float().cpu().numpy(), exp) < tol(dt)
    for form in (1, 2, 6):
        for k in range(2):
            assert torch.equal(res[form][k][0], res[0][k][0]), (form, k)
            assert np.array_equal(res[form][k][1], res[0][k][1]), (form, k)


@pytest.mark.parametrize("M,Cc,n", [(8 * 64 * 64, 128, 3), (2 * 128 * 128, 64, 4), (600, 32, 2), ((8 * 64 * 64, 8 * 32 * 32, 8 * 16 * 16, 8 * 8 * 8), 256, 4)])
def test_bn_bwd_group_equals_separate_launches(M, Cc, n):
    """rua_bn_bwd_group: the one-branch BatchNorm backwards of a ResBlock's dilation branches (model2.py:21-22: one BatchNorm per branch
    behind its first conv) in ONE grid; outputs and dgamma / dbeta bit for bit those of n rua_bn_bwd calls."""
    dt = L.RUA_BF16
    lib = L.lib()
    Ms = list(M) if isinstance(M, tuple) else [M] * n              # unequal pixel counts: the branch BatchNorms of a PSPPooling (model2.py:56-66)
    rng = np.random.default_rng(Ms[0] + Cc)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    xs = [to_dev((rng.standard_normal((Ms[b], Cc)) * 1.3 + 0.2).astype(np.float32), dt) for b in range(n)]
    gs = [to_dev(rng.standard_normal((Ms[b], Cc)).astype(np.float32), dt) for b in range(n)]
    gam = [f(rng.uniform(0.5, 1.5, Cc)) for _ in range(n)]
    coef = [torch.zeros(4, Cc, device=dev()) for _ in range(n)]
    for b in range(n):                                       # any consistent coefficients serve: scale, shift, mean, rstd
        xm = xs[b].float()
        mean, var = xm.mean(0), xm.var(0, unbiased=False)
        coef[b][2], coef[b][3] = mean, torch.rsqrt(var + 1e-3)
        coef[b][0] = gam[b] * coef[b][3]
        coef[b][1] = 0.1 - mean * coef[b][0]
    st2 = [torch.zeros(4 * 2 * Cc, dtype=torch.float64, device=dev()) for _ in range(n)]
    for b in range(n):
        lib.call("rua_col_stats2", gs[b].data_ptr(), xs[b].data_ptr(), coef[b][0].data_ptr(), coef[b][1].data_ptr(), 1, Ms[b], Cc, st2[b].data_ptr(), 4, dt, stream())
    res = []
    for grouped in (False, True):
        dxs = [torch.zeros((Ms[b], Cc), dtype=torch.bfloat16, device=dev()) for b in range(n)]
        dg = [torch.zeros(Cc, device=dev()) for _ in range(n)]; db = [torch.zeros(Cc, device=dev()) for _ in range(n)]
        arr = (L.BnBwdDesc * n)()
        for b in range(n):
            e = arr[b]
            e.x, e.dx, e.M, e.C, e.dtype, e.nb, e.masked, e.accumulate, e.count = xs[b].data_ptr(), dxs[b].data_ptr(), Ms[b], Cc, dt, 1, 1, 0, float(Ms[b])
            br = e.br[0]
            br.g, br.stats2, br.replicas, br.gamma = gs[b].data_ptr(), st2[b].data_ptr(), 4, gam[b].data_ptr()
            br.scale, br.shift, br.mean, br.rstd = [coef[b][i].data_ptr() for i in range(4)]
            br.dgamma, br.dbeta = dg[b].data_ptr(), db[b].data_ptr()
        if grouped:
            lib.call("rua_bn_bwd_group", arr, n, stream())
            assert lib.raw("rua_bn_bwd_group_last_grids")() == 1
        else:
            for b in range(n):
                lib.call("rua_bn_bwd", C.byref(arr[b]), stream())
        torch.cuda.synchronize()
        res.append((dxs, dg, db))
    for b in range(n):
        assert torch.equal(res[0][0][b], res[1][0][b]), b
        assert torch.equal(res[0][1][b], res[1][1][b]) and torch.equal(res[0][2][b], res[1][2][b]), b
    assert float(res[1][0][0].float().abs().sum()) > 0


@pytest.mark.parametrize("M,Cc,n,training", [(8 * 64 * 64, 128, 3, 1), (8 * 32 * 32, 256, 3, 0), (700, 32, 2, 1), ((8 * 64 * 64, 8 * 32 * 32, 8 * 16 * 16, 8 * 8 * 8), 8, 4, 1)])
def test_bn_fwd_group_equals_separate_launches(M, Cc, n, training):
    """rua_bn_fwd_group: the second BatchNorms of a ResBlock's dilation branches where they are materialised (model2.py:21: one per branch, own input,
    own statistics) in ONE grid; outputs, published coefficients and moving statistics bit for bit those of n rua_bn_fwd calls."""
    dt = L.RUA_BF16
    lib = L.lib()
    Ms = list(M) if isinstance(M, tuple) else [M] * n
    rng = np.random.default_rng(Ms[0] + Cc)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    xs = [to_dev((rng.standard_normal((Ms[b], Cc)) * 1.3 + 0.2).astype(np.float32), dt) for b in range(n)]
    gam = [f(rng.uniform(0.5, 1.5, Cc)) for _ in range(n)]; bet = [f(rng.standard_normal(Cc)) for _ in range(n)]
    sts = [torch.zeros(4 * 2 * Cc, dtype=torch.float64, device=dev()) for _ in range(n)]
    for b in range(n):
        lib.call("rua_col_stats", xs[b].data_ptr(), Ms[b], Cc, sts[b].data_ptr(), 4, dt, stream())
    res = []
    for grouped in (False, True):
        outs = [torch.zeros((Ms[b], Cc), dtype=torch.bfloat16, device=dev()) for b in range(n)]
        co = [torch.zeros(4, Cc, device=dev()) for _ in range(n)]
        mm = [f(np.full(Cc, 0.25)) for _ in range(n)]; mv = [f(np.full(Cc, 2.0)) for _ in range(n)]
        arr = (L.BnFwdDesc * n)()
        for b in range(n):
            q = arr[b]
            q.x, q.M, q.C, q.dtype, q.nb, q.relu, q.training = xs[b].data_ptr(), Ms[b], Cc, dt, 1, 1, training
            q.stats, q.replicas, q.count, q.bessel_n, q.momentum, q.eps = sts[b].data_ptr(), 4, float(Ms[b]), float(Ms[b]) * (b + 1), 0.99, 1e-3
            br = q.br[0]
            br.gamma, br.beta, br.moving_mean, br.moving_var = gam[b].data_ptr(), bet[b].data_ptr(), mm[b].data_ptr(), mv[b].data_ptr()
            br.scale, br.shift, br.mean, br.rstd, br.out = (co[b][0].data_ptr(), co[b][1].data_ptr(), co[b][2].data_ptr(), co[b][3].data_ptr(), outs[b].data_ptr())
        if grouped:
            lib.call("rua_bn_fwd_group", arr, n, stream())
            assert lib.raw("rua_bn_fwd_group_last_grids")() == 1
        else:
            for b in range(n):
                lib.call("rua_bn_fwd", C.byref(arr[b]), stream())
        torch.cuda.synchronize()
        res.append((outs, co, mm, mv))
    for b in range(n):
        for k in range(4):
            assert torch.equal(res[0][k][b], res[1][k][b]), (b, k)
    assert float(res[1][0][0].float().abs().sum()) > 0
    xr = xs[0].float()
    if training:
        assert np.allclose(res[1][2][0].cpu().numpy(), 0.25 * 0.99 + 0.01 * xr.mean(0).cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("aux_mode", [2, 3])
def test_conv_small_data_gradient_epilogues(aux_mode):
    """conv_small behind the data gradients of the bottleneck's 1x1 convolutions (model2.py:41-79 backward): ReLU mask from aux with scale / shift
    (aux_mode 2) or aux as the second statistics operand only (aux_mode 3), statistics sum g / sum g * aux, 8 .. 512 pixels, two sources."""
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(aux_mode)
    for N, H, W in ((8, 8, 8), (8, 1, 1)):
        Cout = 320
        segs = [(256, 0), (64, 0)]
        M = N * H * W
        d = L.ConvDesc()
        d.nseg = len(segs)
        keep, exp = [], 0
        for i, (Cs, up) in enumerate(segs):
            x = rng.standard_normal((N, H, W, Cs)).astype(np.float32)
            w = (rng.standard_normal((1, Cout, Cs)) / np.sqrt(320)).astype(np.float32)
            xd, wd = to_dev(x, dt), to_dev(w, dt)
            keep += [xd, wd]
            s = d.seg[i]
            s.x, s.w, s.C, s.Hs, s.Ws, s.up_shift, s.dil, s.taps = xd.data_ptr(), wd.data_ptr(), Cs, H, W, up, 1, 1
            exp = exp + ref_conv_nhwc(rnd(dt, x), rnd(dt, w), None, 1, 1).numpy()
        aux = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
        ms, mt = rng.standard_normal(Cout).astype(np.float32), rng.standard_normal(Cout).astype(np.float32)
        ad, msd, mtd = to_dev(aux, dt), torch.from_numpy(ms).to(dev()), torch.from_numpy(mt).to(dev())
        y = torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device=dev())
        stats = torch.zeros(4 * 2 * Cout, dtype=torch.float64, device=dev())
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, H, W, Cout, 1, dt
        d.aux, d.aux_mode = ad.data_ptr(), aux_mode
        if aux_mode == 2:
            d.mscale, d.mshift = msd.data_ptr(), mtd.data_ptr()
        d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, H, W
        d.stats, d.stats_mode, d.stats_replicas = stats.data_ptr(), 2, 4
        assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 6
        lib.call("rua_conv_fwd", C.byref(d), stream())
        torch.cuda.synchronize()
        a = rnd(dt, aux).numpy().astype(np.float64)
        if aux_mode == 2:
            exp = exp * ((np.float32(ms) * a.astype(np.float32) + np.float32(mt)) > 0)
        assert rel_err(y.float().cpu().numpy(), exp) < tol(dt), (N, H, W)
        st = stats.cpu().numpy().reshape(4, 2 * Cout).sum(0)
        assert rel_err(st[:Cout], exp.sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4
        assert rel_err(st[Cout:], (exp * a).sum(axis=(0, 1, 2))) < 5 * tol(dt) + 1e-4


def test_conv_small_members_of_unequal_size_share_one_launch():
    """rua_conv_fwd_group over the four branch convolutions of the bottleneck PSPPooling (model2.py:47-66 at 8 x 8 x 1024: 512 / 128 / 32 / 8 pooled pixels,
    1024 -> 256, statistics in the epilogue): ONE conv_small_g launch (members with grids of unequal size), outputs and statistics bit-identical to four
    rua_conv_fwd calls; the five per-source data gradients of the fuse conv go as 4 + 1 members the same way (Graph.conv_group splits at RUA_MAX_BRANCH)."""
    dt = L.RUA_BF16
    lib = L.lib()
    rng = np.random.default_rng(123)
    N, Cin, Cout = 8, 1024, 256
    keep, descs = [], []
    for hw in (8, 4, 2, 1):
        x = to_dev(rng.standard_normal((N, hw, hw, Cin)).astype(np.float32), dt)
        w = to_dev((rng.standard_normal((1, Cout, Cin)) / 32).astype(np.float32), dt)
        b = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32)).to(dev())
        ys = [torch.zeros((N, hw, hw, Cout), dtype=torch.bfloat16, device=dev()) for _ in range(2)]
        sts = [torch.zeros(2 * 2 * Cout, dtype=torch.float64, device=dev()) for _ in range(2)]
        keep += [x, w, b, ys, sts]
        pair = []
        for y, st in zip(ys, sts):
            d = L.ConvDesc()
            d.nseg = 1
            sg = d.seg[0]
            sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = x.data_ptr(), w.data_ptr(), Cin, hw, hw, 0, 1, 1
            d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, hw, hw, Cout, 1, dt
            d.bias = b.data_ptr()
            d.y, d.out_stride, d.OH, d.OW = y.data_ptr(), 1, hw, hw
            d.stats, d.stats_mode, d.stats_replicas = st.data_ptr(), 1, 2
            assert lib.raw("rua_conv_kernel_id")(C.byref(d)) == 6
            pair.append(d)
        descs.append(pair)
    for pair in descs:
        lib.call("rua_conv_fwd", C.byref(pair[0]), stream())
    arr = (L.ConvDesc * 4)()
    for i, pair in enumerate(descs):
        C.memmove(C.byref(arr, i * C.sizeof(L.ConvDesc)), C.byref(pair[1]), C.sizeof(L.ConvDesc))
    lib.call("rua_conv_fwd_group", arr, 4, stream())
    assert lib.raw("rua_conv_group_last_grids")() == 1
    torch.cuda.synchronize()
    for i in range(4):
        ys, sts = keep[5 * i + 3], keep[5 * i + 4]
        assert torch.equal(ys[0], ys[1]) and ys[0].float().abs().sum().item() > 0, i
        assert torch.equal(sts[0], sts[1]), i
    lib.set_tuning(conv_group=15)                          # without the conv_small bit: four launches, same results
    try:
        for i in range(4):
            keep[5 * i + 3][1].zero_(); keep[5 * i + 4][1].zero_()
        lib.call("rua_conv_fwd_group", arr, 4, stream())
        assert lib.raw("rua_conv_group_last_grids")() == 4
        torch.cuda.synchronize()
    finally:
        lib.set_tuning(conv_group=31)
    for i in range(4):
        assert torch.equal(keep[5 * i + 3][0], keep[5 * i + 3][1]) and torch.equal(keep[5 * i + 4][0], keep[5 * i + 4][1]), i


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_bn_fwd_output_statistics_from_coefficients(dt):
    """rua_bn_branch.out_stats: the per-channel sum / sum of squares of a training-mode BatchNorm's output WITHOUT ReLU, written from the coefficients
    (sum = M beta, sum of squares = M (beta^2 + gamma^2 var / (var + eps))) - what a rua_col_stats pass over the output returns, up to the output's rounding
    (fp32 storage: 1e-5; bf16: the rounding noise of the stored tensor, 1e-3) - so that the BatchNorms of the next ResBlock (model2.py:17 behind 86) skip that pass."""
    rng = np.random.default_rng(21)
    lib = L.lib()
    M, Cc = 4096, 64
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    x = to_dev((rng.standard_normal((M, Cc)) * 1.7 + 0.4).astype(np.float32), dt)
    gam, bet = f(rng.uniform(0.5, 1.5, Cc)), f(rng.standard_normal(Cc))
    st = torch.zeros(4 * 2 * Cc, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats", x.data_ptr(), M, Cc, st.data_ptr(), 4, dt, stream())
    out = torch.zeros((M, Cc), dtype=tdt(dt), device=dev())
    co = torch.zeros(4, Cc, device=dev()); mm, mv = f(np.zeros(Cc)), f(np.ones(Cc))
    ost = torch.zeros(2 * Cc, dtype=torch.float64, device=dev())
    q = L.BnFwdDesc()
    q.x, q.M, q.C, q.dtype, q.nb, q.relu, q.training = x.data_ptr(), M, Cc, dt, 1, 0, 1
    q.stats, q.replicas, q.count, q.bessel_n, q.momentum, q.eps = st.data_ptr(), 4, float(M), float(M), 0.99, 1e-3
    br = q.br[0]
    br.gamma, br.beta, br.moving_mean, br.moving_var = gam.data_ptr(), bet.data_ptr(), mm.data_ptr(), mv.data_ptr()
    br.scale, br.shift, br.mean, br.rstd, br.out = (co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(), co[3].data_ptr(), out.data_ptr())
    br.out_stats = ost.data_ptr()
    lib.call("rua_bn_fwd", C.byref(q), stream())
    chk = torch.zeros(2 * Cc, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats", out.data_ptr(), M, Cc, chk.data_ptr(), 1, dt, stream())
    torch.cuda.synchronize()
    a, b = ost.cpu().numpy(), chk.cpu().numpy()
    scale = np.abs(b).max()
    assert np.abs(a - b).max() < (1e-5 if dt == L.RUA_F32 else 1e-3) * scale, np.abs(a - b).max() / scale
    q.relu = 1                                                   # with a ReLU the output's statistics are not a function of the coefficients: refused
    assert lib.raw("rua_bn_fwd")(C.byref(q), None) != 0


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
def test_bn_bwd_output_sums_are_the_bias_gradient(dt):
    """rua_bn_bwd_desc.dx_stats: the per-channel sums of what the launch writes to dx - the bias gradient of the convolution that produced x (the stride-2
    convs in front of the encoder ResBlocks, model2.py:103-111) - against a rua_col_stats pass over dx (taken before rounding: tolerance of the storage type)."""
    rng = np.random.default_rng(33)
    lib = L.lib()
    M, Cc, R = 5000, 64, 2
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    x = to_dev((rng.standard_normal((M, Cc)) * 1.4 + 0.3).astype(np.float32), dt)
    skip = to_dev(rng.standard_normal((M, Cc)).astype(np.float32), dt)
    gs = [to_dev(rng.standard_normal((M, Cc)).astype(np.float32), dt) for _ in range(3)]
    gam = [f(rng.uniform(0.5, 1.5, Cc)) for _ in range(3)]
    coef = [torch.zeros(4, Cc, device=dev()) for _ in range(3)]
    xm = x.float()
    mean, var = xm.mean(0), xm.var(0, unbiased=False)
    for b in range(3):
        coef[b][2], coef[b][3] = mean, torch.rsqrt(var + 1e-3)
        coef[b][0] = gam[b] * coef[b][3]
        coef[b][1] = 0.2 * b - mean * coef[b][0]
    st2 = [torch.zeros(R * 2 * Cc, dtype=torch.float64, device=dev()) for _ in range(3)]
    for b in range(3):
        lib.call("rua_col_stats2", gs[b].data_ptr(), x.data_ptr(), coef[b][0].data_ptr(), coef[b][1].data_ptr(), 1, M, Cc, st2[b].data_ptr(), R, dt, stream())
    dx = torch.zeros((M, Cc), dtype=tdt(dt), device=dev())
    dg = [torch.zeros(Cc, device=dev()) for _ in range(3)]; db = [torch.zeros(Cc, device=dev()) for _ in range(3)]
    sums = torch.zeros(8 * 2 * Cc, dtype=torch.float64, device=dev())
    e = L.BnBwdDesc()
    e.x, e.dskip, e.dx, e.M, e.C, e.dtype, e.nb, e.masked, e.accumulate, e.count = x.data_ptr(), skip.data_ptr(), dx.data_ptr(), M, Cc, dt, 3, 1, 0, float(M)
    for b in range(3):
        br = e.br[b]
        br.g, br.stats2, br.replicas, br.gamma = gs[b].data_ptr(), st2[b].data_ptr(), R, gam[b].data_ptr()
        br.scale, br.shift, br.mean, br.rstd = [coef[b][i].data_ptr() for i in range(4)]
        br.dgamma, br.dbeta = dg[b].data_ptr(), db[b].data_ptr()
    e.dx_stats, e.dx_replicas = sums.data_ptr(), 8
    lib.call("rua_bn_bwd", C.byref(e), stream())
    chk = torch.zeros(2 * Cc, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats", dx.data_ptr(), M, Cc, chk.data_ptr(), 1, dt, stream())
    torch.cuda.synchronize()
    got = sums.cpu().numpy().reshape(8, 2 * Cc).sum(0)[:Cc]
    exp = chk.cpu().numpy()[:Cc]
    assert np.abs(got - exp).max() < (1e-4 if dt == L.RUA_F32 else 2e-2) * max(1.0, np.abs(exp).max()), np.abs(got - exp).max()
    assert float(np.abs(exp).max()) > 1.0


@pytest.mark.parametrize("dt", [L.RUA_F32, L.RUA_BF16])
@pytest.mark.parametrize("Cc,use_scratch", [(6, True), (3, False)])
def test_head_bwd_sums_of_dx(dt, Cc, use_scratch):
    """rua_head_bwd_sums: dx, dW, db as rua_head_bwd (bit for bit) plus the per-channel sums of the masked dx it writes - the bias gradient of the 3x3 + ReLU
    conv in front of the head (model2.py:153-171) - on both reduction paths (per-block partials + fixed-order reduce, fp32 atomics)."""
    lib = L.lib()
    rng = np.random.default_rng(Cc)
    M, Cin = 3 * 40 * 24, 32
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    x = to_dev(np.maximum(rng.standard_normal((M, Cin)), 0).astype(np.float32), dt)
    dz, w = f(rng.standard_normal((M, Cc))), f(rng.standard_normal((Cc, Cin)) / 6)
    scratch = torch.zeros(1 << 20, device=dev())
    res = []
    for sums in (False, True):
        dx = torch.zeros((M, Cin), dtype=tdt(dt), device=dev())
        dw, db, ds = torch.zeros((Cc, Cin), device=dev()), torch.zeros(Cc, device=dev()), torch.full((Cin,), 0.5, device=dev())
        sp, sb = (scratch.data_ptr(), scratch.numel() * 4) if use_scratch else (None, 0)
        if sums:
            lib.call("rua_head_bwd_sums", x.data_ptr(), dz.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, dw.data_ptr(), db.data_ptr(), ds.data_ptr(), sp, sb,
                     M, Cin, Cc, dt, 1, stream())
        else:
            lib.call("rua_head_bwd", x.data_ptr(), dz.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, dw.data_ptr(), db.data_ptr(), sp, sb, M, Cin, Cc, dt, 1, stream())
        torch.cuda.synchronize()
        res.append((dx, dw, db, ds))
    assert torch.equal(res[0][0], res[1][0])
    if use_scratch:                                            # (the atomic path adds in arrival order)
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    else:
        assert torch.allclose(res[0][1], res[1][1], rtol=1e-4, atol=1e-4)
    exp = (dz.double() @ w.double()) * (x.double() > 0)
    got = res[1][3].cpu().numpy() - 0.5                          # += semantics
    assert np.abs(got - exp.sum(0).cpu().numpy()).max() < (1e-3 if dt == L.RUA_F32 else 2e-2) * max(1.0, float(exp.sum(0).abs().max()))


def test_bn_bwd_statistics_in_terms_of_the_output():
    """rua_bn_bwd_branch.stats2_out: the BatchNorm backward of a ReLU-less BatchNorm fed with (sum g, sum g * OUT) - what the launch that wrote g takes
    through rua_bn_bwd_desc.dx_stats (slot 1 = sum dx * x) - instead of (sum g, sum g * x) from a rua_col_stats2 pass: same dx, dgamma, dbeta (fp32)."""
    dt = L.RUA_F32
    lib = L.lib()
    rng = np.random.default_rng(44)
    M, Cc, R = 3000, 32, 2
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev())
    x = f(rng.standard_normal((M, Cc)) * 1.4 + 0.3)
    g = f(rng.standard_normal((M, Cc)))
    gam, bet = f(rng.uniform(0.5, 1.5, Cc)), f(rng.standard_normal(Cc))
    mean, var = x.mean(0), x.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-3)
    scale = gam * rstd
    shift = bet - mean * scale
    out = x * scale + shift                                    # the BatchNorm's output (no ReLU): the tensor the next ResBlock reads
    # (1) the writer of g: a one-branch BatchNorm backward over `out` (any coefficients) that also takes sum dx and sum dx * out
    co2 = torch.stack([torch.ones(Cc, device=dev()), torch.zeros(Cc, device=dev()), out.mean(0), torch.rsqrt(out.var(0, unbiased=False) + 1e-3)])
    gg = f(rng.standard_normal((M, Cc)))
    s_in = torch.zeros(R * 2 * Cc, dtype=torch.float64, device=dev())
    lib.call("rua_col_stats2", gg.data_ptr(), out.data_ptr(), co2[0].data_ptr(), co2[1].data_ptr(), 0, M, Cc, s_in.data_ptr(), R, dt, stream())
    gdx = torch.zeros((M, Cc), device=dev())
    sums = torch.zeros(4 * 2 * Cc, dtype=torch.float64, device=dev())
    e = L.BnBwdDesc()
    e.x, e.dx, e.M, e.C, e.dtype, e.nb, e.masked, e.accumulate, e.count = out.data_ptr(), gdx.data_ptr(), M, Cc, dt, 1, 0, 0, float(M)
    br = e.br[0]
    g1 = torch.ones(Cc, device=dev())
    br.g, br.stats2, br.replicas, br.gamma = gg.data_ptr(), s_in.data_ptr(), R, g1.data_ptr()
    br.scale, br.shift, br.mean, br.rstd = [co2[i].data_ptr() for i in range(4)]
    e.dx_stats, e.dx_replicas = sums.data_ptr(), 4
    lib.call("rua_bn_bwd", C.byref(e), stream())
    torch.cuda.synchronize()
    s = sums.cpu().numpy().reshape(4, 2 * Cc).sum(0)
    gd = gdx.double()
    assert np.allclose(s[:Cc], gd.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    assert np.allclose(s[Cc:], (gd * out.double()).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    # (2) the consumer: the backward of the BatchNorm that produced `out`, its gradient being gdx
    res = []
    for mode in (0, 1):
        st = torch.zeros(4 * 2 * Cc, dtype=torch.float64, device=dev())
        if mode == 0:
            lib.call("rua_col_stats2", gdx.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(), 0, M, Cc, st.data_ptr(), 4, dt, stream())
        else:
            st.copy_(sums)
        dx = torch.zeros((M, Cc), device=dev()); dgm, dbt = torch.zeros(Cc, device=dev()), torch.zeros(Cc, device=dev())
        q = L.BnBwdDesc()
        q.x, q.dx, q.M, q.C, q.dtype, q.nb, q.masked, q.accumulate, q.count = x.data_ptr(), dx.data_ptr(), M, Cc, dt, 1, 0, 0, float(M)
        b = q.br[0]
        b.g, b.stats2, b.replicas, b.gamma, b.stats2_out = gdx.data_ptr(), st.data_ptr(), 4, gam.data_ptr(), mode
        b.scale, b.shift, b.mean, b.rstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        b.dgamma, b.dbeta = dgm.data_ptr(), dbt.data_ptr()
        lib.call("rua_bn_bwd", C.byref(q), stream())
        torch.cuda.synchronize()
        res.append((dx.cpu().numpy(), dgm.cpu().numpy(), dbt.cpu().numpy()))
    for a, b_ in zip(res[0], res[1]):
        assert np.abs(a - b_).max() < 2e-4 * max(1.0, np.abs(a).max()), np.abs(a - b_).max()
