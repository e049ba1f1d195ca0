"""Headline benchmark: training patches/sec of the ResUnet-a multitask path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per
GPU, what the driver does), or started plainly - then THIS process, which has made no GPU call, starts exactly that
launcher as a child process (never exec), relays rank 0's JSON line and exits with the child's status
(/root/reference/train_ISPRS.py:347,432: MirroredStrategy over the node's GPUs).

A step = one full train_on_batch on one resident synthetic batch per GPU: forward (training-mode BN),
Tanimoto-dual losses on the four heads, backward, gradient all-reduce (N>1), Adam update, bf16 weight
refresh.  Default workload = BASELINE config 3 (256x256x6, 6 classes, multitask, bs 8/GPU, bf16).
Prints ONE JSON line (rank 0) with the bench contract fields plus `roofline` (dominant kernel, timed
live with events on the launch stream) and `cpu_baseline` (the CPU oracle timed on the host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

torch = None          # imported by main() AFTER the self-launch branch: the launching parent never initialises the GPU

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (patch, channels, classes, multitask, batch/GPU, train GFLOP/patch)
    "cfg3": (256, 6, 6, True, 8, 252.4),
    "cfg2": (256, 6, 6, False, 8, 234.1),
    "cfg5": (128, 7, 2, False, 32, 58.4),
    "cfg4": (512, 6, 6, True, 4, 1072.0),       # d7 extrapolation, 512x512 patches
}
DEPTH = {"cfg4": 7}
BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured streaming copy)


def conv_flops(desc):
    k = sum(desc.seg[i].taps * desc.seg[i].C for i in range(desc.nseg))
    return 2.0 * desc.N * desc.H * desc.W * desc.Cout * k


def wgrad_flops(d):
    return 2.0 * d.N * d.H * d.W * d.Cout * d.C * d.taps


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/collect_traffic.sh ->
    profiles/traffic_summary.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes).  The counters cannot be read from
    inside this process, so this is the value measured on the same workload with rocprofv3; None if it was not collected
    for this workload / dtype."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    if t.get("workload") != args.workload or t.get("dtype") != args.dtype:
        return None
    name = kernel.split("+")[0]
    k = t["kernels"].get(name)
    if k:
        return k["bytes_per_launch"]
    fam = [v for kk, v in t["kernels"].items() if kk.startswith(rocprof_prefixes(name))]
    n = sum(v["launches_seen"] for v in fam)
    return round(sum(v["bytes_per_launch"] * v["launches_seen"] for v in fam) / n) if n else None


def wg_taps_name(d, grouped):
    """The kernel behind rua_wgrad_kind() == 1 as rocprofv3 names it (csrc/conv_mfma.hip::launch_wgrad_taps)."""
    g = "_g" if grouped else ""
    if d.C in (128, 256):
        return f"wgrad_rowsx{g}<{1 if d.C == 256 else 0}>"
    if d.C == 64 and d.W == 128:
        return f"wgrad_rows64{g}"
    return f"wgrad_taps_kernel{g}<{d.C}>"


def rocprof_prefixes(name):
    """rocprofv3 kernel-name prefixes of the template instantiations a bench kernel name folds together (conv_halo<32> =
    conv_halo<32,2> + conv_halo<32,3>, ...; conv_strip<32> = conv_strip32<8,true,5> + conv_strip32<8,false,7> + the round-4
    conv_strip32s<8,...> forms; wgrad_taps_kernel<32> = wgrad_taps_kernel<32> + wgrad_rows32<4,...>)."""
    if name.startswith("conv_strip<"):
        c = name[len("conv_strip<"):-1]
        return ("conv_strip" + c + "<", "conv_strip" + c + "s<")
    if name.startswith("conv_strip_g<"):
        c = name[len("conv_strip_g<"):-1]
        return ("conv_strip" + c + "_g<", "conv_strip" + c + "s_g<")
    if name == "wgrad_taps_kernel<32>":
        return (name, "wgrad_rows32<")
    if name == "wgrad_taps_kernel_g<32>":
        return (name, "wgrad_rows32_g<")
    if name in ("wgrad_rows64", "wgrad_rows64_g"):
        return (name + "<",)
    return (name[:-1] + ",",)


def committed_rocprof_avg(kernel, args):
    """rocprofv3's average duration (us) of `kernel` in the committed profile (profiles/rocprof_avg.json, written by
    profiles/make_readme.py from r01_final/kernel_stats.csv) - printed beside the live figure, never used in it."""
    path = os.path.join(ROOT, "profiles", "rocprof_avg.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    if t.get("workload") != args.workload or t.get("dtype") != args.dtype:
        return None
    return t["avg_us"].get(kernel)


def dmap_name(bm, bn, grouped):
    """The LDS-DMA conv kernel as rocprofv3 names it: its issue form (tuning key dmap_spread) picks the instantiation."""
    from resunet_a_mltsk_keras_amd import _lib as L
    sp = L.lib().get_tuning("dmap_spread")
    form = "w" if (sp & 2 and (bm == 128 or sp & 4)) else "s" if (sp == 1 and bm == 128) else ""
    if grouped:
        return f"conv_dmap_g{form}<{bm},{bn}>"
    return f"conv_dmap_{form}<{bm},{bn}>" if form else f"conv_dmap<{bm},{bn}>"


def ungrouped(kn):
    for a, b in (("conv_dmap_gs<", "conv_dmap_s<"), ("conv_dmap_gw<", "conv_dmap_w<"), ("_g<", "<")):
        if a in kn:
            return kn.replace(a, b)
    return kn


def entry_bytes(name, args):
    """Algorithmic HBM bytes of one C-ABI launch = every activation tensor its descriptor names, once (weights and
    per-channel vectors included where they are not negligible); None for entries without a descriptor rule."""
    es = lambda dt: 2 if dt == 1 else 4
    if name in ("rua_conv_fwd", "rua_conv_fwd_group"):
        ds = [args[0]._obj] if name == "rua_conv_fwd" else [args[0][i] for i in range(args[1])]
        tot = 0
        seen = set()                                        # a tensor several members read (the ResBlock input of the first convs) counts once
        for d in ds:
            e = es(d.dtype)
            for i in range(d.nseg):
                sg = d.seg[i]
                tot += (0 if sg.x in seen else d.N * sg.Hs * sg.Ws * sg.C * e) + sg.taps * sg.C * d.Cout * e
                seen.add(sg.x)
            tot += d.N * d.H * d.W * d.Cout * e * (2 if (d.accumulate or d.out_stride > 1) else 1)       # output (read back when accumulating)
            if d.aux_mode:
                tot += d.N * d.H * d.W * d.Cout * e
        return float(tot)
    if name == "rua_conv_fwd_sum":                          # n convolutions summed into one output (conv_band32: written once)
        from resunet_a_mltsk_keras_amd import _lib as L
        ds = [args[0][i] for i in range(args[1])]
        e = es(ds[0].dtype)
        px = ds[0].N * ds[0].H * ds[0].W
        tot = sum(px * d.seg[0].C * e + d.seg[0].taps * d.seg[0].C * d.Cout * e for d in ds)
        one = L.lib().raw("rua_conv_sum_last_kernel")() in (1, 2)
        tot += px * ds[0].Cout * e * ((1 if one else 2 * len(ds) - 1) + (1 if ds[0].aux_mode else 0))
        return float(tot)
    if name in ("rua_conv_wgrad", "rua_conv_wgrad_group"):
        ds = [args[0]._obj] if name == "rua_conv_wgrad" else [args[0][i] for i in range(args[1])]
        return float(sum((d.N * d.Hs * d.Ws * d.C + d.N * d.H * d.W * d.Cout) * es(d.dtype) + d.taps * d.C * d.Cout * 4 for d in ds))
    if name in ("rua_bn_fwd", "rua_bn_fwd_group"):
        ds = [args[0]._obj] if name == "rua_bn_fwd" else [args[0][i] for i in range(args[1])]
        return float(sum(d.M * d.C * es(d.dtype) * (1 + d.nb) for d in ds if d.x))
    if name in ("rua_bn_bwd", "rua_bn_bwd_group"):
        ds = [args[0]._obj] if name == "rua_bn_bwd" else [args[0][i] for i in range(args[1])]
        return float(sum(d.M * d.C * es(d.dtype) * (d.nb + 1 + (1 if d.dskip else 0) + 1 + (1 if d.accumulate else 0)) for d in ds))
    if name in ("rua_col_stats",):
        return float(args[1] * args[2] * es(args[5]))
    if name in ("rua_col_stats2",):
        return float(2 * args[5] * args[6] * es(args[9]))
    return None


def profile_kernels(eng, g, dtype):
    """One extra (untimed) step with an event pair around every conv / wgrad launch on the launch stream.
    Returns per-kernel-instantiation totals: {name: [launches, seconds, flops]}."""
    from resunet_a_mltsk_keras_amd import _lib as L
    lib = L.lib()
    s = torch.cuda.current_stream().cuda_stream
    sp = C.c_void_p(s)
    tname = "bf16" if dtype == "bf16" else "f32"
    rec = []
    other = []                                              # every C-ABI entry: (name, e0, e1, algorithmic bytes[, flops])
    kbytes = {}                                             # MFMA kernel row -> algorithmic bytes of its launches
    mid = lib.raw("rua_profile_mid_event")
    ev_new, ev_rec, ev_us, ev_del = (lib.raw("rua_prof_event_" + n) for n in ("create", "record", "elapsed_us", "destroy"))
    events = []

    def mark(record=True):                                  # timing event without the system-scope release (include/rua_hip.h)
        e = C.c_void_p(ev_new())
        if not e.value:
            lib.check(-1, "rua_prof_event_create")
        events.append(e)
        if record:
            ev_rec(e, sp)
        return e

    def us(a, b):
        v = C.c_double()
        lib.check(ev_us(a, b, C.byref(v)), "rua_prof_event_elapsed_us")
        return v.value

    # Keep the GPU busy while the instrumented step is enqueued, so every launch is already queued when its turn comes and
    # the event pair brackets GPU execution only (bracketing a launch into an idle queue adds ~10 us of dispatch latency
    # per kernel and would disagree with rocprofv3's per-kernel durations).
    for _ in range(4):
        eng.train_step(None, None, fetch=False)
    # Bracket overhead, calibrated on an idempotent library kernel (the statistics-arena fill): T1 = bracket around one
    # launch, T2 = around two back-to-back launches, so T2 - T1 is what one more kernel costs IN the stream and
    # 2*T1 - T2 is what the two markers add.  (An empty marker pair over-estimates it: ~5 us, part of which overlaps
    # the kernel when there is one.)
    cal1, cal2 = [], []
    for _ in range(12):
        a = mark(); eng._zero_arena(g, s); b = mark(); cal1.append((a, b))
        a = mark(); eng._zero_arena(g, s); eng._zero_arena(g, s); b = mark(); cal2.append((a, b))
    eng._zero_arena(g, s)
    e0 = mark(); eng._prep_weights(s); e1 = mark()            # the per-step weight refresh (data-gradient layout from the optimizer's bf16 copy)
    other.append(("rua_weight_prep", e0, e1, None))
    empty = []
    for pname, plan in (("fwd", g.fwd), ("loss", g.loss_plan), ("bwd", g.bwd)):
        for ci, (fn, name, args, _lane) in enumerate(plan.calls):
            if ci % 16 == 0:                                # empty event pairs: the marker-to-marker cost to subtract
                empty.append((mark(), mark()))
            if fn is None:
                continue                                    # fork / join markers: this pass runs everything on one stream
            if name == "rua_conv_fwd_group":               # the dilation branches of a ResBlock in one grid: one row, N x the FLOPs
                arr, n = args
                e0 = mark()
                rc = fn(*args, sp)
                e1 = mark()
                d0 = arr[0]
                kid = lib.raw("rua_conv_kernel_id")(C.byref(d0))
                bm_, bn_ = lib.raw('rua_conv_tile_bm')(C.byref(d0)), lib.raw('rua_conv_tile_bn')(C.byref(d0))
                # rocprofv3 names the grouped grids conv_igemm_g<...>, conv_dmap_g<...>, conv_strip32_g<...>
                kn = (f"conv_igemm_g<{tname},{bm_},{bn_}>", f"conv_dma<{bm_},{bn_}>", dmap_name(bm_, bn_, True), f"conv_halo<{d0.Cout}>", "conv_pw", f"conv_strip_g<{d0.Cout}>", "conv_small<2>")[kid]
                grids = lib.raw("rua_conv_group_last_grids")()
                nl = 1
                band = lib.raw("rua_conv_group_last_band")()
                if band:                                    # the members as ONE row-streaming launch (1: conv_band64m, 2: conv_band128m at C = 64 / 128)
                    kn = "conv_band64m" if band == 1 else f"conv_band128m<{d0.Cout}>"
                elif grids == n and all(lib.raw("rua_conv_kernel_id")(C.byref(arr[i])) == kid for i in range(n)):
                    kn, nl = ungrouped(kn), n              # not grouped: n launches of the plain kernel
                elif grids != 1:                            # members the launchers could not put into one grid
                    kn = f"{ungrouped(kn)} ({n} members in {grids} launches)"
                fl = sum(conv_flops(arr[i]) for i in range(n))
                rec.append((kn, e0, e1, fl, (d0.N * d0.H * d0.W, d0.Cout, d0.seg[0].taps * d0.seg[0].C, 0, f"group of {n}"), nl))
            elif name == "rua_conv_fwd_sum":               # the branches' second convs summed on chip: one launch (conv_band32) or n
                arr, n = args
                e0 = mark()
                rc = fn(*args, sp)
                e1 = mark()
                which = lib.raw("rua_conv_sum_last_kernel")()
                kn, nl = (("conv_band32", "conv_band64")[which - 1], 1) if which else (f"conv_strip<{arr[0].Cout}>", n)
                fl = sum(conv_flops(arr[i]) for i in range(n))
                rec.append((kn, e0, e1, fl, (arr[0].N * arr[0].H * arr[0].W, arr[0].Cout, n * 9 * arr[0].seg[0].C, 0, f"sum of {n}"), nl))
            elif name == "rua_conv_wgrad_group":          # the branches' weight gradients in one grid (reductions deferred)
                arr, n = args
                e0 = mark()
                rc = fn(*args, sp)
                e1 = mark()
                d0 = arr[0]
                wk = lib.raw("rua_wgrad_kind")(C.byref(d0))
                kn = (f"wgrad_kernel_g", wg_taps_name(d0, True), "wgrad_dmap_g", "wgrad_pw")[wk]
                grids = lib.raw("rua_wgrad_group_last_grids")()
                nl = 1
                if grids == n:                              # this kernel family is not grouped: n launches of the plain kernel
                    kn, nl = kn.replace("_g", ""), n
                elif grids != 1:
                    kn = f"{kn.replace('_g', '')} ({n} members in {grids} launches)"
                fl = sum(wgrad_flops(arr[i]) for i in range(n))
                rec.append((kn, e0, e1, fl, (d0.N * d0.H * d0.W, d0.Cout, d0.C * d0.taps, 0, f"group of {n}"), nl))
            elif name in ("rua_conv_fwd", "rua_conv_wgrad"):
                em = mark(False)                            # recorded by the library between the main kernel and the
                mid(em)                                     # second launch of a two-launch call
                e0 = mark()
                rc = fn(*args, sp)
                e1 = mark()
                mid(None)
                fired = lib.raw("rua_profile_mid_event_fired")() == 1
                d = args[0]._obj
                if name == "rua_conv_fwd":
                    bm_, bn_ = lib.raw('rua_conv_tile_bm')(C.byref(d)), lib.raw('rua_conv_tile_bn')(C.byref(d))
                    two = fired
                    kid = lib.raw("rua_conv_kernel_id")(C.byref(d)); kn = (f"conv_igemm<{tname},{bm_},{bn_}>", f"conv_dma<{bm_},{bn_}>", dmap_name(bm_, bn_, False), f"conv_halo<{d.Cout}>", "conv_pw", f"conv_strip<{d.Cout}>", "conv_small<2>", f"conv_img<{d.W},{1 if d.H * d.W * d.seg[0].C * 2 <= 131072 else 2}>", f"conv_img2<{d.W}>", f"conv_band128m<{d.Cout}> (sum)")[kid]
                    kk = sum(d.seg[i].taps * d.seg[i].C for i in range(d.nseg))
                    flags = ("stats" if d.stats_mode else "") + (f" aux{d.aux_mode}" if d.aux_mode else "") + (" acc" if d.accumulate else "") + (" splitk" if two else "")
                    fl, tag, second = conv_flops(d), (d.N * d.H * d.W, d.Cout, kk, d.seg[0].dil, flags), f"conv_splitk_finish<{tname}>"
                else:
                    wk = lib.raw("rua_wgrad_kind")(C.byref(d)); kn = (f"wgrad_kernel<{tname}>", wg_taps_name(d, False), "wgrad_dmap", "wgrad_pw")[wk]
                    ik = lib.raw("rua_wgrad_img_kind")(C.byref(d)) if wk == 0 else 0
                    if ik:
                        kn = f"wgrad_img<{d.W}>" if ik == 1 else "wgrad_imgs"
                    two = fired
                    fl, tag, second = wgrad_flops(d), (d.N * d.H * d.W, d.Cout, d.C * d.taps, d.dil, ""), ("wgrad_taps_reduce" if wk == 1 else "wgrad_slab_reduce")
                if two:                                     # per-kernel rows, as rocprofv3 names them
                    rec.append((kn, e0, em, fl, tag)); rec.append((second, em, e1, 0.0, tag))
                else:
                    rec.append((kn, e0, e1, fl, tag))
            else:
                e0 = mark()
                rc = fn(*args, sp)
                e1 = mark()
                other.append((name, e0, e1, entry_bytes(name, args)))
            if name in ("rua_conv_fwd", "rua_conv_wgrad", "rua_conv_fwd_group", "rua_conv_wgrad_group", "rua_conv_fwd_sum"):
                other.append((name, e0, e1, entry_bytes(name, args), fl))      # the whole call (incl. a split-K finisher)
                kbytes[kn] = kbytes.get(kn, 0.0) + (entry_bytes(name, args) or 0.0)
            if rc != 0:
                lib.check(rc, name)
    e0 = mark()
    eng.optimizer_step(1.0 / eng.world)
    e1 = mark()
    other.append(("rua_adam_step" if eng.loss.optimizer == "adam" else "rua_sgd_step", e0, e1,
                  float(eng.params.n) * (32.0 if eng.loss.optimizer == "adam" else 24.0)))
    # Second instrumented step: one marker where the composite (ResBlock) changes, nothing inside the blocks - the block
    # times carry no per-kernel brackets.  A composite's launches are contiguous in the plan.
    scopes = {}                                             # (pass, scope) -> [first event, last event, conv+wgrad FLOPs, tensor-pass bytes]
    eng._zero_arena(g, s)
    eng._prep_weights(s)
    for pname, plan in (("fwd", g.fwd), ("loss", g.loss_plan), ("bwd", g.bwd)):
        cur = None
        for ci, (fn, name, args, _lane) in enumerate(plan.calls):
            sc = plan.scopes[ci]
            if sc != cur:
                ev = mark()
                if cur is not None:
                    scopes[(pname, cur)][1] = ev
                if sc is not None:
                    scopes[(pname, sc)] = [ev, None, 0.0, 0.0]
                cur = sc
            if fn is None:
                continue
            if sc is not None:
                scopes[(pname, sc)][3] += entry_bytes(name, args) or 0.0
            if sc is not None and name in ("rua_conv_fwd", "rua_conv_wgrad"):
                scopes[(pname, sc)][2] += conv_flops(args[0]._obj) if name == "rua_conv_fwd" else wgrad_flops(args[0]._obj)
            elif sc is not None and name in ("rua_conv_fwd_group", "rua_conv_fwd_sum"):
                scopes[(pname, sc)][2] += sum(conv_flops(args[0][i]) for i in range(args[1]))
            elif sc is not None and name == "rua_conv_wgrad_group":
                scopes[(pname, sc)][2] += sum(wgrad_flops(args[0][i]) for i in range(args[1]))
            rc = fn(*args, sp)
            if rc != 0:
                lib.check(rc, name)
        if cur is not None:
            scopes[(pname, cur)][1] = mark()
    eng.optimizer_step(1.0 / eng.world)
    torch.cuda.synchronize()
    med = lambda pairs: sorted(us(a, b) for a, b in pairs)[len(pairs) // 2]
    t1, t2, empty_us = med(cal1), med(cal2), med(empty)
    ov = max(2 * t1 - t2, 0.0) * 1e-6                    # seconds the two markers add to a bracket with a kernel inside
    out = {}
    for kn, e0, e1, fl, _, *nl in rec:
        t = out.setdefault(kn, [0, 0.0, 0.0])
        t[0] += nl[0] if nl else 1; t[1] += max(us(e0, e1) * 1e-6 - ov, 1e-7); t[2] += fl
    out["_event_overhead_us"] = ov * 1e6
    # whole composites (every launch of a ResBlock: BN passes, convolutions, weight / data gradients), forward and backward
    out["_blocks"] = {f"{pn}:{sc}": (max(us(e0, e1) * 1e-6 - ov, 1e-7), fl, by) for (pn, sc), (e0, e1, fl, by) in scopes.items() if e1 is not None}
    ent = {}
    for name, e0, e1, by, *fl in other:
        t = ent.setdefault(name, [0, 0.0, 0.0, 0.0])
        t[0] += 1; t[1] += max(us(e0, e1) * 1e-6 - ov, 1e-7); t[2] += by or 0.0; t[3] += fl[0] if fl else 0.0
    out["_entries"] = ent
    out["_kbytes"] = kbytes
    log(f"event bracket calibration: one fill {t1:.2f} us, two fills {t2:.2f} us, empty pair {empty_us:.2f} us -> overhead {ov * 1e6:.2f} us")
    if os.environ.get("RUA_BENCH_DETAIL"):
        groups = {}
        for kn, e0, e1, fl, tag, *_ in rec:
            gkey = (kn,) + tag
            t = groups.setdefault(gkey, [0, 0.0, 0.0])
            t[0] += 1; t[1] += us(e0, e1) * 1e-6; t[2] += fl
        log("per-shape MFMA kernel table (kernel, M, Cout, K, dil, flags): launches, total ms, avg us, TFLOP/s")
        for gkey, (n, sec, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
            log(f"  {str(gkey):90s} n={n:3d} {1e3 * sec:8.3f} ms {1e6 * sec / n:8.1f} us {fl / sec / 1e12:7.1f} TF/s")
    for e in events:
        ev_del(e)
    return out


def d6_ceiling_probe(batch, reps=30):
    """The north star's sub-metric, priced: what the SAME row-streaming kernel family does on the d6 block's product - 4 members x [B x 256 x 256
    pixels x 288] . [288 x 32], model2.py:15-34,102 - when everything the training-mode BatchNorm barrier hangs on it is taken away: no normalise-on-load,
    no bias, no statistics, no epilogue stream - row DMA + MFMA + the output store only (the plain form of conv_strip32s, one grouped launch).  The block is
    two such grids (first and second convolutions, 38.65 GFLOP each at batch 8), so 2 x this probe is a CEILING for the block on this kernel family, and
    the probe against 38.65 us (= 40 % of the dense bf16 peak per grid) says whether the 40 % target is within the family's reach at C = 32."""
    from resunet_a_mltsk_keras_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N, HW, Cc, dils = batch, 256, 32, [1, 3, 15, 31]
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((N, HW, HW, Cc), generator=g).to(dev).to(torch.bfloat16)
    ws = [(torch.randn((9, Cc, Cc), generator=g) / (3 * Cc ** 0.5)).to(dev).to(torch.bfloat16) for _ in dils]
    ys = [torch.empty((N, HW, HW, Cc), device=dev, dtype=torch.bfloat16) for _ in dils]
    arr = (L.ConvDesc * len(dils))()
    for b, dl in enumerate(dils):
        d = arr[b]
        d.nseg = 1
        sg = d.seg[0]
        sg.x, sg.w, sg.C, sg.Hs, sg.Ws, sg.up_shift, sg.dil, sg.taps = x.data_ptr(), ws[b].data_ptr(), Cc, HW, HW, 0, dl, 9
        d.N, d.H, d.W, d.Cout, d.stride, d.dtype = N, HW, HW, Cc, 1, L.RUA_BF16
        d.y, d.out_stride, d.OH, d.OW = ys[b].data_ptr(), 1, HW, HW
    ev_new, ev_rec, ev_us, ev_del = (lib.raw("rua_prof_event_" + n) for n in ("create", "record", "elapsed_us", "destroy"))
    for _ in range(3):
        lib.call("rua_conv_fwd_group", arr, len(dils), s)
    grids = lib.raw("rua_conv_group_last_grids")()
    a, b_ = C.c_void_p(ev_new()), C.c_void_p(ev_new())
    ev_rec(a, s)
    for _ in range(reps):
        lib.call("rua_conv_fwd_group", arr, len(dils), s)
    ev_rec(b_, s)
    v = C.c_double()
    lib.check(ev_us(a, b_, C.byref(v)), "rua_prof_event_elapsed_us")
    ev_del(a); ev_del(b_)
    us = v.value / reps
    gflop = len(dils) * 2.0 * N * HW * HW * Cc * Cc * 9 / 1e9
    return {"probe_us_per_grid": round(us, 1), "gflop_per_grid": round(gflop, 2), "grids_per_launch": int(grids),
            "what": "plain conv_strip32s group: row DMA + MFMA + output store, no BatchNorm / bias / statistics / epilogue stream; back to back, operands warm"}


def host_cores():
    """CPU share actually available: affinity mask capped by the cgroup quota (os.cpu_count() reports the host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(sample_steps=6):
    """CPU oracle (PyTorch-CPU fp32 restatement of the Keras graph) on BASELINE config 1:
    256x256x3, 6 classes, single task, bs 4, weighted CE, Adam - full train steps on all host cores.
    BASELINE.md section 2 asks for 10 / >= 50 / median of 3; at ~3.5 s per step that is impractical inside the default
    run, so this is a bounded sample (one warm-up, `sample_steps` timed steps) and says so: min / median / max are reported."""
    from oracle import resuneta_ref as ref
    from resunet_a_mltsk_keras_amd.synthetic import make_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} host cores ...")
    cfg = ref.RefConfig(input_shape=(256, 256, 3), num_classes=6, multitasking=False)
    params, order = ref.init_params(cfg, 0)
    spec = ref.CompileSpec(loss="weighted_cross_entropy", class_weights=[1.0] * 6, lr=1e-3)
    tr = ref.RefTrainer(cfg, params, order, spec)
    x, y = make_batch(4, 256, 3, 6, False, seed=1234)
    t0 = time.time()
    tr.train_on_batch(x, y)                                    # warm-up
    log(f"cpu warm-up step {time.time() - t0:.1f} s")
    per = []
    for _ in range(sample_steps):
        t0 = time.time()
        tr.train_on_batch(x, y)
        per.append(time.time() - t0)
        log(f"cpu step {per[-1]:.2f} s")
    srt = sorted(per)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    return {"value": round(4 / med, 3), "unit": "patches/s", "cores": cores, "kind": "port",
            "step_s": {"min": round(srt[0], 3), "median": round(med, 3), "max": round(srt[-1], 3), "n": len(srt)},
            "value_range": [round(4 / srt[-1], 3), round(4 / srt[0], 3)],
            "sample": f"{sample_steps} full train steps (fwd+loss+bwd+Adam) of config 1 (256x256x3, 6 classes, bs 4, fp32) "
                      f"after 1 warm-up step; value = 4 patches / median step; PyTorch-CPU oracle standing in for Keras-CPU"}


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 outside a launcher: start the N ranks as a CHILD process (the form the driver
    itself uses), relay rank 0's JSON line, return the child's status.  Nothing here touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = os.environ.get("RUA_BENCH_CHILD") or os.path.abspath(__file__)      # RUA_BENCH_CHILD: the CPU test's stub rank program
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC only on this pool (RCCL needs it)
    log("starting " + " ".join(cmd))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for out in proc.stdout:                                      # rank 0 prints ONE JSON line on stdout; everything else is stderr
        out = out.rstrip("\n")
        if out.startswith("{") and line is None:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        log("the ranks finished without a result line")
        rc = 1
    return rc


def build_engine(workload, dtype, batch, rank, world, args):
    from resunet_a_mltsk_keras_amd import _lib as L
    from resunet_a_mltsk_keras_amd.engine import Engine, LossSpec, ModelConfig
    from resunet_a_mltsk_keras_amd.synthetic import make_batch
    patch, ch, ncls, mt, bs, _ = WORKLOADS[workload]
    B = batch or bs
    depth = DEPTH.get(workload, 6)
    eng = Engine(ModelConfig(input_shape=(patch, patch, ch), num_classes=ncls, multitasking=mt, depth=depth), dtype=dtype, seed=0)
    heads = ["seg", "bound", "dist", "color"]
    eng.compile(LossSpec(kind={h: L.LOSS_TANIMOTO for h in heads}, weight={h: 1.0 for h in heads}, optimizer="adam", lr=1e-3))
    if world > 1 or args.force_dp:
        from resunet_a_mltsk_keras_amd.dist import DataParallel
        DataParallel(eng, bucket_mb=args.bucket_mb, overlap=not args.no_overlap)
    x, y = make_batch(B, patch, ch, ncls, mt, seed=1234 + rank)
    return eng, x, y, B, depth


def also_run(workload, dtype, steps, args):
    """A short run of another BASELINE configuration on this GPU (the `also` block of the N = 1 line): same step, same
    protocol in small (3 warm-up steps incl. plan recording + graph capture, `steps` timed steps between device synchronises)."""
    import gc
    t_all = time.time()
    eng, x, y, B, depth = build_engine(workload, dtype, 0, 0, 1, args)
    eng.train_step(x, y, fetch=False)
    for _ in range(2):
        eng.train_step(None, None, fetch=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(None, None, fetch=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = eng._results(eng.graph(B, True))
    gfl = WORKLOADS[workload][5]
    peak = BF16_DENSE_PEAK_TFLOPS if dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
    out = {"workload": workload, "dtype": dtype, "per_gpu_batch": B, "steps": steps, "ms_per_step": round(1e3 * dt / steps, 3),
           "patches_per_s": round(B * steps / dt, 1), "model_tflops_per_s": round(B * steps / dt * gfl / 1e3, 1),
           "frac_of_peak": round(B * steps / dt * gfl / 1e3 / peak, 4), "loss_last_step": round(res[0], 5),
           "wall_s_incl_build": None}
    del eng, x, y
    gc.collect()
    torch.cuda.empty_cache()
    out["wall_s_incl_build"] = round(time.time() - t_all, 1)
    return out


def main():
    global torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps each; the median block is reported (SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short runs of the other BASELINE configurations (N = 1 line)")
    ap.add_argument("--also-steps", type=int, default=10)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=50.0)
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel step (RCCL group of one rank) on one GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))              # BEFORE anything imports torch.cuda state: the parent stays off the GPU

    import torch as _torch
    torch = _torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} inside a launcher of WORLD_SIZE={world}: the two must agree")
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    elif args.force_dp:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1,
                                device_id=torch.device("cuda", local))

    patch, ch, ncls, mt, bs, gflop_patch = WORKLOADS[args.workload]
    eng, x, y, B, depth = build_engine(args.workload, args.dtype, args.batch, rank, world, args)
    if os.environ.get("RUA_LANES"):                             # experiment: ResBlock branches on parallel graph branches
        eng.use_lanes = True
    t_build = time.time()
    eng.train_step(x, y, fetch=False)                           # builds the plan, uploads the resident batch
    torch.cuda.synchronize()
    g0 = eng.graph(B, True)
    if rank == 0:
        log(f"plan recorded in {time.time() - t_build:.1f} s: {len(g0.fwd.calls)} fwd + {len(g0.loss_plan.calls)} loss + "
            f"{len(g0.bwd.calls)} bwd launches, activations {g0.act_bytes / 2**30:.2f} GiB")
    for _ in range(max(args.warmup - 1, 0)):
        eng.train_step(None, None, fetch=False)
    torch.cuda.synchronize()
    if rank == 0:
        log("warm-up done")

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # SURVEY 8d protocol: 10 warm-up steps, then 3 timed blocks of >= 50 steps, each bracketed by barrier + device
    # synchronise on both sides and taken as the MAX over ranks; the MEDIAN block is the reported one.
    if eng.dist is not None:
        eng.dist.reducer.measure = True                         # event pairs around the wait for the buckets (dp.allreduce_exposed_ms)
    block_dt = []
    own_dt = []                                                 # this rank's own block times (dp.ms_per_step_per_rank)
    host_issue = 0.0                                            # host time spent INSIDE train_step (issue only: nothing in it waits for the GPU)
    for _ in range(max(args.blocks, 1)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.train_step(None, None, fetch=False)
        host_issue += time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        own_dt.append(dt)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        block_dt.append(dt)
    dt = sorted(block_dt)[len(block_dt) // 2]
    res = eng._results(eng.graph(B, True))
    value = B * world * args.steps / dt
    if rank == 0:
        log(f"{len(block_dt)} blocks of {args.steps} steps: {[round(1e3 * d / args.steps, 3) for d in block_dt]} ms/step; median block "
            f"{dt:.3f} s -> {value:.1f} patches/s")

    out = {
        "metric": json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"] if args.workload in ("cfg3", "cfg2")
                  else f"training patches/sec ({patch}x{patch}, {ch}-ch, bs={B}/GPU)",
        "value": round(value, 2), "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic", "h2d": "excluded, batch resident", "timed_blocks": len(block_dt),
        "block_ms_per_step": [round(1e3 * d / args.steps, 3) for d in block_dt],
        "config": {"workload": f"{args.workload}: ResUnet-a d{depth} {'multitask (seg+bound+dist+color) Tanimoto-dual' if mt else 'single-task seg Tanimoto-dual'}, "
                               f"{patch}x{patch}x{ch}, {ncls} classes, Adam, full train step (fwd+loss+bwd+allreduce+update)",
                   "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}", "loss_last_step": round(res[0], 5)},
        "model_tflops_per_s": round(value * gflop_patch / 1e3, 2),
        # host time per step inside train_step (whole-step HIP graph at N = 1; eager C-ABI launches + bucket hooks under data parallel).
        # An upper bound of the issue cost: once the HIP queue is full the calls wait for the GPU (back-pressure), nothing else in them does
        "host_in_train_step_ms_per_step": round(1e3 * host_issue / (args.steps * max(args.blocks, 1)), 3),
        "step_path": ("hip-graph (whole step)" if (eng.use_graph and eng.dist is None) else
                      ("hip-graph pieces + eager collectives" if (eng.use_graph and eng.dp_graph) else "eager launches + bucketed all-reduce overlapped with backward")),
    }
    if eng.dist is not None:
        # data-parallel diagnostics, so that the first multi-GPU record explains itself: ranks counted BY the communicator, bucket
        # sizes, the all-reduce time the backward did not hide (events around the launch stream's wait), every rank's own step time
        from resunet_a_mltsk_keras_amd.dist import dp_report
        torch.cuda.synchronize()
        eng.dist.reducer.measure = False
        out["dp"] = dp_report(eng.dist, 1e3 * sorted(own_dt)[len(own_dt) // 2] / args.steps, torch.device("cuda", torch.cuda.current_device()))
    # every rank runs the instrumented pass (its keep-busy steps contain the gradient all-reduces); rank 0 reports
    prof = profile_kernels(eng, eng.graph(B, True), args.dtype)
    if rank == 0:
        peak = BF16_DENSE_PEAK_TFLOPS if args.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
        ev_ov = prof.pop("_event_overhead_us")
        blocks = prof.pop("_blocks")
        entries = prof.pop("_entries")
        kbytes = prof.pop("_kbytes")
        dom = max((kv for kv in prof.items() if kv[1][2] > 0), key=lambda kv: kv[1][1])   # second launches (no FLOPs) are rows, not candidates
        kn, (n, sec, fl) = dom
        rp = committed_rocprof_avg(kn, args)
        out["roofline"] = {
            "bound": "mfma", "kernel": kn, "launches_per_step": n, "avg_launch_us": round(1e6 * sec / n, 2),
            "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3),
            "achieved": round(fl / sec / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(fl / sec / 1e12 / peak, 4),
            "traffic": measured_traffic(kn, args), "traffic_unit": "HBM bytes per launch (profiles/traffic.json: rocprofv3 PMC)",
            "clock": "HIP events on the launch stream around every launch, calibrated bracket overhead subtracted",
            "event_pair_overhead_us_subtracted": round(ev_ov, 2),
            "rocprof_avg_us_committed_profile": rp,
            # the same fraction on the committed profile's clock (rocprofv3 --kernel-trace --stats average of this kernel)
            "frac_rocprof_clock": round(fl / n / (rp * 1e-6) / 1e12 / peak, 4) if rp else None,
            "all_mfma_kernels": {k: {"launches": v[0], "ms_per_step": round(1e3 * v[1], 3), "tflops": round(v[2] / v[1] / 1e12, 2)}
                                 for k, v in sorted(prof.items())},
            "whole_step_frac_of_peak": round(value / world * gflop_patch / 1e3 / peak, 4),
        }
        # the largest row of the step REGARDLESS of FLOPs: every C-ABI entry timed the same way (an HBM-bound entry is priced
        # against the HBM roof with its algorithmic bytes = the tensors its descriptor names, each once)
        rows = {k: (v[0], v[1], kbytes.get(k, 0.0), v[2]) for k, v in prof.items()}           # the convolution kernels, per instantiation
        rows.update({k: tuple(v) for k, v in entries.items() if not k.startswith("rua_conv")})  # + every other C-ABI entry
        en, (eln, esec, ebytes, efl) = max(rows.items(), key=lambda kv: kv[1][1])
        out["roofline"]["dominant_any"] = {
            "entry": en, "launches_per_step": eln, "ms_per_step": round(1e3 * esec, 3), "avg_launch_us": round(1e6 * esec / eln, 2),
            "bound": "mfma" if efl > 0 else "hbm",
            "algorithmic_bytes_per_launch": round(ebytes / eln) if ebytes else None,
            "achieved_gb_s": round(ebytes / esec / 1e9, 1) if ebytes else None, "hbm_peak_gb_s": HBM_PEAK_GBS,
            "hbm_frac": round(ebytes / esec / 1e9 / HBM_PEAK_GBS, 4) if ebytes else None,
            "mfma_frac": round(efl / esec / 1e12 / peak, 4) if efl > 0 else None,
        }
        out["roofline"]["all_entries"] = {k: {"launches": v[0], "ms_per_step": round(1e3 * v[1], 3),
                                              "gb_s": round(v[2] / v[1] / 1e9, 1) if v[2] else None}
                                          for k, v in sorted(entries.items(), key=lambda kv: -kv[1][1])}
        # north-star sub-metric: the d6 residual atrous block = ResBlock(32,[1,3,15,31]) at full resolution (model2.py:102),
        # its 8 convolutions + BatchNorm passes, forward; and the same for every ResBlock of the network (SURVEY 7 hard-part 2:
        # per-level fractions), forward and backward.  FLOPs: conv (fwd), data + weight gradient (bwd), nothing else counted.
        lv = []
        for key in sorted(k for k in blocks if k.startswith("fwd:") and "ResBlock" in k):
            sec, fl, by = blocks[key]
            row = {"block": key[4:], "fwd_gflop": round(fl / 1e9, 2), "fwd_us": round(1e6 * sec, 1), "fwd_frac": round(fl / sec / 1e12 / peak, 4)}
            if "bwd:" + key[4:] in blocks:
                bsec, bfl, _ = blocks["bwd:" + key[4:]]
                row.update({"bwd_gflop": round(bfl / 1e9, 2), "bwd_us": round(1e6 * bsec, 1), "bwd_frac": round(bfl / bsec / 1e12 / peak, 4)})
            lv.append(row)
        top = next((r for r in lv if r["block"].startswith("enc1:")), None)
        if top is not None:
            sec, fl, by = blocks["fwd:" + top["block"]]
            out["roofline"]["d6_block"] = {"block": top["block"], "gflop": top["fwd_gflop"], "us": top["fwd_us"], "frac": top["fwd_frac"],
                                           "target_frac": 0.40, "pass": "forward, batch %d, every launch of the block (BN + 8 convs)" % B,
                                           # its real limiter: the tensor passes the training-mode BatchNorm barrier leaves
                                           "hbm": {"bound": "hbm", "tensor_pass_bytes": round(by), "achieved_gb_s": round(by / sec / 1e9, 1),
                                                   "peak_gb_s": HBM_PEAK_GBS, "frac": round(by / sec / 1e9 / HBM_PEAK_GBS, 4),
                                                   "note": "bytes = every tensor each launch of the block reads or writes, once per launch"}}
        try:                                                    # the stripped ceiling probe of the d6 block (VERDICT r4 next#2a)
            pr = d6_ceiling_probe(B)
            tgt = pr["gflop_per_grid"] / (0.40 * peak * 1e3) * 1e6                 # us per grid at 40 % of the peak
            pr["us_per_grid_at_40pct"] = round(tgt, 1)
            pr["ceiling_us"] = round(2 * pr["probe_us_per_grid"], 1)
            pr["ceiling_frac"] = round(2 * pr["gflop_per_grid"] * 1e9 / (pr["ceiling_us"] * 1e-6) / 1e12 / peak, 4)
            pr["target_40pct_within_reach_of_this_kernel_family"] = bool(pr["probe_us_per_grid"] <= tgt)
            if "d6_block" in out["roofline"]:
                out["roofline"]["d6_block"]["ceiling_probe"] = pr
            out["roofline"]["d6_block_ceiling_us"] = pr["ceiling_us"]
            out["roofline"]["d6_block_ceiling_frac"] = pr["ceiling_frac"]
            out["roofline"]["d6_block_40pct_reachable"] = pr["target_40pct_within_reach_of_this_kernel_family"]
        except Exception as exc:                                # diagnostics only: never lose the headline line
            log(f"d6 ceiling probe failed: {type(exc).__name__}: {exc}")
        out["roofline"]["resblocks"] = lv
        # every other composite of the step (stem, stride-2 convs, PSPPooling, upsample + combine, heads, losses): microseconds only
        comp = {}
        for key, (sec, fl, _) in blocks.items():
            pn, sc = key.split(":", 1)
            if "ResBlock" not in sc:
                comp.setdefault(sc, {})[pn + "_us"] = round(1e6 * sec, 1)
        out["roofline"]["other_composites"] = comp
        out["roofline"]["composites_total_us"] = round(1e6 * sum(v[0] for v in blocks.values()), 1)
        # the same sub-metrics as FLAT scalars (a record that keeps only the scalar keys of `roofline` still carries them)
        rf = out["roofline"]
        if top is not None:
            rf["d6_block_us"], rf["d6_block_frac"] = top["fwd_us"], top["fwd_frac"]
            rf["d6_block_bwd_us"], rf["d6_block_bwd_frac"] = top.get("bwd_us"), top.get("bwd_frac")
        fr = [r[k] for r in lv for k in ("fwd_frac", "bwd_frac") if k in r]
        if fr:
            rf["worst_resblock_frac"] = min(fr)
            rf["resblocks_ms"] = round(sum(r.get("fwd_us", 0.0) + r.get("bwd_us", 0.0) for r in lv) / 1e3, 3)
        for r in lv:                                            # per level: enc3_fwd_us, enc3_bwd_us, ... (levels 3 - 4 are this round's target)
            tag = r["block"].split(":", 1)[0]
            rf[tag + "_fwd_us"], rf[tag + "_bwd_us"] = r.get("fwd_us"), r.get("bwd_us")
        rf["glue_ms"] = round(sum(v for c in comp.values() for v in c.values()) / 1e3, 3)
        opt_rows = ("rua_adam_step", "rua_sgd_step", "rua_wgrad_reduce_batch", "rua_weight_prep")
        rf["optimizer_ms"] = round(1e3 * sum(v[1] for k, v in entries.items() if k in opt_rows), 3)
        rf["dispatches_per_step"] = eng.count_step_dispatches(B)
    if world == 1 and not args.force_dp and rank == 0:
        if not args.no_also and args.workload == "cfg3" and args.dtype == "bf16" and not args.batch:
            # the other BASELINE configurations, short runs on the same GPU right after the headline one
            del eng, x, y, g0
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            also = {}
            for name, wl, dt_ in (("cfg3_f32", "cfg3", "f32"), ("cfg2", "cfg2", "bf16"), ("cfg5", "cfg5", "bf16"), ("cfg4", "cfg4", "bf16")):
                try:
                    also[name] = also_run(wl, dt_, args.also_steps, args)
                    log(f"also {name}: {also[name]['ms_per_step']} ms/step, {also[name]['patches_per_s']} patches/s")
                except Exception as exc:                       # never lose the headline line to an extra
                    also[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                    log(f"also {name} failed: {also[name]['error']}")
            out["also"] = also
        if not args.no_cpu_baseline:                            # the CPU oracle is timed at N=1 only (the other ranks would idle)
            out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
