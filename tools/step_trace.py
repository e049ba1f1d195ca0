"""Every launch of one cfg3 train step, in plan order, with its composite and its duration (HIP events around each C-ABI call on
a busy stream, bracket overhead subtracted): where the step's time goes outside the ResBlocks.
usage (GPU box): python tools/step_trace.py [--scope psp_mid] [--min-us 0] > gpurun_out/step_trace.txt"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resunet_a_mltsk_keras_amd import _lib as L                     # noqa: E402
from resunet_a_mltsk_keras_amd.engine import Engine, LossSpec, ModelConfig   # noqa: E402
from resunet_a_mltsk_keras_amd.synthetic import make_batch          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scope", default="", help="only composites whose name contains this")
    ap.add_argument("--min-us", type=float, default=0.0)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    eng = Engine(ModelConfig(input_shape=(256, 256, 6), num_classes=6, multitasking=True, depth=6), dtype="bf16", seed=0)
    heads = ["seg", "bound", "dist", "color"]
    eng.compile(LossSpec(kind={h: L.LOSS_TANIMOTO for h in heads}, weight={h: 1.0 for h in heads}, optimizer="adam", lr=1e-3))
    x, y = make_batch(args.batch, 256, 6, 6, True, seed=1234)
    for _ in range(3):
        eng.train_step(x, y, fetch=False)
    torch.cuda.synchronize()
    g = eng.graph(args.batch, True)
    lib = L.lib()
    s = torch.cuda.current_stream().cuda_stream
    sp = C.c_void_p(s)
    ev_new, ev_rec, ev_us = (lib.raw("rua_prof_event_" + n) for n in ("create", "record", "elapsed_us"))

    def mark():
        e = C.c_void_p(ev_new())
        ev_rec(e, sp)
        return e

    def us(a, b):
        v = C.c_double()
        lib.check(ev_us(a, b, C.byref(v)), "elapsed")
        return v.value
    for _ in range(4):                                              # keep the queue full while the instrumented step is enqueued
        eng.train_step(None, None, fetch=False)
    cal1, cal2 = [], []
    for _ in range(12):
        a = mark(); eng._zero_arena(g, s); b = mark(); cal1.append((a, b))
        a = mark(); eng._zero_arena(g, s); eng._zero_arena(g, s); b = mark(); cal2.append((a, b))
    eng._zero_arena(g, s)
    eng._prep_weights(s)
    rows = []
    for pname, plan in (("fwd", g.fwd), ("loss", g.loss_plan), ("bwd", g.bwd)):
        for ci, (fn, name, cargs, _lane) in enumerate(plan.calls):
            if fn is None:
                continue
            e0 = mark()
            rc = fn(*cargs, sp)
            e1 = mark()
            if rc != 0:
                lib.check(rc, name)
            rows.append((pname, plan.scopes[ci] or "-", name, e0, e1))
    eng.optimizer_step(1.0)
    torch.cuda.synchronize()
    med = lambda pairs: sorted(us(a, b) for a, b in pairs)[len(pairs) // 2]
    ov = max(2 * med(cal1) - med(cal2), 0.0)
    tot, per_scope = 0.0, {}
    for pname, sc, name, e0, e1 in rows:
        t = max(us(e0, e1) - ov, 0.0)
        tot += t
        k = (pname, sc)
        per_scope.setdefault(k, [0, 0.0])
        per_scope[k][0] += 1; per_scope[k][1] += t
        if args.scope in sc and t >= args.min_us:
            print(f"{pname:4s} {sc:44s} {name:28s} {t:8.1f} us")
    print(f"# bracket overhead subtracted {ov:.2f} us; sum of launches {tot / 1e3:.3f} ms over {len(rows)} launches")
    for (pname, sc), (n, t) in sorted(per_scope.items(), key=lambda kv: -kv[1][1]):
        print(f"# {pname:4s} {sc:44s} {n:4d} launches {t:9.1f} us")


if __name__ == "__main__":
    main()
