"""What corrupted the piecewise-graph data-parallel step (RUA_DP_GRAPH=1), reproduced: one
rank, RCCL group of one (the all-reduces are then nothing but their stream events), TRIALS runs of STEPS steps each from the same
seed; a run counts as corrupt when its parameters contain a NaN / inf or differ from the eager data-parallel run of the same steps
by more than bf16 noise.  Modes (DPC_MODES, comma separated):
  memset         rua_fill_zero = hipMemsetAsync (memset nodes in the captured pieces), nothing between the replays  -> corrupt on SOME
                 boxes of the pool, then in every run of the process; clean on others
  memset+fence   the same with one eager kernel behind every replay (the round-1/2 workaround)                       -> clean
  sync           the same with a device synchronise behind every replay instead                                     -> still corrupt
  fillk          rua_fill_zero as a kernel (the default since round 3), nothing between the replays                 -> clean
Usage: python tools/dp_graph_check.py [trials] [steps]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(mode, fence, steps, x, y):
    from resunet_a_mltsk_keras_amd import _lib as L
    from resunet_a_mltsk_keras_amd.dist import DataParallel
    from resunet_a_mltsk_keras_amd.engine import Engine, LossSpec, ModelConfig
    os.environ["RUA_DP_GRAPH"] = "1" if mode == "pieces" else "0"
    os.environ["RUA_DP_FENCE"] = "1" if fence else "0"
    heads = ["seg", "bound", "dist", "color"]
    eng = Engine(ModelConfig(input_shape=(128, 128, 6), num_classes=6, multitasking=True), dtype="bf16", seed=0)
    eng.compile(LossSpec(kind={h: L.LOSS_TANIMOTO for h in heads}, weight={h: 1.0 for h in heads}, optimizer="sgd", lr=1e-2))
    DataParallel(eng, bucket_mb=25.0, overlap=True)
    losses = []
    for _ in range(steps):
        losses.append(eng.train_step(x, y)[0])
    torch.cuda.synchronize()
    P = eng.P.cpu().numpy().copy()
    del eng
    torch.cuda.empty_cache()
    return np.array(losses), P


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29581", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from resunet_a_mltsk_keras_amd.synthetic import make_batch
    x, y = make_batch(4, 128, 6, 6, True, seed=3)
    l0, p0 = run("eager", True, steps, x, y)
    print("eager data-parallel reference: losses", np.round(l0, 4), flush=True)
    for fence in [{"1": True, "0": False}.get(v, v) for v in os.environ.get("DPC_MODES", "memset,memset+fence,sync,fillk").split(",")]:
        bad = 0
        from resunet_a_mltsk_keras_amd import _lib as L
        L.lib().set_tuning(fill_kernel=0 if fence in ("memset", "memset+fence", "sync") else 1)
        os.environ["RUA_DP_SYNC"] = "1" if fence == "sync" else "0"
        label = fence
        if fence in ("fillk", "sync", "memset"):
            fence = False
        elif fence == "memset+fence":
            fence = True
        for t in range(trials):
            l, p = run("pieces", fence, steps, x, y)
            finite = np.isfinite(p).all() and np.isfinite(l).all()
            dev = float(np.abs(p - p0).max()) if finite else float("inf")
            ok = finite and dev < 5e-2 and abs(l[-1] - l0[-1]) < 5e-2
            bad += not ok
            print(f"  pieces fence={label} trial {t}: finite={finite} max |dP|={dev:.3e} last loss {l[-1]:.4f} {'ok' if ok else 'CORRUPT'}", flush=True)
        print(f"RESULT fence={label}: {bad} of {trials} runs corrupt", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
