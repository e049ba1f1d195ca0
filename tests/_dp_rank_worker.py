"""One rank of the 2-process data-parallel Engine test (tests/test_api_gpu.py::test_two_process_data_parallel_engine).
Both ranks share cuda:0 of the one-GPU test box; the collectives go through gloo (host-staged, dist.py) so that two
processes on one device can form a group.  argv: rank world port out_dir [resume_checkpoint]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    resume = sys.argv[5] if len(sys.argv) > 5 else None
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from resunet_a_mltsk_keras_amd import keras_api as ka
    from resunet_a_mltsk_keras_amd.engine import HEADS, ModelConfig
    from resunet_a_mltsk_keras_amd.synthetic import make_batch
    C = 4
    x, y = make_batch(4, 64, 3, C, True, seed=21, block=16)               # the GLOBAL batch, identical on every rank
    if resume is None:
        m = ka.Model(ModelConfig(input_shape=(64, 64, 3), num_classes=C, multitasking=True), dtype="f32", seed=11 + rank)
        m.engine.split_k = False                                            # bit-reproducible convolutions (tiny BN batches)
        loss = ka.Tanimoto_dual_loss()
        m.compile(optimizer=ka.SGD(lr=0.05, momentum=0.8), loss={h: loss for h in HEADS}, loss_weights={h: 1.0 for h in HEADS})
    else:
        m = ka.load_model(resume)                                           # resume must attach data parallel too
        m.engine.split_k = False
    assert m.engine.dist is not None and m.engine.world == world
    res = [m.train_on_batch(x, y) for _ in range(2)]                        # Model slices the global batch per rank
    ev = m.test_on_batch(x, y)
    torch.cuda.synchronize()
    np.savez(os.path.join(out, f"rank{rank}.npz"), res=np.asarray(res), ev=np.asarray(ev), P=m.engine.P.cpu().numpy(),
             S=m.engine.S.cpu().numpy(), M1=m.engine.M1.cpu().numpy(), t=m.engine.t)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
