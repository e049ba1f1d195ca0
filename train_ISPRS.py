#!/usr/bin/env python
"""Training CLI of the ResUnet-a path on MI355X — same flags, defaults and results-directory roles as the
reference's train_ISPRS.py (reference train_ISPRS.py:297-338 flags, :55-292 epoch loop, :354-380 dataset listing).

Deviations, all additive or forced by the reference's hard-coded values:
  * `--channels` (default: read from the first patch) replaces the hard-coded `channels = 3` (reference :401) and
    the label buffers take their class count from `--num_classes` instead of the hard-coded 5 (reference :491).
  * patches and labels are paired BY FILE NAME; the reference relies on os.listdir returning the same order in
    every directory (reference :354-374).
  * the default weighted-CE class weights are the reference's five values (reference :424) only when
    --num_classes is 5; otherwise uniform weights.
  * scalars go to `<results_path>/logs/{train,val}/scalars.jsonl` with the reference's TensorBoard tag names
    (tensorboard is not available); the best model is `<results_path>/best_model.h5`: real HDF5 whose `model_weights` group is
    Keras' own weight layout (plus this package's metadata and optimizer state, written without h5py by h5lite).
  * `--dtype {bf16,f32}`, `--seed`: engine options.  Launch with torch.distributed.run for multi-GPU data
    parallel (`-bs` is then the GLOBAL batch, as under MirroredStrategy).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE_WCE_WEIGHTS = [4.34558461, 2.97682037, 3.92124661, 5.67350328, 374.0300152]   # reference train_ISPRS.py:424
TASKS = [("seg", "Segmentation"), ("bound", "Boundary"), ("dist", "Distance"), ("color", "Color")]


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = str(v).lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--resunet_a", help="choose resunet-a model or not", type=str2bool, default=False)
    p.add_argument("--multitasking", help="choose resunet-a multitasking or not", type=str2bool, default=False)
    p.add_argument("--gpu_parallel", help="choose 1 to train one multiple gpu", type=str2bool, default=False)
    p.add_argument("-rp", "--results_path", type=str, default="./results/results_run1",
                   help="Path where to save logs and model checkpoint.")
    p.add_argument("-cp", "--checkpoint_path", type=str, default=None, help="Path where to load model checkpoint to continue training")
    p.add_argument("-dp", "--dataset_path", type=str, default="./DATASETS/patch_size=256_stride=32", help="Path where to load dataset")
    p.add_argument("-bs", "--batch_size", type=int, default=4, help="Batch size on training")
    p.add_argument("-lr", "--learning_rate", type=float, default=1e-3, help="Learning rate on training")
    p.add_argument("--loss", type=str, default="weighted_cross_entropy", choices=["weighted_cross_entropy", "cross_entropy", "tanimoto"])
    p.add_argument("-optm", "--optimizer", type=str, choices=["adam", "sgd"], default="adam")
    p.add_argument("--num_classes", type=int, default=5)
    p.add_argument("--epochs", type=int, default=500)
    p.add_argument("-ps", "--patch_size", type=int, default=256)
    p.add_argument("--bound_weight", type=float, default=1.0)
    p.add_argument("--dist_weight", type=float, default=1.0)
    p.add_argument("--color_weight", type=float, default=1.0)
    # additive
    p.add_argument("--channels", type=int, default=0, help="input bands (0 = read from the first patch)")
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "f32"])
    p.add_argument("--seed", type=int, default=0)
    return p


def compute_mcc(tp, tn, fp, fn):
    den = math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
    return (tp * tn - fp * fn) / den if den > 0 else float("nan")


def list_dataset(root, multitasking):
    """Name-paired patch paths: <root>/train/x.npy with <root>/labels/{seg,bound,dist,color}/x.npy."""
    tdir = os.path.join(root, "train")
    names = sorted(n for n in os.listdir(tdir) if n.endswith(".npy"))
    heads = ["seg", "bound", "dist", "color"] if multitasking else ["seg"]
    for h in heads:
        have = set(os.listdir(os.path.join(root, "labels", h)))
        missing = [n for n in names if n not in have]
        if missing:
            raise FileNotFoundError(f"labels/{h} lacks {len(missing)} patches, e.g. {missing[0]}")
    xs = [os.path.join(tdir, n) for n in names]
    ys = {h: [os.path.join(root, "labels", h, n) for n in names] for h in heads}
    return xs, ys


def split_dataset(xs, ys):
    """train_test_split(test_size=0.2, random_state=42) applied to every list jointly (reference :377-380)."""
    from sklearn.model_selection import train_test_split
    heads = list(ys)
    parts = train_test_split(xs, *[ys[h] for h in heads], test_size=0.2, random_state=42)
    x_tr, x_va = parts[0], parts[1]
    y_tr = {h: parts[2 + 2 * i] for i, h in enumerate(heads)}
    y_va = {h: parts[3 + 2 * i] for i, h in enumerate(heads)}
    return x_tr, y_tr, x_va, y_va


class ScalarLog:
    def __init__(self, path):
        os.makedirs(path, exist_ok=True)
        self.f = open(os.path.join(path, "scalars.jsonl"), "a")

    def scalar(self, tag, value, step):
        self.f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        self.f.flush()



def agree_from_rank0(value, world):
    """Rank 0's value on every rank.  The metrics are already replica-aggregated inside train_on_batch / test_on_batch
    (identical on all ranks); the stop / save decision still goes through one broadcast per epoch, so that no rank can
    ever leave the epoch loop while the others wait in the next gradient all-reduce (reference :280-292 takes ONE
    decision for all replicas)."""
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, 0)
    return float(t.item())


def train_model(args, net, x_tr, y_tr, x_va, y_va, batch_size, epochs, x_shape, n_classes, patience=10, delta=0.001,
                metrics_names=None, rank=0, world=1):
    say = print if rank == 0 else (lambda *a, **k: None)
    say("Start training...\n" + "=" * 60)
    say(f"Training on {len(x_tr)} images\nValidating on {len(x_va)} images\n" + "=" * 60 + f"\nTotal Epochs: {epochs}")
    tw = ScalarLog(os.path.join(args.results_path, "logs", "train")) if rank == 0 else None
    vw = ScalarLog(os.path.join(args.results_path, "logs", "val")) if rank == 0 else None
    from resunet_a_mltsk_keras_amd.loader import PrefetchLoader
    # the reference loads 5*B .npy files serially before every step (train_ISPRS.py:115-141); here worker threads read
    # two batches ahead into pinned buffers
    # under data parallel `batch_size` is the global batch and every rank reads only its own shard of it
    ld_tr = PrefetchLoader(x_tr, y_tr, batch_size, rank=rank, world=world)
    ld_va = PrefetchLoader(x_va, y_va, batch_size, rank=rank, world=world)
    shard = dict(local_shard=True) if world > 1 else {}
    min_loss, cont = float("inf"), 0
    rng = np.random.default_rng(args.seed)
    say(net.output_names)
    for epoch in range(epochs):
        acc_tr = np.zeros(len(metrics_names)); acc_va = np.zeros(len(metrics_names))
        ld_tr.set_order(rng.permutation(len(x_tr)))
        n_tr, n_va = len(ld_tr), len(ld_va)
        for xb, yb in ld_tr:
            acc_tr += np.asarray(net.train_on_batch(x=xb, y=yb if args.multitasking else yb["seg"], return_dict=False, **shard))
        acc_tr /= max(n_tr, 1)
        for xb, yb in ld_va:
            acc_va += np.asarray(net.test_on_batch(x=xb, y=yb if args.multitasking else yb["seg"], **shard))
        acc_va /= max(n_va, 1)
        tm, vm = dict(zip(metrics_names, acc_tr)), dict(zip(metrics_names, acc_va))
        pre = "seg_" if args.multitasking else ""
        mcc = compute_mcc(vm[pre + "true_positives"], vm[pre + "true_negatives"], vm[pre + "false_positives"], vm[pre + "false_negatives"])
        if rank == 0:
            if not args.multitasking:
                print(f"Epoch: {epoch} Training loss: {tm['loss']:.5f} Train acc.: {100 * tm['accuracy']:.5f}% "
                      f"Validation loss: {vm['loss']:.5f} Validation acc.: {100 * vm['accuracy']:.5f}%")
                tw.scalar("Total/Loss", tm["loss"], epoch); tw.scalar("Total/Accuracy", tm["accuracy"], epoch)
                vw.scalar("Total/Loss", vm["loss"], epoch); vw.scalar("Total/Accuracy", vm["accuracy"], epoch); vw.scalar("Total/MCC", mcc, epoch)
            else:
                print(f"+{'-' * 62}+\n| Epoch: {epoch:<53d}|\n| {'Task':8s}{'Loss':>13s}{'Val Loss':>13s}{'Acc %':>13s}{'Val Acc %':>13s} |")
                for key, tag in TASKS:
                    a = (100 * tm["seg_accuracy"], 100 * vm["seg_accuracy"]) if key == "seg" else (0, 0)
                    print(f"| {key.capitalize():8s}{tm[key + '_loss']:13.5f}{vm[key + '_loss']:13.5f}{a[0]:13.5f}{a[1]:13.5f} |")
                    tw.scalar(tag + "/Loss", tm[key + "_loss"], epoch); vw.scalar(tag + "/Loss", vm[key + "_loss"], epoch)
                tw.scalar("Segmentation/Accuracy", tm["seg_accuracy"], epoch); vw.scalar("Segmentation/Accuracy", vm["seg_accuracy"], epoch)
                vw.scalar("Segmentation/MCC", mcc, epoch)
                print(f"| {'Total':8s}{tm['loss']:13.5f}{vm['loss']:13.5f}{0:13d}{0:13d} |\n+{'-' * 62}+")
                tw.scalar("Total/Loss", tm["loss"], epoch); vw.scalar("Total/Loss", vm["loss"], epoch)
        val_loss = agree_from_rank0(vm["loss"], world)
        if val_loss >= min_loss + delta:
            cont += 1
            say(f"EarlyStopping counter: {cont} out of {patience}")
            if cont >= patience:
                say("Early Stopping! \t Training Stopped")
                return net
        else:
            cont, min_loss = 0, val_loss
            say("Saving best model...")
            if rank == 0:
                net.save(os.path.join(args.results_path, "best_model.h5"))
    return None            # like the reference: the model is returned only on early stop (reference :287)


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    say = print if rank == 0 else (lambda *a, **k: None)
    say("=" * 30 + "INITIALIZING" + "=" * 30)
    say(f"GPUS DEVICES: {[torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())]}")
    say(f"Number of devices: {world}")
    if not args.resunet_a:
        sys.exit("--resunet_a False selects the reference's baseline U-Net, which is outside the accelerated path")

    from ResUnet_a.model2 import Resunet_a
    from multitasking_utils import Tanimoto_dual_loss
    from resunet_a_mltsk_keras_amd.keras_api import (SGD, Adam, BinaryCrossentropy, CategoricalCrossentropy, K, MeanSquaredError,
                                                   load_model, weighted_categorical_crossentropy)

    xs, ys = list_dataset(args.dataset_path, args.multitasking)
    x_tr, y_tr, x_va, y_va = split_dataset(xs, ys)
    rows = cols = args.patch_size
    channels = args.channels or int(np.load(xs[0]).shape[-1])
    optm = Adam(lr=args.learning_rate, beta_1=0.9) if args.optimizer == "adam" else SGD(lr=args.learning_rate, momentum=0.8)
    say("=" * 60)
    if args.loss == "cross_entropy":
        say("Using Cross Entropy")
        loss, loss_bound, loss_reg = CategoricalCrossentropy(), BinaryCrossentropy(), MeanSquaredError()
    elif args.loss == "tanimoto":
        say("Using Tanimoto Dual Loss")
        loss = loss_bound = loss_reg = Tanimoto_dual_loss()
    else:
        say("Using Weighted cross entropy")
        weights = REFERENCE_WCE_WEIGHTS if args.num_classes == 5 else [1.0] * args.num_classes
        say(weights)
        loss, loss_bound, loss_reg = weighted_categorical_crossentropy(weights), BinaryCrossentropy(), MeanSquaredError()
    say("=" * 60)

    if args.checkpoint_path is None:
        resuneta = Resunet_a((rows, cols, channels), args.num_classes, args)
        model = resuneta.model
        if rank == 0:
            model.summary()
        if args.multitasking:
            say("Multitasking enabled!")
            lw = {"seg": 1.0, "bound": args.bound_weight, "dist": args.dist_weight, "color": args.color_weight}
            say(f"Loss Weights: {lw}")
            model.compile(optimizer=optm, loss={"seg": loss, "bound": loss_bound, "dist": loss_reg, "color": loss_reg},
                          loss_weights=lw, metrics={"seg": ["accuracy"]})
        else:
            say("Using simple ResUnet-a")
            model.compile(optimizer=optm, loss=loss, metrics=["accuracy"])
        say("ResUnet-a compiled!")
    else:
        say(f"[INFO] loading {args.checkpoint_path}...")
        model = load_model(args.checkpoint_path)
        say(f"[INFO] old learning rate: {K.get_value(model.optimizer.lr)}")
        K.set_value(model.optimizer.lr, args.learning_rate)
        say(f"[INFO] new learning rate: {K.get_value(model.optimizer.lr)}")

    if rank == 0:
        os.makedirs(args.results_path, exist_ok=True)
    if world > 1 and args.batch_size % world:
        sys.exit(f"-bs {args.batch_size} is the GLOBAL batch and must divide by the {world} replicas")
    x_shape = (args.batch_size, rows, cols, channels)
    t0 = time.time()
    train_model(args, model, x_tr, y_tr, x_va, y_va, args.batch_size, args.epochs, x_shape, args.num_classes,
                metrics_names=model.metrics_names, rank=rank, world=world)
    say(f"\nTraining took: {(time.time() - t0) / 3600} \n")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
