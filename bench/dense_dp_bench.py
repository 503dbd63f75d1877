"""Data-parallel training steps of a wide fully connected stack (parallel/dense_trainer.py::DenseDataParallelTrainer):
gradient buckets all-reduced on a side stream under the backward of the layers below.

    python bench/dense_dp_bench.py                                                        # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 bench/dense_dp_bench.py --bucket_mb 32

Prints one JSON line on rank 0: samples/s over all ranks, ms per step, the buckets, and the same steps with the
collectives left out (what the overlap has to hide)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import DeviceDataset  # noqa: E402
from distributed_tensorflow_amd.parallel.dense_trainer import DenseDataParallelTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--widths", default="784,4096,4096,10")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--bucket_mb", type=float, default=32.0)
    ap.add_argument("--graph", type=int, default=0, help="1: hipGraph chunks of 8 steps (collectives on a branch of the graph)")
    ap.add_argument("--force_collectives", type=int, default=0,
                    help="1: a lone rank still issues its all-reduces through a one-rank RCCL communicator")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    elif a.force_collectives:
        dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29871", device_id=device)
    dims = [int(v) for v in a.widths.split(",")]
    rng = np.random.RandomState(rank)
    n = 8 * a.batch
    labels = rng.randint(0, dims[-1], n)
    images = rng.rand(n, dims[0]).astype(np.float32)
    onehot = np.eye(dims[-1], dtype=np.float32)[labels]
    ds = DeviceDataset(images, onehot, device, a.batch, seed=rank)
    out = {}
    for tag, w in (("with_collectives", world), ("compute_only", 1)):
        group = dist.group.WORLD if (a.force_collectives and world == 1 and tag == "with_collectives") else None
        tr = DenseDataParallelTrainer(dims, ["relu"] * (len(dims) - 2), ds, a.batch, 0.01, device, rank, w, seed=1,
                                      bucket_mb=a.bucket_mb, group=group, use_graph=bool(a.graph))
        tr.run_steps(16)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        tr.run_steps(a.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        out[tag] = {"ms_per_step": round(1000 * dt / a.steps, 4), "samples_per_s": round(world * a.batch * a.steps / dt, 1)}
        desc = tr.describe() if tag == "with_collectives" else desc
    if rank == 0:
        print(json.dumps({"widths": dims, "batch_per_rank": a.batch, "ranks": world, **out, "parallelism": desc}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
