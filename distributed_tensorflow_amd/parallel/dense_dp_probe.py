"""Data-parallel training steps of a wide fully connected stack (``dense_trainer.DenseDataParallelTrainer``), as a
function for a process group at hand and as a process of its own::

    python -m distributed_tensorflow_amd.parallel.dense_dp_probe --rank R --world N --device D --port P [--backend gloo]

``bench.py`` runs it as one child per rank: the stage is an extra next to the reference's workload, and whatever happens
in it (a collective that hangs on the node at hand, a fault) must not cost the benchmark's line.  Prints one JSON line
(every rank; the caller reads its own rank's).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch


def _sync(device):
    import torch.distributed as dist

    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def measure(device, rank: int, world: int, agree) -> dict:
    """Synchronous data-parallel steps of a 784-4096-4096-10 stack (parallel/dense_trainer.py): 80 MB of gradients per
    step in buckets, each all-reduced on a side stream under the backward of the layers below.  Every rank calls this at
    the same place; the same steps without collectives say what the overlap has to hide.  Whatever can fail on one rank
    alone (allocation, data, construction) happens before the first collective and is followed by ``agree``: the ranks
    either all measure or all return the error."""
    import numpy as np
    import torch.distributed as dist

    from ..data import DeviceDataset
    from .dense_trainer import DenseDataParallelTrainer

    dims, batch, steps, warm = (784, 4096, 4096, 10), int(os.environ.get("DTF_BENCH_DENSE_DP_BATCH", "4096")), 16, 3
    if device.type == "cpu":        # CPU rehearsal of the flow (gloo): a stack small enough for a test
        dims, batch, steps, warm = (784, 64, 64, 10), 32, 4, 1
    trainers, err = {}, None
    try:
        if os.environ.get("DTF_BENCH_TEST_FAIL") == f"dense_dp:{rank}":       # test hook: this rank cannot set up
            raise MemoryError("injected by DTF_BENCH_TEST_FAIL")
        rng = np.random.RandomState(100 + rank)
        rows = 4 * batch
        images = rng.rand(rows, dims[0]).astype(np.float32)
        onehot = np.eye(dims[-1], dtype=np.float32)[rng.randint(0, dims[-1], rows)]
        ds = DeviceDataset(images, onehot, device, batch, seed=rank)
        for tag, w in (("with_collectives", world), ("compute_only", 1)):
            trainers[tag] = DenseDataParallelTrainer(dims, ["relu"] * (len(dims) - 2), ds, batch, 0.01, device, rank, w, seed=1)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not agree(err is None):
        return {"error": err or "set-up failed on another rank"}
    out = {"model": "-".join(map(str, dims)), "batch_per_rank": batch, "ranks": world, "steps": steps, "warmup": warm,
           "dtype": "fp32", "note": "not the reference's workload: its update rule on a model whose gradients are megabytes"}
    for tag in ("with_collectives", "compute_only"):
        tr = trainers.pop(tag)
        tr.run_steps(warm)
        _sync(device)
        t0 = time.perf_counter()
        tr.run_steps(steps)
        _sync(device)
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dt = float(dt.item())
        out[tag] = {"ms_per_step": round(1000.0 * dt / steps, 4), "samples_per_s": round(world * batch * steps / dt, 1)}
        if tag == "with_collectives":
            out["parallelism"] = tr.describe()
            out["global_step"] = tr.global_step_value()
            ck = torch.tensor([tr.replica_checksum()], dtype=torch.float64, device=device)
            lo, hi = ck.clone(), ck.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            out["replicas_identical"] = bool(float(lo.item()) == float(hi.item()))
            out["last_loss"] = round(tr.last_loss(), 4)
        del tr
    out["exposed_collective_ms_per_step"] = round(out["with_collectives"]["ms_per_step"] - out["compute_only"]["ms_per_step"], 4)
    # what the links give a collective of a bucket's size on its own (bus bandwidth of a ring: 2 (N - 1) / N of the bytes)
    if world > 1:
        floats = (16 << 20) if dist.get_backend() == "nccl" else (1 << 20)
        buf = torch.ones(floats, dtype=torch.float32, device=device)
        for _ in range(2):
            dist.all_reduce(buf)
        _sync(device)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            dist.all_reduce(buf)
        _sync(device)
        dt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        sec = float(dt.item())
        out["allreduce_alone"] = {"megabytes": floats * 4 >> 20, "ms": round(1000.0 * sec, 4),
                                  "bus_GB_per_s": round(2.0 * (world - 1) / world * floats * 4 / sec / 1e9, 1),
                                  "backend": dist.get_backend()}
    return out


def measure_small_batch(device, rank: int, world: int, agree) -> dict:
    """The same trainer where a step is launch-bound (784-1024-1024-10, batch 100 per rank): eager launches against
    hipGraph chunks of 8 steps whose collectives sit on a branch of the graph."""
    import numpy as np
    import torch.distributed as dist

    from ..data import DeviceDataset
    from .dense_trainer import DenseDataParallelTrainer

    dims, batch, steps, warm = (784, 1024, 1024, 10), 100, 160, 16
    if device.type == "cpu":
        dims, batch, steps, warm = (784, 32, 32, 10), 20, 8, 4
    elif dist.get_backend() == "gloo":      # (rehearsal with ranks sharing a GPU: reductions through the host, no graphs)
        steps, warm = 16, 4
    trainers, err = {}, None
    try:
        rng = np.random.RandomState(200 + rank)
        rows = 16 * batch
        images = rng.rand(rows, dims[0]).astype(np.float32)
        onehot = np.eye(dims[-1], dtype=np.float32)[rng.randint(0, dims[-1], rows)]
        ds = DeviceDataset(images, onehot, device, batch, seed=rank)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not agree(err is None):
        return {"error": err or "set-up failed on another rank"}
    out = {"model": "-".join(map(str, dims)), "batch_per_rank": batch, "steps": steps}
    for tag, graph in (("eager", False), ("graph_chunks", True)):
        # (constructed by all ranks at the same place: with graphs the constructor issues one small collective)
        tr = DenseDataParallelTrainer(dims, ["relu"] * (len(dims) - 2), ds, batch, 0.01, device, rank, world, seed=1,
                                      bucket_mb=1.0, use_graph=graph, chunk=8)
        tr.run_steps(warm)
        _sync(device)
        t0 = time.perf_counter()
        tr.run_steps(steps)
        _sync(device)
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        sec = float(dt.item())
        ck = torch.tensor([tr.replica_checksum()], dtype=torch.float64, device=device)
        lo, hi = ck.clone(), ck.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out[tag] = {"ms_per_step": round(1000.0 * sec / steps, 4), "samples_per_s": round(world * batch * steps / sec, 1),
                    "replicas_identical": bool(float(lo.item()) == float(hi.item())), "graphs_used": bool(tr.use_graph)}
        tr.close()
        del tr
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args(argv)
    import datetime

    import torch.distributed as dist

    out = {}
    try:
        dev = torch.device("cuda", args.device)
        torch.cuda.set_device(dev)
        dist.init_process_group(args.backend, rank=args.rank, world_size=args.world,
                                device_id=dev if args.backend == "nccl" else None,
                                init_method=f"tcp://127.0.0.1:{args.port}", timeout=datetime.timedelta(seconds=120))

        def agree(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(int(flag.item()))

        out = measure(dev, args.rank, args.world, agree)
        print(json.dumps(out), flush=True)          # (the caller reads the LAST line: what follows may not get that far)
        try:
            out["small_batch"] = measure_small_batch(dev, args.rank, args.world, agree)
        except Exception as e:  # noqa: BLE001
            out["small_batch"] = {"error": f"{type(e).__name__}: {e}"}
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 — every failure is an answer
        if "with_collectives" in out:      # the measurement is complete: a late failure (the last barrier,
            out = dict(out, late_error=f"{type(e).__name__}: {e}")     # tearing down the group) must not cost it
        else:
            out = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
    return 0 if "error" not in out else 3


if __name__ == "__main__":
    sys.exit(main())
