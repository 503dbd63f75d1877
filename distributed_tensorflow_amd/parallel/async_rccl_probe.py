"""The two-sided asynchronous parameter-server transport (``AsyncSendRecvTrainer``) measured in a process of
its own::

    python -m distributed_tensorflow_amd.parallel.async_rccl_probe --rank R --world N --device D --port P \\
        --steps K --warmup W

Why a separate process.  Every (worker, shard) pair has its own RCCL communicator and service stream, and each
of them can sit in an ``ncclRecv`` that waits for a peer.  The streams of one process share a small pool of
hardware queues (4 by default) and a kernel that waits blocks the queue it sits in — with more waiting streams
than queues, the kernel a peer is waiting FOR can be stuck behind one that waits for that peer.  The pool size is
read when the HIP runtime starts, so this process raises it (GPU_MAX_HW_QUEUES) before it imports torch, which
the benchmark's main process — already initialised, and measured with the default pool — cannot do.  A hang or
crash in here costs the caller nothing but this number.

Prints one JSON line (every rank; the caller reads its own rank's).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--locking", type=int, default=0)
    ap.add_argument("--rows", type=int, default=55000)
    args = ap.parse_args(argv)
    # one hardware queue per stream that may wait: 2 (N - 1) communicator streams + (N - 1) service streams + the
    # worker's own (the runtime's ceiling is 32)
    os.environ["GPU_MAX_HW_QUEUES"] = "32" if args.world > 2 else "8"

    import torch
    import torch.distributed as dist

    from ..data import DeviceDataset, read_data_sets
    from .replicas import build_replica_trainer

    out = {"ok": False, "rank": args.rank, "world": args.world}
    try:
        import datetime

        if args.backend == "nccl":
            dev = torch.device("cuda", args.device)
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", rank=args.rank, world_size=args.world, device_id=dev,
                                    init_method=f"tcp://127.0.0.1:{args.port}", timeout=datetime.timedelta(seconds=120))
        else:
            dev = torch.device("cuda", args.device) if torch.cuda.is_available() else torch.device("cpu")
            if dev.type == "cuda":
                torch.cuda.set_device(dev)
            dist.init_process_group("gloo", rank=args.rank, world_size=args.world,
                                    init_method=f"tcp://127.0.0.1:{args.port}", timeout=datetime.timedelta(seconds=120))
        sizes = (args.rows, max(100, args.rows // 11), max(1000, args.rows // 5))
        if args.rows == 55000:
            sizes = (55000, 5000, 10000)
        mnist = read_data_sets("MNIST_data", one_hot=True, seed=0, sizes=sizes)      # the benchmark's data set
        ds = DeviceDataset(mnist.train.images, mnist.train.labels, dev, 100, seed=1000 + args.rank)
        tx, ty = torch.as_tensor(mnist.test.images).to(dev), torch.as_tensor(mnist.test.labels).to(dev)
        tr = build_replica_trainer("async_rccl", device=dev, dataset=ds, batch=100, lr=0.001, seed=1, rank=args.rank,
                                   world=args.world, chunk=50, use_graph=False, use_locking=bool(args.locking))
        tr.run_steps(args.warmup)
        tr.quiesce()
        t0 = time.perf_counter()
        tr.run_steps(args.steps)
        tr.quiesce()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        gs = tr.global_step_value()
        loss, acc = tr.evaluate(tx, ty)
        out.update(ok=True, ms_per_step=round(1000.0 * dt / args.steps, 5), steps=args.steps, warmup=args.warmup,
                   value=round(args.world * 100 * args.steps / dt, 1), unit="samples/s", global_step=gs,
                   every_push_counted=gs == args.world * (args.steps + args.warmup),
                   test_accuracy_after_run=round(acc, 4), test_loss_after_run=round(loss, 4),
                   hardware_queues=os.environ["GPU_MAX_HW_QUEUES"], parallelism=tr.describe(),
                   communicator_setup_s=round(tr.comm_setup_s, 3), library_calls_per_step=tr.host_calls_per_step())
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 — every failure is an answer
        out["error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 3


if __name__ == "__main__":
    sys.exit(main())
