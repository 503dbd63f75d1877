#!/usr/bin/env python
"""Headline benchmark: training throughput of the reference workload.

Workload = the reference's only model/config (tfsingle.py / tfdist_between*.py):
784-100-10 sigmoid MLP, softmax cross-entropy, plain SGD lr 0.001, batch 100 per
worker, MNIST-shaped synthetic data, fp32 (the reference's dtype), N(0,1)
random-init weights.  One "step" is one worker mini-batch: forward, loss,
accuracy, backward and the SGD update (plus, for N > 1, the gradient exchange
with the GPU-resident parameter shards).

    python bench.py --gpus 1 --steps 5500 --warmup 550
    python bench.py --gpus 8                       # starts its own 8 rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT a torch.distributed environment, ``--gpus N`` (N > 1) makes this
process a launcher: it creates N fresh rank processes (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment) before it
has touched any GPU itself, waits for them and exits with rank 0's status.  The
launcher never initialises HIP and never replaces a running process image.

Prints ONE JSON line on rank 0.  ``value`` = aggregate samples/s over all GPUs;
``vs_baseline`` divides by the reference's derived single-GPU throughput
(55,000 samples / 1.3 s per epoch = 42,300 samples/s on a GTX 1080, README.md:15)
for N = 1 and by its best published multi-worker aggregate (1 PS + 3 workers
async, ~27,400 samples/s, README.md:249-251) for N > 1.  ``sec_per_epoch`` is the
reference's "Total Time" (550 steps + a cost fetch per 100 steps + the full
test-set evaluation, tfdist_between.py:86-111) and is always measured.

With one GPU the line also carries the reference's scripts in their literal form, run as child processes after the
headline: ``sec_per_epoch_per_batch_tier`` (``tfsingle.py``: one ``sess.run`` with feed and four fetches and one event
record per batch; DTF_BENCH_PER_BATCH=0 skips it) and ``per_batch_clusters`` (``tfdist_between.py`` /
``tfdist_between_sync.py`` as separate ps and worker tasks sharing the GPU: asynchronous 1 ps + 1 worker,
1 ps + 2 workers and 2 ps + 2 workers, synchronous 1 ps + 2 workers; recipes with several workers twice, the faster run
reported: client processes that time-share ONE GPU take turns when their hardware queues land on one of its pipes;
DTF_BENCH_CLUSTERS=0 skips it, on the CPU it runs only with =1).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODES = ["single", "auto", "sync", "towers", "xgmi", "xgmi2", "xgmi2c", "xgmi2p", "async", "xgmi_async", "async_rccl",
         "async_lockstep", "ingraph"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5500)
    ap.add_argument("--warmup", type=int, default=550)
    ap.add_argument("--mode", default=None, choices=[None] + MODES,
                    help="N=1 default: single; N>1 default: auto = every synchronous data-parallel exchange that can "
                         "run here is measured with the same warm-up and step count (towers: RCCL all-reduce, "
                         "replicated apply; sync: PS shard per GPU, RCCL reduce-scatter/all-gather; xgmi / xgmi2 / "
                         "xgmi2c: exchanges over peer-mapped HBM, after their trial in child processes) and the "
                         "fastest consistent one is the headline; the asynchronous modes are reported next to it. "
                         "ingraph: ONE client process drives a tower on each of the N GPUs")
    ap.add_argument("--chunk", type=int, default=0,
                    help="steps captured per hipGraph replay (0 = a whole 550-step epoch for one GPU, 50 otherwise)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="run the PyTorch reference path on CPU (debug, tests)")
    return ap.parse_args(argv)


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """Start one fresh process per rank and wait.  This process imports no GPU library."""
    n = int(args.gpus)
    world = 1 if args.mode == "ingraph" else n          # in-graph replication: a single client for all towers
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DTF_BENCH_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # a rank that dies takes the run with it: the others would wait for it in a collective until their own limit
    codes = [None] * world
    failed_at = None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
                if codes[i] not in (None, 0) and failed_at is None:
                    failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 20.0:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()                               # exactly the processes started above
        time.sleep(0.05)
    return codes[0] if codes[0] != 0 else max(abs(c) for c in codes)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    from distributed_tensorflow_amd.bench_support import run_benchmark

    result = run_benchmark(args)
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
