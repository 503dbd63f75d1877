#!/usr/bin/env python
"""Headline benchmark: training throughput of the reference workload.

Workload = the reference's only model/config (tfsingle.py / tfdist_between*.py):
784-100-10 sigmoid MLP, softmax cross-entropy, plain SGD lr 0.001, batch 100 per
worker, MNIST-shaped synthetic data, fp32 (the reference's dtype), N(0,1)
random-init weights.  One "step" is one worker mini-batch: forward, loss,
accuracy, backward and the SGD update (plus, for N > 1, the gradient exchange
with the GPU-resident parameter shards).

    python bench.py --gpus 1 --steps 5500 --warmup 550
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  ``value`` = aggregate samples/s over all GPUs;
``vs_baseline`` divides by the reference's derived single-GPU throughput
(55,000 samples / 1.3 s per epoch = 42,300 samples/s on a GTX 1080, README.md:15)
for N = 1 and by its best published multi-worker aggregate (1 PS + 3 workers
async, ~27,400 samples/s, README.md:249-251) for N > 1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASELINE_SINGLE = 55000.0 / 1.3          # samples/s: reference README.md:15 (1.3 s/epoch)
BASELINE_MULTI = 3 * 55000.0 / 6.03      # samples/s: reference README.md:249-251 (mean of 5.75/5.85/6.5 s)
BATCH = 100
STEPS_PER_EPOCH = 550


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5500)
    ap.add_argument("--warmup", type=int, default=550)
    ap.add_argument("--mode", default=None, choices=[None, "single", "sync", "async", "towers", "xgmi", "xgmi2", "xgmi2c", "xgmi_async", "auto"],
                    help="N=1 default: single; N>1 default: auto = the fastest of sync (PS shard per GPU, RCCL "
                         "reduce-scatter/all-gather), towers (RCCL all-reduce, replicated apply), xgmi (one exchange "
                         "kernel over peer-mapped HBM) and xgmi2 (exchange inside the backward launch), probed before "
                         "the warm-up; the peer-mapped ones only after their trial in child processes succeeded")
    ap.add_argument("--chunk", type=int, default=0,
                    help="steps captured per hipGraph replay (0 = a whole 550-step epoch for one GPU, 50 otherwise)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="run the PyTorch reference path on CPU (debug)")
    return ap.parse_args()


def main():
    args = parse_args()
    from distributed_tensorflow_amd.bench_support import run_benchmark

    result = run_benchmark(args)
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
