"""Per-chunk cost of the one-launch step: many short run_steps() calls in a row, chained (carry) against afresh."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import DeviceDataset, read_data_sets  # noqa: E402
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer  # noqa: E402

dev = torch.device("cuda")
mnist = read_data_sets("MNIST_data", one_hot=True)
for carry in ("1", "0"):
    os.environ["DTF_PIPE_CARRY"] = carry
    for k in (20, 100):
        ds = DeviceDataset(mnist.train.images, mnist.train.labels, dev, 100, seed=0)
        tr = build_replica_trainer(mode="single", device=dev, dataset=ds, batch=100, lr=0.001, seed=1, chunk=550)
        calls = 500 // k
        tr.prepare((k,) * (calls + 1))
        tr.run_steps(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            tr.run_steps(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("carry=%s  %3d calls of run_steps(%3d): %.3f us per step" % (carry, calls, k, dt / (calls * k) * 1e6), flush=True)
        # one call at a time, the device idle before each (what a 20-step benchmark sees)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.run_steps(k) if tr.engine._host_cursor % 550 + k <= 550 else None
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / k * 1e6)
        print("          single calls from an idle device: %s us per step" % " ".join("%.2f" % t for t in ts), flush=True)
