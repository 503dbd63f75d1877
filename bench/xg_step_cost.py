"""Per-step time of the synchronous step variants with ONE replica (no peer traffic): what each design pays
before any xGMI latency.  Usage: python bench/xg_step_cost.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import DeviceDataset  # noqa: E402
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2200
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.rand(55000, 784, generator=g)
y = torch.eye(10)[torch.randint(0, 10, (55000,), generator=g)]
ds = DeviceDataset(x, y, dev, 100, seed=1)
for mode, extra in (("single", {"pipelined": True}), ("single", {"pipelined": False}), ("towers", {}), ("xgmi", {}),
                    ("xgmi2", {}), ("xgmi2c", {}), ("xgmi2p", {}), ("xgmi_async", {})):
    t = build_replica_trainer(mode, device=dev, dataset=ds, batch=100, lr=0.001, seed=1, rank=0, world=1, chunk=50,
                              use_graph=True, **extra)
    t.prepare((550, steps))
    t.run_steps(550)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.run_steps(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:11s} {str(extra):22s} {1e6 * dt / steps:7.2f} us/step   {t.describe()[:70]}", flush=True)
    if hasattr(t, "close"):
        t.close()
