"""One synchronous / asynchronous step variant with ONE replica, for profiling: python bench/xg_one_mode.py MODE [steps] [chunk]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import DeviceDataset  # noqa: E402
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer  # noqa: E402

mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2200
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.rand(55000, 784, generator=g)
y = torch.eye(10)[torch.randint(0, 10, (55000,), generator=g)]
ds = DeviceDataset(x, y, dev, 100, seed=1)
t = build_replica_trainer(mode, device=dev, dataset=ds, batch=100, lr=0.001, seed=1, rank=0, world=1, chunk=chunk,
                          use_graph=True)
t.prepare((550, steps))
t.run_steps(550)
torch.cuda.synchronize()
t0 = time.perf_counter()
t.run_steps(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{mode:11s} {1e6 * dt / steps:7.2f} us/step   {t.describe()}", flush=True)
if hasattr(t, "close"):
    t.close()
