"""In-graph replication: ONE client process driving a tower on every GPU.

The reference abandoned this mode (README.md:261-264); it is provided here as
its own design.  The single client holds one replica of the variables per
tower, gives every tower its own mini-batch of the resident data set, and per
step

1. launches the fused forward/backward on every tower (gradients pre-scaled by
   1/N into that tower's gradient slab),
2. **all-reduces** the N gradient slabs — ``torch.cuda.nccl.all_reduce``, the
   single-process multi-GPU entry of RCCL, over xGMI — so that every tower
   holds the mean gradient,
3. applies the identical sharded-SGD kernel on every tower (replicas stay
   bit-identical; ``global_step`` advances once).

All launches of a step are asynchronous; the client synchronises only when it
reads the cost.  Towers on CPU devices (tests) or several towers on one GPU
reduce through a plain sum instead of RCCL.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from ..data import DeviceDataset
from ..models.mlp import MLPEngine
from ..ops import hip_backend as hb
from ..ops import reference as ref
from ..variables import ShardLayout, assign_shards, mlp_var_specs


class InGraphTowers:
    def __init__(self, devices: Sequence, images, labels, batch: int, lr: float, seed: int = 1,
                 act: str = "sigmoid", dims=(784, 100, 10), data_seed: int = 0):
        self.devices = [torch.device(d) for d in devices]
        self.n = len(self.devices)
        if self.n < 1:
            raise ValueError("need at least one tower")
        specs = mlp_var_specs(*dims)
        assign_shards(specs, 1)
        self.layout = ShardLayout(specs, 1, mode="flat")
        self.params: List[torch.Tensor] = []
        self.grads: List[torch.Tensor] = []
        self.engines: List[MLPEngine] = []
        self.datasets: List[DeviceDataset] = []
        self.lr_t: List[torch.Tensor] = []
        for t, dev in enumerate(self.devices):
            p = self.layout.new_slab(dev)
            if t == 0:
                self.layout.initialize(p, seed)
            else:
                p.copy_(self.params[0])           # towers start as exact replicas
            g = self.layout.new_slab(dev)
            eng = MLPEngine(self.layout.mlp_views(p), self.layout.mlp_views(g), batch=batch, lr=lr, act=act)
            ds = DeviceDataset(images, labels, dev, batch, seed=data_seed + t)   # a different shuffle per tower
            eng.bind_data(ds.x_epoch, ds.y_epoch)
            self.params.append(p)
            self.grads.append(g)
            self.engines.append(eng)
            self.datasets.append(ds)
            self.lr_t.append(eng.lr)
        self.lr = float(lr)
        self.steps = 0
        distinct_gpus = all(d.type == "cuda" for d in self.devices) and \
            len({d.index for d in self.devices}) == self.n
        self.use_rccl = distinct_gpus and self.n > 1
        self.batches_per_epoch = self.engines[0].batches_per_epoch

    def describe(self) -> str:
        how = "torch.cuda.nccl.all_reduce (RCCL)" if self.use_rccl else "in-process sum"
        return f"in-graph: 1 client, {self.n} towers, gradient mean by {how}"

    def _reduce(self):
        if self.n == 1:
            return
        if self.use_rccl:
            import torch.cuda.nccl as nccl

            nccl.all_reduce(self.grads)
        else:
            total = self.grads[0].clone()
            for g in self.grads[1:]:
                total += g.to(total.device)
            for g in self.grads:
                g.copy_(total.to(g.device))

    def step(self):
        pos = self.steps % self.batches_per_epoch
        if pos == 0 and self.steps > 0:
            for ds in self.datasets:
                ds.shuffle_epoch()
        for t, eng in enumerate(self.engines):
            with torch.cuda.device(self.devices[t]) if self.devices[t].type == "cuda" else _null():
                eng.step_grad(pos, scale=1.0 / self.n)
        self._reduce()
        for t in range(self.n):
            dev = self.devices[t]
            gs = self.layout.view(self.params[t], "global_step").view(1)
            if dev.type == "cuda":
                with torch.cuda.device(dev):
                    hb.sgd_apply_(self.params[t], self.grads[t], self.lr_t[t], 1.0, False, gs)
            else:
                ref.sgd_apply_(self.params[t], self.grads[t], self.lr)
                gs.add_(1.0)
        for eng in self.engines:
            eng.advance(1)
        self.steps += 1

    def run_steps(self, n: int):
        for _ in range(n):
            self.step()

    # -- the trainer surface shared with parallel.replicas (used by the example loop) ----------
    @property
    def device(self):
        return self.devices[0]

    @property
    def engine(self):
        return self.engines[0]

    def last_loss(self) -> float:
        """Mean of the towers' last mini-batch costs."""
        return sum(e.last_loss() for e in self.engines) / self.n

    def evaluate(self, x, y):
        return self.engines[0].evaluate(x, y)

    def global_step_value(self) -> int:
        return int(round(float(self.layout.view(self.params[0], "global_step").item())))

    def replicas_identical(self) -> bool:
        ref0 = self.params[0].cpu()
        return all(torch.equal(ref0, p.cpu()) for p in self.params[1:])


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
