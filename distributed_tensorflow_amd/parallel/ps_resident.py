"""Device-resident worker loop against parameter-server shards (between-graph, asynchronous).

The per-batch tier runs ``sess.run(train_op, feed_dict=...)`` once per mini-batch like the
reference (tfdist_between.py:92-96): feed copy, pull, kernels, push, fetch.  This tier keeps the
worker's training split in HBM and turns a step into GPU work only::

    pull     shard k -> local parameter slab          (device copy out of the mapped shard)
    compute  forward + loss + backward                 (2 launches, gradients into the local slab)
    push     p_k -= lr * g_k on every shard            (SGD kernel writing through the mapping,
                                                        global_step += 1 on its owner shard)

With ``transport="ipc"`` all of it is device work on the worker's stream, so chunks of steps are
captured into hipGraphs and the host issues one call per chunk; the update stays the reference's
lock-free asynchronous rule (use_locking=False, tfdist_between.py:65): several such workers —
processes of their own, each replaying its graphs — write the same shards concurrently and never
wait for each other.  With the TCP transport the pull and the push cross the host, so the steps
are issued eagerly (still without any per-step feed).

Synchronous replicas need the chief's token per step, a host handshake, and stay on the
per-batch tier; their device-resident counterpart is ``parallel.replicas.SyncShardTrainer``.
"""
from __future__ import annotations

import torch

from ..models.mlp import MLPEngine
from ..utils.trace import span
from .replicas import _BaseTrainer


class PsResidentTrainer(_BaseTrainer):
    name = "ps-resident"

    def __init__(self, sess, train_op, dataset, batch: int, chunk: int = 50, use_graph: bool = True):
        plan = sess._plan(train_op)
        m = plan["mlp"]
        if m is None:
            raise ValueError("the device-resident tier needs the fused MLP plan (graph not recognised)")
        opt = train_op.attrs["optimizer"]
        self.sess, self.store = sess, sess.store
        self.device = sess.device
        self.ds = dataset
        self.rank, self.world = 0, 1
        self.chunk = max(1, int(chunk))
        self.lr_value = float(opt.lr_value())
        self.use_locking = bool(opt.use_locking)
        self.bump_step = train_op.attrs["global_step"] is not None
        capturable = (self.device.type == "cuda" and type(self.store).__name__ in ("IpcVariableStore", "LocalVariableStore")
                      and not self.use_locking)
        self.use_graph = bool(use_graph) and capturable
        self.layout = sess.layout
        self.params, self.grads = sess.theta, sess.grad
        names = {r: m[r].op_name for r in ("W1", "b1", "W2", "b2")}
        self.engine = MLPEngine({r: self.layout.view(self.params, n) for r, n in names.items()},
                                {r: self.layout.view(self.grads, n) for r, n in names.items()},
                                batch=batch, lr=self.lr_value, act=m["act"])
        self.engine.bind_data(dataset.x_epoch, dataset.y_epoch)
        self._bound_set = getattr(dataset, "active", 0)
        self.lr = self.engine.lr
        self.steps_done = 0
        self._graphs = {}
        if self.device.type == "cuda":
            # Load every kernel of the step once, outside any capture, without touching the shards:
            # a pull, the gradient kernels, and a push with lr = 0 that does not count as a step.
            self.store.pull(self.params)
            self.engine.step_grad(0, 1.0)
            self.store.push_apply(self.grads, 0.0, False, bump_step=False)
            torch.cuda.synchronize(self.device)
            self.engine.reset_step_workspace(0)

    def _one_step(self, batch_index: int):
        with span("pull"):
            self.store.pull(self.params)
        with span("step"):
            self.engine.step_grad(batch_index, 1.0)
        with span("push_apply"):
            self.store.push_apply(self.grads, self.lr_value, self.use_locking, bump_step=self.bump_step)

    def _capture(self, pos: int, n: int, parity: int):
        # No warm-up replay here: a step writes the shared shards, which cannot be rolled back.
        self.engine.reset_step_workspace(parity)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for i in range(n):
                self._one_step(pos + i)
        self.engine.parity = parity
        return graph

    def prepare(self, steps_hint=()):
        try:
            super().prepare(steps_hint)
        except Exception as e:  # noqa: BLE001 — a runtime that cannot capture the mapped-shard ops
            print(f"[ps-resident] graph capture failed ({type(e).__name__}: {e}); issuing steps eagerly", flush=True)
            self.use_graph = False
            self._graphs.clear()
            self.engine.reset_step_workspace(self.engine.parity)

    def evaluate(self, x, y):
        self.store.pull(self.params)
        return self.engine.evaluate(x, y)

    def global_step_value(self) -> int:
        return self.store.global_step()

    def describe(self) -> str:
        return (f"between-graph async worker, device-resident: pull -> 2 launches -> one-sided SGD push "
                f"({type(self.store).__name__}" + (", hipGraph chunks of %d)" % self.chunk if self.use_graph else ", eager)"))
