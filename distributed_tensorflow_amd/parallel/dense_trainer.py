"""Device-resident training of a fully connected stack of any depth (single process).

The counterpart of ``replicas.SingleTrainer`` for graphs that ``session.match_dense_stack`` recognises but that are not
the reference's 784-100-10 network: the variables stay where the session keeps them (views of its flat slab, so
``Saver``, ``sess.run(variable)`` and the Supervisor see the trained values), the training split lives in HBM
(``data.DeviceDataset``), and a window of steps is one hipGraph replay of ``models.dense.DenseStackEngine`` launches —
no feed, no fetch, no host work per step.  Same driver surface as the trainers of ``parallel.replicas`` (what
``examples.mnist_mlp.run_resident_loop`` calls), so the scripts print the reference's progress lines
(tfsingle.py:83-97) unchanged.
"""
from __future__ import annotations

from typing import Tuple

import torch

from ..models.dense import DenseStackEngine


class DenseResidentTrainer:
    name = "dense"

    def __init__(self, sess, train_op, dataset, batch: int, chunk: int = 50, use_graph: bool = True):
        if sess.store.is_remote:
            raise ValueError("the resident dense trainer runs against a local session (no parameter-server tasks)")
        plan = sess._plan(train_op)
        d = plan.get("dense")
        if d is None:
            raise ValueError("the train op's loss is not a fully connected stack with a softmax cross-entropy head")
        self.sess, self.ds = sess, dataset
        self.device = sess.device
        self.chunk = max(1, int(chunk))
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        view = lambda v: None if v is None else sess.layout.view(sess.theta, v.op_name)  # noqa: E731
        gs_var = train_op.attrs["global_step"]
        self.global_step = None if gs_var is None else view(gs_var).view(1)
        self.engine = DenseStackEngine([view(W) for W, _ in d["layers"]], [view(b) for _, b in d["layers"]], d["acts"],
                                       batch=int(batch), lr=train_op.attrs["optimizer"].lr_value(),
                                       global_step=self.global_step)
        self.engine.bind_data(dataset.x_epoch, dataset.y_epoch)
        self._bound_set = getattr(dataset, "active", 0)
        self._graphs = {}

    def describe(self) -> str:
        return ("dense stack " + "-".join(str(n) for n in self.engine.dims) + f", {self.engine.launches_per_step} "
                f"launches per step, " + ("hipGraph chunks" if self.use_graph else "eager"))

    def _schedule(self, cursor: int, n: int):
        bpe, out = self.engine.batches_per_epoch, []
        while n > 0:
            k = min(n, self.chunk, bpe - cursor % bpe)
            out.append(k)
            cursor += k
            n -= k
        return out

    def _bind_active_set(self):
        """A double-buffered data set alternates between two epoch-ordered copies; graphs are kept per copy."""
        index = getattr(self.ds, "active", 0)
        if index != self._bound_set:
            x, y, _ = self.ds.set_tensors(index)
            keep = self.engine._graphs
            self._graphs[self._bound_set] = keep
            self.engine.bind_data(x, y)                 # (drops the engine's graphs: they read the other copy)
            self.engine._graphs = self._graphs.get(index, {})
            self._bound_set = index

    def prepare(self, steps_hint=()):
        """Capture the chunk graphs the hinted run_steps() calls will replay, outside any timed region."""
        if not self.use_graph:
            return
        cursor, bpe = self.engine._host_cursor, self.engine.batches_per_epoch
        for n in steps_hint:
            for k in self._schedule(cursor, n):
                key = (cursor % bpe, k)
                if key not in self.engine._graphs:
                    self.engine._graphs[key] = self.engine._capture(*key)
                cursor += k

    def run_steps(self, n: int):
        """n steps; reshuffles the resident set at every epoch boundary."""
        bpe = self.engine.batches_per_epoch
        for k in self._schedule(self.engine._host_cursor, n):
            cur = self.engine._host_cursor
            if cur % bpe == 0 and cur > 0:
                self.ds.shuffle_epoch()
            self._bind_active_set()
            self.engine.run_chunk(k, use_graph=self.use_graph)
        self.sess._gs_mirror = None         # the step counter moved behind the session's back

    def last_loss(self) -> float:
        return self.engine.last_loss()

    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[float, float]:
        return self.engine.evaluate(x, y)

    def global_step_value(self) -> int:
        return 0 if self.global_step is None else int(round(float(self.global_step.item())))

    def close(self):
        self._graphs.clear()
        self.engine._graphs.clear()
