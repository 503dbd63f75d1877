"""Optimizers: plain SGD and the SyncReplicas barrier.

``GradientDescentOptimizer(lr).minimize(loss, global_step=gs)`` (reference
tfdist_between.py:65-66, tfsingle.py:48-49) builds a *train op*: running it
computes the gradients on the worker and applies ``var -= lr * grad`` **where
the variable lives** — in place for a local session, on the parameter-server
shard for a distributed one — and adds 1 to ``global_step``.  With the default
``use_locking=False`` concurrent workers update lock-free (Hogwild): each
gradient is computed against whatever snapshot the worker pulled.

``SyncReplicasOptimizer(opt, replicas_to_aggregate=N, total_num_replicas=N,
use_locking=True)`` (reference tfdist_between_sync.py:65-68, 86) turns that into
synchronous data-parallel SGD:

* every worker pushes its gradient, tagged with its ``local_step``, into a
  conditional accumulator next to each variable shard; gradients older than the
  accumulator's step are **dropped as stale**;
* the worker then blocks on the shared token queue; the token it receives is
  the new ``global_step`` and becomes its ``local_step``;
* the **chief queue runner** (``get_chief_queue_runner()``; a thread of worker
  0 between ``start_queue_runners`` and ``request_stop``) loops: wait until
  ``replicas_to_aggregate`` gradients are in every accumulator, apply their
  **mean** once, ``global_step += 1``, release ``total_num_replicas`` tokens.

The token queue starts empty (the reference never runs ``get_init_tokens_op``),
so training is strictly lock-step: one update per N worker batches.  A worker
that waits longer than ``operation_timeout_in_ms`` gets a ``TimeoutError``
naming the likely cause instead of the reference's silent hang.
"""
from __future__ import annotations

import threading
import time
from typing import List, Optional, Sequence

from . import graph as G
from .utils.trace import span


class Optimizer:
    def __init__(self, use_locking: bool = False, name: str = "Optimizer"):
        self.use_locking = bool(use_locking)
        self.name = name

    # learning rate of the update rule p -= lr * g (all this framework's optimizers are SGD-shaped)
    def lr_value(self) -> float:
        raise NotImplementedError

    def compute_gradients(self, loss: G.Tensor, var_list: Optional[Sequence[G.Variable]] = None):
        vs = list(var_list) if var_list is not None else [v for v in loss.graph.variables if v.trainable]
        grads = [G.Tensor("gradient", (loss,), {"wrt": v}, name=f"gradients/{v.op_name.replace('/', '_')}_grad",
                          shape=v.shape) for v in vs]
        return list(zip(grads, vs))

    def apply_gradients(self, grads_and_vars, global_step: Optional[G.Variable] = None, name=None) -> G.Tensor:
        gv = [(g, v) for g, v in grads_and_vars if g is not None]
        loss = gv[0][0].inputs[0] if gv and gv[0][0].op == "gradient" else None
        return G.Tensor("train", (), {"optimizer": self, "loss": loss, "grads_and_vars": gv,
                                      "var_list": [v for _, v in gv], "global_step": global_step},
                        name=name or self.name)

    def minimize(self, loss: G.Tensor, global_step: Optional[G.Variable] = None, var_list=None, name=None) -> G.Tensor:
        return self.apply_gradients(self.compute_gradients(loss, var_list), global_step=global_step, name=name)


class GradientDescentOptimizer(Optimizer):
    def __init__(self, learning_rate: float, use_locking: bool = False, name: str = "GradientDescent"):
        super().__init__(use_locking, name)
        self.learning_rate = float(learning_rate)

    def lr_value(self) -> float:
        return self.learning_rate


class QueueRunner:
    """The chief's aggregation loop, runnable on a thread (``Supervisor.start_queue_runners``)."""

    def __init__(self, sync_opt: "SyncReplicasOptimizer"):
        self.sync_opt = sync_opt
        self.name = "sync_replicas_chief_queue_runner"
        self.exceptions: List[BaseException] = []
        self.rounds = 0
        # after the chief's own loop ended: how long to keep waiting for a gradient of the replicas
        # that have not reported the end of theirs (a crashed replica never will)
        self.drain_idle_secs = 10.0

    def create_threads(self, sess, coord=None, daemon: bool = True, start: bool = False):
        # The aggregator shares this process (and its interpreter lock) with the chief's own training loop, and
        # every replica waits on it once per step: hand the lock over quickly (default: 5 ms).
        import sys

        if sys.getswitchinterval() > 2e-4:
            sys.setswitchinterval(2e-4)
        t = threading.Thread(target=self._run, args=(sess, coord), name=self.name, daemon=daemon)
        if start:
            t.start()
        return [t]

    def _run(self, sess, coord):
        opt = self.sync_opt
        store = sess.store
        if not getattr(store, "is_remote", False):
            return   # a local session aggregates inline
        sess.bind_thread()       # (the chief's applies belong on the session's GPU; a new thread starts on device 0)
        clients = store.aggregator_clients()
        required = opt.replicas_to_aggregate
        idle_since = None
        try:
            store.begin_aggregation(clients)
            while True:
                if coord is not None and coord.should_stop():
                    idle_since = idle_since or time.time()
                    if time.time() - idle_since > self.drain_idle_secs:
                        break
                    # The chief's own loop is over.  Replicas that still have batches left must not
                    # be stranded at the barrier: keep aggregating, sized to the replicas still
                    # training, until every one of them has reported the end of its loop.
                    live = store.live_workers(clients)
                    if live <= 0 or live >= store.n_workers:   # all done / nobody reports its end
                        break
                    required = max(1, min(opt.replicas_to_aggregate, live))
                try:
                    if hasattr(store, "take_apply_release"):
                        # take_grad of every accumulator, the apply, the step bump and the token deal: one request
                        # to the shard that keeps global_step (tfdist_between_sync.py:65-68, 86-89)
                        with span("aggregate_apply_release", required=required):
                            store.take_apply_release(clients, required, opt.lr_value(), opt.tokens_per_step, timeout=0.5)
                        self.rounds += getattr(store, "rounds_in_last_call", 1)     # (a native call may apply several)
                        idle_since = None
                        continue
                    with span("aggregate_apply", required=required):
                        step = store.take_apply_all(clients, required, opt.lr_value(), timeout=0.5)
                except TimeoutError:
                    # nothing to aggregate yet; a replica that finished early no longer counts
                    live = store.live_workers(clients)
                    required = max(1, min(opt.replicas_to_aggregate, live if live > 0 else 1))
                    continue
                with span("release_tokens", step=step):
                    store.enqueue_tokens(clients, step, opt.tokens_per_step)
                self.rounds += 1
                idle_since = None
        except BaseException as e:  # surfaced by Supervisor.stop()
            if coord is not None and coord.should_stop() and isinstance(e, (ConnectionError, OSError, RuntimeError)):
                return   # the parameter servers went away after the last replica finished
            self.exceptions.append(e)
            if coord is not None:
                coord.request_stop(e)
        finally:
            for c in clients:
                c.close()


class SyncReplicasOptimizer(Optimizer):
    def __init__(self, opt: Optimizer, replicas_to_aggregate: int, total_num_replicas: Optional[int] = None,
                 variable_averages=None, variables_to_average=None, use_locking: bool = False,
                 name: str = "sync_replicas"):
        super().__init__(use_locking, name)
        self._opt = opt
        self.replicas_to_aggregate = int(replicas_to_aggregate)
        self.total_num_replicas = int(total_num_replicas if total_num_replicas is not None else replicas_to_aggregate)
        self.tokens_per_step = max(self.total_num_replicas, self.replicas_to_aggregate)
        self._runner: Optional[QueueRunner] = None
        self._train_op: Optional[G.Tensor] = None
        self.chief_init_op = None
        self.local_step_init_op = None
        self.ready_for_local_init_op = None

    def lr_value(self) -> float:
        return self._opt.lr_value()

    def compute_gradients(self, *a, **kw):
        return self._opt.compute_gradients(*a, **kw)

    def apply_gradients(self, grads_and_vars, global_step=None, name=None) -> G.Tensor:
        if global_step is None:
            raise ValueError("SyncReplicasOptimizer needs a global_step")
        op = super().apply_gradients(grads_and_vars, global_step, name)
        self._train_op = op
        self.chief_init_op = G.no_op(name="sync_replicas/chief_init")
        self.local_step_init_op = G.no_op(name="sync_replicas/local_step_init")
        self.ready_for_local_init_op = G.no_op(name="sync_replicas/ready_for_local_init")
        return op

    def get_chief_queue_runner(self) -> QueueRunner:
        if self._train_op is None:
            raise ValueError("call minimize()/apply_gradients() before get_chief_queue_runner()")
        if self._runner is None:
            self._runner = QueueRunner(self)
        return self._runner

    def get_init_tokens_op(self, num_tokens: int = -1) -> G.Tensor:
        """Pre-fill the token queue (lets workers run ahead by num_tokens // replicas steps)."""
        if num_tokens == -1:
            num_tokens = self.replicas_to_aggregate
        return G.Tensor("sync_init_tokens", (), {"num_tokens": int(num_tokens), "optimizer": self},
                        name="sync_replicas/init_tokens")
