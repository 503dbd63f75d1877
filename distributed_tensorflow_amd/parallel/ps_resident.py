"""Device-resident worker loop against parameter-server shards (between-graph, asynchronous).

The per-batch tier runs ``sess.run(train_op, feed_dict=...)`` once per mini-batch like the
reference (tfdist_between.py:92-96): feed copy, pull, kernels, push, fetch.  This tier keeps the
worker's training split in HBM and turns a step into GPU work only::

    pull     shard k -> local parameter slab          (device copy out of the mapped shard)
    compute  forward + loss + backward                 (2 launches, gradients into the local slab)
    push     p_k -= lr * g_k on every shard            (SGD kernel writing through the mapping,
                                                        global_step += 1 on its owner shard)

(with GPU-resident shards and no lock the push is not a launch per shard but part of the backward launch: every
backward thread adds ``-lr * g`` of its own gradient element to the shard that holds the variable with one float
atomic through the mapping, ``dtf_mlp_step_async``)

With ``transport="ipc"`` all of it is device work on the worker's stream, so chunks of steps are
captured into hipGraphs and the host issues one call per chunk; the update stays the reference's
lock-free asynchronous rule (use_locking=False, tfdist_between.py:65): several such workers —
processes of their own, each replaying its graphs — write the same shards concurrently and never
wait for each other.  With the TCP transport the pull and the push cross the host, so the steps
are issued eagerly (still without any per-step feed).

Synchronous replicas (``tfdist_between_sync.py --resident``) take the other route,
``PsSyncResidentTrainer``: the chief's accumulate / take_grad / token hand-shake of the per-batch tier is a
host round trip per step, so the device-resident form keeps one replica of the variables per worker and
exchanges gradients INSIDE the backward launch through peer-mapped HBM (``parallel/xgmi.py::PeerInbox``):
the mean of N gradients applied once per step, bit-identical replicas, two launches per step.
"""
from __future__ import annotations

import os

import torch

from ..models.mlp import MLPEngine
from ..utils.trace import span
from .replicas import _BaseTrainer


class PsResidentTrainer(_BaseTrainer):
    name = "ps-resident"

    def __init__(self, sess, train_op, dataset, batch: int, chunk: int = 50, use_graph: bool = True,
                 fused_push="auto"):
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
            if hasattr(self.store, "_lr"):
                self.store._lr(self.lr_value)  # eagerly: a fill recorded into a graph runs only when THAT graph does
            torch.cuda.synchronize(self.device)
            self.engine.reset_step_workspace(0)
        self._push_desc = None
        self._dry = False              # calibration: pushes that add zero and are not counted
        self.push_probe_us = None
        # fold_pull: no pull launch either -- the forward of the pushing step reads W1 / b1 / W2 / b2 from the mapped
        # shards (2 launches per step instead of 3; ``evaluate`` and the step counter pull for themselves)
        self.fold_pull = False
        if (fused_push and self.device.type == "cuda" and not self.use_locking
                and hasattr(self.store, "push_descriptor")):
            from ..ops import hip_backend as hb

            offs = {r: self.layout.var_range[n][0] for r, n in names.items()}
            try:
                # load the pushing backward too, with lr = 0 and no count; shapes the two-launch step does not
                # cover, or more shards than its table holds, keep the push kernels
                self.engine.lr.zero_()
                self.engine.step_async(0, self.store.push_descriptor(offs, bump_step=False))
                self._push_desc = self.store.push_descriptor(offs, bump_step=self.bump_step)
                if os.environ.get("DTF_ASYNC_FOLD_PULL", "1") != "0":
                    try:
                        self.engine.step_async(0, self.store.push_descriptor(offs, bump_step=False, read_shards=True))
                        self.fold_pull = True
                        self._push_desc = self.store.push_descriptor(offs, bump_step=self.bump_step, read_shards=True)
                    except hb.HipKernelError:    # shard boundaries the forward's range lookup does not cover
                        self.fold_pull = False
            except (ValueError, hb.HipKernelError):
                self._push_desc = None
            finally:
                self.engine.lr.fill_(self.lr_value)
                torch.cuda.synchronize(self.device)
                self.engine.reset_step_workspace(0)
            if self._push_desc is not None and fused_push == "auto" and self.use_graph:
                self._choose_push(offs)

    def _choose_push(self, offs):
        """Time the forms of the step against the live shards and keep the fastest one.

        The atomics of the pushing backward travel as 64-byte requests (the 16 lanes of a tile row) and contend
        when several workers hit one shard; a push kernel per shard moves whole vectors but costs a launch per shard.  Which wins depends on how many shards there are, whether they sit
        on this GPU or behind xGMI, and who else is pushing, so it is measured: a chunk of steps of each form is
        captured with a learning rate of zero and no step count (the shards receive ``+ 0``), replayed, timed.
        The same goes for the pull: a forward that reads the variables from the shards saves a launch, but with the
        shards behind xGMI links it issues many small remote reads where a pull moves each shard once in bulk."""
        real_fold = self.fold_pull
        n = min(self.chunk, self.engine.batches_per_epoch)
        times = {}
        self._dry = True
        self.engine.lr.zero_()
        forms = [("fused", False, self.store.push_descriptor(offs, bump_step=False)), ("kernel", False, None)]
        if real_fold:
            forms.insert(0, ("fused, forward from the shards", True,
                             self.store.push_descriptor(offs, bump_step=False, read_shards=True)))
        try:
            for form, fold, desc in forms:
                self._push_desc, self.fold_pull = desc, fold
                g = self._capture(0, n, 0)
                start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    g.replay()
                start.record()
                for _ in range(8):
                    g.replay()
                stop.record()
                stop.synchronize()
                times[form] = 1000.0 * start.elapsed_time(stop) / (8 * n)
                del g
        except Exception:  # noqa: BLE001 — a runtime that cannot capture these steps: prepare() will find out as well
            times = {}
        finally:
            self._dry = False
            self.engine.lr.fill_(self.lr_value)
            self.store._lr(self.lr_value)      # eagerly: a fill recorded into a graph runs only when THAT graph does
            torch.cuda.synchronize(self.device)
            self.engine.reset_step_workspace(0)
        self.push_probe_us = {k: round(v, 2) for k, v in times.items()} or None
        best = min(times, key=times.get) if times else ("fused, forward from the shards" if real_fold else "fused")
        self.fold_pull = best == "fused, forward from the shards"
        self._push_desc = None if best == "kernel" else self.store.push_descriptor(
            offs, bump_step=self.bump_step, read_shards=self.fold_pull)

    def _one_step(self, batch_index: int):
        if self._push_desc is not None and self.fold_pull:
            with span("pull_step_push_apply"):
                self.engine.step_async(batch_index, self._push_desc)
            return
        with span("pull"):
            self.store.pull(self.params)
        if self._push_desc is not None:
            with span("step_push_apply"):
                self.engine.step_async(batch_index, self._push_desc)
            return
        with span("step"):
            self.engine.step_grad(batch_index, 1.0)
        with span("push_apply"):
            if self._dry:
                self.store.push_apply(self.grads, 0.0, False, bump_step=False)
            else:
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

    def fetch_window_async(self, first_step: int, count: int, want_history: bool = False):
        # the step counter lives on its owner shard and is read through the store: fetched at once
        from ..models.mlp import _Fetched

        cost = self.last_loss()
        losses, accs = self.engine.history(first_step, count) if want_history else (None, None)
        return _Fetched((cost, losses, accs, float(self.global_step_value())))

    def describe(self) -> str:
        push = ("forward -> backward that pushes -lr*g per element into the owning shard (float atomics)"
                if self._push_desc is not None else "2 launches -> one-sided SGD push")
        if self.push_probe_us:
            push += " [measured us/step: " + ", ".join(f"{k} {v}" for k, v in self.push_probe_us.items()) + "]"
        pull = "forward that reads the variables from the shards: no pull launch; " if self.fold_pull else "pull -> "
        return (f"between-graph async worker, device-resident: {pull}{push} "
                f"({type(self.store).__name__}" + (", hipGraph chunks of %d)" % self.chunk if self.use_graph else ", eager)"))


MAX_SPINNING_PER_GPU = 3      # processes whose cross-replica backward launches can be resident on one MI355X together


def _device_identity(device: torch.device) -> str:
    """A name that is equal exactly for processes using the same physical GPU of the same host."""
    import socket

    props = torch.cuda.get_device_properties(device)
    tag = getattr(props, "uuid", None)
    if tag is None:
        tag = (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1), getattr(props, "pci_device_id", -1))
    return f"{socket.gethostname()}/{tag}"


class PsSyncResidentTrainer(_BaseTrainer):
    """Synchronous replicas started the reference's way, on the device-resident tier.

    ``SyncReplicasOptimizer(replicas_to_aggregate = total_num_replicas = number of workers)`` means: one update
    per N worker batches, with the mean of their gradients (tfdist_between_sync.py:65-68).  Here every worker
    task keeps a replica of the variables (initialised from the PS shards the chief filled), all workers map
    each other's inboxes over HIP-IPC — the handles travel through the control store of ps task 0, the tasks
    share no ``torch.distributed`` group — and a step is the two-launch cross-replica step of
    ``MLPEngine.step_xg``.  The replicas stay bit-identical; ``publish()`` (the chief calls it at the end of the
    run and before a checkpoint) copies them back into the PS shards.  The workers must share a node.

    Backup workers (``replicas_to_aggregate = R < total_num_replicas``): a step becomes the gradient launches plus the
    one-kernel exchange in its form with arrival tickets (``csrc/xgmi.hip``: the first R gradients to arrive are
    averaged, every worker applies that mean).  Every worker still takes part in every step -- it keeps a replica.
    """

    name = "ps-sync-resident"

    def __init__(self, sess, train_op, dataset, batch: int, chunk: int = 50, use_graph: bool = True,
                 timeout_ms: int = 60000):
        from ..optim import SyncReplicasOptimizer
        from .xgmi import PeerInbox, StoreGroup

        plan = sess._plan(train_op)
        m = plan["mlp"]
        if m is None:
            raise ValueError("the device-resident tier needs the fused MLP plan (graph not recognised)")
        opt = train_op.attrs["optimizer"]
        if not isinstance(opt, SyncReplicasOptimizer):
            raise ValueError("PsSyncResidentTrainer is for SyncReplicasOptimizer train ops")
        self.sess, self.store = sess, sess.store
        self.device = sess.device
        if self.device.type != "cuda":
            raise RuntimeError("the device-resident synchronous tier needs GPU workers")
        self.ds = dataset
        self.rank, self.world = sess.server.task_index, sess.server.num_workers
        if opt.total_num_replicas != self.world or not 1 <= opt.replicas_to_aggregate <= self.world:
            raise ValueError("the resident synchronous tier keeps one replica per worker: total_num_replicas must equal "
                             "the number of workers and replicas_to_aggregate must not exceed it")
        # backup workers (replicas_to_aggregate < total_num_replicas): the mean of the first R gradients to arrive,
        # decided inside the one-kernel exchange; every worker still takes part in every step (it keeps a replica)
        self.required = 0 if opt.replicas_to_aggregate == self.world else int(opt.replicas_to_aggregate)
        self.chunk = max(1, int(chunk))
        self.use_graph = bool(use_graph)
        self.layout = sess.layout
        self.params, self.grads = sess.theta, None
        self.store.pull(self.params)                 # every replica starts from the chief's initial values
        torch.cuda.synchronize(self.device)
        sess.own_replica(self)                       # from here to publish() / close() the shards lag behind
        names = {r: m[r].op_name for r in ("W1", "b1", "W2", "b2")}
        self.global_step = self.layout.view(self.params, "global_step").view(1)
        self.engine = MLPEngine({r: self.layout.view(self.params, n) for r, n in names.items()}, None,
                                batch=batch, lr=float(opt.lr_value()), act=m["act"], global_step=self.global_step)
        self.engine.bind_data(dataset.x_epoch, dataset.y_epoch)
        self._bound_set = getattr(dataset, "active", 0)
        self.lr = self.engine.lr
        self.steps_done = 0
        self._graphs = {}
        group = StoreGroup(sess.server.client("ps", 0), "sync-resident", self.rank, self.world)
        # Where do the workers run?  The exchange inside the backward launch keeps 183 workgroups per worker
        # spinning until every peer's values are in: workers that share a GPU must all be resident on it at once,
        # which holds for at most MAX_SPINNING_PER_GPU of them (more would wait for workgroups that cannot start,
        # until the exchange's time limit).  Such a cluster (e.g. the reference's "1 ps + N workers on gpu0",
        # README.md:42-86, with N > 3) uses the one-kernel exchange instead: a bounded grid after the backward.
        places = [None] * self.world
        group.all_gather_object(places, _device_identity(self.device))
        sharing = max(places.count(p) for p in places)
        forced = os.environ.get("DTF_SYNC_RESIDENT_EXCHANGE", "")
        self.exchange = forced if forced in ("inbox", "kernel") else (
            "kernel" if sharing > MAX_SPINNING_PER_GPU else "inbox")
        if self.required:
            self.exchange = "kernel"         # (the exchange that knows arrival order)
        if self.exchange == "kernel":
            from .xgmi import XgmiExchange

            if self.layout.var_range["global_step"][0] >= 4:
                raise RuntimeError("global_step must live in the first 16 bytes of the slab")
            self.engine.global_step = None
            self.engine._desc = None
            self.xg = XgmiExchange(self.rank, self.world, self.params.numel(), self.device, group=group,
                                   uncached_grads=True, timeout_ms=timeout_ms)
            self.engine.bind_grad_slots([{r: self.layout.view(g, n) for r, n in names.items()} for g in self.xg.grads])
            self._xg_desc = None
            return
        # DTF_XG_COMPACT=1: bare 4-byte words (half the bytes on a link; every worker must agree, the inbox sizes
        # are compared when the handles are exchanged)
        self.xg = PeerInbox(self.rank, self.world, self.params.numel(), self.device, group=group,
                            timeout_ms=timeout_ms, compact=os.environ.get("DTF_XG_COMPACT", "0") == "1")
        self._xg_desc = self.xg.step_descriptor({r: self.layout.var_range[n][0] for r, n in names.items()})

    def _one_step(self, batch_index: int):
        with span("step"):
            if self.exchange == "kernel":
                slot = self.engine.parity
                self.engine.step_grad(batch_index, scale=1.0 / (self.required or self.world), slot=slot)
                self.xg.apply(self.params, self.lr, slot, self.global_step, required=self.required)
            else:
                self.engine.step_xg(batch_index, 1.0 / self.world, self._xg_desc)

    def _capture(self, pos: int, n: int, parity: int):
        # Two warm-up steps (real exchanges, performed by every worker alike) that are rolled back.
        snap = self._snapshot()
        self.engine.reset_step_workspace(parity)
        for _ in range(2):
            self._one_step(pos)
        torch.cuda.synchronize(self.device)
        self._restore(snap)
        self.engine.reset_step_workspace(parity)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for i in range(n):
                self._one_step(pos + i)
        self.engine.parity = parity
        return graph

    def healthy(self) -> bool:
        return not self.xg.failed()

    def publish(self):
        """Copy this replica into the PS shards (chief: end of the run, before a checkpoint)."""
        self.store.publish(self.params)

    def close(self):
        self._graphs.clear()
        self.xg.close()
        if self.sess._replica_owner is self:
            self.sess.own_replica(None)

    def describe(self) -> str:
        how = ("exchange inside the backward launch over peer-mapped HBM" if self.exchange == "inbox" else
               "exchange as one bounded kernel after the backward (more workers share a GPU than can spin on it "
               "together) over peer-mapped HBM")
        if self.required:
            how = (f"exchange as one kernel after the backward over peer-mapped HBM, backup workers: the first "
                   f"{self.required} gradients to arrive count")
        return (f"between-graph sync worker {self.rank} of {self.world}, device-resident: replica per worker, gradient "
                f"{how} (mean of {self.required or self.world}, one update per "
                f"{self.world} batches), {2 if self.exchange == 'inbox' else 3} launches/step"
                + (", hipGraph chunks of %d" % self.chunk if self.use_graph else ", eager"))
