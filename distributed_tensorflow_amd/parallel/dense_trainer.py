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
        with self.sess.run_lock:            # (the Supervisor's checkpoint thread reads the same variables)
            for k in self._schedule(self.engine._host_cursor, n):
                cur = self.engine._host_cursor
                if cur % bpe == 0 and cur > 0:
                    self.ds.shuffle_epoch()
                self._bind_active_set()
                self.engine.run_chunk(k, use_graph=self.use_graph)
            self.sess._gs_mirror = None     # the step counter moved behind the session's back

    def last_loss(self) -> float:
        return self.engine.last_loss()

    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[float, float]:
        return self.engine.evaluate(x, y)

    def global_step_value(self) -> int:
        return 0 if self.global_step is None else int(round(float(self.global_step.item())))

    def close(self):
        self._graphs.clear()
        self.engine._graphs.clear()


class DenseDataParallelTrainer:
    """Synchronous data-parallel training of a fully connected stack, one process per GPU (``torch.distributed``:
    RCCL on GPUs, gloo on CPUs) — the update of ``tfdist_between_sync.py`` (the mean of N workers' gradients applied
    once, tfdist_between_sync.py:65-68) for models whose gradients are megabytes, not the reference's 318 KB.

    Laid out for the all-reduce, not for the model: parameters and gradients are two flat slabs whose ranges follow the
    order in which the backward pass PRODUCES gradients (top layer first).  The gradient slab is cut into buckets of
    about ``bucket_mb`` at layer boundaries; when the backward has written a bucket, a side stream all-reduces it
    and applies ``p -= lr / N * sum`` to the matching parameter range with the SGD kernel (``hb.sgd_apply_``) while the
    main stream goes on with the layers below: the collectives of the upper layers run under the GEMMs of the lower
    ones, and only the last bucket is exposed.  Few, large buckets: a ring all-reduce over xGMI is bound by one link
    per hop (about 50 GB/s effective of a link's 153), so a bucket should be long enough for its bandwidth term to
    dominate its latency term (32 MB: 0.6 ms against some tens of microseconds), and 288 GB of HBM make room for the
    second slab.  All replicas start from the same seed and apply the same sums, so they stay bit-identical.
    """

    name = "dense-dp"

    def __init__(self, dims, acts, dataset, batch: int, lr: float, device, rank: int = 0, world: int = 1, seed: int = 1,
                 bucket_mb: float = 32.0, group=None, use_graph: bool = False, chunk: int = 8):
        from ..ops import hip_backend as hb

        self.hb = hb
        self.device = torch.device(device)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.ds = dataset
        dims = [int(d) for d in dims]
        L = len(dims) - 1
        # slab order: layer L-1 (top) first, weight then bias, down to layer 0
        sizes = []
        for i in range(L - 1, -1, -1):
            sizes += [("W", i, dims[i] * dims[i + 1]), ("b", i, dims[i + 1])]
        # every tensor starts on a 16-byte boundary (vector loads of the GEMMs and of the SGD kernel); the padding
        # holds zeros in both slabs
        pad4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
        total = sum(pad4(n) for _, _, n in sizes)
        self.params = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(total, dtype=torch.float32, device=self.device)
        W, b, gW, gb, self._range = [None] * L, [None] * L, [None] * L, [None] * L, {}
        off = 0
        gen = torch.Generator().manual_seed(int(seed))      # the same values on every rank
        for kind, i, n in sizes:
            self._range[(kind, i)] = (off, off + pad4(n))
            pv, gv = self.params[off:off + n], self.grads[off:off + n]
            if kind == "W":
                W[i], gW[i] = pv.view(dims[i], dims[i + 1]), gv.view(dims[i], dims[i + 1])
                W[i].copy_((torch.randn(dims[i], dims[i + 1], generator=gen) * dims[i] ** -0.5).to(self.device))
            else:
                b[i], gb[i] = pv, gv
                b[i].zero_()
            off += pad4(n)
        self.global_step = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.engine = DenseStackEngine(W, b, acts, batch=int(batch), lr=lr, grads=list(zip(gW, gb)))
        self.engine.bind_data(dataset.x_epoch, dataset.y_epoch)
        self.lr = self.engine.lr
        # buckets: consecutive layers (from the top) until about bucket_mb of gradients
        limit = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets, start, last_layer = [], 0, {}
        for i in range(L - 1, -1, -1):
            end = self._range[("b", i)][1]
            if end - start >= limit or i == 0:
                self.buckets.append((start, end))
                last_layer[i] = len(self.buckets) - 1
                start = end
        self._closes_bucket = last_layer              # layer index -> bucket it completes
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        # GPU ranks over gloo (several ranks sharing one GPU in a rehearsal: RCCL refuses that) reduce through the host:
        # gloo's own handling of device tensors was seen to hang on a 64 MB bucket issued from a side stream
        self._host_reduce = False
        self._reduce = self.world > 1 or group is not None      # an explicit group is used even by a lone rank
        if self._reduce and self.device.type == "cuda":
            import torch.distributed as dist

            self._host_reduce = dist.get_backend(group) == "gloo"
        # hipGraph chunks (small batches, where a step is launch-bound): the side stream joins the capture through the
        # events of _step, so a chunk is ONE graph whose collectives run on their own branch.  Not with gloo (host copies).
        self.use_graph = bool(use_graph) and self.device.type == "cuda" and not self._host_reduce
        self.chunk = max(1, int(chunk))
        self._graphs = {}
        if self.use_graph:
            # load every kernel (and set the communicator up) outside any capture: gradients of zero weight, an update
            # of zero size, one small collective that every rank issues here alike
            import torch.distributed as dist

            self.engine.step_grad(0, 0.0)
            self.hb.sgd_apply_(self.params[:4], self.grads[:4], self.lr, 0.0)
            if self._reduce:
                with torch.cuda.stream(self._side):
                    dist.all_reduce(torch.zeros(4, device=self.device), group=self.group)
            self.grads.zero_()
            torch.cuda.synchronize(self.device)

    def describe(self) -> str:
        mb = [round((b - a) * 4 / (1 << 20), 2) for a, b in self.buckets]
        return (f"dense stack {'-'.join(str(n) for n in self.engine.dims)} dp{self.world}: all-reduce of {len(mb)} "
                f"gradient bucket(s) {mb} MB on a side stream under the backward of the layers below, SGD on the reduced "
                f"range, {'RCCL' if self.device.type == 'cuda' and not self._host_reduce else 'gloo'}"
                + (f", hipGraph chunks of {self.chunk}" if self.use_graph else ""))

    # -- one step -------------------------------------------------------------------------------------------------
    def _reduce_apply(self, k: int):
        import torch.distributed as dist

        a, b = self.buckets[k]
        g = self.grads[a:b]
        if self._reduce and self._host_reduce:
            h = g.cpu()
            dist.all_reduce(h, group=self.group)
            g.copy_(h)
        elif self._reduce:
            dist.all_reduce(g, group=self.group)
        if self.device.type == "cuda":
            self.hb.sgd_apply_(self.params[a:b], g, self.lr, 1.0 / self.world)
        else:
            self.params[a:b].sub_(g, alpha=float(self.lr.item()) / self.world)

    def _step(self, bi: int):
        eng, hb = self.engine, self.hb
        if self.device.type != "cuda":
            eng.step_grad(bi, 1.0)
            for k in range(len(self.buckets)):
                self._reduce_apply(k)
            self.global_step.add_(1.0)
            return
        x, y = eng._batch(bi)
        slot = eng._slot(bi)
        main = torch.cuda.current_stream(self.device)
        loss_slot, correct_slot = eng.loss_sum[slot:slot + 1], eng.correct[slot:slot + 1]
        logits = eng._forward_gpu(x, eng.a)
        if eng.B <= hb.XENT_STEP_MAX_ROWS:
            hb.softmax_xent_step(logits, y, eng.dz[-1], 1.0 / eng.B, loss_slot, correct_slot, self.global_step)
        else:
            hb.softmax_xent_head(logits, y, eng.dz[-1], 1.0 / eng.B, eng._head_scratch, loss_slot, correct_slot,
                                 self.global_step)
        for i in range(eng.L - 1, -1, -1):
            below = eng.a[i - 1] if i > 0 else x
            gw, gb = eng.g[i]
            hb.gemm(hb.GEMM_TN, below, eng.dz[i], out=gw, workspace=eng._ws)
            if eng.b[i] is not None:
                hb.colsum(eng.dz[i], out=gb)
            if i > 0:       # reads W_i: ordered before the side stream's update of W_i by the event below
                hb.gemm(hb.GEMM_NT, eng.dz[i], eng.W[i], out=eng.dz[i - 1], epi=hb.EPI_ACT_BWD, act=eng.acts[i - 1],
                        aux=below, workspace=eng._ws)
            k = self._closes_bucket.get(i)
            if k is not None:
                done = torch.cuda.Event()
                done.record(main)
                self._side.wait_event(done)
                with torch.cuda.stream(self._side):
                    self._reduce_apply(k)
        main.wait_stream(self._side)            # the next forward reads the updated parameters

    def _graph(self, pos: int, k: int):
        key = (pos, k)
        g = self._graphs.get(key)
        if g is None:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for i in range(k):
                    self._step(pos + i)
            self._graphs[key] = g
        return g

    def run_steps(self, n: int):
        eng = self.engine
        bpe = eng.batches_per_epoch
        while n > 0:
            cur = eng._host_cursor
            if cur % bpe == 0 and cur > 0:
                self.ds.shuffle_epoch()
                eng.bind_data(self.ds.x_epoch, self.ds.y_epoch)     # (same storage: the captured chunks stay valid)
            k = min(n, self.chunk, bpe - cur % bpe) if self.use_graph else 1
            if self.use_graph:
                try:
                    self._graph(cur % bpe, k).replay()
                except Exception as e:  # noqa: BLE001 — a runtime that cannot capture the collectives: eager from here on
                    print(f"[dense-dp] graph capture failed ({type(e).__name__}: {e}); issuing steps eagerly", flush=True)
                    self.use_graph = False
                    self._graphs.clear()
                    continue
            else:
                self._step(cur % bpe)
            eng._host_cursor += k
            n -= k

    def prepare(self, steps_hint=()):
        pass

    def last_loss(self) -> float:
        return self.engine.last_loss()

    def evaluate(self, x, y):
        return self.engine.evaluate(x, y)

    def global_step_value(self) -> int:
        return int(round(float(self.global_step.item())))

    def replica_checksum(self) -> float:
        return float(self.params.double().sum())

    def close(self):
        self._graphs.clear()


class PsDenseResidentTrainer:
    """Device-resident asynchronous worker of a between-graph cluster for a fully connected stack of any depth — what
    ``ps_resident.PsResidentTrainer`` is for the reference's network (tfdist_between.py:92-96 without the per-batch
    feed, fetch and host round trips).  A step is GPU work only::

        pull     every parameter-server shard -> the session's parameter slab   (device copies out of the mapped shards)
        compute  ``DenseStackEngine.step_grad``: forward, head, backward, gradients into the session's gradient slab
        push     ``p -= lr * g`` on every shard, written through the mapping; global_step += 1 on its owner shard

    With the ipc transport (shards in the parameter-server tasks' HBM) and ``use_locking=False`` all of it is captured
    into hipGraph chunks; several such worker processes replay their graphs against the same shards without waiting for
    each other — the reference's lock-free asynchronous update (tfdist_between.py:65).  Other transports issue the same
    steps eagerly.  Driver surface of the other resident trainers (``examples.mnist_mlp.run_resident_loop``)."""

    name = "ps-dense-resident"

    def __init__(self, sess, train_op, dataset, batch: int, chunk: int = 50, use_graph: bool = True):
        plan = sess._plan(train_op)
        d = plan.get("dense")
        if d is None:
            raise ValueError("the train op's loss is not a fully connected stack with a softmax cross-entropy head")
        opt = train_op.attrs["optimizer"]
        self.sess, self.store, self.ds = sess, sess.store, dataset
        self.device = sess.device
        self.chunk = max(1, int(chunk))
        self.lr_value = float(opt.lr_value())
        self.use_locking = bool(opt.use_locking)
        self.bump_step = train_op.attrs["global_step"] is not None
        capturable = (self.device.type == "cuda" and not self.use_locking
                      and type(self.store).__name__ in ("IpcVariableStore", "LocalVariableStore"))
        self.use_graph = bool(use_graph) and capturable
        self.params, self.grads = sess.theta, sess.grad
        view = lambda slab, v: None if v is None else sess.layout.view(slab, v.op_name)  # noqa: E731
        self.engine = DenseStackEngine([view(self.params, W) for W, _ in d["layers"]],
                                       [view(self.params, b) for _, b in d["layers"]], d["acts"], batch=int(batch),
                                       lr=self.lr_value,
                                       grads=[(view(self.grads, W), view(self.grads, b)) for W, b in d["layers"]])
        self.engine.bind_data(dataset.x_epoch, dataset.y_epoch)
        self._bound_set = getattr(dataset, "active", 0)
        self._graphs = {}
        if self.device.type == "cuda":
            # load every kernel of a step outside any capture without touching the shards: a pull, the gradient
            # kernels, and a push with lr = 0 that does not count as a step
            self.store.pull(self.params)
            self.engine.step_grad(0, 1.0)
            self.store.push_apply(self.grads, 0.0, False, bump_step=False)
            if hasattr(self.store, "_lr"):
                self.store._lr(self.lr_value)   # eagerly: a fill recorded into a graph runs only when THAT graph does
            torch.cuda.synchronize(self.device)

    def describe(self) -> str:
        return ("between-graph async worker, device-resident, dense stack " + "-".join(str(n) for n in self.engine.dims)
                + f": pull -> {self.engine.launches_per_step} launches -> one-sided SGD push ({type(self.store).__name__}, "
                + ("hipGraph chunks of %d)" % self.chunk if self.use_graph else "eager)"))

    def _schedule(self, cursor: int, n: int):
        bpe, out = self.engine.batches_per_epoch, []
        while n > 0:
            k = min(n, self.chunk, bpe - cursor % bpe)
            out.append(k)
            cursor += k
            n -= k
        return out

    def _one_step(self, bi: int):
        self.store.pull(self.params)
        self.engine.step_grad(bi, 1.0)
        self.store.push_apply(self.grads, self.lr_value, self.use_locking, bump_step=self.bump_step)

    def _graph(self, pos: int, n: int):
        key = (pos, n, self._bound_set)
        g = self._graphs.get(key)
        if g is None:
            # no warm-up replay: a step writes the shared shards, which cannot be rolled back
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for i in range(n):
                    self._one_step(pos + i)
            self._graphs[key] = g
        return g

    def _bind_active_set(self):
        index = getattr(self.ds, "active", 0)
        if index != self._bound_set:
            x, y, _ = self.ds.set_tensors(index)
            self.engine.bind_data(x, y)
            self._bound_set = index

    def prepare(self, steps_hint=()):
        """Capture the chunk graphs of the hinted run_steps() calls for the data copy that is bound now; should the
        runtime refuse to capture the mapped-shard operations, the steps are issued eagerly from then on."""
        if not self.use_graph:
            return
        cursor, bpe = self.engine._host_cursor, self.engine.batches_per_epoch
        try:
            for n in steps_hint:
                for k in self._schedule(cursor, n):
                    self._graph(cursor % bpe, k)
                    cursor += k
        except Exception as e:  # noqa: BLE001
            print(f"[ps-dense-resident] graph capture failed ({type(e).__name__}: {e}); issuing steps eagerly", flush=True)
            self.use_graph = False
            self._graphs.clear()

    def run_steps(self, n: int):
        bpe = self.engine.batches_per_epoch
        with self.sess.run_lock:            # (the Supervisor's checkpoint thread pulls into the same slab)
            for k in self._schedule(self.engine._host_cursor, n):
                cur = self.engine._host_cursor
                if cur % bpe == 0 and cur > 0:
                    self.ds.shuffle_epoch()
                self._bind_active_set()
                pos = cur % bpe
                if self.use_graph:
                    self._graph(pos, k).replay()
                else:
                    for i in range(k):
                        self._one_step(pos + i)
                self.engine.advance(k)
            self.sess._gs_mirror = None     # the step counter moved behind the session's back

    def last_loss(self) -> float:
        return self.engine.last_loss()

    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[float, float]:
        with self.sess.run_lock:
            self.store.pull(self.params)
            return self.engine.evaluate(x, y)

    def global_step_value(self) -> int:
        return self.store.global_step()

    def close(self):
        self._graphs.clear()


class PsDenseSyncResidentTrainer:
    """Synchronous replicas of a deeper stack started the reference's way (``tfdist_between_sync.py --resident --hidden
    ...``), device-resident: ``SyncReplicasOptimizer(replicas_to_aggregate = total_num_replicas = number of workers)``
    is one update per N worker batches with the mean of their gradients (tfdist_between_sync.py:65-68).  Every worker
    task keeps a replica of the variables (pulled from the PS shards the chief initialised); a step is the stack's
    gradient launches (``DenseStackEngine.step_grad`` scaled by 1 / N, into one of two gradient slabs that alternate
    from step to step) and ONE exchange kernel that waits for every worker's slab of that step, sums them in rank order
    through HIP-IPC mappings and applies SGD to the local replica (``xgmi.XgmiExchange``: a bounded grid, so any number
    of workers can share a GPU).  The handles travel through the control store of ps task 0; the replicas stay
    bit-identical; ``publish()`` copies the chief's replica back into the PS shards.  Workers must share a node.
    Backup workers (``replicas_to_aggregate = R < total_num_replicas``): the exchange kernel averages the first R
    gradients to arrive (arrival tickets, ``csrc/xgmi.hip``) and every worker applies that mean."""

    name = "ps-dense-sync-resident"

    def __init__(self, sess, train_op, dataset, batch: int, chunk: int = 50, use_graph: bool = True,
                 timeout_ms: int = 60000):
        from ..optim import SyncReplicasOptimizer
        from .xgmi import StoreGroup, XgmiExchange

        plan = sess._plan(train_op)
        d = plan.get("dense")
        if d is None:
            raise ValueError("the train op's loss is not a fully connected stack with a softmax cross-entropy head")
        opt = train_op.attrs["optimizer"]
        if not isinstance(opt, SyncReplicasOptimizer):
            raise ValueError("PsDenseSyncResidentTrainer is for SyncReplicasOptimizer train ops")
        self.sess, self.store, self.ds = sess, sess.store, dataset
        self.device = sess.device
        if self.device.type != "cuda":
            raise RuntimeError("the device-resident synchronous tier needs GPU workers")
        self.rank, self.world = sess.server.task_index, sess.server.num_workers
        if opt.total_num_replicas != self.world or not 1 <= opt.replicas_to_aggregate <= self.world:
            raise ValueError("the resident synchronous tier keeps one replica per worker: total_num_replicas must equal "
                             "the number of workers and replicas_to_aggregate must not exceed it")
        # backup workers (replicas_to_aggregate < total_num_replicas): the mean of the first R gradients to arrive,
        # decided inside the exchange kernel; every worker still takes part in every step (it keeps a replica)
        self.required = 0 if opt.replicas_to_aggregate == self.world else int(opt.replicas_to_aggregate)
        self.chunk = max(1, int(chunk))
        self.use_graph = bool(use_graph)
        layout = sess.layout
        if train_op.attrs["global_step"] is None or layout.var_range["global_step"][0] >= 4:
            raise RuntimeError("global_step must live in the first 16 bytes of the slab")
        self.params = sess.theta
        self.store.pull(self.params)                 # every replica starts from the chief's initial values
        torch.cuda.synchronize(self.device)
        sess.own_replica(self)                       # from here to publish() / close() the shards lag behind
        self.global_step = layout.view(self.params, "global_step").view(1)
        view = lambda slab, v: None if v is None else layout.view(slab, v.op_name)  # noqa: E731
        group = StoreGroup(sess.server.client("ps", 0), "dense-sync-resident", self.rank, self.world)
        self.xg = XgmiExchange(self.rank, self.world, self.params.numel(), self.device, group=group,
                               uncached_grads=True, timeout_ms=timeout_ms)
        for g in self.xg.grads:                      # whatever no variable covers (step counter, padding) adds zero
            g.zero_()
        self._gsets = [[(view(g, W), view(g, b)) for W, b in d["layers"]] for g in self.xg.grads]
        self.engine = DenseStackEngine([view(self.params, W) for W, _ in d["layers"]],
                                       [view(self.params, b) for _, b in d["layers"]], d["acts"], batch=int(batch),
                                       lr=float(opt.lr_value()), grads=self._gsets[0])
        self.lr = self.engine.lr
        self.engine.bind_data(dataset.x_epoch, dataset.y_epoch)
        self._bound_set = getattr(dataset, "active", 0)
        self._graphs = {}
        self.engine.step_grad(0, 0.0)                # loads the kernels outside any capture (gradients of zero weight)
        for g in self.xg.grads:
            g.zero_()
        torch.cuda.synchronize(self.device)

    def describe(self) -> str:
        return (f"between-graph sync worker {self.rank} of {self.world}, device-resident, dense stack "
                + "-".join(str(n) for n in self.engine.dims) + f": replica per worker, {self.engine.launches_per_step} "
                f"launches + one exchange kernel per step over peer-mapped HBM (mean of "
                + (f"the first {self.required} of {self.world} gradients to arrive" if self.required else f"{self.world}")
                + f", one update per {self.world} batches)" + (", hipGraph chunks of %d" % self.chunk if self.use_graph else ", eager"))

    _schedule = PsDenseResidentTrainer._schedule
    _bind_active_set = PsDenseResidentTrainer._bind_active_set

    def _one_step(self, bi: int, slot: int):
        self.engine.bind_grads(self._gsets[slot])
        self.engine.step_grad(bi, 1.0 / (self.required or self.world))
        self.xg.apply(self.params, self.lr, slot, self.global_step, required=self.required)

    def _graph(self, pos: int, n: int, parity: int):
        key = (pos, n, parity, self._bound_set)
        g = self._graphs.get(key)
        if g is None:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for i in range(n):
                    self._one_step(pos + i, (parity + i) & 1)
            self._graphs[key] = g
        return g

    def prepare(self, steps_hint=()):
        if not self.use_graph:
            return
        cursor, bpe = self.engine._host_cursor, self.engine.batches_per_epoch
        for n in steps_hint:
            for k in self._schedule(cursor, n):
                self._graph(cursor % bpe, k, cursor & 1)
                cursor += k

    def run_steps(self, n: int):
        bpe = self.engine.batches_per_epoch
        with self.sess.run_lock:
            for k in self._schedule(self.engine._host_cursor, n):
                cur = self.engine._host_cursor
                if cur % bpe == 0 and cur > 0:
                    self.ds.shuffle_epoch()
                self._bind_active_set()
                pos = cur % bpe
                if self.use_graph:
                    self._graph(pos, k, cur & 1).replay()
                else:
                    for i in range(k):
                        self._one_step(pos + i, (cur + i) & 1)
                self.engine.advance(k)
            self.sess._gs_mirror = None

    def last_loss(self) -> float:
        return self.engine.last_loss()

    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[float, float]:
        return self.engine.evaluate(x, y)

    def global_step_value(self) -> int:
        return int(round(float(self.global_step.item())))

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
