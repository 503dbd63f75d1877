"""Device-resident replica trainers: the fast tier used by ``bench.py`` and the scripts.

One process per GPU.  Every process is worker *r* AND the owner of PS shard *r*
(GPU-resident: the shard is a range of a flat fp32 slab in that GPU's HBM).

* ``SingleTrainer``   — no PS: fused forward/backward/SGD, steps captured in
  hipGraphs (reference tfsingle.py).
* ``SyncShardTrainer`` — SyncReplicasOptimizer semantics (reference
  tfdist_between_sync.py:65-68) without its queues: gradients are
  **reduce-scattered** to the shard owners (= every accumulator receives all N
  gradients), each owner applies the mean with the SGD kernel on its shard and
  bumps ``global_step`` once, and the new parameters are **all-gathered** back
  (= the pull of the next step).  ``replicas_to_aggregate == total_num_replicas
  == world``; the completion of the collective is the token.
* ``TowerTrainer``    — in-graph-replication semantics (one logical client, N
  towers): gradients are all-reduce-averaged over RCCL and every tower applies
  the identical update to its replica of the variables.
* ``LockstepAsyncRuleTrainer`` — asynchronous update *rule* (every worker's gradient
  is applied separately, unaveraged, to the owning shard; ``global_step``
  advances by N per round) executed in LOCK STEP with collectives: a reference
  point for the arithmetic, not an asynchronous transport.
* ``AsyncSendRecvTrainer`` — between-graph ASYNCHRONOUS parameter servers over
  two-sided library point-to-point calls (RCCL ``ncclSend`` / ``ncclRecv`` on the
  GPUs, gloo in CPU tests): one communicator and one service stream (or thread)
  per (worker, shard) pair, so a worker only ever waits for the shard owner's
  service of ITS OWN request.  The library baseline of the one-sided
  ``XgmiAsyncTrainer``.

Collectives go through ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI
on the GPUs, ``gloo`` in CPU tests).  The message is 318 KB, i.e. latency-bound:
one fused slab per collective, never one call per variable.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch

from ..models.mlp import MLPEngine
from ..utils.trace import span
from ..ops import hip_backend as hb
from ..ops import reference as ref
from ..variables import ShardLayout, assign_shards, mlp_var_specs


class _BaseTrainer:
    name = "base"

    def __init__(self, device, dataset, batch: int, lr: float, seed: int, rank: int = 0, world: int = 1,
                 chunk: int = 50, use_graph: bool = True, layout_mode: str = "flat", act: str = "sigmoid",
                 dims: Tuple[int, int, int] = (784, 100, 10), need_grads: bool = True):
        self.device = torch.device(device)
        self.ds = dataset
        self.rank, self.world = rank, world
        self.chunk = max(1, int(chunk))
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        specs = mlp_var_specs(*dims)
        assign_shards(specs, world)
        self.layout = ShardLayout(specs, world, mode=layout_mode)
        self.params = self.layout.new_slab(self.device)
        self.layout.initialize(self.params, seed)   # same seed everywhere -> identical replicas
        self.grads = self.layout.new_slab(self.device) if need_grads else None
        self.global_step = self.layout.view(self.params, "global_step").view(1)
        self.engine = MLPEngine(
            self.layout.mlp_views(self.params),
            self.layout.mlp_views(self.grads) if need_grads else None,
            batch=batch, lr=lr, act=act, global_step=None,
        )
        self.engine.bind_data(dataset.x_epoch, dataset.y_epoch)
        self._bound_set = getattr(dataset, "active", 0)
        self.lr = self.engine.lr
        self.steps_done = 0
        self._graphs = {}

    # -- to be provided ---------------------------------------------------------
    def _one_step(self, batch_index: int):
        raise NotImplementedError

    def describe(self) -> str:
        return self.name

    # -- common driver -------------------------------------------------------------
    def _schedule(self, cursor: int, n: int):
        """Chunk sizes run_steps() will use for n steps starting at ``cursor``."""
        bpe = self.engine.batches_per_epoch
        out = []
        while n > 0:
            k = min(n, self.chunk, bpe - cursor % bpe)
            out.append(k)
            cursor += k
            n -= k
        return out

    # -- dataset buffers (a double-buffered DeviceDataset alternates between two epoch-ordered copies) ----
    def _bind_set(self, index: int):
        if index == self._bound_set:
            return
        x, y, gram = self.ds.set_tensors(index)
        self.engine.bind_data(x, y)
        if getattr(self, "pipelined", False):
            self.engine.bind_gram(gram)
        self._bound_set = index

    def _set_at(self, cursor: int) -> int:
        """Dataset buffer that will be active when the step counter reads ``cursor``."""
        active = getattr(self.ds, "active", 0)
        if not getattr(self.ds, "double_buffer", False):
            return active
        bpe = self.engine.batches_per_epoch
        now = self.engine._host_cursor
        # the buffers flip at every epoch boundary crossed between now and then
        flips = (cursor // bpe) - (now // bpe) + (1 if (now % bpe == 0 and now > 0 and cursor >= now) else 0)
        return active ^ (flips & 1)

    def prepare(self, steps_hint=()):
        """Capture every chunk graph the hinted run_steps() calls will need, outside any timed region."""
        if not self.use_graph:
            return
        bpe = self.engine.batches_per_epoch
        cursor = self.engine._host_cursor
        parity = self.engine.parity
        keys = set()
        for n in steps_hint:
            for k in self._schedule(cursor, n):
                keys.add((cursor % bpe, k, parity, self._set_at(cursor)))
                cursor += k
                parity ^= k & 1
        live, live_set = self.engine.parity, self._bound_set
        for pos, k, par, st in sorted(keys):
            self._get_graph(pos, k, par, st)
        self._bind_set(live_set)
        self.engine.reset_step_workspace(live)

    # -- a chunk of steps: trainers that keep state in another form between the ends of a chunk hook in here ----
    def _chunk_begin(self, n: int):
        """Called before the first step of a chunk of ``n`` steps (eager, warm-up or under capture)."""

    def _chunk_end(self):
        """Called after the last step of a chunk."""

    def _steps(self, pos: int, n: int, like: int = None, repeat_first: bool = False):
        """``n`` steps from epoch position ``pos`` as one chunk.  ``like``: take the form a chunk of that many steps
        would take (warm-up steps have to launch the kernels of the chunk they precede); ``repeat_first``: every step
        on batch ``pos``."""
        self._chunk_begin(n if like is None else like)
        for i in range(n):
            self._one_step(pos if repeat_first else pos + i)
        self._chunk_end()

    def _run_eager(self, pos: int, n: int):
        self._steps(pos, n)
        self.engine.advance(n)

    def _get_graph(self, pos: int, n: int, parity: int, dset: int = None):
        dset = self._bound_set if dset is None else dset
        key = (pos, n, parity, dset)
        g = self._graphs.get(key)
        if g is None:
            self._bind_set(dset)
            with span("graph_capture", steps=n):
                g = self._capture(pos, n, parity)
            if os.environ.get("DTF_GRAPH_UPLOAD", "1") != "0":
                hb.graph_upload(g)          # the first replay may sit inside somebody's timed region
            self._graphs[key] = g
        return g

    def _snapshot(self):
        return (self.params.clone(), None if self.grads is None else self.grads.clone(),
                self.engine.loss_hist.clone(), self.engine.acc_hist.clone())

    def _restore(self, snap):
        self.params.copy_(snap[0])
        if self.grads is not None:
            self.grads.copy_(snap[1])
        self.engine.loss_hist.copy_(snap[2])
        self.engine.acc_hist.copy_(snap[3])

    def _capture(self, pos: int, n: int, parity: int):
        # Eager warm-up on a side stream (lazy module loads, communicator set-up),
        # with the training state restored afterwards.
        snap = self._snapshot()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._steps(pos, 3, like=n, repeat_first=True)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self._restore(snap)
        self.engine.reset_step_workspace(parity)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._steps(pos, n)             # the engine flips its parity per step
        self.engine.parity = parity         # capture did not execute anything
        return graph

    def run_steps(self, n: int):
        """n worker steps; reshuffles the resident set at every epoch boundary."""
        bpe = self.engine.batches_per_epoch
        for k in self._schedule(self.engine._host_cursor, n):
            cur = self.engine._host_cursor
            if cur % bpe == 0 and cur > 0:
                with span("shuffle_epoch"):
                    self.ds.shuffle_epoch()
            self._bind_set(getattr(self.ds, "active", 0))
            if self.use_graph:
                with span("chunk_replay", steps=k):
                    self._get_graph(cur % bpe, k, self.engine.parity).replay()
                self.engine.parity ^= k & 1
                self.engine._host_cursor += k
            else:
                with span("chunk_eager", steps=k):
                    self._run_eager(cur % bpe, k)
            self.steps_done += k
            if hasattr(self.ds, "stage_ahead"):
                self.ds.stage_ahead()       # host work for the NEXT reshuffle, behind the work queued above

    def evaluate(self, x, y):
        return self.engine.evaluate(x, y)

    def last_loss(self) -> float:
        return self.engine.last_loss()

    def fetch_window_async(self, first_step: int, count: int, want_history: bool = False):
        """Results of the window just run, without waiting for it: see MLPEngine.fetch_window_async.  The fourth
        value of ``result()`` is the shared step counter as it stood right after the window."""
        return self.engine.fetch_window_async(first_step, count, want_history, extra=self.global_step)

    def global_step_value(self) -> int:
        return int(round(float(self.global_step.item())))


class SingleTrainer(_BaseTrainer):
    name = "single"

    def __init__(self, *a, pipelined: bool = True, **kw):
        kw["need_grads"] = False
        super().__init__(*a, **kw)
        # fused path bumps global_step inside the head kernel
        self.engine.global_step = self.global_step
        self.engine._desc = None
        # One launch per step: step t's backward and step t+1's forward share a kernel; needs the Gram
        # blocks of consecutive mini-batches, which the resident dataset refreshes at every reshuffle.
        self.pipelined = False
        if (pipelined and self.device.type == "cuda" and hasattr(self.ds, "enable_pair_gram")
                and hb.mlp_chunk_pipelined_supported(self.engine.B, self.engine.K, self.engine.H, self.engine.C)
                and self.engine.batches_per_epoch >= 2):
            self.engine.bind_gram(self.ds.enable_pair_gram())
            self.pipelined = True

    def _one_step(self, batch_index: int):
        self.engine.step_fused(batch_index)

    def _run_eager(self, pos: int, n: int):
        if self.pipelined:
            self.engine.chunk_pipelined(pos, n)
            self.engine.advance(n)
        else:
            super()._run_eager(pos, n)

    def _get_graph(self, pos: int, n: int, parity: int, dset: int = None):
        return super()._get_graph(pos, n, 0 if self.pipelined else parity, dset)

    def _capture(self, pos: int, n: int, parity: int):
        if not self.pipelined:
            return super()._capture(pos, n, parity)
        return self.engine._capture_pipe(pos, n)     # chunk-local state only: the parity plays no role

    # -- one-launch chunks that chain -----------------------------------------------------------------------
    # A chunk whose successor is known to follow in the same epoch ends with the forward pass of the successor's first
    # batch (and W1 in both layouts), and a chunk that starts where such a chunk ended is nothing but its step launches:
    # no transposition launch, no stand-alone forward (12 us per chunk, which is what a 20-step run_steps() or the six
    # windows of an epoch would otherwise pay each time).  What is carried is tracked by the engine and dropped by
    # anything else that touches it; a graph is captured per (position, length, carry-in, carry-out).
    def _chains(self) -> bool:
        return self.pipelined and self.use_graph and os.environ.get("DTF_PIPE_CARRY", "1") != "0"

    def _pipe_graph(self, pos: int, n: int, cin: bool, cout: bool, dset: int, capture: bool = True):
        key = (pos, n, ("carry", cin, cout), dset)
        g = self._graphs.get(key)
        if g is None and capture:
            self._bind_set(dset)
            with span("graph_capture", steps=n):
                g = self.engine._capture_pipe(pos, n, cin, cout)
            self._graphs[key] = g
        return g

    def prepare(self, steps_hint=()):
        if not self._chains():
            return super().prepare(steps_hint)
        bpe = self.engine.batches_per_epoch
        cursor = self.engine._host_cursor
        carried = None          # every capture below drops what the engine carries, so the first chunk starts afresh
        keys = []
        for n in steps_hint:
            for k in self._schedule(cursor, n):
                pos = cursor % bpe
                cin, cout = carried == pos and pos > 0, pos + k < bpe
                key = (pos, k, cin, cout, self._set_at(cursor))
                if key not in keys:
                    keys.append(key)
                carried = pos + k if cout else None
                cursor += k
        live_set = self._bound_set
        for pos, k, cin, cout, st in keys:
            self._pipe_graph(pos, k, cin, cout, st)
        self._bind_set(live_set)

    def run_steps(self, n: int):
        if not self._chains():
            return super().run_steps(n)
        eng = self.engine
        bpe = eng.batches_per_epoch
        for k in self._schedule(eng._host_cursor, n):
            cur = eng._host_cursor
            if cur % bpe == 0 and cur > 0:
                with span("shuffle_epoch"):
                    self.ds.shuffle_epoch()
                eng.invalidate_carry()
            self._bind_set(getattr(self.ds, "active", 0))
            pos = cur % bpe
            cin, cout = eng.carried_to == pos and pos > 0, pos + k < bpe
            g = self._pipe_graph(pos, k, cin, cout, self._bound_set, capture=False) if cin else None
            if g is None:
                # (capturing runs warm-up steps over whatever was carried: a chunk captured now starts afresh)
                cin = False
                g = self._pipe_graph(pos, k, False, cout, self._bound_set)
            with span("chunk_replay", steps=k):
                g.replay()
            eng.note_carry(pos + k if cout else None)
            eng._host_cursor += k
            self.steps_done += k
            if hasattr(self.ds, "stage_ahead"):
                self.ds.stage_ahead()

    def describe(self):
        how = "1 launch/step (backward of step t + forward of step t+1 in one kernel)" if self.pipelined \
            else "2 fused launches/step"
        return f"single (no PS): {how}, " + ("hipGraph chunks of %d" % self.chunk if self.use_graph else "eager")


class _CollectiveTrainer(_BaseTrainer):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        import torch.distributed as dist

        self.dist = dist
        if self.world > 1 and not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised for world > 1")
        self.my_params = self.layout.shard_view(self.params, self.rank)
        self.my_grads = self.layout.shard_view(self.grads, self.rank)
        a0, b0 = self.layout.shard_range[self.rank]
        o, _ = self.layout.var_range["global_step"]
        self.owns_step = a0 <= o < b0

    def _apply(self, p, g, scale: float, bump: bool):
        if self.device.type == "cuda":
            hb.sgd_apply_(p, g, self.lr, scale=scale, zero_grad=False,
                          global_step=self.global_step if bump else None)
        else:
            ref.sgd_apply_(p, g, float(self.lr.item()), scale)
            if bump:
                self.global_step.add_(1.0)


class SyncShardTrainer(_CollectiveTrainer):
    name = "sync"

    def _one_step(self, batch_index: int):
        d = self.dist
        # gradients pre-scaled by 1/N so that the reduce-scatter SUM is the mean
        self.engine.step_grad(batch_index, scale=1.0 / self.world)
        if self.world > 1:
            d.reduce_scatter_tensor(self.my_grads, self.grads, op=d.ReduceOp.SUM)
        self._apply(self.my_params, self.my_grads, 1.0, bump=self.owns_step)
        if self.world > 1:
            d.all_gather_into_tensor(self.params, self.my_params)

    def describe(self):
        return (f"sync-replicas dp{self.world}: PS shard per GPU, reduce-scatter -> sharded SGD -> all-gather "
                f"(replicas_to_aggregate={self.world})" + (", hipGraph" if self.use_graph else ", eager"))


class TowerTrainer(_CollectiveTrainer):
    name = "towers"

    def _one_step(self, batch_index: int):
        d = self.dist
        self.engine.step_grad(batch_index, scale=1.0 / self.world)
        if self.world > 1:
            d.all_reduce(self.grads, op=d.ReduceOp.SUM)
        self._apply(self.params, self.grads, 1.0, bump=True)

    def describe(self):
        return f"in-graph towers dp{self.world}: all-reduce gradient averaging, replicated apply" + \
            (", hipGraph" if self.use_graph else ", eager")


class LockstepAsyncRuleTrainer(_CollectiveTrainer):
    """Async update RULE in lock step: shard owners apply each worker's gradient separately, but all workers
    move through all-to-all / all-gather together.  Not an asynchronous transport (see AsyncSendRecvTrainer,
    XgmiAsyncTrainer for those)."""

    name = "async_lockstep"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        n = self.layout.shard_numel(self.rank)
        self.inbox = torch.zeros((self.world, n), dtype=torch.float32, device=self.device)

    def _one_step(self, batch_index: int):
        d = self.dist
        self.engine.step_grad(batch_index, scale=1.0)
        if self.world > 1:
            d.all_to_all_single(self.inbox.view(-1), self.grads)
            for w in range(self.world):   # N separate unaveraged applies, N global_step bumps
                self._apply(self.my_params, self.inbox[w], 1.0, bump=self.owns_step)
            d.all_gather_into_tensor(self.params, self.my_params)
        else:
            self._apply(self.my_params, self.my_grads, 1.0, bump=self.owns_step)

    def describe(self):
        return (f"async-rule (lock step) dp{self.world}: all-to-all gradient push, per-worker unaveraged apply on "
                f"the owning shard, all-gather pull" + (", hipGraph" if self.use_graph else ", eager"))


class AsyncSendRecvTrainer(_CollectiveTrainer):
    """Asynchronous parameter servers over two-sided point-to-point calls (``--mode async_rccl``).

    Rank r is worker r and owns shard r of the flat slab.  One worker step (reference tfdist_between.py:56-58,
    65-66: pull, compute, push, apply on the variable's device):

        forward + backward on the worker's private copy of the variables
        for every shard k:  send grad[range k] -> owner k ;  recv fresh params[range k] <- owner k

    and, independently, the owner of shard k serves every worker w on a SERVICE lane of its own:

        recv g <- w ;  shard_k -= lr * g  (SGD kernel on the service stream; global_step += 1 on the shard that
        holds it) ;  send shard_k -> w

    Every ordered (worker, shard) pair has its own 2-rank communicator, so a request is matched by nothing but
    the owner's service lane for that worker: no worker waits for another worker, and the parameters a worker
    receives contain whatever other workers' updates the shard had absorbed by then (staleness <= N - 1
    updates, unbounded in wall time).  ``use_locking=False`` (the reference's default) lets the service lanes'
    whole-vector updates of one shard interleave; ``use_locking=True`` applies with float atomics (GPU) or under
    a mutex (CPU) so that no update is lost.

    With RCCL the calls are stream-ordered: ``run_steps(n)`` enqueues the worker's n steps on the current
    stream and the n services per worker on side streams and returns; nothing blocks the host.  With gloo the
    calls block, so the services run in threads for the duration of ``run_steps``.  Every rank must call
    ``run_steps`` with the same n (each owner serves exactly n requests per worker).
    """

    name = "async_rccl"

    def __init__(self, *a, use_locking: bool = False, **kw):
        kw["use_graph"] = False          # point-to-point calls on several communicators: eager launches
        super().__init__(*a, **kw)
        d = self.dist
        self.use_locking = bool(use_locking)
        self.backend = d.get_backend() if self.world > 1 else "none"
        self.stream_ordered = self.backend == "nccl"
        # the authoritative copy of this rank's shard (the worker trains on self.params, a private copy)
        self.shard = self.my_params.clone()
        a0, b0 = self.layout.shard_range[self.rank]
        o, _ = self.layout.var_range["global_step"]
        self.shard_step = self.shard[o - a0:o - a0 + 1] if self.owns_step else None
        n = self.layout.shard_numel(self.rank)
        self.inbox = torch.zeros((self.world, n), dtype=torch.float32, device=self.device)
        self.outbox = torch.zeros((self.world, n), dtype=torch.float32, device=self.device)
        # pair[(w, k)]: worker w <-> owner k; every rank creates every group, in the same order.  (Creating only the
        # 2 (N - 1) groups a rank belongs to -- new_group(..., use_local_synchronization=True) -- does not work here:
        # such groups are named by a hash of their SORTED ranks, so the two communicators a pair needs, one per
        # direction, get the same name and the second creation hangs; seen with gloo at 4 and 8 ranks.  With RCCL a
        # group is only a record until its first call, so the N (N - 1) records cost little: comm_setup_s says how much.)
        self.pair = {}
        import time as _time

        t_comm = _time.perf_counter()
        if self.world > 1:
            for w in range(self.world):
                for k in range(self.world):
                    if w != k:
                        self.pair[(w, k)] = d.new_group([w, k])
        self.comm_setup_s = _time.perf_counter() - t_comm
        self._lock = None
        if not self.stream_ordered:
            import threading

            self._lock = threading.Lock()
        self._side = {}
        if self.stream_ordered:
            self._side = {w: torch.cuda.Stream(device=self.device) for w in range(self.world) if w != self.rank}
        self.served = 0

    # -- the owner's side -----------------------------------------------------------------------------
    def _apply_to_shard(self, g):
        if self.device.type == "cuda":
            if self.use_locking:
                hb.sgd_apply_atomic_(self.shard, g, self.lr, 1.0, self.shard_step)
            else:
                hb.sgd_apply_(self.shard, g, self.lr, 1.0, False, self.shard_step)
        elif self._lock is not None and self.use_locking:
            with self._lock:
                ref.sgd_apply_(self.shard, g, float(self.lr.item()), 1.0)
                if self.shard_step is not None:
                    self.shard_step.add_(1.0)
        else:
            ref.sgd_apply_(self.shard, g, float(self.lr.item()), 1.0)
            if self.shard_step is not None:
                self.shard_step.add_(1.0)
        self.served += 1

    def _serve_once(self, w: int):
        d, grp = self.dist, self.pair[(w, self.rank)]
        d.recv(self.inbox[w], src=w, group=grp)
        self._apply_to_shard(self.inbox[w])
        if self.use_locking or not self.stream_ordered:
            self.outbox[w].copy_(self.shard)      # a snapshot: the shard keeps changing under other lanes
            d.send(self.outbox[w], dst=w, group=grp)
        else:
            # unlocked (the reference's default): the answer is read straight out of the live shard, as an unlocked
            # read of a variable that other lanes are updating is in the reference -- no staging copy
            d.send(self.shard, dst=w, group=grp)

    def _service_thread(self, w: int, n: int, errors: list):
        try:
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
            for _ in range(n):
                self._serve_once(w)
        except Exception as e:  # noqa: BLE001 — reported by run_steps on the main thread
            errors.append((w, e))

    # -- the worker's side ----------------------------------------------------------------------------
    def _one_step(self, batch_index: int):
        d = self.dist
        self.engine.step_grad(batch_index, scale=1.0)
        if self.stream_ordered:
            # per peer ONE coalesced call (ncclGroupStart / End): the push of this worker's gradient range and the
            # receive of the owner's answer, which the owner sends after it has applied that gradient
            works = []
            for k in range(self.world):
                if k == self.rank:
                    continue
                a, b = self.layout.shard_range[k]
                grp = self.pair[(self.rank, k)]
                works += d.batch_isend_irecv([d.P2POp(d.isend, self.grads[a:b], k, group=grp),
                                              d.P2POp(d.irecv, self.params[a:b], k, group=grp)])
            self._apply_to_shard(self.my_grads)          # own shard: no communicator between a rank and itself
            self.my_params.copy_(self.shard)
            for wk in works:
                wk.wait()                                # (stream-ordered: the host does not block)
            return
        for k in range(self.world):
            a, b = self.layout.shard_range[k]
            if k == self.rank:
                continue
            grp = self.pair[(self.rank, k)]
            d.send(self.grads[a:b], dst=k, group=grp)
        # own shard: no communicator between a rank and itself
        self._apply_to_shard(self.my_grads)
        self.my_params.copy_(self.shard)
        for k in range(self.world):
            a, b = self.layout.shard_range[k]
            if k == self.rank:
                continue
            d.recv(self.params[a:b], src=k, group=self.pair[(self.rank, k)])

    def _run_eager(self, pos: int, n: int):
        if self.world == 1:
            return super()._run_eager(pos, n)
        if self.stream_ordered:
            main = torch.cuda.current_stream(self.device)
            for s in self._side.values():
                s.wait_stream(main)
            for i in range(n):
                self._one_step(pos + i)
                for w, s in self._side.items():
                    with torch.cuda.stream(s):
                        self._serve_once(w)
            self.engine.advance(n)
            return
        import threading

        errors: list = []
        threads = [threading.Thread(target=self._service_thread, args=(w, n, errors), daemon=True)
                   for w in range(self.world) if w != self.rank]
        for t in threads:
            t.start()
        for i in range(n):
            self._one_step(pos + i)
        for t in threads:
            t.join()
        if errors:
            raise RuntimeError(f"service lane for worker {errors[0][0]} failed: {errors[0][1]}")
        self.engine.advance(n)

    def quiesce(self):
        """Wait until every request enqueued so far has been served (side streams drained)."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if self.world > 1:
            self.dist.barrier()

    def evaluate(self, x, y):
        self._refresh()
        return self.engine.evaluate(x, y)

    def _refresh(self):
        """Private copy <- the authoritative shards (a collective: every rank calls it)."""
        self.quiesce()
        self.my_params.copy_(self.shard)
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.params, self.shard.clone())

    def global_step_value(self) -> int:
        self._refresh()
        return int(round(float(self.global_step.item())))

    def host_calls_per_step(self) -> int:
        """Library calls a rank issues per step: as a worker (one coalesced send + receive per peer with RCCL, two calls
        per peer otherwise) and as the owner of its shard (a receive and a send per served worker)."""
        peers = self.world - 1
        return peers * (1 if self.stream_ordered else 2) + 2 * peers

    def describe(self):
        lane = "side HIP stream" if self.stream_ordered else "service thread"
        return (f"async PS dp{self.world} over {self.backend} send/recv (the two-sided library yardstick of the one-sided "
                f"shards): one communicator and one {lane} per (worker, shard) pair, owner applies with the SGD kernel "
                f"({'atomic adds' if self.use_locking else 'unlocked whole-vector update'}), eager, "
                f"{self.host_calls_per_step()} library calls per rank and step")


class XgmiTowerTrainer(_BaseTrainer):
    """Synchronous replicas whose gradient exchange is ONE kernel over peer-mapped HBM.

    Same update as ``TowerTrainer`` (the mean of N gradients applied once to every replica), but the
    all-reduce and the SGD apply are a single launch that reads the peers' gradient slabs directly over
    xGMI (``csrc/xgmi.hip``, ``parallel/xgmi.py``): 3 launches per step and no library collective.
    The sum is taken in rank order on every replica, so the replicas stay bit-identical;
    ``replica_checksum()`` lets a caller verify exactly that.
    """

    name = "xgmi"

    def __init__(self, *a, group=None, local_peers: bool = False, uncached_grads: bool = False,
                 timeout_ms: int = 20000, required: int = 0, **kw):
        kw["need_grads"] = False
        super().__init__(*a, **kw)
        if self.device.type != "cuda":
            raise RuntimeError("the peer-mapped exchange is a GPU path")
        from .xgmi import XgmiExchange

        if self.layout.var_range["global_step"][0] >= 4:
            raise RuntimeError("global_step must live in the first 16 bytes of the slab")
        # backup replicas (SyncReplicasOptimizer with replicas_to_aggregate < total_num_replicas): an update is the
        # mean of the first ``required`` gradients to arrive; 0 = every replica's
        if not 0 <= int(required) <= self.world:
            raise ValueError("required must lie in 0 .. world")
        self.required = 0 if int(required) == self.world else int(required)
        self.xg = XgmiExchange(self.rank, self.world, self.params.numel(), self.device, group=group,
                               uncached_grads=uncached_grads, timeout_ms=timeout_ms, connect=not local_peers)
        self.engine.bind_grad_slots([self.layout.mlp_views(g) for g in self.xg.grads])
        self.group = group

    def _one_step(self, batch_index: int):
        # The slot follows the engine's step parity: consecutive steps alternate between the two slabs,
        # which is what lets a peer still read step t's gradient while step t+1 is being computed.
        slot = self.engine.parity
        self.engine.step_grad(batch_index, scale=1.0 / (self.required or self.world), slot=slot)
        self.xg.apply(self.params, self.lr, slot, self.global_step, required=self.required)

    def _capture(self, pos: int, n: int, parity: int):
        # Warm-up exchanges are real exchanges (every replica performs the same number of them) whose
        # results are rolled back; two of them, so that the slots keep alternating.
        snap = self._snapshot()
        self.engine.reset_step_workspace(parity)
        self._steps(pos, 2, like=n, repeat_first=True)
        torch.cuda.synchronize(self.device)
        self._restore(snap)
        self.engine.reset_step_workspace(parity)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._steps(pos, n)
        self.engine.parity = parity
        return graph

    def replica_checksum(self) -> int:
        """Bitwise checksum of the local replica of the variables (equal on all replicas when in sync)."""
        return int(self.params.view(torch.int32).to(torch.int64).sum().item())

    def healthy(self) -> bool:
        return not self.xg.failed()

    def close(self):
        self._graphs.clear()
        self.xg.close()

    def describe(self):
        what = (f"sum of {self.world} slabs" if not self.required else
                f"sum of the first {self.required} of {self.world} slabs to arrive,")
        return (f"sync towers dp{self.world}: one-kernel gradient exchange over peer-mapped HBM "
                f"(announce, wait, {what} in rank order, SGD apply), 3 launches/step"
                + (", hipGraph chunks of %d" % self.chunk if self.use_graph else ", eager"))


class XgmiFusedTrainer(_BaseTrainer):
    """Synchronous replicas whose gradient exchange happens INSIDE the backward launch: 2 launches per step.

    Same update as ``XgmiTowerTrainer`` (the mean of N gradients applied once to every replica, sums in rank
    order, replicas bit-identical).  Every backward thread stores its own element of the 1/N-scaled gradient as
    an 8-byte {exchange number, value} word straight into every peer's inbox, polls its own inbox for the
    peers' words and applies the in-place SGD update (``csrc/mlp_step2.hip``, ``bwd_tile16_xg_kernel``;
    ``parallel/xgmi.py::PeerInbox``).  No flag array, no barrier, no separate exchange kernel: the exchange adds
    one xGMI write latency to the backward.
    """

    name = "xgmi2"

    def __init__(self, *a, group=None, timeout_ms: int = 20000, compact: bool = False, local_peers: bool = False,
                 two_phase: bool = False, **kw):
        kw["need_grads"] = False
        super().__init__(*a, **kw)
        if self.device.type != "cuda":
            raise RuntimeError("the peer-mapped exchange is a GPU path")
        from .xgmi import PeerInbox

        self.engine.global_step = self.global_step
        self.engine._desc = None
        self.compact = bool(compact)
        self.two_phase = bool(two_phase)
        self.name = "xgmi2p" if self.two_phase else ("xgmi2c" if self.compact else "xgmi2")
        self.xg = PeerInbox(self.rank, self.world, self.params.numel(), self.device, group=group,
                            timeout_ms=timeout_ms, compact=self.compact, connect=not local_peers,
                            two_phase=self.two_phase)
        self._xg_offs = {role: self.layout.var_range[name][0] for role, name in
                         (("W1", "weights/Variable"), ("b1", "biases/Variable"), ("W2", "weights/Variable_1"),
                          ("b2", "biases/Variable_1"))}
        # Between the ends of a chunk W1 lives TRANSPOSED ([H][K]): the forward tiles read 8 contiguous rows of that
        # image with 16-byte loads instead of 784 pieces of 32 bytes of W1[K][H], the backward updates it in place;
        # one transposition launch at either end of the chunk (every replica takes the same decision: same chunk
        # lengths everywhere).  DTF_XG_W1T=0: W1 as it is.
        self._W1 = self.layout.mlp_views(self.params)["W1"]
        self.use_w1t = (os.environ.get("DTF_XG_W1T", "1") != "0" and self.engine.K % 4 == 0 and self.engine.K <= 1024
                        and self._W1.data_ptr() % 16 == 0)
        self._w1t = torch.empty((self.engine.H, self.engine.K), dtype=torch.float32, device=self.device) \
            if self.use_w1t else None
        self._in_w1t = False
        self._xg_desc = self._xg_desc_t = None
        if not local_peers:
            self._make_descriptors()

    W1T_MIN_STEPS = 8      # two transposition launches per chunk: shorter chunks keep W1 as it is

    def _make_descriptors(self):
        self._xg_desc = self.xg.step_descriptor(self._xg_offs)
        if self.use_w1t:
            self._xg_desc_t = self.xg.step_descriptor(self._xg_offs)
            self._xg_desc_t.w1t = self._w1t.data_ptr()

    def connect_local(self, towers):
        """Replicas driven by ONE process (in-graph towers): wire the inboxes with plain pointers."""
        self.xg.connect_local([t.xg for t in towers if t is not self])
        self._make_descriptors()

    def _chunk_begin(self, n: int):
        self._in_w1t = self.use_w1t and n >= self.W1T_MIN_STEPS
        if self._in_w1t:
            hb.transpose(self._W1, self._w1t)

    def _chunk_end(self):
        if self._in_w1t:
            hb.transpose(self._w1t, self._W1)
            self._in_w1t = False

    def _one_step(self, batch_index: int):
        self.engine.step_xg(batch_index, 1.0 / self.world, self._xg_desc_t if self._in_w1t else self._xg_desc)

    _capture = XgmiTowerTrainer._capture
    replica_checksum = XgmiTowerTrainer.replica_checksum
    healthy = XgmiTowerTrainer.healthy
    close = XgmiTowerTrainer.close

    def describe(self):
        if self.two_phase:
            return (f"sync towers dp{self.world}: two-phase gradient exchange inside the backward launch over "
                    f"peer-mapped HBM (partials to the element's owner, the PS shard of that GPU, which sums in rank "
                    f"order, applies SGD and returns the new value to every peer: 2/{self.world} of the slab per link), "
                    f"2 launches/step" + (", hipGraph chunks of %d" % self.chunk if self.use_graph else ", eager"))
        words = ("bare 4-byte values, unwritten words marked by a reserved NaN pattern," if self.compact
                 else "{number, value} words")
        return (f"sync towers dp{self.world}: gradient exchange inside the backward launch ({words} "
                f"stored into the peers' inboxes over peer-mapped HBM, rank-ordered sum, in-place SGD), "
                f"2 launches/step" + (", hipGraph chunks of %d" % self.chunk if self.use_graph else ", eager"))


class XgmiAsyncTrainer(_BaseTrainer):
    """Between-graph ASYNCHRONOUS workers on N GPUs with GPU-resident PS shards reached one-sidedly.

    Rank r is worker r and owns PS shard r (``parallel/xgmi.py::PeerShards``).  A step is

        pull   every shard -> private slab        (1 launch, remote shards over xGMI)
        fwd + bwd on the worker's own mini-batch   (2 launches)
        push   shard -= lr * g on every shard      (in place through the mappings, global_step += 1)

    The push is either a launch of its own (read-modify-write of whole vectors, 4 launches per step) or, with
    ``fused_push`` (the default), part of the backward launch: every backward thread applies ``-lr * g`` of its
    own gradient element to the owning shard as one fire-and-forget float atomic (``dtf_mlp_step_async``,
    3 launches per step, and concurrent pushes of several workers lose no update).  Either way there is no barrier, flag or matching receive anywhere: workers run at their own pace against whatever the
    shards hold when they pull, and concurrent pushes interleave without a lock — the reference's default
    (GradientDescentOptimizer, use_locking=False, tfdist_between.py:65-66).  ``global_step`` counts every
    worker's push, so it advances by N per "round" (README.md:70-72 of the reference observes the same).
    """

    name = "xgmi_async"

    def __init__(self, *a, group=None, fused_push="auto", fold_pull="auto", **kw):
        super().__init__(*a, **kw)
        if self.device.type != "cuda":
            raise RuntimeError("the peer-mapped shards are a GPU path")
        from .xgmi import PeerShards

        self.group = group
        # "auto" (both): start from the two-launch form and, with peers, let _choose_form measure; a bool forces a form
        calibrate = fused_push == "auto" and fold_pull == "auto"
        fused_push = True if fused_push == "auto" else fused_push
        fold_pull = True if fold_pull == "auto" else fold_pull
        self.fused_push = bool(fused_push)
        # fold_pull: no pull launch either -- the forward reads W1 / b1 / W2 / b2 from the shards through the
        # mappings and snapshots W2 / b2 for the backward (2 launches per step; needs the fused push)
        self.fold_pull = bool(fold_pull) and self.fused_push and os.environ.get("DTF_ASYNC_FOLD_PULL", "1") != "0"
        chunk = self.layout.shard_numel(0)
        self.shards = PeerShards(self.rank, self.world, chunk, self.device, group=group)
        if self.shards.total != self.params.numel():
            raise RuntimeError("layout and shard sizes disagree")
        self.shards.publish(self.params)          # same seed everywhere: every rank publishes its own range
        self.step_index = self.layout.var_range["global_step"][0]
        # load every kernel of the step outside any capture without changing a shard (lr = 0, no count)
        zero_lr = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.shards.pull(self.params)
        self.engine.step_grad(0, 1.0)
        self.shards.push_sgd(self.grads, zero_lr, -1)
        torch.cuda.synchronize(self.device)
        self.engine.reset_step_workspace(0)
        if self.fused_push:
            offs = {role: self.layout.var_range[name][0] for role, name in
                    (("W1", "weights/Variable"), ("b1", "biases/Variable"), ("W2", "weights/Variable_1"),
                     ("b2", "biases/Variable_1"))}
            # same for the pushing backward: the engine reads lr from its own tensor, so a zero there for one step
            keep = self.engine.lr.clone()
            self.engine.lr.zero_()
            if self.fold_pull:
                try:
                    self.engine.step_async(0, self.shards.step_descriptor(offs, -1, read_shards=True))
                except hb.HipKernelError:       # shards too short for the forward's range lookup: keep the pull launch
                    self.fold_pull = False
            self.engine.step_async(0, self.shards.step_descriptor(offs, -1))
            self.engine.lr.copy_(keep)
            torch.cuda.synchronize(self.device)
            self.engine.reset_step_workspace(0)
            self._push_desc = self.shards.step_descriptor(offs, self.step_index, read_shards=self.fold_pull)
            self.form_probe_us = None
            if (calibrate and self.world > 1 and self.use_graph and os.environ.get("DTF_ASYNC_CALIBRATE", "1") != "0"):
                self._choose_form(offs)

    def _choose_form(self, offs):
        """Which form of the step is fastest HERE is measured, not assumed: with the shards behind xGMI links a forward
        that reads the variables from them issues many small remote reads where a pull moves each shard once in bulk,
        and the atomics of the pushing backward contend differently from a push kernel's whole vectors.  The forms --
        forward from the shards + pushing backward (2 launches), pull + pushing backward (3), pull + gradient launches +
        push kernel (4) -- are each captured as a chunk with a learning rate of zero and no step count (the shards receive
        ``+ 0`` while the other workers go on), replayed and timed with device events; every worker decides for itself."""
        want = (self.fold_pull, self.fused_push)
        n = min(self.chunk, self.engine.batches_per_epoch, 20)
        times = {}
        keep_lr = self.engine.lr.clone()
        self.engine.lr.zero_()
        self._dry = True
        try:
            for name, fold, fused in (("forward from the shards", True, True), ("pull, pushing backward", False, True),
                                      ("pull, push kernel", False, False)):
                if fold and not self.fold_pull:
                    continue
                self.fold_pull, self.fused_push = fold, fused
                self._push_desc = self.shards.step_descriptor(offs, -1, read_shards=fold) if fused else None
                g = self._capture(0, n, 0)
                start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(2):
                    g.replay()
                start.record()
                for _ in range(6):
                    g.replay()
                stop.record()
                stop.synchronize()
                times[name] = (1000.0 * start.elapsed_time(stop) / (6 * n), fold, fused)
                del g
        except Exception:  # noqa: BLE001 -- a runtime that cannot capture these steps: keep the default form
            times = {}
        finally:
            self._dry = False
            self.engine.lr.copy_(keep_lr)
            torch.cuda.synchronize(self.device)
            self.engine.reset_step_workspace(0)
        if times:
            best = min(times.values(), key=lambda v: v[0])
            self.fold_pull, self.fused_push = best[1], best[2]
            self.form_probe_us = {k: round(v[0], 2) for k, v in times.items()}
        else:
            self.fold_pull, self.fused_push = want
        self._push_desc = (self.shards.step_descriptor(offs, self.step_index, read_shards=self.fold_pull)
                           if self.fused_push else None)

    _dry = False

    def _one_step(self, batch_index: int):
        if self.fold_pull:
            with span("pull_step_push_apply"):
                self.engine.step_async(batch_index, self._push_desc)
            return
        with span("pull"):
            self.shards.pull(self.params)
        if self.fused_push:
            with span("step_push_apply"):
                self.engine.step_async(batch_index, self._push_desc)
            return
        with span("step"):
            self.engine.step_grad(batch_index, 1.0)
        with span("push_apply"):
            self.shards.push_sgd(self.grads, self.lr, -1 if self._dry else self.step_index)

    def _capture(self, pos: int, n: int, parity: int):
        # No warm-up replay: a step writes the shared shards, which cannot be rolled back.
        self.engine.reset_step_workspace(parity)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._steps(pos, n)
        self.engine.parity = parity
        return graph

    def evaluate(self, x, y):
        self.shards.pull(self.params)
        return self.engine.evaluate(x, y)

    def global_step_value(self) -> int:
        self.shards.pull(self.params)
        return int(round(float(self.global_step.item())))

    def fetch_window_async(self, first_step: int, count: int, want_history: bool = False):
        if self.fold_pull:          # the private copy (and the step counter in it) is not refreshed by the steps
            self.shards.pull(self.params)
        return super().fetch_window_async(first_step, count, want_history)

    def close(self):
        self._graphs.clear()
        self.shards.close()

    def describe(self):
        probe = getattr(self, "form_probe_us", None)
        probed = (" [measured us/step: " + ", ".join(f"{k} {v}" for k, v in probe.items()) + "]") if probe else ""
        if self.fold_pull:
            return (f"async PS dp{self.world}: shard per GPU in HBM, forward that reads the variables from the shards -> "
                    f"backward that pushes -lr*g per element into the owning shard with float atomics (2 launches/step) "
                    f"through peer mappings, no barrier" + probed
                    + (", hipGraph chunks of %d" % self.chunk if self.use_graph else ", eager"))
        push = ("forward -> backward that pushes -lr*g per element into the owning shard with float atomics "
                "(3 launches/step)" if self.fused_push else
                "2 launches -> one-sided lock-free SGD push (1 launch)")
        return (f"async PS dp{self.world}: shard per GPU in HBM, one-sided pull (1 launch) -> {push} "
                f"through peer mappings, no barrier" + probed
                + (", hipGraph chunks of %d" % self.chunk if self.use_graph else ", eager"))


def build_replica_trainer(mode: str, **kw):
    cls = {"single": SingleTrainer, "sync": SyncShardTrainer, "towers": TowerTrainer,
           "async_lockstep": LockstepAsyncRuleTrainer, "async": LockstepAsyncRuleTrainer,
           "async_rccl": AsyncSendRecvTrainer, "xgmi": XgmiTowerTrainer, "xgmi2": XgmiFusedTrainer,
           "xgmi2c": XgmiFusedTrainer, "xgmi2p": XgmiFusedTrainer, "xgmi_async": XgmiAsyncTrainer}[mode]
    if mode == "xgmi2c":        # the exchange inside the backward launch with 4-byte words
        kw["compact"] = True
    if mode == "xgmi2p":        # ... through the owner of every element (reduce-scatter shaped)
        kw["two_phase"] = True
    if mode == "single" and kw.get("world", 1) != 1:
        raise ValueError("single mode is one process")
    return cls(**kw)
