"""Variable stores: where a session's variables live and how gradients reach them.

``LocalVariableStore``   single process (reference tfsingle.py): the slab on the training
                         device IS the variable storage.
``RemoteVariableStore``  between-graph replication (reference tfdist_between*.py): shard k
                         is a float buffer inside ps task k's native server
                         (runtime/ps_server.cpp).  Pull = read every shard into the worker's
                         local slab; async push = server-side ``p += -lr * g`` per shard
                         (ApplyGradientDescent executed on the PS, lock-free unless
                         ``use_locking``); sync push = conditional accumulators + token queue
                         (SyncReplicasOptimizer).

Message sizes for the reference model: 318,040 B pulled and pushed per step and worker,
split by the round-robin placement (2 PS: 4,044 B to ps0, 314,000 B to ps1).
"""
from __future__ import annotations

import struct
from typing import List, Optional

import numpy as np
import torch

from ..ops import hip_backend as hb
from ..variables import ShardLayout

INIT_KEY = "__variables_initialized__"
TOKEN_Q = "__sync_token_q__"          # + "/<worker index>": every worker owns a token queue
FINISHED_KEY = "__workers_finished__"   # counter on ps task 0: workers whose training loop has ended


def token_key(worker: int) -> str:
    return f"{TOKEN_Q}/{worker}"


def spread_tokens(n: int, n_workers: int) -> List[int]:
    """How many of n tokens each worker's queue receives (as even as possible)."""
    return [n // n_workers + (1 if w < n % n_workers else 0) for w in range(n_workers)]
SHARD_BUF = "shard"
SHARD_ACC = "shard_acc"


class LocalVariableStore:
    is_remote = False

    def __init__(self, layout: ShardLayout, device: torch.device):
        self.layout = layout
        self.device = device
        self.slab = layout.new_slab(device)
        self._initialized = False
        o, _ = layout.var_range.get("global_step", (None, None))
        self._gs_off = o
        self._lr_t: Optional[torch.Tensor] = None

    def bind_local(self) -> torch.Tensor:
        return self.slab

    def initialized(self) -> bool:
        return self._initialized

    def wait_initialized(self, timeout: float) -> bool:
        return self._initialized

    def initialize(self, seed: int, only: Optional[List[str]] = None):
        if only is None:
            self.layout.initialize(self.slab, seed)
        else:
            from ..variables import fill_initial_value

            for s in self.layout.specs:
                if s.name in only:
                    fill_initial_value(self.layout.view(self.slab, s.name).view(-1), s, seed)
        self._initialized = True

    def pull(self, local: torch.Tensor, locked: bool = False):
        return  # the local slab is the storage

    def publish(self, local: torch.Tensor):
        if local.data_ptr() != self.slab.data_ptr():
            self.slab.copy_(local)

    def write_all(self, local: torch.Tensor):
        if local.data_ptr() != self.slab.data_ptr():
            self.slab.copy_(local)
        self._initialized = True

    def push_apply(self, grads: torch.Tensor, lr: float, use_locking: bool = False, bump_step: bool = True):
        gs = None
        if bump_step and self._gs_off is not None:
            gs = self.slab[self._gs_off:self._gs_off + 1]
        if self.slab.is_cuda:
            if self._lr_t is None or float(self._lr_cached) != float(lr):
                self._lr_t = torch.full((1,), float(lr), dtype=torch.float32, device=self.device)
                self._lr_cached = float(lr)
            hb.sgd_apply_(self.slab, grads, self._lr_t, 1.0, False, gs)
        else:
            self.slab.sub_(grads, alpha=float(lr))
            if gs is not None:
                gs.add_(1.0)

    def global_step(self) -> int:
        if self._gs_off is None:
            return 0
        return int(round(float(self.slab[self._gs_off].item())))

    def close(self):
        pass


class RemoteVariableStore:
    """Client side of the parameter servers of a cluster (one instance per worker session)."""

    is_remote = True

    def __init__(self, layout: ShardLayout, server, device: torch.device):
        self.layout = layout
        self.server = server
        self.device = device
        self.n = layout.num_shards
        if self.n != server.num_ps:
            raise ValueError("layout shard count must equal the number of ps tasks")
        self.timeout = server.config.operation_timeout_in_ms / 1000.0
        self._host = np.zeros(layout.total, dtype=np.float32)     # staging for device <-> socket
        self._host_t = torch.from_numpy(self._host)
        self._ghost = np.zeros(layout.total, dtype=np.float32)
        self._ghost_t = torch.from_numpy(self._ghost)
        self.local_step = 0
        self._gs_shard = layout.owner_of("global_step") if "global_step" in layout.var_range else 0
        o, _ = layout.var_range.get("global_step", (0, 1))
        self._gs_off_in_shard = o - layout.shard_range[self._gs_shard][0]
        self.dropped = 0
        self._agg_pos = 0
        self.n_workers = max(1, server.num_workers)
        self.widx = server.task_index

    def _c(self, k: int):
        return self.server.client("ps", k)

    # -- lifecycle -------------------------------------------------------------------
    def initialized(self) -> bool:
        try:
            self._c(0).get(INIT_KEY, timeout=0.0)
            return True
        except TimeoutError:
            return False

    def wait_initialized(self, timeout: float) -> bool:
        try:
            self._c(0).get(INIT_KEY, timeout=timeout)
            return True
        except TimeoutError:
            return False

    def _create_buffers(self):
        for k in range(self.n):
            self._c(k).buf_create(SHARD_BUF, self.layout.shard_numel(k))

    def initialize(self, seed: int, only: Optional[List[str]] = None):
        """Chief only: run the initializers and store the result on every shard."""
        self._create_buffers()
        tmp = torch.zeros(self.layout.total, dtype=torch.float32, device=self.device)
        self.layout.initialize(tmp, seed)
        self.write_all(tmp)

    def publish(self, local: torch.Tensor):
        """Overwrite the shards with ``local`` (a replica that was trained elsewhere); no initialisation side effects."""
        self._host_t.copy_(local)
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b > a:
                self._c(k).buf_write(SHARD_BUF, self._host[a:b])

    def write_all(self, local: torch.Tensor):
        self._create_buffers()
        self._host_t.copy_(local)
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b > a:
                self._c(k).buf_write(SHARD_BUF, self._host[a:b])
        self._c(0).set(INIT_KEY, b"1")

    # -- data plane -----------------------------------------------------------------------
    def pull(self, local: torch.Tensor, locked: bool = False):
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b > a:
                self._c(k).buf_read(SHARD_BUF, b - a, out=self._host[a:b], locked=locked)
        local.copy_(self._host_t, non_blocking=False)

    def _stage_grads(self, grads: torch.Tensor):
        self._ghost_t.copy_(grads)

    def push_apply(self, grads: torch.Tensor, lr: float, use_locking: bool = False, bump_step: bool = True):
        """Asynchronous update: every shard applies this worker's gradient immediately."""
        self._stage_grads(grads)
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b > a:
                # (the shard that holds global_step counts the update in the same request)
                self._c(k).buf_axpy(SHARD_BUF, -float(lr), self._ghost[a:b], locked=use_locking,
                                    bump=self._gs_off_in_shard if (bump_step and k == self._gs_shard) else None)

    # -- SyncReplicas protocol ---------------------------------------------------------------
    def push_accumulate(self, grads: torch.Tensor) -> bool:
        """Worker: add the gradient to every shard's accumulator (dropped if stale)."""
        self._stage_grads(grads)
        accepted = True
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b > a:
                accepted &= self._c(k).acc_apply(SHARD_ACC, self._ghost[a:b], self.local_step)
        if not accepted:
            self.dropped += 1
        return accepted

    def push_accumulate_and_wait(self, grads: torch.Tensor) -> int:
        """``push_accumulate`` + ``dequeue_token`` with one request fewer: the push to the shard that keeps
        ``global_step`` (sent last) also blocks for this worker's token."""
        self._stage_grads(grads)
        accepted = True
        order = [k for k in range(self.n) if k != self._gs_shard] + [self._gs_shard]
        tok = None
        for k in order:
            a, b = self.layout.shard_range[k]
            if k == self._gs_shard:
                try:
                    ok, tok = self._c(k).acc_apply_then_pop(SHARD_ACC, self._ghost[a:b], self.local_step,
                                                            token_key(self.widx), timeout=self.timeout)
                except TimeoutError:
                    raise TimeoutError(
                        f"sync barrier: no token within {self.timeout:.0f}s — a replica or the chief aggregator "
                        f"is gone (replicas_to_aggregate == total_num_replicas leaves no spare)") from None
                accepted &= ok
            elif b > a:
                accepted &= self._c(k).acc_apply(SHARD_ACC, self._ghost[a:b], self.local_step)
        if not accepted:
            self.dropped += 1
        self.local_step = struct.unpack("<q", tok)[0]
        return self.local_step

    def dequeue_token(self) -> int:
        """Worker: block until the chief released this worker's token; it carries the new global step.

        Tokens are dealt one per worker and round, so a fast worker cannot take a slow one's
        token: with replicas_to_aggregate == total_num_replicas every replica contributes exactly
        once per update and all of them finish on the same global step.
        """
        try:
            tok = self._c(self._gs_shard).qpop(token_key(self.widx), timeout=self.timeout)
        except TimeoutError:
            raise TimeoutError(
                f"sync barrier: no token within {self.timeout:.0f}s — a replica or the chief aggregator "
                f"is gone (replicas_to_aggregate == total_num_replicas leaves no spare)") from None
        self.local_step = struct.unpack("<q", tok)[0]
        return self.local_step

    def enqueue_tokens(self, clients, step: int, n: int):
        tok = struct.pack("<q", int(step))
        per_worker = spread_tokens(n, self.n_workers)
        if per_worker[0] > 0 and len(set(per_worker)) == 1:
            # the usual round (replicas_to_aggregate == number of workers): one request feeds every worker's queue
            clients[self._gs_shard].qpush_fanout(TOKEN_Q, tok, self.n_workers, copies=per_worker[0])
            return
        for w, copies in enumerate(per_worker):
            if copies:
                clients[self._gs_shard].qpush(token_key(w), tok, copies=copies)

    def live_workers(self, clients) -> int:
        """Workers still inside their training loop (the chief aggregator sizes its rounds by this)."""
        return self.n_workers - clients[0].add(FINISHED_KEY, 0)

    def begin_aggregation(self, clients):
        """Chief aggregator, once: accumulator steps start at the current global step (a restored run
        does not start at 0), the base the workers' ``local_step`` tags are compared against."""
        step = self.global_step_via(clients[self._gs_shard])
        for k in range(self.n):
            if self.layout.shard_numel(k):
                clients[k].acc_set_step(SHARD_ACC, step)

    def global_step_via(self, client) -> int:
        v = client.buf_read(SHARD_BUF, 1, offset=self._gs_off_in_shard)
        return int(round(float(v[0])))

    def sync_local_step(self):
        """Worker, before its first synchronous push: local_step := global_step (TF's local_step_init_op)."""
        self.local_step = self.global_step()

    def take_apply_all(self, clients, num_required: int, lr: float, timeout: float) -> int:
        """Chief aggregator: per shard wait for N gradients and apply their mean; then global_step += 1.

        Resumable: a TimeoutError leaves the round where it was (shards already
        applied are not taken again), so the caller can poll its stop flag and retry.
        """
        while self._agg_pos < self.n:
            k = self._agg_pos
            a, b = self.layout.shard_range[k]
            if b > a:
                clients[k].acc_take_apply(SHARD_ACC, SHARD_BUF, num_required, lr, timeout=timeout)
            self._agg_pos += 1
        self._agg_pos = 0
        return clients[self._gs_shard].buf_add_scalar(SHARD_BUF, self._gs_off_in_shard, 1.0)

    def take_apply_release(self, clients, num_required: int, lr: float, tokens: int, timeout: float) -> int:
        """Chief aggregator, a whole round: every shard takes and applies its mean gradient; the request to the
        shard that keeps ``global_step`` (sent last) also bumps the step and deals the tokens, server-side."""
        per_worker = spread_tokens(tokens, self.n_workers)
        if not (per_worker[0] > 0 and len(set(per_worker)) == 1):
            step = self.take_apply_all(clients, num_required, lr, timeout)      # uneven deal: separate requests
            self.enqueue_tokens(clients, step, tokens)
            return step
        order = [k for k in range(self.n) if k != self._gs_shard and self.layout.shard_numel(k)]
        while self._agg_pos < len(order):
            clients[order[self._agg_pos]].acc_take_apply(SHARD_ACC, SHARD_BUF, num_required, lr, timeout=timeout)
            self._agg_pos += 1
        step = clients[self._gs_shard].acc_take_apply_release(
            SHARD_ACC, SHARD_BUF, num_required, lr, self._gs_off_in_shard, TOKEN_Q, self.n_workers,
            copies=per_worker[0], timeout=timeout)
        self._agg_pos = 0
        return step

    def aggregator_clients(self):
        return [self.server.new_client("ps", k) for k in range(self.n)]

    def global_step(self) -> int:
        v = self._c(self._gs_shard).buf_read(SHARD_BUF, 1, offset=self._gs_off_in_shard)
        return int(round(float(v[0])))

    def close(self):
        pass
