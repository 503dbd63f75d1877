"""GPU-resident parameter-server shards shared over HIP-IPC (one node).

``ConfigProto(transport="ipc")``.  The ps task allocates its shard in the HBM of
its GPU and exports the allocation as an IPC memory handle; workers map it and
from then on move **no tensor bytes through any server**:

* pull         one device-to-device copy per shard (same GPU, or peer-to-peer
               over xGMI when the shard lives on another GPU) on the worker's
               stream;
* async push   the worker launches the sharded SGD kernel directly on the
               mapped shard (``p -= lr*g`` written through the mapping): a
               one-sided, lock-free Hogwild update, the meaning of
               ``use_locking=False`` in the reference (tfdist_between.py:65);
               ``use_locking=True`` brackets the launch with a mutex kept by the
               ps task's native server;
* sync push    the worker copies its gradient into one of ITS two mailbox rows
               next to the shard (no atomics, no write conflicts; the rows
               alternate with the worker's pushes), synchronises its stream,
               and with ONE request to the ps control plane queues an arrival
               record (worker index, ``local_step``, mailbox row) and blocks for
               its token; the chief aggregator takes the arrivals of a round
               with one request, keeps the required number of fresh ones
               (records older than the accumulator step are dropped as stale),
               applies ``p -= lr/n * sum(their mailboxes)`` with ONE launch per
               shard on its own stream, bumps ``global_step`` and deals one
               token to every worker.  With ``replicas_to_aggregate ==
               total_num_replicas`` a mailbox row is rewritten only after its
               owner received the token of the round that consumed it.  With
               backup workers (``replicas_to_aggregate < total_num_replicas``) a
               replica can hold spare tokens and run ahead of the aggregator;
               the two rows per worker keep a gradient that is being summed
               apart from the one being written, as long as the aggregator is
               less than two of that worker's steps behind.

This is the path for tasks that share a GPU — RCCL refuses two ranks on one
device — e.g. "1 PS + 1 worker on one MI355X" and the reference's scenario of a
PS and two workers on gpu0 (README.md:42-86), and it is the truly asynchronous
data plane: no worker ever waits for a matching receive.

Only the control plane (allocation request, handle exchange, arrival counters,
tokens, locks) uses the native task servers.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import time
from typing import List, Optional

import torch

from ..ops import hip_backend as hb
from ..runtime import native as rt
from ..utils.trace import span
from ..variables import ShardLayout
from .store import FINISHED_KEY, INIT_KEY, TOKEN_Q, spread_tokens, token_key

ALLOC_KEY = "__ipc_alloc__"
HANDLE_KEY = "__ipc_handle__"
ARRIVAL_Q = "__ipc_arrival_q__"
BOARD_KEY = "__ipc_barrier_board__"       # name of the shared-memory board of the barrier (empty: use the sockets)
ACC_STEP = "__ipc_acc_step__"
LOCK_Q = "__ipc_lock__"


_SHARD_MAGIC = 0x44544653           # "DTFS"
_SHARD_BLOB = struct.Struct("<Iqq64s")


def _allocate_shard(rows: int, cols: int, dev: torch.device) -> bytes:
    """``rows x cols`` zeroed floats in UNCACHED memory of ``dev``; returns the blob the workers map it from.

    A shard is written by workers on its own GPU (through that GPU's L2) and by workers on other GPUs (over xGMI,
    at the memory): ordinary device memory is made coherent between the two only at kernel boundaries, uncached
    (fine-grained) memory always is.  The allocation lives as long as the ps task."""
    from .xgmi import _alloc, _export as export_handle

    with torch.cuda.device(dev):
        ptr = _alloc(rows * cols * 4, True)
    return _SHARD_BLOB.pack(_SHARD_MAGIC, int(rows), int(cols), export_handle(ptr))


def _import(blob: bytes, device: torch.device) -> torch.Tensor:
    """Maps a shard published by ``_allocate_shard`` into this process, as a tensor of ``device``'s kernels."""
    from .xgmi import _DeviceBuffer, _open

    # A fixed-layout record (magic, rows, cols, 64-byte IPC handle): whoever can write the control store can hand
    # over numbers and a handle, never code; the sizes are bounded before anything is mapped.
    if len(blob) != _SHARD_BLOB.size:
        raise ValueError("shard record of the wrong size in the control store")
    magic, rows, cols, handle = _SHARD_BLOB.unpack(blob)
    if magic != _SHARD_MAGIC or not (1 <= rows <= 4096) or not (4 <= cols <= (1 << 32)):
        raise ValueError("malformed shard record in the control store")
    with torch.cuda.device(device):
        ptr = _open(handle)
        buf = _DeviceBuffer(ptr, rows * cols * 4, "<f4", 4, False)
        return torch.as_tensor(buf, device=device).view(rows, cols)


def serve_ipc_shard(server, poll: float = 0.2) -> bool:
    """Body of ``Server.join()`` for a ps task under the ipc transport.

    Waits for the chief's allocation request ({numel, rows}), allocates
    ``rows x numel`` floats on this task's GPU (row 0 = the variables, rows 1..N
    = one gradient mailbox per worker), publishes the IPC handle and then
    blocks until the cluster releases the task.  The allocation is never freed:
    it lives as long as the process, which keeps the mapping valid in the workers.
    """
    from ..runtime import native as rt

    me = rt.StoreClient("127.0.0.1", server.port, timeout=30)
    try:
        while True:
            if server._native.wait(0.0):
                return True
            try:
                req = me.get(ALLOC_KEY, timeout=poll)
            except TimeoutError:
                continue
            numel, rows = struct.unpack("<qq", req)
            dev = server.device
            if dev.type != "cuda":
                raise RuntimeError("transport='ipc' needs a GPU in the ps task")
            blob = _allocate_shard(rows, max(numel, 4), dev)
            me.qpush(LOCK_Q, b"1")                 # the shard mutex starts released
            me.set(HANDLE_KEY, blob)
            me.delete(ALLOC_KEY)
            if server.config.log_device_placement:
                print(f"[ps {server.task_index}] shard of {numel} floats + {rows - 1} mailboxes resident on {dev}",
                      flush=True)
            return server._native.wait(None)
    finally:
        me.close()


class IpcVariableStore:
    is_remote = True

    def __init__(self, layout: ShardLayout, server, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("transport='ipc' needs GPU workers")
        hb.require()
        self.layout = layout
        self.server = server
        self.device = device
        self.n = layout.num_shards
        self.n_workers = server.num_workers
        self.widx = server.task_index
        self.timeout = server.config.operation_timeout_in_ms / 1000.0
        self.local_step = 0
        self.dropped = 0
        self._maps: List[Optional[torch.Tensor]] = [None] * self.n
        self._gs_shard = layout.owner_of("global_step") if "global_step" in layout.var_range else 0
        o, _ = layout.var_range.get("global_step", (0, 1))
        self._gs_off = o - layout.shard_range[self._gs_shard][0]
        self._lr_t = torch.zeros(1, dtype=torch.float32, device=device)
        self._lr_val = None
        self._arrived: List[tuple] = []   # aggregator: (worker, mailbox row) of the gradients of the open round
        self._pushes = 0                   # this worker's pushes so far (its two mailbox rows alternate)
        self._agg_plans = {}               # aggregator: launch arguments per combination of mailbox rows
        self._acc_step = 0                 # aggregator: rounds applied so far (the accumulator step)

    def _c(self, k: int):
        return self.server.client("ps", k)

    # -- the barrier's control plane: shared memory when every task runs on this node, the task server otherwise ----
    def _all_tasks_local(self) -> bool:
        import socket

        here = {"localhost", "127.0.0.1", "::1", "0.0.0.0", socket.gethostname()}
        spec = self.server.cluster
        return all(addr.rsplit(":", 1)[0] in here for job in spec.jobs for addr in spec.job_tasks(job))

    def _create_board(self):
        """Chief, when the variables are initialised: create the board and publish its name (an empty name tells the
        other tasks to stay on the sockets: tasks on several hosts, or DTF_SYNC_BOARD=0)."""
        import os

        if getattr(self, "_board_known", False):        # (a restore after the initialisation: same board)
            self._c(self._gs_shard).set(BOARD_KEY, self._board.name.encode() if self._board is not None else b"")
            return
        self._board, name = None, b""
        if os.environ.get("DTF_SYNC_BOARD", "1") != "0" and self._all_tasks_local():
            try:
                port = self.server.cluster.task_address("ps", self._gs_shard).rsplit(":", 1)[1]
                self._board = rt.Board(rt.Board.unique_name(port), self.n_workers, create=True)
                name = self._board.name.encode()
            except Exception as e:  # noqa: BLE001 — no shared memory here: the sockets still work
                print(f"[ipc store] barrier board unavailable ({e}); using the task server", flush=True)
        self._board_known = True
        self._c(self._gs_shard).set(BOARD_KEY, name)

    def _get_board(self):
        if not getattr(self, "_board_known", False):
            name = bytes(self._c(self._gs_shard).get(BOARD_KEY, timeout=self.timeout))
            self._board = rt.Board(name.decode(), self.n_workers, create=False) if name else None
            self._board_known = True
        return self._board

    def _lr(self, lr: float) -> torch.Tensor:
        if self._lr_val != lr:
            self._lr_t.fill_(float(lr))
            self._lr_val = lr
        return self._lr_t

    # -- mapping ---------------------------------------------------------------------
    def _map(self, k: int) -> torch.Tensor:
        m = self._maps[k]
        if m is None:
            blob = self._c(k).get(HANDLE_KEY, timeout=self.timeout)
            # (a shard in another GPU's HBM is mapped with peer access enabled by the open call itself)
            m = _import(blob, self.device)
            self._maps[k] = m
        return m

    def _params(self, k: int) -> torch.Tensor:
        return self._map(k)[0, :self.layout.shard_numel(k)] if self.layout.shard_numel(k) else self._map(k)[0, :0]

    def _mailbox(self, k: int, worker: int, slot: int = 0) -> torch.Tensor:
        return self._map(k)[1 + 2 * worker + (slot & 1), :self.layout.shard_numel(k)]

    # -- lifecycle ---------------------------------------------------------------------
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

    def _request_allocation(self):
        for k in range(self.n):
            self._c(k).set(ALLOC_KEY, struct.pack("<qq", self.layout.shard_numel(k), 1 + 2 * self.n_workers))
        for k in range(self.n):
            self._map(k)

    def initialize(self, seed: int, only=None):
        """Chief: have every ps task allocate its shard, then run the initializers INTO the shards."""
        self._request_allocation()
        for k in range(self.n):
            if self.layout.shard_numel(k):
                self.layout.initialize(self._params(k), seed, shard=k, slab_offset=self.layout.shard_range[k][0])
        torch.cuda.synchronize(self.device)
        self._create_board()
        self._c(0).set(INIT_KEY, b"1")

    def publish(self, local: torch.Tensor):
        """Overwrite the shards with ``local`` (a replica that was trained elsewhere); no initialisation side effects."""
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b > a:
                self._params(k).copy_(local[a:b])
        torch.cuda.synchronize(self.device)

    def write_all(self, local: torch.Tensor):
        self._request_allocation()
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b > a:
                self._params(k).copy_(local[a:b])
        torch.cuda.synchronize(self.device)
        self._create_board()
        self._c(0).set(INIT_KEY, b"1")

    # -- data plane --------------------------------------------------------------------------
    def pull(self, local: torch.Tensor, locked: bool = False):
        """Every shard into the local slab: one launch reading all the mapped shards (a copy per shard when there
        are more shards than the kernel's table holds)."""
        tab = self._pull_table()
        if tab is not None and local.is_contiguous() and local.numel() == tab[2]:
            hb._check(hb.require().dtf_xgmi_pull_ranges(tab[0], tab[1], self.n, local.data_ptr(),
                                                        torch.cuda.current_stream(self.device).cuda_stream),
                      "dtf_xgmi_pull_ranges")
            return
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b > a:
                local[a:b].copy_(self._params(k), non_blocking=True)

    def _pull_table(self):
        """Host arrays for ``dtf_xgmi_pull_ranges``: first-element address and slab range of every shard."""
        tab = getattr(self, "_pull_tab", False)
        if tab is False:
            tab = None
            total = self.layout.shard_range[self.n - 1][1]
            if self.n <= hb.require().dtf_xgmi_max_ranks() and total % 4 == 0:
                ptrs = (C.c_uint64 * self.n)(*[self._params(k).data_ptr() if self.layout.shard_numel(k) else 0
                                               for k in range(self.n)])
                starts = (C.c_int64 * (self.n + 1))(*[self.layout.shard_range[k][0] for k in range(self.n)], total)
                tab = (ptrs, starts, total)
            self._pull_tab = tab
        return tab

    def push_apply(self, grads: torch.Tensor, lr: float, use_locking: bool = False, bump_step: bool = True):
        lr_t = self._lr(lr)
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            if b <= a:
                continue
            p = self._params(k)
            gs = p[self._gs_off:self._gs_off + 1] if (bump_step and k == self._gs_shard) else None
            if use_locking:
                self._c(k).qpop(LOCK_Q, timeout=self.timeout)
                try:
                    hb.sgd_apply_(p, grads[a:b], lr_t, 1.0, False, gs)
                    torch.cuda.current_stream(self.device).synchronize()
                finally:
                    self._c(k).qpush(LOCK_Q, b"1")
            else:
                hb.sgd_apply_(p, grads[a:b], lr_t, 1.0, False, gs)

    def step_counter_view(self) -> torch.Tensor:
        """The live ``global_step`` element in its owner's shard (one float of mapped device memory)."""
        view = getattr(self, "_gs_view", None)
        if view is None:
            view = self._gs_view = self._params(self._gs_shard)[self._gs_off:self._gs_off + 1]
        return view

    def push_descriptor(self, offsets: dict, bump_step: bool = True, read_shards: bool = False) -> "hb.DtfXgDesc":
        """Arguments of ``hb.mlp_step_async`` for these shards: the backward launch itself applies ``-lr * g`` of
        every gradient element to the shard that holds the variable (lock-free; tfdist_between.py:65-66).
        ``offsets`` maps W1 / b1 / W2 / b2 to their element offsets in the local slab.  ``read_shards``: the step's
        forward reads the variables from the mapped shards themselves -- no ``pull`` in front of the step."""
        if self.n > hb.require().dtf_xgmi_max_ranks():
            raise ValueError("more parameter-server shards than the step's shard table holds")
        d = hb.DtfXgDesc()
        for k in range(self.n):
            a, b = self.layout.shard_range[k]
            d.shard_start[k] = a
            d.inbox[k] = self._params(k).data_ptr() if b > a else 0
        d.shard_start[self.n] = self.layout.shard_range[self.n - 1][1]
        d.state = None
        d.n = 0
        d.offW1, d.offb1, d.offW2, d.offb2 = (int(offsets[k]) for k in ("W1", "b1", "W2", "b2"))
        d.rank, d.world, d.timeout_ms = self.widx, self.n, 0
        d.step_index = self.layout.var_range["global_step"][0] if (bump_step and "global_step" in self.layout.var_range) else -1
        d.compact = 1 if read_shards else 0
        return d

    # -- SyncReplicas protocol ------------------------------------------------------------------
    def _deposit(self, grads: torch.Tensor) -> bytes:
        slot = self._pushes & 1
        self._pushes += 1
        with span("deposit_gradient"):
            for k in range(self.n):
                a, b = self.layout.shard_range[k]
                if b > a:
                    self._mailbox(k, self.widx, slot).copy_(grads[a:b], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()   # bytes are in HBM before the arrival is queued
        return struct.pack("<qqq", self.widx, self.local_step, slot)

    def push_accumulate(self, grads: torch.Tensor) -> bool:
        rec = self._deposit(grads)
        board = self._get_board()
        if board is not None:
            board.arrive(*struct.unpack("<qqq", rec))
        else:
            self._c(self._gs_shard).qpush(ARRIVAL_Q, rec)
        return True

    def _announce_and_wait(self, rec: bytes) -> int:
        """Announce the deposited gradient and wait for this worker's token: through the shared-memory board (no
        system call unless somebody sleeps), or as one request to the task server."""
        board = self._get_board()
        try:
            if board is not None:
                board.arrive(*struct.unpack("<qqq", rec))
                with span("token_wait"):
                    self.local_step = board.wait_token(self.widx, self.timeout)
            else:
                tok = self._c(self._gs_shard).qpush_pop(ARRIVAL_Q, rec, token_key(self.widx), timeout=self.timeout)
                self.local_step = struct.unpack("<q", tok)[0]
        except TimeoutError:
            raise self._no_token() from None
        return self.local_step

    # The gradient can be produced IN the mailbox: the backward kernel's output pointers are views of this worker's
    # mailbox rows (one per variable, in the shard that owns the variable), so a synchronous push moves no bytes.
    def mailbox_grad_slots(self, names: dict):
        """[{role: tensor}, {role: tensor}]: views of this worker's two mailbox rows shaped like the variables
        ``names[role]`` — gradient outputs for ``MLPEngine.bind_grad_slots``."""
        slots = []
        for slot in (0, 1):
            views = {}
            for role, name in names.items():
                k = self.layout.owner_of(name)
                o, n = self.layout.var_range[name]
                a = self.layout.shard_range[k][0]
                views[role] = self._map(k)[1 + 2 * self.widx + slot, o - a:o - a + n].view(self.layout.by_name[name].shape)
            slots.append(views)
        return slots

    def next_mailbox_slot(self) -> int:
        return self._pushes & 1

    def push_deposited_and_wait(self) -> int:
        """The gradient of this step already sits in mailbox row ``next_mailbox_slot()`` and the stream has been
        synchronised since: announce it and wait for the token (one request)."""
        slot = self._pushes & 1
        self._pushes += 1
        return self._announce_and_wait(struct.pack("<qqq", self.widx, self.local_step, slot))

    def _no_token(self):
        return TimeoutError(
            f"sync barrier: no token within {self.timeout:.0f}s — a replica or the chief aggregator "
            f"is gone (replicas_to_aggregate == total_num_replicas leaves no spare)")

    def dequeue_token(self) -> int:
        board = self._get_board()
        try:
            if board is not None:
                self.local_step = board.wait_token(self.widx, self.timeout)
            else:
                tok = self._c(self._gs_shard).qpop(token_key(self.widx), timeout=self.timeout)
                self.local_step = struct.unpack("<q", tok)[0]
        except TimeoutError:
            raise self._no_token() from None
        return self.local_step

    def push_accumulate_and_wait(self, grads: torch.Tensor) -> int:
        """``push_accumulate`` + ``dequeue_token`` as ONE hand-off to the control plane."""
        return self._announce_and_wait(self._deposit(grads))

    def aggregator_clients(self):
        return [self.server.new_client("ps", k) for k in range(self.n)]

    def live_workers(self, clients) -> int:
        return self.n_workers - clients[0].add(FINISHED_KEY, 0)

    def begin_aggregation(self, clients):
        self._acc_step = self.global_step()

    def sync_local_step(self):
        self.local_step = self.global_step()

    def take_apply_all(self, clients, num_required: int, lr: float, timeout: float) -> int:
        """Chief aggregator: collect num_required fresh arrivals, then p -= lr/n * sum(their mailboxes).

        Resumable: arrivals already popped stay in ``_arrived`` across a TimeoutError.
        """
        if num_required > self.n_workers:
            raise ValueError("replicas_to_aggregate exceeds the number of workers")
        ctl = clients[self._gs_shard]
        deadline = time.time() + timeout
        while len(self._arrived) < num_required:
            # the arrivals still missing, in one request (TimeoutError when none came)
            with span("wait_arrivals"):
                board = self._get_board()
                if board is not None:
                    arrivals = board.take(num_required - len(self._arrived), max(0.0, deadline - time.time()))
                else:
                    got, recs = ctl.qpop_n(ARRIVAL_Q, num_required - len(self._arrived),
                                           timeout=max(0.0, deadline - time.time()))
                    arrivals = [struct.unpack_from("<qqq", recs, 24 * i) for i in range(got)]
            for w, step, slot in arrivals:
                if step < self._acc_step:
                    self.dropped += 1              # stale: its round was applied without it
                elif all(w != a for a, _ in self._arrived):
                    self._arrived.append((int(w), int(slot)))
            if len(self._arrived) < num_required and time.time() >= deadline:
                raise TimeoutError("sync aggregation: not enough fresh gradients yet")
        stream = getattr(self, "_agg_stream", None)
        if stream is None:
            stream = self._agg_stream = torch.cuda.Stream(device=self.device)
            self._agg_lr = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._agg_lr_val = None
        n = len(self._arrived)
        with span("sum_apply"), torch.cuda.stream(stream):
            if self._agg_lr_val != lr:
                self._agg_lr.fill_(float(lr))
                self._agg_lr_val = lr
            if n <= 16:             # take_grad of every accumulator + the apply + the step bump: one launch per shard
                key = tuple(self._arrived)
                plan = self._agg_plans.get(key)
                if plan is None:    # device addresses of this combination of mailbox rows, built once
                    plan = []
                    for k in range(self.n):
                        if not self.layout.shard_numel(k):
                            continue
                        p = self._params(k)
                        gs = p[self._gs_off:self._gs_off + 1] if k == self._gs_shard else None
                        tab = (C.c_uint64 * n)(*[self._mailbox(k, w, slot).data_ptr() for w, slot in self._arrived])
                        plan.append((p.data_ptr(), tab, p.numel(), None if gs is None else gs.data_ptr()))
                    self._agg_plans[key] = plan
                lib, sh = hb.require(), stream.cuda_stream
                for pp, tab, numel, gsp in plan:
                    hb._check(lib.dtf_sum_apply(pp, tab, n, numel, self._agg_lr.data_ptr(), 1.0 / n, gsp, sh),
                              "dtf_sum_apply")
            else:
                for k in range(self.n):
                    if not self.layout.shard_numel(k):
                        continue
                    p = self._params(k)
                    gs = p[self._gs_off:self._gs_off + 1] if k == self._gs_shard else None
                    for i, (w, slot) in enumerate(self._arrived):
                        hb.sgd_apply_(p, self._mailbox(k, w, slot), self._agg_lr, 1.0 / n, False, gs if i == n - 1 else None)
        with span("sum_apply_sync"):
            stream.synchronize()                  # the tokens must not overtake the update
        self._arrived = []
        self._acc_step += 1
        # == global_step: read from the shard when the aggregation began, bumped once per round by the last apply
        # above (no device-to-host copy on the path every replica is waiting on)
        return self._acc_step

    def take_apply_release(self, clients, num_required: int, lr: float, tokens: int, timeout: float) -> int:
        """Chief aggregator, a whole round: arrivals (one request), the update (one launch per shard), tokens (one
        request).  A TimeoutError leaves the round open."""
        done = self._native_round(num_required, lr, tokens, timeout)
        if done is not None:
            return done
        step = self.take_apply_all(clients, num_required, lr, timeout)
        with span("release_tokens"):
            self.enqueue_tokens(clients, step, tokens)
        self.rounds_in_last_call = 1 + getattr(self, "_carry_rounds", 0)
        self._carry_rounds = 0
        return step

    #: rounds that one native call may apply before the interpreter looks at the learning rate and at the coordinator again
    NATIVE_ROUNDS_PER_CALL = 32

    def _native_round(self, num_required: int, lr: float, tokens: int, timeout: float) -> Optional[int]:
        """The common rounds -- every worker arrives, in step, with the same mailbox row -- inside ONE native call
        (``dtf_sync_rounds``: arrivals, one ``dtf_sum_apply`` per shard, the wait for them, the tokens, and the same again
        for up to NATIVE_ROUNDS_PER_CALL rounds): the replicas wait for exactly this stretch, and the interpreter's share
        of it (context managers, ctypes calls, and above all the lock that the chief's thread shares with the training
        loop of the chief's own replica, which fell 50 - 70 us behind the other replicas every round) was most of it.
        Returns the new step, or None with whatever arrived moved into ``_arrived`` for the general path
        (DTF_SYNC_NATIVE_ROUND=0: never used).  ``rounds_in_last_call`` says how many rounds the call applied."""
        self.rounds_in_last_call = 1
        board = self._get_board()
        per_worker = spread_tokens(tokens, self.n_workers)
        if (board is None or self._arrived or num_required != self.n_workers or self.n_workers > 16
                or per_worker[0] <= 0 or len(set(per_worker)) != 1
                or os.environ.get("DTF_SYNC_NATIVE_ROUND", "1") == "0"):
            return None
        plan = getattr(self, "_round_plan", None)
        if plan is None:
            shards = [k for k in range(self.n) if self.layout.shard_numel(k)]
            params = [self._params(k) for k in shards]
            tabs = [(C.c_uint64 * self.n_workers)(*[self._mailbox(k, w, slot).data_ptr() for w in range(self.n_workers)])
                    for slot in (0, 1) for k in shards]
            gs = [p[self._gs_off:self._gs_off + 1].data_ptr() if k == self._gs_shard else None
                  for k, p in zip(shards, params)]
            plan = self._round_plan = {
                "n": len(shards), "keep": (params, tabs),
                "params": (C.c_void_p * len(shards))(*[p.data_ptr() for p in params]),
                "tabs": (C.c_void_p * len(tabs))(*[C.cast(t, C.c_void_p).value for t in tabs]),
                "numel": (C.c_int64 * len(shards))(*[p.numel() for p in params]),
                "gs": (C.c_void_p * len(shards))(*gs),
                "out": (C.c_int64 * (3 * 16))(), "got": C.c_int(0), "rounds": C.c_int(0),
                "entry": board.native_entry_points(),
            }
            self._agg_stream = getattr(self, "_agg_stream", None) or torch.cuda.Stream(device=self.device)
            if getattr(self, "_agg_lr", None) is None:
                self._agg_lr = torch.zeros(1, dtype=torch.float32, device=self.device)
                self._agg_lr_val = None
            lib = hb.require()
            lib.dtf_sync_rounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.dtf_sync_rounds.restype = C.c_int
        if self._agg_lr_val != lr:
            with torch.cuda.stream(self._agg_stream):
                self._agg_lr.fill_(float(lr))
            self._agg_stream.synchronize()
            self._agg_lr_val = lr
        take, release, handle = plan["entry"]
        with span("native_rounds"), torch.cuda.device(self.device):
            rc = hb.require().dtf_sync_rounds(take, release, handle, self.n_workers, self._acc_step,
                                             max(0, int(timeout * 1000)), plan["n"], plan["params"], plan["tabs"],
                                             plan["numel"], plan["gs"], self._agg_lr.data_ptr(), per_worker[0],
                                             self._agg_stream.cuda_stream, self.NATIVE_ROUNDS_PER_CALL, plan["out"],
                                             C.byref(plan["got"]), C.byref(plan["rounds"]))
        done = plan["rounds"].value
        self._acc_step += done
        if rc not in (0, 1, 2):
            hb._check(rc, "dtf_sync_rounds")
        out = plan["out"]
        for i in range(plan["got"].value if rc != 0 else 0):      # the general path's bookkeeping for what was taken
            w, step, slot = out[3 * i], out[3 * i + 1], out[3 * i + 2]
            if step < self._acc_step:
                self.dropped += 1
            elif all(w != a for a, _ in self._arrived):
                self._arrived.append((int(w), int(slot)))
        if done > 0 and (rc == 0 or not self._arrived or rc == 2):
            self.rounds_in_last_call = done
            return self._acc_step
        self._carry_rounds = getattr(self, "_carry_rounds", 0) + done      # counted when the general path ends its round
        if rc == 2 and len(self._arrived) < num_required:
            raise TimeoutError("sync aggregation: not enough fresh gradients yet")
        return None

    def enqueue_tokens(self, clients, step: int, n: int):
        per_worker = spread_tokens(n, self.n_workers)
        board = self._get_board()
        if board is not None:
            if per_worker[0] > 0 and len(set(per_worker)) == 1:
                board.release(-1, int(step), per_worker[0])
            else:
                for w, copies in enumerate(per_worker):
                    board.release(w, int(step), copies)
            return
        tok = struct.pack("<q", int(step))
        if per_worker[0] > 0 and len(set(per_worker)) == 1:
            # the usual round (replicas_to_aggregate == number of workers): one request feeds every worker's queue
            clients[self._gs_shard].qpush_fanout(TOKEN_Q, tok, self.n_workers, copies=per_worker[0])
            return
        for w, copies in enumerate(per_worker):
            if copies:
                clients[self._gs_shard].qpush(token_key(w), tok, copies=copies)

    def global_step(self) -> int:
        torch.cuda.current_stream(self.device).synchronize()
        return int(round(float(self._params(self._gs_shard)[self._gs_off].item())))

    def close(self):
        pass
