"""Peer-mapped gradient exchange: the buffers and handles behind ``csrc/xgmi.hip``.

Every replica (one process per GPU, or several processes on one GPU in tests) owns

* two gradient slabs — the backward kernel of step *t* writes slab ``t & 1`` —
* an array of arrival flags in uncached memory, one word per peer,
* three words of local state (exchange number, finished blocks, error),

exports the first two as HIP-IPC handles and maps those of every peer; the addresses travel in the
arguments of every launch.  ``apply()`` then launches ONE kernel that announces the
local gradient, waits for the peers' announcements, sums the N slabs in rank order straight out
of the peers' HBM and applies ``p -= lr * sum`` to the local replica of the variables.

This is the synchronous data-parallel update of the reference (the mean gradient applied once
per N worker batches, tfdist_between_sync.py:65-68) with the all-reduce and the optimizer fused
into one launch; it replaces a library all-reduce whose ring steps dominate a 318 KB message.

The handles travel through any ``torch.distributed`` group (``all_gather_object``); the group is
used for set-up only, never per step.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from ..ops import hip_backend as hb

HANDLE_BYTES = 64


class _DeviceBuffer:
    """Memory from ``dtf_xgmi_alloc`` exposed to torch through ``__cuda_array_interface__``."""

    def __init__(self, ptr: int, nbytes: int, typestr: str, itemsize: int, owner: bool):
        self.ptr, self.nbytes, self.owner = ptr, nbytes, owner
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr,
                                         "data": (ptr, False), "version": 2, "strides": None}


# ---- exchange memory lives as long as the process ------------------------------------------------------------------
# Exchange buffers (inboxes, gradient slabs, flag arrays, parameter-server shards) are taken from a pool that never
# gives memory back to the driver, and mappings of other processes' buffers are kept open and reused.  Freeing and
# re-allocating such buffers while a process goes on building engines is what the pool rules out: on this
# platform (MI355X, ROCm 7.2, SPX) a process that had freed and re-allocated uncached exchange buffers a few times
# ended up with ONE of the eight XCDs reading other values than the remaining seven (and than the host) from
# ordinary device memory allocated afterwards, across kernel boundaries -- the exchange number of a tower, then
# parameters, finally a permutation.  bench/stress/ingraph_identity.py is the reproducer (variant "free": the old
# behaviour); with the pool twelve rounds of two engines each stay bit-identical.
_pool_free: dict = {}       # (device index, bytes, uncached) -> [address, ...]
_pool_all: dict = {}        # address -> (device index, bytes, uncached)
_pool_enabled = True        # the stress test's "free" variant switches it off to show the failure


def _alloc(nbytes: int, uncached: bool, fill: int = 0) -> int:
    """``nbytes`` of device memory on the current device, every byte set to ``fill``."""
    lib = hb.require()
    key = (torch.cuda.current_device(), int(nbytes), bool(uncached))
    free = _pool_free.get(key)
    if _pool_enabled and free:
        ptr = free.pop()
        hb._check(lib.dtf_xgmi_fill(ptr, int(fill) & 0xFF, int(nbytes)), "dtf_xgmi_fill")
        return ptr
    out = C.c_void_p()
    hb._check(lib.dtf_xgmi_alloc(C.byref(out), int(nbytes), int(uncached)), "dtf_xgmi_alloc")
    ptr = int(out.value)
    if fill:
        hb._check(lib.dtf_xgmi_fill(ptr, int(fill) & 0xFF, int(nbytes)), "dtf_xgmi_fill")
    _pool_all[ptr] = key
    return ptr


def _release(ptr: int) -> None:
    """Back into the pool (the memory stays allocated and mapped; the next taker re-fills it)."""
    if not ptr:
        return
    key = _pool_all.get(ptr)
    if key is None:
        return
    if not _pool_enabled:
        del _pool_all[ptr]
        with torch.cuda.device(key[0]):
            hb.require().dtf_xgmi_free(ptr)
        return
    free = _pool_free.setdefault(key, [])
    if ptr not in free:
        free.append(ptr)


def _export(ptr: int) -> bytes:
    buf = C.create_string_buffer(HANDLE_BYTES)
    hb._check(hb.require().dtf_xgmi_export(ptr, buf), "dtf_xgmi_export")
    return buf.raw


_opened: dict = {}          # (device index, handle bytes) -> [address, users]


def _open(handle: bytes) -> int:
    """Map another process's buffer; a handle this process has mapped before resolves to the mapping it has."""
    key = (torch.cuda.current_device(), bytes(handle))
    hit = _opened.get(key)
    if hit is not None:
        hit[1] += 1
        return hit[0]
    out = C.c_void_p()
    hb._check(hb.require().dtf_xgmi_open(C.create_string_buffer(handle, HANDLE_BYTES), C.byref(out)), "dtf_xgmi_open")
    _opened[key] = [int(out.value), 1]
    return int(out.value)


def _close(ptr: int) -> None:
    """Drop one use of a mapping.  The mapping itself stays (see the note on the pool above)."""
    for hit in _opened.values():
        if hit[0] == ptr and hit[1] > 0:
            hit[1] -= 1
            return


def _encode(obj) -> bytes:
    """JSON with byte strings as {"__b64__": ...}: what crosses the control store is data, never code."""
    import base64
    import json

    def conv(o):
        if isinstance(o, (bytes, bytearray)):
            return {"__b64__": base64.b64encode(bytes(o)).decode("ascii")}
        if isinstance(o, dict):
            return {str(k): conv(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [conv(v) for v in o]
        if o is None or isinstance(o, (bool, int, float, str)):
            return o
        raise TypeError(f"cannot send {type(o).__name__} through the control store")

    return json.dumps(conv(obj)).encode("utf-8")


def _decode(blob: bytes):
    import base64
    import json

    def conv(o):
        if isinstance(o, dict):
            if set(o) == {"__b64__"}:
                return base64.b64decode(o["__b64__"])
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, list):
            return [conv(v) for v in o]
        return o

    return conv(json.loads(bytes(blob).decode("utf-8")))


class StoreGroup:
    """``all_gather_object`` / ``barrier`` over a task server's control store, for processes that were started
    the reference's way (``--job_name/--task_index``) and share no ``torch.distributed`` group.  Objects travel
    as JSON (byte strings base64-coded): a peer that can reach the store can hand over data, not code."""

    def __init__(self, client, name: str, rank: int, world: int, timeout: float = 120.0):
        self.client, self.name, self.rank, self.world, self.timeout = client, name, int(rank), int(world), timeout
        self._round = 0
        self._nbar = 0

    def all_gather_object(self, out: list, obj) -> None:
        self._round += 1
        base = f"__grp__/{self.name}/{self._round}/"
        self.client.set(base + str(self.rank), _encode(obj))
        for r in range(self.world):       # blobs written by the other tasks of this job
            out[r] = _decode(self.client.get(base + str(r), timeout=self.timeout))

    def barrier(self) -> None:
        self._nbar += 1
        key = f"__grp__/{self.name}/barrier"
        self.client.add(key, 1)
        self.client.wait_ge(key, self.world * self._nbar, timeout=self.timeout)


def _all_gather(group, obj, world: int) -> list:
    out: List[Optional[dict]] = [None] * world
    if hasattr(group, "all_gather_object"):
        group.all_gather_object(out, obj)
    else:
        import torch.distributed as dist

        dist.all_gather_object(out, obj, group=group)
    return out


def _barrier(group) -> None:
    if hasattr(group, "barrier") and hasattr(group, "all_gather_object"):
        group.barrier()
    else:
        import torch.distributed as dist

        dist.barrier(group=group)


def _agree(group, world: int, error: Optional[str], what: str) -> None:
    """Collective go / no-go that doubles as a barrier: every rank reports whether its LOCAL part of a set-up
    step worked, and every rank raises together when any of them failed.  A rank that raised on its own
    (a refused mapping, an allocation that failed) would leave the others waiting in the next collective."""
    if world <= 1:
        if error:
            raise RuntimeError(f"{what}: {error}")
        return
    seen = _all_gather(group, {"error": error}, world)
    bad = [(r, s["error"]) for r, s in enumerate(seen) if s and s.get("error")]
    if bad:
        raise RuntimeError(f"{what} failed on rank(s) " + "; ".join(f"{r}: {e}" for r, e in bad))


def _export_all(alloc) -> dict:
    """Run ``alloc()`` (allocation + export, local work only) and return its dict, or {"error": ...}."""
    try:
        return alloc()
    except Exception as e:  # noqa: BLE001 — reported to every rank by the caller
        return {"error": f"{type(e).__name__}: {e}"}


class XgmiExchange:
    """Buffers and peer mappings of one replica.

    ``numel``: floats per gradient slab (padded to a multiple of 4 internally).  ``group``: the
    ``torch.distributed`` group used to swap handles (None with ``world == 1`` or when the peers
    are handed in through ``connect_local``).  ``uncached_grads``: keep the gradient slabs out of
    every L2 as well (slower stores in the backward kernel; the flags always are).
    """

    def __init__(self, rank: int, world: int, numel: int, device: torch.device, group=None,
                 uncached_grads: bool = False, timeout_ms: int = 5000, connect: bool = True):
        lib = hb.require()
        if not 1 <= world <= lib.dtf_xgmi_max_ranks():
            raise ValueError(f"world must be 1..{lib.dtf_xgmi_max_ranks()}")
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        self.numel = int(numel)
        self.padded = (self.numel + 3) // 4 * 4
        self.timeout_ms = int(timeout_ms)
        self._max = lib.dtf_xgmi_max_ranks()
        self._closed = False
        self._mapped: List[int] = []
        self._grad_ptr = self._flag_ptr = 0
        self._alloc_error: Optional[str] = None
        self.grads = []
        try:
            with torch.cuda.device(self.device):
                slab_bytes = 2 * self.padded * 4
                self._grad_ptr = _alloc(slab_bytes, uncached_grads)
                self._flag_ptr = _alloc(4096, True)
                self._grad_buf = _DeviceBuffer(self._grad_ptr, slab_bytes, "<f4", 4, True)
                both = torch.as_tensor(self._grad_buf, device=self.device)
                self.grads = [both[: self.padded][: self.numel], both[self.padded:][: self.numel]]
                self.state = torch.zeros(4, dtype=torch.int32, device=self.device)
                self.state[0] = 1                               # exchanges are numbered from 1; flags start at 0
        except Exception as e:  # noqa: BLE001 — with peers, the failure is announced to them in connect()
            self._alloc_error = f"{type(e).__name__}: {e}"
            if not connect or self.world == 1:
                raise
        self._peer_grad = [0] * self.world
        self._peer_flag = [0] * self.world
        self._peer_grad[self.rank], self._peer_flag[self.rank] = self._grad_ptr, self._flag_ptr
        if connect:
            self.connect(group)

    # -- wiring -----------------------------------------------------------------------
    def handles(self) -> dict:
        return {"rank": self.rank, "grad": _export(self._grad_ptr), "flag": _export(self._flag_ptr),
                "padded": self.padded}

    def connect(self, group=None):
        """Swap IPC handles over ``group`` and map every peer's buffers.  Every step that can fail on one rank
        only (allocation, export, opening a handle) is followed by a collective verdict, so that the ranks
        either all go on or all raise."""
        if self.world > 1:
            mine = {"error": self._alloc_error} if self._alloc_error else _export_all(self.handles)
            everyone = _all_gather(group, mine, self.world)
            bad = [(r, h["error"]) for r, h in enumerate(everyone) if h.get("error")]
            if bad:
                raise RuntimeError("peer-mapped exchange: allocation / export failed on rank(s) "
                                   + "; ".join(f"{r}: {e}" for r, e in bad))
            error = None
            try:
                with torch.cuda.device(self.device):
                    for h in everyone:
                        r = h["rank"]
                        if r == self.rank:
                            continue
                        if h["padded"] != self.padded:
                            raise RuntimeError("replicas disagree on the slab size")
                        self._peer_grad[r] = _open(h["grad"])
                        self._mapped.append(self._peer_grad[r])
                        self._peer_flag[r] = _open(h["flag"])
                        self._mapped.append(self._peer_flag[r])
            except Exception as e:  # noqa: BLE001
                error = f"{type(e).__name__}: {e}"
            # (also the barrier: nobody announces into a flag array that is not mapped yet)
            _agree(group, self.world, error, "peer-mapped exchange: mapping the peers' buffers")
        self._write_table()

    def connect_local(self, peers: List["XgmiExchange"]):
        """Replicas that live in ONE process (tests, in-graph towers on one GPU): plain pointers."""
        for p in peers:
            self._peer_grad[p.rank], self._peer_flag[p.rank] = p._grad_ptr, p._flag_ptr
        self._write_table()

    def _write_table(self):
        """Host arrays of device addresses handed to every launch: slabs of slot 0, slabs of slot 1, flag arrays."""
        u64 = C.c_uint64 * self.world
        self._slot_tabs = [u64(*self._peer_grad), u64(*[g + self.padded * 4 for g in self._peer_grad])]
        self._flag_tab = u64(*self._peer_flag)

    # -- per step ----------------------------------------------------------------------
    def apply(self, params: torch.Tensor, lr: torch.Tensor, slot: int, global_step: Optional[torch.Tensor] = None,
              required: int = 0):
        """``params -= lr * sum_r grads_r[slot]`` (+ ``global_step += 1``) on the current stream.

        ``required`` in 1 .. world - 1 (backup replicas, ``SyncReplicasOptimizer(replicas_to_aggregate < total)``): the
        sum runs over the first ``required`` replicas to arrive in this exchange -- every replica must pass the same
        number and scale its gradients by 1 / required; ``last_members()`` tells which ones counted."""
        if params.numel() != self.numel:
            raise ValueError("parameter slab and gradient slabs differ in size")
        if not 0 <= int(required) <= self.world:
            raise ValueError("required must lie in 0 .. world")
        hb._check(hb.require().dtf_xgmi_allreduce_apply_subset(
            self._slot_tabs[int(slot) & 1], self._flag_tab, self._flag_ptr, params.data_ptr(), self.numel, lr.data_ptr(),
            None if global_step is None else global_step.data_ptr(), self.state.data_ptr(), self.rank, self.world,
            int(required), self.timeout_ms, torch.cuda.current_stream(self.device).cuda_stream),
            "dtf_xgmi_allreduce_apply")

    def last_members(self) -> List[int]:
        """Ranks whose gradients counted in the last exchange with backup replicas (synchronises)."""
        mask = int(self.state[3].item())
        return [r for r in range(self.world) if (mask >> r) & 1]

    def failed(self) -> bool:
        """True once an exchange gave up waiting for a peer (reads the device word: synchronises)."""
        return bool(int(self.state[2].item()))

    def exchanges_done(self) -> int:
        return int(self.state[0].item()) - 1

    def close(self):
        if self._closed:
            return
        self._closed = True
        lib = hb.require()
        try:
            torch.cuda.synchronize(self.device)
        except Exception:  # noqa: BLE001 — tearing down after a device error
            pass
        with torch.cuda.device(self.device):
            for p in self._mapped:
                _close(p)
            self.grads = []
            _release(self._grad_ptr)
            _release(self._flag_ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class PeerShards:
    """GPU-resident parameter-server shards that every worker reaches one-sidedly (asynchronous mode).

    The variables are one flat slab of ``world * chunk`` floats; shard *k* (elements ``[k*chunk, (k+1)*chunk)``)
    is authoritative in rank *k*'s buffer.  Every rank allocates a slab-sized buffer, exports it and maps the
    others, after which

    * ``pull(local)``      copies every shard into the worker's private slab — one launch, remote shards
      read over xGMI;
    * ``push_sgd(g, lr)``  applies ``shard -= lr * g`` to every shard in place through the mappings and bumps
      ``global_step`` atomically — one launch, lock-free (``use_locking=False``, tfdist_between.py:65).

    Nothing here waits for another worker.
    """

    def __init__(self, rank: int, world: int, chunk: int, device: torch.device, group=None,
                 uncached: Optional[bool] = None):
        lib = hb.require()
        if not 1 <= world <= lib.dtf_xgmi_max_ranks():
            raise ValueError(f"world must be 1..{lib.dtf_xgmi_max_ranks()}")
        if chunk <= 0 or chunk % 4:
            raise ValueError("shards must be a positive multiple of 4 floats")
        self.rank, self.world, self.chunk, self.device = int(rank), int(world), int(chunk), torch.device(device)
        self.total = self.chunk * self.world
        self._closed = False
        self._mapped: List[int] = []
        # A shard is updated by atomics of its own GPU (executed in that GPU's L2, which keeps the line) AND by
        # atomics arriving over xGMI (executed at the memory): ordinary device memory is only made coherent between
        # the two at kernel boundaries, so updates could overwrite each other.  The shards are therefore allocated
        # uncached (fine-grained: every access goes to memory), which measured no slower than cached shards with
        # all workers on one GPU (13.6 against 14.6 us/step; DTF_SHARDS_UNCACHED=0 brings the cached ones back).
        if uncached is None:
            import os

            uncached = os.environ.get("DTF_SHARDS_UNCACHED", "1") != "0"
        self.uncached = bool(uncached)
        self._ptr = 0
        self.buffer = None

        def local():
            with torch.cuda.device(self.device):
                self._ptr = _alloc(self.total * 4, self.uncached)
                self._buf = _DeviceBuffer(self._ptr, self.total * 4, "<f4", 4, True)
                self.buffer = torch.as_tensor(self._buf, device=self.device)
                return {"rank": self.rank, "total": self.total,
                        "handle": _export(self._ptr) if self.world > 1 else b""}

        mine = _export_all(local)
        if self.world == 1 and mine.get("error"):
            raise RuntimeError(mine["error"])
        ptrs = [0] * self.world
        ptrs[self.rank] = self._ptr
        if self.world > 1:
            # every local step that may fail is followed by a verdict of ALL ranks (see _agree)
            everyone = _all_gather(group, mine, self.world)
            bad = [(r, h["error"]) for r, h in enumerate(everyone) if h.get("error")]
            if bad:
                raise RuntimeError("peer-mapped shards: allocation / export failed on rank(s) "
                                   + "; ".join(f"{r}: {e}" for r, e in bad))
            error = None
            try:
                with torch.cuda.device(self.device):
                    for h in everyone:
                        if h["rank"] == self.rank:
                            continue
                        if h["total"] != self.total:
                            raise RuntimeError("workers disagree on the slab size")
                        ptrs[h["rank"]] = _open(h["handle"])
                        self._mapped.append(ptrs[h["rank"]])
            except Exception as e:  # noqa: BLE001
                error = f"{type(e).__name__}: {e}"
            _agree(group, self.world, error, "peer-mapped shards: mapping the peers' buffers")
        self._tab = (C.c_uint64 * self.world)(*ptrs)
        self._group = group

    @property
    def my_shard(self) -> torch.Tensor:
        return self.buffer[self.rank * self.chunk:(self.rank + 1) * self.chunk]

    def publish(self, slab: torch.Tensor):
        """Initialise this rank's shard from its range of ``slab``; returns once every rank has done so."""
        a = self.rank * self.chunk
        self.my_shard.copy_(slab[a:a + self.chunk])
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            _barrier(self._group)

    def pull(self, local: torch.Tensor):
        if local.numel() != self.total:
            raise ValueError("local slab size differs from the sharded slab")
        hb._check(hb.require().dtf_xgmi_pull(self._tab, self.world, self.chunk, local.data_ptr(),
                                             torch.cuda.current_stream(self.device).cuda_stream), "dtf_xgmi_pull")

    def push_sgd(self, grads: torch.Tensor, lr: torch.Tensor, step_index: int = -1):
        if grads.numel() != self.total:
            raise ValueError("gradient slab size differs from the sharded slab")
        hb._check(hb.require().dtf_xgmi_push_sgd(self._tab, self.world, self.chunk, grads.data_ptr(), lr.data_ptr(),
                                                 int(step_index),
                                                 torch.cuda.current_stream(self.device).cuda_stream),
                  "dtf_xgmi_push_sgd")

    def step_descriptor(self, offsets: dict, step_index: int = -1, read_shards: bool = False) -> "hb.DtfXgDesc":
        """Arguments of ``hb.mlp_step_async``: the shard buffers of every rank, the element offsets of W1 / b1 /
        W2 / b2 inside the sharded slab and the element that counts the pushes (< 0: not counted).  ``read_shards``:
        the step's forward reads the variables from the shards themselves (no ``pull`` before the step)."""
        d = hb.DtfXgDesc()
        for r in range(self.world):
            d.inbox[r] = self._tab[r]
        d.state = None
        d.n = self.chunk
        d.offW1, d.offb1, d.offW2, d.offb2 = (int(offsets[k]) for k in ("W1", "b1", "W2", "b2"))
        d.rank, d.world, d.timeout_ms = self.rank, self.world, 0
        d.step_index = int(step_index)
        d.compact = 1 if read_shards else 0
        return d

    def close(self):
        if self._closed:
            return
        self._closed = True
        lib = hb.require()
        try:
            torch.cuda.synchronize(self.device)
        except Exception:  # noqa: BLE001 — tearing down after a device error
            pass
        with torch.cuda.device(self.device):
            for p in self._mapped:
                _close(p)
            self.buffer = None
            _release(self._ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass



class PeerInbox:
    """Inboxes of the exchange that runs INSIDE the backward launch (``hb.mlp_step_xg``).

    Every replica owns, in uncached memory, two slots (even / odd exchange numbers) of ``world`` rows of ``numel`` 8-byte
    words.  Row *s* of a replica's inbox is written by replica *s* only: one word {exchange number, value} per
    gradient element, stored straight into the peer's HBM; the owner polls its own inbox.  The number in the
    upper half of a word is the arrival flag of its lower half.
    """

    def __init__(self, rank: int, world: int, numel: int, device: torch.device, group=None, timeout_ms: int = 5000,
                 compact: bool = False, connect: bool = True, two_phase: bool = False):
        lib = hb.require()
        if not 1 <= world <= lib.dtf_xgmi_max_ranks():
            raise ValueError(f"world must be 1..{lib.dtf_xgmi_max_ranks()}")
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        self.numel = int(numel)
        self.timeout_ms = int(timeout_ms)
        # compact: bare 4-byte values, a word nobody has written yet holds 0xffffffff (half the bytes on a link)
        self.compact = bool(compact)
        # two_phase: partials travel to the OWNER of an element only (the slab cut into `world` ranges), which answers
        # with the new parameter value: 2 / world of the slab per link and step, one more hop (8-byte words)
        self.two_phase = bool(two_phase)
        if self.compact and self.two_phase:
            raise ValueError("compact words and the two-phase exchange are separate variants")
        self.slot_bytes = self.world * self.numel * (4 if self.compact else 8)
        self._closed = False
        self._mapped: List[int] = []
        self._ptr = 0

        def local():
            with torch.cuda.device(self.device):
                self._ptr = _alloc(2 * self.slot_bytes, True, fill=0xFF if self.compact else 0)
                # [0]: number of the current exchange, bumped by every forward launch (so numbers start at 1 and
                # an empty inbox, all zeros, never matches), [2]: error, [3..12]: who gave up waiting for whom
                # (xg_report_timeout in csrc/mlp_step2.hip, read by diagnose())
                self.state = torch.zeros(16, dtype=torch.int32, device=self.device)
                return {"rank": self.rank, "bytes": self.slot_bytes + (1 if self.two_phase else 0),   # (variant tag)
                        "handle": _export(self._ptr) if (self.world > 1 and connect) else b""}

        mine = _export_all(local)
        if (self.world == 1 or not connect) and mine.get("error"):
            raise RuntimeError(mine["error"])
        self._ptrs = [0] * self.world
        self._ptrs[self.rank] = self._ptr
        if self.world > 1 and connect:
            everyone = _all_gather(group, mine, self.world)
            bad = [(r, h["error"]) for r, h in enumerate(everyone) if h.get("error")]
            if bad:
                raise RuntimeError("peer inboxes: allocation / export failed on rank(s) "
                                   + "; ".join(f"{r}: {e}" for r, e in bad))
            error = None
            try:
                with torch.cuda.device(self.device):
                    for h in everyone:
                        if h["rank"] == self.rank:
                            continue
                        if h["bytes"] != self.slot_bytes + (1 if self.two_phase else 0):
                            raise RuntimeError("replicas disagree on the inbox size")
                        self._ptrs[h["rank"]] = _open(h["handle"])
                        self._mapped.append(self._ptrs[h["rank"]])
            except Exception as e:  # noqa: BLE001
                error = f"{type(e).__name__}: {e}"
            # (also the barrier: nobody writes into an inbox that is not mapped yet)
            _agree(group, self.world, error, "peer inboxes: mapping the peers' buffers")

    def connect_local(self, peers: List["PeerInbox"]):
        """Replicas that live in ONE process (in-graph towers): the peers' inboxes are plain pointers.  Towers
        on different GPUs need peer access between the devices (``hb.enable_peer_access``) first."""
        for p in peers:
            if p.slot_bytes != self.slot_bytes or p.world != self.world:
                raise ValueError("towers disagree on the inbox size")
            self._ptrs[p.rank] = p._ptr

    def step_descriptor(self, offsets: dict) -> "hb.DtfXgDesc":
        """Arguments of ``hb.mlp_step_xg``; ``offsets`` maps W1 / b1 / W2 / b2 to the element offsets of their
        gradients inside a row.  (The kernel picks the slot from the exchange number.)"""
        d = hb.DtfXgDesc()
        for r in range(self.world):
            d.inbox[r] = self._ptrs[r]
        d.state = self.state.data_ptr()
        d.n = self.numel
        d.offW1, d.offb1, d.offW2, d.offb2 = (int(offsets[k]) for k in ("W1", "b1", "W2", "b2"))
        d.rank, d.world, d.timeout_ms = self.rank, self.world, self.timeout_ms
        d.compact = 2 if self.two_phase else int(self.compact)
        return d

    def failed(self) -> bool:
        return bool(int(self.state[2].item()))

    def exchanges_done(self) -> int:
        return int(self.state[0].item())

    def diagnose(self) -> Optional[dict]:
        """After a failed exchange: the record of the first thread that gave up (which workgroup and thread, which
        peer's word was missing for which element of which exchange, what the word held instead), how many threads
        gave up, and what that word of the inbox holds NOW in both slots (a word that has arrived meanwhile means a
        peer that was late, not one that never sent).  None while nothing has failed."""
        st = [int(v) & 0xFFFFFFFF for v in self.state.cpu().tolist()]
        if not st[2]:
            return None
        rec = {"rank": self.rank, "exchanges_started": st[0], "threads_gave_up": st[12], "workgroup": st[4],
               "thread": st[5], "waiting_for_rank": st[6] if st[6] < self.world else -1, "exchange": st[7],
               "word_seen": (st[8], st[9]), "element": st[10], "inbox": hex(self._ptr)}
        peer, e = rec["waiting_for_rank"], st[10]
        if 0 <= peer < self.world and e < self.numel:
            width = 4 if self.compact else 8
            now = []
            for slot in (0, 1):
                addr = self._ptr + slot * self.slot_bytes + (peer * self.numel + e) * width
                view = torch.as_tensor(_DeviceBuffer(addr, width, "<i4", 4, False), device=self.device)
                now.append(tuple(int(v) & 0xFFFFFFFF for v in view.cpu().tolist()))
            rec["word_now_slot0"], rec["word_now_slot1"] = now     # 8-byte words: (value bits, exchange number)
        return rec

    def close(self):
        if self._closed:
            return
        self._closed = True
        lib = hb.require()
        try:
            torch.cuda.synchronize(self.device)
        except Exception:  # noqa: BLE001 — tearing down after a device error
            pass
        with torch.cuda.device(self.device):
            for p in self._mapped:
                _close(p)
            _release(self._ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
