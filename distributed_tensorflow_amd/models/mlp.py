"""The reference workload as a device-resident engine.

``MLPEngine`` binds the fused gfx950 kernels (csrc/mlp_gemm.hip, mlp_head.hip)
to a set of parameter / gradient tensors — normally views into the flat
per-shard slabs of :mod:`distributed_tensorflow_amd.variables` — and exposes the
three step flavours every training mode is built from:

* ``step_fused``  forward + loss + backward + in-place SGD (single process;
  reference tfsingle.py:46-49, 78-80) in three kernel launches;
* ``step_grad``   forward + loss + backward into the gradient slab (workers of
  the PS modes; the push/apply happens in the transport);
* ``evaluate``    forward-only accuracy/loss over a large batch (the per-epoch
  test pass, reference tfdist_between.py:109).

The training set stays resident in HBM (55,000 x 784 fp32 = 172 MB of 288 GB),
so a step needs no host->device copy.  The index of the mini-batch inside the
epoch is a plain kernel argument, so a chunk of n steps starting at epoch
position p is captured once into a hipGraph keyed by (p, n) and replayed every
epoch with one host call (``run_chunk``); the resident set is reshuffled in
place between epochs, which keeps those graphs valid.  (A device-side cursor
would cost every kernel a dependent global load — about 1 us — before its first
useful load.)

On CPU tensors every method runs the PyTorch reference implementation with the
same semantics, which is both the soft-placement fallback and the test oracle.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

from ..ops import hip_backend as hb
from ..ops import reference as ref


class _Fetched:
    """A window's results that are already on the host."""

    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


class _PendingFetch:
    """A window's results on their way into a pinned staging row (see MLPEngine.fetch_window_async)."""

    def __init__(self, slot, event, count, has_extra):
        self._slot, self._event, self._count, self._has_extra = slot, event, count, has_extra
        self._value = None

    def result(self):
        if self._value is None:
            self._event.synchronize()
            buf, n = self._slot["buf"], self._count
            self._value = (float(buf[0]), buf[2:2 + n].clone() if n else None,
                           buf[2 + n:2 + 2 * n].clone() if n else None, float(buf[1]) if self._has_extra else None)
            self._slot["free"] = True
        return self._value


class MLPEngine:
    PARAM_NAMES = ("W1", "b1", "W2", "b2")

    def __init__(
        self,
        params: Dict[str, torch.Tensor],
        grads: Optional[Dict[str, torch.Tensor]] = None,
        *,
        batch: int,
        lr: float,
        act="sigmoid",
        hist_cap: int = 8192,
        global_step: Optional[torch.Tensor] = None,
        ksplit: int = 0,
    ):
        self.p = {k: params[k] for k in self.PARAM_NAMES}
        self.g = None if grads is None else {k: grads[k] for k in self.PARAM_NAMES}
        W1, W2 = self.p["W1"], self.p["W2"]
        self.device = W1.device
        self.K, self.H = W1.shape
        self.C = W2.shape[1]
        if W2.shape[0] != self.H or self.p["b1"].numel() != self.H or self.p["b2"].numel() != self.C:
            raise ValueError("inconsistent MLP parameter shapes")
        self.B = int(batch)
        self.act = ref.act_code(act)
        self.on_gpu = self.device.type == "cuda"
        if self.on_gpu:
            hb.require()
        dev = self.device
        self.lr = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.hist_cap = int(hist_cap)
        self.loss_hist = torch.zeros((self.hist_cap,), dtype=torch.float32, device=dev)
        self.acc_hist = torch.zeros((self.hist_cap,), dtype=torch.float32, device=dev)
        self.z2 = torch.zeros((self.B, self.H), dtype=torch.float32, device=dev)
        self.dz2 = torch.zeros((self.B, self.H), dtype=torch.float32, device=dev)
        self.shadow = torch.zeros((self.H * self.C + self.C,), dtype=torch.float32, device=dev)
        # Logit ping-pong accumulators of the two-launch step.  Invariant: before a step
        # that uses half `parity`, that half is all zeros; the step leaves it dirty and
        # clears the other half, so consecutive steps alternate the parity.
        self.zws = torch.zeros((2 * 2048,), dtype=torch.float32, device=dev)
        self.parity = 0
        self.global_step = global_step
        self.ksplit = int(ksplit)
        self.x = None
        self.y = None
        self.batches_per_epoch = 0
        self._host_cursor = 0  # mini-batches consumed so far
        self._desc = None
        self._graphs = {}
        self._eval_ws = None
        self._eval_acc = None
        self._gram = None
        self._pipe = None
        self._pipe_bufs = None
        # one-launch chunks that chain: the batch (position in the epoch) whose forward pass the last chunk left behind
        # together with the transposed image of the current W1, or None (see chunk_pipelined)
        self._carry_pos = None
        self._grad_slots = None
        self._slot_descs = {}

    # -- data ---------------------------------------------------------------
    def bind_data(self, x: torch.Tensor, y: torch.Tensor, batches_per_epoch: Optional[int] = None):
        """Attach resident sample/label matrices; batch b is rows [b*B, (b+1)*B)."""
        if x.device != self.device or y.device != self.device:
            raise ValueError("dataset must live on the engine's device")
        if x.dtype != torch.float32 or y.dtype != torch.float32 or not x.is_contiguous() or not y.is_contiguous():
            raise ValueError("dataset must be contiguous float32")
        if x.shape[1] != self.K or y.shape[1] != self.C or x.shape[0] != y.shape[0]:
            raise ValueError("dataset shape does not match the model")
        nb = x.shape[0] // self.B
        if batches_per_epoch is None:
            batches_per_epoch = nb
        if batches_per_epoch < 1 or batches_per_epoch > nb:
            raise ValueError("not enough rows for the requested batches_per_epoch")
        rebuilt = (self.x is None or self.x.data_ptr() != x.data_ptr() or self.y.data_ptr() != y.data_ptr()
                   or self.batches_per_epoch != batches_per_epoch)
        self.x, self.y, self.batches_per_epoch = x, y, int(batches_per_epoch)
        if rebuilt:
            self._desc = None
            self._graphs.clear()
            self._carry_pos = None

    def bind_gram(self, gram: Optional[torch.Tensor]):
        """Attach the Gram blocks of consecutive mini-batches of the bound (epoch-ordered) set:
        ``gram[j, m, i] = 1 + x_{j+1}[m] . x_j[i]`` ([batches_per_epoch - 1, B, ld], zero for i >= B; see
        :meth:`DeviceDataset.enable_pair_gram`).  With them, fused steps run with ONE launch per step
        (``chunk_pipelined``).  The tensor is read in place: refresh its CONTENTS when the set is reshuffled."""
        self._carry_pos = None
        if gram is None:
            self._gram = self._pipe = None
            return
        if not self.on_gpu:
            raise ValueError("the pipelined chunk is a GPU path")
        if gram.dtype != torch.float32 or not gram.is_contiguous() or gram.device != self.device:
            raise ValueError("gram must be a contiguous float32 tensor on the engine's device")
        nb, b, ld = gram.shape
        if b != self.B or nb < self.batches_per_epoch - 1 or ld % 16 or not (self.B <= ld <= 128):
            raise ValueError("gram shape does not match the bound set")
        if not hb.mlp_chunk_pipelined_supported(self.B, self.K, self.H, self.C):
            raise ValueError("shape not supported by the pipelined chunk")
        self._gram = gram
        self._pipe = None

    @property
    def pipelined(self) -> bool:
        return self._gram is not None

    def _pipe_descriptor(self) -> "hb.DtfPipeDesc":
        if self._pipe is None:
            if self._pipe_bufs is None:
                dev = self.device
                self._pipe_bufs = {
                    "tW1": torch.empty_like(self.p["W1"]), "tb1": torch.empty_like(self.p["b1"]),
                    "tW2": torch.empty_like(self.p["W2"]), "tb2": torch.empty_like(self.p["b2"]),
                    "a2b": torch.zeros((self.B, self.H), dtype=torch.float32, device=dev),
                    "zws3": torch.zeros((3 * 2048,), dtype=torch.float32, device=dev),
                    "w1t": torch.zeros((2, self.H, self.K), dtype=torch.float32, device=dev),
                }
            q = hb.DtfPipeDesc()
            for k, t in self._pipe_bufs.items():
                setattr(q, k, t.data_ptr())
            q.gram = self._gram.data_ptr()
            q.gram_ld = self._gram.shape[2]
            self._pipe = q
        return self._pipe

    def chunk_pipelined(self, pos: int, n: int, carry_in: bool = False, carry_out: bool = False):
        """Enqueue n fused steps on batches pos .. pos+n-1 of the epoch, one launch per step (no cursor change).

        Chunks chain: with ``carry_out`` the last launch also runs the forward pass of batch pos+n and keeps W1 in
        both layouts, and the next chunk, started at pos+n with ``carry_in``, consists of its n step launches only (no
        transposition, no stand-alone forward).  ``carry_in`` is the CALLER's promise that nothing touched the
        parameters, the bound set or this engine in between (``carried_to`` says where the last chunk left off; every
        other step method, a rebind and ``invalidate_carry()`` reset it).  When a chunk is captured into a graph the
        bookkeeping is the replaying caller's (``note_carry``)."""
        if self._gram is None:
            raise RuntimeError("bind_gram() first")
        if pos < 0 or pos + n > self.batches_per_epoch:
            raise ValueError("a pipelined chunk must stay inside one epoch")
        if carry_out and pos + n >= self.batches_per_epoch:
            raise ValueError("no batch after the chunk in this epoch to run the forward pass of")
        flags = (hb.PIPE_CARRY_IN if carry_in else 0) | (hb.PIPE_CARRY_OUT if carry_out else 0)
        hb.mlp_chunk_pipelined(self._descriptor(), self._pipe_descriptor(), pos, n, flags)
        if not torch.cuda.is_current_stream_capturing():
            self._carry_pos = pos + n if carry_out else None

    @property
    def carried_to(self) -> Optional[int]:
        """Position in the epoch at which a chunk started with ``carry_in`` would be valid, or None."""
        return self._carry_pos

    def note_carry(self, pos: Optional[int]):
        """Record what a replayed graph of ``chunk_pipelined(..., carry_out=...)`` left behind."""
        self._carry_pos = pos

    def invalidate_carry(self):
        """Call after writing the parameters or the bound set behind the engine's back."""
        self._carry_pos = None

    def set_cursor(self, value: int):
        self._host_cursor = int(value)

    def set_lr(self, lr: float):
        self.lr.fill_(float(lr))

    # -- descriptor -----------------------------------------------------------
    def _descriptor(self) -> "hb.DtfMlpDesc":
        if self._desc is not None:
            return self._desc
        if self.x is None:
            raise RuntimeError("bind_data() must be called before stepping")
        d = hb.DtfMlpDesc()
        d.in_dim, d.hidden, d.classes, d.batch = self.K, self.H, self.C, self.B
        d.act, d.ksplit = self.act, self.ksplit
        d.batches_per_epoch, d.hist_cap = self.batches_per_epoch, self.hist_cap
        d.x, d.y = self.x.data_ptr(), self.y.data_ptr()
        for k in self.PARAM_NAMES:
            t = self.p[k]
            if not t.is_contiguous() or t.dtype != torch.float32:
                raise ValueError(f"parameter {k} must be contiguous float32")
            setattr(d, k, t.data_ptr())
            if self.g is not None:
                gt = self.g[k]
                if not gt.is_contiguous() or gt.shape != t.shape:
                    raise ValueError(f"gradient {k} must be contiguous and parameter-shaped")
                setattr(d, "g" + k, gt.data_ptr())
        d.z2, d.dz2 = self.z2.data_ptr(), self.dz2.data_ptr()
        d.cursor, d.lr = None, self.lr.data_ptr()
        d.loss_hist, d.acc_hist = self.loss_hist.data_ptr(), self.acc_hist.data_ptr()
        d.global_step = self.global_step.data_ptr() if self.global_step is not None else None
        d.shadow = self.shadow.data_ptr()
        d.zws = self.zws.data_ptr()
        self._desc = d
        return d

    # -- CPU reference path -----------------------------------------------------
    def _cpu_batch(self, batch_index: int):
        b = batch_index % self.batches_per_epoch
        sl = slice(b * self.B, (b + 1) * self.B)
        return self.x[sl], self.y[sl]

    def _cpu_step(self, batch_index: int, fused: bool, scale: float):
        xb, yb = self._cpu_batch(batch_index)
        out = ref.mlp_grads(xb, yb, self.p["W1"], self.p["b1"], self.p["W2"], self.p["b2"], self.act)
        slot = (batch_index % self.batches_per_epoch) % self.hist_cap
        self.loss_hist[slot] = out["loss"]
        self.acc_hist[slot] = out["acc"]
        lr = float(self.lr.item())
        for k in self.PARAM_NAMES:
            if fused:
                self.p[k].sub_(out["d" + k], alpha=lr)
            else:
                self.g[k].copy_(out["d" + k] * scale)
        if fused and self.global_step is not None:
            self.global_step.add_(1.0)

    # -- steps ----------------------------------------------------------------
    def _bi(self, batch_index: Optional[int]) -> int:
        if batch_index is None:
            batch_index = self._host_cursor
        return int(batch_index) % self.batches_per_epoch

    def step_fused(self, batch_index: Optional[int] = None):
        """One SGD step on mini-batch ``batch_index`` of the epoch (default: the cursor's)."""
        self._carry_pos = None
        bi = self._bi(batch_index)
        if self.on_gpu:
            hb.mlp_step_fused(self._descriptor(), bi, self.parity)
            self.parity ^= 1
        else:
            self._cpu_step(bi, True, 1.0)

    def bind_grad_slots(self, slots):
        """Alternative gradient outputs: ``slots[s]`` is a dict like ``grads``; ``step_grad(slot=s)`` writes
        there.  Used by exchanges that double-buffer the gradient (peers read step t's while t+1 is computed)."""
        for g in slots:
            for k in self.PARAM_NAMES:
                if g[k].shape != self.p[k].shape or not g[k].is_contiguous() or g[k].dtype != torch.float32:
                    raise ValueError(f"gradient slot {k} must be contiguous float32 and parameter-shaped")
        self._grad_slots = [{k: g[k] for k in self.PARAM_NAMES} for g in slots]
        self._slot_descs = {}

    def _slot_descriptor(self, slot: int) -> "hb.DtfMlpDesc":
        base = self._descriptor()
        cached = self._slot_descs.get(slot)
        if cached is not None and cached[0] is base:
            return cached[1]
        d = hb.DtfMlpDesc.from_buffer_copy(base)
        for k in self.PARAM_NAMES:
            setattr(d, "g" + k, self._grad_slots[slot][k].data_ptr())
        self._slot_descs[slot] = (base, d)
        return d

    def step_grad(self, batch_index: Optional[int] = None, scale: float = 1.0, slot: Optional[int] = None,
                  shards: Optional["hb.DtfXgDesc"] = None):
        """Gradients of mini-batch ``batch_index`` into the gradient tensors (or into gradient slot ``slot``,
        see :meth:`bind_grad_slots`); no update.  ``shards`` (GPU): the forward reads the variables from these
        mapped parameter-server shards instead of the engine's own copy -- no pull in front of the step."""
        self._carry_pos = None
        if self.g is None and slot is None:
            raise RuntimeError("engine was built without gradient tensors")
        bi = self._bi(batch_index)
        if self.on_gpu:
            d = self._descriptor() if slot is None else self._slot_descriptor(slot)
            if shards is not None:
                hb.mlp_step_grad_shards(d, bi, self.parity, scale, shards)
            else:
                hb.mlp_step_grad(d, bi, self.parity, scale)
            self.parity ^= 1
        elif slot is None:
            self._cpu_step(bi, False, scale)
        else:
            keep, self.g = self.g, self._grad_slots[slot]
            try:
                self._cpu_step(bi, False, scale)
            finally:
                self.g = keep

    def step_xg(self, batch_index: Optional[int], scale: float, xg: "hb.DtfXgDesc"):
        """Fused step of a synchronous replica: the backward's workgroups exchange their partial gradients
        (scaled by ``scale`` = 1/N) with the peers described by ``xg`` and apply the rank-ordered sum in place."""
        self._carry_pos = None
        if not self.on_gpu:
            raise RuntimeError("the cross-replica step is a GPU path")
        hb.mlp_step_xg(self._descriptor(), self._bi(batch_index), self.parity, scale, xg)
        self.parity ^= 1

    def step_async(self, batch_index: Optional[int], shards: "hb.DtfXgDesc"):
        """Step of an asynchronous worker: forward, then a backward whose threads apply ``-lr * g`` of their own
        gradient element directly to the parameter shard that owns it (``PeerShards.step_descriptor``).  The
        engine's own parameters are only read; the caller refreshes them with a pull."""
        self._carry_pos = None
        if not self.on_gpu:
            raise RuntimeError("the shard-pushing step is a GPU path")
        hb.mlp_step_async(self._descriptor(), self._bi(batch_index), self.parity, shards)
        self.parity ^= 1

    def advance(self, n: int = 1):
        self._host_cursor += n

    def run_chunk(self, n: int, use_graph: bool = True):
        """``n`` fused steps on consecutive batches starting at the cursor, then cursor += n.

        On the GPU the 3n launches are captured once per (epoch position, n)
        into a hipGraph and replayed; the host issues a single call per chunk.
        """
        if n <= 0:
            return
        pos = self._host_cursor % self.batches_per_epoch
        if self.on_gpu and self._gram is not None and pos + n <= self.batches_per_epoch:
            if not use_graph:
                self.chunk_pipelined(pos, n)
            else:
                key = ("pipe", pos, n)
                g = self._graphs.get(key)
                if g is None:
                    g = self._capture_pipe(pos, n)
                    self._graphs[key] = g
                g.replay()
                self._carry_pos = None
            self._host_cursor += n
            return
        if not self.on_gpu or not use_graph:
            for i in range(n):
                self.step_fused(pos + i)
            self.advance(n)
            return
        key = (pos, n, self.parity)
        g = self._graphs.get(key)
        if g is None:
            g = self._capture(pos, n)
            self._graphs[key] = g
        g.replay()
        self.parity ^= n & 1
        self._host_cursor += n

    def _capture(self, pos: int, n: int) -> "torch.cuda.CUDAGraph":
        self._descriptor()
        # Warm up on a side stream (module load, attribute set) without
        # disturbing training state: snapshot and restore what a step mutates.
        saved = {k: v.clone() for k, v in self.p.items()}
        saved_gs = self.global_step.clone() if self.global_step is not None else None
        saved_hist = (self.loss_hist.clone(), self.acc_hist.clone())
        parity0 = self.parity
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self.step_fused(pos)
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        for k, v in saved.items():
            self.p[k].copy_(v)
        if saved_gs is not None:
            self.global_step.copy_(saved_gs)
        self.loss_hist.copy_(saved_hist[0])
        self.acc_hist.copy_(saved_hist[1])
        self.reset_step_workspace(parity0)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(n):
                hb.mlp_step_fused(self._desc, (pos + i) % self.batches_per_epoch, parity0 ^ (i & 1))
        return graph

    def _capture_pipe(self, pos: int, n: int, carry_in: bool = False, carry_out: bool = False) -> "torch.cuda.CUDAGraph":
        self._descriptor()
        # Warm up (module load) on a side stream with the training state rolled back afterwards.
        saved = {k: v.clone() for k, v in self.p.items()}
        saved_gs = self.global_step.clone() if self.global_step is not None else None
        saved_hist = (self.loss_hist.clone(), self.acc_hist.clone())
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self.chunk_pipelined(pos, min(n, 2))
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        for k, v in saved.items():
            self.p[k].copy_(v)
        if saved_gs is not None:
            self.global_step.copy_(saved_gs)
        self.loss_hist.copy_(saved_hist[0])
        self.acc_hist.copy_(saved_hist[1])
        self._carry_pos = None          # the warm-up steps overwrote whatever a previous chunk had left behind
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.chunk_pipelined(pos, n, carry_in, carry_out)
        if os.environ.get("DTF_GRAPH_UPLOAD", "1") != "0":
            hb.graph_upload(graph)      # the first replay may sit inside somebody's timed region
        return graph

    def reset_step_workspace(self, parity: int = 0):
        """Re-establish the ping-pong invariant (after warm-up steps whose effects are rolled back)."""
        self._carry_pos = None
        self.zws.zero_()
        self.parity = int(parity) & 1

    # -- evaluation ---------------------------------------------------------------
    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[float, float]:
        """Mean loss and accuracy of the current parameters on (x, y); syncs the host."""
        if not self.on_gpu:
            loss, acc = ref.mlp_loss_acc(x, y, self.p["W1"], self.p["b1"], self.p["W2"], self.p["b2"], self.act)
            return float(loss), float(acc)
        M = x.shape[0]
        if self._eval_ws is None or self._eval_ws.numel() < M * self.H:
            self._eval_ws = torch.empty((M * self.H,), dtype=torch.float32, device=self.device)
            self._eval_acc = (torch.zeros((1,), dtype=torch.float32, device=self.device),
                              torch.zeros((1,), dtype=torch.int32, device=self.device))
        loss_sum, correct = self._eval_acc
        loss_sum.zero_()
        correct.zero_()
        hb.mlp_eval(x, y, self.p["W1"], self.p["b1"], self.p["W2"], self.p["b2"], self.act, self._eval_ws,
                    loss_sum, correct)
        return float(loss_sum.item()) / M, float(correct.item()) / M

    # -- history --------------------------------------------------------------------
    def _slot(self, step: int) -> int:
        return (step % self.batches_per_epoch) % self.hist_cap

    def history(self, first_step: int, count: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per-step (loss, batch accuracy) for steps [first_step, first_step+count) as CPU tensors.

        The ring is indexed by position in the epoch, so a window must be read
        before the same positions are trained on again.
        """
        if count > min(self.hist_cap, self.batches_per_epoch):
            raise ValueError("history window larger than the ring")
        idx = torch.tensor([self._slot(s) for s in range(first_step, first_step + count)],
                           dtype=torch.long, device=self.device)
        return self.loss_hist[idx].cpu(), self.acc_hist[idx].cpu()

    def last_loss(self) -> float:
        return float(self.loss_hist[self._slot(self._host_cursor - 1)].item())

    def fetch_window_async(self, first_step: int, count: int, want_history: bool = False, extra=None):
        """Queue the copies of a finished window's results behind it and return a handle; ``handle.result()`` waits
        for those copies only, so the caller can launch the next window first and read this one while it runs.

        result() -> (cost of the window's last step, per-step losses, per-step accuracies, extra value) with the two
        series None unless ``want_history``; ``extra``: an optional one-element device tensor (the step counter)
        copied at the same point of the stream."""
        last = self._slot(first_step + count - 1)
        if not self.on_gpu:
            losses = accs = None
            if want_history:
                losses, accs = self.history(first_step, count)
            return _Fetched((float(self.loss_hist[last]), losses, accs, None if extra is None else float(extra.item())))
        pool = getattr(self, "_fetch_pool", None)
        if pool is None:
            pool = self._fetch_pool = []
        # a pinned staging row per window in flight (reused once its copy has been read)
        n = 2 + (2 * count if want_history else 0)
        slot = next((b for b in pool if b["free"] and b["buf"].numel() >= n), None)
        if slot is None:
            slot = {"buf": torch.empty(max(n, 2 + 2 * min(self.hist_cap, self.batches_per_epoch)),
                                       dtype=torch.float32).pin_memory(), "free": True}
            pool.append(slot)
        slot["free"] = False
        buf = slot["buf"]
        buf[0:1].copy_(self.loss_hist[last:last + 1], non_blocking=True)
        if extra is not None:
            buf[1:2].copy_(extra.reshape(1).to(torch.float32), non_blocking=True)
        if want_history:
            if count > min(self.hist_cap, self.batches_per_epoch):
                raise ValueError("history window larger than the ring")
            idx = torch.tensor([self._slot(s) for s in range(first_step, first_step + count)], dtype=torch.long)
            idx = idx.to(self.device, non_blocking=True)
            buf[2:2 + count].copy_(self.loss_hist[idx], non_blocking=True)
            buf[2 + count:2 + 2 * count].copy_(self.acc_hist[idx], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return _PendingFetch(slot, ev, count if want_history else 0, extra is not None)
