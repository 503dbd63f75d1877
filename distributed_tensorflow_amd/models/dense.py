"""Fully connected stacks of any depth and width on the hand-written dense-layer kernels.

The reference trains exactly one network, 784-100-10 (tfdist_between.py:48-59); ``MLPEngine`` is built around that
shape (one or two launches per step).  ``DenseStackEngine`` is the general form of the same graph,

    a_0 = x;   a_i = act_i(a_{i-1} W_i + b_i)  (i = 1 .. L-1);   logits = a_{L-1} W_L + b_L;
    loss = mean(-sum(y_ * log softmax(logits), 1));   accuracy = mean(argmax(logits) == argmax(y_))

for any number of layers, any widths, any number of classes and sigmoid / relu / tanh / linear hidden layers, with
every launch a kernel of ``csrc/gemm.hip``:

* forward   one GEMM per layer with bias + activation in its epilogue,
* head      softmax cross-entropy, accuracy and d(loss)/d(logits) in one launch,
* backward  per layer, from the top: the input gradient as an NT GEMM whose epilogue multiplies by the activation's
            derivative (taken from the activation's OUTPUT, so pre-activations are never kept), then the weight
            gradient as a TN GEMM whose epilogue IS the optimizer (``W -= lr * x^T dz``, the ApplyGradientDescent of
            tfdist_between.py:65-66) or a plain store into a gradient tensor (parameter-server modes), and the bias
            gradient as a column sum with the same two endings.

3 L + 1 launches per fused step of L layers: L forward GEMMs, the head (its last workgroup to arrive adds the
per-workgroup sums in a fixed order, stores the step's loss and accuracy counts and bumps the step counter: nothing is
cleared beforehand; batches of up to 16 rows use a one-workgroup head), L - 1 input-gradient and L weight-gradient GEMMs,
and ONE column-sum launch for the bias gradients of all layers (batches beyond 4096 rows: one per
layer, with row segments), captured into hipGraph chunks by ``run_chunk`` when the mini-batches come from a resident set.
All launches go to the current stream in that order.  (``DTF_DENSE_OVERLAP=1`` captures a step as a graph with BRANCHES
instead — weight gradients next to the input-gradient GEMMs of the layers below them, bias sums next to the last weight
gradients — which shortens the critical path on paper and measured slower: edges
between the streams of a hipGraph cost more than the launches they take off the chain.)  On CPU tensors every method runs the same mathematics
in PyTorch, which is the soft-placement path and the test oracle.
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..ops import hip_backend as hb

ACT_CODES = {"sigmoid": 0, "relu": 1, "tanh": 2, "linear": hb.ACT_NONE, "identity": hb.ACT_NONE, None: hb.ACT_NONE}


def _act_code(act) -> int:
    if isinstance(act, int):
        return act
    return ACT_CODES[act]


def _act_fwd(z: torch.Tensor, act: int) -> torch.Tensor:
    if act == 0:
        return torch.sigmoid(z)
    if act == 1:
        return torch.relu(z)
    if act == 2:
        return torch.tanh(z)
    return z


def _act_bwd_from_output(a: torch.Tensor, act: int) -> torch.Tensor:
    if act == 0:
        return a * (1.0 - a)
    if act == 1:
        return (a > 0).to(a.dtype)
    if act == 2:
        return 1.0 - a * a
    return torch.ones_like(a)


class DenseStackEngine:
    """``weights[i]``: [n_{i}, n_{i+1}] float32, ``biases[i]``: [n_{i+1}] or None, ``acts``: activation of every layer
    but the last (names or codes).  ``grads``: optional list of (gW, gb) tensors for ``step_grad``.  Parameters are
    normally views into the flat per-shard slabs of :mod:`distributed_tensorflow_amd.variables`."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], acts: Sequence,
                 *, batch: int, lr: float, grads: Optional[Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]]] = None,
                 global_step: Optional[torch.Tensor] = None, hist_cap: int = 8192):
        self.W = list(weights)
        self.b = list(biases)
        self.L = len(self.W)
        if self.L < 1 or len(self.b) != self.L or len(acts) != self.L - 1:
            raise ValueError("a stack of L layers has L weights, L biases (or None) and L - 1 hidden activations")
        self.acts = [_act_code(a) for a in acts]
        self.device = self.W[0].device
        self.dims = [self.W[0].shape[0]] + [w.shape[1] for w in self.W]
        for i, w in enumerate(self.W):
            if w.dim() != 2 or w.shape[0] != self.dims[i] or w.dtype != torch.float32 or w.device != self.device \
                    or not w.is_contiguous():
                raise ValueError(f"weight {i} must be a contiguous float32 [{self.dims[i]}, n] matrix on {self.device}")
            bi = self.b[i]
            if bi is not None and (bi.numel() != w.shape[1] or bi.dtype != torch.float32 or bi.device != self.device):
                raise ValueError(f"bias {i} must have {w.shape[1]} float32 elements on {self.device}")
        self.g = None
        if grads is not None:
            self.g = [(gw, gb) for gw, gb in grads]
            for i, (gw, gb) in enumerate(self.g):
                if gw.shape != self.W[i].shape or not gw.is_contiguous() or (self.b[i] is not None and gb is None):
                    raise ValueError(f"gradient tensors of layer {i} must be shaped like its parameters")
        self.B = int(batch)
        self.on_gpu = self.device.type == "cuda"
        if self.on_gpu:
            hb.require()
        dev = self.device
        self.lr = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.global_step = global_step
        self.hist_cap = int(hist_cap)
        # per-step sums, indexed by position in the epoch: row loss sum and number of correct rows
        self.loss_sum = torch.zeros((self.hist_cap,), dtype=torch.float32, device=dev)
        self.correct = torch.zeros((self.hist_cap,), dtype=torch.int32, device=dev)
        # activations a_1 .. a_{L-1}, logits, and their gradients (dz_i shaped like a_i)
        self.a = [torch.empty((self.B, n), dtype=torch.float32, device=dev) for n in self.dims[1:]]
        self.dz = [torch.empty((self.B, n), dtype=torch.float32, device=dev) for n in self.dims[1:]]
        self.x = self.y = None
        self.batches_per_epoch = 0
        self._host_cursor = 0
        self._graphs: Dict[tuple, "torch.cuda.CUDAGraph"] = {}
        self._eval_bufs: Dict[int, list] = {}
        self._colsum_plans: Dict[str, object] = {}
        # K-split workspace of this engine's GEMMs (a batch-100 layer has 26 tiles for 256 CUs): 34 MB, one per engine,
        # so two engines on two streams never share counters
        self._ws = hb.gemm_workspace(dev) if self.on_gpu else None
        # captured steps as graphs with branches: measured SLOWER than the chain (a dependency between two streams of a
        # hipGraph costs more than the 1.6 us between two launches of one stream: 76 against 59 us per step of
        # 784-100-10, 122 against 114 us of 784-1024-1024-10 at batch 100), so it is an option, not the default
        self.overlap = os.environ.get("DTF_DENSE_OVERLAP", "0") == "1"
        self._head_scratch = hb.xent_head_scratch(self.B, dev) if self.on_gpu else None
        self._side = self._ws_side = None

    @property
    def launches_per_step(self) -> int:
        """L forward GEMMs, the head (which also stores the step's statistics and bumps the step counter), L - 1 input
        gradients, L weight gradients, one launch for the bias gradients (batches beyond 4096 rows: one per bias)."""
        biases = sum(1 for b in self.b if b is not None)
        return 3 * self.L + (min(1, biases) if self.B <= 4096 else biases)

    # -- data -----------------------------------------------------------------------------------------------------
    def bind_data(self, x: torch.Tensor, y: torch.Tensor, batches_per_epoch: Optional[int] = None):
        """Attach resident sample / label matrices; batch b is rows [b*B, (b+1)*B)."""
        if x.device != self.device or y.device != self.device:
            raise ValueError("dataset must live on the engine's device")
        if x.dtype != torch.float32 or y.dtype != torch.float32 or not x.is_contiguous() or not y.is_contiguous():
            raise ValueError("dataset must be contiguous float32")
        if x.shape[1] != self.dims[0] or y.shape[1] != self.dims[-1] or x.shape[0] != y.shape[0]:
            raise ValueError("dataset shape does not match the model")
        nb = x.shape[0] // self.B
        bpe = nb if batches_per_epoch is None else int(batches_per_epoch)
        if bpe < 1 or bpe > nb:
            raise ValueError("not enough rows for the requested batches_per_epoch")
        if self.x is None or self.x.data_ptr() != x.data_ptr() or self.y.data_ptr() != y.data_ptr() \
                or self.batches_per_epoch != bpe:
            self._graphs.clear()
        self.x, self.y, self.batches_per_epoch = x, y, bpe

    def bind_grads(self, grads: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]]):
        """Point ``step_grad`` at another set of gradient tensors (e.g. the two slots of a double-buffered exchange)."""
        grads = [(gw, gb) for gw, gb in grads]
        for i, (gw, gb) in enumerate(grads):
            if gw.shape != self.W[i].shape or not gw.is_contiguous() or (self.b[i] is not None and gb is None):
                raise ValueError(f"gradient tensors of layer {i} must be shaped like its parameters")
        self.g = grads

    def set_lr(self, lr: float):
        self.lr.fill_(float(lr))

    def set_cursor(self, value: int):
        self._host_cursor = int(value)

    def _batch(self, bi: int):
        sl = slice(bi * self.B, (bi + 1) * self.B)
        return self.x[sl], self.y[sl]

    def _slot(self, step: int) -> int:
        return (step % self.batches_per_epoch) % self.hist_cap

    # -- one step -------------------------------------------------------------------------------------------------
    def _forward_gpu(self, x: torch.Tensor, acts_out: List[torch.Tensor]):
        h = x
        for i in range(self.L):
            act = self.acts[i] if i < self.L - 1 else hb.ACT_NONE
            hb.gemm(hb.GEMM_NN, h, self.W[i], out=acts_out[i], epi=hb.EPI_BIAS_ACT, act=act, bias=self.b[i],
                    workspace=self._ws)
            h = acts_out[i]
        return h

    def _branch_streams(self):
        """Three side streams with a K-split workspace each (GEMMs that run at the same time must not share counters)."""
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(3)]
            self._ws_side = [hb.gemm_workspace(self.device) for _ in range(2)]
        return self._side, self._ws_side

    def _step_gpu(self, bi: int, fused: bool, scale: float):
        """With ``self.overlap`` a CAPTURED step is a graph with branches, not a chain: the weight gradient of layer i
        needs only dz_i and the activations below, so it runs NEXT TO the input-gradient GEMMs of the layers below it
        (after the one of its own layer, which still reads the old W_i), the bias column sums run next to the last
        weight gradients.  Eager launches stay on one stream."""
        x, y = self._batch(bi)
        slot = self._slot(bi)
        loss_slot, correct_slot = self.loss_sum[slot:slot + 1], self.correct[slot:slot + 1]
        gs = self.global_step if fused else None
        gscale = (1.0 if fused else scale) / self.B
        main = torch.cuda.current_stream(self.device)
        branch = self.overlap and torch.cuda.is_current_stream_capturing()
        side, ws_side = self._branch_streams() if branch else (None, None)

        used = set()

        def fork(s):        # stream s goes on from this point of the main stream
            ev = torch.cuda.Event()
            ev.record(main)
            s.wait_event(ev)

        logits = self._forward_gpu(x, self.a)
        if self.B <= hb.XENT_STEP_MAX_ROWS:
            # one workgroup: loss and count stored in a fixed order, the step counter bumped in the same launch
            hb.softmax_xent_step(logits, y, self.dz[-1], gscale, loss_slot, correct_slot, gs)
        else:
            # four rows per workgroup; the last workgroup to arrive adds the partial sums in a fixed order, stores them
            # and bumps the step counter: nothing to clear beforehand, no launch of its own for the counter
            hb.softmax_xent_head(logits, y, self.dz[-1], gscale, self._head_scratch, loss_slot, correct_slot, gs)
        for i in range(self.L - 1, -1, -1):
            below = self.a[i - 1] if i > 0 else x
            if i > 0:       # before W_i changes: dz_{i-1} = (dz_i W_i^T) * act'(a_{i-1})
                hb.gemm(hb.GEMM_NT, self.dz[i], self.W[i], out=self.dz[i - 1], epi=hb.EPI_ACT_BWD,
                        act=self.acts[i - 1], aux=below, workspace=self._ws)
            if branch:
                fork(side[i & 1])
                used.add(i & 1)
            with torch.cuda.stream(side[i & 1]) if branch else contextlib.nullcontext():
                ws = ws_side[i & 1] if branch else self._ws
                if fused:
                    hb.gemm(hb.GEMM_TN, below, self.dz[i], out=self.W[i], epi=hb.EPI_SGD, lr=self.lr, scale=1.0,
                            workspace=ws)
                else:
                    hb.gemm(hb.GEMM_TN, below, self.dz[i], out=self.g[i][0], workspace=ws)
        if branch:
            fork(side[2])
            with torch.cuda.stream(side[2]):
                self._bias_gradients(fused)
            for k in sorted(used | {2}):
                main.wait_stream(side[k])
        else:
            self._bias_gradients(fused)

    def _bias_gradients(self, fused: bool):
        """Column sums of every layer's dz into the biases (``b -= lr * sum``) or the bias-gradient tensors: ONE launch
        at the end of the backward pass (every dz is still in place); batches beyond 4096 rows take a launch per layer,
        whose row segments keep the tall sums parallel."""
        pairs = [(self.dz[i], self.b[i] if fused else self.g[i][1]) for i in range(self.L) if self.b[i] is not None]
        if not pairs:
            return
        if self.B > 4096 or len(pairs) > 16:
            for z, out in pairs:
                hb.colsum(z, out=out, sgd=fused, lr=self.lr if fused else None)
            return
        key = "fused" if fused else ("grad", pairs[0][1].data_ptr())     # (gradient tensors can be re-bound)
        plan = self._colsum_plans.get(key)
        if plan is None:
            plan = self._colsum_plans[key] = hb.ColsumPlan(pairs)
        hb.colsum_multi(plan, sgd=fused, lr=self.lr if fused else None)

    def _step_cpu(self, bi: int, fused: bool, scale: float):
        x, y = self._batch(bi)
        acts = [x]
        for i in range(self.L):
            z = acts[-1] @ self.W[i]
            if self.b[i] is not None:
                z = z + self.b[i]
            acts.append(_act_fwd(z, self.acts[i]) if i < self.L - 1 else z)
        logits = acts[-1]
        logp = torch.log_softmax(logits, dim=1)
        slot = self._slot(bi)
        self.loss_sum[slot] = -(y * logp).sum()
        self.correct[slot] = int((logits.argmax(1) == y.argmax(1)).sum())
        dz = (logp.exp() * y.sum(1, keepdim=True) - y) * ((1.0 if fused else scale) / self.B)
        lr = float(self.lr.item())
        for i in range(self.L - 1, -1, -1):
            below = acts[i]
            dz_below = (dz @ self.W[i].t()) * _act_bwd_from_output(below, self.acts[i - 1]) if i > 0 else None
            gw, gb = below.t() @ dz, dz.sum(0)
            if fused:
                self.W[i].sub_(gw, alpha=lr)
                if self.b[i] is not None:
                    self.b[i].sub_(gb, alpha=lr)
            else:
                self.g[i][0].copy_(gw)
                if self.b[i] is not None:
                    self.g[i][1].copy_(gb.view_as(self.g[i][1]))
            dz = dz_below
        if fused and self.global_step is not None:
            self.global_step.add_(1.0)

    def _bi(self, batch_index: Optional[int]) -> int:
        if self.x is None:
            raise RuntimeError("bind_data() must be called before stepping")
        return int(self._host_cursor if batch_index is None else batch_index) % self.batches_per_epoch

    def step_fused(self, batch_index: Optional[int] = None):
        """One SGD step on mini-batch ``batch_index`` of the epoch (default: the cursor's); parameters updated in place."""
        (self._step_gpu if self.on_gpu else self._step_cpu)(self._bi(batch_index), True, 1.0)

    def step_grad(self, batch_index: Optional[int] = None, scale: float = 1.0):
        """Gradients (times ``scale``) of mini-batch ``batch_index`` into the gradient tensors; no update."""
        if self.g is None:
            raise RuntimeError("engine was built without gradient tensors")
        (self._step_gpu if self.on_gpu else self._step_cpu)(self._bi(batch_index), False, scale)

    def advance(self, n: int = 1):
        self._host_cursor += n

    def run_chunk(self, n: int, use_graph: bool = True):
        """``n`` fused steps on consecutive batches starting at the cursor (wrapping at the end of the epoch), then
        cursor += n.  On the GPU the launches are captured once per (epoch position, n) and replayed."""
        if n <= 0:
            return
        pos = self._host_cursor % self.batches_per_epoch
        if not self.on_gpu or not use_graph:
            for i in range(n):
                self.step_fused(pos + i)
            self._host_cursor += n
            return
        key = (pos, n)
        g = self._graphs.get(key)
        if g is None:
            g = self._capture(pos, n)
            self._graphs[key] = g
        g.replay()
        self._host_cursor += n

    def _capture(self, pos: int, n: int) -> "torch.cuda.CUDAGraph":
        # warm up on a side stream (module load, kernel attributes) with the training state rolled back afterwards
        saved = [(w.clone(), None if b is None else b.clone()) for w, b in zip(self.W, self.b)]
        saved_gs = self.global_step.clone() if self.global_step is not None else None
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self.step_fused(pos)
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        for (w0, b0), w, b in zip(saved, self.W, self.b):
            w.copy_(w0)
            if b is not None:
                b.copy_(b0)
        if saved_gs is not None:
            self.global_step.copy_(saved_gs)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(n):
                self._step_gpu((pos + i) % self.batches_per_epoch, True, 1.0)
        return graph

    # -- evaluation / history ----------------------------------------------------------------------------------------
    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[float, float]:
        """Mean loss and accuracy of the current parameters on (x, y); syncs the host."""
        M = x.shape[0]
        if not self.on_gpu:
            h = x
            for i in range(self.L):
                z = h @ self.W[i]
                if self.b[i] is not None:
                    z = z + self.b[i]
                h = _act_fwd(z, self.acts[i]) if i < self.L - 1 else z
            logp = torch.log_softmax(h, dim=1)
            return float(-(y * logp).sum() / M), float((h.argmax(1) == y.argmax(1)).float().mean())
        bufs = self._eval_bufs.get(M)
        if bufs is None:
            self._eval_bufs.clear()
            bufs = [torch.empty((M, n), dtype=torch.float32, device=self.device) for n in self.dims[1:]]
            bufs.append(torch.zeros((1,), dtype=torch.float32, device=self.device))
            bufs.append(torch.zeros((1,), dtype=torch.int32, device=self.device))
            self._eval_bufs[M] = bufs
        loss_sum, correct = bufs[-2], bufs[-1]
        loss_sum.zero_()
        correct.zero_()
        logits = self._forward_gpu(x.contiguous(), bufs[:self.L])
        hb.softmax_xent(logits, y.contiguous(), loss_sum=loss_sum, correct=correct)
        return float(loss_sum.item()) / M, float(correct.item()) / M

    def history(self, first_step: int, count: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per-step (mean loss, batch accuracy) of steps [first_step, first_step + count) as CPU tensors (the ring is
        indexed by position in the epoch: read a window before the same positions are trained on again)."""
        if count > min(self.hist_cap, self.batches_per_epoch):
            raise ValueError("history window larger than the ring")
        idx = torch.tensor([self._slot(s) for s in range(first_step, first_step + count)], dtype=torch.long,
                           device=self.device)
        return (self.loss_sum[idx] / self.B).cpu(), (self.correct[idx].float() / self.B).cpu()

    def last_loss(self) -> float:
        return float(self.loss_sum[self._slot(self._host_cursor - 1)].item()) / self.B

    def last_stats(self, batch_index: int = None) -> Tuple[float, float]:
        """(mean loss, batch accuracy) of the step on ``batch_index`` (default: the last one the cursor passed)."""
        slot = self._slot(self._host_cursor - 1 if batch_index is None else batch_index)
        return float(self.loss_sum[slot].item()) / self.B, float(self.correct[slot].item()) / self.B
