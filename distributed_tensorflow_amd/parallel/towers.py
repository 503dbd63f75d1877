"""In-graph replication: ONE client process driving a tower on every GPU.

The reference abandoned this mode (README.md:261-264); it is provided here as
its own design, in two forms:

* :class:`InGraphEngine` — the engine.  Every tower has its own device, HIP
  stream, replica of the variables, independently shuffled resident data set
  and hipGraph chunks of the two-launch cross-replica step: the gradient mean
  is taken INSIDE the towers' backward launches, every thread storing its
  gradient element straight into the other towers' inboxes (peer access between
  the GPUs of one process: plain pointers, no IPC handles) and polling its own.
  The towers' graphs therefore need nothing from the host between them: the
  client issues ONE graph launch per tower per chunk of steps and synchronises
  only when it reads a cost.
* :class:`InGraphTowers` — the library baseline: a host loop that launches the
  fused forward/backward on every tower, all-reduces the gradient slabs with
  ``torch.cuda.nccl.all_reduce`` (the single-process multi-GPU entry of RCCL)
  and applies the sharded-SGD kernel on every tower.

InGraphTowers: the single client holds one replica of the variables per
tower, gives every tower its own mini-batch of the resident data set, and per
step

1. launches the fused forward/backward on every tower (gradients pre-scaled by
   1/N into that tower's gradient slab),
2. **all-reduces** the N gradient slabs — ``torch.cuda.nccl.all_reduce``, the
   single-process multi-GPU entry of RCCL, over xGMI — so that every tower
   holds the mean gradient,
3. applies the identical sharded-SGD kernel on every tower (replicas stay
   bit-identical; ``global_step`` advances once).

All launches of a step are asynchronous; the client synchronises only when it
reads the cost.  Towers on CPU devices (tests) or several towers on one GPU
reduce through a plain sum instead of RCCL.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from ..data import DeviceDataset
from ..models.mlp import MLPEngine
from ..ops import hip_backend as hb
from ..ops import reference as ref
from ..variables import ShardLayout, assign_shards, mlp_var_specs


class InGraphTowers:
    def __init__(self, devices: Sequence, images, labels, batch: int, lr: float, seed: int = 1,
                 act: str = "sigmoid", dims=(784, 100, 10), data_seed: int = 0):
        self.devices = [torch.device(d) for d in devices]
        self.n = len(self.devices)
        if self.n < 1:
            raise ValueError("need at least one tower")
        specs = mlp_var_specs(*dims)
        assign_shards(specs, 1)
        self.layout = ShardLayout(specs, 1, mode="flat")
        self.params: List[torch.Tensor] = []
        self.grads: List[torch.Tensor] = []
        self.engines: List[MLPEngine] = []
        self.datasets: List[DeviceDataset] = []
        self.lr_t: List[torch.Tensor] = []
        for t, dev in enumerate(self.devices):
            p = self.layout.new_slab(dev)
            if t == 0:
                self.layout.initialize(p, seed)
            else:
                p.copy_(self.params[0])           # towers start as exact replicas
            g = self.layout.new_slab(dev)
            eng = MLPEngine(self.layout.mlp_views(p), self.layout.mlp_views(g), batch=batch, lr=lr, act=act)
            ds = DeviceDataset(images, labels, dev, batch, seed=data_seed + t)   # a different shuffle per tower
            eng.bind_data(ds.x_epoch, ds.y_epoch)
            self.params.append(p)
            self.grads.append(g)
            self.engines.append(eng)
            self.datasets.append(ds)
            self.lr_t.append(eng.lr)
        self.lr = float(lr)
        self.steps = 0
        distinct_gpus = all(d.type == "cuda" for d in self.devices) and \
            len({d.index for d in self.devices}) == self.n
        self.use_rccl = distinct_gpus and self.n > 1
        self.batches_per_epoch = self.engines[0].batches_per_epoch

    def describe(self) -> str:
        how = "torch.cuda.nccl.all_reduce (RCCL)" if self.use_rccl else "in-process sum"
        return f"in-graph: 1 client, {self.n} towers, gradient mean by {how}"

    def _reduce(self):
        if self.n == 1:
            return
        if self.use_rccl:
            import torch.cuda.nccl as nccl

            try:
                nccl.all_reduce(self.grads)
                return
            except (RuntimeError, AttributeError) as e:
                # (a build of the library without the single-process binding: the sum below gives the same mean)
                print(f"[towers] torch.cuda.nccl.all_reduce is not usable here ({type(e).__name__}: {e}); "
                      f"summing the towers' gradients in this process instead", flush=True)
                self.use_rccl = False
        total = self.grads[0].clone()
        for g in self.grads[1:]:
            total += g.to(total.device)
        for g in self.grads:
            g.copy_(total.to(g.device))

    def step(self):
        pos = self.steps % self.batches_per_epoch
        if pos == 0 and self.steps > 0:
            for ds in self.datasets:
                ds.shuffle_epoch()
        for t, eng in enumerate(self.engines):
            with torch.cuda.device(self.devices[t]) if self.devices[t].type == "cuda" else _null():
                eng.step_grad(pos, scale=1.0 / self.n)
        self._reduce()
        for t in range(self.n):
            dev = self.devices[t]
            gs = self.layout.view(self.params[t], "global_step").view(1)
            if dev.type == "cuda":
                with torch.cuda.device(dev):
                    hb.sgd_apply_(self.params[t], self.grads[t], self.lr_t[t], 1.0, False, gs)
            else:
                ref.sgd_apply_(self.params[t], self.grads[t], self.lr)
                gs.add_(1.0)
        for eng in self.engines:
            eng.advance(1)
        self.steps += 1

    def run_steps(self, n: int):
        for _ in range(n):
            self.step()

    # -- the trainer surface shared with parallel.replicas (used by the example loop) ----------
    @property
    def device(self):
        return self.devices[0]

    @property
    def engine(self):
        return self.engines[0]

    def last_loss(self) -> float:
        """Mean of the towers' last mini-batch costs."""
        return sum(e.last_loss() for e in self.engines) / self.n

    def evaluate(self, x, y):
        return self.engines[0].evaluate(x, y)

    def global_step_value(self) -> int:
        return int(round(float(self.layout.view(self.params[0], "global_step").item())))

    def replicas_identical(self) -> bool:
        ref0 = self.params[0].cpu()
        return all(torch.equal(ref0, p.cpu()) for p in self.params[1:])


class DenseInGraphTowers:
    """In-graph replication of a fully connected stack of any depth (``tfdist.py --hidden 512,256``): the host loop of
    :class:`InGraphTowers` around ``models.dense.DenseStackEngine`` towers.  Every tower holds a replica of the
    variables in one flat slab ([global_step][W_1][b_1]...[W_L][b_L], every tensor on a 16-byte boundary) and a
    gradient slab of the same layout; per step every tower computes its gradients scaled by 1 / N on its own device,
    the N slabs are all-reduced (RCCL's single-process entry between distinct GPUs, a plain sum otherwise) and every
    tower applies the same SGD launch, so the replicas stay bit-identical and ``global_step`` advances once."""

    def __init__(self, devices: Sequence, images, labels, batch: int, lr: float, hidden: Sequence[int],
                 act: str = "relu", seed: int = 1, data_seed: int = 0):
        from ..models.dense import DenseStackEngine

        self.devices = [torch.device(d) for d in devices]
        self.n = len(self.devices)
        if self.n < 1:
            raise ValueError("need at least one tower")
        dims = [int(images.shape[1])] + [int(h) for h in hidden] + [int(labels.shape[1])]
        L = len(dims) - 1
        pad4 = lambda k: (k + 3) // 4 * 4  # noqa: E731
        ranges, off = [], 4                      # the first 16 bytes hold the step counter
        for i in range(L):
            w = (off, off + dims[i] * dims[i + 1])
            off = pad4(w[1])
            b = (off, off + dims[i + 1])
            off = pad4(b[1])
            ranges.append((w, b))
        gen = torch.Generator().manual_seed(int(seed))
        first = torch.zeros(off, dtype=torch.float32)
        for i, (w, _) in enumerate(ranges):      # stddev 1 / sqrt(fan-in), biases zero (examples.mnist_mlp._build_deep_model)
            first[w[0]:w[1]] = (torch.randn(dims[i], dims[i + 1], generator=gen) * dims[i] ** -0.5).reshape(-1)
        self.params: List[torch.Tensor] = []
        self.grads: List[torch.Tensor] = []
        self.engines, self.datasets = [], []
        for t, dev in enumerate(self.devices):
            p, g = first.clone().to(dev), torch.zeros(off, dtype=torch.float32, device=dev)     # (own copy per tower)
            views = lambda slab: [(slab[w[0]:w[1]].view(dims[i], dims[i + 1]), slab[b[0]:b[1]])  # noqa: E731
                                  for i, (w, b) in enumerate(ranges)]
            pv = views(p)
            with torch.cuda.device(dev) if dev.type == "cuda" else _null():
                eng = DenseStackEngine([w for w, _ in pv], [b for _, b in pv], [act] * (L - 1), batch=int(batch), lr=lr,
                                       grads=views(g))
                ds = DeviceDataset(images, labels, dev, batch, seed=data_seed + t)   # a different shuffle per tower
            eng.bind_data(ds.x_epoch, ds.y_epoch)
            self.params.append(p)
            self.grads.append(g)
            self.engines.append(eng)
            self.datasets.append(ds)
        self.lr = float(lr)
        self.steps = 0
        distinct_gpus = all(d.type == "cuda" for d in self.devices) and len({d.index for d in self.devices}) == self.n
        self.use_rccl = distinct_gpus and self.n > 1
        self.batches_per_epoch = self.engines[0].batches_per_epoch

    def describe(self) -> str:
        how = "torch.cuda.nccl.all_reduce (RCCL)" if self.use_rccl else "in-process sum"
        return (f"in-graph: 1 client, {self.n} towers of a dense stack " + "-".join(str(d) for d in self.engines[0].dims)
                + f", {self.engines[0].launches_per_step} launches per tower and step, gradient mean by {how}")

    _reduce = InGraphTowers._reduce

    def step(self):
        pos = self.steps % self.batches_per_epoch
        if pos == 0 and self.steps > 0:
            for ds, eng in zip(self.datasets, self.engines):
                ds.shuffle_epoch()
                eng.bind_data(ds.x_epoch, ds.y_epoch)
        for t, eng in enumerate(self.engines):
            with torch.cuda.device(self.devices[t]) if self.devices[t].type == "cuda" else _null():
                eng.step_grad(pos, scale=1.0 / self.n)
        self._reduce()
        for t, dev in enumerate(self.devices):
            gs = self.params[t][0:1]
            if dev.type == "cuda":
                with torch.cuda.device(dev):
                    hb.sgd_apply_(self.params[t], self.grads[t], self.engines[t].lr, 1.0, False, gs)
            else:
                self.params[t].sub_(self.grads[t], alpha=self.lr)
                gs.add_(1.0)
        for eng in self.engines:
            eng.advance(1)
        self.steps += 1

    def run_steps(self, n: int):
        for _ in range(n):
            self.step()

    @property
    def device(self):
        return self.devices[0]

    @property
    def engine(self):
        return self.engines[0]

    def last_loss(self) -> float:
        return sum(e.last_loss() for e in self.engines) / self.n

    def evaluate(self, x, y):
        return self.engines[0].evaluate(x, y)

    def global_step_value(self) -> int:
        return int(round(float(self.params[0][0].item())))

    def replicas_identical(self) -> bool:
        ref0 = self.params[0].cpu()
        return all(torch.equal(ref0, p.cpu()) for p in self.params[1:])


class DenseInGraphEngine:
    """In-graph replication of a fully connected stack of any depth as an ENGINE (``tfdist.py --hidden 512,256`` on
    GPUs): every tower has its own device, HIP stream, replica of the variables in one flat slab, shuffled resident data
    set and hipGraph chunks of

        the stack's gradient launches (scaled by 1 / N, into one of two gradient slabs that alternate per step)
        ONE exchange kernel: waits for every tower's slab of this step, sums the N slabs in tower order straight out
        of the peers' memory (peer access between the GPUs of one process: plain pointers) and applies SGD

    so the client issues one graph launch per tower per chunk and synchronises only when it reads a cost; the replicas
    stay bit-identical and ``global_step`` advances once per step.  Replaces the host loop of
    :class:`DenseInGraphTowers` (kept for CPU towers).  Towers may share a GPU (tests, one-GPU rehearsal): their
    exchange kernels then have to be resident together (a bounded grid of 256 workgroups each)."""

    def __init__(self, devices: Sequence, images, labels, batch: int, lr: float, hidden: Sequence[int], act: str = "relu",
                 seed: int = 1, chunk: int = 50, data_seed: int = 0, timeout_ms: int = 20000, use_graph: bool = True):
        from ..models.dense import DenseStackEngine
        from .xgmi import XgmiExchange

        self.devices = [torch.device(d) for d in devices]
        if any(d.type != "cuda" for d in self.devices):
            raise RuntimeError("the exchange kernel is a GPU path (CPU towers: DenseInGraphTowers)")
        self.n = len(self.devices)
        self.chunk = max(1, int(chunk))
        self.use_graph = bool(use_graph)
        self._closed = False
        dims = [int(images.shape[1])] + [int(h) for h in hidden] + [int(labels.shape[1])]
        L = len(dims) - 1
        pad4 = lambda k: (k + 3) // 4 * 4  # noqa: E731
        ranges, off = [], 4                      # the first 16 bytes hold the step counter
        for i in range(L):
            w = (off, off + dims[i] * dims[i + 1])
            off = pad4(w[1])
            b = (off, off + dims[i + 1])
            off = pad4(b[1])
            ranges.append((w, b))
        gen = torch.Generator().manual_seed(int(seed))
        first = torch.zeros(off, dtype=torch.float32)
        for i, (w, _) in enumerate(ranges):      # stddev 1 / sqrt(fan-in), biases zero (examples.mnist_mlp._build_deep_model)
            first[w[0]:w[1]] = (torch.randn(dims[i], dims[i + 1], generator=gen) * dims[i] ** -0.5).reshape(-1)
        views = lambda slab: [(slab[w[0]:w[1]].view(dims[i], dims[i + 1]), slab[b[0]:b[1]])  # noqa: E731
                              for i, (w, b) in enumerate(ranges)]
        distinct = sorted({d.index for d in self.devices})
        for a in distinct:                      # the exchange kernel of every tower reads every other tower's slabs
            for b in distinct:
                hb.enable_peer_access(torch.device("cuda", a), torch.device("cuda", b))
        self.streams, self._spare_streams = [], []
        self.params, self.engines, self.datasets, self.xgs, self._gsets, self.steps = [], [], [], [], [], []
        for t, dev in enumerate(self.devices):
            with torch.cuda.device(dev):
                st = InGraphEngine._stream_for_tower(self, dev)
                with torch.cuda.stream(st):
                    p = first.clone().to(dev)
                    xg = XgmiExchange(t, self.n, off, dev, uncached_grads=False, timeout_ms=timeout_ms, connect=False)
                    for g in xg.grads:                  # whatever no variable covers (step counter, padding) adds zero
                        g.zero_()
                    gsets = [views(g) for g in xg.grads]
                    pv = views(p)
                    eng = DenseStackEngine([w for w, _ in pv], [b for _, b in pv], [act] * (L - 1), batch=int(batch), lr=lr,
                                           grads=gsets[0])
                    ds = DeviceDataset(images, labels, dev, batch, seed=data_seed + t)   # a different shuffle per tower
                    eng.bind_data(ds.x_epoch, ds.y_epoch)
                    eng.step_grad(0, 0.0)               # loads the kernels outside any capture (gradients of zero weight)
                    for g in xg.grads:
                        g.zero_()
                st.synchronize()
            self.streams.append(st)
            self.params.append(p)
            self.engines.append(eng)
            self.datasets.append(ds)
            self.xgs.append(xg)
            self._gsets.append(gsets)
        for xg in self.xgs:
            xg.connect_local(self.xgs)
        self.lr = float(lr)
        self.batches_per_epoch = self.engines[0].batches_per_epoch
        self._graphs = [dict() for _ in range(self.n)]
        self._graph_keys = set()
        # Two real exchanges among all towers with a learning rate of zero (slot 0, then slot 1: the slots keep
        # alternating into step 0): the exchange kernel is loaded, and has met its peers, before any capture.
        zero_lr = [torch.zeros(1, dtype=torch.float32, device=d) for d in self.devices]
        for slot in (0, 1):
            for t in range(self.n):
                with self._on(t):
                    self.xgs[t].apply(self.params[t], zero_lr[t], slot, None)
        self.synchronize()
        if not self.healthy():
            raise RuntimeError("the towers' exchange kernels did not meet (are they resident together?)")

    _on = None          # (set below: the helpers of InGraphEngine that only use .devices / .streams)

    def describe(self) -> str:
        where = "one GPU each" if len({d.index for d in self.devices}) == self.n else "sharing a GPU"
        return (f"in-graph: 1 client, {self.n} towers of a dense stack " + "-".join(str(d) for d in self.engines[0].dims)
                + f" ({where}), a stream and hipGraph chunks of {self.chunk} steps per tower, {self.engines[0].launches_per_step} gradient "
                f"launches + one exchange kernel per step (sum of the towers' slabs in tower order through peer access, SGD), "
                f"1 host call per tower per chunk")

    def host_calls_per_chunk(self) -> int:
        return 1 if self.use_graph else (self.engines[0].launches_per_step + 1) * self.chunk

    def _one_step(self, t: int, bi: int, slot: int):
        eng = self.engines[t]
        eng.bind_grads(self._gsets[t][slot])
        eng.step_grad(bi, 1.0 / self.n)
        self.xgs[t].apply(self.params[t], eng.lr, slot, self.params[t][0:1])

    def _schedule(self, cursor: int, n: int):
        bpe, out = self.batches_per_epoch, []
        while n > 0:
            k = min(n, self.chunk, bpe - cursor % bpe)
            out.append(k)
            cursor += k
            n -= k
        return out

    def _capture_all(self, pos: int, k: int, parity: int):
        InGraphEngine._check_streams(self)
        self.synchronize()
        for t in range(self.n):
            with torch.cuda.device(self.devices[t]):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=self.streams[t], capture_error_mode="thread_local"):
                    for i in range(k):
                        self._one_step(t, pos + i, (parity + i) & 1)
                with torch.cuda.stream(self.streams[t]):
                    hb.graph_upload(graph)
                self._graphs[t][(pos, k, parity)] = graph
        self._graph_keys.add((pos, k, parity))

    def prepare(self, steps_hint=()):
        if not self.use_graph:
            return
        cursor = self.engines[0]._host_cursor
        for n in steps_hint:
            for k in self._schedule(cursor, n):
                key = (cursor % self.batches_per_epoch, k, cursor & 1)
                if key not in self._graph_keys:
                    self._capture_all(*key)
                cursor += k

    def run_steps(self, n: int):
        bpe = self.batches_per_epoch
        for k in self._schedule(self.engines[0]._host_cursor, n):
            cur = self.engines[0]._host_cursor
            pos, parity = cur % bpe, cur & 1
            if pos == 0 and cur > 0:
                for t in range(self.n):
                    with self._on(t):
                        self.datasets[t].shuffle_epoch()          # in place: captured graphs stay valid
            if self.use_graph and (pos, k, parity) not in self._graph_keys:
                self._capture_all(pos, k, parity)
            for t in range(self.n):
                with self._on(t):
                    if self.use_graph:
                        self._graphs[t][(pos, k, parity)].replay()
                    else:
                        for i in range(k):
                            self._one_step(t, pos + i, (parity + i) & 1)
                    self.datasets[t].stage_ahead()
            for eng in self.engines:
                eng.advance(k)

    # -- the trainer surface shared with the other tower classes (used by the example loop) ----------
    @property
    def device(self):
        return self.devices[0]

    @property
    def engine(self):
        return self.engines[0]

    def last_loss(self) -> float:
        self.synchronize()
        return sum(e.last_loss() for e in self.engines) / self.n

    def evaluate(self, x, y):
        self.synchronize()
        with torch.cuda.device(self.devices[0]):
            return self.engines[0].evaluate(x, y)

    def global_step_value(self) -> int:
        self.synchronize()
        return int(round(float(self.params[0][0].item())))

    def replicas_identical(self) -> bool:
        self.synchronize()
        ref0 = self.params[0].cpu()
        return all(torch.equal(ref0, p.cpu()) for p in self.params[1:])

    def healthy(self) -> bool:
        self.synchronize()
        return not any(xg.failed() for xg in self.xgs)

    def diagnostics(self) -> dict:
        self.synchronize()
        return {"streams": [hex(s.cuda_stream) for s in self.streams], "exchanges": [xg.exchanges_done() for xg in self.xgs],
                "failed": [xg.failed() for xg in self.xgs]}

    def close(self):
        if self._closed:
            return
        self._closed = True
        try:
            self.synchronize()
        except Exception:  # noqa: BLE001 — tearing down after a device error
            pass
        for g in self._graphs:
            g.clear()
        for t, xg in enumerate(self.xgs):
            with torch.cuda.device(self.devices[t]):
                xg.close()


class InGraphEngine:
    """One client, N towers, O(1) host calls per tower per chunk (see the module docstring).

    ``exchange="inbox"``: gradient mean inside the backward launches (``compact=True``: 4-byte words);
    ``exchange="rccl"``: the :class:`InGraphTowers` host loop (eager; the baseline).
    Several towers may share one GPU (tests, one-GPU rehearsal): their kernels then have to be resident together,
    which holds for up to 3 towers on an MI355X.
    """

    def __init__(self, devices: Sequence, images, labels, batch: int, lr: float, seed: int = 1, chunk: int = 50,
                 exchange: str = "inbox", compact: bool = False, data_seed: int = 0, timeout_ms: int = 20000,
                 use_graph: bool = True):
        if exchange not in ("inbox", "rccl"):
            raise ValueError(exchange)
        self.devices = [torch.device(d) for d in devices]
        self.n = len(self.devices)
        self.exchange = exchange
        self.chunk = max(1, int(chunk))
        self.use_graph = bool(use_graph)
        self._closed = False
        if exchange == "rccl":
            self._loop = InGraphTowers(self.devices, images, labels, batch, lr, seed=seed, data_seed=data_seed)
            self.towers = []
            return
        if any(d.type != "cuda" for d in self.devices):
            raise RuntimeError("the in-launch exchange is a GPU path")
        from .replicas import XgmiFusedTrainer

        distinct = sorted({d.index for d in self.devices})
        for a in distinct:                      # kernels of every tower store into every other tower's inbox
            for b in distinct:
                hb.enable_peer_access(torch.device("cuda", a), torch.device("cuda", b))
        self._loop = None
        self.streams = []
        self.towers = []
        self._spare_streams = []
        for t, dev in enumerate(self.devices):
            with torch.cuda.device(dev):
                st = self._stream_for_tower(dev)
                with torch.cuda.stream(st):
                    ds = DeviceDataset(images, labels, dev, batch, seed=data_seed + t)   # a different shuffle per tower
                    tr = XgmiFusedTrainer(device=dev, dataset=ds, batch=batch, lr=lr, seed=seed, rank=t, world=self.n,
                                          chunk=self.chunk, use_graph=self.use_graph, compact=compact,
                                          timeout_ms=timeout_ms, local_peers=True)
                st.synchronize()
            self.streams.append(st)
            self.towers.append(tr)
        for tr in self.towers:
            tr.connect_local(self.towers)
        self.batches_per_epoch = self.towers[0].engine.batches_per_epoch
        self._graph_keys = set()

    # -- helpers --------------------------------------------------------------------------------------
    def _stream_for_tower(self, dev):
        """A stream whose kernels can run while the kernels of every earlier tower on the same GPU wait for them.

        The streams of a process share a few hardware queues and a queue runs its kernels in order; which queue a new
        stream lands on depends on what the process did before.  Towers on one GPU wait for each other inside their
        backward launches, so two of them on one queue would wait until the exchange's time limit."""
        mates = [s for s, d in zip(self.streams, self.devices) if d == dev]
        for _ in range(16):
            st = torch.cuda.Stream(device=dev)
            if all(hb.streams_independent(m, st) for m in mates):
                return st
            self._spare_streams.append(st)           # keep it, so that the pool hands out a different one next
        raise RuntimeError(f"no stream on {dev} found that runs beside the {len(mates)} tower stream(s) already there: "
                           "the hardware queues of this process are taken (GPU_MAX_HW_QUEUES)")

    def _on(self, t: int):
        import contextlib

        stack = contextlib.ExitStack()
        stack.enter_context(torch.cuda.device(self.devices[t]))
        stack.enter_context(torch.cuda.stream(self.streams[t]))
        return stack

    def synchronize(self):
        for d in {d for d in self.devices if d.type == "cuda"}:
            torch.cuda.synchronize(d)

    def host_calls_per_chunk(self) -> int:
        """Host calls per tower per chunk of steps in steady state (inside an epoch)."""
        return 1 if (self._loop is None and self.use_graph) else 3 * self.chunk

    def describe(self) -> str:
        if self._loop is not None:
            return self._loop.describe() + ", host loop (eager)"
        where = "one GPU each" if len({d.index for d in self.devices}) == self.n else "sharing a GPU"
        return (f"in-graph: 1 client, {self.n} towers ({where}), a stream and hipGraph chunks of {self.chunk} steps per "
                f"tower, gradient mean inside the towers' backward launches (words stored into the other towers' "
                f"inboxes through peer access), 2 launches/step, 1 host call per tower per chunk")

    def _check_streams(self):
        """Towers that share a GPU wait for each other inside their kernels: their streams must (still) sit on
        different hardware queues.  Checked at creation and again before every capture."""
        for a in range(self.n):
            for b in range(a + 1, self.n):
                if self.devices[a] == self.devices[b]:
                    with torch.cuda.device(self.devices[a]):
                        if not hb.streams_independent(self.streams[a], self.streams[b]):
                            raise RuntimeError(f"towers {a} and {b} on {self.devices[a]}: streams "
                                               f"{self.streams[a].cuda_stream:#x} / {self.streams[b].cuda_stream:#x} "
                                               f"share a hardware queue")

    def diagnostics(self) -> dict:
        """Stream handles, inbox addresses and, after a failed exchange, every tower's record of it."""
        if self._loop is not None:
            return {}
        self.synchronize()
        return {"streams": [hex(s.cuda_stream) for s in self.streams],
                "spare_streams": [hex(s.cuda_stream) for s in self._spare_streams],
                "inboxes": [hex(tr.xg._ptr) for tr in self.towers],
                "states": [hex(tr.xg.state.data_ptr()) for tr in self.towers],
                "exchanges": [tr.xg.exchanges_done() for tr in self.towers],
                "failed": [tr.xg.diagnose() for tr in self.towers],
                "gather_errors": [hb.gather_errors(d) for d in sorted(set(self.devices), key=str)]}

    # -- capture: the towers' warm-up exchanges have to be in flight TOGETHER --------------------------
    def _capture_all(self, pos: int, k: int, parity: int):
        tw = self.towers
        self._check_streams()
        snaps = []
        for t, tr in enumerate(tw):
            with self._on(t):
                snaps.append(tr._snapshot())
                tr.engine.reset_step_workspace(parity)
        import os
        debug = os.environ.get("DTF_TOWERS_DEBUG") == "1"
        for t, tr in enumerate(tw):
            with self._on(t):
                tr._chunk_begin(k)                # (the warm-up steps launch the kernels of the chunk captured below)
        for w in range(2):                        # real exchanges among all towers, rolled back below
            before = [tr.xg.exchanges_done() for tr in tw] if debug else None
            for t, tr in enumerate(tw):
                with self._on(t):
                    tr._one_step(pos)
            if debug:
                self.synchronize()
                after = [tr.xg.exchanges_done() for tr in tw]
                bad = [tr.xg.failed() for tr in tw]
                if any(bad) or any(a != b + 1 for a, b in zip(after, before)):
                    print(f"[towers debug] capture ({pos},{k},{parity}) warm-up {w}: exchange numbers {before} -> {after}, "
                          f"failed {bad}, raw state {[tr.xg.state.cpu().tolist() for tr in tw]}", flush=True)
        for t, tr in enumerate(tw):
            with self._on(t):
                tr._chunk_end()
        self.synchronize()
        for t, tr in enumerate(tw):
            with self._on(t):
                tr._restore(snaps[t])
                tr.engine.reset_step_workspace(parity)
        self.synchronize()
        for t, tr in enumerate(tw):
            with torch.cuda.device(self.devices[t]):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=self.streams[t], capture_error_mode="thread_local"):
                    tr._steps(pos, k)
                tr.engine.parity = parity
                with torch.cuda.stream(self.streams[t]):
                    hb.graph_upload(graph)          # not during the first replay, which somebody may be timing
                tr._graphs[(pos, k, parity, tr._bound_set)] = graph
        self._graph_keys.add((pos, k, parity))

    def prepare(self, steps_hint=()):
        """Capture every chunk graph the hinted ``run_steps`` calls will need."""
        if self._loop is not None or not self.use_graph:
            return
        lead = self.towers[0]
        cursor, parity = lead.engine._host_cursor, lead.engine.parity
        live = parity
        keys = []
        for n in steps_hint:
            for k in lead._schedule(cursor, n):
                key = (cursor % self.batches_per_epoch, k, parity)
                if key not in self._graph_keys and key not in keys:
                    keys.append(key)
                cursor += k
                parity ^= k & 1
        for pos, k, par in keys:
            self._capture_all(pos, k, par)
        for t, tr in enumerate(self.towers):
            with self._on(t):
                tr.engine.reset_step_workspace(live)
        self.synchronize()

    def run_steps(self, n: int):
        if self._loop is not None:
            return self._loop.run_steps(n)
        lead = self.towers[0]
        bpe = self.batches_per_epoch
        for k in lead._schedule(lead.engine._host_cursor, n):
            cur = lead.engine._host_cursor
            if self.use_graph and (cur % bpe, k, lead.engine.parity) not in self._graph_keys:
                self._capture_all(cur % bpe, k, lead.engine.parity)
            for t, tr in enumerate(self.towers):
                with self._on(t):
                    tr.run_steps(k)               # one graph launch (plus the reshuffle at an epoch boundary)

    # -- the trainer surface --------------------------------------------------------------------------
    @property
    def device(self):
        return self.devices[0]

    @property
    def engine(self):
        """Tower 0's engine (cursor, per-step history: read after ``last_loss()``, which waits for the towers)."""
        return self._loop.engine if self._loop is not None else self.towers[0].engine

    def last_loss(self) -> float:
        if self._loop is not None:
            return self._loop.last_loss()
        self.synchronize()
        return sum(tr.last_loss() for tr in self.towers) / self.n

    def evaluate(self, x, y):
        if self._loop is not None:
            return self._loop.evaluate(x, y)
        self.synchronize()
        with torch.cuda.device(self.devices[0]):
            return self.towers[0].evaluate(x, y)

    def global_step_value(self) -> int:
        if self._loop is not None:
            return self._loop.global_step_value()
        self.synchronize()
        return self.towers[0].global_step_value()

    def replicas_identical(self) -> bool:
        if self._loop is not None:
            return self._loop.replicas_identical()
        self.synchronize()
        ref0 = self.towers[0].params.cpu()
        return all(torch.equal(ref0, tr.params.cpu()) for tr in self.towers[1:])

    def healthy(self) -> bool:
        if self._loop is not None:
            return True
        self.synchronize()
        return all(tr.healthy() for tr in self.towers)

    def close(self):
        if self._closed:
            return
        self._closed = True
        try:
            self.synchronize()
        except Exception:  # noqa: BLE001 — tearing down after a device error
            pass
        for t, tr in enumerate(self.towers):
            with torch.cuda.device(self.devices[t]):
                tr.close()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# helpers of InGraphEngine that only use .devices / .streams serve the dense engine as well
DenseInGraphEngine._on = InGraphEngine._on
DenseInGraphEngine.synchronize = InGraphEngine.synchronize
