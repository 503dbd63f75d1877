"""``Session``: executes fetches against a variable store.

``Session(config=...)`` is the single-process form (reference tfsingle.py:67-68);
``Session(server.target)`` — normally obtained from
``Supervisor.prepare_or_wait_for_session`` — binds to the task's ``Server`` and
keeps the variables on the parameter-server shards (reference
tfdist_between.py:83, 93-95).

``run(fetches, feed_dict)`` per call:

1. **pull**: a distributed session copies every shard into the worker's local
   slab (the Recv of W1, b1, W2, b2 and global_step of the reference's step);
2. a **train op** among the fetches computes loss and gradients on the worker
   device — through the fused HIP pipeline when the loss is the reference's
   MLP pattern (:func:`match_mlp`), otherwise op by op with PyTorch autograd —
   then **pushes**: in place (local), server-side SGD per shard (async), or
   accumulate + wait for a token (SyncReplicas);
3. remaining fetches are evaluated with the values of that same forward pass,
   i.e. *before* the update, like TensorFlow's single ``sess.run``.

The per-step host round trip of this API (feed, fetch of the cost) is what the
reference pays too; the device-resident fast tier (``parallel.replicas``,
``train.loop``) removes it.
"""
from __future__ import annotations

import contextlib
import os
import time
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import graph as G
from .cluster import ConfigProto, lookup_server
from .utils.trace import Collector, span
from .models.mlp import MLPEngine
from .ops import gemm as gemm_op
from .optim import Optimizer, SyncReplicasOptimizer
from .parallel.store import LocalVariableStore, RemoteVariableStore
from .placement import DeviceSpec
from .summary import Summary
from .variables import ShardLayout

_TORCH_DTYPES = {"float32": torch.float32, "float64": torch.float64, "int32": torch.int32, "int64": torch.int64,
                 "bool": torch.bool}

_session_stack: List["Session"] = []


def get_default_session() -> Optional["Session"]:
    return _session_stack[-1] if _session_stack else None


# ---------------------------------------------------------------------------
# pattern matching: the reference's model, loss and accuracy
# ---------------------------------------------------------------------------

def _is(node, op):
    return isinstance(node, G.Tensor) and node.op == op


def _linear(node):
    """add(matmul(x, W), b) -> (x, W, b)"""
    if not _is(node, "add"):
        return None
    for mm, b in (node.inputs, node.inputs[::-1]):
        if _is(mm, "matmul") and isinstance(b, G.Variable) and not mm.attrs["transpose_a"] \
                and not mm.attrs["transpose_b"] and isinstance(mm.inputs[1], G.Variable):
            return mm.inputs[0], mm.inputs[1], b
    return None


def match_mlp(loss) -> Optional[Dict[str, Any]]:
    """Recognise  mean(-sum(y_ * log(softmax(act(x W1 + b1) W2 + b2)), 1))  (tfdist_between.py:55-62)
    or mean(softmax_cross_entropy_with_logits(labels=y_, logits=...))."""
    head = _match_head(loss)
    if head is None:
        return None
    y_, probs, logits = head
    l2 = _linear(logits)
    if l2 is None:
        return None
    a2, W2, b2 = l2
    if a2.op not in ("sigmoid", "relu"):
        return None
    l1 = _linear(a2.inputs[0])
    if l1 is None:
        return None
    x, W1, b1 = l1
    if not (_is(x, "placeholder") and _is(y_, "placeholder")):
        return None
    if len(W1.shape) != 2 or len(W2.shape) != 2 or W2.shape[1] > 16 or len(b1.shape) != 1 or len(b2.shape) != 1:
        return None
    return {"x": x, "y_": y_, "W1": W1, "b1": b1, "W2": W2, "b2": b2, "act": a2.op, "logits": logits,
            "probs": probs, "loss": loss, "hidden": a2}


def _match_head(loss):
    """(y_, probs or None, logits) of  mean(-sum(y_ * log(softmax(logits)), 1))  or
    mean(softmax_cross_entropy_with_logits(labels=y_, logits=logits)), else None."""
    if not _is(loss, "reduce_mean"):
        return None
    inner = loss.inputs[0]
    if _is(inner, "softmax_xent"):
        y_, logits = inner.inputs
        return y_, None, logits
    if _is(inner, "neg") and _is(inner.inputs[0], "reduce_sum") and inner.inputs[0].attrs["axis"] in ((1,), (-1,)):
        prod = inner.inputs[0].inputs[0]
        if _is(prod, "mul"):
            for a, b in (prod.inputs, prod.inputs[::-1]):
                if _is(b, "log") and _is(b.inputs[0], "softmax"):
                    return a, b.inputs[0], b.inputs[0].inputs[0]
    return None


def _linear_or_matmul(node):
    """add(matmul(h, W), b) -> (h, W, b);  matmul(h, W) -> (h, W, None)"""
    lin = _linear(node)
    if lin is not None:
        return lin
    if _is(node, "matmul") and not node.attrs["transpose_a"] and not node.attrs["transpose_b"] \
            and isinstance(node.inputs[1], G.Variable):
        return node.inputs[0], node.inputs[1], None
    return None


def match_dense_stack(loss) -> Optional[Dict[str, Any]]:
    """Recognise the reference's graph with ANY number of fully connected layers: a softmax cross-entropy head
    (either spelling, see :func:`_match_head`) over  logits = h_L W_L (+ b_L),  h_i = act_i(h_{i-1} W_i (+ b_i)),
    h_0 = a placeholder, act_i in sigmoid / relu / tanh  (tfdist_between.py:55-62 is the two-layer case).  Served by
    ``models.dense.DenseStackEngine``."""
    head = _match_head(loss)
    if head is None:
        return None
    y_, probs, logits = head
    layers, acts, node = [], [], logits
    while True:
        lin = _linear_or_matmul(node)
        if lin is None:
            return None
        h, W, b = lin
        if len(W.shape) != 2 or (b is not None and len(b.shape) != 1):
            return None
        layers.append((W, b))
        if _is(h, "placeholder"):
            x = h
            break
        if not isinstance(h, G.Tensor) or h.op not in ("sigmoid", "relu", "tanh"):
            return None
        acts.append(h.op)
        node = h.inputs[0]
    if not _is(y_, "placeholder"):
        return None
    layers.reverse()
    acts.reverse()
    variables = [v for W, b in layers for v in ((W, b) if b is not None else (W,))]
    if len(set(variables)) != len(variables):
        return None         # shared weights: the in-place update of one layer would change another
    return {"x": x, "y_": y_, "layers": layers, "acts": acts, "logits": logits, "probs": probs, "loss": loss,
            "variables": variables}


def match_accuracy(node, m) -> bool:
    """mean(cast(equal(argmax(y, 1), argmax(y_, 1)), float32))  (tfdist_between.py:68-70)"""
    if not _is(node, "reduce_mean") or not _is(node.inputs[0], "cast"):
        return False
    eq = node.inputs[0].inputs[0]
    if not _is(eq, "equal"):
        return False
    a, b = eq.inputs
    if not (_is(a, "argmax") and _is(b, "argmax") and a.attrs["axis"] in (1, -1) and b.attrs["axis"] in (1, -1)):
        return False
    srcs = {a.inputs[0], b.inputs[0]}
    return m["y_"] in srcs and (m["probs"] in srcs or m["logits"] in srcs)


# ---------------------------------------------------------------------------
class _PendingScalar:
    """A scalar on its way into the session's pinned host words (see ``Session._host_words``)."""
    __slots__ = ("words", "event", "index")

    def __init__(self, words, event, index):
        self.words, self.event, self.index = words, event, index

    def get(self) -> float:
        self.event.synchronize()          # returns at once when a later wait on the stream has already passed it
        return float(self.words[self.index])


class Session:
    def __init__(self, target: str = "", graph: Optional[G.Graph] = None, config: Optional[ConfigProto] = None):
        self.graph = graph or G.get_default_graph()
        self.server = lookup_server(target) if target else None
        if target and self.server is None:
            raise ValueError(f"unknown session target {target!r}: create the task's Server first")
        self.config = config or (self.server.config if self.server else ConfigProto())
        self.device = self._resolve_device()
        self.bind_thread()
        specs = [v.spec for v in self.graph.variables]
        n_shards = self.server.num_ps if (self.server and self.server.num_ps > 0) else 1
        if n_shards == 1:
            for s in specs:
                s.shard = 0
        self.layout = ShardLayout(specs, n_shards, mode="by_variable")
        if self.server is not None and self.server.num_ps > 0:
            self.store = self._make_remote_store()
            self.theta = self.layout.new_slab(self.device)
        else:
            self.store = LocalVariableStore(self.layout, self.device)
            self.theta = self.store.bind_local()
        self.grad = self.layout.new_slab(self.device)
        self._engines: Dict[Any, MLPEngine] = {}
        self._plans: Dict[int, Any] = {}          # train op -> what runs it (fused model, dense stack, accuracy nodes)
        self._reads_vars: Dict[Any, bool] = {}    # node -> whether evaluating it reads variables
        self._replica_owner = None
        self._closed = False
        self.fused_steps = 0
        self._local_step_ready = False
        self.generic_steps = 0
        self.pushed_steps = 0       # fused steps of an asynchronous worker whose backward launch pushed the update itself
        self.dense_steps = 0        # steps of fully connected stacks other than the reference's (models/dense.py)
        # host mirror of global_step for a local session (None: unknown, re-read from the device): saves one
        # blocking device-to-host copy per sess.run that fetches it next to a train op
        self._gs_mirror: Optional[int] = None
        # Pinned host words that the per-batch step's results land in ([0:2] loss and accuracy of the fused step, [2]
        # a scalar variable fetched next to a train op): asynchronous copies and ONE event wait per sess.run instead
        # of a blocking ``.cpu()`` per fetched value (each ~20 us: a staged copy into pageable memory).
        self._host_words = self._host_evt = self._host_evt_var = None
        if self.device.type == "cuda":
            self._host_words = torch.zeros(4, dtype=torch.float32).pin_memory()
            self._host_evt, self._host_evt_var = torch.cuda.Event(), torch.cuda.Event()
        # One thread at a time inside the session: a Supervisor's checkpoint thread pulls into the same slab and
        # staging buffers a training step is using (re-entrant: run() -> _pull(), Saver -> var_tensor()).
        import threading

        self.run_lock = threading.RLock()
        if self.config.log_device_placement:
            self._log_placement()

    def _make_remote_store(self):
        if self.config.transport == "ipc":
            from .parallel.ps_ipc import IpcVariableStore

            return IpcVariableStore(self.layout, self.server, self.device)
        return RemoteVariableStore(self.layout, self.server, self.device)

    # -- device -----------------------------------------------------------------
    def _resolve_device(self) -> torch.device:
        if self.server is not None:
            return self.server.device
        want_gpu, idx = None, 0
        for n in self.graph.nodes:
            d = DeviceSpec.from_string(n.device)
            if d.device_type:
                want_gpu, idx = d.device_type == "gpu", d.device_index or 0
                break
        if want_gpu is False:
            return torch.device("cpu")
        if torch.cuda.is_available():
            return torch.device("cuda", idx % torch.cuda.device_count())
        if want_gpu and not self.config.allow_soft_placement:
            raise RuntimeError("graph requests /gpu:%d but no GPU is visible and allow_soft_placement=False" % idx)
        return torch.device("cpu")

    def _log_placement(self):
        print("Device mapping:", flush=True)
        for v in self.graph.variables:
            where = v.device or f"/job:localhost/{self.device}"
            print(f"{v.op_name}: (VariableV2): {where}", flush=True)
        worker = None
        for n in self.graph.nodes:
            if n.op not in ("variable", "const") and n.device:
                worker = n.device
                break
        print(f"<compute ops>: {worker or '/job:localhost'} -> {self.device}", flush=True)
        print(self.layout.describe(), flush=True)

    # -- context ------------------------------------------------------------------
    def __enter__(self):
        _session_stack.append(self)
        return self

    def __exit__(self, *exc):
        if _session_stack and _session_stack[-1] is self:
            _session_stack.pop()
        self.close()
        return False

    def as_default(self):
        return self

    def bind_thread(self) -> None:
        """Make the session's GPU the calling thread's current device.  The native launches go to the current device's
        current stream and set per-device kernel attributes there, and the current device is a property of the THREAD
        (a new thread starts on device 0): the constructor does this for its own thread, a service thread that works
        on the session's store (the Supervisor's checkpoint loop, the chief's aggregation loop) calls it first.  With
        worker k on GPU k (tfdist_between.py:34) device 0 would be the wrong one for every worker but the first."""
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)

    def _on_device(self):
        """Context for entry points any thread may call: the session's GPU is the current device inside."""
        if self.device.type != "cuda" or torch.cuda.current_device() == self.device.index:
            return contextlib.nullcontext()
        return torch.cuda.device(self.device)

    def close(self):
        if not self._closed:
            with self._on_device():
                self.store.close()
            self._closed = True

    # -- variables ------------------------------------------------------------------
    def var_tensor(self, v: G.Variable) -> torch.Tensor:
        self._gs_mirror = None      # the caller may write through the view (Saver.restore, assign)
        return self.layout.view(self.theta, v.op_name)

    def _var_view(self, v: G.Variable) -> torch.Tensor:
        """Read-only use inside the session: does not invalidate the host mirror."""
        return self.layout.view(self.theta, v.op_name)

    def _pull(self):
        # A device-resident synchronous trainer keeps THE replica of this worker in self.theta and only publishes it to
        # the parameter-server shards at the end (``own_replica``): until then the shards hold older values, and a pull
        # (a checkpoint of the Supervisor's thread, an evaluation) would silently set this worker back.
        if self._replica_owner is not None:
            return
        with self.run_lock, self._on_device(), span("pull"):
            self.store.pull(self.theta)

    def own_replica(self, owner) -> None:
        """``owner`` (a trainer) declares self.theta authoritative for this worker; ``None`` hands it back."""
        self._replica_owner = owner

    def _run_init(self, variables: Sequence[G.Variable]):
        self._gs_mirror = None
        seed = self.graph.seed if self.graph.seed is not None else int(time.time())
        all_vars = len(variables) == len(self.graph.variables)
        self.store.initialize(seed, None if all_vars else [v.op_name for v in variables])
        if self.store.is_remote:
            self._pull()

    # -- evaluation -------------------------------------------------------------------
    def run(self, fetches, feed_dict: Optional[Dict[G.Tensor, Any]] = None, options=None, run_metadata=None):
        """``options=RunOptions(trace_level=...)`` records this call's phases into ``run_metadata.step_stats``."""
        if options is not None and getattr(options, "wants_trace", False):
            col = run_metadata.step_stats.collector if run_metadata is not None else Collector()
            col.sync = bool(options.wants_device_time)
            with self.run_lock, self._on_device(), col, span("session.run"):
                return self._run(fetches, feed_dict)
        with self.run_lock, self._on_device():
            return self._run(fetches, feed_dict)

    def _run(self, fetches, feed_dict):
        if self._closed:
            raise RuntimeError("session is closed")
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        feed = {}
        for k, v in (feed_dict or {}).items():
            feed[k] = v
        cache: Dict[G.Tensor, Any] = {}
        needs_vars = any(self._reads_variables(f) for f in flist if isinstance(f, G.Tensor))
        shard_step = self._shard_served_call(flist, feed) if (needs_vars and self.store.is_remote) else None
        if needs_vars and self.store.is_remote and shard_step is None:
            self._pull()
        # train ops first: they define the forward pass the other fetches see.  Variables fetched
        # in the same call (global_step) are read BEFORE the update — the reference prints
        # step+1 for that reason (tfdist_between.py:102).
        trains = [f for f in flist if _is(f, "train")]
        if trains:
            gs_var = trains[0].attrs["global_step"] if len(trains) == 1 else None
            mirror_ok = gs_var is not None and not self.store.is_remote
            pending_used = False
            for f in flist:
                if _is(f, "variable"):
                    if mirror_ok and f is gs_var:
                        if self._gs_mirror is None:
                            self._gs_mirror = int(round(float(self._var_view(f).item())))
                        cache[f] = torch.tensor(float(self._gs_mirror))         # host value, pre-update
                    elif shard_step == "pushed" and f is gs_var:
                        # a call served by the fused asynchronous step has no pull in front, and the step itself hands
                        # back the counter as it was before its push (the reference's fetch next to the train op reads
                        # it unordered, tfdist_between.py:94,102): _run_train puts it into the cache
                        pass
                    else:
                        # (a synchronous step that reads the variables from the shards has no pull in front either:
                        # the counter is read where it lives)
                        view = self.store.step_counter_view() if (shard_step == "deposit" and f is gs_var) \
                            else self._var_view(f)
                        if self._host_words is not None and view.numel() == 1 and not pending_used:
                            with self._on_device():
                                self._host_words[2:3].copy_(view.reshape(1), non_blocking=True)
                                pending = _PendingScalar(self._host_words, self._host_evt_var, 2)
                                # (with the device named: record() on its own asks torch.cuda.current_stream(), whose
                                # device lookup is a driver call of 11 us)
                                pending.event.record(torch.cuda.current_stream(self.device))
                            cache[f] = pending
                            pending_used = True
                        else:
                            cache[f] = view.clone()
            if not mirror_ok:
                self._gs_mirror = None
        for f in trains:
            self._run_train(f, feed, cache)
            if self._gs_mirror is not None:
                self._gs_mirror += 1
        out = []
        for f in flist:
            out.append(self._fetch(f, feed, cache))
        return out[0] if single else out

    def _reads_variables(self, node: G.Tensor, _seen=None) -> bool:
        if node.op in ("variable", "train"):
            return True
        if node.op in ("init_all", "init_vars", "no_op", "sync_init_tokens"):
            return False
        key = node
        if key in self._reads_vars:
            return self._reads_vars[key]
        r = any(self._reads_variables(i) for i in node.inputs)
        if node.op in ("merge_summary", "scalar_summary", "gradient"):
            r = True
        self._reads_vars[key] = r
        return r

    def _fetch(self, f, feed, cache):
        if f is None:
            return None
        if not isinstance(f, G.Tensor):
            raise TypeError(f"cannot fetch {f!r}")
        if f.op in ("train", "init_all", "init_vars", "no_op", "group", "sync_init_tokens"):
            if f.op in ("init_all", "init_vars"):
                self._run_init(f.attrs["variables"])
            elif f.op == "group":
                for i in f.inputs:
                    self._fetch(i, feed, cache)
            elif f.op == "sync_init_tokens":
                self._init_tokens(f)
            return None
        # whole-batch evaluation of the matched loss / accuracy on the fused kernels
        fused = self._try_fused_eval(f, feed, cache)
        if fused is not None:
            return fused
        v = self._eval(f, feed, cache)
        return self._to_host(v)

    @staticmethod
    def _to_host(v):
        if isinstance(v, _PendingScalar):
            return v.get()
        if isinstance(v, torch.Tensor):
            a = v.detach().cpu().numpy()
            return a.item() if a.ndim == 0 else a
        return v

    def _feed_tensor(self, node, feed):
        if node not in feed:
            raise ValueError(f"placeholder {node.name} must be fed")
        val = feed[node]
        if isinstance(val, torch.Tensor):
            return val.to(self.device, _TORCH_DTYPES[node.dtype])
        return torch.as_tensor(np.asarray(val), dtype=_TORCH_DTYPES[node.dtype]).to(self.device)

    def _eval(self, node: G.Tensor, feed, cache, var_override=None):
        if node in cache:
            r = cache[node]
            return torch.tensor(r.get()) if isinstance(r, _PendingScalar) else r
        op = node.op
        ins = node.inputs
        ev = lambda n: self._eval(n, feed, cache, var_override)  # noqa: E731
        if op == "variable":
            r = var_override[node] if (var_override and node in var_override) else self._var_view(node)
        elif op == "placeholder":
            r = self._feed_tensor(node, feed)
        elif op == "const":
            r = torch.as_tensor(node.attrs["value"]).to(self.device)
        elif op == "random_normal":
            g = torch.Generator().manual_seed(int(node.attrs["seed"] or self.graph.seed or 0) + node.id)
            r = (torch.randn(node.attrs["shape"], generator=g) * node.attrs["stddev"] + node.attrs["mean"]).to(self.device)
        elif op == "zeros":
            r = torch.zeros(node.attrs["shape"], device=self.device)
        elif op == "ones":
            r = torch.ones(node.attrs["shape"], device=self.device)
        elif op == "matmul":
            # the hand-written fp32 MFMA GEMM (forward and, through autograd, both gradient products) on a GPU;
            # torch.matmul for what it does not cover
            r = gemm_op.matmul(ev(ins[0]), ev(ins[1]), node.attrs["transpose_a"], node.attrs["transpose_b"])
        elif op == "add":
            r = ev(ins[0]) + ev(ins[1])
        elif op == "sub":
            r = ev(ins[0]) - ev(ins[1])
        elif op == "mul":
            r = ev(ins[0]) * ev(ins[1])
        elif op == "div":
            r = ev(ins[0]) / ev(ins[1])
        elif op == "maximum":
            r = torch.maximum(ev(ins[0]), ev(ins[1]))
        elif op == "minimum":
            r = torch.minimum(ev(ins[0]), ev(ins[1]))
        elif op == "equal":
            r = ev(ins[0]) == ev(ins[1])
        elif op == "greater":
            r = ev(ins[0]) > ev(ins[1])
        elif op == "less":
            r = ev(ins[0]) < ev(ins[1])
        elif op == "neg":
            r = -ev(ins[0])
        elif op == "log":
            r = torch.log(ev(ins[0]))
        elif op == "exp":
            r = torch.exp(ev(ins[0]))
        elif op == "sqrt":
            r = torch.sqrt(ev(ins[0]))
        elif op == "square":
            r = ev(ins[0]) ** 2
        elif op == "abs":
            r = ev(ins[0]).abs()
        elif op == "sigmoid":
            r = torch.sigmoid(ev(ins[0]))
        elif op == "tanh":
            r = torch.tanh(ev(ins[0]))
        elif op == "relu":
            r = torch.relu(ev(ins[0]))
        elif op in ("identity",):
            r = ev(ins[0])
        elif op == "stop_gradient":
            r = ev(ins[0]).detach()
        elif op == "softmax":
            r = torch.softmax(ev(ins[0]), dim=node.attrs["axis"])
        elif op == "log_softmax":
            r = torch.log_softmax(ev(ins[0]), dim=node.attrs["axis"])
        elif op == "softmax_xent":
            labels, logits = ev(ins[0]), ev(ins[1])
            r = -(labels * torch.log_softmax(logits, dim=-1)).sum(dim=-1)
        elif op in ("reduce_sum", "reduce_mean", "reduce_max"):
            x = ev(ins[0])
            ax, kd = node.attrs["axis"], node.attrs["keep_dims"]
            if op == "reduce_sum":
                r = x.sum() if ax is None else x.sum(dim=ax, keepdim=kd)
            elif op == "reduce_mean":
                x = x if x.is_floating_point() else x.float()
                r = x.mean() if ax is None else x.mean(dim=ax, keepdim=kd)
            else:
                r = x.max() if ax is None else x.amax(dim=ax, keepdim=kd)
        elif op == "argmax":
            r = ev(ins[0]).argmax(dim=node.attrs["axis"])
        elif op == "cast":
            r = ev(ins[0]).to(_TORCH_DTYPES[node.attrs["dtype"]])
        elif op == "reshape":
            r = ev(ins[0]).reshape(node.attrs["shape"])
        elif op == "transpose":
            x = ev(ins[0])
            r = x.permute(*node.attrs["perm"]) if node.attrs["perm"] else x.t()
        elif op == "scalar_summary":
            r = Summary([(node.attrs["tag"], float(ev(ins[0])))])
        elif op == "merge_summary":
            vals = []
            for i in ins:
                vals.extend(ev(i).values)
            r = Summary(vals)
        elif op in ("assign", "assign_add"):
            val = ev(ins[1])
            if self.store.is_remote:
                raise NotImplementedError("assign on parameter-server variables: use an optimizer or Saver.restore")
            dst = self.var_tensor(ins[0])
            dst.copy_(val) if op == "assign" else dst.add_(val)
            r = dst
        elif op == "gradient":
            loss, var = ins[0], node.attrs["wrt"]
            leaf = self._var_view(var).detach().clone().requires_grad_(True)
            val = self._eval(loss, feed, {}, {var: leaf})
            (r,) = torch.autograd.grad(val, [leaf], allow_unused=True)
            if r is None:
                r = torch.zeros_like(leaf)
        else:
            raise NotImplementedError(f"op {op!r} ({node.name})")
        cache[node] = r
        return r

    # -- training ------------------------------------------------------------------------
    def _engine_for(self, m, batch: int) -> MLPEngine:
        key = (id(m["loss"]), batch)
        eng = self._engines.get(key)
        if eng is None:
            names = {r: m[r].op_name for r in ("W1", "b1", "W2", "b2")}
            params = {r: self.layout.view(self.theta, n) for r, n in names.items()}
            grads = {r: self.layout.view(self.grad, n) for r, n in names.items()}
            eng = MLPEngine(params, grads, batch=batch, lr=0.0, act=m["act"], hist_cap=1)
            # loss and batch accuracy side by side: ONE 8-byte device-to-host copy per step
            # ([2]: the step counter before the push of a fused asynchronous step, see _async_descriptor)
            eng._stats = torch.zeros(4, dtype=torch.float32, device=self.device)
            eng.loss_hist, eng.acc_hist = eng._stats[0:1], eng._stats[1:2]
            K, C = params["W1"].shape[0], params["W2"].shape[1]
            # the fed batch travels as ONE host-to-device copy: x and y side by side in a pinned host buffer
            # (reused every step: the step's blocking statistics fetch orders it after the copy) and in one
            # device buffer the kernels read through two views
            stage = torch.zeros(batch * (K + C), dtype=torch.float32, device=self.device)
            eng._stage_dev = stage
            eng._x_stage = stage[: batch * K].view(batch, K)
            eng._y_stage = stage[batch * K:].view(batch, C)
            eng._stage_host = None
            if self.device.type == "cuda" and (batch * K) % 4 == 0:
                host = torch.zeros(batch * (K + C), dtype=torch.float32).pin_memory()
                eng._stage_host = host
                eng._stage_host_x = host[: batch * K].view(batch, K).numpy()
                eng._stage_host_y = host[batch * K:].view(batch, C).numpy()
            eng.bind_data(eng._x_stage, eng._y_stage, batches_per_epoch=1)
            # a local session applies plain SGD inside the step's own two launches (no separate optimizer launch)
            eng.global_step = None
            self._engines[key] = eng
        return eng

    @staticmethod
    def _as_f32(v) -> torch.Tensor:
        if isinstance(v, torch.Tensor):
            return v.to(torch.float32)
        return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))

    def _dense_engine_for(self, d, batch: int):
        """The engine of a matched fully connected stack (see :func:`match_dense_stack`) for one batch size: parameters
        and gradients are views of the session's slabs, the fed batch lands in a staging pair it reads as batch 0."""
        from .models.dense import DenseStackEngine

        key = ("dense", id(d["loss"]), batch)
        eng = self._engines.get(key)
        if eng is None:
            view = lambda slab, v: None if v is None else self.layout.view(slab, v.op_name)  # noqa: E731
            weights = [view(self.theta, W) for W, _ in d["layers"]]
            biases = [view(self.theta, b) for _, b in d["layers"]]
            grads = [(view(self.grad, W), view(self.grad, b)) for W, b in d["layers"]]
            eng = DenseStackEngine(weights, biases, d["acts"], batch=batch, lr=0.0, grads=grads, hist_cap=1)
            eng._x_stage = torch.zeros((batch, eng.dims[0]), dtype=torch.float32, device=self.device)
            eng._y_stage = torch.zeros((batch, eng.dims[-1]), dtype=torch.float32, device=self.device)
            eng.bind_data(eng._x_stage, eng._y_stage, batches_per_epoch=1)
            self._engines[key] = eng
        return eng

    def _stage(self, eng: MLPEngine, m, feed):
        x, y = feed.get(m["x"]), feed.get(m["y_"])
        if x is None or y is None:
            raise ValueError("both input placeholders must be fed")
        if eng._stage_host is not None and not isinstance(x, torch.Tensor) and not isinstance(y, torch.Tensor):
            np.copyto(eng._stage_host_x, x, casting="same_kind")
            np.copyto(eng._stage_host_y, y, casting="same_kind")
            eng._stage_dev.copy_(eng._stage_host, non_blocking=True)
            return
        xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        yt = y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32))
        eng._x_stage.copy_(xt, non_blocking=True)
        eng._y_stage.copy_(yt, non_blocking=True)

    def _plan(self, train_op):
        plan = self._plans.get(id(train_op))
        if plan is None:
            loss = train_op.attrs["loss"]
            m = match_mlp(loss) if loss is not None else None
            if m is not None:
                wanted = {m[r] for r in ("W1", "b1", "W2", "b2")}
                if set(train_op.attrs["var_list"]) != wanted:
                    m = None   # custom var_list: take the generic path
            dense = None
            if m is None and loss is not None and os.environ.get("DTF_DENSE_ENGINE", "1") != "0":
                dense = match_dense_stack(loss)
                if dense is not None and set(train_op.attrs["var_list"]) != set(dense["variables"]):
                    dense = None
            head = m if m is not None else dense
            acc_nodes = [n for n in self.graph.nodes if head is not None and match_accuracy(n, head)]
            plan = {"mlp": m, "dense": dense, "acc_nodes": acc_nodes}
            self._plans[id(train_op)] = plan
        return plan

    def _async_descriptor(self, train_op, eng: MLPEngine, m):
        """The shard table of the fused asynchronous step (``MLPEngine.step_async`` with a forward that reads the
        variables from the mapped shards and a backward that pushes ``-lr * g`` per element: 2 launches, no pull, no
        separate apply), or None when this train op / store / model does not qualify.  Decided once per engine."""
        desc = getattr(eng, "_async_desc", False)
        if desc is not False:
            return desc
        desc = None
        opt = train_op.attrs["optimizer"]
        if (self.device.type == "cuda" and hasattr(self.store, "push_descriptor") and not isinstance(opt, SyncReplicasOptimizer)
                and not opt.use_locking and os.environ.get("DTF_ASYNC_FUSED_PER_BATCH", "1") != "0"):
            names = {r: m[r].op_name for r in ("W1", "b1", "W2", "b2")}
            trainable = {s.name for s in self.layout.specs if s.trainable}
            if trainable == set(names.values()):
                from .ops import hip_backend as hb

                offs = {r: self.layout.var_range[n][0] for r, n in names.items()}
                try:
                    desc = self.store.push_descriptor(offs, bump_step=train_op.attrs["global_step"] is not None,
                                                      read_shards=True)
                    desc.state = eng._stats[2:3].data_ptr()      # receives global_step as it was before the step's push
                    keep = eng.lr.clone()
                    eng.lr.zero_()                       # loads the kernels and checks the shard table: adds zero, no count
                    eng.step_async(0, self.store.push_descriptor(offs, bump_step=False, read_shards=True))
                    eng.lr.copy_(keep)
                    eng.reset_step_workspace(0)
                except (ValueError, hb.HipKernelError):
                    desc = None
        eng._async_desc = desc
        return desc

    def _shard_served_call(self, flist, feed):
        """For a ``sess.run`` of ONE train op of the fused model on mapped shards whose other fetches are served by the
        step itself (cost, accuracy, summaries of them) or are its step counter: how the step works on the shards, so
        that the call needs no pull -- "pushed" (asynchronous: forward from the shards, backward that pushes) or
        "deposit" (synchronous replicas: forward from the shards, gradient into the worker's mailbox).  None: pull as
        usual."""
        trains = [f for f in flist if _is(f, "train")]
        if len(trains) != 1 or not hasattr(self.store, "push_descriptor"):
            return None
        train_op = trains[0]
        plan = self._plan(train_op)
        m = plan["mlp"]
        if m is None or m["x"] not in feed or m["y_"] not in feed:
            return None
        gs_var = train_op.attrs["global_step"]
        for f in flist:
            if f is train_op or f is None:
                continue
            if _is(f, "variable"):
                if f is not gs_var:
                    return None
            elif not self._served_by_step(f, m, plan):
                return None
        opt = train_op.attrs["optimizer"]
        if not self._local_step_ready:           # (the shards are allocated and mapped before the first view of them)
            self.store.sync_local_step()
            self._local_step_ready = True
        eng = self._engine_for(m, int(feed[m["x"]].shape[0]))
        if isinstance(opt, SyncReplicasOptimizer):
            if self._mailbox_gradients(eng, m) and self._sync_shards(eng, m) is not None:
                return "deposit"
            return None
        if opt.use_locking or self._async_descriptor(train_op, eng, m) is None:
            return None
        return "pushed"

    def _sync_shards(self, eng: MLPEngine, m):
        """Shard table for the gradient step of a synchronous replica whose forward reads the variables from the mapped
        shards (``MLPEngine.step_grad(shards=...)``), or None.  Decided once per engine."""
        desc = getattr(eng, "_sync_shards_desc", False)
        if desc is not False:
            return desc
        desc = None
        if (self.device.type == "cuda" and hasattr(self.store, "push_descriptor")
                and os.environ.get("DTF_SYNC_FOLD_PULL", "1") != "0"):
            from .ops import hip_backend as hb

            offs = {r: self.layout.var_range[m[r].op_name][0] for r in ("W1", "b1", "W2", "b2")}
            try:
                desc = self.store.push_descriptor(offs, bump_step=False, read_shards=True)
                # loads the kernels and checks the shard table (zero times the gradient, into the session's own slab)
                eng.step_grad(0, 0.0, shards=desc)
                eng.reset_step_workspace(0)
            except (ValueError, RuntimeError, hb.HipKernelError):
                desc = None
        eng._sync_shards_desc = desc
        return desc

    def _served_by_step(self, f, m, plan, _depth: int = 0) -> bool:
        """True when ``f`` is computed from the fused step's own results (its loss, its batch accuracy, summaries of
        them) without reading a variable."""
        if not isinstance(f, G.Tensor) or _depth > 8:
            return False
        if f is m["loss"] or f in plan["acc_nodes"]:
            return True
        if f.op in ("scalar_summary", "merge_summary"):
            return all(self._served_by_step(i, m, plan, _depth + 1) for i in f.inputs)
        return False

    def _run_train(self, train_op, feed, cache):
        opt: Optimizer = train_op.attrs["optimizer"]
        plan = self._plan(train_op)
        m = plan["mlp"]
        lr = opt.lr_value()
        sync = isinstance(opt, SyncReplicasOptimizer) and self.store.is_remote
        applied = False
        if m is not None and m["x"] in feed:
            xv = feed[m["x"]]
            batch = int(xv.shape[0])
            eng = self._engine_for(m, batch)
            with span("stage"):
                self._stage(eng, m, feed)
            with span("step"):
                if not self.store.is_remote:
                    # no parameter server: the SGD update happens inside the step's two launches
                    self._bind_local_update(eng, train_op, lr)
                    eng.step_fused(0)
                    applied = True
                elif sync and self._mailbox_gradients(eng, m):
                    # the backward kernel writes this step's gradient straight into the worker's mailbox row
                    eng.step_grad(0, 1.0, slot=self.store.next_mailbox_slot(), shards=self._sync_shards(eng, m))
                    applied = "deposited"
                elif not sync and self._async_descriptor(train_op, eng, m) is not None:
                    # asynchronous worker on mapped shards: the forward reads the variables from the shards, the
                    # backward pushes -lr * g per element and counts the step (no pull, no separate apply)
                    if getattr(eng, "_async_lr", None) != lr:
                        eng.set_lr(lr)
                        eng._async_lr = lr
                    eng.step_async(0, eng._async_desc)
                    applied = "pushed"
                    self.pushed_steps += 1
                else:
                    eng.step_grad(0, 1.0)
            with span("fetch_stats"):
                # the step's only wait for the device
                host = self._stats_to_host(eng._stats[:3] if applied == "pushed" else eng._stats[:2])
            if applied == "pushed" and train_op.attrs["global_step"] is not None:
                cache[train_op.attrs["global_step"]] = host[2]
            cache[m["loss"]] = host[0]
            cache[("acc", id(m["loss"]))] = host[1]
            for n in plan["acc_nodes"]:
                cache[n] = cache[("acc", id(m["loss"]))]
            self.fused_steps += 1
        elif plan.get("dense") is not None and plan["dense"]["x"] in feed and plan["dense"]["y_"] in feed:
            d = plan["dense"]
            eng = self._dense_engine_for(d, int(feed[d["x"]].shape[0]))
            with span("stage"):
                eng._x_stage.copy_(self._as_f32(feed[d["x"]]), non_blocking=True)
                eng._y_stage.copy_(self._as_f32(feed[d["y_"]]), non_blocking=True)
            with span("step"):
                if not self.store.is_remote:
                    gs_var = train_op.attrs["global_step"]
                    eng.set_lr(lr) if getattr(eng, "_lr_val", None) != lr else None
                    eng._lr_val = lr
                    eng.global_step = None if gs_var is None else self._var_view(gs_var).view(1)
                    eng.step_fused(0)
                    applied = True
                else:
                    eng.step_grad(0, 1.0)
            with span("fetch_stats"):
                host = self._stats_to_host(torch.stack((eng.loss_sum[0], eng.correct[0].to(torch.float32)))) / float(eng.B)
            cache[d["loss"]] = host[0]
            cache[("acc", id(d["loss"]))] = host[1]
            for n in plan["acc_nodes"]:
                cache[n] = host[1]
            self.fused_steps += 1
            self.dense_steps += 1
        else:
            self.grad.zero_()
            var_list = train_op.attrs["var_list"]
            leaves = {v: self._var_view(v).detach().clone().requires_grad_(True) for v in var_list}
            fcache: Dict[G.Tensor, Any] = {}
            with span("step_generic"):
                loss_val = self._eval(train_op.attrs["loss"], feed, fcache, leaves)
                grads = torch.autograd.grad(loss_val, [leaves[v] for v in var_list], allow_unused=True)
            for v, g in zip(var_list, grads):
                if g is not None:
                    self.layout.view(self.grad, v.op_name).copy_(g)
            for k, val in fcache.items():   # forward values seen by the other fetches
                cache[k] = val.detach() if isinstance(val, torch.Tensor) else val
            self.generic_steps += 1
        # ---- push ----
        if applied == "deposited":      # (the statistics fetch above synchronised the stream: the bytes are in HBM)
            if not self._local_step_ready:
                self.store.sync_local_step()
                self._local_step_ready = True
            with span("push_accumulate_token_wait"):
                self.store.push_deposited_and_wait()
            return
        if applied:
            return
        if sync:
            if not self._local_step_ready:
                self.store.sync_local_step()
                self._local_step_ready = True
            if hasattr(self.store, "push_accumulate_and_wait"):
                with span("push_accumulate_token_wait"):      # deposit + wait for the token: one round trip
                    self.store.push_accumulate_and_wait(self.grad)
            else:
                with span("push_accumulate"):
                    self.store.push_accumulate(self.grad)
                with span("token_wait"):
                    self.store.dequeue_token()
        elif isinstance(opt, SyncReplicasOptimizer):
            # one local replica: aggregate of one
            with span("push_apply"):
                self.store.push_apply(self.grad, lr, True, bump_step=train_op.attrs["global_step"] is not None)
        else:
            with span("push_apply"):
                self.store.push_apply(self.grad, lr, opt.use_locking,
                                      bump_step=train_op.attrs["global_step"] is not None)

    def _stats_to_host(self, stats: torch.Tensor) -> torch.Tensor:
        """The two statistics of a fused step on the host; waits for the step."""
        if self._host_words is None or stats.device.type != "cuda":
            return stats.cpu()
        n = stats.numel()            # 2, or 3 with the step counter of a fused asynchronous step
        with self._on_device():
            self._host_words[0:n].copy_(stats, non_blocking=True)
            # (with the device named: record() on its own asks torch.cuda.current_stream(), whose device lookup is a
            # driver call of 11 us; _on_device() instead of a device guard per step for the same reason)
            self._host_evt.record(torch.cuda.current_stream(self.device))
        self._host_evt.synchronize()
        return self._host_words[0:n].clone()

    def _mailbox_gradients(self, eng: MLPEngine, m) -> bool:
        """Synchronous replicas on a transport with GPU-resident mailboxes: let the engine write its gradients INTO
        this worker's mailbox rows (no copy per step).  Only when the fused model covers every trainable variable."""
        state = getattr(eng, "_mailbox_state", None)
        if state is None:
            state = False
            if hasattr(self.store, "mailbox_grad_slots") and os.environ.get("DTF_MAILBOX_GRADIENTS", "1") != "0":
                names = {r: m[r].op_name for r in ("W1", "b1", "W2", "b2")}
                trainable = {s.name for s in self.layout.specs if s.trainable}
                if trainable == set(names.values()):
                    if not self._local_step_ready:       # (allocation of the shards precedes the first mapping)
                        self.store.sync_local_step()
                        self._local_step_ready = True
                    eng.bind_grad_slots(self.store.mailbox_grad_slots(names))
                    state = True
            eng._mailbox_state = state
        return state

    def _bind_local_update(self, eng: MLPEngine, train_op, lr: float):
        """Point the engine's fused step at this train op's learning rate and step counter (cached per engine)."""
        gs_var = train_op.attrs["global_step"]
        key = (float(lr), id(gs_var))
        if getattr(eng, "_local_update_key", None) == key:
            return
        eng.set_lr(lr)
        eng.global_step = None if gs_var is None else self._var_view(gs_var).view(1)
        eng._desc = None
        eng._local_update_key = key

    def _init_tokens(self, node):
        opt = node.attrs["optimizer"]
        if self.store.is_remote:
            clients = [self.server.client("ps", k) for k in range(self.store.n)]
            self.store.enqueue_tokens(clients, self.store.global_step(), node.attrs["num_tokens"])

    def _try_fused_eval(self, f, feed, cache):
        """Loss / accuracy of the matched model over a fed batch, on the fused forward kernels."""
        for plan in list(self._plans.values()):
            if plan.get("mlp") is None and plan.get("dense") is None:
                continue
            m = plan["mlp"] if plan.get("mlp") is not None else plan["dense"]
            is_loss = f is m["loss"]
            is_acc = (not is_loss) and match_accuracy(f, m)
            if not (is_loss or is_acc):
                continue
            if is_loss and f in cache:
                return float(cache[f])
            if is_acc and ("acc", id(m["loss"])) in cache:
                return float(cache[("acc", id(m["loss"]))])
            if m["x"] not in feed or m["y_"] not in feed:
                return None
            if self.device.type != "cuda":
                return None
            x = torch.as_tensor(np.ascontiguousarray(feed[m["x"]], dtype=np.float32)).to(self.device)
            y = torch.as_tensor(np.ascontiguousarray(feed[m["y_"]], dtype=np.float32)).to(self.device)
            eng = self._engine_for(m, 1) if plan.get("mlp") is not None else self._dense_engine_for(m, 1)
            loss, acc = eng.evaluate(x, y)
            cache[m["loss"]] = torch.tensor(loss)
            cache[("acc", id(m["loss"]))] = torch.tensor(acc)
            return loss if is_loss else acc
        return None


InteractiveSession = Session
