"""Host-side ranges: profiler markers and a per-run timeline.

The reference's only tracing aids are ``log_device_placement`` and wall-clock
prints (tfdist_between.py:15, :100-111); TF-1.x users additionally expect
``sess.run(..., options=RunOptions(trace_level=FULL_TRACE), run_metadata=md)``
followed by ``timeline.Timeline(md.step_stats).generate_chrome_trace_format()``.
Both are served by one mechanism here:

* ``span(name)`` brackets a phase of the training loop (``pull``, ``stage``,
  ``step``, ``push``, ``apply``, ``token`` ...).  When tracing is off it returns a
  shared no-op object: one global lookup and two empty method calls per phase.
* With ``DTF_TRACE=roctx`` (or ``enable(roctx=True)``) every span also pushes a
  range into the ROCm marker library, so ``rocprofv3 --marker-trace`` lays the
  framework's phases over the kernel trace.
* With a ``Collector`` active (``DTF_TRACE=timeline`` installs a process-wide
  one, ``RunOptions`` installs one for a single ``run`` call) spans are recorded
  as (name, start, duration, thread, args) and can be written in the Chrome
  trace-event format that ``chrome://tracing`` / Perfetto read.

Spans measure HOST time.  A phase that only enqueues device work is as long as
its enqueue; pass ``sync=True`` to ``Collector`` to drain the device at the end
of each span, which serialises the step but attributes device time to phases.
"""
from __future__ import annotations

import ctypes
import json
import os
import threading
import time
from typing import Any, Dict, List, Optional

__all__ = ["span", "mark", "enable", "disable", "enabled", "Collector", "RunOptions", "RunMetadata", "Timeline",
           "StepStats", "active_collector"]

_ROCTX_NAMES = ("librocprofiler-sdk-roctx.so", "libroctx64.so")


class _Roctx:
    """Lazy ctypes binding of the marker library (push / pop / mark)."""

    def __init__(self):
        self.lib = None
        self.tried = False

    def load(self) -> bool:
        if self.tried:
            return self.lib is not None
        self.tried = True
        roots = [os.environ.get("ROCM_PATH", "/opt/rocm") + "/lib/", ""]
        for root in roots:
            for name in _ROCTX_NAMES:
                try:
                    lib = ctypes.CDLL(root + name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.argtypes = []
                    lib.roctxRangePop.restype = ctypes.c_int
                    lib.roctxMarkA.argtypes = [ctypes.c_char_p]
                    lib.roctxMarkA.restype = None
                    self.lib = lib
                    return True
                except (OSError, AttributeError):
                    continue
        return False


_roctx = _Roctx()


class _State:
    __slots__ = ("roctx", "collector", "on")

    def __init__(self):
        self.roctx = False
        self.collector: Optional["Collector"] = None
        self.on = False

    def refresh(self):
        self.on = self.roctx or self.collector is not None


_state = _State()
_tls = threading.local()


class _NullSpan:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def set(self, **_):
        pass


_NULL = _NullSpan()


class _Span:
    __slots__ = ("name", "args", "t0", "pushed", "col")

    def __init__(self, name: str, args: Optional[Dict[str, Any]]):
        self.name = name
        self.args = args
        self.pushed = False
        self.col = None

    def set(self, **kw):
        """Attach values known only inside the span (bytes moved, step number)."""
        if self.args is None:
            self.args = {}
        self.args.update(kw)

    def __enter__(self):
        if _state.roctx and _roctx.lib is not None:
            _roctx.lib.roctxRangePushA(self.name.encode())
            self.pushed = True
        self.col = _tls.__dict__.get("collector") or _state.collector
        self.t0 = time.perf_counter_ns()
        return self

    def __exit__(self, *exc):
        col = self.col
        if col is not None:
            if col.sync:
                col._drain()
            t1 = time.perf_counter_ns()
            col._add(self.name, self.t0, t1 - self.t0, self.args)
        if self.pushed:
            _roctx.lib.roctxRangePop()
        return False


def span(name: str, **args):
    """Context manager around one phase; free when tracing is off."""
    if not _state.on and "collector" not in _tls.__dict__:
        return _NULL
    return _Span(name, args or None)


def mark(name: str, **args):
    """Instant event (a checkpoint written, a stale gradient dropped)."""
    if not _state.on and "collector" not in _tls.__dict__:
        return
    if _state.roctx and _roctx.lib is not None:
        _roctx.lib.roctxMarkA(name.encode())
    col = _tls.__dict__.get("collector") or _state.collector
    if col is not None:
        col._add(name, time.perf_counter_ns(), 0, args or None, instant=True)


def enabled() -> bool:
    return _state.on or "collector" in _tls.__dict__


def active_collector() -> Optional["Collector"]:
    return _tls.__dict__.get("collector") or _state.collector


def enable(roctx: bool = False, timeline: bool = False, sync: bool = False) -> Optional["Collector"]:
    """Process-wide switch.  Returns the process collector when ``timeline`` is set."""
    if roctx:
        if not _roctx.load():
            raise RuntimeError("the ROCm marker library (%s) could not be loaded" % " / ".join(_ROCTX_NAMES))
        _state.roctx = True
    if timeline and _state.collector is None:
        _state.collector = Collector(sync=sync)
    _state.refresh()
    return _state.collector


def disable():
    _state.roctx = False
    _state.collector = None
    _state.refresh()


class Collector:
    """Records spans.  ``with Collector() as c:`` scopes it to the current thread."""

    def __init__(self, sync: bool = False, process_name: Optional[str] = None, capacity: int = 1 << 20):
        self.sync = sync
        self.process_name = process_name or "pid %d" % os.getpid()
        self.capacity = capacity
        self.events: List[tuple] = []
        self.dropped = 0
        self._lock = threading.Lock()
        self._prev = None

    # -- scoping ---------------------------------------------------------------------
    def __enter__(self):
        self._prev = _tls.__dict__.get("collector")
        _tls.collector = self
        return self

    def __exit__(self, *exc):
        if self._prev is None:
            _tls.__dict__.pop("collector", None)
        else:
            _tls.collector = self._prev
        return False

    # -- recording -------------------------------------------------------------------
    def _drain(self):
        import torch

        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()

    def _add(self, name, t0_ns, dur_ns, args, instant=False):
        ev = (name, t0_ns, dur_ns, threading.get_ident(), args, instant)
        with self._lock:
            if len(self.events) < self.capacity:
                self.events.append(ev)
            else:
                self.dropped += 1

    def clear(self):
        with self._lock:
            self.events = []
            self.dropped = 0

    # -- reporting -------------------------------------------------------------------
    def totals(self) -> Dict[str, Dict[str, float]]:
        """name -> {count, total_ms, mean_us, max_us} over the recorded spans."""
        out: Dict[str, Dict[str, float]] = {}
        for name, _t0, dur, _tid, _a, instant in self.events:
            if instant:
                continue
            d = out.setdefault(name, {"count": 0, "total_ms": 0.0, "max_us": 0.0})
            d["count"] += 1
            d["total_ms"] += dur / 1e6
            d["max_us"] = max(d["max_us"], dur / 1e3)
        for d in out.values():
            d["mean_us"] = d["total_ms"] * 1e3 / max(1, d["count"])
        return out

    def format_totals(self) -> str:
        rows = sorted(self.totals().items(), key=lambda kv: -kv[1]["total_ms"])
        lines = ["%-24s %8s %12s %10s %10s" % ("span", "count", "total ms", "mean us", "max us")]
        for name, d in rows:
            lines.append("%-24s %8d %12.3f %10.1f %10.1f" % (name, d["count"], d["total_ms"], d["mean_us"], d["max_us"]))
        return "\n".join(lines)

    def chrome_trace(self, pid: Optional[int] = None) -> Dict[str, Any]:
        pid = os.getpid() if pid is None else pid
        tids: Dict[int, int] = {}
        evs: List[Dict[str, Any]] = [{"name": "process_name", "ph": "M", "pid": pid, "args": {"name": self.process_name}}]
        base = min((e[1] for e in self.events), default=0)
        for name, t0, dur, tid, args, instant in self.events:
            t = tids.setdefault(tid, len(tids))
            ev = {"name": name, "cat": "dtf", "pid": pid, "tid": t, "ts": (t0 - base) / 1e3}
            if instant:
                ev.update(ph="i", s="t")
            else:
                ev.update(ph="X", dur=dur / 1e3)
            if args:
                ev["args"] = {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in args.items()}
            evs.append(ev)
        for tid, t in tids.items():
            evs.append({"name": "thread_name", "ph": "M", "pid": pid, "tid": t, "args": {"name": "thread %d" % tid}})
        return {"traceEvents": evs, "displayTimeUnit": "ns", "otherData": {"dropped": self.dropped}}

    def save(self, path: str) -> str:
        d = os.path.dirname(os.path.abspath(path))
        os.makedirs(d, exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "w") as f:
            json.dump(self.chrome_trace(), f)
        os.replace(tmp, path)
        return path


# ---------------------------------------------------------------------------
# TF-1.x surface: RunOptions / RunMetadata / timeline.Timeline
class RunOptions:
    NO_TRACE = 0
    SOFTWARE_TRACE = 1
    HARDWARE_TRACE = 2
    FULL_TRACE = 3

    def __init__(self, trace_level: int = 0, timeout_in_ms: int = 0, **_ignored):
        self.trace_level = trace_level
        self.timeout_in_ms = timeout_in_ms

    @property
    def wants_trace(self) -> bool:
        return self.trace_level != RunOptions.NO_TRACE

    @property
    def wants_device_time(self) -> bool:
        """HARDWARE / FULL: drain the device at each phase end so device time lands in the phase."""
        return self.trace_level in (RunOptions.HARDWARE_TRACE, RunOptions.FULL_TRACE)


class StepStats:
    """What one traced ``run`` call recorded."""

    def __init__(self, collector: Optional[Collector] = None):
        self.collector = collector or Collector()

    @property
    def events(self):
        return self.collector.events

    def totals(self):
        return self.collector.totals()


class RunMetadata:
    def __init__(self):
        self.step_stats = StepStats()


class Timeline:
    """``Timeline(run_metadata.step_stats).generate_chrome_trace_format()`` as in tensorflow.python.client."""

    def __init__(self, step_stats: StepStats, graph=None):
        self.step_stats = step_stats

    def generate_chrome_trace_format(self, show_dataflow: bool = True, show_memory: bool = False) -> str:
        return json.dumps(self.step_stats.collector.chrome_trace())


def _from_env():
    v = os.environ.get("DTF_TRACE", "").strip().lower()
    if not v or v in ("0", "off", "false"):
        return
    want = set(v.replace("+", ",").split(","))
    if "1" in want or "all" in want:
        want |= {"roctx", "timeline"}
    if "roctx" in want:
        try:
            enable(roctx=True)
        except RuntimeError:
            pass
    if "timeline" in want:
        col = enable(timeline=True, sync="sync" in want)
        out = os.environ.get("DTF_TRACE_FILE")
        if out:
            import atexit

            atexit.register(lambda: col.events and col.save(out.replace("%p", str(os.getpid()))))


_from_env()
