"""ctypes binding of ``_lib/libdtf_rt.so`` (runtime/ps_server.cpp, runtime/records.cpp).

* :class:`TaskServer`  — in-process native server of one cluster task (control
  plane + host-memory parameter store).
* :class:`StoreClient` — connection to a task server; every method is one
  request/response and releases the GIL while it blocks.
* CRC32C / event-file / tensor-bundle writers.
"""
from __future__ import annotations

import ctypes as C
import struct
import time
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

_LIB_PATH = Path(__file__).resolve().parent.parent / "_lib" / "libdtf_rt.so"
_lib = None

OK, TIMEOUT, BAD_REQUEST, IO_ERROR, TOO_SMALL = 0, 1, 2, 3, 4
FLAG_LOCK = 1
FLAG_FANOUT = 2
FLAG_RELEASE = 4
FLAG_THEN_POP = 8
FLAG_BUMP = 16
(OP_PING, OP_SET, OP_GET, OP_ADD, OP_WAIT_GE, OP_QPUSH, OP_QPOP, OP_DEL, OP_SHUTDOWN, OP_QLEN,
 OP_BUF_CREATE, OP_BUF_WRITE, OP_BUF_READ, OP_BUF_AXPY, OP_ACC_APPLY, OP_ACC_TAKE_APPLY, OP_ACC_SET_STEP,
 OP_BUF_ADD_SCALAR, OP_ACC_INFO, OP_QPUSH_POP, OP_QPOP_N) = range(21)


class RuntimeError_(RuntimeError):
    pass


class StoreTimeout(TimeoutError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            from .. import build

            build.build_runtime()
        L = C.CDLL(str(_LIB_PATH))
        vp, i32, i64, u64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
        L.dtf_server_start.restype = vp
        L.dtf_server_start.argtypes = [C.c_char_p, i32, C.POINTER(i32)]
        L.dtf_server_port.argtypes = [vp]
        L.dtf_server_wait.argtypes = [vp, i32]
        L.dtf_server_stats.argtypes = [vp, C.POINTER(u64)]
        L.dtf_server_stats.restype = None
        L.dtf_server_stop.argtypes = [vp]
        L.dtf_server_stop.restype = None
        L.dtf_client_connect.restype = vp
        L.dtf_client_connect.argtypes = [C.c_char_p, i32, i32]
        L.dtf_client_close.argtypes = [vp]
        L.dtf_client_close.restype = None
        L.dtf_client_call.argtypes = [vp, i32, C.c_char_p, i32, i64, i64, f64, vp, u64, C.POINTER(i64), vp, u64,
                                      C.POINTER(u64)]
        L.dtf_crc32c.restype = C.c_uint32
        L.dtf_crc32c.argtypes = [vp, u64]
        L.dtf_crc32c_masked.restype = C.c_uint32
        L.dtf_crc32c_masked.argtypes = [vp, u64]
        L.dtf_events_open.restype = vp
        L.dtf_events_open.argtypes = [C.c_char_p, C.c_char_p]
        L.dtf_events_path.restype = C.c_char_p
        L.dtf_events_path.argtypes = [vp]
        L.dtf_events_write_scalars.argtypes = [vp, f64, i64, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]
        L.dtf_events_write_scalar_series.argtypes = [vp, f64, i32, C.POINTER(i64), i32, C.POINTER(C.c_char_p),
                                                     C.POINTER(C.c_float)]
        L.dtf_events_write_raw.argtypes = [vp, vp, u64]
        L.dtf_events_flush.argtypes = [vp]
        L.dtf_events_close.argtypes = [vp]
        L.dtf_events_close.restype = None
        L.dtf_bundle_begin.restype = vp
        L.dtf_bundle_begin.argtypes = [C.c_char_p]
        L.dtf_bundle_add_f32.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32]
        L.dtf_bundle_finish.argtypes = [vp]
        L.dtf_board_open.restype = vp
        L.dtf_board_open.argtypes = [C.c_char_p, i32, i32]
        L.dtf_board_close.argtypes = [vp]
        L.dtf_board_close.restype = None
        L.dtf_board_unlink.argtypes = [vp]
        L.dtf_board_arrive.argtypes = [vp, i64, i64, i64]
        L.dtf_board_take.argtypes = [vp, i32, i32, C.POINTER(i64), C.POINTER(i32)]
        L.dtf_board_release.argtypes = [vp, i64, i64, i32]
        L.dtf_board_wait_token.argtypes = [vp, i64, i32, C.POINTER(i64)]
        L.dtf_board_attached.argtypes = [vp]
        _lib = L
    return _lib


def crc32c(data: bytes) -> int:
    return lib().dtf_crc32c(data, len(data))


def crc32c_masked(data: bytes) -> int:
    return lib().dtf_crc32c_masked(data, len(data))


class TaskServer:
    """The native server of one task; ``port=0`` picks a free port."""

    def __init__(self, host: str = "0.0.0.0", port: int = 0):
        out = C.c_int(0)
        self._h = lib().dtf_server_start(host.encode(), int(port), C.byref(out))
        if not self._h:
            raise OSError(f"cannot bind task server on {host}:{port}")
        self.port = out.value

    def wait(self, timeout: Optional[float] = None) -> bool:
        """Block until a client requested shutdown; returns False on timeout."""
        if timeout is not None:
            return bool(lib().dtf_server_wait(self._h, int(timeout * 1000)))
        while True:  # wake up regularly so that Ctrl-C works
            if lib().dtf_server_wait(self._h, 500):
                return True

    def stats(self) -> Tuple[int, int, int]:
        arr = (C.c_uint64 * 3)()
        lib().dtf_server_stats(self._h, arr)
        return int(arr[0]), int(arr[1]), int(arr[2])

    def stop(self):
        if self._h:
            lib().dtf_server_stop(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.stop()
        except Exception:
            pass


class StoreClient:
    def __init__(self, host: str, port: int, timeout: float = 60.0):
        self.addr = (host, int(port))
        self._h = lib().dtf_client_connect(host.encode(), int(port), int(timeout * 1000))
        if not self._h:
            raise ConnectionError(f"cannot reach task server {host}:{port} within {timeout}s")
        self._out = C.create_string_buffer(1 << 16)

    def close(self):
        if self._h:
            lib().dtf_client_close(self._h)
            self._h = None

    # -- raw call ---------------------------------------------------------------
    def _call(self, op: int, key: str = "", flags: int = 0, a0: int = 0, a1: int = 0, f0: float = 0.0,
              payload=None, payload_len: int = 0, out: Optional[np.ndarray] = None) -> Tuple[int, int, bytes]:
        if isinstance(payload, (bytes, bytearray)):
            buf, n = bytes(payload), len(payload)
        elif isinstance(payload, np.ndarray):
            assert payload.flags["C_CONTIGUOUS"]
            buf, n = payload.ctypes.data_as(C.c_void_p), payload.nbytes
        elif payload is None:
            buf, n = None, 0
        else:
            buf, n = payload, payload_len
        val, olen = C.c_int64(0), C.c_uint64(0)
        if out is not None:
            obuf, ocap = out.ctypes.data_as(C.c_void_p), out.nbytes
        else:
            obuf, ocap = C.cast(self._out, C.c_void_p), len(self._out)
        st = lib().dtf_client_call(self._h, op, key.encode(), flags, int(a0), int(a1), float(f0), buf, n,
                                   C.byref(val), obuf, ocap, C.byref(olen))
        if st == TOO_SMALL and out is None:
            self._out = C.create_string_buffer(int(olen.value) * 2)
            raise RuntimeError_("response larger than the receive buffer; retry")
        data = b""
        if out is None and st == OK and olen.value:
            data = self._out.raw[:olen.value]
        return st, int(val.value), data

    def _ok(self, st: int, what: str):
        if st == TIMEOUT:
            raise StoreTimeout(what)
        if st != OK:
            raise RuntimeError_(f"{what}: status {st} from {self.addr}")

    # -- control plane ------------------------------------------------------------
    def ping(self):
        self._ok(self._call(OP_PING)[0], "ping")

    def set(self, key: str, value: bytes):
        self._ok(self._call(OP_SET, key, payload=value)[0], f"set {key}")

    def get(self, key: str, timeout: float = 60.0) -> bytes:
        for _ in range(2):
            try:
                st, _, data = self._call(OP_GET, key, a0=int(timeout * 1000))
            except RuntimeError_:
                continue
            self._ok(st, f"get {key}")
            return data
        raise RuntimeError_(f"get {key} failed")

    def add(self, key: str, delta: int = 1) -> int:
        st, v, _ = self._call(OP_ADD, key, a0=delta)
        self._ok(st, f"add {key}")
        return v

    def wait_ge(self, key: str, value: int, timeout: float = 60.0) -> int:
        st, v, _ = self._call(OP_WAIT_GE, key, a0=value, a1=int(timeout * 1000))
        self._ok(st, f"wait {key} >= {value} (at {v})")
        return v

    def qpush(self, key: str, value: bytes, copies: int = 1) -> int:
        st, v, _ = self._call(OP_QPUSH, key, a0=copies, payload=value)
        self._ok(st, f"qpush {key}")
        return v

    def qpush_fanout(self, prefix: str, value: bytes, fanout: int, copies: int = 1) -> int:
        """``copies`` of ``value`` into each of the queues ``prefix/0 .. prefix/(fanout-1)`` in ONE request."""
        st, v, _ = self._call(OP_QPUSH, prefix, flags=FLAG_FANOUT, a0=copies, a1=fanout, payload=value)
        self._ok(st, f"qpush_fanout {prefix}")
        return v

    def qpop(self, key: str, timeout: float = 60.0) -> bytes:
        st, _, data = self._call(OP_QPOP, key, a0=int(timeout * 1000))
        self._ok(st, f"qpop {key}")
        return data

    def qpush_pop(self, push_key: str, value: bytes, pop_key: str, timeout: float = 60.0) -> bytes:
        """Push ``value`` onto ``push_key`` and block for an item of ``pop_key``: ONE round trip."""
        name = pop_key.encode()
        st, _, data = self._call(OP_QPUSH_POP, push_key, a0=int(timeout * 1000),
                                 payload=len(name).to_bytes(2, "little") + name + bytes(value))
        self._ok(st, f"qpush_pop {push_key} -> {pop_key}")
        return data

    def qpop_n(self, key: str, count: int, timeout: float = 60.0) -> Tuple[int, bytes]:
        """Up to ``count`` items in one request: waits until ``count`` are queued, returns those present when the
        time is up (TimeoutError if none).  -> (items returned, their bytes back to back)."""
        st, v, data = self._call(OP_QPOP_N, key, a0=int(count), a1=int(timeout * 1000))
        self._ok(st, f"qpop_n {key}")
        return v, data

    def qlen(self, key: str) -> int:
        st, v, _ = self._call(OP_QLEN, key)
        self._ok(st, f"qlen {key}")
        return v

    def delete(self, key: str) -> int:
        st, v, _ = self._call(OP_DEL, key)
        self._ok(st, f"del {key}")
        return v

    def shutdown_server(self):
        self._ok(self._call(OP_SHUTDOWN)[0], "shutdown")

    # -- parameter store ------------------------------------------------------------
    def buf_create(self, key: str, numel: int) -> bool:
        st, v, _ = self._call(OP_BUF_CREATE, key, a0=numel)
        self._ok(st, f"buf_create {key}")
        return bool(v)

    def buf_write(self, key: str, values: np.ndarray, offset: int = 0):
        values = np.ascontiguousarray(values, dtype=np.float32)
        self._ok(self._call(OP_BUF_WRITE, key, a0=offset, payload=values)[0], f"buf_write {key}")

    def buf_read(self, key: str, numel: int, offset: int = 0, out: Optional[np.ndarray] = None,
                 locked: bool = False) -> np.ndarray:
        if out is None:
            out = np.empty(numel, dtype=np.float32)
        st, _, _ = self._call(OP_BUF_READ, key, flags=FLAG_LOCK if locked else 0, a0=offset, a1=numel, out=out)
        self._ok(st, f"buf_read {key}")
        return out

    def buf_axpy(self, key: str, alpha: float, g: np.ndarray, offset: int = 0, locked: bool = False,
                 bump: Optional[int] = None) -> int:
        """buf[offset:offset+len(g)] += alpha * g, executed on the server.  ``bump``: element of the buffer that counts
        the updates (global_step): one is added to it in the same request and its new value returned."""
        g = np.ascontiguousarray(g, dtype=np.float32)
        flags = (FLAG_LOCK if locked else 0) | (FLAG_BUMP if bump is not None else 0)
        st, v, _ = self._call(OP_BUF_AXPY, key, flags=flags, a0=offset, a1=int(bump) if bump is not None else 0, f0=alpha,
                              payload=g)
        self._ok(st, f"buf_axpy {key}")
        return v

    def buf_add_scalar(self, key: str, offset: int, delta: float) -> int:
        st, v, _ = self._call(OP_BUF_ADD_SCALAR, key, a0=offset, f0=delta)
        self._ok(st, f"buf_add_scalar {key}")
        return v

    def acc_apply(self, key: str, g: np.ndarray, local_step: int) -> bool:
        """Push a gradient into a conditional accumulator; False = dropped as stale."""
        g = np.ascontiguousarray(g, dtype=np.float32)
        st, v, _ = self._call(OP_ACC_APPLY, key, a0=local_step, payload=g)
        self._ok(st, f"acc_apply {key}")
        return bool(v)

    def acc_apply_then_pop(self, key: str, g: np.ndarray, local_step: int, pop_key: str,
                           timeout: float = 60.0) -> Tuple[bool, bytes]:
        """``acc_apply`` and then a blocking ``qpop(pop_key)`` in ONE round trip -> (accepted, popped item)."""
        g = np.ascontiguousarray(g, dtype=np.float32)
        st, v, data = self._call(OP_ACC_APPLY, key + "\x01" + pop_key, flags=FLAG_THEN_POP, a0=local_step,
                                 a1=int(timeout * 1000), payload=g)
        self._ok(st, f"acc_apply_then_pop {key} -> {pop_key}")
        return bool(v & 1), data

    def acc_take_apply(self, key: str, var_key: str, num_required: int, lr: float, timeout: float = 60.0) -> int:
        """Wait for num_required gradients, apply their mean to var_key with rate lr; returns the count taken."""
        st, v, _ = self._call(OP_ACC_TAKE_APPLY, key, a0=num_required, a1=int(timeout * 1000), f0=lr,
                              payload=var_key.encode())
        self._ok(st, f"acc_take_apply {key} (have {v} of {num_required})")
        return v

    def acc_take_apply_release(self, key: str, var_key: str, num_required: int, lr: float, step_offset: int,
                               token_prefix: str, fanout: int, copies: int = 1, timeout: float = 60.0) -> int:
        """The chief's whole round in ONE request: wait for num_required gradients, apply their mean to var_key,
        bump the step counter var_key[step_offset] and deal ``copies`` tokens carrying the new step into the queues
        ``token_prefix/0 .. token_prefix/(fanout-1)``.  Returns the new step."""
        payload = (var_key.encode() + b"\0" + token_prefix.encode() + b"\0"
                   + struct.pack("<qqq", int(step_offset), int(fanout), int(copies)))
        st, v, _ = self._call(OP_ACC_TAKE_APPLY, key, flags=FLAG_RELEASE, a0=num_required, a1=int(timeout * 1000),
                              f0=lr, payload=payload)
        self._ok(st, f"acc_take_apply_release {key}")
        return v

    def acc_set_step(self, key: str, step: int):
        self._ok(self._call(OP_ACC_SET_STEP, key, a0=step)[0], f"acc_set_step {key}")

    def acc_info(self, key: str) -> Tuple[int, int, int]:
        st, _, data = self._call(OP_ACC_INFO, key)
        self._ok(st, f"acc_info {key}")
        return struct.unpack("<qqq", data)


class EventsWriter:
    """Append-only TensorBoard event file (TFRecord framing, masked CRC32C)."""

    def __init__(self, logdir: str, suffix: str = ""):
        Path(logdir).mkdir(parents=True, exist_ok=True)
        self._h = lib().dtf_events_open(str(logdir).encode(), suffix.encode())
        if not self._h:
            raise OSError(f"cannot open an event file under {logdir}")
        self.path = lib().dtf_events_path(self._h).decode()

    def write_scalars(self, step: int, tags: Sequence[str], values: Sequence[float], wall_time: Optional[float] = None):
        n = len(tags)
        ctags = (C.c_char_p * n)(*[t.encode() for t in tags])
        cvals = (C.c_float * n)(*[float(v) for v in values])
        st = lib().dtf_events_write_scalars(self._h, wall_time or time.time(), int(step), n, ctags, cvals)
        if st != OK:
            raise OSError("event write failed")

    def write_series(self, steps: np.ndarray, tags: Sequence[str], values: np.ndarray,
                     wall_time: Optional[float] = None):
        """One event per step; ``values`` is [len(steps), len(tags)] float32."""
        steps = np.ascontiguousarray(steps, dtype=np.int64)
        values = np.ascontiguousarray(values, dtype=np.float32)
        assert values.shape == (len(steps), len(tags))
        ctags = (C.c_char_p * len(tags))(*[t.encode() for t in tags])
        st = lib().dtf_events_write_scalar_series(
            self._h, wall_time or time.time(), len(steps), steps.ctypes.data_as(C.POINTER(C.c_int64)), len(tags),
            ctags, values.ctypes.data_as(C.POINTER(C.c_float)))
        if st != OK:
            raise OSError("event write failed")

    def write_raw(self, event_bytes: bytes):
        if lib().dtf_events_write_raw(self._h, event_bytes, len(event_bytes)) != OK:
            raise OSError("event write failed")

    def flush(self):
        lib().dtf_events_flush(self._h)

    def close(self):
        if self._h:
            lib().dtf_events_close(self._h)
            self._h = None


def write_bundle(prefix: str, tensors: List[Tuple[str, np.ndarray]]):
    """Write ``<prefix>.index`` and ``<prefix>.data-00000-of-00001`` (float32 tensors)."""
    h = lib().dtf_bundle_begin(str(prefix).encode())
    for name, arr in tensors:
        arr = np.asarray(arr, dtype=np.float32, order="C")     # keeps 0-d scalars 0-d
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        st = lib().dtf_bundle_add_f32(h, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim)
        if st != OK:
            raise OSError("bundle add failed")
    if lib().dtf_bundle_finish(h) != OK:
        raise OSError(f"cannot write checkpoint {prefix}")


class Board:
    """The synchronous barrier's control plane for tasks that share a node (``shm_board.cpp``): an arrival queue and a
    token queue per worker in POSIX shared memory, waited on with futexes.  The chief creates it under a unique name
    and publishes the name; every other task opens it.  Thread-safe."""

    def __init__(self, name: str, workers: int, create: bool):
        self.name, self.workers = name, int(workers)
        self._h = lib().dtf_board_open(name.encode(), int(workers), 1 if create else 0)
        if not self._h:
            raise RuntimeError_(f"cannot {'create' if create else 'open'} the barrier board {name!r} for {workers} workers")

    @staticmethod
    def unique_name(tag: str) -> str:
        import os
        import time as _time

        return f"/dtf_board_{tag}_{os.getpid()}_{int(_time.time() * 1e6) & 0xFFFFFFFF:x}"

    def arrive(self, worker: int, step: int, slot: int):
        if lib().dtf_board_arrive(self._h, int(worker), int(step), int(slot)) != 0:
            raise RuntimeError_("barrier board: the arrival queue is full (the aggregator is gone)")

    def take(self, want: int, timeout: float):
        """Up to ``want`` arrivals as (worker, step, slot) tuples; waits ``timeout`` s for the first (TimeoutError)."""
        buf = (C.c_int64 * (3 * int(want)))()
        got = C.c_int(0)
        rc = lib().dtf_board_take(self._h, int(want), max(0, int(timeout * 1000)), buf, C.byref(got))
        if rc == 1:
            raise StoreTimeout("no arrival")
        if rc != 0:
            raise RuntimeError_(f"barrier board: take failed ({rc})")
        return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(got.value)]

    def release(self, worker: int, step: int, copies: int = 1):
        """``copies`` tokens carrying ``step`` for ``worker`` (-1: for every worker)."""
        if copies > 0 and lib().dtf_board_release(self._h, int(worker), int(step), int(copies)) != 0:
            raise RuntimeError_("barrier board: a token queue is full (a replica is gone)")

    def wait_token(self, worker: int, timeout: float) -> int:
        step = C.c_int64(0)
        rc = lib().dtf_board_wait_token(self._h, int(worker), max(0, int(timeout * 1000)), C.byref(step))
        if rc == 1:
            raise StoreTimeout("no token")
        if rc != 0:
            raise RuntimeError_(f"barrier board: wait failed ({rc})")
        return int(step.value)

    def attached(self) -> int:
        return int(lib().dtf_board_attached(self._h))

    def native_entry_points(self):
        """(address of dtf_board_take, address of dtf_board_release, the board's handle): for code in another library
        that runs a whole round of the chief without the interpreter (``dtf_sync_round`` of the kernel library)."""
        l = lib()
        return (C.cast(l.dtf_board_take, C.c_void_p).value, C.cast(l.dtf_board_release, C.c_void_p).value, self._h)

    def close(self):
        if self._h:
            lib().dtf_board_close(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
