"""Scalar summaries and TensorBoard event files (reference tfdist_between.py:72-74, 84, 96).

``summary.scalar("cost", cross_entropy)``, ``summary.merge_all()`` and
``summary.FileWriter(logs_path, graph=...)`` / ``writer.add_summary(summary, step)``
with the reference's semantics: evaluating the merged op yields one ``Summary``
with a (tag, value) pair per scalar; every worker opens its own uniquely named
event file in the same directory.  Records are written by the native writer
(runtime/records.cpp) in TFRecord framing, readable by TensorBoard;
:func:`read_events` is a dependency-free reader used by the tests.
"""
from __future__ import annotations

import os
import struct
import time
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import graph as G
from .runtime import native as rt


class Summary:
    """Evaluated summary: list of (tag, simple_value)."""

    def __init__(self, values: Sequence[Tuple[str, float]]):
        self.values = list(values)

    def SerializeToString(self) -> bytes:  # noqa: N802  (protobuf naming)
        out = b""
        for tag, v in self.values:
            t = tag.encode()
            val = b"\x0a" + _varint(len(t)) + t + b"\x15" + struct.pack("<f", v)
            out += b"\x0a" + _varint(len(val)) + val
        return out

    def __repr__(self):
        return "Summary(%s)" % ", ".join(f"{t}={v:.6g}" for t, v in self.values)


def _varint(n: int) -> bytes:
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def scalar(name: str, tensor: G.Tensor, collections=None) -> G.Tensor:
    g = tensor.graph
    tag = g.scope_prefix() + name
    node = G.Tensor("scalar_summary", (tensor,), {"tag": tag}, name=name, dtype="string")
    g.add_to_collection(G.GraphKeys.SUMMARIES, node)
    return node


def merge(inputs: Sequence[G.Tensor], name=None) -> G.Tensor:
    return G.Tensor("merge_summary", tuple(inputs), {}, name=name, dtype="string")


def merge_all(key: str = G.GraphKeys.SUMMARIES) -> Optional[G.Tensor]:
    nodes = G.get_default_graph().get_collection(key)
    if not nodes:
        return None
    return merge(nodes)


def encode_graph_def(graph: G.Graph) -> bytes:
    """GraphDef{ repeated NodeDef node = 1 { name=1, op=2, input=3, device=4 } } — enough for the graph tab."""
    out = bytearray()
    for n in graph.nodes:
        nd = bytearray()
        for field, s in ((1, n.op_name), (2, G._default_op_name(n.op))):
            b = s.encode()
            nd += bytes([(field << 3) | 2]) + _varint(len(b)) + b
        for i in n.inputs:
            b = i.op_name.encode()
            nd += bytes([(3 << 3) | 2]) + _varint(len(b)) + b
        if n.device:
            b = n.device.encode()
            nd += bytes([(4 << 3) | 2]) + _varint(len(b)) + b
        out += b"\x0a" + _varint(len(nd)) + bytes(nd)
    return bytes(out)


class FileWriter:
    def __init__(self, logdir: str, graph: Optional[G.Graph] = None, max_queue: int = 10, flush_secs: int = 120,
                 filename_suffix: str = ""):
        self.logdir = logdir
        suffix = filename_suffix or f"{os.getpid()}"
        self._w = rt.EventsWriter(logdir, suffix)
        self._last_flush = time.time()
        self._flush_secs = flush_secs
        if graph is not None:
            self.add_graph(graph)

    @property
    def path(self) -> str:
        return self._w.path

    def get_logdir(self) -> str:
        return self.logdir

    def add_graph(self, graph: G.Graph, global_step=None):
        gd = graph.as_graph_def_bytes()
        ev = b"\x09" + struct.pack("<d", time.time()) + b"\x22" + _varint(len(gd)) + gd   # Event.graph_def = 4
        self._w.write_raw(ev)

    def add_summary(self, summary, global_step=None):
        if summary is None:
            return
        step = int(global_step) if global_step is not None else 0
        if isinstance(summary, Summary):
            self._w.write_scalars(step, [t for t, _ in summary.values], [v for _, v in summary.values])
        elif isinstance(summary, (bytes, bytearray)):
            ev = (b"\x09" + struct.pack("<d", time.time()) + b"\x10" + _varint(step) + b"\x2a"
                  + _varint(len(summary)) + bytes(summary))
            self._w.write_raw(ev)
        else:
            raise TypeError("add_summary expects a Summary or serialized bytes")
        if time.time() - self._last_flush > self._flush_secs:
            self.flush()

    def add_scalar_series(self, steps, tags: Sequence[str], values):
        """Bulk form used by the device-resident loops: one event per step from a [steps, tags] array."""
        self._w.write_series(np.asarray(steps), list(tags), np.asarray(values, dtype=np.float32))

    def flush(self):
        self._w.flush()
        self._last_flush = time.time()

    def close(self):
        self._w.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# -- dependency-free reader (tests, tools) ------------------------------------------------
def _read_varint(buf: bytes, pos: int):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _parse_fields(buf: bytes):
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported wire type")
        out.append((field, wire, v))
    return out


def read_records(path: str) -> Iterator[bytes]:
    """Yield TFRecord payloads, verifying both masked CRC32C checksums."""
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        head = data[pos:pos + 12]
        (n,) = struct.unpack("<Q", head[:8])
        (hc,) = struct.unpack("<I", head[8:12])
        if rt.crc32c_masked(head[:8]) != hc:
            raise ValueError("corrupt record length")
        payload = data[pos + 12:pos + 12 + n]
        (dc,) = struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])
        if rt.crc32c_masked(payload) != dc:
            raise ValueError("corrupt record payload")
        yield payload
        pos += 16 + n


def read_events(path: str) -> List[dict]:
    """Decode an event file into dicts: {wall_time, step, file_version?, scalars{tag: value}, graph_def?}."""
    events = []
    for rec in read_records(path):
        ev = {"step": 0, "scalars": {}}
        for field, wire, v in _parse_fields(rec):
            if field == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif field == 2:
                ev["step"] = v
            elif field == 3:
                ev["file_version"] = v.decode()
            elif field == 4:
                ev["graph_def"] = v
            elif field == 5:
                for f2, _, val in _parse_fields(v):
                    if f2 == 1:
                        tag, num = None, None
                        for f3, _, x in _parse_fields(val):
                            if f3 == 1:
                                tag = x.decode()
                            elif f3 == 2:
                                num = struct.unpack("<f", x)[0]
                        ev["scalars"][tag] = num
        events.append(ev)
    return events
