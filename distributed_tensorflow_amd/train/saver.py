"""``Saver``: checkpoints in TensorFlow's on-disk layout.

What ``Supervisor(logdir=...)`` would make the reference write:
``<logdir>/model.ckpt-<global_step>.index`` + ``.data-00000-of-00001`` (tensor
bundle), ``.meta`` and the text ``checkpoint`` state file, with the variables
stored under their graph names (``global_step``, ``weights/Variable``,
``weights/Variable_1``, ``biases/Variable``, ``biases/Variable_1``).  Only the
chief saves; the shards are gathered to the chief through a normal pull.

The bundle is written by the native writer (runtime/records.cpp).  Byte-level
interchange with TensorFlow's reader is not verified here (TensorFlow is not
installable in this environment) — "parity unpinned"; layout, naming, CRCs and
round-tripping through :func:`read_bundle` are tested, and the FORMAT is checked
by a decoder that shares no code with this package: the table of the ``.index``
file taken apart by hand, its ``BundleHeaderProto`` / ``BundleEntryProto`` values
parsed by Google's protobuf runtime under TensorFlow's field numbers
(tests/test_saver_supervisor_summary.py).
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..runtime import native as rt
from ..summary import _parse_fields, _read_varint, _varint
from ..utils.trace import mark, span

CHECKPOINT_STATE = "checkpoint"


def _read_block(data: bytes, offset: int, size: int) -> bytes:
    block = data[offset:offset + size]
    ctype = data[offset + size]
    (crc,) = struct.unpack("<I", data[offset + size + 1:offset + size + 5])
    if rt.crc32c_masked(block + bytes([ctype])) != crc:
        raise ValueError("checkpoint index: block checksum mismatch")
    if ctype != 0:
        raise ValueError("checkpoint index: compressed blocks are not supported")
    return block


def _block_entries(block: bytes):
    (n_restarts,) = struct.unpack("<I", block[-4:])
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_bundle(prefix: str) -> Dict[str, np.ndarray]:
    """Load every float32 tensor of ``<prefix>.index`` / ``.data-00000-of-00001``."""
    with open(prefix + ".index", "rb") as f:
        idx = f.read()
    with open(prefix + ".data-00000-of-00001", "rb") as f:
        data = f.read()
    footer = idx[-48:]
    if struct.unpack("<Q", footer[40:])[0] != 0xdb4775248b80fb57:
        raise ValueError("not a table file")
    pos = 0
    _, pos = _read_varint(footer, pos)
    _, pos = _read_varint(footer, pos)
    ioff, pos = _read_varint(footer, pos)
    isize, pos = _read_varint(footer, pos)
    out: Dict[str, np.ndarray] = {}
    for _, handle in _block_entries(_read_block(idx, ioff, isize)):
        boff, p = _read_varint(handle, 0)
        bsize, p = _read_varint(handle, p)
        for key, val in _block_entries(_read_block(idx, boff, bsize)):
            if key == b"":
                continue   # BundleHeaderProto
            shape, offset, size, crc = [], 0, 0, None
            for field, wire, v in _parse_fields(val):
                if field == 2:
                    for f2, _, dim in _parse_fields(v):
                        if f2 == 2:
                            d = 0
                            for f3, _, x in _parse_fields(dim):
                                if f3 == 1:
                                    d = x
                            shape.append(d)
                elif field == 4:
                    offset = v
                elif field == 5:
                    size = v
                elif field == 6:
                    (crc,) = struct.unpack("<I", v)
            raw = data[offset:offset + size]
            if crc is not None and rt.crc32c_masked(raw) != crc:
                raise ValueError(f"checkpoint tensor {key.decode()}: checksum mismatch")
            out[key.decode()] = np.frombuffer(raw, dtype=np.float32).reshape(shape).copy()
    return out


def latest_checkpoint(checkpoint_dir: str, latest_filename: str = CHECKPOINT_STATE) -> Optional[str]:
    path = os.path.join(checkpoint_dir, latest_filename)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for line in f:
            m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
            if m:
                p = m.group(1)
                if not os.path.isabs(p):
                    p = os.path.join(checkpoint_dir, p)
                return p if os.path.exists(p + ".index") else None
    return None


def update_checkpoint_state(save_dir: str, model_checkpoint_path: str, all_paths: Sequence[str],
                            latest_filename: str = CHECKPOINT_STATE):
    rel = lambda p: os.path.basename(p) if os.path.dirname(os.path.abspath(p)) == os.path.abspath(save_dir) else p  # noqa: E731
    tmp = os.path.join(save_dir, latest_filename + ".tmp")
    with open(tmp, "w") as f:
        f.write(f'model_checkpoint_path: "{rel(model_checkpoint_path)}"\n')
        for p in all_paths:
            f.write(f'all_model_checkpoint_paths: "{rel(p)}"\n')
    os.replace(tmp, os.path.join(save_dir, latest_filename))


import contextlib

_NO_LOCK = contextlib.nullcontext()


class Saver:
    def __init__(self, var_list=None, max_to_keep: int = 5, name: str = "save"):
        self.var_list = var_list
        self.max_to_keep = max_to_keep
        self.last_checkpoints: List[str] = []

    def _variables(self, sess):
        return list(self.var_list) if self.var_list is not None else list(sess.graph.variables)

    def save(self, sess, save_path: str, global_step=None, latest_filename: str = CHECKPOINT_STATE,
             write_meta_graph: bool = True) -> str:
        # (the session's lock: a save from the Supervisor's checkpoint thread must not interleave with a step)
        with getattr(sess, "run_lock", _NO_LOCK), span("checkpoint_save"):
            return self._save(sess, save_path, global_step, latest_filename, write_meta_graph)

    def _save(self, sess, save_path, global_step, latest_filename, write_meta_graph) -> str:
        if sess.store.is_remote:
            sess._pull()
        if global_step is not None:
            if not isinstance(global_step, (int, np.integer)):
                global_step = int(round(float(sess.var_tensor(global_step).item())))
            prefix = f"{save_path}-{int(global_step)}"
        else:
            prefix = save_path
        save_dir = os.path.dirname(os.path.abspath(prefix))
        os.makedirs(save_dir, exist_ok=True)
        tensors = [(v.op_name, sess.var_tensor(v).detach().cpu().numpy()) for v in self._variables(sess)]
        rt.write_bundle(prefix, tensors)
        if write_meta_graph:
            self._write_meta(sess, prefix)
        if prefix in self.last_checkpoints:
            self.last_checkpoints.remove(prefix)
        self.last_checkpoints.append(prefix)
        while self.max_to_keep and len(self.last_checkpoints) > self.max_to_keep:
            old = self.last_checkpoints.pop(0)
            for ext in (".index", ".data-00000-of-00001", ".meta"):
                try:
                    os.remove(old + ext)
                except OSError:
                    pass
        update_checkpoint_state(save_dir, prefix, self.last_checkpoints, latest_filename)
        return prefix

    @staticmethod
    def _write_meta(sess, prefix: str):
        """MetaGraphDef{ graph_def = 2, saver_def = 3 { filename_tensor_name=1, save=2, restore=3, version=7 } }."""
        gd = sess.graph.as_graph_def_bytes()

        def s(field, text):
            b = text.encode()
            return bytes([(field << 3) | 2]) + _varint(len(b)) + b

        saver = s(1, "save/Const:0") + s(2, "save/control_dependency:0") + s(3, "save/restore_all") + b"\x38\x02"
        meta = b"\x12" + _varint(len(gd)) + gd + b"\x1a" + _varint(len(saver)) + saver
        with open(prefix + ".meta.tmp", "wb") as f:
            f.write(meta)
        os.replace(prefix + ".meta.tmp", prefix + ".meta")

    def restore(self, sess, save_path: str):
        with getattr(sess, "run_lock", _NO_LOCK):
            self._restore(sess, save_path)

    def _restore(self, sess, save_path: str):
        mark("checkpoint_restore", path=save_path)
        values = read_bundle(save_path)
        for v in self._variables(sess):
            if v.op_name not in values:
                raise KeyError(f"variable {v.op_name} not found in checkpoint {save_path}")
            arr = values[v.op_name]
            if tuple(arr.shape) != tuple(v.shape):
                raise ValueError(f"shape mismatch for {v.op_name}: checkpoint {arr.shape} vs graph {v.shape}")
            sess.var_tensor(v).copy_(torch.from_numpy(arr).reshape(v.shape))
        sess.store.write_all(sess.theta)

    def recover_last_checkpoints(self, paths: Sequence[str]):
        self.last_checkpoints = [p for p in paths if os.path.exists(p + ".index")]
