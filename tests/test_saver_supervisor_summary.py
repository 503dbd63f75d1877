"""Checkpoint layout, Supervisor lifecycle, event files."""
import os
import threading
import time

import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.examples.mnist_mlp import build_model
from distributed_tensorflow_amd.summary import read_events
from distributed_tensorflow_amd.train.saver import latest_checkpoint, read_bundle

from cluster_utils import cluster_dict

NAMES = ["global_step", "weights/Variable", "weights/Variable_1", "biases/Variable", "biases/Variable_1"]


def _batch(seed, n=20):
    rng = np.random.RandomState(seed)
    return rng.rand(n, 784).astype(np.float32), np.eye(10, dtype=np.float32)[rng.randint(0, 10, n)]


def test_saver_layout_roundtrip_and_retention(tmp_path):
    with tf.Graph().as_default():
        m = build_model(0.05)
        saver = tf.train.Saver(max_to_keep=2)
        with tf.Session() as sess:
            sess.run(m.init_op)
            x, y = _batch(0)
            paths = []
            for _ in range(3):
                sess.run(m.train_op, {m.x: x, m.y_: y})
                paths.append(saver.save(sess, str(tmp_path / "model.ckpt"), global_step=m.global_step))
            assert [os.path.basename(p) for p in paths] == ["model.ckpt-1", "model.ckpt-2", "model.ckpt-3"]
            files = sorted(os.listdir(tmp_path))
            assert files == ["checkpoint", "model.ckpt-2.data-00000-of-00001", "model.ckpt-2.index", "model.ckpt-2.meta",
                             "model.ckpt-3.data-00000-of-00001", "model.ckpt-3.index", "model.ckpt-3.meta"]
            state = (tmp_path / "checkpoint").read_text().splitlines()
            assert state[0] == 'model_checkpoint_path: "model.ckpt-3"'
            assert state[1:] == ['all_model_checkpoint_paths: "model.ckpt-2"', 'all_model_checkpoint_paths: "model.ckpt-3"']
            assert latest_checkpoint(str(tmp_path)) == str(tmp_path / "model.ckpt-3")
            bundle = read_bundle(paths[-1])
            assert sorted(bundle) == sorted(NAMES)
            assert bundle["weights/Variable"].shape == (784, 100) and float(bundle["global_step"]) == 3.0
            np.testing.assert_array_equal(bundle["weights/Variable_1"], sess.var_tensor(m.W2).numpy())
            want = {k: sess.var_tensor(getattr(m, k)).clone() for k in ("W1", "b1", "W2", "b2")}
    # restore into a fresh graph/session
    with tf.Graph().as_default():
        m2 = build_model(0.05)
        with tf.Session() as sess2:
            sess2.run(m2.init_op)
            tf.train.Saver().restore(sess2, paths[-1])
            for k in want:
                torch.testing.assert_close(sess2.var_tensor(getattr(m2, k)), want[k])
            assert sess2.run(m2.global_step) == 3.0
    # a corrupted data file is detected by the per-tensor CRC
    data = tmp_path / "model.ckpt-3.data-00000-of-00001"
    raw = bytearray(data.read_bytes())
    raw[100] ^= 0xFF
    data.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        read_bundle(paths[-1])


def test_supervisor_nonchief_waits_for_chief_then_sees_its_values():
    cluster = tf.train.ClusterSpec(cluster_dict(1, 2))
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
    ps = tf.train.Server(cluster, "ps", 0, config=cfg)
    ws = [tf.train.Server(cluster, "worker", i, config=cfg) for i in range(2)]
    got = {}
    try:
        def nonchief():
            with tf.Graph().as_default() as g:
                with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:1", cluster=cluster)):
                    m = build_model(0.05)
                sv = tf.train.Supervisor(graph=g, is_chief=False, global_step=m.global_step, init_op=m.init_op)
                t0 = time.time()
                sess = sv.prepare_or_wait_for_session(ws[1].target)
                got["waited"] = time.time() - t0
                got["W1"] = sess.var_tensor(m.W1).clone()
                sv.stop()

        t = threading.Thread(target=nonchief)
        t.start()
        time.sleep(0.6)                       # the chief shows up late
        assert "W1" not in got                # ... and the non-chief is still waiting
        with tf.Graph().as_default() as g:
            with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:0", cluster=cluster)):
                m = build_model(0.05)
            sv = tf.train.Supervisor(graph=g, is_chief=True, global_step=m.global_step, init_op=m.init_op)
            sess = sv.prepare_or_wait_for_session(ws[0].target)
            t.join(10)
            assert got["waited"] >= 0.5 and got["waited"] < 5      # event-driven, no 30 s poll
            torch.testing.assert_close(got["W1"], sess.var_tensor(m.W1))
            sv.stop()
    finally:
        for s in [ps] + ws:
            s.stop()


def test_supervisor_chief_only_checkpoint_and_resume(tmp_path):
    logdir = str(tmp_path / "train")
    x, y = _batch(4)

    def run(steps, expect_start):
        cluster = tf.train.ClusterSpec(cluster_dict(1, 1))
        cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
        ps = tf.train.Server(cluster, "ps", 0, config=cfg)
        w = tf.train.Server(cluster, "worker", 0, config=cfg)
        try:
            with tf.Graph().as_default() as g:
                with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:0", cluster=cluster)):
                    m = build_model(0.05)
                sv = tf.train.Supervisor(graph=g, is_chief=True, global_step=m.global_step, init_op=m.init_op,
                                         logdir=logdir, save_model_secs=3600)
                with sv.prepare_or_wait_for_session(w.target) as sess:
                    assert sess.store.global_step() == expect_start
                    for _ in range(steps):
                        sess.run(m.train_op, {m.x: x, m.y_: y})
                    W1 = sess.var_tensor(m.W1).clone()
                    sess._pull()
                    W1 = sess.var_tensor(m.W1).clone()
                sv.stop()                     # final checkpoint by the chief
                return W1
        finally:
            ps.stop()
            w.stop()

    W_a = run(3, 0)
    assert os.path.exists(os.path.join(logdir, "model.ckpt-3.index"))
    assert latest_checkpoint(logdir).endswith("model.ckpt-3")
    W_b = run(2, 3)                           # a restarted cluster resumes at step 3 with the saved weights
    assert os.path.exists(os.path.join(logdir, "model.ckpt-5.index"))
    assert not torch.equal(W_a, W_b)
    np.testing.assert_array_equal(read_bundle(os.path.join(logdir, "model.ckpt-3"))["weights/Variable"], W_a.numpy())


def test_event_file_contents(tmp_path):
    with tf.Graph().as_default() as g:
        m = build_model(0.05)
        with tf.Session() as sess:
            sess.run(m.init_op)
            w = tf.summary.FileWriter(str(tmp_path), graph=g)
            x, y = _batch(5)
            costs = []
            for _ in range(4):
                _, c, s, step = sess.run([m.train_op, m.cross_entropy, m.summary_op, m.global_step], {m.x: x, m.y_: y})
                w.add_summary(s, step)
                costs.append(c)
            w.add_scalar_series(np.arange(10, 13), ["cost", "accuracy"], np.arange(6, dtype=np.float32).reshape(3, 2))
            w.close()
    ev = read_events(w.path)
    assert os.path.basename(w.path).startswith("events.out.tfevents.")
    assert ev[0]["file_version"] == "brain.Event:2" and b"weights/Variable" in ev[1]["graph_def"]
    scal = [e for e in ev if e["scalars"]]
    assert [e["step"] for e in scal] == [0, 1, 2, 3, 10, 11, 12]
    np.testing.assert_allclose([e["scalars"]["cost"] for e in scal[:4]], costs, rtol=1e-6)
    assert set(scal[0]["scalars"]) == {"cost", "accuracy"} and scal[-1]["scalars"] == {"cost": 4.0, "accuracy": 5.0}


def _tf_event_classes():
    """``Event`` / ``Summary`` message classes built at run time from the field numbers of TensorFlow's
    ``core/util/event.proto`` and ``core/framework/summary.proto`` (the subset the reference's FileWriter emits):
    Google's protobuf runtime, not this package's own reader, decodes the records below."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="dtf_test_event.proto", package="dtftest", syntax="proto3")
    summary = fd.message_type.add(name="Summary")
    value = summary.nested_type.add(name="Value")
    value.field.add(name="tag", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    value.field.add(name="simple_value", number=2, type=F.TYPE_FLOAT, label=F.LABEL_OPTIONAL)
    value.field.add(name="node_name", number=7, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    summary.field.add(name="value", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED,
                      type_name=".dtftest.Summary.Value")
    event = fd.message_type.add(name="Event")
    event.field.add(name="wall_time", number=1, type=F.TYPE_DOUBLE, label=F.LABEL_OPTIONAL)
    event.field.add(name="step", number=2, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    event.field.add(name="file_version", number=3, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    event.field.add(name="graph_def", number=4, type=F.TYPE_BYTES, label=F.LABEL_OPTIONAL)
    event.field.add(name="summary", number=5, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name=".dtftest.Summary")
    # core/framework/graph.proto + node_def.proto, the fields the graph tab reads
    node = fd.message_type.add(name="NodeDef")
    node.field.add(name="name", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    node.field.add(name="op", number=2, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    node.field.add(name="input", number=3, type=F.TYPE_STRING, label=F.LABEL_REPEATED)
    node.field.add(name="device", number=4, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    graph = fd.message_type.add(name="GraphDef")
    graph.field.add(name="node", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".dtftest.NodeDef")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:      # older protobuf runtimes
        get = message_factory.MessageFactory(pool).GetPrototype
    event_cls = get(pool.FindMessageTypeByName("dtftest.Event"))
    event_cls.GraphDef = get(pool.FindMessageTypeByName("dtftest.GraphDef"))
    return event_cls


def _crc32c(data: bytes) -> int:
    """Castagnoli CRC, bit by bit (independent of the table-driven native one)."""
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def _masked(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_event_file_is_a_tfrecord_stream_of_event_protos_for_an_independent_decoder(tmp_path):
    """The event file the reference's ``tf.summary.FileWriter`` + ``add_summary`` leave behind (tfsingle.py:63,86) is a
    TFRecord stream (length, masked CRC-32C of the length, payload, masked CRC-32C of the payload) of ``Event``
    protocol buffers.  Here the framing is taken apart by hand with a bit-wise CRC, and the payloads are parsed by
    Google's protobuf runtime under TensorFlow's field numbers: what TensorBoard's reader does.  (No TensorFlow in this
    image: byte-for-byte parity with a TensorFlow-written file stays unpinned; the FORMAT is checked.)"""
    import struct

    Event = _tf_event_classes()
    with tf.Graph().as_default() as g:
        m = build_model(0.05)
        with tf.Session() as sess:
            sess.run(m.init_op)
            w = tf.summary.FileWriter(str(tmp_path), graph=g)
            x, y = _batch(5)
            costs = []
            for _ in range(3):
                _, c, s, step = sess.run([m.train_op, m.cross_entropy, m.summary_op, m.global_step], {m.x: x, m.y_: y})
                w.add_summary(s, step)
                costs.append(float(c))
            w.close()
    blob = open(w.path, "rb").read()
    events, pos = [], 0
    while pos < len(blob):
        (length,) = struct.unpack_from("<Q", blob, pos)
        (len_crc,) = struct.unpack_from("<I", blob, pos + 8)
        assert len_crc == _masked(_crc32c(blob[pos:pos + 8]))
        payload = blob[pos + 12:pos + 12 + length]
        (data_crc,) = struct.unpack_from("<I", blob, pos + 12 + length)
        assert len(payload) == length and data_crc == _masked(_crc32c(payload))
        ev = Event()
        ev.ParseFromString(payload)
        events.append(ev)
        pos += 12 + length + 4
    assert pos == len(blob)
    assert events[0].file_version == "brain.Event:2" and events[0].wall_time > 1e9
    graphs = [e.graph_def for e in events if len(e.graph_def)]
    assert len(graphs) == 1
    gd = Event.GraphDef()
    gd.ParseFromString(graphs[0])                      # what TensorBoard's graph tab reads (tfsingle.py:63)
    nodes = {n.name: n for n in gd.node}
    assert {"weights/Variable", "weights/Variable_1", "biases/Variable", "biases/Variable_1", "global_step",
            "input/x-input", "input/y-input"} <= set(nodes)
    assert nodes["input/x-input"].op == "Placeholder" and nodes["weights/Variable"].op.startswith("Variable")
    assert all(i.split(":")[0].lstrip("^") in nodes for n in gd.node for i in n.input)     # every edge has both ends
    scal = [e for e in events if len(e.summary.value)]
    assert [e.step for e in scal] == [0, 1, 2]
    for e, c in zip(scal, costs):
        got = {v.tag: v.simple_value for v in e.summary.value}
        assert set(got) == {"cost", "accuracy"} and abs(got["cost"] - c) <= 1e-6 * max(1.0, abs(c))


def _bundle_classes():
    """``BundleHeaderProto`` / ``BundleEntryProto`` built at run time from the field numbers of TensorFlow's
    ``core/protobuf/tensor_bundle.proto`` (+ ``TensorShapeProto``): decoded by Google's protobuf runtime."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="dtf_test_bundle.proto", package="dtfbundle", syntax="proto3")
    shape = fd.message_type.add(name="TensorShapeProto")
    dim = shape.nested_type.add(name="Dim")
    dim.field.add(name="size", number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    dim.field.add(name="name", number=2, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    shape.field.add(name="dim", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED,
                    type_name=".dtfbundle.TensorShapeProto.Dim")
    shape.field.add(name="unknown_rank", number=3, type=F.TYPE_BOOL, label=F.LABEL_OPTIONAL)
    header = fd.message_type.add(name="BundleHeaderProto")
    header.field.add(name="num_shards", number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    header.field.add(name="endianness", number=2, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)     # enum: LITTLE = 0
    entry = fd.message_type.add(name="BundleEntryProto")
    entry.field.add(name="dtype", number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)           # enum: DT_FLOAT = 1
    entry.field.add(name="shape", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL,
                    type_name=".dtfbundle.TensorShapeProto")
    entry.field.add(name="shard_id", number=3, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    entry.field.add(name="offset", number=4, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    entry.field.add(name="size", number=5, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    entry.field.add(name="crc32c", number=6, type=F.TYPE_FIXED32, label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return get(pool.FindMessageTypeByName("dtfbundle.BundleHeaderProto")), get(pool.FindMessageTypeByName("dtfbundle.BundleEntryProto"))


def _varint_at(buf: bytes, pos: int):
    val = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _table_block(buf: bytes, offset: int, size: int):
    """Entries of one block of a LevelDB-format table (the checkpoint's .index file): checks the block trailer
    (type byte 0 = uncompressed, masked CRC-32C over block + type) and undoes the key prefix compression."""
    import struct

    block, ctype = buf[offset:offset + size], buf[offset + size]
    (crc,) = struct.unpack_from("<I", buf, offset + size + 1)
    assert ctype == 0 and crc == _masked(_crc32c(block + bytes([ctype])))
    (restarts,) = struct.unpack_from("<I", block, len(block) - 4)
    end, pos, key, out = len(block) - 4 - 4 * restarts, 0, b"", []
    while pos < end:
        shared, pos = _varint_at(block, pos)
        fresh, pos = _varint_at(block, pos)
        vlen, pos = _varint_at(block, pos)
        key = key[:shared] + block[pos:pos + fresh]
        pos += fresh
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def test_checkpoint_is_a_tensor_bundle_for_an_independent_decoder(tmp_path):
    """What ``Supervisor(logdir=...)`` / ``Saver.save`` write is TensorFlow's tensor bundle: ``.index`` = a LevelDB-format
    table (footer with the magic number, index block, data blocks with prefix-compressed keys and CRC trailers) that maps
    "" to a ``BundleHeaderProto`` and every variable name to a ``BundleEntryProto``; ``.data-00000-of-00001`` = the raw
    little-endian tensors at the entries' offsets, each with its masked CRC-32C.  The table is taken apart by hand here
    and the protos are parsed by Google's protobuf runtime under TensorFlow's field numbers.  (No TensorFlow in this
    image: byte-for-byte parity with a TensorFlow-written checkpoint stays unpinned; the FORMAT is checked.)"""
    import struct

    Header, Entry = _bundle_classes()
    with tf.Graph().as_default():
        m = build_model(0.05)
        with tf.Session() as sess:
            sess.run(m.init_op)
            x, y = _batch(0)
            sess.run(m.train_op, {m.x: x, m.y_: y})
            prefix = tf.train.Saver().save(sess, str(tmp_path / "model.ckpt"), global_step=m.global_step)
            want = {v.op_name: sess.var_tensor(v).detach().cpu().numpy().copy() for v in sess.graph.variables}
    idx = open(prefix + ".index", "rb").read()
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    footer = idx[-48:]
    assert struct.unpack("<Q", footer[40:])[0] == 0xDB4775248B80FB57          # the table format's magic number
    _, pos = _varint_at(footer, 0)                                            # metaindex handle: offset, size
    _, pos = _varint_at(footer, pos)
    index_off, pos = _varint_at(footer, pos)
    index_size, pos = _varint_at(footer, pos)
    entries = {}
    for _, handle in _table_block(idx, index_off, index_size):                # index block: one handle per data block
        off, p = _varint_at(handle, 0)
        size, _ = _varint_at(handle, p)
        for key, value in _table_block(idx, off, size):
            assert key not in entries
            entries[key] = value
    assert list(entries) == sorted(entries)                                   # a table's keys are in byte order
    header = Header()
    header.ParseFromString(entries.pop(b""))
    assert header.num_shards == 1 and header.endianness == 0
    assert sorted(k.decode() for k in entries) == sorted(want)
    for key, blob in entries.items():
        e = Entry()
        e.ParseFromString(blob)
        name = key.decode()
        assert e.dtype == 1 and e.shard_id == 0                               # DT_FLOAT, the one data file
        assert [d.size for d in e.shape.dim] == list(want[name].shape)
        raw = data[e.offset:e.offset + e.size]
        assert len(raw) == e.size == want[name].size * 4 and e.crc32c == _masked(_crc32c(raw))
        np.testing.assert_array_equal(np.frombuffer(raw, dtype="<f4").reshape(want[name].shape), want[name])
