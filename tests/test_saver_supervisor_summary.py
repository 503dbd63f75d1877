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
