"""Fully connected stacks of any depth: pattern matcher, engine (CPU oracle path) and session integration, incl. a
parameter-server run; the GPU kernels behind the same engine are covered by test_dense_stack_gpu.py."""
import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.models import DenseStackEngine
from distributed_tensorflow_amd.session import match_dense_stack, match_mlp

from cluster_utils import cluster_dict

DIMS = (20, 16, 12, 5)
ACTS = ("sigmoid", "relu")


def build_deep(lr, dims=DIMS, acts=ACTS, bias_last=True, xent_op=False):
    """The reference's graph (tfdist_between.py:37-70) with more layers."""
    g = type("M", (), {})()
    g.global_step = tf.get_variable("global_step", [], initializer=tf.constant_initializer(0), trainable=False)
    g.x = tf.placeholder(tf.float32, [None, dims[0]])
    g.y_ = tf.placeholder(tf.float32, [None, dims[-1]])
    tf.set_random_seed(1)
    g.W = [tf.Variable(tf.random_normal([dims[i], dims[i + 1]], stddev=0.5)) for i in range(len(dims) - 1)]
    g.b = [tf.Variable(tf.zeros([dims[i + 1]])) for i in range(len(dims) - 1)]
    h = g.x
    for i in range(len(dims) - 2):
        h = getattr(tf.nn, acts[i])(tf.add(tf.matmul(h, g.W[i]), g.b[i]))
    logits = tf.add(tf.matmul(h, g.W[-1]), g.b[-1]) if bias_last else tf.matmul(h, g.W[-1])
    if xent_op:
        g.loss = tf.reduce_mean(tf.nn.softmax_cross_entropy_with_logits(labels=g.y_, logits=logits))
        y = tf.nn.softmax(logits)
    else:
        y = tf.nn.softmax(logits)
        g.loss = tf.reduce_mean(-tf.reduce_sum(g.y_ * tf.log(y), reduction_indices=[1]))
    g.accuracy = tf.reduce_mean(tf.cast(tf.equal(tf.argmax(y, 1), tf.argmax(g.y_, 1)), tf.float32))
    g.train_op = tf.train.GradientDescentOptimizer(lr).minimize(g.loss, global_step=g.global_step)
    g.init_op = tf.global_variables_initializer()
    return g


def _batch(seed, n=30, dims=DIMS):
    rng = np.random.RandomState(seed)
    return rng.rand(n, dims[0]).astype(np.float32), np.eye(dims[-1], dtype=np.float32)[rng.randint(0, dims[-1], n)]


def _oracle_step(Ws, bs, acts, x, y, lr):
    """One SGD step by autograd in fp64; returns (loss, accuracy) before the update."""
    Wd = [w.double().requires_grad_(True) for w in Ws]
    bd = [b.double().requires_grad_(True) for b in bs]
    h = x.double()
    fns = {"sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh, "linear": lambda z: z}
    for i in range(len(Wd)):
        z = h @ Wd[i] + bd[i]
        h = fns[acts[i]](z) if i < len(Wd) - 1 else z
    loss = -(y.double() * torch.log_softmax(h, 1)).sum(1).mean()
    acc = (h.argmax(1) == y.argmax(1)).double().mean()
    loss.backward()
    with torch.no_grad():
        for w, wd in zip(Ws, Wd):
            w.copy_((wd - lr * wd.grad).float())
        for b, b_d in zip(bs, bd):
            b.copy_((b_d - lr * b_d.grad).float())
    return float(loss.detach()), float(acc)


def test_matcher_accepts_deep_stacks_and_rejects_others():
    with tf.Graph().as_default():
        m = build_deep(0.1)
        assert match_mlp(m.loss) is None            # three layers: not the reference's shape
        d = match_dense_stack(m.loss)
        assert d is not None and d["acts"] == list(ACTS) and [w for w, _ in d["layers"]] == m.W
        assert [b for _, b in d["layers"]] == m.b and d["x"] is m.x and d["y_"] is m.y_
        assert match_dense_stack(m.accuracy) is None
    with tf.Graph().as_default():
        m = build_deep(0.1, dims=(8, 6, 6, 6, 3), acts=("tanh", "relu", "sigmoid"), bias_last=False, xent_op=True)
        d = match_dense_stack(m.loss)
        assert d is not None and d["acts"] == ["tanh", "relu", "sigmoid"] and d["layers"][-1][1] is None
    with tf.Graph().as_default():                   # a layer that is not matmul + bias: no match
        x = tf.placeholder(tf.float32, [None, 4])
        y_ = tf.placeholder(tf.float32, [None, 3])
        w = tf.Variable(tf.zeros([4, 3]))
        loss = tf.reduce_mean(-tf.reduce_sum(y_ * tf.log(tf.nn.softmax(tf.square(tf.matmul(x, w)))), reduction_indices=[1]))
        assert match_dense_stack(loss) is None
    with tf.Graph().as_default():                   # one weight used by two layers: no match (in-place updates)
        x = tf.placeholder(tf.float32, [None, 4])
        y_ = tf.placeholder(tf.float32, [None, 4])
        w = tf.Variable(tf.zeros([4, 4]))
        logits = tf.matmul(tf.nn.relu(tf.matmul(x, w)), w)
        loss = tf.reduce_mean(tf.nn.softmax_cross_entropy_with_logits(labels=y_, logits=logits))
        assert match_dense_stack(loss) is None


@pytest.mark.parametrize("dims,acts", [((20, 16, 12, 5), ("sigmoid", "relu")), ((9, 7), ()),
                                       ((10, 8, 8, 8, 4), ("tanh", "linear", "sigmoid"))])
def test_engine_cpu_steps_match_autograd(dims, acts):
    gen = torch.Generator().manual_seed(3)
    Ws = [torch.randn(dims[i], dims[i + 1], generator=gen) * 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=gen) * 0.1 for i in range(len(dims) - 1)]
    x = torch.rand(3 * 25, dims[0], generator=gen)
    y = torch.eye(dims[-1])[torch.randint(0, dims[-1], (3 * 25,), generator=gen)]
    gs = torch.zeros(1)
    eng = DenseStackEngine([w.clone() for w in Ws], [b.clone() for b in bs], acts, batch=25, lr=0.2, global_step=gs)
    eng.bind_data(x, y)
    for step in range(4):
        eng.run_chunk(1)
        want = _oracle_step(Ws, bs, acts, x[(step % 3) * 25:(step % 3 + 1) * 25], y[(step % 3) * 25:(step % 3 + 1) * 25], 0.2)
        got = eng.last_stats()
        assert abs(got[0] - want[0]) < 1e-5 and abs(got[1] - want[1]) < 1e-6
    for w, we in zip(Ws, eng.W):
        torch.testing.assert_close(we, w, rtol=1e-5, atol=1e-6)
    for b, be in zip(bs, eng.b):
        torch.testing.assert_close(be, b, rtol=1e-5, atol=1e-6)
    assert float(gs) == 4.0
    loss, acc = eng.evaluate(x, y)
    assert np.isfinite(loss) and 0.0 <= acc <= 1.0


def test_engine_step_grad_matches_autograd_scaled():
    gen = torch.Generator().manual_seed(5)
    dims, acts = (12, 10, 6), ("sigmoid",)
    Ws = [torch.randn(dims[i], dims[i + 1], generator=gen) for i in range(2)]
    bs = [torch.zeros(dims[i + 1]) for i in range(2)]
    grads = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in zip(Ws, bs)]
    x, y = torch.rand(16, 12, generator=gen), torch.eye(6)[torch.randint(0, 6, (16,), generator=gen)]
    eng = DenseStackEngine(Ws, bs, acts, batch=16, lr=0.0, grads=grads)
    eng.bind_data(x, y)
    eng.step_grad(0, scale=0.5)
    W2 = [w.clone() for w in Ws]
    b2 = [b.clone() for b in bs]
    _oracle_step(W2, b2, acts, x, y, 1.0)           # W2 = W - grad
    for (gw, gb), w, w2, b, bb in zip(grads, Ws, W2, bs, b2):
        torch.testing.assert_close(gw, 0.5 * (w - w2), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(gb, 0.5 * (b - bb), rtol=1e-4, atol=1e-6)


def test_session_runs_deep_graph_on_the_engine_like_op_by_op_autograd(monkeypatch):
    x, y = _batch(0)

    def run(n):
        with tf.Graph().as_default():
            m = build_deep(0.1)
            with tf.Session() as sess:
                sess.run(m.init_op)
                out = [sess.run([m.train_op, m.loss, m.accuracy, m.global_step], feed_dict={m.x: x, m.y_: y})[1:]
                       for _ in range(n)]
                final = [sess.var_tensor(v).clone() for v in m.W + m.b]
                ev = sess.run([m.loss, m.accuracy], feed_dict={m.x: x, m.y_: y})
                return out, final, ev, (sess.dense_steps, sess.generic_steps, sess.fused_steps)

    out_d, fin_d, ev_d, counts_d = run(4)
    assert counts_d == (4, 0, 4)
    monkeypatch.setenv("DTF_DENSE_ENGINE", "0")
    out_g, fin_g, ev_g, counts_g = run(4)
    assert counts_g == (0, 4, 0)
    np.testing.assert_allclose(np.array(out_d, dtype=np.float64), np.array(out_g, dtype=np.float64), rtol=1e-5, atol=1e-6)
    for a, b in zip(fin_d, fin_g):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ev_d, ev_g, rtol=1e-5, atol=1e-6)
    assert [o[2] for o in out_d] == [0.0, 1.0, 2.0, 3.0]          # global_step fetched before the update


def test_deep_graph_against_parameter_servers_async():
    """Two PS shards, one worker: the engine writes gradients into the session's slab, the transport applies them."""
    cd = cluster_dict(2, 1)
    cluster = tf.train.ClusterSpec(cd)
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
    servers = [tf.train.Server(cluster, "ps", i, config=cfg) for i in range(2)]
    worker = tf.train.Server(cluster, "worker", 0, config=cfg)
    x, y = _batch(1)
    try:
        g = tf.Graph()
        with g.as_default():
            with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:0", cluster=cluster)):
                m = build_deep(0.1)
        sv = tf.train.Supervisor(graph=g, is_chief=True, global_step=m.global_step, init_op=m.init_op)
        sess = sv.prepare_or_wait_for_session(worker.target)
        Ws = [sess.var_tensor(v).clone() for v in m.W]
        bs = [sess.var_tensor(v).clone() for v in m.b]
        costs = [sess.run([m.train_op, m.loss], {m.x: x, m.y_: y})[1] for _ in range(3)]
        assert sess.dense_steps == 3 and sess.generic_steps == 0 and sess.store.global_step() == 3
        want = [_oracle_step(Ws, bs, ACTS, torch.tensor(x), torch.tensor(y), 0.1)[0] for _ in range(3)]
        np.testing.assert_allclose(costs, want, rtol=1e-5, atol=1e-6)
        sess._pull()
        for v, w in zip(m.W + m.b, Ws + bs):
            torch.testing.assert_close(sess.var_tensor(v), w, rtol=1e-5, atol=1e-6)
        sess.close()
        sv.stop()
    finally:
        for s in servers + [worker]:
            s.stop()


def test_resident_async_worker_of_a_deep_stack_against_two_ps_shards():
    """PsDenseResidentTrainer (here on the CPU: eager pull -> step_grad -> push per step) against the fp64 autograd
    oracle on the same batches; the trained values are on the parameter-server shards afterwards."""
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.examples.mnist_mlp import build_model
    from distributed_tensorflow_amd.parallel.dense_trainer import PsDenseResidentTrainer

    rng = np.random.RandomState(0)
    protos = rng.rand(10, 784).astype(np.float32)
    labels = rng.randint(0, 10, 600)
    images = (protos[labels] + 0.3 * rng.rand(600, 784)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]
    cd = cluster_dict(2, 1)
    cluster = tf.train.ClusterSpec(cd)
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
    servers = [tf.train.Server(cluster, "ps", i, config=cfg) for i in range(2)]
    worker = tf.train.Server(cluster, "worker", 0, config=cfg)
    try:
        g = tf.Graph()
        with g.as_default():
            with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:0", cluster=cluster)):
                m = build_model(0.05, "relu", hidden=(32, 16))
        sv = tf.train.Supervisor(graph=g, is_chief=True, global_step=m.global_step, init_op=m.init_op)
        sess = sv.prepare_or_wait_for_session(worker.target)
        start = [sess.var_tensor(v).clone() for v in m.weights + m.biases]
        ds = DeviceDataset(images, onehot, sess.device, 50, seed=3)
        fed = [(ds.x_epoch[i * 50:(i + 1) * 50].clone(), ds.y_epoch[i * 50:(i + 1) * 50].clone()) for i in range(7)]
        tr = PsDenseResidentTrainer(sess, m.train_op, ds, 50, chunk=3)
        assert "784-32-16-10" in tr.describe() and "eager" in tr.describe()
        tr.prepare(steps_hint=(7,))
        tr.run_steps(7)
        assert tr.global_step_value() == 7 and sess.store.global_step() == 7
        sess._pull()
        trained = [sess.var_tensor(v).clone() for v in m.weights + m.biases]
        loss_tr = tr.last_loss()
        tr.close()
        sess.close()
        sv.stop()
    finally:
        for s in servers + [worker]:
            s.stop()
    Ws, bs = [w.clone() for w in start[:3]], [b.clone() for b in start[3:]]
    for x, y in fed:
        want_loss = _oracle_step(Ws, bs, ("relu", "relu"), x, y, 0.05)[0]
    for got, want in zip(trained, Ws + bs):
        torch.testing.assert_close(got, want.view_as(got), rtol=1e-4, atol=1e-5)
    assert abs(loss_tr - want_loss) < 1e-4


def test_resident_trainer_steps_the_sessions_variables():
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.examples.mnist_mlp import build_model
    from distributed_tensorflow_amd.parallel.dense_trainer import DenseResidentTrainer

    rng = np.random.RandomState(0)
    protos = rng.rand(10, 784).astype(np.float32)
    labels = rng.randint(0, 10, 600)
    images = (protos[labels] + 0.3 * rng.rand(600, 784)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]
    with tf.Graph().as_default():
        m = build_model(0.05, "relu", hidden=(32, 16))
        assert len(m.weights) == 3 and m.weights[0].shape == (784, 32)
        with tf.Session() as sess:
            sess.run(m.init_op)
            w0 = sess.var_tensor(m.weights[0]).clone()
            ds = DeviceDataset(images, onehot, sess.device, 50, seed=3)
            tr = DenseResidentTrainer(sess, m.train_op, ds, 50, chunk=5)
            assert "784-32-16-10" in tr.describe() and tr.engine.batches_per_epoch == 12
            first = tr.evaluate(torch.as_tensor(images), torch.as_tensor(onehot))
            tr.run_steps(30)                    # two and a half epochs: crosses two reshuffles
            assert tr.global_step_value() == 30 and sess.run(m.global_step) == 30.0
            last = tr.evaluate(torch.as_tensor(images), torch.as_tensor(onehot))
            assert last[0] < first[0] and last[1] > first[1]
            assert not torch.equal(sess.var_tensor(m.weights[0]), w0)           # the session's own variables moved
            # the session evaluates the same numbers from the same variables
            loss, acc = sess.run([m.cross_entropy, m.accuracy], feed_dict={m.x: images, m.y_: onehot})
            assert abs(loss - last[0]) < 1e-5 and abs(acc - last[1]) < 1e-6
            # and a step through sess.run continues from where the trainer stopped
            _, step = sess.run([m.train_op, m.global_step], feed_dict={m.x: images[:50], m.y_: onehot[:50]})
            assert step == 30.0 and sess.dense_steps == 1
            tr.close()


def test_between_graph_script_with_resident_workers_of_a_deeper_stack(tmp_path):
    """tfdist_between.py --resident --hidden 48,24 on CPU tasks (tcp shards, eager steps): 1 ps + 2 workers."""
    import re

    from cluster_utils import launch_script

    args = ["--epochs=2", "--train_size=2200", "--freq=11", "--lr=0.05", "--resident", "--hidden=48,24",
            "--activation=relu", f"--logs_path={tmp_path}"]
    res = launch_script("tfdist_between.py", 1, 2, args, timeout=280)
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    finals = []
    for i in range(2):
        out = res[f"worker{i}"][1]
        assert "dense stack 784-48-24-10" in out and "eager" in out and out.rstrip().endswith("Done"), out[-2000:]
        finals.append([int(s) for s in re.findall(r"Step: (\d+),", out)][-1])
        costs = [float(c) for c in re.findall(r"Cost: ([\d.]+)", out)]
        assert costs[-1] < costs[0]
    assert 2 * 22 * 2 - 3 <= max(finals) <= 2 * 22 * 2         # both workers' pushes are updates of the shared step


def test_in_graph_towers_of_a_deeper_stack_apply_the_mean_gradient(tmp_path):
    """Two CPU towers: one step equals the fp64 autograd step on both towers' batches taken together; replicas stay
    identical across a reshuffle; and the script form, ``tfdist.py --hidden 48,24 --num_towers 2``."""
    import os
    import subprocess
    import sys

    from cluster_utils import ROOT
    from distributed_tensorflow_amd.parallel.towers import DenseInGraphTowers

    rng = np.random.RandomState(0)
    labels = rng.randint(0, 10, 400)
    images = (rng.rand(10, 784).astype(np.float32)[labels] + 0.3 * rng.rand(400, 784)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]
    tw = DenseInGraphTowers(["cpu", "cpu"], images, onehot, batch=100, lr=0.05, hidden=(32, 16), act="tanh", seed=1)
    assert "784-32-16-10" in tw.describe() and tw.batches_per_epoch == 4
    Ws = [w.clone() for w in tw.engines[0].W]
    bs = [b.clone() for b in tw.engines[0].b]
    x = torch.cat([ds.x_epoch[:100] for ds in tw.datasets])
    y = torch.cat([ds.y_epoch[:100] for ds in tw.datasets])
    tw.step()
    assert tw.replicas_identical() and tw.global_step_value() == 1
    _oracle_step(Ws, bs, ("tanh", "tanh"), x, y, 0.05)
    for got, want in zip(tw.engines[1].W + tw.engines[1].b, Ws + bs):
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    tw.run_steps(5)                       # crosses the 4-batch epoch boundary (reshuffle)
    assert tw.replicas_identical() and tw.global_step_value() == 6

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tfdist.py"), "--hidden", "48,24", "--activation", "relu",
                          "--num_towers", "2", "--lr", "0.05", "--epochs", "2", "--train_size", "2200", "--logs_path",
                          str(tmp_path)], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=300, env=dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert out.returncode == 0, out.stdout[-2000:]
    assert "2 towers of a dense stack 784-48-24-10" in out.stdout and out.stdout.rstrip().endswith("Done")
    assert "Step: 44,  Epoch:  2,  Batch:  22 of  22" in out.stdout
    accs = [float(line.split(":")[1]) for line in out.stdout.splitlines() if line.startswith("Test-Accuracy")]
    assert len(accs) == 2 and accs[-1] > 0.5


def test_deeper_stack_checkpoint_and_resume_through_the_supervisor(tmp_path):
    """A deeper stack trained by the device-resident PS worker loop, checkpointed by the chief's Supervisor at the end
    and resumed by a restarted cluster: every variable of the stack is in the bundle under its TF name."""
    import os

    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.examples.mnist_mlp import build_model
    from distributed_tensorflow_amd.parallel.dense_trainer import PsDenseResidentTrainer
    from distributed_tensorflow_amd.train.saver import latest_checkpoint, read_bundle

    logdir = str(tmp_path / "train")
    rng = np.random.RandomState(0)
    labels = rng.randint(0, 10, 300)
    images = (rng.rand(10, 784).astype(np.float32)[labels] + 0.3 * rng.rand(300, 784)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]

    def run(steps, expect_start):
        cluster = tf.train.ClusterSpec(cluster_dict(1, 1))
        cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
        ps = tf.train.Server(cluster, "ps", 0, config=cfg)
        w = tf.train.Server(cluster, "worker", 0, config=cfg)
        try:
            with tf.Graph().as_default() as g:
                with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:0", cluster=cluster)):
                    m = build_model(0.05, "relu", hidden=(32, 16))
                sv = tf.train.Supervisor(graph=g, is_chief=True, global_step=m.global_step, init_op=m.init_op,
                                         logdir=logdir, save_model_secs=3600)
                with sv.prepare_or_wait_for_session(w.target) as sess:
                    assert sess.store.global_step() == expect_start
                    ds = DeviceDataset(images, onehot, sess.device, 50, seed=3)
                    tr = PsDenseResidentTrainer(sess, m.train_op, ds, 50, chunk=4)
                    tr.run_steps(steps)
                    sess._pull()
                    values = {v.op_name: sess.var_tensor(v).clone() for v in m.weights + m.biases}
                    tr.close()
                sv.stop()                     # final checkpoint by the chief
                return values
        finally:
            ps.stop()
            w.stop()

    a = run(7, 0)
    assert latest_checkpoint(logdir).endswith("model.ckpt-7") and os.path.exists(os.path.join(logdir, "model.ckpt-7.index"))
    bundle = read_bundle(os.path.join(logdir, "model.ckpt-7"))
    assert len(a) == 6 and set(a) <= set(bundle) and "global_step" in bundle
    for name, value in a.items():
        np.testing.assert_array_equal(bundle[name].reshape(-1), value.numpy().reshape(-1))
    b = run(3, 7)                             # a restarted cluster resumes at step 7 with the saved variables
    assert latest_checkpoint(logdir).endswith("model.ckpt-10")
    assert any(not torch.equal(a[k], b[k]) for k in a)


def test_tfsingle_script_with_hidden_layers(tmp_path):
    import os
    import subprocess
    import sys

    from cluster_utils import ROOT

    for extra in (["--resident"], []):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tfsingle.py"), "--hidden", "48,24", "--activation", "relu",
                              "--lr", "0.05", "--epochs", "2", "--train_size", "2200", "--logs_path", str(tmp_path), *extra],
                             cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                             env=dict(os.environ, PYTHONPATH=ROOT))
        assert out.returncode == 0, out.stdout[-2000:]
        assert "Step: 44,  Epoch:  2,  Batch:  22 of  22" in out.stdout and out.stdout.rstrip().endswith("Done")
        accs = [float(line.split(":")[1]) for line in out.stdout.splitlines() if line.startswith("Test-Accuracy")]
        assert len(accs) == 2 and accs[-1] > 0.5


@pytest.mark.parametrize("bucket_mb", [32.0, 0.001])
def test_data_parallel_dense_stack_applies_the_mean_gradient(bucket_mb):
    """Two gloo ranks: bucketed all-reduce + SGD per bucket = one process applying the mean of both gradients."""
    import json
    import os
    import subprocess
    import sys

    from cluster_utils import ROOT, free_ports

    port = free_ports(1)[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dense_dp_worker.py"), str(bucket_mb)]
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="2"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert res["replicas_identical"] and res["global_step"] == 14
    assert res["one_step_error"] < 1e-5
    assert res["buckets"] == (1 if bucket_mb > 1 else 3)        # 1 KB buckets: one per layer
    assert res["acc"] > 0.5 and "all-reduce" in res["desc"]
