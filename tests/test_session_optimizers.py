"""Session + optimizer semantics on CPU, single process, PS tasks on localhost ports."""
import threading

import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.examples.mnist_mlp import build_model
from distributed_tensorflow_amd.ops import reference as ref
from distributed_tensorflow_amd.session import match_accuracy, match_mlp

from cluster_utils import cluster_dict


def _batch(seed, n=20):
    rng = np.random.RandomState(seed)
    x = rng.rand(n, 784).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.randint(0, 10, n)]
    return x, y


def _params(sess, m):
    return {k: sess.var_tensor(getattr(m, k)).clone() for k in ("W1", "b1", "W2", "b2")}


def test_pattern_matcher_recognises_reference_graph():
    with tf.Graph().as_default():
        m = build_model(0.001)
        mm = match_mlp(m.cross_entropy)
        assert mm is not None and mm["act"] == "sigmoid" and mm["W1"] is m.W1 and mm["b2"] is m.b2
        assert match_accuracy(m.accuracy, mm)
        assert match_mlp(m.accuracy) is None
    with tf.Graph().as_default():
        m = build_model(0.001, "relu")
        assert match_mlp(m.cross_entropy)["act"] == "relu"


def test_fused_plan_equals_generic_autograd_and_manual_reference():
    x, y = _batch(0)
    with tf.Graph().as_default():
        m = build_model(0.1)
        with tf.Session() as sess:
            sess.run(m.init_op)
            p0 = _params(sess, m)
            for _ in range(3):
                _, cost, acc, step = sess.run([m.train_op, m.cross_entropy, m.accuracy, m.global_step],
                                              feed_dict={m.x: x, m.y_: y})
            assert sess.fused_steps == 3 and sess.generic_steps == 0 and step == 2.0
            # manual SGD with the closed-form gradients
            p = {k: v.clone() for k, v in p0.items()}
            for _ in range(3):
                out = ref.mlp_grads(torch.tensor(x), torch.tensor(y), p["W1"], p["b1"], p["W2"], p["b2"])
                for k in p:
                    p[k] -= 0.1 * out["d" + k]
            for k in p:
                torch.testing.assert_close(_params(sess, m)[k], p[k], rtol=1e-5, atol=1e-6)
            assert abs(cost - float(out["loss"])) < 1e-5 and abs(acc - float(out["acc"])) < 1e-6
            assert sess.run(m.global_step) == 3.0
    # a var_list that is not exactly the four MLP variables defeats the pattern -> op-by-op autograd
    with tf.Graph().as_default():
        m = build_model(0.1)
        part = tf.train.GradientDescentOptimizer(0.1).minimize(m.cross_entropy, var_list=[m.W2, m.b2])
        with tf.Session() as sess:
            sess.run(m.init_op)
            p0 = _params(sess, m)
            sess.run(part, feed_dict={m.x: x, m.y_: y})
            assert sess.generic_steps == 1 and sess.fused_steps == 0
            out = ref.mlp_grads(torch.tensor(x), torch.tensor(y), p0["W1"], p0["b1"], p0["W2"], p0["b2"])
            p1 = _params(sess, m)
            torch.testing.assert_close(p1["W2"], p0["W2"] - 0.1 * out["dW2"], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(p1["W1"], p0["W1"])          # frozen
    # generic path: gradients by autograd agree with the closed form
    with tf.Graph().as_default():
        m = build_model(0.1)
        gW2 = tf.train.GradientDescentOptimizer(0.1).compute_gradients(m.cross_entropy, [m.W2])[0][0]
        with tf.Session() as sess:
            sess.run(m.init_op)
            p = _params(sess, m)
            g = sess.run(gW2, feed_dict={m.x: x, m.y_: y})
            want = ref.mlp_grads(torch.tensor(x), torch.tensor(y), p["W1"], p["b1"], p["W2"], p["b2"])["dW2"]
            np.testing.assert_allclose(g, want.numpy(), rtol=1e-4, atol=1e-6)


def test_generic_train_path_for_non_mlp_model():
    with tf.Graph().as_default():
        gs = tf.train.get_or_create_global_step()
        x = tf.placeholder(tf.float32, [None, 3])
        t = tf.placeholder(tf.float32, [None, 1])
        w = tf.Variable(tf.zeros([3, 1]))
        loss = tf.reduce_mean(tf.square(tf.matmul(x, w) - t))
        train = tf.train.GradientDescentOptimizer(0.1).minimize(loss, global_step=gs)
        with tf.Session() as sess:
            sess.run(tf.global_variables_initializer())
            xs = np.random.RandomState(0).randn(64, 3).astype(np.float32)
            ts = xs @ np.array([[1.0], [-2.0], [0.5]], dtype=np.float32)
            first = sess.run(loss, {x: xs, t: ts})
            for _ in range(200):
                sess.run(train, {x: xs, t: ts})
            assert sess.run(loss, {x: xs, t: ts}) < 1e-3 * first and sess.generic_steps == 200
            assert sess.run(gs) == 200.0


def _worker_graph(cluster, task, lr, sync):
    g = tf.Graph()
    with g.as_default():
        with tf.device(tf.train.replica_device_setter(worker_device=f"/job:worker/task:{task}", cluster=cluster)):
            m = build_model(lr, sync_replicas=cluster.num_tasks("worker") if sync else 0)
    return g, m


@pytest.mark.parametrize("n_ps", [1, 2])
def test_async_ps_n_pushes_n_applies(n_ps):
    cd = cluster_dict(n_ps, 2)
    cluster = tf.train.ClusterSpec(cd)
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
    servers = [tf.train.Server(cluster, "ps", i, config=cfg) for i in range(n_ps)]
    workers = [tf.train.Server(cluster, "worker", i, config=cfg) for i in range(2)]
    try:
        sessions = []
        for i in range(2):
            g, m = _worker_graph(cluster, i, 0.05, sync=False)
            sv = tf.train.Supervisor(graph=g, is_chief=(i == 0), global_step=m.global_step, init_op=m.init_op)
            sessions.append((sv, sv.prepare_or_wait_for_session(workers[i].target), m))
        (sv0, s0, m0), (sv1, s1, m1) = sessions
        p0 = _params(s0, m0)
        torch.testing.assert_close(_params(s1, m1)["W1"], p0["W1"])   # non-chief sees the chief's init
        xa, ya = _batch(1)
        xb, yb = _batch(2)
        # worker0 pushes on the initial snapshot, then worker1 (which pulls the updated params), then worker0
        s0.run(m0.train_op, {m0.x: xa, m0.y_: ya})
        s1.run(m1.train_op, {m1.x: xb, m1.y_: yb})
        _, step = s0.run([m0.train_op, m0.global_step], {m0.x: xa, m0.y_: ya})
        assert step == 2.0 and s0.store.global_step() == 3      # N pushes -> N applies
        p = {k: v.clone() for k, v in p0.items()}
        for xx, yy in ((xa, ya), (xb, yb), (xa, ya)):
            out = ref.mlp_grads(torch.tensor(xx), torch.tensor(yy), p["W1"], p["b1"], p["W2"], p["b2"])
            for k in p:
                p[k] -= 0.05 * out["d" + k]
        s1._pull()
        for k in p:
            torch.testing.assert_close(_params(s1, m1)[k], p[k], rtol=1e-5, atol=1e-6)
        for sv, s, _ in sessions:
            s.close()
            sv.stop()
        assert all(srv.join(timeout=5) for srv in servers)       # released once both workers are done
    finally:
        for s in servers + workers:
            s.stop()


@pytest.mark.parametrize("n_ps", [1, 2])
def test_sync_replicas_one_mean_apply_per_n_pushes(n_ps):
    cd = cluster_dict(n_ps, 2)
    cluster = tf.train.ClusterSpec(cd)
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
    servers = [tf.train.Server(cluster, "ps", i, config=cfg) for i in range(n_ps)]
    workers = [tf.train.Server(cluster, "worker", i, config=cfg) for i in range(2)]
    steps, lr = 3, 0.05
    batches = [[_batch(10 + 2 * s + w) for s in range(steps)] for w in range(2)]
    seen, errors = {}, []
    try:
        sessions = []
        for i in range(2):
            g, m = _worker_graph(cluster, i, lr, sync=True)
            sv = tf.train.Supervisor(graph=g, is_chief=(i == 0), global_step=m.global_step, init_op=m.init_op)
            sess = sv.prepare_or_wait_for_session(workers[i].target)
            sessions.append((sv, sess, m))
        sv0, s0, m0 = sessions[0]
        p0 = _params(s0, m0)
        sv0.start_queue_runners(s0, [m0.sync_opt.get_chief_queue_runner()])

        def work(i):
            try:
                sv, sess, m = sessions[i]
                gs = []
                for s in range(steps):
                    xx, yy = batches[i][s]
                    _, step = sess.run([m.train_op, m.global_step], {m.x: xx, m.y_: yy})
                    gs.append(step)
                sess._pull()
                seen[i] = (gs, _params(sess, m))
            except BaseException as e:   # noqa: BLE001
                errors.append(e)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in ts]
        [t.join(60) for t in ts]
        assert not errors, errors
        assert s0.store.global_step() == steps                      # one update per N=2 worker batches
        assert seen[0][0] == [0.0, 1.0, 2.0] and seen[1][0] == [0.0, 1.0, 2.0]   # lock-step
        p = {k: v.clone() for k, v in p0.items()}
        for s in range(steps):
            gsum = {k: torch.zeros_like(v) for k, v in p.items()}
            for w in range(2):
                xx, yy = batches[w][s]
                out = ref.mlp_grads(torch.tensor(xx), torch.tensor(yy), p["W1"], p["b1"], p["W2"], p["b2"])
                for k in p:
                    gsum[k] += out["d" + k]
            for k in p:
                p[k] -= lr * gsum[k] / 2                            # the MEAN of the two gradients
        for w in range(2):
            for k in p:
                torch.testing.assert_close(seen[w][1][k], p[k], rtol=1e-5, atol=1e-6)
        sv0.request_stop()
        for sv, s, _ in reversed(sessions):     # the chief's stop() waits for the replicas still training
            s.close()
            sv.stop()
        assert m0.sync_opt.get_chief_queue_runner().rounds == steps
    finally:
        for s in servers + workers:
            s.stop()


def test_sync_barrier_times_out_with_clear_error_instead_of_hanging():
    cd = cluster_dict(1, 2)
    cluster = tf.train.ClusterSpec(cd)
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=300)
    ps = tf.train.Server(cluster, "ps", 0, config=cfg)
    w0 = tf.train.Server(cluster, "worker", 0, config=cfg)
    try:
        g, m = _worker_graph(cluster, 0, 0.05, sync=True)
        sv = tf.train.Supervisor(graph=g, is_chief=True, global_step=m.global_step, init_op=m.init_op)
        sess = sv.prepare_or_wait_for_session(w0.target)
        m.sync_opt.get_chief_queue_runner().drain_idle_secs = 0.2
        sv.start_queue_runners(sess, [m.sync_opt.get_chief_queue_runner()])
        xx, yy = _batch(3)
        with pytest.raises(TimeoutError, match="sync barrier"):    # the second replica never pushes
            sess.run(m.train_op, {m.x: xx, m.y_: yy})
        sv.stop()
    finally:
        ps.stop()
        w0.stop()


def test_sync_replicas_with_a_backup_worker_drops_the_stale_gradient():
    """replicas_to_aggregate=2 of total_num_replicas=3: the update uses the first two gradients,
    the third one (same local_step, arriving after the take) is refused as stale."""
    from distributed_tensorflow_amd.parallel.store import SHARD_ACC

    cluster = tf.train.ClusterSpec(cluster_dict(1, 3))
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
    ps = tf.train.Server(cluster, "ps", 0, config=cfg)
    ws = [tf.train.Server(cluster, "worker", i, config=cfg) for i in range(3)]
    try:
        sessions = []
        for i in range(3):
            g = tf.Graph()
            with g.as_default():
                with tf.device(tf.train.replica_device_setter(worker_device=f"/job:worker/task:{i}", cluster=cluster)):
                    m = build_model(0.05)
                    opt = tf.train.SyncReplicasOptimizer(tf.train.GradientDescentOptimizer(0.05), replicas_to_aggregate=2,
                                                         total_num_replicas=3, use_locking=True)
                    m.train_op = opt.minimize(m.cross_entropy, global_step=m.global_step)
                    m.sync_opt = opt
            sv = tf.train.Supervisor(graph=g, is_chief=(i == 0), global_step=m.global_step, init_op=m.init_op)
            sessions.append((sv, sv.prepare_or_wait_for_session(ws[i].target), m))
        sv0, s0, m0 = sessions[0]
        assert m0.sync_opt.tokens_per_step == 3
        sv0.start_queue_runners(s0, [m0.sync_opt.get_chief_queue_runner()])
        done, errors = [], []

        def work(i, delay):
            try:
                import time as _t
                _t.sleep(delay)
                sv, sess, m = sessions[i]
                xx, yy = _batch(20 + i)
                sess.run(m.train_op, {m.x: xx, m.y_: yy})
                done.append(i)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        ts = [threading.Thread(target=work, args=(i, 0.0 if i < 2 else 1.0)) for i in range(3)]
        [t.start() for t in ts]
        [t.join(60) for t in ts]
        assert not errors and sorted(done) == [0, 1, 2]
        assert s0.store.global_step() == 1                         # one aggregate of two gradients
        assert sessions[2][1].store.dropped == 1                   # the late third gradient was stale
        count, step, dropped = ws[0].client("ps", 0).acc_info(SHARD_ACC)
        assert (count, step, dropped) == (0, 1, 1)
        sv0.request_stop()
        for sv, s, _ in reversed(sessions):
            s.close()
            sv.stop()
    finally:
        for s in [ps] + ws:
            s.stop()


def test_sync_chief_keeps_aggregating_for_a_replica_that_has_batches_left():
    """The chief runs 2 steps and stops; the other replica has 4.  Its last two gradients are
    aggregated alone (rounds sized to the replicas still training) instead of hanging at the barrier."""
    cluster = tf.train.ClusterSpec(cluster_dict(1, 2))
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
    ps = tf.train.Server(cluster, "ps", 0, config=cfg)
    ws = [tf.train.Server(cluster, "worker", i, config=cfg) for i in range(2)]
    try:
        sessions = []
        for i in range(2):
            g, m = _worker_graph(cluster, i, 0.05, sync=True)
            sv = tf.train.Supervisor(graph=g, is_chief=(i == 0), global_step=m.global_step, init_op=m.init_op)
            sessions.append((sv, sv.prepare_or_wait_for_session(ws[i].target), m))
        sv0, s0, m0 = sessions[0]
        runner = m0.sync_opt.get_chief_queue_runner()
        sv0.start_queue_runners(s0, [runner])
        steps_seen, errors = {}, []

        def work(i, n):
            try:
                sv, sess, m = sessions[i]
                out = []
                for s in range(n):
                    xx, yy = _batch(40 + 2 * s + i)
                    out.append(sess.run([m.train_op, m.global_step], {m.x: xx, m.y_: yy})[1])
                steps_seen[i] = out
                if i == 0:
                    sv.request_stop()
                sv.stop()
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        ts = [threading.Thread(target=work, args=(0, 2)), threading.Thread(target=work, args=(1, 4))]
        [t.start() for t in ts]
        [t.join(60) for t in ts]
        assert not errors, errors
        assert steps_seen[0] == [0.0, 1.0] and steps_seen[1] == [0.0, 1.0, 2.0, 3.0]
        assert runner.rounds == 4 and not runner.exceptions
        assert ps.join(timeout=5.0) is True        # both workers done -> the ps was released
    finally:
        for s in [ps] + ws:
            s.stop()


def test_async_worker_runs_alone_when_its_peer_never_shows_up():
    """Async mode has no barrier: the chief trains to completion in a 2-worker cluster with one worker missing;
    the parameter server keeps serving (it is released only when ALL workers have reported done)."""
    cluster = tf.train.ClusterSpec(cluster_dict(1, 2))
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=5000)
    ps = tf.train.Server(cluster, "ps", 0, config=cfg)
    w0 = tf.train.Server(cluster, "worker", 0, config=cfg)
    try:
        g, m = _worker_graph(cluster, 0, 0.05, sync=False)
        sv = tf.train.Supervisor(graph=g, is_chief=True, global_step=m.global_step, init_op=m.init_op)
        with sv.prepare_or_wait_for_session(w0.target) as sess:
            xx, yy = _batch(7)
            for _ in range(5):
                sess.run(m.train_op, {m.x: xx, m.y_: yy})
            assert sess.store.global_step() == 5
        sv.stop()
        assert ps.join(timeout=0.3) is False
    finally:
        ps.stop()
        w0.stop()
