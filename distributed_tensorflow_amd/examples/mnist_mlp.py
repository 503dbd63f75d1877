"""Shared body of the five training scripts at the repository root.

Builds the reference's graph (names, creation order and placement as in
tfdist_between.py:37-75) through the framework's TF-style API and runs the
reference's training loop with its exact stdout contract
(tfdist_between.py:86-112)::

    Step: %d,  Epoch: %2d,  Batch: %3d of %3d,  Cost: %.4f,  AvgTime: %3.2fms
    Test-Accuracy: %2.2f
    Total Time: %3.2fs
    Final Cost: %.4f

Two execution tiers produce those lines:

* ``run_session_loop``  — one ``sess.run`` per batch with a NumPy feed, a cost
  fetch and an event-file write per step, like the reference;
* ``run_resident_loop`` — the device-resident tier: the training split lives in
  HBM, ``freq`` steps are replayed from one hipGraph, the host reads the cost
  only when a progress line is due, and the per-step (cost, accuracy) summaries
  are flushed to the event file in bulk.  Same model, same updates, same log.
"""
from __future__ import annotations

import os
import time
from types import SimpleNamespace
from typing import Optional

import numpy as np

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.data import DeviceDataset, read_data_sets


def define_hparam_flags():
    f = tf.app.flags
    f.DEFINE_integer("batch_size", 100, "mini-batch size per worker")
    f.DEFINE_float("lr", 0.001, "SGD learning rate")
    f.DEFINE_integer("epochs", 100, "passes over the training split")
    f.DEFINE_string("logs_path", "./logs", "directory of the TensorBoard event files")
    f.DEFINE_integer("freq", 100, "progress line every this many batches")
    f.DEFINE_string("data_dir", "MNIST_data", "idx files if present, else synthetic MNIST-shaped data")
    f.DEFINE_integer("train_size", 55000, "rows of the training split (synthetic data only)")
    f.DEFINE_string("logdir", "", "checkpoint directory (empty: no checkpoints, like the reference)")
    f.DEFINE_float("save_model_secs", 600, "checkpoint period of the chief")
    f.DEFINE_string("activation", "sigmoid", "hidden activation: sigmoid (reference), relu or tanh")
    f.DEFINE_string("hidden", "100", "widths of the hidden layers, comma-separated: 100 is the reference's network, "
                                     "e.g. 512,256 a deeper stack on the dense-layer kernels")
    f.DEFINE_boolean("resident", False, "device-resident hipGraph loop instead of one sess.run per batch.  Synchronous "
                     "replicas in this tier keep one replica per worker and sum the workers' gradients inside their "
                     "kernels (with backup workers, --replicas_to_aggregate below the number of workers: the first "
                     "gradients to arrive, decided inside the exchange kernel)")
    f.DEFINE_integer("replicas_to_aggregate", 0, "synchronous scripts: gradients per update (0: the number of workers, "
                     "as the reference; fewer: the slowest workers' gradients of a step are dropped -- backup workers)")
    f.DEFINE_boolean("summaries", True, "write cost/accuracy summaries every step")
    f.DEFINE_integer("max_steps", 0, "stop after this many batches in total (0: run all epochs)")
    f.DEFINE_string("transport", "tcp", "parameter-server transport: tcp | ipc")


def load_data(flags):
    sizes = (int(flags.train_size), 5000, 10000) if int(flags.train_size) == 55000 else \
        (int(flags.train_size), max(100, int(flags.train_size) // 11), max(1000, int(flags.train_size) // 5))
    return read_data_sets(flags.data_dir, one_hot=True, sizes=sizes)


def hidden_widths(flags) -> tuple:
    return tuple(int(v) for v in str(getattr(flags, "hidden", "100")).split(",") if v.strip())


def build_model(lr: float, activation: str = "sigmoid", sync_replicas: int = 0, hidden=(100,), aggregate: int = 0):
    """The reference graph; call inside the ``tf.device(...)`` scope that places it.  ``hidden``: widths of the hidden
    layers — (100,) is the reference's network (tfdist_between.py:48-59), more entries give the same graph with more
    fully connected layers (variables still created weights first, then biases, as the reference does)."""
    hidden = tuple(int(h) for h in hidden)
    if len(hidden) != 1:
        return _build_deep_model(lr, activation, sync_replicas, hidden, aggregate)
    global_step = tf.get_variable("global_step", [], initializer=tf.constant_initializer(0), trainable=False)
    with tf.name_scope("input"):
        x = tf.placeholder(tf.float32, shape=[None, 784], name="x-input")
        y_ = tf.placeholder(tf.float32, shape=[None, 10], name="y-input")
    tf.set_random_seed(1)
    with tf.name_scope("weights"):
        W1 = tf.Variable(tf.random_normal([784, hidden[0]]))
        W2 = tf.Variable(tf.random_normal([hidden[0], 10]))
    with tf.name_scope("biases"):
        b1 = tf.Variable(tf.zeros([hidden[0]]))
        b2 = tf.Variable(tf.zeros([10]))
    act = {"relu": tf.nn.relu, "tanh": tf.nn.tanh}.get(activation, tf.nn.sigmoid)
    with tf.name_scope("softmax"):
        a2 = act(tf.add(tf.matmul(x, W1), b1))
        y = tf.nn.softmax(tf.add(tf.matmul(a2, W2), b2))
    with tf.name_scope("cross_entropy"):
        cross_entropy = tf.reduce_mean(-tf.reduce_sum(y_ * tf.log(y), reduction_indices=[1]))
    with tf.name_scope("train"):
        opt = tf.train.GradientDescentOptimizer(lr)
        sync_opt = None
        if sync_replicas:
            sync_opt = tf.train.SyncReplicasOptimizer(opt, replicas_to_aggregate=aggregate or sync_replicas,
                                                      total_num_replicas=sync_replicas, use_locking=True)
            train_op = sync_opt.minimize(cross_entropy, global_step=global_step)
        else:
            train_op = opt.minimize(cross_entropy, global_step=global_step)
    with tf.name_scope("Accuracy"):
        correct = tf.equal(tf.argmax(y, 1), tf.argmax(y_, 1))
        accuracy = tf.reduce_mean(tf.cast(correct, tf.float32))
    tf.summary.scalar("cost", cross_entropy)
    tf.summary.scalar("accuracy", accuracy)
    return SimpleNamespace(global_step=global_step, x=x, y_=y_, W1=W1, W2=W2, b1=b1, b2=b2, y=y,
                           cross_entropy=cross_entropy, train_op=train_op, accuracy=accuracy,
                           summary_op=tf.summary.merge_all(), init_op=tf.global_variables_initializer(),
                           sync_opt=sync_opt)


def _build_deep_model(lr: float, activation: str, sync_replicas: int, hidden, aggregate: int = 0):
    """The reference graph with ``len(hidden)`` hidden layers (none: softmax regression).  Weights are drawn with
    standard deviation 1 / sqrt(fan-in): the reference's N(0, 1) saturates every unit of a second hidden layer."""
    dims = (784,) + tuple(hidden) + (10,)
    global_step = tf.get_variable("global_step", [], initializer=tf.constant_initializer(0), trainable=False)
    with tf.name_scope("input"):
        x = tf.placeholder(tf.float32, shape=[None, 784], name="x-input")
        y_ = tf.placeholder(tf.float32, shape=[None, 10], name="y-input")
    tf.set_random_seed(1)
    with tf.name_scope("weights"):
        weights = [tf.Variable(tf.random_normal([dims[i], dims[i + 1]], stddev=dims[i] ** -0.5))
                   for i in range(len(dims) - 1)]
    with tf.name_scope("biases"):
        biases = [tf.Variable(tf.zeros([dims[i + 1]])) for i in range(len(dims) - 1)]
    act = {"relu": tf.nn.relu, "tanh": tf.nn.tanh}.get(activation, tf.nn.sigmoid)
    with tf.name_scope("softmax"):
        h = x
        for W, b in zip(weights[:-1], biases[:-1]):
            h = act(tf.add(tf.matmul(h, W), b))
        y = tf.nn.softmax(tf.add(tf.matmul(h, weights[-1]), biases[-1]))
    with tf.name_scope("cross_entropy"):
        cross_entropy = tf.reduce_mean(-tf.reduce_sum(y_ * tf.log(y), reduction_indices=[1]))
    with tf.name_scope("train"):
        opt = tf.train.GradientDescentOptimizer(lr)
        sync_opt = None
        if sync_replicas:
            sync_opt = tf.train.SyncReplicasOptimizer(opt, replicas_to_aggregate=aggregate or sync_replicas,
                                                      total_num_replicas=sync_replicas, use_locking=True)
            train_op = sync_opt.minimize(cross_entropy, global_step=global_step)
        else:
            train_op = opt.minimize(cross_entropy, global_step=global_step)
    with tf.name_scope("Accuracy"):
        correct = tf.equal(tf.argmax(y, 1), tf.argmax(y_, 1))
        accuracy = tf.reduce_mean(tf.cast(correct, tf.float32))
    tf.summary.scalar("cost", cross_entropy)
    tf.summary.scalar("accuracy", accuracy)
    return SimpleNamespace(global_step=global_step, x=x, y_=y_, weights=weights, biases=biases, y=y,
                           cross_entropy=cross_entropy, train_op=train_op, accuracy=accuracy,
                           summary_op=tf.summary.merge_all(), init_op=tf.global_variables_initializer(),
                           sync_opt=sync_opt)


def progress_line(step, epoch, i, batch_count, cost, elapsed, freq) -> str:
    # elapsed is always divided by freq, also for the shorter last window (tfdist_between.py:100-105)
    return ("Step: %d,  Epoch: %2d,  Batch: %3d of %3d,  Cost: %.4f,  AvgTime: %3.2fms"
            % (step + 1, epoch + 1, i + 1, batch_count, cost, float(elapsed * 1000 / freq)))


def run_session_loop(sess, m, mnist, flags, writer=None, begin_time: Optional[float] = None) -> dict:
    """One ``sess.run`` per batch — the reference's loop."""
    begin_time = begin_time or time.time()
    freq, batch_size = int(flags.freq), int(flags.batch_size)
    start_time = time.time()
    total, cost, acc, epoch_times = 0, float("nan"), float("nan"), []
    fetches = [m.train_op, m.cross_entropy, m.summary_op if flags.summaries else None, m.global_step]
    for epoch in range(int(flags.epochs)):
        batch_count = int(mnist.train.num_examples / batch_size)
        count = 0
        for i in range(batch_count):
            batch_x, batch_y = mnist.train.next_batch(batch_size)
            _, cost, summary, step = sess.run(fetches, feed_dict={m.x: batch_x, m.y_: batch_y})
            if writer is not None and summary is not None:
                writer.add_summary(summary, step)
            count += 1
            total += 1
            if count % freq == 0 or i + 1 == batch_count:
                elapsed = time.time() - start_time
                start_time = time.time()
                print(progress_line(step, epoch, i, batch_count, cost, elapsed, freq), flush=True)
                count = 0
            if flags.max_steps and total >= int(flags.max_steps):
                break
        acc = sess.run(m.accuracy, feed_dict={m.x: mnist.test.images, m.y_: mnist.test.labels})
        print("Test-Accuracy: %2.2f" % acc)
        epoch_times.append(time.time() - begin_time)
        print("Total Time: %3.2fs" % float(epoch_times[-1]))
        begin_time = time.time()
        print("Final Cost: %.4f" % cost, flush=True)
        if flags.max_steps and total >= int(flags.max_steps):
            break
    return _report({"steps": total, "final_cost": float(cost), "test_accuracy": float(acc), "epoch_times": epoch_times})


def _report(result: dict) -> dict:
    """DTF_EPOCH_TIMES_FILE=path: the loop's result (the epoch times as floats; stdout rounds them to 10 ms, which is
    a sixth of an epoch of this tier) is also written there as JSON, for harnesses that time the scripts."""
    path = os.environ.get("DTF_EPOCH_TIMES_FILE")
    if path:
        import json

        with open(path, "w") as f:
            json.dump(result, f)
    return result


def run_resident_loop(trainer, mnist, flags, writer=None, begin_time: Optional[float] = None,
                      step_of=lambda trainer: trainer.global_step_value()) -> dict:
    """Device-resident tier driven by a ``parallel.replicas`` trainer; same stdout contract."""
    import torch

    begin_time = begin_time or time.time()
    freq, batch_size = int(flags.freq), int(flags.batch_size)
    dev = trainer.device
    test_x = torch.as_tensor(mnist.test.images).to(dev)
    test_y = torch.as_tensor(mnist.test.labels).to(dev)
    batch_count = trainer.engine.batches_per_epoch
    start_time = time.time()
    total, cost, acc, epoch_times = 0, float("nan"), float("nan"), []
    want_history = writer is not None and bool(flags.summaries)
    pipelined = hasattr(trainer, "fetch_window_async")
    state = {"start": start_time, "cost": cost}

    def emit(pending):
        """Print (and log) a finished window; with a pipelined trainer the NEXT window is already running."""
        handle, epoch_, last_i, first_, k_ = pending
        if pipelined:
            cost_, losses, accs, gstep = handle.result()
            step = int(round(gstep)) - 1         # value a fetch in the last run of the window would have seen
        else:
            cost_, step, losses, accs = handle
        if want_history:
            steps = np.arange(step - k_ + 1, step + 1)
            writer.add_scalar_series(steps, ["cost", "accuracy"], np.stack([losses.numpy(), accs.numpy()], 1))
        now = time.time()
        print(progress_line(step, epoch_, last_i, batch_count, cost_, now - state["start"], freq), flush=True)
        state["start"], state["cost"] = now, cost_

    for epoch in range(int(flags.epochs)):
        i = 0
        pending = None
        while i < batch_count:
            k = min(freq, batch_count - i)
            if flags.max_steps:
                k = min(k, int(flags.max_steps) - total)
            first = trainer.engine._host_cursor
            trainer.run_steps(k)
            i += k
            total += k
            if pipelined:
                # the window's cost, step counter and history are copied behind it; they are read (one wait per
                # window, as the reference's per-step fetch implies) only after the next window has been launched
                handle = trainer.fetch_window_async(first, k, want_history)
            else:
                cost_now = trainer.last_loss()  # the only host sync of the window
                losses, accs = trainer.engine.history(first, k) if want_history else (None, None)
                handle = (cost_now, step_of(trainer) - 1, losses, accs)
            if pending is not None:
                emit(pending)
            pending = (handle, epoch, i - 1, first, k)
            if flags.max_steps and total >= int(flags.max_steps):
                break
        if pending is not None:
            emit(pending)
        cost = state["cost"]
        _, acc = trainer.evaluate(test_x, test_y)
        print("Test-Accuracy: %2.2f" % acc)
        epoch_times.append(time.time() - begin_time)
        print("Total Time: %3.2fs" % float(epoch_times[-1]))
        begin_time = time.time()
        print("Final Cost: %.4f" % cost, flush=True)
        if flags.max_steps and total >= int(flags.max_steps):
            break
    return _report({"steps": total, "final_cost": float(cost), "test_accuracy": float(acc), "epoch_times": epoch_times})


def make_config(flags) -> "tf.ConfigProto":
    config = tf.ConfigProto(allow_soft_placement=True, log_device_placement=True)
    config.gpu_options.allow_growth = True
    config.transport = getattr(flags, "transport", "tcp")
    return config


def run_worker(server, cluster, flags, sync: bool) -> dict:
    """Between-graph worker: own graph, own session, variables on the PS shards."""
    mnist = load_data(flags)
    n_workers = cluster.num_tasks("worker")
    with tf.device(tf.train.replica_device_setter(
            worker_device="/job:worker/task:%d/gpu:%d" % (flags.task_index, flags.task_index), cluster=cluster)):
        m = build_model(flags.lr, flags.activation, sync_replicas=n_workers if sync else 0, hidden=hidden_widths(flags),
                        aggregate=int(flags.replicas_to_aggregate) if sync else 0)
        print("Ready to go")
    is_chief = flags.task_index == 0
    sv = tf.train.Supervisor(is_chief=is_chief, global_step=m.global_step, init_op=m.init_op,
                             logdir=flags.logdir or None, save_model_secs=flags.save_model_secs)
    begin_time = time.time()
    chief_queue_runner = m.sync_opt.get_chief_queue_runner() if sync else None
    with sv.prepare_or_wait_for_session(server.target) as sess:
        if sync and is_chief:
            sv.start_queue_runners(sess, [chief_queue_runner])
        writer = tf.summary.FileWriter(flags.logs_path, graph=tf.get_default_graph()) if flags.summaries else None
        if flags.resident and hidden_widths(flags) != (100,) and sync and sess.device.type != "cuda":
            print("--resident with synchronous replicas of a deeper stack needs GPU workers; "
                  "this stack runs one sess.run per batch", flush=True)
            result = run_session_loop(sess, m, mnist, flags, writer, begin_time)
        elif flags.resident and hidden_widths(flags) != (100,) and sync:
            from ..data import DeviceDataset
            from ..parallel.dense_trainer import PsDenseSyncResidentTrainer

            ds = DeviceDataset(mnist.train.images, mnist.train.labels, sess.device, flags.batch_size,
                               seed=1000 + flags.task_index)
            trainer = PsDenseSyncResidentTrainer(sess, m.train_op, ds, flags.batch_size, chunk=min(50, int(flags.freq)))
            trainer.prepare(steps_hint=(ds.batches_per_epoch,))
            print(trainer.describe(), flush=True)
            result = run_resident_loop(trainer, mnist, flags, writer, begin_time)
            if not trainer.healthy():
                raise RuntimeError("a replica never delivered its gradients: the synchronous run is void")
            if is_chief:
                trainer.publish()           # the PS shards hold the trained variables again
            trainer.close()
        elif flags.resident and hidden_widths(flags) != (100,):
            from ..data import DeviceDataset
            from ..parallel.dense_trainer import PsDenseResidentTrainer

            ds = DeviceDataset(mnist.train.images, mnist.train.labels, sess.device, flags.batch_size,
                               seed=1000 + flags.task_index)
            trainer = PsDenseResidentTrainer(sess, m.train_op, ds, flags.batch_size, chunk=min(50, int(flags.freq)))
            trainer.prepare(steps_hint=(ds.batches_per_epoch,))
            print(trainer.describe(), flush=True)
            result = run_resident_loop(trainer, mnist, flags, writer, begin_time)
            trainer.close()
        elif flags.resident and sync:
            from ..data import DeviceDataset
            from ..parallel.ps_resident import PsSyncResidentTrainer

            ds = DeviceDataset(mnist.train.images, mnist.train.labels, sess.device, flags.batch_size,
                               seed=1000 + flags.task_index)
            trainer = PsSyncResidentTrainer(sess, m.train_op, ds, flags.batch_size, chunk=min(50, int(flags.freq)))
            trainer.prepare(steps_hint=(ds.batches_per_epoch,))
            print(trainer.describe(), flush=True)
            result = run_resident_loop(trainer, mnist, flags, writer, begin_time)
            if not trainer.healthy():
                raise RuntimeError("a replica never delivered its gradients: the synchronous run is void")
            if is_chief:
                trainer.publish()           # the PS shards hold the trained variables again
            trainer.close()
        elif flags.resident and not sync:
            from ..data import DeviceDataset
            from ..parallel.ps_resident import PsResidentTrainer

            ds = DeviceDataset(mnist.train.images, mnist.train.labels, sess.device, flags.batch_size,
                               seed=1000 + flags.task_index)
            # DTF_FUSED_PUSH=1 / 0 forces the push into the backward launch / into a launch per shard;
            # by default the worker measures both against the live shards and keeps the faster
            forced = os.environ.get("DTF_FUSED_PUSH", "auto")
            trainer = PsResidentTrainer(sess, m.train_op, ds, flags.batch_size, chunk=min(50, int(flags.freq)),
                                        fused_push="auto" if forced == "auto" else forced != "0")
            trainer.prepare(steps_hint=(ds.batches_per_epoch,))
            print(trainer.describe(), flush=True)
            result = run_resident_loop(trainer, mnist, flags, writer, begin_time)
        else:
            result = run_session_loop(sess, m, mnist, flags, writer, begin_time)
        if writer is not None:
            writer.close()
        if sync and is_chief:
            sv.request_stop()
        result["global_step"] = sess.store.global_step()
        result["fused_steps"] = sess.fused_steps
        if sess.fused_steps or sess.generic_steps:
            print(f"[session] per-batch steps: {sess.fused_steps} on the fused kernels ({sess.pushed_steps} of them pushed "
                  f"their update from the backward launch), {sess.generic_steps} generic", flush=True)
    sv.stop()
    print("Done")
    return result
