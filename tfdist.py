#!/usr/bin/env python
"""In-graph replication: a single client process with one tower per GPU and gradient
averaging across the towers.  (The reference's tfdist.py is an early between-graph script and its README
abandons in-graph replication; this script provides that mode.)

    python tfdist.py                    # one tower per visible GPU
    python tfdist.py --num_towers 4     # first four GPUs (or CPU towers when no GPU is visible)
    python tfdist.py --exchange rccl    # RCCL all-reduce driven by a host loop (the library baseline)

``--exchange inbox`` (default on GPUs): every tower has its own stream and hipGraph chunks, and the gradient
mean is taken inside the towers' backward launches through peer access — one host call per tower per chunk.

Every tower trains on its own shuffled copy of the data with batch_size samples per step,
so one step consumes num_towers * batch_size samples and makes ONE averaged update.
"""
import os
import time

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.examples import mnist_mlp

tf.app.flags.DEFINE_integer("num_towers", 0, "number of towers (0: one per visible GPU, at least 1)")
tf.app.flags.DEFINE_string("exchange", "auto", "gradient mean across towers: inbox (inside the backward launches, "
                           "hipGraph chunks per tower), rccl (all-reduce, host loop), auto")
mnist_mlp.define_hparam_flags()
FLAGS = tf.app.flags.FLAGS


def main(_):
    import torch

    from distributed_tensorflow_amd.parallel.towers import DenseInGraphEngine, DenseInGraphTowers, InGraphEngine, InGraphTowers

    mnist = mnist_mlp.load_data(FLAGS)
    print("worker setting up ...", flush=True)
    n_gpu = torch.cuda.device_count()          # (counting the GPUs does not start the HIP runtime)
    n = FLAGS.num_towers or max(n_gpu, 1)
    # Towers that share a GPU must all be resident on it at once to swap gradients inside their launches, each on a
    # hardware queue of its own: at most two per GPU with the four queues a process gets by default, four per GPU
    # with sixteen (more towers' waiting workgroups no longer fit the chip together).  The runtime reads the number
    # when it starts, at the first GPU call: ask for it here.
    per_gpu = -(-n // n_gpu) if n_gpu else 0
    if 2 < per_gpu <= 4 and FLAGS.exchange != "rccl" and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 16:
        os.environ["GPU_MAX_HW_QUEUES"] = "16"
    per_gpu_limit = 4 if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 16 else 2
    if n_gpu and not torch.cuda.is_available():
        n_gpu = 0
    devices = [f"cuda:{i % n_gpu}" for i in range(n)] if n_gpu else ["cpu"] * n
    begin_time = time.time()
    exchange = FLAGS.exchange
    if exchange == "auto":
        crowded = n_gpu and per_gpu > per_gpu_limit
        exchange = "inbox" if (n_gpu and not crowded and FLAGS.activation == "sigmoid") else "rccl"
    hidden = mnist_mlp.hidden_widths(FLAGS)
    if hidden != (100,) and n_gpu and FLAGS.exchange != "rccl" and per_gpu <= per_gpu_limit:
        # any other stack on GPUs: a stream and hipGraph chunks per tower, the stack's gradient launches and one
        # exchange kernel per step (sum of the towers' gradient slabs through peer access + SGD)
        towers = DenseInGraphEngine(devices, mnist.train.images, mnist.train.labels, FLAGS.batch_size, FLAGS.lr,
                                    hidden=hidden, act=FLAGS.activation, seed=1, chunk=min(50, int(FLAGS.freq)))
    elif hidden != (100,):
        # CPU towers, or --exchange rccl: the host-loop towers around the dense-layer engine (all-reduce of the slabs)
        towers = DenseInGraphTowers(devices, mnist.train.images, mnist.train.labels, FLAGS.batch_size, FLAGS.lr,
                                    hidden=hidden, act=FLAGS.activation, seed=1)
    elif exchange == "inbox" and n_gpu:
        towers = InGraphEngine(devices, mnist.train.images, mnist.train.labels, FLAGS.batch_size, FLAGS.lr, seed=1,
                               chunk=min(50, int(FLAGS.freq)))
    else:
        towers = InGraphTowers(devices, mnist.train.images, mnist.train.labels, FLAGS.batch_size, FLAGS.lr,
                               seed=1, act=FLAGS.activation)
    for i, d in enumerate(devices):
        print(f"tower_{i}: /job:localhost/replica:0/task:0/{d.replace('cuda', 'gpu')}")
    print(towers.describe())
    print("Ready to go")
    writer = tf.summary.FileWriter(FLAGS.logs_path) if FLAGS.summaries else None
    mnist_mlp.run_resident_loop(towers, mnist, FLAGS, writer, begin_time)
    if writer is not None:
        writer.close()
    # towers apply the same mean gradient, so their replicas must be the same bits; an exchange that gave up waiting
    # (skipping its update from then on) or replicas that differ void the run instead of passing silently
    healthy = towers.healthy() if hasattr(towers, "healthy") else True
    if not healthy or not towers.replicas_identical():
        detail = towers.diagnostics() if hasattr(towers, "diagnostics") else {}
        raise RuntimeError(f"in-graph towers out of step (healthy={healthy}): {detail}")
    if hasattr(towers, "close"):
        towers.close()
    print("Done")
    return 0


if __name__ == "__main__":
    tf.app.run(main)
