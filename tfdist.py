#!/usr/bin/env python
"""In-graph replication: a single client process with one tower per GPU and RCCL gradient
averaging.  (The reference's tfdist.py is an early between-graph script and its README
abandons in-graph replication; this script provides that mode.)

    python tfdist.py                    # one tower per visible GPU
    python tfdist.py --num_towers 4     # first four GPUs (or CPU towers when no GPU is visible)

Every tower trains on its own shuffled copy of the data with batch_size samples per step,
so one step consumes num_towers * batch_size samples and makes ONE averaged update.
"""
import time

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.examples import mnist_mlp

tf.app.flags.DEFINE_integer("num_towers", 0, "number of towers (0: one per visible GPU, at least 1)")
mnist_mlp.define_hparam_flags()
FLAGS = tf.app.flags.FLAGS


def main(_):
    import torch

    from distributed_tensorflow_amd.parallel.towers import InGraphTowers

    mnist = mnist_mlp.load_data(FLAGS)
    print("worker setting up ...", flush=True)
    n_gpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    n = FLAGS.num_towers or max(n_gpu, 1)
    devices = [f"cuda:{i % n_gpu}" for i in range(n)] if n_gpu else ["cpu"] * n
    begin_time = time.time()
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
    print("Done")
    return 0


if __name__ == "__main__":
    tf.app.run(main)
