#!/usr/bin/env python
"""Single-process baseline (equivalent of the reference's tfsingle.py): no cluster, no
parameter server, the model pinned to /gpu:0 with soft placement.

    python tfsingle.py                 # one sess.run per batch, like the reference
    python tfsingle.py --resident      # device-resident data + hipGraph replay (fast tier)
    python tfsingle.py --resident --hidden 512,256 --activation relu --lr 0.05    # a deeper stack, dense-layer kernels
"""
import time

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.examples import mnist_mlp

mnist_mlp.define_hparam_flags()
FLAGS = tf.app.flags.FLAGS


def main(_):
    mnist = mnist_mlp.load_data(FLAGS)
    print("worker setting up ...", flush=True)
    begin_time = time.time()
    hidden = mnist_mlp.hidden_widths(FLAGS)
    if FLAGS.resident and hidden != (100,):
        # any other fully connected stack: variables in a session, steps replayed from hipGraphs of dense-layer launches
        import torch

        from distributed_tensorflow_amd.data import DeviceDataset
        from distributed_tensorflow_amd.parallel.dense_trainer import DenseResidentTrainer

        with tf.device("/gpu:0"):
            m = mnist_mlp.build_model(FLAGS.lr, FLAGS.activation, hidden=hidden)
            print("Ready to go")
        with tf.Session(config=mnist_mlp.make_config(FLAGS)) as sess:
            sess.run(m.init_op)
            ds = DeviceDataset(mnist.train.images, mnist.train.labels, sess.device, FLAGS.batch_size, seed=0)
            trainer = DenseResidentTrainer(sess, m.train_op, ds, FLAGS.batch_size, chunk=min(FLAGS.freq, 550))
            print(trainer.describe(), flush=True)
            writer = tf.summary.FileWriter(FLAGS.logs_path, graph=tf.get_default_graph()) if FLAGS.summaries else None
            mnist_mlp.run_resident_loop(trainer, mnist, FLAGS, writer, begin_time)
            trainer.close()
    elif FLAGS.resident:
        import torch

        from distributed_tensorflow_amd.data import DeviceDataset
        from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

        device = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
        ds = DeviceDataset(mnist.train.images, mnist.train.labels, device, FLAGS.batch_size, seed=0)
        trainer = build_replica_trainer(mode="single", device=device, dataset=ds, batch=FLAGS.batch_size,
                                        lr=FLAGS.lr, seed=1, chunk=min(FLAGS.freq, 550), act=FLAGS.activation)
        print("Ready to go")
        writer = tf.summary.FileWriter(FLAGS.logs_path) if FLAGS.summaries else None
        mnist_mlp.run_resident_loop(trainer, mnist, FLAGS, writer, begin_time)
    else:
        with tf.device("/gpu:0"):
            m = mnist_mlp.build_model(FLAGS.lr, FLAGS.activation, hidden=hidden)
            print("Ready to go")
        with tf.Session(config=mnist_mlp.make_config(FLAGS)) as sess:
            sess.run(m.init_op)
            writer = tf.summary.FileWriter(FLAGS.logs_path, graph=tf.get_default_graph()) if FLAGS.summaries else None
            mnist_mlp.run_session_loop(sess, m, mnist, FLAGS, writer, begin_time)
    if writer is not None:
        writer.close()
    print("Done")
    return 0


if __name__ == "__main__":
    tf.app.run(main)
