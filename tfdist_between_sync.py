#!/usr/bin/env python
"""Between-graph SYNCHRONOUS parameter-server training (equivalent of the reference's
tfdist_between_sync.py): SyncReplicasOptimizer with replicas_to_aggregate =
total_num_replicas = number of workers, the chief (worker 0) running the aggregation
queue runner.  One averaged update per N worker batches; launch exactly like
tfdist_between.py.
"""
import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.examples import mnist_mlp
from settings import ps_svrs, worker_svrs

tf.app.flags.DEFINE_string("job_name", "", "ps or worker")
tf.app.flags.DEFINE_integer("task_index", 0, "Index of task within the job")
mnist_mlp.define_hparam_flags()
FLAGS = tf.app.flags.FLAGS


def main(_):
    cluster = tf.train.ClusterSpec({"ps": ps_svrs, "worker": worker_svrs})
    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index,
                             config=mnist_mlp.make_config(FLAGS))
    if FLAGS.job_name == "ps":
        print("ps setting up ...", flush=True)
        server.join()
    elif FLAGS.job_name == "worker":
        print("worker setting up ...", flush=True)
        mnist_mlp.run_worker(server, cluster, FLAGS, sync=True)
    server.stop()
    return 0


if __name__ == "__main__":
    tf.app.run(main)
