"""``train`` namespace: the ``tf.train.*`` names the reference uses, plus the device-resident loop."""
from ..cluster import ClusterSpec, Server  # noqa: F401
from ..optim import GradientDescentOptimizer, Optimizer, QueueRunner, SyncReplicasOptimizer  # noqa: F401
from ..placement import (GreedyLoadBalancingStrategy, RoundRobinStrategy,  # noqa: F401
                         replica_device_setter)
from .saver import Saver, latest_checkpoint, read_bundle  # noqa: F401
from .supervisor import Coordinator, Supervisor  # noqa: F401


def get_global_step(graph=None):
    from .. import graph as G

    g = graph or G.get_default_graph()
    for v in g.variables:
        if v.op_name == "global_step":
            return v
    return None


def get_or_create_global_step(graph=None):
    from .. import graph as G

    v = get_global_step(graph)
    if v is None:
        v = G.get_variable("global_step", [], initializer=G.constant_initializer(0), trainable=False)
    return v
