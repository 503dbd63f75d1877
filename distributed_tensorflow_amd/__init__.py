"""MI355X-native parameter-server training framework.

``import distributed_tensorflow_amd as tf`` gives the TF-1.x surface the
reference scripts are written against (graph building, ``tf.train.*``,
``tf.app.flags``, ``tf.summary.*``, ``tf.Session``); the execution underneath is
PyTorch-ROCm tensors, hand-written gfx950 HIP kernels and RCCL / HIP-IPC /
native-TCP parameter-server transports.
"""
__version__ = "0.1.0"

from . import flags as _flags_mod
from . import summary, train  # noqa: F401
from .cluster import ClusterSpec, ConfigProto, GPUOptions, Server  # noqa: F401
from .graph import *  # noqa: F401,F403
from .graph import (Graph, GraphKeys, Tensor, Variable, get_default_graph, nn,  # noqa: F401
                    reset_default_graph)
from .session import InteractiveSession, Session, get_default_session  # noqa: F401
from .utils.trace import RunMetadata, RunOptions  # noqa: F401


class _App:
    flags = _flags_mod.flags
    run = staticmethod(_flags_mod.run)


app = _App()
flags = _flags_mod.flags


def is_gpu_available() -> bool:
    import torch

    return torch.cuda.is_available()
