"""``from distributed_tensorflow_amd.client import timeline`` mirrors ``tensorflow.python.client.timeline``."""
from . import timeline  # noqa: F401
