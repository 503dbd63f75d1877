"""``from distributed_tensorflow_amd.client import timeline`` mirrors ``tensorflow.python.client.timeline``:
``timeline.Timeline(run_metadata.step_stats).generate_chrome_trace_format()`` (the class lives in ``utils/trace.py``)."""
from ..utils import trace as timeline  # noqa: F401
