"""``timeline.Timeline(run_metadata.step_stats).generate_chrome_trace_format()``."""
from ..utils.trace import Timeline  # noqa: F401
