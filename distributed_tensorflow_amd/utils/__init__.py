"""Cross-cutting helpers: host-side tracing (profiler ranges, per-run timelines)."""
from . import trace  # noqa: F401
from .trace import Collector, RunMetadata, RunOptions, Timeline, mark, span  # noqa: F401
