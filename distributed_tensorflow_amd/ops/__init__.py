"""Fused ops: gfx950 HIP kernels (``hip_backend``) and their PyTorch fp32 oracle (``reference``)."""
from . import hip_backend, reference  # noqa: F401
from .reference import ACT_RELU, ACT_SIGMOID, act_code  # noqa: F401
