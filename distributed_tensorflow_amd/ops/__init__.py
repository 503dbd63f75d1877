"""Fused ops: gfx950 HIP kernels (``hip_backend``), their PyTorch fp32 oracle (``reference``) and the differentiable
MatMul on the hand-written GEMM (``gemm``)."""
from . import gemm, hip_backend, reference  # noqa: F401
from .reference import ACT_RELU, ACT_SIGMOID, act_code  # noqa: F401
