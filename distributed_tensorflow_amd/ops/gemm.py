"""MatMul of the graph executor on the hand-written fp32 MFMA GEMM (csrc/gemm.hip), differentiable.

``matmul(a, b, transpose_a, transpose_b)`` is what a ``tf.matmul`` node of an arbitrary graph evaluates to on a GPU
session (reference tfdist_between.py:56,58 are two such nodes): the forward product is one launch of the NN / NT / TN
kernel, and autograd's two gradient products are one launch each of the matching layouts — no library GEMM.  Inputs the
kernel does not cover (other dtypes, batched operands, CPU tensors) take ``torch.matmul``.
"""
from __future__ import annotations

import torch

from . import hip_backend as hb


def _rows(t: torch.Tensor) -> torch.Tensor:
    """A view or copy of ``t`` whose rows are contiguous (what the kernel's leading-dimension addressing needs)."""
    return t if (t.stride(1) == 1 and t.stride(0) >= t.shape[1]) else t.contiguous()


def supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    return (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2
            and b.dim() == 2 and a.numel() > 0 and b.numel() > 0 and hb.available())


class _MatMul(torch.autograd.Function):
    """``layout`` NN: a [M,K] b [K,N]; NT: a [M,K] b [N,K] (b transposed); TN: a [K,M] b [K,N] (a transposed)."""

    @staticmethod
    def forward(ctx, a, b, layout):
        a, b = _rows(a.detach()), _rows(b.detach())
        ctx.save_for_backward(a, b)
        ctx.layout = layout
        return hb.gemm(layout, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _rows(g)
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if ctx.layout == hb.GEMM_NN:      # c = a b
            da = hb.gemm(hb.GEMM_NT, g, b) if need_a else None       # g b^T
            db = hb.gemm(hb.GEMM_TN, a, g) if need_b else None       # a^T g
        elif ctx.layout == hb.GEMM_NT:    # c = a b^T
            da = hb.gemm(hb.GEMM_NN, g, b) if need_a else None       # g b
            db = hb.gemm(hb.GEMM_TN, g, a) if need_b else None       # g^T a
        else:                             # c = a^T b
            da = hb.gemm(hb.GEMM_NT, b, g) if need_a else None       # b g^T
            db = hb.gemm(hb.GEMM_NN, a, g) if need_b else None       # a g
        return da, db, None


def matmul(a: torch.Tensor, b: torch.Tensor, transpose_a: bool = False, transpose_b: bool = False) -> torch.Tensor:
    if not supported(a, b):
        return (a.t() if transpose_a else a) @ (b.t() if transpose_b else b)
    if transpose_a and transpose_b:
        a, transpose_a = a.t(), False        # (copied into row-major form by the kernel wrapper)
    layout = hb.GEMM_TN if transpose_a else (hb.GEMM_NT if transpose_b else hb.GEMM_NN)
    return _MatMul.apply(a, b, layout)
