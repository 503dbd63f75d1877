"""ctypes binding of ``_lib/libdtf_hip.so`` (the gfx950 kernels in ``csrc/``).

The library is built in-tree by :mod:`distributed_tensorflow_amd.build` and is
loaded lazily.  On a machine with a GPU a missing library is a hard error —
there is no silent PyTorch fallback for CUDA tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch

_LIB_PATH = Path(os.environ.get("DTF_HIP_LIB") or Path(__file__).resolve().parent.parent / "_lib" / "libdtf_hip.so")   # DTF_HIP_LIB: a variant build (tuning experiments)
_lib: Optional[C.CDLL] = None
_load_error: Optional[str] = None

ERR_NAMES = {1: "bad shape/alignment", 2: "null pointer", 3: "launch failed", 4: "unsupported configuration"}


class HipKernelError(RuntimeError):
    pass


class DtfMlpDesc(C.Structure):
    """Field-for-field mirror of ``struct DtfMlpDesc`` in csrc/dtf_api.h."""

    _fields_ = [
        ("in_dim", C.c_int32),
        ("hidden", C.c_int32),
        ("classes", C.c_int32),
        ("batch", C.c_int32),
        ("act", C.c_int32),
        ("ksplit", C.c_int32),
        ("batches_per_epoch", C.c_int32),
        ("hist_cap", C.c_int32),
        ("x", C.c_void_p),
        ("y", C.c_void_p),
        ("W1", C.c_void_p),
        ("b1", C.c_void_p),
        ("W2", C.c_void_p),
        ("b2", C.c_void_p),
        ("gW1", C.c_void_p),
        ("gb1", C.c_void_p),
        ("gW2", C.c_void_p),
        ("gb2", C.c_void_p),
        ("z2", C.c_void_p),
        ("dz2", C.c_void_p),
        ("cursor", C.c_void_p),
        ("lr", C.c_void_p),
        ("loss_hist", C.c_void_p),
        ("acc_hist", C.c_void_p),
        ("global_step", C.c_void_p),
        ("shadow", C.c_void_p),
        ("zws", C.c_void_p),
    ]


class DtfPipeDesc(C.Structure):
    """Mirror of ``struct DtfPipeDesc`` (state of the one-launch-per-step chunk)."""

    _fields_ = [
        ("tW1", C.c_void_p),
        ("tb1", C.c_void_p),
        ("tW2", C.c_void_p),
        ("tb2", C.c_void_p),
        ("a2b", C.c_void_p),
        ("zws3", C.c_void_p),
        ("gram", C.c_void_p),
        ("gram_ld", C.c_int32),
        ("reserved", C.c_int32),
        ("w1t", C.c_void_p),
    ]


class DtfXgDesc(C.Structure):
    """Mirror of ``struct DtfXgDesc`` (the replicas' inboxes of the cross-replica two-launch step)."""

    _fields_ = [
        ("inbox", C.c_uint64 * 16),
        ("state", C.c_void_p),
        ("n", C.c_int64),
        ("offW1", C.c_int64),
        ("offb1", C.c_int64),
        ("offW2", C.c_int64),
        ("offb2", C.c_int64),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("timeout_ms", C.c_int32),
        ("compact", C.c_int32),
        ("step_index", C.c_int64),
        ("shard_start", C.c_int64 * 17),
        ("w1t", C.c_void_p),
    ]


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float
    lib.dtf_mlp_chunk_pipelined.argtypes = [C.POINTER(DtfMlpDesc), C.POINTER(DtfPipeDesc), i32, i32, vp]
    lib.dtf_mlp_chunk_pipelined_ex.argtypes = [C.POINTER(DtfMlpDesc), C.POINTER(DtfPipeDesc), i32, i32, i32, vp]
    lib.dtf_mlp_chunk_pipelined_supported.argtypes = [i32, i32, i32, i32]
    lib.dtf_mlp_step_fused.argtypes = [C.POINTER(DtfMlpDesc), i32, i32, vp]
    lib.dtf_mlp_step_grad.argtypes = [C.POINTER(DtfMlpDesc), i32, i32, f32, vp]
    lib.dtf_advance_cursor.argtypes = [vp, i32, vp]
    lib.dtf_mlp_eval.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dtf_linear_bias_act.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.dtf_dw_gemm.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp, vp]
    lib.dtf_sgd_apply.argtypes = [vp, vp, i64, vp, f32, i32, vp, vp]
    lib.dtf_accumulate.argtypes = [vp, vp, i64, vp]
    lib.dtf_sgd_apply_atomic.argtypes = [vp, vp, i64, vp, f32, vp, vp]
    lib.dtf_sgd_apply_atomic.restype = C.c_int
    lib.dtf_sum_apply.argtypes = [vp, vp, i32, i64, vp, f32, vp, vp]
    lib.dtf_sum_apply.restype = C.c_int
    lib.dtf_philox_normal.argtypes = [vp, i64, u64, u64, f32, f32, vp]
    lib.dtf_gather_rows.argtypes = [vp, vp, vp, i64, i32, i64, vp, vp]
    lib.dtf_hip_build_info.restype = C.c_char_p
    lib.dtf_gather_gram_supported.argtypes = [i32, i32]
    lib.dtf_gather_gram_supported.restype = C.c_int
    lib.dtf_gather_gram.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.dtf_gather_gram.restype = C.c_int
    lib.dtf_xgmi_max_ranks.argtypes = []
    lib.dtf_xgmi_alloc.argtypes = [C.POINTER(C.c_void_p), i64, i32]
    lib.dtf_xgmi_free.argtypes = [vp]
    lib.dtf_xgmi_export.argtypes = [vp, C.c_char_p]
    lib.dtf_xgmi_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.dtf_xgmi_close.argtypes = [vp]
    lib.dtf_xgmi_allreduce_apply.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, vp]
    lib.dtf_xgmi_allreduce_apply_subset.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.dtf_mlp_step_xg.argtypes = [C.POINTER(DtfMlpDesc), i32, i32, f32, C.POINTER(DtfXgDesc), vp]
    lib.dtf_mlp_step_xg.restype = C.c_int
    lib.dtf_mlp_step_async.argtypes = [C.POINTER(DtfMlpDesc), i32, i32, C.POINTER(DtfXgDesc), vp]
    lib.dtf_mlp_step_async.restype = C.c_int
    lib.dtf_mlp_step_grad_shards.argtypes = [C.POINTER(DtfMlpDesc), i32, i32, f32, C.POINTER(DtfXgDesc), vp]
    lib.dtf_mlp_step_grad_shards.restype = C.c_int
    lib.dtf_enable_peer_access.argtypes = [i32, i32]
    lib.dtf_enable_peer_access.restype = C.c_int
    lib.dtf_xgmi_pull.argtypes = [vp, i32, i64, vp, vp]
    lib.dtf_xgmi_push_sgd.argtypes = [vp, i32, i64, vp, vp, i64, vp]
    lib.dtf_xgmi_pull_ranges.argtypes = [vp, vp, i32, vp, vp]
    lib.dtf_xgmi_fill.argtypes = [vp, i32, i64]
    lib.dtf_transpose.argtypes = [vp, vp, i32, i32, vp]
    lib.dtf_transpose.restype = C.c_int
    for name in ("dtf_xgmi_pull", "dtf_xgmi_push_sgd", "dtf_xgmi_pull_ranges", "dtf_xgmi_fill",
                 "dtf_xgmi_max_ranks", "dtf_xgmi_alloc", "dtf_xgmi_free",
                 "dtf_xgmi_export", "dtf_xgmi_open", "dtf_xgmi_close", "dtf_xgmi_allreduce_apply",
                 "dtf_xgmi_allreduce_apply_subset"):
        getattr(lib, name).restype = C.c_int
    lib.dtf_gemm_f32.argtypes = [i32, i32, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, vp, vp, i64, vp, f32, i32, vp]
    lib.dtf_gemm_f32_ws.argtypes = [i32, i32, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, vp, vp, i64, vp, f32, i32, vp,
                                    i64, i32, vp]
    lib.dtf_gemm_f32_ws.restype = C.c_int
    lib.dtf_colsum.argtypes = [vp, i64, i32, i32, vp, i32, f32, vp, vp]
    lib.dtf_softmax_xent.argtypes = [vp, vp, i32, i32, vp, f32, vp, vp, vp, vp]
    lib.dtf_dense_step_begin.argtypes = [vp, vp, vp, vp]
    lib.dtf_colsum_multi.argtypes = [i32, vp, vp, vp, vp, vp, i32, f32, vp, vp]
    lib.dtf_colsum_multi.restype = C.c_int
    lib.dtf_softmax_xent_step.argtypes = [vp, vp, i32, i32, vp, f32, vp, vp, vp, vp]
    lib.dtf_softmax_xent_head.argtypes = [vp, vp, i32, i32, vp, f32, vp, C.c_int64, vp, vp, vp, vp]
    for name in ("dtf_gemm_f32", "dtf_colsum", "dtf_softmax_xent", "dtf_dense_step_begin", "dtf_softmax_xent_step",
                 "dtf_softmax_xent_head"):
        getattr(lib, name).restype = C.c_int
    lib.dtf_debug_force_generic_path.argtypes = [i32]
    lib.dtf_debug_force_generic_path.restype = None
    lib.dtf_debug_set_fwd_tile8.argtypes = [i32]
    lib.dtf_debug_set_fwd_tile8.restype = None
    lib.dtf_debug_set_head_stamps.argtypes = [vp]
    lib.dtf_debug_set_head_stamps.restype = None
    for name in ("dtf_mlp_step_fused", "dtf_mlp_step_grad", "dtf_advance_cursor", "dtf_mlp_eval",
                 "dtf_linear_bias_act", "dtf_dw_gemm", "dtf_sgd_apply", "dtf_accumulate",
                 "dtf_philox_normal", "dtf_gather_rows"):
        getattr(lib, name).restype = C.c_int


def load(required: bool = False) -> Optional[C.CDLL]:
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is None:
        try:
            if "DTF_HIP_LIB" not in os.environ:
                try:   # a no-op when the stamp matches the sources; builds in-tree otherwise (fresh checkout, stale
                    from .. import build as _build      # library; hipcc cross-compiles, no GPU needed)

                    _build.build_hip()
                except Exception as e:  # noqa: BLE001
                    if not _LIB_PATH.exists():
                        raise OSError(f"{_LIB_PATH} not built and the in-tree build failed: {e}") from None
            lib = C.CDLL(str(_LIB_PATH), mode=C.RTLD_GLOBAL)
            try:
                _declare(lib)
                lib.dtf_abi_sizeof.argtypes = [C.c_int]
            except AttributeError as e:     # a library older than this binding lacks a symbol
                raise OSError(f"{_LIB_PATH} was built from other sources ({e}); rebuild with "
                              f"python -m distributed_tensorflow_amd.build") from None
            # the ctypes mirrors must have the library's layout: a stale .so next to newer Python (or the reverse)
            # would otherwise read descriptor fields at the wrong offsets
            lib.dtf_abi_sizeof.argtypes = [C.c_int]
            lib.dtf_abi_sizeof.restype = C.c_int
            for which, mirror in enumerate((DtfMlpDesc, DtfPipeDesc, DtfXgDesc)):
                if lib.dtf_abi_sizeof(which) != C.sizeof(mirror):
                    raise OSError(f"{_LIB_PATH} was built from other sources: sizeof({mirror.__name__}) is "
                                  f"{lib.dtf_abi_sizeof(which)} there and {C.sizeof(mirror)} here; rebuild with "
                                  f"python -m distributed_tensorflow_amd.build")
            if os.environ.get("DTF_PIPE_PARTS") is not None:       # timing experiments (results are then wrong)
                lib.dtf_debug_set_pipe_parts.argtypes = [C.c_int]
                lib.dtf_debug_set_pipe_parts(int(os.environ["DTF_PIPE_PARTS"]))
            if os.environ.get("DTF_PIPE_ORDER") is not None:       # tuning experiments (see step_pipe_kernel)
                lib.dtf_debug_set_pipe_order.argtypes = [C.c_int]
                lib.dtf_debug_set_pipe_order(int(os.environ["DTF_PIPE_ORDER"]))
            _lib = lib
            return _lib
        except OSError as e:  # pragma: no cover - depends on the machine
            _load_error = str(e)
    if required:
        raise HipKernelError(f"HIP kernel library unavailable: {_load_error}")
    return None


def available() -> bool:
    """True when the kernels can actually run: library loadable AND a GPU present."""
    return torch.cuda.is_available() and load() is not None


def require() -> C.CDLL:
    return load(required=True)


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipKernelError(f"{what}: {ERR_NAMES.get(rc, rc)}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(operand: Optional[torch.Tensor] = None) -> int:
    """Handle of the current stream of the current device.  (``torch.cuda.current_stream()`` resolves its device
    through ``torch.cuda.is_available()``, a driver call of ~12 us: too much for a call made before every launch.)

    ``operand``: a tensor of the launch.  The kernels run on the calling thread's CURRENT device (and set their
    per-device attributes there); an operand that lives on another GPU means the caller forgot to make that GPU
    current (``torch.cuda.device(...)``; a new thread starts on device 0), and the launch would read memory the
    current device has no mapping of: refuse it loudly instead."""
    if _raw_stream is not None and _raw_device is not None:
        dev = _raw_device()
        if operand is not None and operand.device.index != dev:
            raise HipKernelError(f"operand on {operand.device} but the current device of this thread is cuda:{dev}: "
                                 f"enter torch.cuda.device({str(operand.device)!r}) around the call")
        return _raw_stream(dev)
    if operand is not None and operand.device.index != torch.cuda.current_device():
        raise HipKernelError(f"operand on {operand.device} but the current device of this thread is "
                             f"cuda:{torch.cuda.current_device()}")
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise HipKernelError(f"{name} must be a contiguous float32 CUDA tensor, got {t.dtype} {t.device}")
    return t


# ---------------------------------------------------------------------------
# thin op wrappers
# ---------------------------------------------------------------------------

def linear_bias_act(x, W, b, act: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = require()
    _f32c(x, "x"), _f32c(W, "W"), _f32c(b, "b")
    M, K = x.shape
    K2, H = W.shape
    if K2 != K or b.numel() != H:
        raise HipKernelError(f"linear_bias_act shape mismatch x{tuple(x.shape)} W{tuple(W.shape)} b{tuple(b.shape)}")
    if out is None:
        out = torch.empty((M, H), device=x.device, dtype=torch.float32)
    _check(lib.dtf_linear_bias_act(_ptr(x), _ptr(W), _ptr(b), _ptr(out), M, K, H, act, 1, _stream(x)),
           "dtf_linear_bias_act")
    return out


def linear_splitk(x, W, ksplit: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Raw x@W through the split-K/atomic path (accumulates into ``out``, zeroed if not given)."""
    lib = require()
    _f32c(x, "x"), _f32c(W, "W")
    M, K = x.shape
    H = W.shape[1]
    if W.shape[0] != K:
        raise HipKernelError("linear_splitk shape mismatch")
    if out is None:
        out = torch.zeros((M, H), device=x.device, dtype=torch.float32)
    _check(lib.dtf_linear_bias_act(_ptr(x), _ptr(W), None, _ptr(out), M, K, H, 0, int(ksplit), _stream(x)),
           "dtf_linear_bias_act(splitk)")
    return out


def dw_gemm(x, dz, out: Optional[torch.Tensor] = None, mode: int = 0, scale: float = 1.0,
            lr: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = require()
    _f32c(x, "x"), _f32c(dz, "dz")
    M, K = x.shape
    M2, H = dz.shape
    if M2 != M:
        raise HipKernelError("dw_gemm: batch mismatch")
    if out is None:
        out = torch.zeros((K, H), device=x.device, dtype=torch.float32)
    if tuple(out.shape) != (K, H):
        raise HipKernelError("dw_gemm: bad output shape")
    _check(lib.dtf_dw_gemm(_ptr(x), _ptr(dz), _ptr(out), M, K, H, mode, float(scale), _ptr(lr), _stream(x)),
           "dtf_dw_gemm")
    return out


GEMM_NN, GEMM_NT, GEMM_TN = 0, 1, 2
EPI_NONE, EPI_BIAS_ACT, EPI_ACT_BWD, EPI_SGD = 0, 1, 2, 3
ACT_NONE = -1


def _f32_2d(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        raise HipKernelError(f"{name} must be a 2-D float32 CUDA tensor with contiguous rows, got {t.dtype} "
                             f"{tuple(t.shape)} strides {t.stride()} on {t.device}")
    return t


def gemm(layout: int, A, B, out: Optional[torch.Tensor] = None, *, epi: int = EPI_NONE, act: int = ACT_NONE,
         bias: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None, lr: Optional[torch.Tensor] = None,
         scale: float = 1.0, tile: int = 0, workspace: Optional[torch.Tensor] = None, splits: int = 0) -> torch.Tensor:
    """``out = epilogue(op(A) @ op(B))`` on the fp32 matrix cores (csrc/gemm.hip).

    ``workspace``: a float32 tensor on the same device, zero-filled once by the caller and from then on used only by
    calls on one stream at a time (``gemm_workspace()``); with it, shapes with few tiles and a long K are split along K
    over several workgroups per tile (``splits``: 0 = chosen from the shape, n = at most n).

    layout ``GEMM_NN``: A [M,K], B [K,N]; ``GEMM_NT``: A [M,K], B [N,K]; ``GEMM_TN``: A [K,M], B [K,N].  Operands are
    row-major with contiguous rows (row strides may exceed the row length).  Epilogues: ``EPI_NONE`` ``scale * acc``;
    ``EPI_BIAS_ACT`` ``act(acc + bias)``; ``EPI_ACT_BWD`` ``acc * act'(aux)`` (aux = the activation's output);
    ``EPI_SGD`` ``out -= lr * scale * acc`` (lr: one-element device tensor).

    ``tile``: 0 = chosen from the shape; 128 / 129 = 128 x 128 tiles with four / eight waves; 64 = 64 x 64 tiles (eight
    waves); 32 = 32 x 64 tiles."""
    lib = require()
    _f32_2d(A, "A"), _f32_2d(B, "B")
    if layout == GEMM_NN:
        (M, K), (K2, N) = A.shape, B.shape
    elif layout == GEMM_NT:
        (M, K), (N, K2) = A.shape, B.shape
    elif layout == GEMM_TN:
        (K, M), (K2, N) = A.shape, B.shape
    else:
        raise HipKernelError(f"gemm: unknown layout {layout}")
    if K != K2:
        raise HipKernelError(f"gemm: inner dimensions differ, A{tuple(A.shape)} B{tuple(B.shape)} layout {layout}")
    if out is None:
        if epi == EPI_SGD:
            raise HipKernelError("gemm: the SGD epilogue updates `out` in place")
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    _f32_2d(out, "out")
    if tuple(out.shape) != (M, N) or out.device != A.device or B.device != A.device:
        raise HipKernelError(f"gemm: output {tuple(out.shape)} on {out.device} for a {M}x{N} product on {A.device}")
    if bias is not None and (_f32c(bias, "bias").numel() != N or bias.device != A.device):
        raise HipKernelError("gemm: bias must have one element per output column")
    if epi == EPI_ACT_BWD and (aux is None or tuple(_f32_2d(aux, "aux").shape) != (M, N) or aux.device != A.device):
        raise HipKernelError("gemm: the activation-gradient epilogue needs the activation's output, shaped like out")
    if epi == EPI_SGD and (lr is None or _f32c(lr, "lr").numel() != 1):
        raise HipKernelError("gemm: the SGD epilogue needs a one-element device learning rate")
    if workspace is not None and (_f32c(workspace, "workspace").device != A.device):
        raise HipKernelError("gemm: the workspace must live on the operands' device")
    _check(lib.dtf_gemm_f32_ws(layout, epi, _ptr(A), _ptr(B), _ptr(out), M, N, K, A.stride(0), B.stride(0), out.stride(0),
                               int(act), _ptr(bias), _ptr(aux), aux.stride(0) if aux is not None else 0, _ptr(lr),
                               float(scale), int(tile), _ptr(workspace), workspace.numel() if workspace is not None else 0,
                               int(splits), _stream(A)), "dtf_gemm_f32")
    return out


def gemm_workspace(device, tiles: int = 128, splits: int = 16) -> torch.Tensor:
    """A zero-filled workspace large enough for any K split ``gemm`` chooses (at most 128 tiles of 64 x 64 x 16 ranges)."""
    return torch.zeros(1024 + tiles * splits * 64 * 64, dtype=torch.float32, device=device)


def colsum(z, out: Optional[torch.Tensor] = None, *, sgd: bool = False, lr: Optional[torch.Tensor] = None,
           scale: float = 1.0) -> torch.Tensor:
    """``out = scale * z.sum(0)`` or, with ``sgd``, ``out -= lr * scale * z.sum(0)`` (bias gradients)."""
    lib = require()
    _f32_2d(z, "z")
    M, N = z.shape
    if out is None:
        if sgd:
            raise HipKernelError("colsum: the SGD form updates `out` in place")
        out = torch.empty((N,), device=z.device, dtype=torch.float32)
    if _f32c(out, "out").numel() != N or out.device != z.device:
        raise HipKernelError("colsum: one output element per column")
    if sgd and (lr is None or _f32c(lr, "lr").numel() != 1):
        raise HipKernelError("colsum: the SGD form needs a one-element device learning rate")
    _check(lib.dtf_colsum(_ptr(z), z.stride(0), M, N, _ptr(out), EPI_SGD if sgd else EPI_NONE, float(scale), _ptr(lr),
                          _stream(z)), "dtf_colsum")
    return out


class ColsumPlan:
    """Column sums of several matrices in ONE launch (``colsum_multi``): the host arrays of addresses and shapes are
    built once for a fixed set of tensors (the dz / bias pairs of an engine)."""

    def __init__(self, pairs):
        pairs = list(pairs)
        if not 1 <= len(pairs) <= 16:
            raise HipKernelError("colsum_multi: 1 to 16 matrices")
        for z, out in pairs:
            _f32_2d(z, "z")
            if _f32c(out, "out").numel() != z.shape[1] or out.device != z.device:
                raise HipKernelError("colsum_multi: one output element per column")
        n = len(pairs)
        self.count = n
        self.keep = pairs
        self.z = (C.c_uint64 * n)(*[z.data_ptr() for z, _ in pairs])
        self.out = (C.c_uint64 * n)(*[o.data_ptr() for _, o in pairs])
        self.ldz = (C.c_int64 * n)(*[z.stride(0) for z, _ in pairs])
        self.M = (C.c_int32 * n)(*[z.shape[0] for z, _ in pairs])
        self.N = (C.c_int32 * n)(*[z.shape[1] for z, _ in pairs])


def colsum_multi(plan: ColsumPlan, *, sgd: bool = False, lr: Optional[torch.Tensor] = None, scale: float = 1.0):
    """``out_t = scale * z_t.sum(0)`` (or ``out_t -= lr * scale * z_t.sum(0)`` with ``sgd``) for every pair of the plan."""
    if sgd and (lr is None or _f32c(lr, "lr").numel() != 1):
        raise HipKernelError("colsum_multi: the SGD form needs a one-element device learning rate")
    _check(require().dtf_colsum_multi(plan.count, plan.z, plan.ldz, plan.M, plan.N, plan.out, EPI_SGD if sgd else EPI_NONE,
                                      float(scale), _ptr(lr), _stream()), "dtf_colsum_multi")


def softmax_xent(logits, labels, *, dz: Optional[torch.Tensor] = None, grad_scale: float = 1.0,
                 loss_sum: Optional[torch.Tensor] = None, correct: Optional[torch.Tensor] = None,
                 row_loss: Optional[torch.Tensor] = None):
    """Softmax cross-entropy head: accumulates the sum of the row losses into ``loss_sum`` (float32[1]) and the
    number of rows classified like their labels into ``correct`` (int32[1]); ``dz`` (shaped like logits) receives
    ``grad_scale * d(row loss)/d(logits)``; ``row_loss`` (float32[M]) the per-row losses."""
    lib = require()
    _f32c(logits, "logits"), _f32c(labels, "labels")
    if logits.dim() != 2 or labels.shape != logits.shape or labels.device != logits.device:
        raise HipKernelError("softmax_xent: logits and labels must be equally shaped 2-D tensors on one device")
    M, Cn = logits.shape
    if dz is not None and (_f32c(dz, "dz").shape != logits.shape or dz.device != logits.device):
        raise HipKernelError("softmax_xent: dz must be shaped like the logits")
    if loss_sum is not None and (_f32c(loss_sum, "loss_sum").numel() != 1):
        raise HipKernelError("softmax_xent: loss_sum is one float")
    if correct is not None and (correct.dtype != torch.int32 or correct.numel() != 1 or not correct.is_cuda):
        raise HipKernelError("softmax_xent: correct is one int32 on the GPU")
    if row_loss is not None and _f32c(row_loss, "row_loss").numel() != M:
        raise HipKernelError("softmax_xent: row_loss has one element per row")
    _check(lib.dtf_softmax_xent(_ptr(logits), _ptr(labels), M, Cn, _ptr(dz), float(grad_scale), _ptr(loss_sum),
                                _ptr(correct), _ptr(row_loss), _stream(logits)), "dtf_softmax_xent")


# Up to here the one-workgroup head is used: one row per wave.  Beyond, its waves walk several rows one after the other,
# each a chain of two memory round trips, and the many-workgroup head plus the accumulator-reset launch is faster
# (batch 100: 61 against 72 us per step of a 784-100-10 stack; batch 1024: 210 against 350).
XENT_STEP_MAX_ROWS = 16


def softmax_xent_step(logits, labels, dz: torch.Tensor, grad_scale: float, loss_out: torch.Tensor,
                      correct_out: torch.Tensor, global_step: Optional[torch.Tensor] = None):
    """The head of a training step as ONE launch of one workgroup: stores the step's loss sum (float32[1]) and correct
    count (int32[1]), writes ``dz = grad_scale * d(row loss)/d(logits)`` and bumps ``global_step`` (float32[1] or None)."""
    lib = require()
    _f32c(logits, "logits"), _f32c(labels, "labels"), _f32c(dz, "dz"), _f32c(loss_out, "loss_out")
    if logits.dim() != 2 or labels.shape != logits.shape or dz.shape != logits.shape:
        raise HipKernelError("softmax_xent_step: logits, labels and dz must be equally shaped 2-D tensors")
    if correct_out.dtype != torch.int32 or correct_out.numel() != 1 or loss_out.numel() != 1:
        raise HipKernelError("softmax_xent_step: one float32 loss and one int32 count")
    M, Cn = logits.shape
    _check(lib.dtf_softmax_xent_step(_ptr(logits), _ptr(labels), M, Cn, _ptr(dz), float(grad_scale), _ptr(loss_out),
                                     _ptr(correct_out), _ptr(global_step), _stream(logits)), "dtf_softmax_xent_step")


def xent_head_scratch(rows: int, device) -> torch.Tensor:
    """Scratch of :func:`softmax_xent_head` for batches of up to ``rows`` rows (zeroed: its last word is the ticket)."""
    return torch.zeros(2 * ((int(rows) + 3) // 4) + 4, dtype=torch.float32, device=device)


def softmax_xent_head(logits, labels, dz: torch.Tensor, grad_scale: float, scratch: torch.Tensor, loss_out: torch.Tensor,
                      correct_out: torch.Tensor, global_step: Optional[torch.Tensor] = None):
    """The head of a training step over many workgroups in ONE launch with nothing to clear first: per-workgroup sums meet
    in ``scratch`` (``xent_head_scratch``; one per engine, steps in stream order), the last workgroup adds them in a fixed
    order, stores the loss sum (float32[1]) and the correct count (int32[1]) and bumps ``global_step``."""
    lib = require()
    _f32c(logits, "logits"), _f32c(labels, "labels"), _f32c(dz, "dz"), _f32c(loss_out, "loss_out"), _f32c(scratch, "scratch")
    if logits.dim() != 2 or labels.shape != logits.shape or dz.shape != logits.shape:
        raise HipKernelError("softmax_xent_head: logits, labels and dz must be equally shaped 2-D tensors")
    if correct_out.dtype != torch.int32 or correct_out.numel() != 1 or loss_out.numel() != 1:
        raise HipKernelError("softmax_xent_head: one float32 loss and one int32 count")
    M, Cn = logits.shape
    _check(lib.dtf_softmax_xent_head(_ptr(logits), _ptr(labels), M, Cn, _ptr(dz), float(grad_scale), _ptr(scratch),
                                     scratch.numel(), _ptr(loss_out), _ptr(correct_out), _ptr(global_step), _stream(logits)),
           "dtf_softmax_xent_head")


def dense_step_begin(loss_sum: Optional[torch.Tensor], correct: Optional[torch.Tensor],
                     global_step: Optional[torch.Tensor]):
    """One launch at the start of a dense-stack step: clears the step's loss / accuracy accumulators (one float, one
    int32) and bumps the step counter (one float); each may be None."""
    _check(require().dtf_dense_step_begin(_ptr(loss_sum), _ptr(correct), _ptr(global_step), _stream()),
           "dtf_dense_step_begin")


def sgd_apply_(p, g, lr: torch.Tensor, scale: float = 1.0, zero_grad: bool = False,
               global_step: Optional[torch.Tensor] = None):
    lib = require()
    _f32c(p, "p"), _f32c(g, "g"), _f32c(lr, "lr")
    if p.numel() != g.numel():
        raise HipKernelError("sgd_apply: size mismatch")
    _check(lib.dtf_sgd_apply(_ptr(p), _ptr(g), p.numel(), _ptr(lr), float(scale), int(zero_grad),
                             _ptr(global_step), _stream(p)), "dtf_sgd_apply")
    return p


def sgd_apply_atomic_(p, g, lr: torch.Tensor, scale: float = 1.0, global_step: Optional[torch.Tensor] = None):
    """``p -= lr * scale * g`` with one float atomic per element (no update lost under concurrent applies)."""
    lib = require()
    _f32c(p, "p"), _f32c(g, "g"), _f32c(lr, "lr")
    if p.numel() != g.numel():
        raise HipKernelError("sgd_apply_atomic: size mismatch")
    _check(lib.dtf_sgd_apply_atomic(_ptr(p), _ptr(g), p.numel(), _ptr(lr), float(scale), _ptr(global_step),
                                    _stream(p)), "dtf_sgd_apply_atomic")
    return p


def sum_apply_(p, grads, lr: torch.Tensor, scale: float = 1.0, global_step: Optional[torch.Tensor] = None):
    """``p -= lr * scale * sum(grads)`` in ONE launch; ``grads``: up to 16 tensors shaped like ``p`` (or their
    device addresses), summed in list order."""
    lib = require()
    _f32c(p, "p"), _f32c(lr, "lr")
    ptrs = []
    for g in grads:
        if isinstance(g, torch.Tensor):
            _f32c(g, "grad")
            if g.numel() != p.numel():
                raise HipKernelError("sum_apply: size mismatch")
            ptrs.append(g.data_ptr())
        else:
            ptrs.append(int(g))
    if not 1 <= len(ptrs) <= 16:
        raise HipKernelError("sum_apply: 1..16 gradient sources")
    tab = (C.c_uint64 * len(ptrs))(*ptrs)
    _check(lib.dtf_sum_apply(_ptr(p), tab, len(ptrs), p.numel(), _ptr(lr), float(scale), _ptr(global_step),
                             _stream(p)), "dtf_sum_apply")
    return p


def accumulate_(acc, g):
    lib = require()
    _f32c(acc, "acc"), _f32c(g, "g")
    if acc.numel() != g.numel():
        raise HipKernelError("accumulate: size mismatch")
    _check(lib.dtf_accumulate(_ptr(acc), _ptr(g), acc.numel(), _stream(acc)), "dtf_accumulate")
    return acc


def philox_normal_(out, seed: int, offset: int = 0, mean: float = 0.0, stddev: float = 1.0):
    lib = require()
    _f32c(out, "out")
    _check(lib.dtf_philox_normal(_ptr(out), out.numel(), int(seed) & (2**64 - 1), int(offset), float(mean),
                                 float(stddev), _stream(out)), "dtf_philox_normal")
    return out


_gather_err = {}


def _gather_err_word(device: torch.device) -> torch.Tensor:
    """Two ints per device: how many row indices a gather found outside its source, and where the first one sat."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _gather_err.get(key)
    if t is None and not torch.cuda.is_current_stream_capturing():
        # (never allocated inside a stream capture: the word has to outlive every graph that holds its address)
        t = _gather_err[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return t


def gather_errors(device=None) -> int:
    """Number of out-of-range row indices the gathers on ``device`` have clamped so far (synchronises).  Anything but
    0 means a permutation was corrupted or read before it had arrived: the caller should stop trusting the data set."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _gather_err.get(key)
    return 0 if t is None else int(t[0].item())


def reset_gather_errors(device=None) -> None:
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key in _gather_err:
        _gather_err[key].zero_()


def gather_rows(src, idx, out: Optional[torch.Tensor] = None):
    lib = require()
    _f32c(src, "src")
    if idx.dtype != torch.int64 or not idx.is_cuda or not idx.is_contiguous():
        raise HipKernelError("gather_rows: idx must be contiguous int64 CUDA tensor")
    if idx.device != src.device:
        raise HipKernelError(f"gather_rows: idx on {idx.device}, src on {src.device}")
    rows, cols = idx.numel(), src.shape[1]
    if out is None:
        out = torch.empty((rows, cols), device=src.device, dtype=torch.float32)
    if out.shape[0] < rows or out.shape[1] != cols or out.device != src.device:
        raise HipKernelError("gather_rows: bad output shape")
    if torch.cuda.current_device() != src.device.index:
        raise HipKernelError(f"gather_rows: launched with cuda:{torch.cuda.current_device()} current on tensors of "
                             f"{src.device} (the launch would go to another device's stream)")
    err = _gather_err_word(src.device)
    _check(lib.dtf_gather_rows(_ptr(src), _ptr(idx), _ptr(out), rows, cols, src.shape[0], _ptr(err), _stream(src)),
           "dtf_gather_rows")
    return out


def gather_gram_supported(B: int, K: int) -> bool:
    return bool(require().dtf_gather_gram_supported(int(B), int(K)))


def gather_gram(src, perm, dst, gram, B: int, nb: int, ysrc=None, ydst=None):
    """Epoch refresh in ONE launch: ``dst[r] = src[perm[r]]`` for r < nb * B (``dst`` None: not stored),
    ``gram[j, m, i] = 1 + x_{j+1}[m] . x_j[i]`` over the epoch-ordered rows and, with ``ysrc`` / ``ydst``, the
    labels ``ydst[r] = ysrc[perm[r]]`` (csrc/gram.hip)."""
    lib = require()
    _f32c(src, "src"), _f32c(gram, "gram")
    if dst is not None:
        _f32c(dst, "dst")
    if perm.dtype != torch.int64 or not perm.is_cuda or not perm.is_contiguous() or perm.numel() < nb * B:
        raise HipKernelError("gather_gram: perm must be a contiguous int64 CUDA tensor of at least nb * B entries")
    K = src.shape[1]
    if gram.dim() != 3 or gram.shape[0] < nb - 1 or gram.shape[1] != B or (dst is not None and (
            dst.shape[1] != K or dst.shape[0] < nb * B)):
        raise HipKernelError("gather_gram: shape mismatch")
    C_ = 0
    if ydst is not None:
        _f32c(ysrc, "ysrc"), _f32c(ydst, "ydst")
        C_ = ysrc.shape[1]
        if ysrc.shape[0] != src.shape[0] or ydst.shape[1] != C_ or ydst.shape[0] < nb * B:
            raise HipKernelError("gather_gram: label shape mismatch")
    if perm.device != src.device or gram.device != src.device or torch.cuda.current_device() != src.device.index:
        raise HipKernelError(f"gather_gram: tensors on {src.device} / {perm.device} / {gram.device}, launched with "
                             f"cuda:{torch.cuda.current_device()} current")
    _check(lib.dtf_gather_gram(_ptr(src), _ptr(perm), _ptr(dst), _ptr(gram), src.shape[0], int(B), K, int(nb),
                               gram.shape[2], _ptr(ysrc) if ydst is not None else None, _ptr(ydst), C_, _stream(src)),
           "dtf_gather_gram")
    return gram


def mlp_eval(x, y, W1, b1, W2, b2, act: int, a2_ws, loss_sum, correct):
    lib = require()
    for n, t in (("x", x), ("y", y), ("W1", W1), ("b1", b1), ("W2", W2), ("b2", b2), ("a2_ws", a2_ws),
                 ("loss_sum", loss_sum)):
        _f32c(t, n)
    M, K = x.shape
    H, Cc = W2.shape
    if W1.shape != (K, H) or y.shape != (M, Cc) or a2_ws.numel() < M * H or correct.dtype != torch.int32:
        raise HipKernelError("mlp_eval: shape mismatch")
    _check(lib.dtf_mlp_eval(_ptr(x), _ptr(y), M, K, H, Cc, act, _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2),
                            _ptr(a2_ws), _ptr(loss_sum), _ptr(correct), _stream(x)), "dtf_mlp_eval")


def mlp_step_fused(desc: DtfMlpDesc, batch_index: int, parity: int):
    _check(require().dtf_mlp_step_fused(C.byref(desc), int(batch_index), int(parity), _stream()),
           "dtf_mlp_step_fused")


PIPE_CARRY_IN, PIPE_CARRY_OUT = 1, 2


def mlp_chunk_pipelined(desc: DtfMlpDesc, pipe: DtfPipeDesc, first_batch: int, n: int, flags: int = 0):
    """n fused steps with one launch per step (csrc/mlp_step2.hip: step_pipe_kernel).  ``flags``: PIPE_CARRY_IN (the
    previous chunk ended here with PIPE_CARRY_OUT: no prologue, no stand-alone forward) | PIPE_CARRY_OUT (the last
    launch also holds the forward of the batch after the chunk)."""
    _check(require().dtf_mlp_chunk_pipelined_ex(C.byref(desc), C.byref(pipe), int(first_batch), int(n), int(flags),
                                                _stream()), "dtf_mlp_chunk_pipelined")


def mlp_chunk_pipelined_supported(B: int, K: int, H: int, Cn: int) -> bool:
    return bool(require().dtf_mlp_chunk_pipelined_supported(int(B), int(K), int(H), int(Cn)))


def mlp_step_grad(desc: DtfMlpDesc, batch_index: int, parity: int, grad_scale: float = 1.0):
    _check(require().dtf_mlp_step_grad(C.byref(desc), int(batch_index), int(parity), float(grad_scale),
                                       _stream()), "dtf_mlp_step_grad")


def mlp_step_async(desc: DtfMlpDesc, batch_index: int, parity: int, xg: DtfXgDesc):
    """Forward + a backward that pushes ``shard -= lr * g`` element-wise into the shards described by ``xg``."""
    _check(require().dtf_mlp_step_async(C.byref(desc), int(batch_index), int(parity), C.byref(xg), _stream()),
           "dtf_mlp_step_async")


def mlp_step_grad_shards(desc: DtfMlpDesc, batch_index: int, parity: int, grad_scale: float, shards: DtfXgDesc):
    """Gradient step whose forward reads the variables from mapped parameter-server shards (no pull in front)."""
    _check(require().dtf_mlp_step_grad_shards(C.byref(desc), int(batch_index), int(parity), float(grad_scale),
                                              C.byref(shards), _stream()), "dtf_mlp_step_grad_shards")


def mlp_step_xg(desc: DtfMlpDesc, batch_index: int, parity: int, grad_scale: float, xg: DtfXgDesc):
    _check(require().dtf_mlp_step_xg(C.byref(desc), int(batch_index), int(parity), float(grad_scale), C.byref(xg),
                                     _stream()), "dtf_mlp_step_xg")


def transpose(src: torch.Tensor, dst: torch.Tensor):
    """``dst[c, r] = src[r, c]`` for contiguous float32 matrices (one launch)."""
    _f32c(src, "src"), _f32c(dst, "dst")
    if src.dim() != 2 or dst.shape != (src.shape[1], src.shape[0]) or dst.device != src.device:
        raise HipKernelError("transpose: dst must be shaped like src.T on the same device")
    _check(require().dtf_transpose(_ptr(src), _ptr(dst), src.shape[0], src.shape[1], _stream(src)), "dtf_transpose")


def advance_cursor(cursor: torch.Tensor, n: int):
    _check(require().dtf_advance_cursor(_ptr(cursor), int(n), _stream(cursor)), "dtf_advance_cursor")


def enable_peer_access(device: torch.device, peer: torch.device) -> None:
    """Kernels launched on ``device`` may dereference tensors that live on ``peer`` afterwards."""
    if device.type != "cuda" or peer.type != "cuda":
        return
    a = device.index if device.index is not None else torch.cuda.current_device()
    b = peer.index if peer.index is not None else torch.cuda.current_device()
    if a != b:
        _check(require().dtf_enable_peer_access(int(a), int(b)), f"peer access cuda:{a} -> cuda:{b}")


_hiprt = None


def graph_upload(graph: "torch.cuda.CUDAGraph") -> bool:
    """Have the runtime place an instantiated graph on the device now (``hipGraphUpload``) instead of during its
    first launch.  False when the handle or the entry point is not available (nothing is lost: the first replay does it)."""
    global _hiprt
    try:
        exec_handle = graph.raw_cuda_graph_exec()
        if _hiprt is None:
            _hiprt = C.CDLL("libamdhip64.so")
            _hiprt.hipGraphUpload.argtypes = [C.c_void_p, C.c_void_p]
            _hiprt.hipGraphUpload.restype = C.c_int
        return _hiprt.hipGraphUpload(exec_handle, _stream()) == 0
    except Exception:  # noqa: BLE001 — an optimisation only
        return False


def streams_independent(a: "torch.cuda.Stream", b: "torch.cuda.Stream", timeout_us: int = 20000) -> bool:
    """Whether a kernel on ``b`` runs while a kernel on ``a`` waits for it (both directions are checked): False when the
    two streams of this process share a hardware queue."""
    lib = require()
    lib.dtf_streams_independent.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.dtf_streams_independent.restype = C.c_int
    for x, y in ((a, b), (b, a)):
        rc = lib.dtf_streams_independent(x.cuda_stream, y.cuda_stream, int(timeout_us))
        if rc < 0:
            raise HipKernelError("dtf_streams_independent: HIP error")
        if rc == 0:
            return False
    return True


def build_info() -> str:
    lib = load()
    return lib.dtf_hip_build_info().decode() if lib is not None else "unavailable"


def force_generic_path(on: bool):
    """Route steps through the generic three-launch kernels even where the two-launch path applies."""
    require().dtf_debug_force_generic_path(1 if on else 0)


def set_forward_tile8(on: bool):
    """Two-launch path: forward with 8x8 output tiles (default) or the generic 16x16-tile kernel."""
    require().dtf_debug_set_fwd_tile8(1 if on else 0)

