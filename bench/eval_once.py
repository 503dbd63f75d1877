"""Evaluation of the 10,000-row test set: one-kernel path against the GEMM + head pair (run under rocprofv3 for kernel times)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip_backend as hb  # noqa: E402

dev = torch.device("cuda")
M, K, H = 10000, 784, 100
g = torch.Generator().manual_seed(0)
x = torch.rand(M, K, generator=g).to(dev)
y = torch.nn.functional.one_hot(torch.randint(0, 10, (M,), generator=g), 10).float().to(dev)
W1, b1 = torch.randn(K, H, generator=g).to(dev) * 0.05, torch.randn(H, generator=g).to(dev)
W2, b2 = torch.randn(H, 10, generator=g).to(dev), torch.randn(10, generator=g).to(dev)
ws = torch.empty(M * H, device=dev)
ls, cc = torch.zeros(1, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
lib = hb.require()
lib.dtf_debug_set_eval_fused.argtypes = [C.c_int]
if len(sys.argv) > 1 and sys.argv[1] == "pmc":                  # a few plain launches for a counter run
    for _ in range(5):
        hb.mlp_eval(x, y, W1, b1, W2, b2, 0, ws, ls, cc)
    torch.cuda.synchronize()
    sys.exit(0)
for fused in (1, 0):
    lib.dtf_debug_set_eval_fused(fused)
    for _ in range(5):
        hb.mlp_eval(x, y, W1, b1, W2, b2, 0, ws, ls, cc)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            hb.mlp_eval(x, y, W1, b1, W2, b2, 0, ws, ls, cc)
    gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print("%s: %.1f us per evaluation of %d rows (%.1f TFLOP/s first layer)" % (
        "one kernel" if fused else "GEMM + head", dt * 1e6, M, 2.0 * M * K * H / dt / 1e12), flush=True)
lib.dtf_debug_set_eval_fused(1)
if len(sys.argv) > 1 and sys.argv[1] == "parts":
    lib.dtf_debug_set_eval_parts.argtypes = [C.c_int]
    for parts, what in ((1, "loads + LDS writes only"), (2, "products only"), (0, "neither"), (3, "all")):
        lib.dtf_debug_set_eval_parts(parts)
        for _ in range(3):
            hb.mlp_eval(x, y, W1, b1, W2, b2, 0, ws, ls, cc)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                hb.mlp_eval(x, y, W1, b1, W2, b2, 0, ws, ls, cc)
        gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            gr.replay()
        torch.cuda.synchronize()
        print("  parts %d (%s): %.1f us" % (parts, what, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
    lib.dtf_debug_set_eval_parts(3)
    stamps = torch.zeros(8, dtype=torch.int64, device=dev)
    lib.dtf_debug_set_eval_stamps.argtypes = [C.c_void_p]
    for parts in (0, 3):
        lib.dtf_debug_set_eval_parts(parts)
        lib.dtf_debug_set_eval_stamps(stamps.data_ptr())
        hb.mlp_eval(x, y, W1, b1, W2, b2, 0, ws, ls, cc)
        torch.cuda.synchronize()
        lib.dtf_debug_set_eval_stamps(None)
        t = stamps.cpu().tolist()
        print("  parts %d, workgroup 100, ns since its start: small operands staged %d, first chunk in LDS %d, K loop done %d, "
              "activations in LDS %d, losses summed %d, atomics done %d" % ((parts,) + tuple((v - t[0]) * 10 for v in t[1:7])))
    lib.dtf_debug_set_eval_parts(3)
