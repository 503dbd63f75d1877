"""Throughput of the fp32 MFMA GEMM (csrc/gemm.hip) next to the library GEMM (torch.matmul -> rocBLAS / hipBLASLt).

    python bench/gemm_bench.py [--iters 20] [--shapes 4096x4096x4096,...]

One line per (shape, layout): microseconds and TFLOP/s of both, the ratio, the tile the kernel chose, and the largest
element-wise difference between the two results.  fp32 MFMA peak of an MI355X: 256 CUs x 256 FLOP/clk x 2.4 GHz = 157 TFLOP/s.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip_backend as hb  # noqa: E402

DEFAULT = "4096x4096x4096,8192x8192x4096,3072x3072x3072,2048x2048x2048,1024x1024x1024,10000x100x784,4096x1024x784,100x1024x784,100x784x1024,784x1024x100"


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default=DEFAULT)
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--workspace", type=int, default=1, help="1: let shapes with few tiles split K (the engines' setting)")
    ap.add_argument("--split_knobs", default="", help="max_tiles,target_workgroups of the K split (tuning)")
    a = ap.parse_args()
    if a.split_knobs:
        hb.require().dtf_debug_set_gemm_split(*(int(v) for v in a.split_knobs.split(",")))
    dev = torch.device("cuda")
    ws = hb.gemm_workspace(dev, tiles=1024, splits=8) if a.workspace else None
    print(f"{'M x N x K':>20} {'layout':>6} {'tile':>4} {'ours us':>10} {'TFLOP/s':>8} {'library us':>10} {'TFLOP/s':>8} {'ours/lib':>8} {'max diff':>9}")
    for shape in a.shapes.split(","):
        M, N, K = (int(v) for v in shape.split("x"))
        flop = 2.0 * M * N * K
        for layout, name in ((hb.GEMM_NN, "NN"), (hb.GEMM_NT, "NT"), (hb.GEMM_TN, "TN")):
            A = torch.randn((K, M) if layout == hb.GEMM_TN else (M, K), device=dev)
            B = torch.randn((N, K) if layout == hb.GEMM_NT else (K, N), device=dev)
            out = torch.empty(M, N, device=dev)
            lib_out = torch.empty(M, N, device=dev)
            At = A.t() if layout == hb.GEMM_TN else A
            Bt = B.t() if layout == hb.GEMM_NT else B
            t_lib = timed(lambda: torch.matmul(At, Bt, out=lib_out), a.iters)
            for tile in (int(t) for t in a.tiles.split(",")):
                t_ours = timed(lambda: hb.gemm(layout, A, B, out=out, tile=tile, workspace=ws), a.iters)
                diff = float((out - lib_out).abs().max())
                print(f"{shape:>20} {name:>6} {tile:>4} {t_ours:>10.1f} {flop / t_ours / 1e6:>8.1f} {t_lib:>10.1f} "
                      f"{flop / t_lib / 1e6:>8.1f} {t_lib / t_ours:>8.2f} {diff:>9.2e}", flush=True)


if __name__ == "__main__":
    main()
