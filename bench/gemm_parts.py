"""Where the unpinned K loop of the GEMM loses time: 4096^3 (or MxNxK given as the first argument, with the tile as the
second) with parts of a chunk switched off (results are then wrong).
The first line is the kernel as it is (pinned steady-state chunk); every other line runs the unpinned chunk."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from distributed_tensorflow_amd.ops import hip_backend as hb
lib = hb.require()
dev = torch.device("cuda")
M, N, K = (int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (4096, 4096, 4096)
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
Bn = torch.randn(K, N, device=dev)
def timed(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
for _ in range(30):          # clocks up before the first line
    hb.gemm(hb.GEMM_NT, A, B, out=out, tile=tile)
for parts, what in ((15, "the kernel (pinned chunk)"), (31, "unpinned chunk, all parts"), (11, "no barrier"), (14, "no global loads"), (13, "no LDS writes"), (7, "no LDS reads"), (12, "no loads, no LDS writes"), (4, "MFMA + barrier only"), (0, "MFMA only")):
    lib.dtf_debug_set_gemm_parts(parts)
    t_nt = timed(lambda: hb.gemm(hb.GEMM_NT, A, B, out=out, tile=tile))
    t_nn = timed(lambda: hb.gemm(hb.GEMM_NN, A, Bn, out=out, tile=tile))
    f = 2.0 * M * N * K
    print(f"{what:28s} NT {t_nt:8.1f} us {f/t_nt/1e6:6.1f} TFLOP/s   NN {t_nn:8.1f} us {f/t_nn/1e6:6.1f} TFLOP/s", flush=True)
