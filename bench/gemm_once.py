"""A few launches of one GEMM shape, for rocprofv3 (kernel trace or counters).
    python bench/gemm_once.py 4096x4096x4096 NT [tile] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip_backend as hb  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1].split("x"))
layout = {"NN": hb.GEMM_NN, "NT": hb.GEMM_NT, "TN": hb.GEMM_TN}[sys.argv[2]]
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda")
A = torch.randn((K, M) if layout == hb.GEMM_TN else (M, K), device=dev)
B = torch.randn((N, K) if layout == hb.GEMM_NT else (K, N), device=dev)
out = torch.empty(M, N, device=dev)
library = os.environ.get("DTF_GEMM_ONCE_LIBRARY") == "1"       # the library GEMM instead (for counter comparisons)
At = A.t() if layout == hb.GEMM_TN else A
Bt = B.t() if layout == hb.GEMM_NT else B
for _ in range(iters):
    if library:
        torch.matmul(At, Bt, out=out)
    else:
        hb.gemm(layout, A, B, out=out, tile=tile)
torch.cuda.synchronize()
print("done", float(out[0, 0]))
