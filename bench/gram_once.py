"""A few launches of the fused gather + Gram kernel (for rocprofv3 --pmc / --kernel-trace runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from distributed_tensorflow_amd.ops import hip_backend as hb  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
src = torch.rand(55000, 784, generator=g).to(dev)
perm = torch.randperm(55000, generator=g).to(dev)
dst = torch.empty_like(src)
gram = torch.zeros(549, 100, 112, device=dev)
ysrc = torch.rand(55000, 10, generator=g).to(dev)
ydst = torch.empty_like(ysrc)
if len(sys.argv) > 1:
    import ctypes

    hb.require().dtf_debug_set_gram_parts.argtypes = [ctypes.c_int]
    hb.require().dtf_debug_set_gram_parts(int(sys.argv[1]))
for _ in range(6):
    hb.gather_gram(src, perm, dst, gram, 100, 550, ysrc, ydst)
torch.cuda.synchronize()
