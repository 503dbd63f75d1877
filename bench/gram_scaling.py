"""Time of the fused gather + Gram kernel against the number of workgroups (2 per pair of batches): does a CU that
holds three workgroups finish them in less than three times the time of one?"""
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
ysrc = torch.rand(55000, 10, generator=g).to(dev)
ydst = torch.empty_like(ysrc)
for pairs in (64, 128, 256, 384, 512, 549):
    nb = pairs + 1
    gram = torch.zeros(pairs, 100, 112, device=dev)
    for _ in range(3):
        hb.gather_gram(src, perm, dst, gram, 100, nb, ysrc, ydst)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        hb.gather_gram(src, perm, dst, gram, 100, nb, ysrc, ydst)
    e1.record()
    torch.cuda.synchronize()
    print("%4d pairs = %4d workgroups (%.2f per CU): %.1f us" % (pairs, 2 * pairs, 2 * pairs / 256, e0.elapsed_time(e1) * 100), flush=True)
