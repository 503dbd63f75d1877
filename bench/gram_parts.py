"""Where the time of the fused gather + Gram kernel goes: the kernel with some of its parts switched off
(csrc/gram.hip, dtf_debug_set_gram_parts), 20 launches each, device events."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from distributed_tensorflow_amd.ops import hip_backend as hb  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    src = torch.rand(55000, 784, generator=g).to(dev)
    perm = torch.randperm(55000, generator=g).to(dev)
    dst = torch.empty_like(src)
    gram = torch.zeros(549, 100, 112, device=dev)
    ysrc = torch.rand(55000, 10, generator=g).to(dev)
    ydst = torch.empty_like(ysrc)
    lib = hb.require()
    lib.dtf_debug_set_gram_parts.argtypes = [ctypes.c_int]
    names = {7: "all", 3: "loads + products (no row stores)", 5: "loads + stores (no products)", 2: "products only",
             1: "loads only", 0: "staging + barriers only"}
    for parts, name in names.items():
        lib.dtf_debug_set_gram_parts(parts)
        for _ in range(3):
            hb.gather_gram(src, perm, dst, gram, 100, 550, ysrc, ydst)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            hb.gather_gram(src, perm, dst, gram, 100, 550, ysrc, ydst)
        e1.record()
        torch.cuda.synchronize()
        print(f"parts={parts} ({name}): {e0.elapsed_time(e1) / 20 * 1000:.1f} us", flush=True)
    lib.dtf_debug_set_gram_parts(7)


if __name__ == "__main__":
    main()
