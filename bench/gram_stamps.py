"""Timeline of the fused gather + Gram kernel (csrc/gram.hip): wall-clock stamps of wave 0 of every workgroup.

    python bench/gram_stamps.py [parts]
"""
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
    lib.dtf_debug_set_gram_stamps.argtypes = [ctypes.c_void_p]
    lib.dtf_debug_set_gram_parts(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
    stamps = torch.zeros(256, 24, dtype=torch.int64, device=dev)
    for _ in range(3):
        hb.gather_gram(src, perm, dst, gram, 100, 550, ysrc, ydst)
    lib.dtf_debug_set_gram_stamps(stamps.data_ptr())
    hb.gather_gram(src, perm, dst, gram, 100, 550, ysrc, ydst)
    torch.cuda.synchronize()
    lib.dtf_debug_set_gram_stamps(None)
    lib.dtf_debug_set_gram_parts(7)
    t = stamps.cpu().double() / 100.0            # us
    t0 = t[:, 0].min()
    t = t - t0
    names = ["start", "before the chunk loop"] + [f"chunk {c} may be written" for c in range(13)]
    cols = [0, 1] + [2 + c for c in range(13)]
    names += ["after the loop", "end"]
    cols += [22, 23]
    print("us since the first workgroup started: min / median / max over 256 workgroups")
    for n, c in zip(names, cols):
        v = t[:, c]
        print(f"  {n:26s} {v.min():8.2f} {v.median():8.2f} {v.max():8.2f}")
    d = t[:, 3:15] - t[:, 2:14]
    print(f"chunk period (first barrier to first barrier): median {d.median():.2f} us, min {d.min():.2f}, max {d.max():.2f}")


if __name__ == "__main__":
    main()
