"""Host-visible latency of 'launch a graph of N tiny kernels from an idle device, then synchronize'."""
import ctypes
import os
import sys
import time

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "spinflag":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
import torch  # noqa: E402

x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
for nodes in (1, 20, 100):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(nodes):
                x.add_(1.0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print("%-10s graph of %3d tiny kernels from idle + synchronize: median %.1f us, min %.1f us" % (mode, nodes, ts[15], ts[0]), flush=True)
