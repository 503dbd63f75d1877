"""Training steps of fully connected stacks on the dense-layer kernels (models/dense.py) next to the same steps written
with torch.nn.functional + autograd + an SGD loop (library GEMMs), both replayed from graphs where possible.

    python bench/dense_bench.py [--steps 200]

One line per (layer widths, batch): microseconds per step of both, samples/s, and the TFLOP/s the step's GEMMs reach
(6 * batch * sum(n_i * n_{i+1}) FLOP per step, minus the input gradient of the first layer)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.models import DenseStackEngine  # noqa: E402

CONFIGS = [((784, 100, 10), 100), ((784, 1024, 1024, 10), 100), ((784, 1024, 1024, 10), 1024), ((784, 4096, 4096, 10), 4096),
           ((784, 2048, 2048, 2048, 10), 8192)]


def torch_step_factory(Ws, bs, lr):
    def step(x, y):
        h = x
        for i in range(len(Ws)):
            h = torch.addmm(bs[i], h, Ws[i])
            if i < len(Ws) - 1:
                h = torch.relu(h)
        loss = -(y * torch.log_softmax(h, 1)).sum(1).mean()
        grads = torch.autograd.grad(loss, Ws + bs)
        with torch.no_grad():
            for p, g in zip(Ws + bs, grads):
                p.sub_(g, alpha=lr)
        return loss
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(f"{'widths':>28} {'batch':>6} {'ours us/step':>12} {'samples/s':>12} {'TFLOP/s':>8} {'torch us/step':>13} {'ours/torch':>10}")
    for dims, B in CONFIGS:
        gen = torch.Generator().manual_seed(0)
        Ws = [(torch.randn(dims[i], dims[i + 1], generator=gen) / dims[i] ** 0.5).to(dev) for i in range(len(dims) - 1)]
        bs = [torch.zeros(dims[i + 1], device=dev) for i in range(len(dims) - 1)]
        nb = 8
        x = torch.rand(nb * B, dims[0], generator=gen).to(dev)
        y = torch.eye(dims[-1])[torch.randint(0, dims[-1], (nb * B,), generator=gen)].to(dev)
        eng = DenseStackEngine([w.clone() for w in Ws], [b.clone() for b in bs], ["relu"] * (len(dims) - 2), batch=B, lr=0.01)
        eng.bind_data(x, y)
        eng.run_chunk(nb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps // nb):
            eng.run_chunk(nb)
        torch.cuda.synchronize()
        ours = (time.perf_counter() - t0) / (a.steps // nb * nb)
        flop = 6.0 * B * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)) - 2.0 * B * dims[0] * dims[1]
        Wt = [w.clone().requires_grad_(True) for w in Ws]
        bt = [b.clone().requires_grad_(True) for b in bs]
        step = torch_step_factory(Wt, bt, 0.01)
        for i in range(3):
            step(x[:B], y[:B])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            j = i % nb
            step(x[j * B:(j + 1) * B], y[j * B:(j + 1) * B])
        torch.cuda.synchronize()
        ref = (time.perf_counter() - t0) / a.steps
        print(f"{'-'.join(map(str, dims)):>28} {B:>6} {ours * 1e6:>12.1f} {B / ours:>12.0f} {flop / ours / 1e12:>8.1f} "
              f"{ref * 1e6:>13.1f} {ref / ours:>10.2f}", flush=True)


if __name__ == "__main__":
    main()
