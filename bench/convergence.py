"""Convergence check of the resident tier: the reference's 100-epoch recipe (SGD lr 0.001, batch 100,
550 steps/epoch) on the synthetic MNIST-shaped set; prints test accuracy every 10 epochs and a JSON summary.

    python bench/convergence.py [--epochs 100] [--workers_equiv 1]

``--workers_equiv N`` mimics N asynchronous workers' update count on one GPU by running N x 550 steps per
epoch (the reference's accuracy rises with the worker count only because the number of updates does,
README.md:70-72).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import DeviceDataset, read_data_sets  # noqa: E402
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--workers_equiv", type=int, default=1)
    ap.add_argument("--lr", type=float, default=0.001)
    args = ap.parse_args()
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    mnist = read_data_sets("MNIST_data", one_hot=True)
    ds = DeviceDataset(mnist.train.images, mnist.train.labels, dev, 100, seed=0)
    tr = build_replica_trainer(mode="single", device=dev, dataset=ds, batch=100, lr=args.lr, seed=1, chunk=50)
    tx, ty = torch.as_tensor(mnist.test.images).to(dev), torch.as_tensor(mnist.test.labels).to(dev)
    t0 = time.time()
    curve = []
    for ep in range(args.epochs):
        tr.run_steps(550 * args.workers_equiv)
        if (ep + 1) % 10 == 0 or ep == 0:
            loss, acc = tr.evaluate(tx, ty)
            curve.append((ep + 1, round(acc, 4), round(loss, 4)))
            print("epoch %3d  test-accuracy %.4f  test-loss %.4f  (%.1fs)" % (ep + 1, acc, loss, time.time() - t0), flush=True)
    print(json.dumps({"epochs": args.epochs, "updates": tr.global_step_value(), "final_test_accuracy": curve[-1][1],
                      "data": mnist.source, "seconds": round(time.time() - t0, 2), "curve": curve}))


if __name__ == "__main__":
    main()
