"""Convergence table: the reference's recipe (N(0,1) init, SGD lr 0.001, batch 100, 550 steps/epoch) on a synthetic
MNIST-shaped set, every execution tier next to the plain PyTorch fp32 oracle at EQUAL update counts.

    python bench/convergence.py [--variant hard|easy] [--epochs 100] [--tiers oracle,single,two_launch,async,towers,cluster]

``--variant hard`` (default) is the slow-learning set calibrated so that the oracle lands on the reference's MNIST
figures (72 % after 55,000 updates, 80 % after 110,000: README.md:15,66-68); ``easy`` saturates near 96 %.
Tiers: ``oracle`` plain PyTorch ops (autograd-free closed form, ops/reference.py) on the same device;
``single`` the one-launch pipelined step; ``two_launch`` the fused two-launch step; ``async`` one asynchronous
worker against a GPU-resident shard (one-sided pull / push in the backward); ``towers`` synchronous replicas of
one (gradient slab + sharded SGD kernel); ``cluster`` the between-graph scripts as separate processes
(``--resident``, ipc transport): 1 ps + 2 asynchronous workers (110,000 updates), 1 ps + 2 synchronous workers (55,000
updates), 1 ps + 3 asynchronous workers (165,000 updates), 1 ps + 3 synchronous workers with a backup worker (55,000).  Prints one line per checkpoint and a JSON summary.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distributed_tensorflow_amd.data import DeviceDataset, read_data_sets  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer  # noqa: E402
from distributed_tensorflow_amd.variables import ShardLayout, assign_shards, mlp_var_specs  # noqa: E402


def oracle_curve(mnist, dev, epochs, checkpoints, lr=0.001, batch=100, seed=1, data_seed=0):
    """The recipe in plain PyTorch fp32 ops, same initial values as every tier (variables.py initializers)."""
    specs = mlp_var_specs()
    assign_shards(specs, 1)
    layout = ShardLayout(specs, 1, mode="flat")
    slab = layout.new_slab(dev)
    layout.initialize(slab, seed)
    p = {k: v.clone() for k, v in layout.mlp_views(slab).items()}
    x = torch.as_tensor(mnist.train.images).to(dev)
    y = torch.as_tensor(mnist.train.labels).to(dev)
    tx, ty = torch.as_tensor(mnist.test.images).to(dev), torch.as_tensor(mnist.test.labels).to(dev)
    gen = torch.Generator().manual_seed(data_seed)
    out, step = {}, 0
    nb = x.shape[0] // batch
    for _ in range(epochs):
        perm = torch.randperm(x.shape[0], generator=gen).to(dev)
        for i in range(nb):
            idx = perm[i * batch:(i + 1) * batch]
            g = ref.mlp_grads(x[idx], y[idx], p["W1"], p["b1"], p["W2"], p["b2"])
            for k in p:
                p[k].sub_(g["d" + k], alpha=lr)
        step += nb
        if step in checkpoints:
            out[step] = round(float(ref.mlp_loss_acc(tx, ty, p["W1"], p["b1"], p["W2"], p["b2"])[1]), 4)
    return out


def trainer_curve(mode, mnist, dev, epochs, checkpoints, **kw):
    ds = DeviceDataset(mnist.train.images, mnist.train.labels, dev, 100, seed=0)
    tr = build_replica_trainer(mode=mode, device=dev, dataset=ds, batch=100, lr=0.001, seed=1, chunk=50, **kw)
    tx, ty = torch.as_tensor(mnist.test.images).to(dev), torch.as_tensor(mnist.test.labels).to(dev)
    out, step = {}, 0
    for _ in range(epochs):
        tr.run_steps(550)
        step += 550
        if step in checkpoints:
            out[step] = round(tr.evaluate(tx, ty)[1], 4)
    if hasattr(tr, "close"):
        tr.close()
    return out


def cluster_final(script, workers, epochs, variant, extra=()):
    env = dict(os.environ, DTF_DATA_VARIANT=variant)
    cmd = [sys.executable, os.path.join(ROOT, "bench", "run_cluster.py"), "--script", script, "--ps", "1", "--workers",
           str(workers), "--transport", "ipc", "--epochs", str(epochs), "--timeout", "900", "--", "--resident", *extra]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1000)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if not line:
        return {"error": (p.stdout + p.stderr)[-500:]}
    doc = json.loads(line[-1])
    return {name: {"final_accuracy": t.get("final_accuracy"), "last_step": t.get("last_step"),
                   "steady_epoch_s": t.get("steady_epoch_s")} for name, t in doc["tasks"].items() if name.startswith("worker")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="hard", choices=["hard", "easy"])
    ap.add_argument("--epochs", type=int, default=200, help="550-step epochs (100 = 55,000 updates)")
    ap.add_argument("--tiers", default="oracle,single,two_launch,async,towers,cluster")
    args = ap.parse_args()
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    mnist = read_data_sets("MNIST_data", one_hot=True, variant=args.variant)
    checkpoints = {550 * e for e in (10, 50, 100, 200, 300) if e <= args.epochs}
    tiers = args.tiers.split(",")
    table, t0 = {}, time.time()
    for tier in tiers:
        if tier == "oracle":
            table[tier] = oracle_curve(mnist, dev, args.epochs, checkpoints)
        elif tier == "single":
            table[tier] = trainer_curve("single", mnist, dev, args.epochs, checkpoints)
        elif tier == "two_launch":
            table[tier] = trainer_curve("single", mnist, dev, args.epochs, checkpoints, pipelined=False)
        elif tier == "async" and dev.type == "cuda":
            table[tier] = trainer_curve("xgmi_async", mnist, dev, args.epochs, checkpoints)
        elif tier == "towers":
            table[tier] = trainer_curve("towers", mnist, dev, args.epochs, checkpoints)
        elif tier == "cluster" and dev.type == "cuda":
            table["cluster_async_1ps_2w_100_epochs"] = cluster_final("tfdist_between.py", 2, 100, args.variant)
            table["cluster_sync_1ps_2w_100_epochs"] = cluster_final("tfdist_between_sync.py", 2, 100, args.variant)
            # the reference's last recipe (README.md:231-254 there): three asynchronous workers, 165,000 updates
            table["cluster_async_1ps_3w_100_epochs"] = cluster_final("tfdist_between.py", 3, 100, args.variant)
            # three synchronous workers of which the first two gradients of a step count (a backup worker): 55,000 updates
            table["cluster_sync_1ps_3w_backup_100_epochs"] = cluster_final("tfdist_between_sync.py", 3, 100, args.variant,
                                                                           ("--replicas_to_aggregate=2",))
        else:
            continue
        print(f"{tier:>11}: {table.get(tier, '')}  ({time.time() - t0:.1f}s)", flush=True)
    print(json.dumps({"variant": args.variant, "data": mnist.source, "updates_per_epoch": 550, "table": table,
                      "reference_mnist": {"55000": 0.72, "110000": 0.80, "165000": "0.82-0.83"},
                      "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
