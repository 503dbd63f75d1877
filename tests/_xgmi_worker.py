"""One replica of the peer-mapped exchange test (several of these share the GPU).

usage: _xgmi_worker.py RANK WORLD PORT STEPS USE_GRAPH UNCACHED [sync|sync2|sync2c|sync2p|sync_backup|async|async_sep|async_auto]
Prints one JSON line: health, bitwise agreement of the replicas, distance to a CPU fp32 reference
(sync), or the state of the shared shards after every worker finished (async).
"""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd.data import DeviceDataset          # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref         # noqa: E402
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer  # noqa: E402


def run_async(rank, world, steps, use_graph, dev, ds, B, lr, fused_push=True):
    # fused_push="auto": the worker measures the forms of its step against the live shards and keeps the fastest
    t = build_replica_trainer("xgmi_async", device=dev, dataset=ds, batch=B, lr=lr, seed=1, rank=rank, world=world,
                              chunk=3, use_graph=bool(use_graph), fused_push=fused_push)
    p0 = t.params.clone()
    loss0, _ = t.evaluate(ds.x_epoch, ds.y_epoch)
    t.prepare((steps,))
    dist.barrier()
    t.run_steps(steps)
    torch.cuda.synchronize()
    dist.barrier()                       # every worker's pushes have landed
    gs = t.global_step_value()           # pulls the shards
    loss1, _ = t.evaluate(ds.x_epoch, ds.y_epoch)
    mine = {"params": t.params.cpu(), "gs": gs}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    identical = all(torch.equal(e["params"], everyone[0]["params"]) for e in everyone)
    moved = float((t.params - p0)[4:].abs().max())
    finite = bool(torch.isfinite(t.params).all())
    dist.barrier()
    t.close()
    print(json.dumps({"rank": rank, "identical": identical, "finite": finite, "moved": moved, "global_step": gs,
                      "loss_before": loss0, "loss_after": loss1, "describe": t.describe()}), flush=True)
    dist.destroy_process_group()


def main():
    rank, world, port, steps, use_graph, uncached = (int(v) for v in sys.argv[1:7])
    mode = sys.argv[7] if len(sys.argv) > 7 else "sync"
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, rows, lr = 100, 800, 0.05
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(rows, 784, generator=g)
    y = torch.eye(10)[torch.randint(0, 10, (rows,), generator=g)]
    ds = DeviceDataset(x, y, dev, B, seed=1000 + rank)
    if mode in ("async", "async_sep", "async_auto"):
        return run_async(rank, world, steps, use_graph, dev, ds, B, lr,
                         fused_push="auto" if mode == "async_auto" else mode == "async")
    backup = mode == "sync_backup"          # the first world - 1 gradients of a step count (backup replicas)
    extra = {} if mode in ("sync2", "sync2c", "sync2p") else {"uncached_grads": bool(uncached)}
    if backup:
        extra["required"] = world - 1
    t = build_replica_trainer({"sync2": "xgmi2", "sync2c": "xgmi2c", "sync2p": "xgmi2p"}.get(mode, "xgmi"), device=dev, dataset=ds, batch=B, lr=lr, seed=1, rank=rank, world=world,
                              chunk=3, use_graph=bool(use_graph), timeout_ms=30000, **extra)
    p0 = t.params.clone().cpu()
    members = []
    if backup and not use_graph:
        import time

        dist.barrier()
        for s in range(steps):               # step by step: which replicas counted is read back after every exchange
            if rank == (s % world):
                time.sleep(0.02)             # a straggler per step (who is dropped is still up to the arrival order)
            t.run_steps(1)
            members.append(t.xg.last_members())
    else:
        t.prepare((steps,))
        dist.barrier()
        t.run_steps(steps)
    torch.cuda.synchronize()
    healthy = t.healthy()
    mine = {"params": t.params.cpu(), "x": ds.x_epoch.cpu(), "y": ds.y_epoch.cpu(), "sum": t.replica_checksum(),
            "exchanges": t.xg.exchanges_done(), "members": members}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    identical = all(torch.equal(e["params"], everyone[0]["params"]) for e in everyone)
    same_sum = len({e["sum"] for e in everyone}) == 1
    # fp32 reference on the CPU: the mean of the replicas' gradients applied once per step
    lay = t.layout
    p = p0.clone()
    views = lay.mlp_views(p)
    for s in range(steps):
        acc = {k: torch.zeros_like(v) for k, v in views.items()}
        counted = members[s] if members else list(range(world))
        for r, e in enumerate(everyone):
            if r not in counted:
                continue
            xb, yb = e["x"][s * B:(s + 1) * B], e["y"][s * B:(s + 1) * B]
            out = ref.mlp_grads(xb, yb, views["W1"], views["b1"], views["W2"], views["b2"])
            for k in acc:
                acc[k] += out["d" + k] / len(counted)
        for k in views:
            views[k].sub_(acc[k], alpha=lr)
    err = float((mine["params"][4:] - p[4:]).abs().max())
    scale = float(p[4:].abs().max())
    moved = float((p - p0)[4:].abs().max())
    gs = t.global_step_value()
    dist.barrier()
    t.close()
    print(json.dumps({"rank": rank, "healthy": healthy, "identical": identical, "same_sum": same_sum,
                      "max_err": err, "scale": scale, "moved": moved, "global_step": gs,
                      "exchanges": mine["exchanges"], "describe": t.describe(), "members": members,
                      "same_members": all(e["members"] == members for e in everyone)}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
