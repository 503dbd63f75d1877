"""Rank body for the collective-trainer tests (launched with torch.distributed.run)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from distributed_tensorflow_amd.data import DeviceDataset, read_data_sets  # noqa: E402
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer  # noqa: E402


def main():
    mode, device_kind, use_graph = sys.argv[1], sys.argv[2], sys.argv[3] == "graph"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if device_kind == "cuda":
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        dist.init_process_group("nccl", device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo")
    mnist = read_data_sets("MNIST_data", one_hot=True, sizes=(1100, 100, 1000))
    ds = DeviceDataset(mnist.train.images, mnist.train.labels, device, 100, seed=100 + rank)
    tr = build_replica_trainer(mode=mode, device=device, dataset=ds, batch=100, lr=0.05, seed=1, rank=rank,
                               world=world, chunk=4, use_graph=use_graph)
    tr.prepare(steps_hint=(13,))
    tr.run_steps(13)          # crosses an epoch boundary (11 batches per epoch)
    global_step = tr.global_step_value()      # (a collective for the point-to-point transport: every rank calls it)
    params = tr.params.clone()
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    tx, ty = torch.as_tensor(mnist.test.images).to(device), torch.as_tensor(mnist.test.labels).to(device)
    loss, acc = tr.evaluate(tx, ty)
    if rank == 0:
        print("RESULT " + json.dumps({"mode": mode, "world": world, "replicas_identical": same,
                                      "global_step": global_step, "served": getattr(tr, "served", None), "loss": loss, "acc": acc,
                                      "checksum": float(params.double().sum()), "desc": tr.describe()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
