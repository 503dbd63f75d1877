"""Rank body of the data-parallel dense-stack test (launched with torch.distributed.run)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from distributed_tensorflow_amd.data import DeviceDataset  # noqa: E402
from distributed_tensorflow_amd.parallel.dense_trainer import DenseDataParallelTrainer  # noqa: E402


def main():
    bucket_mb = float(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = torch.device("cpu")
    dist.init_process_group("gloo")
    rng = np.random.RandomState(7)
    labels = rng.randint(0, 10, 600)
    images = (rng.rand(10, 64).astype(np.float32)[labels] + 0.3 * rng.rand(600, 64)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]
    ds = DeviceDataset(images, onehot, device, 50, seed=100 + rank)
    tr = DenseDataParallelTrainer((64, 48, 32, 10), ("relu", "tanh"), ds, 50, 0.2, device, rank, world, seed=3,
                                  bucket_mb=bucket_mb)
    # the oracle: ONE process that sees both ranks' batches and applies the mean gradient (recomputed on rank 0)
    p0 = tr.params.clone()
    first_batches = [None] * world
    dist.all_gather_object(first_batches, (ds.x_epoch[:50].clone(), ds.y_epoch[:50].clone()))
    tr.run_steps(1)
    after_one = tr.params.clone()
    tr.run_steps(13)            # 12 batches per epoch: crosses a reshuffle
    gathered = [torch.zeros_like(tr.params) for _ in range(world)]
    dist.all_gather(gathered, tr.params)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    loss, acc = tr.evaluate(torch.as_tensor(images), torch.as_tensor(onehot))
    if rank == 0:
        # mean gradient of the two first batches by autograd, in the slab's order
        from distributed_tensorflow_amd.parallel.dense_trainer import DenseDataParallelTrainer as T  # noqa: F401
        ref = T((64, 48, 32, 10), ("relu", "tanh"), ds, 50, 0.2, device, 0, 1, seed=3, bucket_mb=bucket_mb)
        assert torch.equal(ref.params, p0)
        leaves = ref.params.clone().double().requires_grad_(True)
        total = 0.0
        for x, y in first_batches:
            h = x.double()
            for i in range(3):
                a, b = ref._range[("W", i)]
                c, d = ref._range[("b", i)]
                n_in, n_out = ref.engine.dims[i], ref.engine.dims[i + 1]
                z = h @ leaves[a:a + n_in * n_out].view(n_in, n_out) + leaves[c:c + n_out]
                h = (torch.relu, torch.tanh)[i](z) if i < 2 else z
            total = total + (-(y.double() * torch.log_softmax(h, 1)).sum(1).mean()) / world
        total.backward()
        want = (p0.double() - 0.2 * leaves.grad).float()
        err = float((after_one - want).abs().max())
        print("RESULT " + json.dumps({"replicas_identical": same, "global_step": tr.global_step_value(), "loss": loss,
                                      "acc": acc, "one_step_error": err, "buckets": len(tr.buckets),
                                      "desc": tr.describe()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
