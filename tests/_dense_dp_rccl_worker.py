"""One rank, a real RCCL communicator: the data-parallel dense-stack trainer issues its bucket all-reduces through it
(side stream, events against the backward on the main stream) and must end bit-identical to the same run without."""
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
    port = int(sys.argv[1])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}", device_id=device)
    rng = np.random.RandomState(7)
    labels = rng.randint(0, 10, 2400)
    images = (rng.rand(10, 784).astype(np.float32)[labels] + 0.3 * rng.rand(2400, 784)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]
    runs = []
    for group, use_graph in ((dist.group.WORLD, False), (None, False), (dist.group.WORLD, True)):
        ds = DeviceDataset(images, onehot, device, 200, seed=5)
        tr = DenseDataParallelTrainer((784, 1024, 512, 10), ("relu", "tanh"), ds, 200, 0.05, device, seed=3, bucket_mb=1.0,
                                      group=group, use_graph=use_graph, chunk=5)
        tr.run_steps(30)            # 12 batches per epoch: two reshuffles
        torch.cuda.synchronize()
        runs.append((tr.params.clone(), tr.global_step_value(), tr.last_loss(), len(tr.buckets), tr.describe(),
                     sorted(tr._graphs), tr.use_graph))
        tr.close()
    print("RESULT " + json.dumps({"identical": bool(torch.equal(runs[0][0], runs[1][0])), "global_step": runs[0][1],
                                  "loss": runs[0][2], "loss_without": runs[1][2], "buckets": runs[0][3], "desc": runs[0][4],
                                  "backend": dist.get_backend(),
                                  "graph_identical": bool(torch.equal(runs[0][0], runs[2][0])), "graph_step": runs[2][1],
                                  "graph_keys": runs[2][5], "graph_used": runs[2][6], "graph_desc": runs[2][4]}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
