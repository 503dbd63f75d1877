"""Cost of one epoch refresh of the resident training set (gather + Gram blocks), per kernel under rocprofv3 or
by device events:  python bench/refresh_cost.py [reps]   (DTF_GRAM_KERNEL=0: the library path)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from distributed_tensorflow_amd.data import DeviceDataset, read_data_sets  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    mnist = read_data_sets("MNIST_data", one_hot=True, seed=0)
    ds = DeviceDataset(mnist.train.images, mnist.train.labels, dev, 100, seed=1000)
    ds.enable_pair_gram()
    for _ in range(3):
        ds.shuffle_epoch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ds.shuffle_epoch()
    e1.record()
    torch.cuda.synchronize()
    print(f"fused={ds._fused_refresh()} refresh {e0.elapsed_time(e1) / reps * 1000:.1f} us per epoch", flush=True)


if __name__ == "__main__":
    main()
