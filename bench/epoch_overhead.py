"""Step time of the device-resident single-GPU trainer inside an epoch and across epoch boundaries (the reshuffle
gather and the Gram-block GEMM run between epochs)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import DeviceDataset, read_data_sets
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer
dev = torch.device("cuda")
mnist = read_data_sets("MNIST_data", one_hot=True, seed=0)
ds = DeviceDataset(mnist.train.images, mnist.train.labels, dev, 100, seed=1000)
tr = build_replica_trainer(mode="single", device=dev, dataset=ds, batch=100, lr=0.001, seed=1, rank=0, world=1, chunk=50, use_graph=True)
tr.prepare(steps_hint=(550, 5500, 550))
tr.run_steps(550); torch.cuda.synchronize()
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e6
# inside one epoch: 500 steps without touching the boundary
tr.run_steps(50)      # crosses into the next epoch (shuffle happens here)
print("500 steps inside an epoch: %.3f us/step" % (timed(lambda: tr.run_steps(500)) / 500))
print("50 steps incl. epoch refresh: %.1f us total" % timed(lambda: tr.run_steps(50)))
print("5500 steps (10 epochs): %.3f us/step" % (timed(lambda: tr.run_steps(5500)) / 5500))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); tr.run_steps(5500); e1.record(); torch.cuda.synchronize()
print("5500 steps, event timed: %.3f us/step" % (e0.elapsed_time(e1) * 1000 / 5500))
print(tr.describe())
