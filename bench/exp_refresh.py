import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import DeviceDataset
x = np.random.rand(55000, 784).astype(np.float32); y = np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 55000)]
ds = DeviceDataset(x, y, "cuda", 100, seed=1)
ds.enable_pair_gram()
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("shuffle_epoch total  %.3f ms" % t(ds.shuffle_epoch))
s = ds._sets[0]
perm = torch.randperm(55000).cuda()
from distributed_tensorflow_amd.ops import hip_backend as hb
print("randperm host        %.3f ms" % t(lambda: torch.randperm(55000)))
print("perm H2D             %.3f ms" % t(lambda: torch.randperm(55000).to("cuda")))
print("gather x             %.3f ms" % t(lambda: hb.gather_rows(ds.x, perm, s["x"])))
print("gather y             %.3f ms" % t(lambda: hb.gather_rows(ds.y, perm, s["y"])))
xb = s["x"].view(550, 100, 784)
ones = torch.ones(1, 1, 1, device="cuda")
print("baddbmm              %.3f ms" % t(lambda: torch.baddbmm(ones, xb[1:], xb[:-1].transpose(1, 2), out=s["tmp"])))
print("copy into padded     %.3f ms" % t(lambda: s["gram"][:, :, :100].copy_(s["tmp"])))
