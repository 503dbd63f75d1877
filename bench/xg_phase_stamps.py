"""Phase stamps (ns, thread 0 of one workgroup) of the backward launch with ONE replica: the plain fused backward
(`single`, two launches per step) next to the cross-replica one (`xgmi2`), for a main tile workgroup, another one and
the statistics workgroup.  Shows what the exchange machinery costs before any peer traffic.

    python bench/xg_phase_stamps.py
"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import DeviceDataset
from distributed_tensorflow_amd.ops import hip_backend as hb
from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.rand(5500, 784, generator=g); y = torch.eye(10)[torch.randint(0, 10, (5500,), generator=g)]
ds = DeviceDataset(x, y, dev, 100, seed=1)
lib = hb.require()
lib.dtf_debug_set_bwd_stamp_block.argtypes = [C.c_int]; lib.dtf_debug_set_bwd_stamp_block.restype = None
lib.dtf_debug_set_bwd_stamps.argtypes = [C.c_void_p]; lib.dtf_debug_set_bwd_stamps.restype = None
names = ["start", "loads issued", "stage+softmax", "contraction", "partials+sync", "reduce(+store)", "db1", "end(+exchange)"]
for mode, extra in (("single", {"pipelined": False}), ("xgmi2", {})):
    t = build_replica_trainer(mode, device=dev, dataset=ds, batch=100, lr=0.001, seed=1, rank=0, world=1, chunk=50, use_graph=False, **extra)
    for block in (1, 30, -1):
        lib.dtf_debug_set_bwd_stamp_block(block)
        stamps = torch.zeros(32, dtype=torch.int64, device=dev)
        t.run_steps(6); torch.cuda.synchronize()
        lib.dtf_debug_set_bwd_stamps(stamps.data_ptr())
        t.run_steps(4); torch.cuda.synchronize()
        lib.dtf_debug_set_bwd_stamps(None)
        s = stamps.cpu().view(16, 2)
        print(mode, "block", block, " ".join("%s=%d" % (names[i], (s[i, 0] - s[i - 1, 0]) * 10) for i in range(1, 8)), "total=%d ns" % ((s[7, 0] - s[0, 0]) * 10), flush=True)
    if hasattr(t, "close"): t.close()
