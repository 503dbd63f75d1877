import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from distributed_tensorflow_amd.ops import hip_backend as hb
import probe_kernels as P
lib = hb.require()
lib.dtf_debug_set_pipe_parts.argtypes = [C.c_int]
for parts in (3, 1, 2, 0, 3):
    lib.dtf_debug_set_pipe_parts(3)
    eng = P.make_engine()
    nb, B = eng.batches_per_epoch, eng.B
    xb = eng.x[:nb * B].view(nb, B, -1)
    gram = torch.zeros(nb - 1, B, 112, device="cuda")
    gram[:, :, :B] = 1.0 + torch.bmm(xb[1:], xb[:-1].transpose(1, 2))
    eng.bind_gram(gram)
    eng.chunk_pipelined(0, 2)
    torch.cuda.synchronize()
    lib.dtf_debug_set_pipe_parts(parts)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.chunk_pipelined(0, 200)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print("parts", parts, "per step: %.3f us" % (e0.elapsed_time(e1) * 1000 / (20 * 200)), flush=True)
lib.dtf_debug_set_pipe_parts(3)
