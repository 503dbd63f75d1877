"""Fixed cost of one hipGraph chunk of the one-launch-per-step path (memset + stand-alone forward + graph launch):
replays chunks of 200 / 50 / 10 / 2 steps and derives the per-chunk constant; also cycles 11 distinct graphs."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import probe_kernels as P
def run(n, reps):
    eng = P.make_engine()
    nb, B = eng.batches_per_epoch, eng.B
    xb = eng.x[:nb * B].view(nb, B, -1)
    gram = torch.zeros(nb - 1, B, 112, device="cuda")
    gram[:, :, :B] = 1.0 + torch.bmm(xb[1:], xb[:-1].transpose(1, 2))
    eng.bind_gram(gram)
    eng.chunk_pipelined(0, 2); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.chunk_pipelined(0, n)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    print("chunk of %3d steps: %8.1f us per replay, %.3f us per step" % (n, us, us / n), flush=True)
    return us
a = run(200, 20); b = run(50, 80); c = run(10, 200); d = run(2, 200)
print("per-chunk fixed cost ~ %.1f us (from 200 vs 50), %.1f us (50 vs 10)" % ((b - a * 50 / 200) / (1 - 50 / 200), (c - b * 10 / 50) / (1 - 10 / 50)))
eng = P.make_engine()
nb, B = eng.batches_per_epoch, eng.B
xb = eng.x[:nb * B].view(nb, B, -1)
gram = torch.zeros(nb - 1, B, 112, device="cuda")
gram[:, :, :B] = 1.0 + torch.bmm(xb[1:], xb[:-1].transpose(1, 2))
eng.bind_gram(gram)
eng.chunk_pipelined(0, 2); torch.cuda.synchronize()
graphs = []
for pos in range(0, 550, 50):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.chunk_pipelined(pos, 50)
    graphs.append(g)
for g in graphs: g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    for g in graphs: g.replay()
e1.record(); torch.cuda.synchronize()
print("11 different graphs cycled: %.3f us per step" % (e0.elapsed_time(e1) * 1000 / (10 * 550)))
