"""Per-kernel probes: event-timed launch loops and in-kernel phase stamps of the head kernel.

    python bench/probe_kernels.py            (on the GPU box)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.models import MLPEngine  # noqa: E402
from distributed_tensorflow_amd.ops import hip_backend as hb  # noqa: E402


def make_engine(B=100):
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(1)
    p = {"W1": torch.randn(784, 100, generator=g), "b1": torch.zeros(100),
         "W2": torch.randn(100, 10, generator=g), "b2": torch.zeros(10)}
    p = {k: v.to(dev) for k, v in p.items()}
    grads = {k: torch.zeros_like(v) for k, v in p.items()}
    eng = MLPEngine(p, grads, batch=B, lr=0.001, global_step=torch.zeros(1, device=dev))
    x = torch.rand(550 * B, 784, generator=g).to(dev)
    y = torch.nn.functional.one_hot(torch.randint(0, 10, (550 * B,), generator=g), 10).float().to(dev)
    eng.bind_data(x, y)
    return eng


def timed(fn, iters=300, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


def main():
    eng = make_engine()
    lib = hb.require()
    d = eng._descriptor()
    x, W1, b1 = eng.x[:100], eng.p["W1"], eng.p["b1"]
    z = torch.zeros(100, 100, device="cuda")
    dz = torch.randn(100, 100, device="cuda") * 0.01
    out = torch.zeros(784, 100, device="cuda")
    empty = torch.zeros(1, device="cuda")
    print("us per launch, graph-replayed back to back:")
    print("  empty torch fill      %.2f" % timed(lambda: empty.zero_()))
    for ks in (1, 4, 7, 13):
        if ks == 1:
            f = lambda: hb.linear_bias_act(x, W1, b1, 0, out=z)
        else:
            f = lambda ks=ks: hb.linear_splitk(x, W1, ks, out=z)
        print("  fwd gemm ksplit=%-2d     %.2f" % (ks, timed(f)))
    print("  dw gemm (store)       %.2f" % timed(lambda: hb.dw_gemm(x, dz, out=out, mode=0)))
    print("  dw gemm (sgd)         %.2f" % timed(lambda: hb.dw_gemm(x, dz, out=out, mode=1, lr=eng.lr)))
    hb.force_generic_path(True)
    print("  step_grad (3 kernels) %.2f" % timed(lambda: hb.mlp_step_grad(d, 0, 0, 1.0)))
    print("  step_fused(3 kernels) %.2f" % timed(lambda: hb.mlp_step_fused(d, 0, 0)))
    hb.force_generic_path(False)
    par = [0]

    def alt(fn):
        def run():
            fn(par[0])
            par[0] ^= 1
        return run

    print("  step_grad (2 kernels) %.2f" % timed(alt(lambda q: hb.mlp_step_grad(d, 0, q, 1.0)), iters=300, warm=20))
    eng.zws.zero_(); par[0] = 0
    print("  step_fused(2 kernels) %.2f" % timed(alt(lambda q: hb.mlp_step_fused(d, 0, q)), iters=300, warm=20))
    eng.zws.zero_()
    p = torch.zeros(79512, device="cuda")
    gg = torch.zeros(79512, device="cuda")
    print("  sgd_apply 79.5k       %.2f" % timed(lambda: hb.sgd_apply_(p, gg, eng.lr)))

    stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
    lib.dtf_debug_set_head_stamps(stamps.data_ptr())
    hb.force_generic_path(True)
    for _ in range(3):
        hb.mlp_step_fused(d, 0, 0)
    torch.cuda.synchronize()
    hb.force_generic_path(False)
    lib.dtf_debug_set_head_stamps(None)
    s = stamps.cpu().view(16, 2)
    names = ["start", "loads issued", "small staged+sync", "a2 staged+sync", "phase1 logits/softmax",
             "dW2 tiles (wave0)", "phase2 done+sync", "end"]
    print("head kernel phases (thread 0): wall ns (100 MHz clock) / shader cycles")
    for i in range(1, 8):
        print("  %-24s %7d ns %8d cyc" % (names[i], (s[i, 0] - s[i - 1, 0]) * 10, s[i, 1] - s[i - 1, 1]))
    print("  %-24s %7d ns %8d cyc" % ("total", (s[7, 0] - s[0, 0]) * 10, s[7, 1] - s[0, 1]))


def bwd_stamps(block=1, label="main block 1"):
    eng = make_engine()
    lib = hb.require()
    lib.dtf_debug_set_bwd_stamp_block.argtypes = [C.c_int]
    lib.dtf_debug_set_bwd_stamp_block.restype = None
    lib.dtf_debug_set_bwd_stamp_block(block)
    d = eng._descriptor()
    stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
    lib.dtf_debug_set_bwd_stamps.argtypes = [C.c_void_p]
    lib.dtf_debug_set_bwd_stamps.restype = None
    lib.dtf_debug_set_bwd_stamps(stamps.data_ptr())
    for i in range(4):
        hb.mlp_step_fused(d, 0, i & 1)
    torch.cuda.synchronize()
    lib.dtf_debug_set_bwd_stamps(None)
    s = stamps.cpu().view(16, 2)
    names = ["start", "loads issued", "stage+softmax (2 syncs)", "contraction loop", "partials to LDS+sync",
             "reduce+store", "db1", "end"]
    print("bwd_all phases (%s, thread 0): wall ns / shader cycles" % label)
    for i in range(1, 8):
        print("  %-24s %7d ns %8d cyc" % (names[i], (s[i, 0] - s[i - 1, 0]) * 10, s[i, 1] - s[i - 1, 1]))
    print("  %-24s %7d ns %8d cyc" % ("total", (s[7, 0] - s[0, 0]) * 10, s[7, 1] - s[0, 1]))


def block_stamps():
    eng = make_engine()
    lib = hb.require()
    d = eng._descriptor()
    n = 2 * (256 + 256)
    stamps = torch.zeros(n, dtype=torch.int64, device="cuda")
    lib.dtf_debug_set_block_stamps.argtypes = [C.c_void_p]
    lib.dtf_debug_set_block_stamps.restype = None
    g = torch.cuda.CUDAGraph()
    for q in (0, 1):
        hb.mlp_step_fused(d, 0, q)
    torch.cuda.synchronize()
    lib.dtf_debug_set_block_stamps(stamps.data_ptr())
    with torch.cuda.graph(g):
        for i in range(6):
            hb.mlp_step_fused(d, 0, i & 1)
    lib.dtf_debug_set_block_stamps(None)
    g.replay()
    torch.cuda.synchronize()
    s = stamps.cpu().view(-1, 2)
    nf = int((s[:256, 0] != 0).sum())
    nb = int((s[256:, 0] != 0).sum())
    f, b = s[:nf], s[256:256 + nb]   # 100 dW1 tiles + 4 dW2 tiles + 1 statistics block
    t0 = int(f[:, 0].min())
    rel = lambda x: (int(x) - t0) * 10   # noqa: E731
    print("grid timing of the LAST captured step (ns, relative to the first forward block start):")
    print("  fwd  blocks start %5d..%5d   end %5d..%5d" % (rel(f[:, 0].min()), rel(f[:, 0].max()), rel(f[:, 1].min()), rel(f[:, 1].max())))
    print("  bwd  blocks start %5d..%5d   end %5d..%5d" % (rel(b[:, 0].min()), rel(b[:, 0].max()), rel(b[:, 1].min()), rel(b[:, 1].max())))
    dur = (b[:, 1] - b[:, 0]) * 10
    print("  bwd  block durations: min %d median %d max %d; last 8 blocks (dW2 tiles, statistics): %s" % (int(dur.min()), int(dur.median()), int(dur.max()), [int(x) for x in dur[-8:]]))
    durf = (f[:, 1] - f[:, 0]) * 10
    print("  fwd  block durations: min %d median %d max %d" % (int(durf.min()), int(durf.median()), int(durf.max())))
    print("  fwd  per block (id: start/end): " + " ".join("%d:%d/%d" % (i, rel(f[i, 0]), rel(f[i, 1])) for i in range(nf)))
    print("  bwd  per block (id: start/end): " + " ".join("%d:%d/%d" % (i, rel(b[i, 0]), rel(b[i, 1])) for i in range(nb)))


def pipe_block_stamps():
    """Per-block timeline of one launch of the one-launch-per-step kernel (forward tiles of step t+1 +
    backward tiles of step t)."""
    eng = make_engine()
    lib = hb.require()
    nb, B = eng.batches_per_epoch, eng.B
    xb = eng.x[:nb * B].view(nb, B, -1)
    gram = torch.zeros(nb - 1, B, (B + 15) // 16 * 16, device="cuda")
    gram[:, :, :B] = 1.0 + torch.bmm(xb[1:], xb[:-1].transpose(1, 2))
    eng.bind_gram(gram)
    stamps = torch.zeros(2 * 1024, dtype=torch.int64, device="cuda")
    lib.dtf_debug_set_block_stamps.argtypes = [C.c_void_p]
    lib.dtf_debug_set_block_stamps.restype = None
    eng.chunk_pipelined(0, 2)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    lib.dtf_debug_set_block_stamps(stamps.data_ptr())
    with torch.cuda.graph(g):
        eng.chunk_pipelined(0, 6)
    lib.dtf_debug_set_block_stamps(None)
    g.replay()
    torch.cuda.synchronize()
    s = stamps.cpu().view(-1, 2)
    na = int(os.environ.get("DTF_PROBE_NA", "5"))
    nd = (784 + 16 * na - 1) // (16 * na) * 7          # dW1 tiles
    nbk = nd + 7 + 1                                    # + dW2 tiles, statistics
    b, f = s[256:256 + nbk], s[768:768 + 169]
    t0 = int(min(f[:, 0].min(), b[:, 0].min()))
    rel = lambda x: (int(x) - t0) * 10   # noqa: E731
    print("one-launch step, second-to-last launch of a 6-step chunk (ns from the first block start):")
    print("  fwd tiles  start %5d..%5d   end %5d..%5d" % (rel(f[:, 0].min()), rel(f[:, 0].max()), rel(f[:, 1].min()), rel(f[:, 1].max())))
    print("  dW1 tiles  start %5d..%5d   end %5d..%5d" % (rel(b[:nd, 0].min()), rel(b[:nd, 0].max()), rel(b[:nd, 1].min()), rel(b[:nd, 1].max())))
    print("  dW2 tiles + statistics (start/end): " + " ".join("%d/%d" % (rel(b[i, 0]), rel(b[i, 1])) for i in range(nd, nbk)))
    durf = (f[:, 1] - f[:, 0]) * 10
    durb = (b[:nd, 1] - b[:nd, 0]) * 10
    print("  durations: fwd tiles min %d median %d max %d; dW1 tiles min %d median %d max %d" % (
        int(durf.min()), int(durf.median()), int(durf.max()), int(durb.min()), int(durb.median()), int(durb.max())))
    late = b[:nd][b[:nd, 0] > b[:nd, 0].min() + 100]  # started more than 1 us after the first: second round on their CU
    if len(late):
        print("  dW1 tiles of the second round: %d, start %d..%d, end %d..%d" % (
            len(late), rel(late[:, 0].min()), rel(late[:, 0].max()), rel(late[:, 1].min()), rel(late[:, 1].max())))
    ends = sorted([rel(x) for x in f[:, 1]] + [rel(x) for x in b[:, 1]])
    print("  all tiles: 50%% ended by %d, 90%% by %d, 99%% by %d, last %d" % (
        ends[len(ends) // 2], ends[len(ends) * 9 // 10], ends[len(ends) * 99 // 100], ends[-1]))


def pipe_fwd_stamps(block=512 + 30, names=None):
    """Phase stamps of one forward tile of the one-launch step (block < 512: of a backward tile)."""
    eng = make_engine()
    lib = hb.require()
    nb, B = eng.batches_per_epoch, eng.B
    xb = eng.x[:nb * B].view(nb, B, -1)
    gram = torch.zeros(nb - 1, B, (B + 15) // 16 * 16, device="cuda")
    gram[:, :, :B] = 1.0 + torch.bmm(xb[1:], xb[:-1].transpose(1, 2))
    eng.bind_gram(gram)
    stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
    lib.dtf_debug_set_bwd_stamps.argtypes = [C.c_void_p]
    lib.dtf_debug_set_bwd_stamp_block.argtypes = [C.c_int]
    eng.chunk_pipelined(0, 2)
    torch.cuda.synchronize()
    lib.dtf_debug_set_bwd_stamps(stamps.data_ptr())
    lib.dtf_debug_set_bwd_stamp_block(block)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.chunk_pipelined(0, 6)
    lib.dtf_debug_set_bwd_stamps(None)
    lib.dtf_debug_set_bwd_stamp_block(1)
    g.replay()
    torch.cuda.synchronize()
    s = stamps.cpu().view(16, 2)
    if names is None:
        names = ["start", "offsets + loads issued", "stage + softmax (2 syncs)", "contraction + dz2 + a2^T dz3 (+sync)",
                 "correction + partials to LDS (+sync)", "reduce + activation + store (+sync)", "logit partials + atomics", "end"]
        print("one-launch step, forward tile %d (thread 0): wall ns / shader cycles" % (block - 512))
    else:
        print("one-launch step, backward tile %d (thread 0): wall ns / shader cycles" % block)
    for i in range(1, 8):
        print("  %-40s %7d ns %8d cyc" % (names[i], (s[i, 0] - s[i - 1, 0]) * 10, s[i, 1] - s[i - 1, 1]))
    print("  %-40s %7d ns" % ("total", (s[7, 0] - s[0, 0]) * 10))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pipe":
        pipe_block_stamps()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bwd":
        for blk in (8, 40):
            pipe_fwd_stamps(blk, ["start", "loads issued", "stage+softmax (2 syncs)", "contraction loop",
                                  "partials to LDS+sync", "reduce+store", "db1", "end"])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fwd":
        for blk in (512 + 30, 512 + 100, 512 + 160):
            pipe_fwd_stamps(blk)
        sys.exit(0)
    pipe_fwd_stamps()
    pipe_block_stamps()
    block_stamps()
    bwd_stamps(1, "main block 1")
    bwd_stamps(-1, "stats block: 3=softmax done,4=db2,5=hist stores,6=global_step rmw")
    sys.exit(0)
    main()
