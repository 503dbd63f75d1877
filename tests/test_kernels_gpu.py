"""Every HIP kernel against a plain PyTorch fp32 oracle of the same op."""
import pytest
import torch

from distributed_tensorflow_amd.models import MLPEngine
from distributed_tensorflow_amd.ops import hip_backend as hb
from distributed_tensorflow_amd.ops import reference as ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def test_library_loaded():
    assert hb.available(), "libdtf_hip.so must be loadable on a GPU box"
    assert "gfx950" in hb.build_info()


@pytest.mark.parametrize("M,K,H", [(100, 784, 100), (10000, 784, 100), (1, 8, 1), (33, 50, 37), (128, 785, 130)])
@pytest.mark.parametrize("act", [ref.ACT_SIGMOID, ref.ACT_RELU])
def test_linear_bias_act(M, K, H, act):
    x, W, b = _rand(M, K, seed=1), _rand(K, H, scale=0.1, seed=2), _rand(H, seed=3)
    out = hb.linear_bias_act(x, W, b, act)
    want = ref.linear_bias_act(x.double(), W.double(), b.double(), act).float()
    torch.testing.assert_close(out, want, rtol=2e-5, atol=2e-5)


def test_mfma_orientation_asymmetric():
    # identity A with an asymmetric B exposes a transposed tile write
    K = H = 64
    x = torch.eye(K, device=DEV)
    W = (torch.arange(K * H, device=DEV, dtype=torch.float32).reshape(K, H)) / (K * H)
    out = hb.linear_bias_act(x, W, torch.zeros(H, device=DEV), ref.ACT_RELU)
    torch.testing.assert_close(out, W)


@pytest.mark.parametrize("ksplit", [2, 7, 13])
def test_linear_splitk_atomic(ksplit):
    x, W = _rand(100, 784, seed=4), _rand(784, 100, seed=5)
    out = hb.linear_splitk(x, W, ksplit)
    want = (x.double() @ W.double()).float()
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,K,H", [(100, 784, 100), (7, 33, 5), (600, 784, 100)])
def test_dw_gemm_modes(M, K, H):
    x, dz = _rand(M, K, seed=6), _rand(M, H, scale=0.01, seed=7)
    want = (x.double().t() @ dz.double()).float()
    got = hb.dw_gemm(x, dz, mode=0, scale=0.5)
    torch.testing.assert_close(got, 0.5 * want, rtol=1e-4, atol=1e-5)
    W = _rand(K, H, seed=8)
    W0 = W.clone()
    lr = torch.tensor([0.1], device=DEV)
    hb.dw_gemm(x, dz, out=W, mode=1, lr=lr)
    torch.testing.assert_close(W, W0 - 0.1 * want, rtol=1e-4, atol=1e-5)
    acc = torch.ones(K, H, device=DEV)
    hb.dw_gemm(x, dz, out=acc, mode=2)
    torch.testing.assert_close(acc, 1.0 + want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n", [4, 79510, 1000003])
def test_sgd_apply_and_accumulate(n):
    p, g = _rand(n, seed=9), _rand(n, seed=10)
    p0, g0 = p.clone(), g.clone()
    lr = torch.tensor([0.25], device=DEV)
    gs = torch.zeros(1, device=DEV)
    hb.sgd_apply_(p, g, lr, scale=0.5, zero_grad=True, global_step=gs)
    torch.testing.assert_close(p, p0 - 0.125 * g0)
    assert float(g.abs().max()) == 0.0 and float(gs) == 1.0
    acc = _rand(n, seed=11)
    a0 = acc.clone()
    hb.accumulate_(acc, g0)
    torch.testing.assert_close(acc, a0 + g0)


def test_philox_normal_matches_host_and_is_offset_consistent():
    out = torch.empty(64, device=DEV)
    hb.philox_normal_(out, seed=1, offset=0)
    host = ref.philox_normal_host(64, seed=1)
    torch.testing.assert_close(out.cpu(), host, rtol=1e-3, atol=1e-3)
    part = torch.empty(13, device=DEV)
    hb.philox_normal_(part, seed=1, offset=21)
    torch.testing.assert_close(part, out[21:34])
    big = torch.empty(784 * 100, device=DEV)
    hb.philox_normal_(big, seed=1, offset=0, mean=0.0, stddev=1.0)
    assert abs(float(big.mean())) < 0.02 and abs(float(big.std()) - 1.0) < 0.02


def test_gather_rows():
    src = _rand(1000, 784, seed=12)
    idx = torch.randperm(1000).to(DEV)
    torch.testing.assert_close(hb.gather_rows(src, idx), src[idx])
    src2 = _rand(100, 10, seed=13)
    idx2 = torch.randperm(100).to(DEV)
    torch.testing.assert_close(hb.gather_rows(src2, idx2), src2[idx2])


def _engine(device, B=100, K=784, H=100, C=10, act="sigmoid", with_grads=True, scale=0.1, nb=3):
    g = torch.Generator().manual_seed(42)
    p = {"W1": torch.randn(K, H, generator=g) * scale, "b1": torch.randn(H, generator=g) * scale,
         "W2": torch.randn(H, C, generator=g), "b2": torch.randn(C, generator=g) * scale}
    p = {k: v.to(device) for k, v in p.items()}
    grads = {k: torch.zeros_like(v) for k, v in p.items()} if with_grads else None
    gs = torch.zeros(1, device=device)
    eng = MLPEngine(p, grads, batch=B, lr=0.05, act=act, global_step=gs, hist_cap=64)
    x = torch.rand(nb * B, K, generator=g)
    lab = torch.randint(0, C, (nb * B,), generator=g)
    y = torch.nn.functional.one_hot(lab, C).float()
    eng.bind_data(x.to(device), y.to(device))
    return eng


@pytest.fixture(params=["two_launch", "two_launch_fwd16", "generic"])
def step_path(request):
    """Every step implementation: the 2-launch small-batch path with the default 8x8 forward tiles and with the
    generic 16x16-tile forward kernel (the fallback for K not a multiple of 4), and the generic 3-launch path."""
    hb.force_generic_path(request.param == "generic")
    hb.set_forward_tile8(request.param != "two_launch_fwd16")
    yield request.param
    hb.force_generic_path(False)
    hb.set_forward_tile8(True)


@pytest.mark.parametrize("B,K,H,C,act", [(100, 784, 100, 10, "sigmoid"), (100, 784, 100, 10, "relu"),
                                         (37, 50, 24, 3, "sigmoid"), (128, 64, 128, 16, "relu"),
                                         (1, 20, 7, 2, "sigmoid"), (300, 784, 100, 10, "sigmoid")])
def test_step_grad_matches_reference(B, K, H, C, act, step_path):
    eng = _engine(DEV, B, K, H, C, act)
    eng.step_grad(2, scale=0.5)
    torch.cuda.synchronize()
    xb, yb = eng.x[2 * B:3 * B].double(), eng.y[2 * B:3 * B].double()
    want = ref.mlp_grads(xb, yb, *(eng.p[k].double() for k in ("W1", "b1", "W2", "b2")), eng.act)
    for k in ("W1", "b1", "W2", "b2"):
        torch.testing.assert_close(eng.g[k], 0.5 * want["d" + k].float(), rtol=5e-4, atol=2e-6, msg=lambda m: f"{k}: {m}")
    slot = 2 % eng.hist_cap
    assert abs(float(eng.loss_hist[slot]) - float(want["loss"])) < 1e-4
    assert abs(float(eng.acc_hist[slot]) - float(want["acc"])) < 1e-6
    if step_path == "generic" and B == 100:
        assert float(eng.z2.abs().max()) == 0.0, "split-K accumulator must be recycled to zero"


def test_fused_steps_track_cpu_reference(step_path):
    gpu = _engine(DEV, with_grads=False)
    cpu = _engine("cpu", with_grads=False)
    for e in (gpu, cpu):
        e.run_chunk(3, use_graph=False)
        e.run_chunk(2, use_graph=False)
    for k in MLPEngine.PARAM_NAMES:
        torch.testing.assert_close(gpu.p[k].cpu(), cpu.p[k], rtol=1e-4, atol=1e-5)
    assert float(gpu.global_step) == 5.0 and gpu._host_cursor == 5
    # 3 batches per epoch: positions 0,1,2,0,1 -> the ring holds the losses of steps 3,4,2
    torch.testing.assert_close(gpu.loss_hist[:3].cpu(), cpu.loss_hist[:3], rtol=1e-4, atol=1e-5)


def test_graph_replay_equals_eager(step_path):
    a = _engine(DEV, with_grads=False)
    b = _engine(DEV, with_grads=False)
    a.run_chunk(3, use_graph=True)
    a.run_chunk(3, use_graph=True)
    b.run_chunk(6, use_graph=False)
    torch.cuda.synchronize()
    for k in MLPEngine.PARAM_NAMES:
        # both paths accumulate partial sums with f32 atomics, whose order varies run to run
        torch.testing.assert_close(a.p[k], b.p[k], rtol=1e-4, atol=1e-6)
    assert a._host_cursor == 6 and float(a.global_step) == 6.0


def test_eval_matches_reference():
    eng = _engine(DEV, with_grads=False)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(10000, 784, generator=g).to(DEV)
    y = torch.nn.functional.one_hot(torch.randint(0, 10, (10000,), generator=g), 10).float().to(DEV)
    loss, acc = eng.evaluate(x, y)
    wl, wa = ref.mlp_loss_acc(x.double(), y.double(), *(eng.p[k].double() for k in ("W1", "b1", "W2", "b2")), eng.act)
    assert abs(loss - float(wl)) < 1e-3 * max(1.0, float(wl))
    assert abs(acc - float(wa)) < 1e-4


@pytest.mark.parametrize("M,K,H,act,ncls", [(10000, 784, 100, 0, 10), (77, 784, 100, 1, 10), (48, 64, 112, 0, 16),
                                              (1000, 260, 37, 0, 3), (1, 4, 1, 0, 2), (49, 1024, 5, 1, 10)])
def test_one_kernel_evaluation_matches_reference_and_two_kernel_path(M, K, H, act, ncls):
    """eval.hip against the fp64 oracle and against the first-layer GEMM + head kernel pair it replaces."""
    import ctypes as C
    lib = hb.require()
    g = torch.Generator().manual_seed(M + K)
    x = torch.rand(M, K, generator=g).to(DEV)
    y = torch.nn.functional.one_hot(torch.randint(0, ncls, (M,), generator=g), ncls).float().to(DEV)
    W1 = (torch.randn(K, H, generator=g) * 0.1).to(DEV)
    b1 = torch.randn(H, generator=g).to(DEV)
    W2 = torch.randn(H, ncls, generator=g).to(DEV)
    b2 = torch.randn(ncls, generator=g).to(DEV)
    ws = torch.empty(M * H, device=DEV)
    out = {}
    lib.dtf_debug_set_eval_fused.argtypes = [C.c_int]
    try:
        for fused in (1, 0):
            lib.dtf_debug_set_eval_fused(fused)
            ls, cc = torch.zeros(1, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
            hb.mlp_eval(x, y, W1, b1, W2, b2, act, ws, ls, cc)
            out[fused] = (float(ls.item()) / M, int(cc.item()))
    finally:
        lib.dtf_debug_set_eval_fused(1)
    wl, wa = ref.mlp_loss_acc(x.double(), y.double(), W1.double(), b1.double(), W2.double(), b2.double(), act)
    for fused in (1, 0):
        assert abs(out[fused][0] - float(wl)) < 1e-4 * max(1.0, float(wl)), (fused, out, float(wl))
        assert abs(out[fused][1] - round(float(wa) * M)) <= 1, (fused, out, float(wa) * M)


def test_log_zero_edge_case_is_finite(step_path):
    # saturated logits: the reference's log(softmax) would produce -inf * 0 = NaN
    eng = _engine(DEV, scale=1.0)
    eng.p["W2"].mul_(200.0)
    eng.step_grad()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.loss_hist[0]) and all(torch.isfinite(v).all() for v in eng.g.values())


def _pair_gram(eng):
    nb, B = eng.batches_per_epoch, eng.B
    xb = eng.x[:nb * B].view(nb, B, -1)
    ld = (B + 15) // 16 * 16
    gram = torch.zeros(nb - 1, B, ld, device=eng.device)
    gram[:, :, :B] = 1.0 + torch.bmm(xb[1:], xb[:-1].transpose(1, 2))
    return gram


@pytest.mark.parametrize("B,K,H,C,act", [(100, 784, 100, 10, "sigmoid"), (100, 784, 100, 10, "relu"),
                                         (64, 256, 48, 10, "sigmoid"), (37, 100, 20, 7, "sigmoid"),
                                         (128, 1024, 128, 16, "sigmoid")])
@pytest.mark.parametrize("n", [1, 2, 5, 6])
def test_pipelined_chunk_tracks_cpu_reference(B, K, H, C, act, n):
    """One launch per step (the forward of step t+1 from the pre-update parameters plus the Gram correction)
    gives the parameters, losses and step count of n ordinary fused steps."""
    gpu = _engine(DEV, B, K, H, C, act, with_grads=False, nb=7)
    cpu = _engine("cpu", B, K, H, C, act, with_grads=False, nb=7)
    gpu.bind_gram(_pair_gram(gpu))
    gpu.set_cursor(1)
    cpu.set_cursor(1)
    gpu.run_chunk(n, use_graph=False)
    cpu.run_chunk(n, use_graph=False)
    torch.cuda.synchronize()
    for k in MLPEngine.PARAM_NAMES:
        torch.testing.assert_close(gpu.p[k].cpu(), cpu.p[k], rtol=2e-4, atol=2e-5)
    assert float(gpu.global_step) == float(n) and gpu._host_cursor == 1 + n
    torch.testing.assert_close(gpu.loss_hist[1:1 + n].cpu(), cpu.loss_hist[1:1 + n], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(gpu.acc_hist[1:1 + n].cpu(), cpu.acc_hist[1:1 + n], rtol=0, atol=1e-6)


def test_pipelined_chunk_graph_replay_and_two_launch_agree():
    a = _engine(DEV, with_grads=False, nb=8)
    b = _engine(DEV, with_grads=False, nb=8)
    a.bind_gram(_pair_gram(a))
    a.run_chunk(4, use_graph=True)
    a.run_chunk(3, use_graph=True)       # odd chunk: starts from the twin parameter set
    a.run_chunk(4, use_graph=True)       # crosses the epoch end: falls back to the two-launch step
    b.run_chunk(11, use_graph=False)
    torch.cuda.synchronize()
    for k in MLPEngine.PARAM_NAMES:
        torch.testing.assert_close(a.p[k], b.p[k], rtol=2e-4, atol=2e-5)
    assert a._host_cursor == 11 and float(a.global_step) == 11.0


@pytest.mark.parametrize("n", [1, 2, 4, 5])
def test_pipelined_graphs_can_be_replayed_in_any_order(n):
    """A chunk graph must not depend on what the previous chunk left in the work buffers (logit accumulators,
    twin parameters): replay two graphs alternately and compare every replay with the eager chunk."""
    g = _engine(DEV, with_grads=False, nb=30)
    e = _engine(DEV, with_grads=False, nb=30)
    for t in (g, e):
        t.bind_gram(_pair_gram(t))
    for pos in (7, 0, 7, 0, 7):
        g.set_cursor(pos)
        e.set_cursor(pos)
        g.run_chunk(n, use_graph=True)
        e.run_chunk(n, use_graph=False)
        torch.cuda.synchronize()
        for k in MLPEngine.PARAM_NAMES:
            torch.testing.assert_close(g.p[k], e.p[k], rtol=1e-5, atol=1e-6)
            g.p[k].copy_(e.p[k])


@pytest.mark.parametrize("rows,B,K", [(55000, 100, 784), (1234, 100, 784), (700, 100, 100), (450, 33, 20), (260, 104, 64)])
def test_gather_gram_matches_gather_then_batched_matmul(rows, B, K):
    """csrc/gram.hip: epoch-ordered copy + Gram blocks of consecutive mini-batches in one launch, against
    index_select and a float64 batched product (and, at the training shape, torch.baddbmm in fp32)."""
    g = torch.Generator().manual_seed(11)
    src = torch.rand(rows, K, generator=g).to(DEV)
    perm = torch.randperm(rows, generator=g).to(DEV)
    nb = rows // B
    ld = (B + 15) // 16 * 16
    dst = torch.full((rows, K), -7.0, device=DEV)
    gram = torch.full((nb - 1, B, ld), -7.0, device=DEV)
    assert hb.gather_gram_supported(B, K)
    ysrc = torch.rand(rows, 10, generator=g).to(DEV)
    ydst = torch.full((rows, 10), -7.0, device=DEV)
    hb.gather_gram(src, perm, dst, gram, B, nb, ysrc, ydst)
    assert torch.equal(ydst[:nb * B], ysrc[perm[:nb * B]]) and bool((ydst[nb * B:] == -7.0).all())
    want_rows = src[perm[:nb * B]]
    assert torch.equal(dst[:nb * B], want_rows)                       # a gather moves bits
    assert bool((dst[nb * B:] == -7.0).all())                         # rows beyond the last whole batch: untouched
    xb = want_rows.view(nb, B, K).double()
    want = 1.0 + torch.bmm(xb[1:], xb[:-1].transpose(1, 2))
    torch.testing.assert_close(gram[:, :, :B], want.float(), rtol=1e-5, atol=1e-5)
    assert bool((gram[:, :, B:] == 0).all())
    if (rows, B, K) == (55000, 100, 784):
        lib = torch.baddbmm(torch.ones(1, 1, 1, device=DEV), xb[1:].float(), xb[:-1].float().transpose(1, 2))
        torch.testing.assert_close(gram[:, :, :B], lib, rtol=1e-5, atol=1e-5)
    # Gram blocks only (no copy), from an already ordered set
    gram2 = torch.zeros_like(gram)
    ident = torch.arange(rows, device=DEV)
    hb.gather_gram(dst, ident, None, gram2, B, nb)
    assert torch.equal(gram2, gram)


def test_device_dataset_refresh_uses_the_fused_kernel_and_matches_the_library_path(monkeypatch):
    from distributed_tensorflow_amd.data import DeviceDataset

    g = torch.Generator().manual_seed(5)
    x, y = torch.rand(1230, 784, generator=g), torch.eye(10)[torch.randint(0, 10, (1230,), generator=g)]
    a = DeviceDataset(x, y, DEV, 100, seed=3)
    a.enable_pair_gram()
    a.shuffle_epoch()
    monkeypatch.setenv("DTF_GRAM_KERNEL", "0")
    b = DeviceDataset(x, y, DEV, 100, seed=3)
    b.enable_pair_gram()
    b.shuffle_epoch()
    assert a._fused_refresh() is False       # (the variable is read per call)
    assert torch.equal(a.x_epoch, b.x_epoch) and torch.equal(a.y_epoch, b.y_epoch)
    torch.testing.assert_close(a.gram, b.gram, rtol=1e-5, atol=1e-5)


def test_drawing_the_next_order_late_does_not_change_the_orders():
    """stage_ahead() only moves the host's draw of the next permutation behind queued work: same sequence either way."""
    from distributed_tensorflow_amd.data import DeviceDataset

    g = torch.Generator().manual_seed(6)
    x, y = torch.rand(530, 64, generator=g), torch.eye(10)[torch.randint(0, 10, (530,), generator=g)]
    a, b = DeviceDataset(x, y, DEV, 100, seed=4), DeviceDataset(x, y, DEV, 100, seed=4)
    for epoch in range(4):
        a.shuffle_epoch()
        a.stage_ahead()
        a.stage_ahead()                       # nothing left to do: no second draw
        b.shuffle_epoch()                     # draws when the next reshuffle takes it
        assert torch.equal(a.x_epoch, b.x_epoch) and torch.equal(a.y_epoch, b.y_epoch), epoch


@pytest.mark.parametrize("windows", [(5, 5, 2, 5, 5), (1, 1, 3, 7, 1, 2), (12, 12), (4, 3, 5, 1, 11)])
def test_chained_one_launch_chunks_give_the_results_of_chunks_that_start_afresh(windows, monkeypatch):
    """A chunk that starts where the previous one ended (forward pass and transposed W1 carried over: no prologue)
    against the same windows with every chunk starting afresh, and both against the plain two-launch step."""
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

    g = torch.Generator().manual_seed(8)
    x, y = torch.rand(1230, 784, generator=g), torch.eye(10)[torch.randint(0, 10, (1230,), generator=g)]

    def run(carry, pipelined=True):
        monkeypatch.setenv("DTF_PIPE_CARRY", "1" if carry else "0")
        ds = DeviceDataset(x, y, DEV, 100, seed=1)
        tr = build_replica_trainer(mode="single", device=DEV, dataset=ds, batch=100, lr=0.01, seed=2, chunk=12,
                                   pipelined=pipelined)
        tr.prepare(windows[:3])                 # some graphs ahead of time, the rest captured on the way
        costs = []
        for k in windows:
            tr.run_steps(k)
            costs.append(tr.last_loss())
        torch.cuda.synchronize()
        return tr, costs

    a, ca = run(True)
    b, cb = run(False)
    c, cc = run(False, pipelined=False)
    assert a.global_step_value() == b.global_step_value() == c.global_step_value() == sum(windows)
    if windows != (12, 12):                     # (two whole epochs: nothing to chain)
        assert any(k[2] in (("carry", True, True), ("carry", True, False)) for k in a._graphs), "no chunk chained"
    for other, co in ((b, cb), (c, cc)):
        # (separate runs: the logit partial sums meet through float atomics, whose order varies run to run)
        torch.testing.assert_close(a.params, other.params, rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(torch.tensor(ca), torch.tensor(co), rtol=2e-4, atol=2e-5)


def test_what_a_chunk_carries_is_dropped_by_anything_else_that_steps_the_engine():
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

    g = torch.Generator().manual_seed(9)
    x, y = torch.rand(1230, 784, generator=g), torch.eye(10)[torch.randint(0, 10, (1230,), generator=g)]
    tx, ty = x[:200].to(DEV), y[:200].to(DEV)

    def trainer():
        ds = DeviceDataset(x, y, DEV, 100, seed=1)
        return build_replica_trainer(mode="single", device=DEV, dataset=ds, batch=100, lr=0.01, seed=2, chunk=12)

    a, b = trainer(), trainer()
    a.run_steps(3)
    assert a.engine.carried_to == 3
    a.evaluate(tx, ty)                                  # reads the parameters only
    assert a.engine.carried_to == 3
    a.engine.step_fused(3)                              # a two-launch step behind the trainer's back (batch 3)
    a.engine.advance(1)
    assert a.engine.carried_to is None
    a.run_steps(4)                                      # starts afresh at batch 4, chains from there
    assert a.engine.carried_to == 8
    a.engine.invalidate_carry()
    a.run_steps(4)                                      # the last chunk of the epoch leaves nothing behind
    assert a.engine.carried_to is None and a.global_step_value() == 12
    b.run_steps(12)
    torch.cuda.synchronize()
    torch.testing.assert_close(a.params, b.params, rtol=2e-4, atol=2e-5)


def test_window_results_fetched_behind_the_next_window_equal_the_blocking_reads():
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

    g = torch.Generator().manual_seed(7)
    x, y = torch.rand(1230, 784, generator=g), torch.eye(10)[torch.randint(0, 10, (1230,), generator=g)]
    runs = []
    for pipelined in (True, False):
        ds = DeviceDataset(x, y, DEV, 100, seed=1)
        tr = build_replica_trainer(mode="single", device=DEV, dataset=ds, batch=100, lr=0.01, seed=2, chunk=5)
        out, pending = [], None
        for k in (5, 5, 2, 5, 5):            # 12 batches per epoch: the fourth window starts a new epoch
            first = tr.engine._host_cursor
            tr.run_steps(k)
            if pipelined:
                h = tr.fetch_window_async(first, k, want_history=True)
                if pending is not None:
                    out.append(pending.result())
                pending = h
            else:
                losses, accs = tr.engine.history(first, k)
                out.append((tr.last_loss(), losses, accs, float(tr.global_step_value())))
        if pending is not None:
            out.append(pending.result())
        runs.append(out)
    for (c0, l0, a0, g0), (c1, l1, a1, g1) in zip(*runs):
        # (two training runs: the logit partial sums meet through float atomics, whose order varies run to run)
        assert abs(c0 - c1) < 1e-4 * max(1.0, abs(c1)) and g0 == g1
        assert float(l0[-1]) == c0 and float(l1[-1]) == c1
        torch.testing.assert_close(l0, l1, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(a0, a1, rtol=0, atol=0.011)
    assert [r[3] for r in runs[0]] == [5.0, 10.0, 12.0, 17.0, 22.0]


@pytest.mark.parametrize("n,count", [(79516, 1), (79516, 2), (79516, 8), (1003, 3), (5, 16)])
def test_sum_apply_is_n_mailboxes_in_one_launch(n, count):
    p = _rand(n, seed=20)
    gs = [_rand(n, seed=21 + i) for i in range(count)]
    lr = torch.tensor([0.05], device=DEV)
    step = torch.zeros(1, device=DEV)
    want = p.double()
    total = torch.zeros(n, dtype=torch.float32, device=DEV)
    for gi in gs:
        total = total + gi                       # index order, fp32: what the kernel does
    want = (p - (0.05 * 0.5) * total)
    hb.sum_apply_(p, gs, lr, scale=0.5, global_step=step)
    torch.testing.assert_close(p, want, rtol=1e-6, atol=1e-6)
    assert float(step) == 1.0


def test_sgd_apply_atomic_loses_no_update_under_concurrent_streams():
    n = 79516
    p = torch.zeros(n, device=DEV)
    g = torch.ones(n, device=DEV)
    lr = torch.tensor([1.0], device=DEV)
    step = torch.zeros(1, device=DEV)
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()
    for rep in range(25):
        for s in streams:
            with torch.cuda.stream(s):
                hb.sgd_apply_atomic_(p, g, lr, 1.0, step)
    torch.cuda.synchronize()
    assert float(p.min()) == float(p.max()) == -100.0 and float(step) == 100.0


def test_native_rounds_of_the_synchronous_chief_apply_regular_rounds_and_hand_back_the_rest():
    """``dtf_sync_rounds`` (csrc/optim.hip) against a scripted barrier board: regular rounds (one arrival per worker, in
    step, same mailbox row) are applied as p -= lr * mean(g) with the step counter bumped and tokens of step + 1 dealt;
    an arrival out of step ends the call with the arrivals handed back and nothing applied; no arrivals: time-out."""
    import ctypes as C

    lib = hb.require()
    lib.dtf_sync_rounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dtf_sync_rounds.restype = C.c_int
    workers, n = 3, 1000
    script = []            # per take(): list of (worker, step, slot)
    released = []
    TAKE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int))
    RELEASE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int)

    def take(board, want, timeout_ms, out, got):
        if not script:
            got[0] = 0
            return 1
        batch = script.pop(0)[:want]
        for i, rec in enumerate(batch):
            for j in range(3):
                out[3 * i + j] = rec[j]
        got[0] = len(batch)
        return 0

    def release(board, worker, step, copies):
        released.append((worker, step, copies))
        return 0

    take_c, release_c = TAKE(take), RELEASE(release)
    g = torch.Generator().manual_seed(4)
    p = torch.randn(n + 4, generator=g).to(DEV)            # [n parameters][step counter + padding]
    boxes = [[torch.randn(n + 4, generator=g).to(DEV) for _ in range(workers)] for _ in range(2)]     # [slot][worker]
    for slot in boxes:
        for b in slot:
            b[n:] = 0.0
    lr = torch.full((1,), 0.5, device=DEV)
    params = (C.c_void_p * 1)(p.data_ptr())
    tabs_keep = [(C.c_uint64 * workers)(*[b.data_ptr() for b in boxes[slot]]) for slot in (0, 1)]
    tabs = (C.c_void_p * 2)(*[C.cast(t, C.c_void_p).value for t in tabs_keep])
    numel = (C.c_int64 * 1)(n + 4)
    gs = (C.c_void_p * 1)(p[n:n + 1].data_ptr())
    out, got, rounds = (C.c_int64 * 48)(), C.c_int(0), C.c_int(0)
    stream = torch.cuda.current_stream().cuda_stream

    def call(step, max_rounds):
        return lib.dtf_sync_rounds(C.cast(take_c, C.c_void_p), C.cast(release_c, C.c_void_p), 1, workers, step, 50, 1, params,
                                   tabs, numel, gs, lr.data_ptr(), 2, stream, max_rounds, out, C.byref(got), C.byref(rounds))

    want = p.clone()
    # two regular rounds (arrivals in two takes, any order), rows 0 and 1
    script[:] = [[(2, 7, 0)], [(0, 7, 0), (1, 7, 0)], [(1, 8, 1), (0, 8, 1), (2, 8, 1)]]
    assert call(7, 2) == 0 and rounds.value == 2 and released == [(-1, 8, 2), (-1, 9, 2)]
    for slot in (0, 1):
        want[:n] -= 0.5 * sum(b[:n] for b in boxes[slot]) / workers
    want[n] += 2.0
    torch.cuda.synchronize()
    torch.testing.assert_close(p, want, rtol=1e-6, atol=1e-6)
    # an arrival of an older step: handed back, nothing applied, no token
    script[:] = [[(0, 9, 0), (1, 8, 0), (2, 9, 0)]]
    del released[:]
    assert call(9, 4) == 1 and rounds.value == 0 and got.value == 3 and not released
    assert [tuple(out[3 * i:3 * i + 3]) for i in range(3)] == [(0, 9, 0), (1, 8, 0), (2, 9, 0)]
    # the same worker twice, and mixed mailbox rows: not a regular round either
    script[:] = [[(0, 9, 0), (0, 9, 0), (2, 9, 0)]]
    assert call(9, 1) == 1 and rounds.value == 0
    script[:] = [[(0, 9, 0), (1, 9, 1), (2, 9, 0)]]
    assert call(9, 1) == 1 and rounds.value == 0
    # one regular round, then nothing arrives: the round counts, the call ends with a time-out and two arrivals in hand
    script[:] = [[(0, 9, 1), (1, 9, 1), (2, 9, 1)], [(1, 10, 0), (2, 10, 0)]]
    assert call(9, 3) == 2 and rounds.value == 1 and got.value == 2 and released == [(-1, 10, 2)]
    want[:n] -= 0.5 * sum(b[:n] for b in boxes[1]) / workers
    want[n] += 1.0
    torch.cuda.synchronize()
    torch.testing.assert_close(p, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("where", [0, 501, 1002])
def test_applies_from_two_streams_never_lose_a_count_of_a_step_counter_inside_the_slab(where):
    """The counter is a variable of the slab like the others (zero gradient): the unlocked apply and the sum-apply bump it
    with one atomic and keep every plain write away from it, wherever in the slab it lies -- two workers that apply to one
    shard at the same time without a lock (use_locking=False, tfdist_between.py:65) may lose updates, never counts."""
    n = 1003
    p = torch.zeros(n, device=DEV)
    g = torch.zeros(n, device=DEV)
    lr = torch.full((1,), 0.1, device=DEV)
    gs = p[where:where + 1]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(150):
        for s in streams:
            with torch.cuda.stream(s):
                hb.sgd_apply_(p, g, lr, 1.0, False, gs)
                hb.sum_apply_(p, [g, g], lr, 0.5, gs)
    torch.cuda.synchronize()
    assert float(p[where]) == 600.0 and float(p.abs().sum()) == 600.0


def test_a_launch_with_an_operand_of_another_gpu_than_the_threads_current_one_is_refused(monkeypatch):
    """The kernels run on the calling thread's current device; with worker k on GPU k (tfdist_between.py:34) a thread
    still on device 0 must hear about it, not fault.  (One GPU here: the current device is faked.)"""
    from distributed_tensorflow_amd.ops import hip_backend as hb

    p = torch.zeros(64, device="cuda")
    g = torch.ones(64, device="cuda")
    lr = torch.tensor([0.5], device="cuda")
    hb.sgd_apply_(p, g, lr)
    assert torch.allclose(p, torch.full_like(p, -0.5))
    monkeypatch.setattr(hb, "_raw_device", lambda: 1)
    with pytest.raises(hb.HipKernelError, match="current device"):
        hb.sgd_apply_(p, g, lr)


def test_a_session_makes_its_gpu_the_current_device_of_its_service_threads():
    import threading
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.examples.mnist_mlp import build_model

    g = tf.Graph()
    with g.as_default():
        build_model(0.1)
    sess = tf.Session(graph=g)
    seen = []

    def service():
        sess.bind_thread()
        seen.append(torch.cuda.current_device())
        with sess._on_device():
            seen.append(torch.cuda.current_device())

    t = threading.Thread(target=service)
    t.start()
    t.join()
    assert seen == [sess.device.index, sess.device.index]
    sess.close()
