"""The dense-layer kernels (csrc/gemm.hip) against plain PyTorch oracles evaluated in fp64."""
import pytest
import torch

from distributed_tensorflow_amd.ops import gemm as gemm_op
from distributed_tensorflow_amd.ops import hip_backend as hb

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _operands(layout, M, N, K, seed):
    a = _rand(*((K, M) if layout == hb.GEMM_TN else (M, K)), seed=seed)
    b = _rand(*((N, K) if layout == hb.GEMM_NT else (K, N)), seed=seed + 1)
    ad = a.double().t() if layout == hb.GEMM_TN else a.double()
    bd = b.double().t() if layout == hb.GEMM_NT else b.double()
    return a, b, ad @ bd


SHAPES = [(128, 128, 64), (100, 100, 784), (1, 1, 1), (33, 37, 50), (130, 257, 785), (100, 10, 100), (784, 100, 100),
          (256, 384, 1030), (5, 300, 7)]


@pytest.mark.parametrize("layout", [hb.GEMM_NN, hb.GEMM_NT, hb.GEMM_TN])
@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("tile", [64, 128, 129])        # 129: the 128-tile with eight waves
def test_gemm_plain(layout, M, N, K, tile):
    a, b, want = _operands(layout, M, N, K, seed=11)
    got = hb.gemm(layout, a, b, tile=tile)
    torch.testing.assert_close(got, want.float(), rtol=2e-5, atol=2e-5 * K ** 0.5)


@pytest.mark.parametrize("layout", [hb.GEMM_NN, hb.GEMM_NT, hb.GEMM_TN])
@pytest.mark.parametrize("M,N,K", SHAPES + [(10000, 100, 784), (10000, 100, 781)])
def test_gemm_half_height_tile(layout, M, N, K):
    """The 32 x 64 tile (chosen by itself for shapes like 10000 x 100 x 784, where between one and one and a half
    64-tiles per CU would leave the CUs with two to set the time).  With A transposed in memory the two halves of a
    wave stage neighbouring k-quads of the tile's 32 rows; K = 781 ends inside a k-quad of the upper half."""
    a, b, want = _operands(layout, M, N, K, seed=12)
    got = hb.gemm(layout, a, b, tile=32)
    torch.testing.assert_close(got, want.float(), rtol=2e-5, atol=2e-5 * K ** 0.5)
    if M == 10000:
        torch.testing.assert_close(hb.gemm(layout, a, b), want.float(), rtol=2e-5, atol=2e-5 * K ** 0.5)


def test_gemm_orientation_asymmetric():
    # an identity on one side and an asymmetric matrix on the other expose a transposed tile read or write
    n = 96
    eye = torch.eye(n, device=DEV)
    w = torch.arange(n * n, device=DEV, dtype=torch.float32).reshape(n, n) / (n * n)
    torch.testing.assert_close(hb.gemm(hb.GEMM_NN, eye, w), w)
    torch.testing.assert_close(hb.gemm(hb.GEMM_NN, w, eye), w)
    torch.testing.assert_close(hb.gemm(hb.GEMM_NT, eye, w), w.t())
    torch.testing.assert_close(hb.gemm(hb.GEMM_TN, w, eye), w.t())


def test_gemm_strided_rows_and_unaligned_base():
    # operands that are column ranges of wider matrices (row stride > row length) and start off a 16-byte boundary
    big_a, big_b = _rand(70, 203, seed=1), _rand(90, 131, seed=2)
    a, b = big_a[:, 3:93], big_b[:, 1:120]          # [70, 90] x [90, 119]
    out_big = torch.zeros(70, 140, device=DEV)
    out = out_big[:, 5:124]
    hb.gemm(hb.GEMM_NN, a, b, out=out)
    torch.testing.assert_close(out, (a.double() @ b.double()).float(), rtol=2e-5, atol=2e-4)
    assert float(out_big[:, :5].abs().sum()) == 0.0 and float(out_big[:, 124:].abs().sum()) == 0.0


@pytest.mark.parametrize("act,fn", [(0, torch.sigmoid), (1, torch.relu), (2, torch.tanh), (hb.ACT_NONE, lambda z: z)])
def test_gemm_bias_act_epilogue(act, fn):
    x, W, bias = _rand(100, 300, seed=3), _rand(300, 77, scale=0.1, seed=4), _rand(77, seed=5)
    got = hb.gemm(hb.GEMM_NN, x, W, epi=hb.EPI_BIAS_ACT, act=act, bias=bias)
    want = fn(x.double() @ W.double() + bias.double()).float()
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("act", [0, 1, 2, hb.ACT_NONE])
def test_gemm_activation_gradient_epilogue(act):
    dz, W = _rand(100, 50, scale=0.1, seed=6), _rand(200, 50, seed=7)
    pre = _rand(100, 200, seed=8).double()
    a = {0: torch.sigmoid, 1: torch.relu, 2: torch.tanh, hb.ACT_NONE: lambda z: z}[act](pre)
    d = {0: a * (1 - a), 1: (a > 0).double(), 2: 1 - a * a, hb.ACT_NONE: torch.ones_like(a)}[act]
    got = hb.gemm(hb.GEMM_NT, dz, W, epi=hb.EPI_ACT_BWD, act=act, aux=a.float().contiguous())
    want = ((dz.double() @ W.double().t()) * d).float()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)


def test_gemm_sgd_epilogue_and_colsum():
    x, dz = _rand(100, 784, seed=9), _rand(100, 300, scale=0.01, seed=10)
    W, b = _rand(784, 300, seed=11), _rand(300, seed=12)
    W0, b0 = W.clone(), b.clone()
    lr = torch.tensor([0.05], device=DEV)
    hb.gemm(hb.GEMM_TN, x, dz, out=W, epi=hb.EPI_SGD, lr=lr, scale=0.5)
    hb.colsum(dz, out=b, sgd=True, lr=lr, scale=0.5)
    torch.testing.assert_close(W, (W0.double() - 0.025 * (x.double().t() @ dz.double())).float(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b, (b0.double() - 0.025 * dz.double().sum(0)).float(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(hb.colsum(dz, scale=2.0), (2.0 * dz.double().sum(0)).float(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,C", [(100, 10), (7, 1), (33, 100), (64, 1000)])
def test_softmax_xent_head(M, C):
    z = _rand(M, C, scale=3.0, seed=13)
    y = torch.zeros(M, C, device=DEV)
    y[torch.arange(M), torch.randint(0, C, (M,), generator=torch.Generator().manual_seed(14))] = 1.0
    if M > 3:
        y[3] = torch.softmax(_rand(C, seed=15), 0)       # a soft label row
    zd = z.double().requires_grad_(True)
    rows = -(y.double() * torch.log_softmax(zd, 1)).sum(1)
    rows.sum().backward()
    dz = torch.empty_like(z)
    loss_sum = torch.zeros(1, device=DEV)
    correct = torch.zeros(1, dtype=torch.int32, device=DEV)
    row_loss = torch.empty(M, device=DEV)
    hb.softmax_xent(z, y, dz=dz, grad_scale=1.0 / M, loss_sum=loss_sum, correct=correct, row_loss=row_loss)
    torch.testing.assert_close(row_loss, rows.detach().float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss_sum[0], rows.sum().detach().float(), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(dz, (zd.grad / M).float(), rtol=1e-4, atol=1e-6)
    assert int(correct.item()) == int((z.argmax(1) == y.argmax(1)).sum().item())


def test_softmax_xent_saturated_logits_stay_finite():
    # the reference takes log(softmax(z)), which underflows to log(0) here (tfdist_between.py:59-62)
    z = torch.tensor([[200.0, -200.0, 0.0], [-500.0, 500.0, 0.0]], device=DEV)
    y = torch.tensor([[0.0, 1.0, 0.0], [0.0, 1.0, 0.0]], device=DEV)
    loss_sum = torch.zeros(1, device=DEV)
    dz = torch.empty_like(z)
    hb.softmax_xent(z, y, dz=dz, loss_sum=loss_sum)
    assert torch.isfinite(loss_sum).all() and torch.isfinite(dz).all()
    torch.testing.assert_close(loss_sum[0], torch.tensor(400.0, device=DEV))


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_matmul_autograd_matches_torch(ta, tb):
    M, K, N = 45, 70, 33
    a = _rand(*((K, M) if ta else (M, K)), seed=16).requires_grad_(True)
    b = _rand(*((N, K) if tb else (K, N)), seed=17).requires_grad_(True)
    out = gemm_op.matmul(a, b, ta, tb)
    w = _rand(M, N, seed=18)
    (out * w).sum().backward()
    a2, b2 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    ref = (a2.t() if ta else a2) @ (b2.t() if tb else b2)
    (ref * w.double()).sum().backward()
    torch.testing.assert_close(out, ref.float(), rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(a.grad, a2.grad.float(), rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(b.grad, b2.grad.float(), rtol=2e-5, atol=1e-4)


def test_gemm_rejects_mismatched_operands():
    with pytest.raises(hb.HipKernelError):
        hb.gemm(hb.GEMM_NN, _rand(4, 5), _rand(6, 7))
    with pytest.raises(hb.HipKernelError):
        hb.gemm(hb.GEMM_NN, _rand(4, 5), _rand(5, 7), out=torch.zeros(4, 8, device=DEV))
    with pytest.raises(hb.HipKernelError):
        hb.gemm(hb.GEMM_NN, _rand(4, 5), _rand(5, 7), epi=hb.EPI_ACT_BWD)


@pytest.mark.parametrize("layout", [hb.GEMM_NN, hb.GEMM_NT, hb.GEMM_TN])
@pytest.mark.parametrize("M,N,K,splits", [(100, 1024, 784, 0), (100, 784, 1024, 0), (64, 64, 4096, 16), (33, 130, 517, 3),
                                          (100, 10, 2048, 0), (7, 5, 129, 2)])
def test_gemm_k_split_matches_oracle_and_leaves_counters_clean(layout, M, N, K, splits):
    a, b, want = _operands(layout, M, N, K, seed=21)
    ws = hb.gemm_workspace(DEV)
    for _ in range(3):          # the counters must be back at zero after every launch
        got = hb.gemm(layout, a, b, workspace=ws, splits=splits)
        torch.testing.assert_close(got, want.float(), rtol=2e-5, atol=2e-5 * K ** 0.5)
    assert int(ws[:1024].view(torch.int32).abs().sum()) == 0
    # same bits whatever the arrival order of the workgroups: a fixed order of the partial sums
    again = hb.gemm(layout, a, b, workspace=ws, splits=splits)
    assert torch.equal(got, again)


def test_gemm_k_split_with_every_epilogue():
    ws = hb.gemm_workspace(DEV)
    x, W, bias = _rand(100, 784, seed=3), _rand(784, 300, scale=0.1, seed=4), _rand(300, seed=5)
    got = hb.gemm(hb.GEMM_NN, x, W, epi=hb.EPI_BIAS_ACT, act=0, bias=bias, workspace=ws)
    torch.testing.assert_close(got, torch.sigmoid(x.double() @ W.double() + bias.double()).float(), rtol=2e-5, atol=2e-5)
    dz, W2 = _rand(100, 1024, scale=0.1, seed=6), _rand(200, 1024, seed=7)
    a = torch.sigmoid(_rand(100, 200, seed=8))
    got = hb.gemm(hb.GEMM_NT, dz, W2, epi=hb.EPI_ACT_BWD, act=0, aux=a, workspace=ws)
    want = ((dz.double() @ W2.double().t()) * (a.double() * (1 - a.double()))).float()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    xb, dzb, Wb = _rand(2048, 100, seed=9), _rand(2048, 64, scale=0.01, seed=10), _rand(100, 64, seed=11)
    W0 = Wb.clone()
    lr = torch.tensor([0.05], device=DEV)
    hb.gemm(hb.GEMM_TN, xb, dzb, out=Wb, epi=hb.EPI_SGD, lr=lr, scale=0.5, workspace=ws)
    torch.testing.assert_close(Wb, (W0.double() - 0.025 * (xb.double().t() @ dzb.double())).float(), rtol=1e-5, atol=1e-6)


def test_colsum_tall_matrix_uses_row_segments():
    z = _rand(5000, 130, scale=0.1, seed=12)
    torch.testing.assert_close(hb.colsum(z), z.double().sum(0).float(), rtol=1e-4, atol=1e-4)
    b = _rand(130, seed=13)
    b0 = b.clone()
    lr = torch.tensor([0.1], device=DEV)
    hb.colsum(z, out=b, sgd=True, lr=lr)
    torch.testing.assert_close(b, (b0.double() - 0.1 * z.double().sum(0)).float(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,C", [(100, 10), (1, 3), (1000, 37), (2048, 10)])
def test_softmax_xent_step_one_workgroup_head(M, C):
    z = _rand(M, C, scale=3.0, seed=31)
    y = torch.zeros(M, C, device=DEV)
    y[torch.arange(M), torch.randint(0, C, (M,), generator=torch.Generator().manual_seed(32))] = 1.0
    zd = z.double().requires_grad_(True)
    rows = -(y.double() * torch.log_softmax(zd, 1)).sum(1)
    rows.sum().backward()
    dz = torch.empty_like(z)
    loss = torch.full((1,), 123.0, device=DEV)                    # stored, not accumulated: old contents do not matter
    correct = torch.full((1,), 77, dtype=torch.int32, device=DEV)
    gs = torch.tensor([41.0], device=DEV)
    hb.softmax_xent_step(z, y, dz, 1.0 / M, loss, correct, gs)
    torch.testing.assert_close(loss[0], rows.sum().detach().float(), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(dz, (zd.grad / M).float(), rtol=1e-4, atol=1e-6)
    assert int(correct.item()) == int((z.argmax(1) == y.argmax(1)).sum().item()) and float(gs) == 42.0
    again = torch.empty(1, device=DEV)
    hb.softmax_xent_step(z, y, dz, 1.0 / M, again, correct, None)
    assert torch.equal(again, loss) and float(gs) == 42.0         # a fixed order of additions; no counter given, none bumped


@pytest.mark.parametrize("M,C", [(100, 10), (17, 3), (1000, 37), (4096, 10), (10001, 5)])
def test_softmax_xent_head_many_workgroups_stores_without_a_reset(M, C):
    """Per-workgroup sums meet in a scratch buffer, the last workgroup to arrive adds them in a fixed order and STORES
    the statistics: stale outputs do not matter, the ticket is back at zero afterwards, repeated launches give the same bits."""
    z = _rand(M, C, scale=3.0, seed=61)
    y = torch.zeros(M, C, device=DEV)
    y[torch.arange(M), torch.randint(0, C, (M,), generator=torch.Generator().manual_seed(62))] = 1.0
    zd = z.double().requires_grad_(True)
    rows = -(y.double() * torch.log_softmax(zd, 1)).sum(1)
    rows.sum().backward()
    dz = torch.empty_like(z)
    scratch = hb.xent_head_scratch(M, DEV)
    loss = torch.full((1,), 123.0, device=DEV)
    correct = torch.full((1,), 77, dtype=torch.int32, device=DEV)
    gs = torch.tensor([41.0], device=DEV)
    hb.softmax_xent_head(z, y, dz, 1.0 / M, scratch, loss, correct, gs)
    torch.testing.assert_close(loss[0], rows.sum().detach().float(), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(dz, (zd.grad / M).float(), rtol=1e-4, atol=1e-6)
    assert int(correct.item()) == int((z.argmax(1) == y.argmax(1)).sum().item()) and float(gs) == 42.0
    blocks = (M + 3) // 4
    assert int(scratch[2 * blocks:2 * blocks + 1].view(torch.int32).item()) == 0
    for _ in range(3):
        again, c2 = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
        hb.softmax_xent_head(z, y, dz, 1.0 / M, scratch, again, c2, None)
        assert torch.equal(again, loss) and torch.equal(c2, correct) and float(gs) == 42.0
    with pytest.raises(hb.HipKernelError):
        hb.softmax_xent_head(z, y, dz, 1.0 / M, scratch[:blocks], loss, correct, gs)        # scratch too small


def test_colsum_multi_one_launch_for_several_matrices():
    zs = [_rand(100, n, scale=0.1, seed=40 + n) for n in (10, 300, 64, 1)]
    outs = [_rand(n, seed=50 + n) for n in (10, 300, 64, 1)]
    before = [o.clone() for o in outs]
    plan = hb.ColsumPlan(list(zip(zs, outs)))
    lr = torch.tensor([0.2], device=DEV)
    hb.colsum_multi(plan, sgd=True, lr=lr, scale=0.5)
    for z, o, o0 in zip(zs, outs, before):
        torch.testing.assert_close(o, (o0.double() - 0.1 * z.double().sum(0)).float(), rtol=1e-5, atol=1e-6)
    hb.colsum_multi(plan, scale=2.0)
    for z, o in zip(zs, outs):
        torch.testing.assert_close(o, (2.0 * z.double().sum(0)).float(), rtol=1e-5, atol=1e-6)


def test_gemm_workspace_too_small_falls_back_to_one_workgroup_per_tile():
    a, b, want = _operands(hb.GEMM_NN, 100, 256, 2048, seed=61)
    small = torch.zeros(1024 + 3 * 64 * 64, dtype=torch.float32, device=DEV)      # room for less than two ranges of 8 tiles
    got = hb.gemm(hb.GEMM_NN, a, b, workspace=small)
    torch.testing.assert_close(got, want.float(), rtol=2e-5, atol=2e-5 * 2048 ** 0.5)
    assert int(small[:1024].view(torch.int32).abs().sum()) == 0
    tiny = torch.zeros(16, dtype=torch.float32, device=DEV)                       # not even the counters: ignored
    torch.testing.assert_close(hb.gemm(hb.GEMM_NN, a, b, workspace=tiny), want.float(), rtol=2e-5, atol=2e-5 * 2048 ** 0.5)


def test_gemm_refuses_leading_dimensions_beyond_its_32_bit_offsets():
    # a row stride of 5 M floats: 128 rows of a tile span more than 2 GB
    wide = torch.empty(0, device=DEV)
    try:
        wide = torch.empty((130, 5_000_000), dtype=torch.float32, device=DEV)     # 2.6 GB
    except RuntimeError:
        pytest.skip("no room for the 2.6 GB operand")
    with pytest.raises(hb.HipKernelError, match="unsupported"):
        hb.gemm(hb.GEMM_NT, wide[:, :64], wide[:, 64:128])
