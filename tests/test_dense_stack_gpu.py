"""DenseStackEngine on the GPU kernels against the fp64 autograd oracle; session integration on a GPU session."""
import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.models import DenseStackEngine

from test_dense_stack import _batch, _oracle_step, build_deep

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stack(dims, seed=3, scale=0.5):
    gen = torch.Generator().manual_seed(seed)
    Ws = [torch.randn(dims[i], dims[i + 1], generator=gen) * scale / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=gen) * 0.1 for i in range(len(dims) - 1)]
    return Ws, bs, gen


@pytest.mark.parametrize("dims,acts,B", [((784, 100, 10), ("sigmoid",), 100),
                                         ((784, 300, 200, 10), ("relu", "sigmoid"), 100),
                                         ((50, 33, 17, 9, 3), ("tanh", "linear", "relu"), 7),
                                         ((64, 1000), (), 33)])
def test_fused_steps_match_autograd(dims, acts, B):
    Ws, bs, gen = _stack(dims)
    nb = 3
    x = torch.rand(nb * B, dims[0], generator=gen)
    y = torch.eye(dims[-1])[torch.randint(0, dims[-1], (nb * B,), generator=gen)]
    gs = torch.zeros(1, device=DEV)
    eng = DenseStackEngine([w.to(DEV) for w in Ws], [b.to(DEV) for b in bs], acts, batch=B, lr=0.5, global_step=gs)
    eng.bind_data(x.to(DEV), y.to(DEV))
    for step in range(5):
        eng.run_chunk(1, use_graph=False)
        sl = slice((step % nb) * B, (step % nb + 1) * B)
        want = _oracle_step(Ws, bs, acts, x[sl], y[sl], 0.5)
        got = eng.last_stats()
        assert abs(got[0] - want[0]) < 2e-4 * max(1.0, abs(want[0])) and abs(got[1] - want[1]) < 1e-6, (step, got, want)
    for w, we in zip(Ws, eng.W):
        torch.testing.assert_close(we.cpu(), w, rtol=2e-4, atol=2e-5)
    for b, be in zip(bs, eng.b):
        torch.testing.assert_close(be.cpu(), b, rtol=2e-4, atol=2e-5)
    assert float(gs) == 5.0
    loss, acc = eng.evaluate(x.to(DEV), y.to(DEV))
    h = x.double()
    fns = {"sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh, "linear": lambda z: z}
    for i in range(len(Ws)):
        z = h @ Ws[i].double() + bs[i].double()
        h = fns[acts[i]](z) if i < len(Ws) - 1 else z
    want_loss = float(-(y.double() * torch.log_softmax(h, 1)).sum(1).mean())
    assert abs(loss - want_loss) < 2e-4 * max(1.0, want_loss)
    assert abs(acc - float((h.argmax(1) == y.argmax(1)).double().mean())) < 2.0 / x.shape[0]


@pytest.mark.parametrize("branched", [False, True])
def test_graph_chunks_equal_eager_steps(branched):
    """``branched``: the captured step as a graph with side-stream branches (``DTF_DENSE_OVERLAP=1``; measured slower
    than the chain, kept as an option) must be the same arithmetic."""
    dims, acts, B = (784, 256, 128, 10), ("relu", "relu"), 100
    Ws, bs, gen = _stack(dims)
    x = torch.rand(6 * B, dims[0], generator=gen).to(DEV)
    y = torch.eye(10)[torch.randint(0, 10, (6 * B,), generator=gen)].to(DEV)
    engines = []
    for use_graph in (False, True):
        eng = DenseStackEngine([w.to(DEV) for w in Ws], [b.to(DEV) for b in bs], acts, batch=B, lr=0.1,
                               global_step=torch.zeros(1, device=DEV))
        eng.bind_data(x, y)
        eng.overlap = branched
        for n in (4, 4, 4):             # 12 steps over a 6-batch epoch
            eng.run_chunk(n, use_graph=use_graph)
        engines.append(eng)
    torch.cuda.synchronize()
    a, b = engines
    for wa, wb in zip(a.W + a.b, b.W + b.b):
        assert torch.equal(wa, wb)
    assert float(a.global_step) == float(b.global_step) == 12.0
    la, lb = a.history(6, 6), b.history(6, 6)
    # (the step's loss is a sum of per-workgroup atomics: equal up to the order of 25 additions)
    torch.testing.assert_close(la[0], lb[0], rtol=1e-6, atol=0)
    assert torch.equal(la[1], lb[1])
    assert set(b._graphs) == {(0, 4), (4, 4), (2, 4)}      # (4, 4) wraps around the end of the epoch


def test_step_grad_matches_autograd_scaled():
    dims, acts, B = (100, 64, 10), ("sigmoid",), 50
    Ws, bs, gen = _stack(dims)
    x = torch.rand(B, dims[0], generator=gen)
    y = torch.eye(10)[torch.randint(0, 10, (B,), generator=gen)]
    grads = [(torch.zeros_like(w, device=DEV), torch.zeros_like(b, device=DEV)) for w, b in zip(Ws, bs)]
    eng = DenseStackEngine([w.to(DEV) for w in Ws], [b.to(DEV) for b in bs], acts, batch=B, lr=0.0, grads=grads)
    eng.bind_data(x.to(DEV), y.to(DEV))
    eng.step_grad(0, scale=0.25)
    W2, b2 = [w.clone() for w in Ws], [b.clone() for b in bs]
    _oracle_step(W2, b2, acts, x, y, 1.0)
    for (gw, gb), w, w2, b, bb in zip(grads, Ws, W2, bs, b2):
        torch.testing.assert_close(gw.cpu(), 0.25 * (w - w2), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(gb.cpu(), 0.25 * (b - bb), rtol=1e-4, atol=1e-6)
    for w, we in zip(Ws, eng.W):
        assert torch.equal(we.cpu(), w)         # parameters untouched


def test_session_on_gpu_runs_deep_graph_on_the_kernels(monkeypatch):
    x, y = _batch(0, n=100, dims=(784, 128, 64, 10))

    def run(n):
        with tf.Graph().as_default():
            with tf.device("/gpu:0"):
                m = build_deep(0.05, dims=(784, 128, 64, 10), acts=("sigmoid", "relu"))
            with tf.Session() as sess:
                assert sess.device.type == "cuda"
                sess.run(m.init_op)
                out = [sess.run([m.train_op, m.loss, m.accuracy], feed_dict={m.x: x, m.y_: y})[1:] for _ in range(n)]
                ev = sess.run([m.loss, m.accuracy], feed_dict={m.x: x, m.y_: y})
                return out, ev, (sess.dense_steps, sess.generic_steps)

    out_d, ev_d, counts_d = run(5)
    assert counts_d == (5, 0)
    monkeypatch.setenv("DTF_DENSE_ENGINE", "0")         # op-by-op: MatMul nodes on the same GEMM through autograd
    out_g, ev_g, counts_g = run(5)
    assert counts_g == (0, 5)
    np.testing.assert_allclose(np.array(out_d, dtype=np.float64), np.array(out_g, dtype=np.float64), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(ev_d, ev_g, rtol=2e-4, atol=1e-5)
    assert out_d[-1][0] < out_d[0][0]


def test_resident_trainer_graph_replay_equals_eager_across_reshuffles():
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.examples.mnist_mlp import build_model
    from distributed_tensorflow_amd.parallel.dense_trainer import DenseResidentTrainer

    rng = np.random.RandomState(0)
    labels = rng.randint(0, 10, 1200)
    images = (rng.rand(10, 784).astype(np.float32)[labels] + 0.3 * rng.rand(1200, 784)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]
    finals = []
    for use_graph in (False, True):
        with tf.Graph().as_default():
            with tf.device("/gpu:0"):
                m = build_model(0.05, "relu", hidden=(256, 128))
            with tf.Session() as sess:
                sess.run(m.init_op)
                ds = DeviceDataset(images, onehot, sess.device, 100, seed=3)
                tr = DenseResidentTrainer(sess, m.train_op, ds, 100, chunk=5, use_graph=use_graph)
                tr.prepare((30,))
                tr.run_steps(30)                # 12 batches per epoch: two reshuffles
                assert tr.global_step_value() == 30
                finals.append([sess.var_tensor(v).clone() for v in m.weights + m.biases] + [tr.evaluate(
                    torch.as_tensor(images).cuda(), torch.as_tensor(onehot).cuda())])
                tr.close()
    for a, b in zip(finals[0][:-1], finals[1][:-1]):
        assert torch.equal(a, b)
    assert finals[0][-1][1] > 0.8


def test_data_parallel_trainer_gpu_streams_match_cpu_run():
    """One rank on the GPU (buckets reduced and applied on the side stream under the backward of the layers below)
    against the same trainer on the CPU: same seed, same shuffles, same steps."""
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.parallel.dense_trainer import DenseDataParallelTrainer

    rng = np.random.RandomState(7)
    labels = rng.randint(0, 10, 1200)
    images = (rng.rand(10, 784).astype(np.float32)[labels] + 0.3 * rng.rand(1200, 784)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]
    runs = []
    for dev in ("cpu", "cuda"):
        ds = DeviceDataset(images, onehot, torch.device(dev), 100, seed=5)
        tr = DenseDataParallelTrainer((784, 256, 128, 10), ("relu", "sigmoid"), ds, 100, 0.1, dev, seed=3, bucket_mb=0.1)
        assert len(tr.buckets) == 2
        tr.run_steps(15)            # 12 batches per epoch: one reshuffle
        runs.append((tr.params.cpu(), tr.global_step_value(), tr.last_loss()))
    assert runs[0][1] == runs[1][1] == 15
    torch.testing.assert_close(runs[1][0], runs[0][0], rtol=2e-4, atol=2e-5)
    assert abs(runs[0][2] - runs[1][2]) < 1e-3


def test_data_parallel_trainer_reduces_through_rccl_on_its_side_stream():
    """A one-rank RCCL communicator carries the bucket all-reduces (own process: the communicator is process-wide)."""
    import json
    import os
    import subprocess
    import sys

    from cluster_utils import ROOT, free_ports

    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dense_dp_rccl_worker.py"), str(free_ports(1)[0])],
                       cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert res["backend"] == "nccl" and "RCCL" in res["desc"] and res["buckets"] == 2
    assert res["identical"] is True and res["global_step"] == 30           # parameters: bit for bit
    assert abs(res["loss"] - res["loss_without"]) < 1e-6 and res["loss"] < 1.0
    # and as hipGraph chunks of 5 steps (the collectives on their own branch of the graph): the same bits again
    assert res["graph_used"] is True and res["graph_identical"] is True and res["graph_step"] == 30, res
    assert "hipGraph chunks of 5" in res["graph_desc"] and [0, 5] in res["graph_keys"] and [10, 2] in res["graph_keys"]


def test_stack_without_biases_and_with_tall_batch():
    """No bias anywhere (no column-sum launch), and a batch beyond the one-launch column sum's 4096 rows."""
    dims, acts = (96, 80, 12), ("tanh",)
    Ws, _, gen = _stack(dims)
    for B in (64, 4200):
        x = torch.rand(B, dims[0], generator=gen)
        y = torch.eye(dims[-1])[torch.randint(0, dims[-1], (B,), generator=gen)]
        for with_bias in (False, True):
            W = [w.clone() for w in Ws]
            bs = [torch.zeros(n) for n in dims[1:]]
            eng = DenseStackEngine([w.to(DEV) for w in W], [b.to(DEV) for b in bs] if with_bias else [None, None], acts,
                                   batch=B, lr=0.3)
            eng.bind_data(x.to(DEV), y.to(DEV))
            eng.run_chunk(2, use_graph=False)
            if with_bias:
                for _ in range(2):
                    _oracle_step(W, bs, acts, x, y, 0.3)
                for b, be in zip(bs, eng.b):
                    torch.testing.assert_close(be.cpu(), b, rtol=2e-4, atol=2e-5)
            else:
                Wd = [w.double() for w in W]
                for _ in range(2):
                    leaves = [w.clone().requires_grad_(True) for w in Wd]
                    h = torch.tanh(x.double() @ leaves[0]) @ leaves[1]
                    (-(y.double() * torch.log_softmax(h, 1)).sum(1).mean()).backward()
                    Wd = [(w - 0.3 * w.grad).detach() for w in leaves]
                W = [w.float() for w in Wd]
            for w, we in zip(W, eng.W):
                torch.testing.assert_close(we.cpu(), w, rtol=2e-4, atol=2e-5)


@pytest.mark.timeout(300)
def test_in_graph_engine_of_a_dense_stack_one_graph_launch_per_tower_and_chunk():
    """``DenseInGraphEngine``: two towers sharing the GPU, a stream and hipGraph chunks per tower, the stack's gradient
    launches and ONE exchange kernel per step.  One step equals the fp64 oracle's step on the mean gradient of both
    towers' batches; the replicas stay bit-identical across a reshuffle; graph replay equals eager launches."""
    from distributed_tensorflow_amd.parallel.towers import DenseInGraphEngine

    rng = np.random.RandomState(0)
    labels = rng.randint(0, 10, 400)
    images = (rng.rand(10, 784).astype(np.float32)[labels] + 0.3 * rng.rand(400, 784)).astype(np.float32)
    onehot = np.eye(10, dtype=np.float32)[labels]
    finals = {}
    for use_graph in (True, False):
        tw = DenseInGraphEngine(["cuda:0", "cuda:0"], images, onehot, batch=100, lr=0.05, hidden=(32, 16), act="tanh",
                                seed=1, chunk=3, use_graph=use_graph, timeout_ms=5000)
        try:
            assert "784-32-16-10" in tw.describe() and tw.batches_per_epoch == 4
            assert tw.host_calls_per_chunk() == (1 if use_graph else 3 * (tw.engines[0].launches_per_step + 1))
            Ws = [w.detach().cpu().clone() for w in tw.engines[0].W]
            bs = [b.detach().cpu().clone() for b in tw.engines[0].b]
            x = torch.cat([ds.x_epoch[:100].cpu() for ds in tw.datasets])
            y = torch.cat([ds.y_epoch[:100].cpu() for ds in tw.datasets])
            tw.prepare((1, 5))
            tw.run_steps(1)
            assert tw.healthy() and tw.replicas_identical() and tw.global_step_value() == 1
            _oracle_step(Ws, bs, ("tanh", "tanh"), x, y, 0.05)      # the mean over both towers' batches
            for got, want in zip(tw.engines[1].W + tw.engines[1].b, Ws + bs):
                torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-6)
            tw.run_steps(5)                       # crosses the 4-batch epoch boundary (every tower reshuffles its set)
            assert tw.healthy() and tw.replicas_identical() and tw.global_step_value() == 6
            loss, acc = tw.evaluate(torch.as_tensor(images).cuda(), torch.as_tensor(onehot).cuda())
            assert np.isfinite(loss) and 0.0 <= acc <= 1.0
            finals[use_graph] = tw.params[0].cpu().clone()
        finally:
            tw.close()
    torch.testing.assert_close(finals[True], finals[False], rtol=1e-5, atol=1e-6)
