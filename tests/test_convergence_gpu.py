"""Accuracy as a test that can fail: on the slow-learning synthetic set (calibrated so that the plain PyTorch
fp32 oracle of the reference recipe reaches the reference's MNIST figures, 72 % at 55,000 updates) every GPU
tier has to land where the oracle lands, at equal update counts, and the one-launch pipelined step has to follow
the oracle's loss curve step by step through a whole epoch.  Parity with real MNIST stays unpinned (no files,
no network)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.fixture(scope="module")
def hard_set():
    from distributed_tensorflow_amd.data import read_data_sets

    return read_data_sets("MNIST_data", one_hot=True, variant="hard")


@pytest.mark.timeout(600)
def test_every_tier_lands_on_the_oracle_accuracy_at_equal_update_counts(hard_set):
    import convergence as cv

    dev = torch.device("cuda")
    marks = {5500, 27500, 55000}
    oracle = cv.oracle_curve(hard_set, dev, 100, marks)
    # the calibration itself (a different shuffle than the CPU run it was made with): the reference's 72 % band
    assert 0.68 <= oracle[55000] <= 0.76, oracle
    assert oracle[5500] < 0.30 < oracle[27500] < oracle[55000], oracle      # slow start: a 96 % plateau cannot hide here
    tiers = {"single (1 launch/step)": cv.trainer_curve("single", hard_set, dev, 100, marks),
             "two launches/step": cv.trainer_curve("single", hard_set, dev, 100, marks, pipelined=False),
             "async worker, resident shard": cv.trainer_curve("xgmi_async", hard_set, dev, 100, marks),
             "towers of one": cv.trainer_curve("towers", hard_set, dev, 100, marks)}
    for name, curve in tiers.items():
        for step in (27500, 55000):
            assert abs(curve[step] - oracle[step]) <= 0.015, (name, step, curve, oracle)


@pytest.mark.timeout(300)
def test_pipelined_step_follows_the_oracle_loss_curve_through_an_epoch(hard_set):
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.ops import reference as ref
    from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

    dev = torch.device("cuda")
    ds = DeviceDataset(hard_set.train.images, hard_set.train.labels, dev, 100, seed=0)
    tr = build_replica_trainer(mode="single", device=dev, dataset=ds, batch=100, lr=0.001, seed=1, chunk=550)
    assert tr.pipelined
    p = {k: v.clone() for k, v in tr.layout.mlp_views(tr.params).items()}
    x, y = ds.x_epoch.clone(), ds.y_epoch.clone()
    tr.run_steps(550)
    got_loss, got_acc = tr.engine.history(0, 550)
    want_loss, want_acc = [], []
    for i in range(550):
        g = ref.mlp_grads(x[100 * i:100 * i + 100], y[100 * i:100 * i + 100], p["W1"], p["b1"], p["W2"], p["b2"])
        want_loss.append(float(g["loss"]))
        want_acc.append(float(g["acc"]))
        for k in p:
            p[k].sub_(g["d" + k], alpha=0.001)
    want_loss, want_acc = torch.tensor(want_loss), torch.tensor(want_acc)
    torch.testing.assert_close(got_loss, want_loss, rtol=2e-3, atol=2e-3)
    assert float((got_acc - want_acc).abs().max()) <= 0.02 + 1e-6          # at most two rows of a batch flip
    assert float((got_acc - want_acc).abs().mean()) <= 0.002
    for k, v in tr.layout.mlp_views(tr.params).items():
        torch.testing.assert_close(v, p[k], rtol=2e-3, atol=2e-4)
