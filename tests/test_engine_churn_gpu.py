"""Many engines and trainers built and closed in ONE process.

A process that frees exchange buffers when an engine closes and allocates new ones for the next engine ends up, after
a few engines, with one XCD reading other values than the other seven from ordinary device memory (the exchange number
of a tower, then parameters): `bench/stress/ingraph_identity.py` with the variant "free" shows it.  Exchange memory is
therefore pooled for the life of the process (`parallel/xgmi.py`); these tests build what `bench.py`'s rank builds in a
row — and more — in the pytest process itself and ask every engine for `healthy()` and bit-identical replicas.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(rows=600):
    rng = np.random.RandomState(0)
    images = rng.rand(rows, 784).astype(np.float32)
    labels = np.eye(10, dtype=np.float32)[rng.randint(0, 10, rows)]
    return images, labels


def _ingraph_round(images, labels, compact):
    from distributed_tensorflow_amd.parallel.towers import InGraphEngine

    eng = InGraphEngine(["cuda:0"] * 2, images, labels, batch=100, lr=0.05, seed=1, chunk=4, compact=compact,
                        timeout_ms=3000)
    try:
        eng.prepare((5,))
        eng.run_steps(5)
        assert eng.healthy(), eng.diagnostics()
        assert eng.replicas_identical()
        eng.run_steps(9)                        # crosses an epoch boundary: reshuffle, new captures
        assert eng.healthy(), eng.diagnostics()
        assert eng.replicas_identical()
        assert eng.global_step_value() == 14
        assert not any(eng.diagnostics()["gather_errors"])
        return eng.towers[0].params.cpu().clone()
    finally:
        eng.close()


@pytest.mark.timeout(300)
def test_twelve_ingraph_engines_in_one_process_stay_healthy_and_agree():
    images, labels = _data()
    first = {}
    for i in range(6):
        for compact in (False, True):
            got = _ingraph_round(images, labels, compact)
            if compact not in first:
                first[compact] = got
            else:
                # same seed, same data: every engine retraces the first one (not bit for bit: a tower's logits are
                # summed with float atomics, in whatever order its forward workgroups arrive)
                err = float((got - first[compact]).abs().max())
                assert err <= 1e-4 * float(first[compact].abs().max()), f"engine {2 * i + int(compact)}: {err}"


@pytest.mark.timeout(300)
def test_eight_trainers_of_mixed_modes_in_one_process():
    """single -> xgmi -> xgmi2 -> xgmi2c -> xgmi2p -> xgmi_async -> in-graph -> in-graph (4-byte words): what a rank of
    `bench.py --gpus N` builds and retires in a row, as one replica each (world 1) plus two two-tower engines."""
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.ops import hip_backend as hb
    from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

    images, labels = _data()
    dev = torch.device("cuda", 0)
    finals = {}
    for mode in ("single", "xgmi", "xgmi2", "xgmi2c", "xgmi2p", "xgmi_async"):
        ds = DeviceDataset(images, labels, dev, 100, seed=3)
        tr = build_replica_trainer(mode=mode, device=dev, dataset=ds, batch=100, lr=0.05, seed=1, rank=0, world=1,
                                   chunk=4, use_graph=True)
        try:
            tr.prepare((5, 9))
            tr.run_steps(5)
            tr.run_steps(9)
            torch.cuda.synchronize(dev)
            if hasattr(tr, "healthy"):
                assert tr.healthy(), mode
            assert tr.global_step_value() == 14, mode
            if mode == "xgmi_async":
                tr.shards.pull(tr.params)
            torch.cuda.synchronize(dev)
            finals[mode] = tr.params.cpu().clone()
            assert torch.isfinite(finals[mode]).all(), mode
        finally:
            if hasattr(tr, "close"):
                tr.close()
    # one replica, one data order, one update rule: the exchange variants compute what the plain step computes
    ref = finals["single"]
    for mode, got in finals.items():
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-4 * scale, mode
    for compact in (False, True):
        _ingraph_round(images, labels, compact)
    assert hb.gather_errors(dev) == 0


def test_gather_rows_reports_a_bad_permutation_instead_of_faulting():
    from distributed_tensorflow_amd.ops import hip_backend as hb

    dev = torch.device("cuda", 0)
    src = torch.arange(40, dtype=torch.float32, device=dev).view(10, 4).contiguous()
    before = hb.gather_errors(dev)
    idx = torch.tensor([0, 3, 1 << 40, -7, 9], dtype=torch.int64, device=dev)
    out = hb.gather_rows(src, idx)
    torch.cuda.synchronize(dev)
    assert hb.gather_errors(dev) == before + 2
    assert torch.equal(out[0], src[0]) and torch.equal(out[1], src[3]) and torch.equal(out[4], src[9])
    assert torch.equal(out[2], src[9]) and torch.equal(out[3], src[0])      # clamped into the source
    hb.reset_gather_errors(dev)
