"""CPU-side pieces of the peer-mapped exchange: gradient slots of the engine, and a trial that cannot run."""
import torch

from distributed_tensorflow_amd import bench_support
from distributed_tensorflow_amd.models import MLPEngine
from distributed_tensorflow_amd.ops import reference as ref


def _engine():
    g = torch.Generator().manual_seed(3)
    p = {"W1": torch.randn(20, 8, generator=g), "b1": torch.zeros(8), "W2": torch.randn(8, 4, generator=g),
         "b2": torch.zeros(4)}
    x = torch.rand(12, 20, generator=g)
    y = torch.eye(4)[torch.randint(0, 4, (12,), generator=g)]
    eng = MLPEngine(p, None, batch=6, lr=0.1)
    eng.bind_data(x, y)
    return eng, p, x, y


def test_step_grad_writes_into_the_requested_slot_only():
    eng, p, x, y = _engine()
    slots = [{k: torch.zeros_like(v) for k, v in p.items()} for _ in range(2)]
    eng.bind_grad_slots(slots)
    eng.step_grad(1, scale=0.5, slot=1)
    want = ref.mlp_grads(x[6:12], y[6:12], p["W1"], p["b1"], p["W2"], p["b2"])
    for k in p:
        torch.testing.assert_close(slots[1][k], 0.5 * want["d" + k])
        assert float(slots[0][k].abs().sum()) == 0.0
    eng.step_grad(0, scale=1.0, slot=0)
    want0 = ref.mlp_grads(x[0:6], y[0:6], p["W1"], p["b1"], p["W2"], p["b2"])
    torch.testing.assert_close(slots[0]["W1"], want0["dW1"])
    torch.testing.assert_close(slots[1]["W1"], 0.5 * want["dW1"])     # untouched by the second call


def test_slot_shapes_are_checked():
    eng, p, _, _ = _engine()
    bad = {k: torch.zeros_like(v) for k, v in p.items()}
    bad["W2"] = torch.zeros(4, 8)
    try:
        eng.bind_grad_slots([bad])
    except ValueError as e:
        assert "W2" in str(e)
    else:
        raise AssertionError("a transposed gradient slot was accepted")


def test_a_trial_that_cannot_run_is_an_answer_not_an_error(monkeypatch):
    # no GPU here: the child fails while setting its device and must come back as {"ok": False, ...}
    monkeypatch.setenv("MASTER_PORT", "29733")
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")      # must not leak into the child's rendezvous
    res = bench_support._trial_xgmi(rank=0, local_rank=0, world=1, chunk=10)
    assert res["ok"] is False and res["uncached_grads"] is False
    assert "error" in res
