"""Plain PyTorch fp32 implementations of every fused op.

They serve two purposes: (1) the CPU / soft-placement execution path (the
reference runs with ``allow_soft_placement=True``, tfdist_between.py:15), and
(2) the numerical oracle the HIP kernels are tested against.

Model (reference tfdist_between.py:55-62, 68-70)::

    a2 = act(x @ W1 + b1);  z3 = a2 @ W2 + b2;  y = softmax(z3)
    loss = mean(-sum(y_ * log(y), 1));  acc = mean(argmax(y,1) == argmax(y_,1))
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

ACT_SIGMOID = 0
ACT_RELU = 1
ACT_NAMES = {"sigmoid": ACT_SIGMOID, "relu": ACT_RELU}


def act_code(act) -> int:
    if isinstance(act, str):
        return ACT_NAMES[act]
    return int(act)


def activation(z: torch.Tensor, act: int) -> torch.Tensor:
    return torch.sigmoid(z) if act == ACT_SIGMOID else torch.relu(z)


def activation_grad_from_output(a: torch.Tensor, act: int) -> torch.Tensor:
    if act == ACT_SIGMOID:
        return a * (1.0 - a)
    return (a > 0).to(a.dtype)


def linear_bias_act(x, W, b, act: int):
    return activation(x @ W + b, act)


def mlp_forward(x, W1, b1, W2, b2, act: int = ACT_SIGMOID):
    a2 = linear_bias_act(x, W1, b1, act)
    z3 = a2 @ W2 + b2
    return a2, z3


def softmax_xent_acc(z3, y) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns (per-row loss, correct flags, softmax probabilities); stable log-softmax."""
    logp = torch.log_softmax(z3, dim=1)
    row_loss = -(y * logp).sum(dim=1)
    correct = (z3.argmax(dim=1) == y.argmax(dim=1)).to(z3.dtype)
    return row_loss, correct, logp.exp()


def mlp_loss_acc(x, y, W1, b1, W2, b2, act: int = ACT_SIGMOID):
    _, z3 = mlp_forward(x, W1, b1, W2, b2, act)
    row_loss, correct, _ = softmax_xent_acc(z3, y)
    return row_loss.mean(), correct.mean()


def mlp_grads(x, y, W1, b1, W2, b2, act: int = ACT_SIGMOID) -> Dict[str, torch.Tensor]:
    """Closed-form backward pass (what autodiff of the reference graph yields)."""
    B = x.shape[0]
    a2, z3 = mlp_forward(x, W1, b1, W2, b2, act)
    row_loss, correct, p = softmax_xent_acc(z3, y)
    dz3 = (p * y.sum(dim=1, keepdim=True) - y) / B
    dW2 = a2.t() @ dz3
    db2 = dz3.sum(dim=0)
    da2 = dz3 @ W2.t()
    dz2 = da2 * activation_grad_from_output(a2, act)
    dW1 = x.t() @ dz2
    db1 = dz2.sum(dim=0)
    return {
        "loss": row_loss.mean(),
        "acc": correct.mean(),
        "dW1": dW1,
        "db1": db1,
        "dW2": dW2,
        "db2": db2,
        "dz2": dz2,
        "a2": a2,
    }


def sgd_apply_(p: torch.Tensor, g: torch.Tensor, lr: float, scale: float = 1.0, zero_grad: bool = False):
    p.sub_(g, alpha=float(lr) * float(scale))
    if zero_grad:
        g.zero_()
    return p


def accumulate_(acc: torch.Tensor, g: torch.Tensor):
    acc.add_(g)
    return acc


# --- Philox4x32-10 (host mirror of csrc/optim.hip) --------------------------

_M0 = 0xD2511F53
_M1 = 0xCD9E8D57
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = 0xFFFFFFFF


def _philox4x32_10(group: int, seed: int):
    c0, c1, c2, c3 = group & _MASK, (group >> 32) & _MASK, 0, 0
    k0, k1 = seed & _MASK, (seed >> 32) & _MASK
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> 32, p0 & _MASK
        hi1, lo1 = p1 >> 32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK, lo1, (hi0 ^ c3 ^ k1) & _MASK, lo0
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def philox_normal_host(n: int, seed: int, offset: int = 0, mean: float = 0.0, stddev: float = 1.0):
    """Slow scalar mirror of the device generator; for tests on small n only."""
    out = []
    s = 2.0 ** -32
    for gi in range(offset, offset + n):
        c = _philox4x32_10(gi >> 2, seed)
        comp = gi & 3
        if comp < 2:
            u_r, u_t = c[0], c[1]
        else:
            u_r, u_t = c[2], c[3]
        ur = min((float(u_r) + 1.0) * s, 1.0)
        ut = float(u_t) * s
        r = math.sqrt(-2.0 * math.log(ur))
        v = r * (math.cos(2 * math.pi * ut) if comp % 2 == 0 else math.sin(2 * math.pi * ut))
        out.append(mean + stddev * v)
    return torch.tensor(out, dtype=torch.float32)


def random_normal(shape, seed: int, offset: int = 0, mean: float = 0.0, stddev: float = 1.0,
                  device="cpu") -> torch.Tensor:
    """Deterministic N(mean, stddev) fill for the CPU path (torch generator keyed by seed+offset)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed((int(seed) * 1000003 + int(offset)) & 0x7FFFFFFFFFFFFFFF)
    t = torch.empty(tuple(shape), dtype=torch.float32)
    t.normal_(mean, stddev, generator=gen)
    return t.to(device)
