"""Variable specs, shard layouts and flat fp32 slabs.

The "graph" of this framework is a fixed fused pipeline over flat parameter
slabs; *placement* decides which PS shard owns which slab range.

* :class:`VarSpec` — name, shape, initializer, trainable flag, owning shard
  (reference tfdist_between.py:37, 47-53: ``global_step`` scalar 0,
  W1/W2 ~ N(0,1), b1/b2 zeros, graph seed 1).
* :class:`ShardLayout` — maps variables to element ranges.  Two modes:

  - ``"by_variable"``: shard k stores exactly the variables assigned to it by
    the device setter, contiguously, in creation order (TensorFlow semantics;
    shards can be wildly uneven — 78,500 vs 1,011 elements for 2 PS tasks);
  - ``"flat"``: the variables are laid out back to back in creation order and
    the flat range is cut into ``num_shards`` equal chunks — every shard the
    same size, so reduce-scatter / all-gather need no padding logic.

  In both modes a worker's local copy is ONE contiguous slab that is the
  concatenation of all shard ranges, so pull/push of shard k is one contiguous
  copy and every variable is a contiguous view (kernels need that).
* Initial values are a pure function of (graph seed, variable index, element
  index): any process — chief, PS shard owner, a test — reproduces them
  without communication.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

ALIGN = 4  # elements (16 bytes): every variable and shard range starts 16-B aligned


def _round_up(n: int, a: int) -> int:
    return (n + a - 1) // a * a


@dataclass
class VarSpec:
    name: str
    shape: Tuple[int, ...]
    init: Tuple = ("zeros",)          # ("zeros",) | ("constant", v) | ("normal", mean, stddev)
    trainable: bool = True
    shard: int = 0
    index: int = 0                     # creation index in the graph
    device: str = ""

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= int(s)
        return n


def mlp_var_specs(in_dim: int = 784, hidden: int = 100, classes: int = 10) -> List[VarSpec]:
    """The reference model's variables in the reference's creation order and TF names."""
    return [
        VarSpec("global_step", (), ("constant", 0.0), trainable=False, index=0),
        VarSpec("weights/Variable", (in_dim, hidden), ("normal", 0.0, 1.0), index=1),
        VarSpec("weights/Variable_1", (hidden, classes), ("normal", 0.0, 1.0), index=2),
        VarSpec("biases/Variable", (hidden,), ("zeros",), index=3),
        VarSpec("biases/Variable_1", (classes,), ("zeros",), index=4),
    ]


MLP_ROLE_NAMES = {"W1": "weights/Variable", "W2": "weights/Variable_1", "b1": "biases/Variable",
                  "b2": "biases/Variable_1"}


def assign_shards(specs: Sequence[VarSpec], num_shards: int, strategy=None) -> None:
    """Apply a placement strategy in creation order (default: round-robin by count)."""
    from .placement import RoundRobinStrategy

    strat = strategy or RoundRobinStrategy(max(num_shards, 1))
    for s in sorted(specs, key=lambda v: v.index):
        s.shard = strat(s.name, 4 * s.numel) if num_shards > 0 else 0
        s.device = f"/job:ps/task:{s.shard}"


class ShardLayout:
    def __init__(self, specs: Sequence[VarSpec], num_shards: int, mode: str = "by_variable"):
        if mode not in ("by_variable", "flat"):
            raise ValueError(mode)
        self.specs = sorted(specs, key=lambda v: v.index)
        self.num_shards = max(int(num_shards), 1)
        self.mode = mode
        self.var_range: Dict[str, Tuple[int, int]] = {}
        self.shard_range: List[Tuple[int, int]] = []
        if mode == "by_variable":
            off = 0
            for k in range(self.num_shards):
                start = off
                for s in self.specs:
                    if s.shard == k:
                        self.var_range[s.name] = (off, s.numel)
                        off += _round_up(max(s.numel, 1), ALIGN)
                self.shard_range.append((start, off))
            self.total = off
        else:
            off = 0
            for s in self.specs:
                self.var_range[s.name] = (off, s.numel)
                off += _round_up(max(s.numel, 1), ALIGN)
            chunk = _round_up((off + self.num_shards - 1) // self.num_shards, ALIGN)
            self.total = chunk * self.num_shards
            self.shard_range = [(k * chunk, (k + 1) * chunk) for k in range(self.num_shards)]
        self.by_name = {s.name: s for s in self.specs}

    # -- queries ---------------------------------------------------------------
    def shard_numel(self, k: int) -> int:
        a, b = self.shard_range[k]
        return b - a

    def vars_of_shard(self, k: int) -> List[VarSpec]:
        """Variables whose storage intersects shard k (whole variables in by_variable mode)."""
        a, b = self.shard_range[k]
        out = []
        for s in self.specs:
            o, n = self.var_range[s.name]
            if o < b and o + max(n, 1) > a:
                out.append(s)
        return out

    def owner_of(self, name: str) -> int:
        o, _ = self.var_range[name]
        for k, (a, b) in enumerate(self.shard_range):
            if a <= o < b:
                return k
        raise KeyError(name)

    def describe(self) -> str:
        lines = []
        for k, (a, b) in enumerate(self.shard_range):
            names = ", ".join(f"{s.name}{list(s.shape)}" for s in self.vars_of_shard(k))
            lines.append(f"/job:ps/task:{k}: elements [{a}, {b}) = {b - a} ({4 * (b - a)} B): {names}")
        return "\n".join(lines)

    # -- slabs ---------------------------------------------------------------------
    def new_slab(self, device, zero: bool = True) -> torch.Tensor:
        f = torch.zeros if zero else torch.empty
        return f((self.total,), dtype=torch.float32, device=device)

    def view(self, slab: torch.Tensor, name: str) -> torch.Tensor:
        o, n = self.var_range[name]
        return slab[o:o + n].view(self.by_name[name].shape)

    def views(self, slab: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {s.name: self.view(slab, s.name) for s in self.specs}

    def shard_view(self, slab: torch.Tensor, k: int) -> torch.Tensor:
        a, b = self.shard_range[k]
        return slab[a:b]

    def mlp_views(self, slab: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {role: self.view(slab, name) for role, name in MLP_ROLE_NAMES.items()}

    # -- initialisation --------------------------------------------------------------
    def initialize(self, slab: torch.Tensor, seed: int, shard: Optional[int] = None,
                   slab_offset: int = 0) -> None:
        """Run the variables' initializers into ``slab``.

        ``slab`` is either a full local slab (``slab_offset`` = 0) or the storage
        of one shard (``slab_offset`` = that shard's start).  Only variables
        intersecting ``shard`` are touched when it is given.
        """
        lo, hi = (0, self.total) if shard is None else self.shard_range[shard]
        for s in self.specs:
            o, n = self.var_range[s.name]
            a, b = max(o, lo), min(o + n, hi)
            if a >= b:
                continue
            dst = slab[a - slab_offset:b - slab_offset]
            fill_initial_value(dst, s, seed, first_element=a - o)


def fill_initial_value(dst: torch.Tensor, spec: VarSpec, seed: int, first_element: int = 0) -> None:
    """dst <- elements [first_element, first_element + dst.numel()) of the variable's initial value."""
    kind = spec.init[0]
    if kind == "zeros":
        dst.zero_()
    elif kind == "constant":
        dst.fill_(float(spec.init[1]))
    elif kind == "normal":
        mean, std = float(spec.init[1]), float(spec.init[2])
        vseed = (int(seed) << 20) ^ (spec.index + 1)
        if dst.is_cuda:
            from .ops import hip_backend as hb

            if dst.is_contiguous():
                hb.philox_normal_(dst, vseed, offset=first_element, mean=mean, stddev=std)
            else:  # pragma: no cover
                tmp = torch.empty(dst.numel(), dtype=torch.float32, device=dst.device)
                hb.philox_normal_(tmp, vseed, offset=first_element, mean=mean, stddev=std)
                dst.copy_(tmp.view_as(dst))
        else:
            from .ops import reference as ref

            full = ref.random_normal((spec.numel,), vseed, 0, mean, std)
            dst.copy_(full[first_element:first_element + dst.numel()].view_as(dst))
    else:
        raise ValueError(f"unknown initializer {spec.init}")
