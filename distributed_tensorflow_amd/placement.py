"""Device strings and variable placement on parameter-server shards.

``replica_device_setter`` reproduces the behaviour the reference relies on
(tfdist_between.py:33-35, README.md:20-22): every *variable* goes to
``/job:ps/task:(i mod num_ps)`` where ``i`` counts variables in creation order;
every other op goes to the worker device.  With the reference's creation order
(global_step, W1, W2, b1, b2) and two PS tasks this yields
ps0 = {global_step, W2, b2} and ps1 = {W1, b1}.

Two further strategies exist because round-robin-by-count is badly unbalanced
for this model (W1 holds 98.6 % of the bytes):

* ``GreedyLoadBalancingStrategy`` — next variable goes to the least-loaded shard
  (TensorFlow's optional ``ps_strategy`` of the same name);
* ``"flat"`` layout in :mod:`variables` — the flattened parameter space is cut
  into equal ranges, one per shard, so reduce-scatter / all-gather move equal
  counts (one PS shard per GPU on an 8-GPU node).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Callable, Optional, Sequence


@dataclass(frozen=True)
class DeviceSpec:
    job: Optional[str] = None
    task: Optional[int] = None
    device_type: Optional[str] = None  # "gpu" | "cpu"
    device_index: Optional[int] = None

    _RE = re.compile(r"/(job|replica|task|device|gpu|cpu):([^/]+)")

    @classmethod
    def from_string(cls, s: Optional[str]) -> "DeviceSpec":
        if not s:
            return cls()
        job = task = dtype = dindex = None
        for key, val in cls._RE.findall(s.lower()):
            if key == "job":
                job = val
            elif key == "task":
                task = int(val)
            elif key in ("gpu", "cpu"):
                dtype, dindex = key, int(val)
            elif key == "device":
                # /device:GPU:0
                parts = val.split(":")
                dtype = parts[0]
                dindex = int(parts[1]) if len(parts) > 1 else 0
        return cls(job, task, dtype, dindex)

    def to_string(self) -> str:
        s = ""
        if self.job is not None:
            s += f"/job:{self.job}"
            if self.task is not None:
                s += f"/task:{self.task}"
        if self.device_type is not None:
            s += f"/{self.device_type}:{self.device_index or 0}"
        return s

    def merged_over(self, outer: "DeviceSpec") -> "DeviceSpec":
        """Fields of self win; missing ones are inherited (nested ``device`` scopes)."""
        return DeviceSpec(
            self.job if self.job is not None else outer.job,
            self.task if self.task is not None else outer.task,
            self.device_type if self.device_type is not None else outer.device_type,
            self.device_index if self.device_index is not None else outer.device_index,
        )


class RoundRobinStrategy:
    """i-th variable (creation order) -> shard i mod num_tasks."""

    def __init__(self, num_tasks: int):
        self.num_tasks = int(num_tasks)
        self._next = 0

    def __call__(self, name: str, nbytes: int) -> int:
        t = self._next
        self._next = (self._next + 1) % self.num_tasks
        return t


class GreedyLoadBalancingStrategy:
    """Next variable -> currently least-loaded shard (ties to the lowest index)."""

    def __init__(self, num_tasks: int):
        self.num_tasks = int(num_tasks)
        self.loads = [0] * self.num_tasks

    def __call__(self, name: str, nbytes: int) -> int:
        t = min(range(self.num_tasks), key=lambda i: (self.loads[i], i))
        self.loads[t] += int(nbytes)
        return t


class ReplicaDeviceSetter:
    """Device function object returned by :func:`replica_device_setter`."""

    def __init__(self, num_ps: int, worker_device: str, ps_device: str = "/job:ps",
                 ps_strategy: Optional[Callable[[str, int], int]] = None):
        self.num_ps = int(num_ps)
        self.worker_device = worker_device
        self.ps_job = DeviceSpec.from_string(ps_device).job or "ps"
        self.ps_strategy = ps_strategy or RoundRobinStrategy(max(self.num_ps, 1))
        self.assignments = []  # (variable name, device string) in creation order

    def variable_device(self, name: str, nbytes: int) -> str:
        if self.num_ps == 0:
            dev = self.worker_device
        else:
            dev = f"/job:{self.ps_job}/task:{self.ps_strategy(name, nbytes)}"
        self.assignments.append((name, dev))
        return dev

    def op_device(self) -> str:
        return self.worker_device

    def __call__(self, op_kind: str, name: str = "", nbytes: int = 0) -> str:
        return self.variable_device(name, nbytes) if op_kind == "variable" else self.op_device()


def replica_device_setter(ps_tasks: int = 0, ps_device: str = "/job:ps", worker_device: str = "/job:worker",
                          cluster=None, ps_strategy=None) -> ReplicaDeviceSetter:
    """Same signature subset as ``tf.train.replica_device_setter`` (cluster or ps_tasks)."""
    if cluster is not None:
        spec = cluster.as_dict() if hasattr(cluster, "as_dict") else dict(cluster)
        job = DeviceSpec.from_string(ps_device).job or "ps"
        ps_tasks = len(spec.get(job, []))
    return ReplicaDeviceSetter(ps_tasks, worker_device, ps_device, ps_strategy)


def placement_report(assignments: Sequence) -> str:
    """Human-readable variable -> device map (the ``log_device_placement`` output)."""
    return "\n".join(f"{name}: ({dev})" for name, dev in assignments)
