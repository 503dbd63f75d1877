"""Cluster runtime: ``ClusterSpec``, ``ConfigProto`` and the per-task ``Server``.

Surface of the reference (tfdist_between.py:9, 15-17, 27-29, 83)::

    cluster = ClusterSpec({"ps": ps_svrs, "worker": worker_svrs})
    config = ConfigProto(allow_soft_placement=True, log_device_placement=True)
    config.gpu_options.allow_growth = True
    server = Server(cluster, job_name=..., task_index=..., config=config)
    server.join()          # parameter-server tasks
    server.target          # workers: the session master

One OS process per task, started by hand.  A ``Server`` binds the task's own
``host:port`` from the cluster spec with the native task server
(runtime/ps_server.cpp), which replaces TensorFlow's gRPC server:

* a **ps** task hosts the shard's variables (host memory through the native
  server, or GPU memory exported over HIP-IPC — see ``parallel/ps_ipc.py``) and
  blocks in ``join()``.  Unlike the reference, where the PS has to be killed by
  hand, ``join()`` returns once every worker has reported completion
  (``ps_exit_when_workers_done``), and raises nothing if it never happens.
* a **worker** task maps itself to GPU ``task_index`` (reference
  tfdist_between.py:34: worker k -> gpu:k) when one exists, otherwise — with
  ``allow_soft_placement`` — to the CPU.
"""
from __future__ import annotations

import socket
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Union

import torch

from .runtime import native as rt


class ClusterSpec:
    def __init__(self, cluster: Union[Dict[str, Sequence[str]], "ClusterSpec"]):
        if isinstance(cluster, ClusterSpec):
            cluster = cluster.as_dict()
        self._spec: Dict[str, List[str]] = {}
        for job, tasks in cluster.items():
            if isinstance(tasks, dict):
                tasks = [tasks[i] for i in sorted(tasks)]
            self._spec[str(job)] = [str(t) for t in tasks]

    def as_dict(self) -> Dict[str, List[str]]:
        return {k: list(v) for k, v in self._spec.items()}

    @property
    def jobs(self) -> List[str]:
        return list(self._spec)

    def num_tasks(self, job_name: str) -> int:
        return len(self._spec.get(job_name, []))

    def task_address(self, job_name: str, task_index: int) -> str:
        try:
            return self._spec[job_name][task_index]
        except (KeyError, IndexError):
            raise ValueError(f"no task {task_index} in job {job_name!r} of {self._spec}") from None

    def job_tasks(self, job_name: str) -> List[str]:
        return list(self._spec.get(job_name, []))

    def __repr__(self):
        return f"ClusterSpec({self._spec})"

    def __eq__(self, other):
        return isinstance(other, ClusterSpec) and self._spec == other._spec


@dataclass
class GPUOptions:
    # PyTorch's caching allocator already grows on demand and never grabs the whole
    # device, which is what allow_growth=True asks of TensorFlow (README.md:49-55): several
    # tasks can share one GPU either way.  The field is kept for source compatibility.
    allow_growth: bool = True
    per_process_gpu_memory_fraction: float = 0.0
    visible_device_list: str = ""


@dataclass
class ConfigProto:
    allow_soft_placement: bool = False
    log_device_placement: bool = False
    gpu_options: GPUOptions = field(default_factory=GPUOptions)
    device_count: Dict[str, int] = field(default_factory=dict)
    # framework extensions
    transport: str = "tcp"                    # "tcp" (host-memory shards) | "ipc" (GPU-resident shards, one node)
    ps_exit_when_workers_done: bool = True
    operation_timeout_in_ms: int = 120000     # sync-barrier / rendezvous timeout instead of a silent hang


def _split_addr(addr: str):
    host, _, port = addr.rpartition(":")
    return host or "127.0.0.1", int(port)


def _resolvable(host: str) -> str:
    """Cluster files name hosts like 'server1'; fall back to loopback when a name does not resolve."""
    try:
        socket.getaddrinfo(host, None)
        return host
    except OSError:
        return "127.0.0.1"


_SERVERS: Dict[str, "Server"] = {}


def lookup_server(target: str) -> Optional["Server"]:
    return _SERVERS.get(target)


def _bind_address(host: str) -> str:
    """Interface the task server listens on: the one the cluster spec names.  ``localhost`` clusters (every test,
    every one-node run) stay on the loopback interface; a name or address of this machine binds that address only;
    anything that cannot be bound here falls back to all interfaces (``DTF_BIND_ADDRESS`` overrides).  The server
    has no authentication: it belongs on a loopback or a trusted cluster network."""
    import os
    import socket

    forced = os.environ.get("DTF_BIND_ADDRESS")
    if forced:
        return forced
    if host in ("localhost", "127.0.0.1", "", "::1"):
        return "127.0.0.1"
    try:
        ip = socket.gethostbyname(host)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as probe:
            probe.bind((ip, 0))              # only an address of this machine can be bound
        return ip
    except OSError:
        return "0.0.0.0"


class Server:
    def __init__(self, server_or_cluster_def, job_name: Optional[str] = None, task_index: Optional[int] = None,
                 protocol: Optional[str] = None, config: Optional[ConfigProto] = None, start: bool = True):
        self.cluster = ClusterSpec(server_or_cluster_def)
        self.job_name = job_name or ""
        self.task_index = int(task_index or 0)
        self.config = config or ConfigProto()
        if self.job_name not in self.cluster.jobs:
            raise ValueError(f"job_name={self.job_name!r} is not a job of the cluster {self.cluster.jobs}; "
                             f"launch with --job_name=ps|worker --task_index=N")
        self.address = self.cluster.task_address(self.job_name, self.task_index)
        self.host, self.port = _split_addr(self.address)
        self.target = f"dtf://{self.address}"
        self._native: Optional[rt.TaskServer] = None
        self._clients: Dict[str, rt.StoreClient] = {}
        self._joined = False
        if start:
            self.start()

    # -- lifecycle -------------------------------------------------------------
    def start(self):
        if self._native is None:
            self._native = rt.TaskServer(_bind_address(self.host), self.port)
            _SERVERS[self.target] = self
            if self.config.log_device_placement:
                print(f"[server] /job:{self.job_name}/task:{self.task_index} listening on :{self.port} "
                      f"(device {self.device})", flush=True)

    def join(self, timeout: Optional[float] = None) -> bool:
        """Block like ``tf.train.Server.join``; returns True once the cluster asked this task to exit."""
        if self._native is None:
            self.start()
        if self.job_name == "ps" and self.config.transport == "ipc" and timeout is None:
            from .parallel.ps_ipc import serve_ipc_shard

            done = serve_ipc_shard(self)      # allocates the GPU-resident shard on request, then waits
        else:
            done = self._native.wait(timeout)
        self._joined = True
        return done

    def stop(self):
        for c in self._clients.values():
            c.close()
        self._clients.clear()
        if self._native is not None:
            self._native.stop()
            self._native = None
        _SERVERS.pop(self.target, None)

    # -- topology ------------------------------------------------------------------
    @property
    def num_ps(self) -> int:
        return self.cluster.num_tasks("ps")

    @property
    def num_workers(self) -> int:
        return self.cluster.num_tasks("worker")

    @property
    def is_chief(self) -> bool:
        return self.job_name == "worker" and self.task_index == 0

    @property
    def device(self) -> torch.device:
        """worker k -> cuda:k (wrapping over the visible GPUs); soft placement falls back to the CPU."""
        if torch.cuda.is_available() and self.config.device_count.get("GPU", 1) != 0:
            n = torch.cuda.device_count()
            return torch.device("cuda", self.task_index % n)
        return torch.device("cpu")

    def client(self, job: str, task: int) -> rt.StoreClient:
        """Connection to another task's native server (opened on first use, retried until it is up)."""
        key = f"{job}:{task}"
        c = self._clients.get(key)
        if c is None:
            host, port = _split_addr(self.cluster.task_address(job, task))
            c = rt.StoreClient(_resolvable(host), port, timeout=self.config.operation_timeout_in_ms / 1000.0)
            self._clients[key] = c
        return c

    def new_client(self, job: str, task: int) -> rt.StoreClient:
        """A private connection (for threads that block in the server, e.g. the chief aggregator)."""
        host, port = _split_addr(self.cluster.task_address(job, task))
        return rt.StoreClient(_resolvable(host), port, timeout=self.config.operation_timeout_in_ms / 1000.0)

    def report_training_finished(self):
        """Count this worker's training loop as over (it stops contributing gradients).

        A synchronous chief keeps its aggregator alive until every worker has said so, and sizes
        the rounds by the workers that have not, so that nobody is stranded at the barrier."""
        if self.num_ps == 0 or self.job_name != "worker" or getattr(self, "_finished_reported", False):
            return
        self._finished_reported = True
        from .parallel.store import FINISHED_KEY

        self.client("ps", 0).add(FINISHED_KEY, 1)

    def report_worker_done(self):
        """Count this worker as finished; the last one releases the parameter servers from join()."""
        if self.num_ps == 0 or not self.config.ps_exit_when_workers_done:
            return
        done = self.client("ps", 0).add("__workers_done__", 1)
        if done >= self.num_workers:
            for k in range(self.num_ps):
                try:
                    self.client("ps", k).shutdown_server()
                except Exception:
                    pass
