"""Implementation of ``bench.py`` (kept in the package so tests can exercise it on CPU).

Flow of one rank (``run_benchmark``):

1. (several GPUs, mode auto) trial of the peer-mapped exchanges in a CHILD process per rank, before this process
   initialises the GPU, under a wall-time budget;
2. rendezvous (RCCL on GPUs, gloo on CPU), ``ranks_seen`` = all-reduce of ones;
3. resident data set (timed: ``setup_s``);
4. every candidate exchange runs the SAME measurement — W warm-up steps, then exactly K steps bracketed by a
   barrier + device synchronisation, maximum over ranks — library collectives first, peer-mapped exchanges after;
   a candidate whose build / capture fails on ANY rank is dropped on ALL ranks (collective verdicts, never a
   rank leaving a loop on its own); the fastest consistent one is the headline;
5. the reference's "Total Time" epoch with the headline trainer (``sec_per_epoch``, always);
6. the asynchronous modes on the SAME data set in this process (one-sided peer-mapped shards, two-sided
   send/recv), then in-graph replication (one client, N towers) on rank 0;
7. rank 0 prints one JSON line.

From step 4 on rank 0 holds a complete result; a watchdog thread prints it and ends the process should any later,
optional stage hang, and ends the process with an error should the run hang before there is a result.
"""
from __future__ import annotations

import json
import os
import sys
import threading
import time
from typing import Optional

import torch

BASELINE_SINGLE = 55000.0 / 1.3          # samples/s: reference README.md:15 (1.3 s/epoch)
BASELINE_MULTI = 3 * 55000.0 / 6.03      # samples/s: reference README.md:249-251 (mean of 5.75 / 5.85 / 6.5 s)
BATCH = 100
STEPS_PER_EPOCH = 550
LR = 0.001
PEER_MODES = ("xgmi", "xgmi2", "xgmi2c", "xgmi2p")      # synchronous exchanges over peer-mapped HBM (parallel/xgmi.py)
SYNC_MODES = ("towers", "sync") + PEER_MODES
ASYNC_MODES = ("xgmi_async", "async_rccl", "async_lockstep")


def _dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _sync(device, distributed: bool):
    import torch.distributed as dist

    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _aligned_start(device, distributed: bool, lead_s: float = 0.002) -> None:
    """The opening bracket of a timed region: barrier + device synchronisation and then, on every rank, a wait for the
    SAME instant of the node's monotonic clock (rank 0 names it).  Ranks leave a barrier up to milliseconds apart, and
    replicas that wait for each other inside their kernels would book that skew as step time: with 20 timed steps of
    10 us it is most of the bracket."""
    import torch.distributed as dist

    _sync(device, distributed)
    if not distributed:
        return
    t = torch.tensor([time.monotonic() + lead_s], dtype=torch.float64, device=device)
    dist.broadcast(t, src=0)
    target = float(t.item())
    if target - time.monotonic() > 1.0:         # clocks that do not agree (not one node): no common instant
        return
    while time.monotonic() < target:
        pass


class Watchdog:
    """Bounds the wall time of the run.  ``hold(result)`` registers what rank 0 prints if the deadline passes
    (or ``emit`` is never reached); without a held result an expiry ends the process with status 3."""

    def __init__(self, rank: int, seconds: float):
        self.rank, self.deadline = rank, time.monotonic() + seconds
        self._t0 = time.monotonic()
        self._lock = threading.Lock()
        self._result = None
        self._emitted = False
        self._stage = "start"
        self._stage_deadline = None
        self._stage_began = time.monotonic()
        self.stage_s = {}                   # wall seconds per stage, in the order the stages ran
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def stage(self, name: str, limit_s: Optional[float] = None):
        """Names what runs now; ``limit_s`` bounds THIS stage (an optional one that may hang in a collective)."""
        now = time.monotonic()
        self.stage_s[self._stage] = round(self.stage_s.get(self._stage, 0.0) + now - self._stage_began, 2)
        if self.rank == 0 and os.environ.get("DTF_BENCH_QUIET") != "1":
            print(f"[bench] +{now - self._t0:6.1f} s  {name}", file=sys.stderr, flush=True)
        self._stage, self._stage_began = name, now
        self._stage_deadline = None if limit_s is None else now + limit_s

    def stages(self) -> dict:
        """Seconds per stage so far, the running one included."""
        out = dict(self.stage_s)
        out[self._stage] = round(out.get(self._stage, 0.0) + time.monotonic() - self._stage_began, 2)
        return out

    def hold(self, result: Optional[dict]):
        with self._lock:
            self._result = None if result is None else dict(result)

    def extend_to(self, seconds_from_now: float):
        self.deadline = max(self.deadline, time.monotonic() + seconds_from_now)

    def emit(self, result: Optional[dict]) -> Optional[dict]:
        """The normal end: returns the result to print (None when the watchdog already printed one)."""
        with self._lock:
            if self._emitted:
                return None
            self._emitted = True
            return result

    def _run(self):
        while time.monotonic() < min(self.deadline, self._stage_deadline or self.deadline):
            time.sleep(0.25)
            if self._emitted:
                return
        with self._lock:
            if self._emitted:
                return
            self._emitted = True
            res = self._result
        print(f"[bench rank {self.rank}] wall-time limit reached in stage '{self._stage}'", file=sys.stderr, flush=True)
        try:                        # where every thread of this rank stands: the one diagnosis a hang leaves behind
            import faulthandler

            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        except Exception:  # noqa: BLE001
            pass
        if self.rank == 0 and res is not None:
            res["truncated_at_stage"] = self._stage
            res["stage_s"] = self.stages()
            print(json.dumps(res), flush=True)
        os._exit(0 if res is not None or self.rank != 0 else 3)


def _trial_xgmi(rank: int, local_rank: int, world: int, chunk: int, budget_s: float = 75.0) -> dict:
    """Run the peer-mapped exchanges once in a child process per rank (see parallel/xgmi_probe.py).

    Called before this process initialises the GPU.  Whatever happens to the child - a refused
    mapping, a timeout in the exchange, replicas that disagree, a crash - stays in the child.  The child
    stops starting new stages once ``budget_s`` is spent; it is killed ``budget_s + 45`` s after its start."""
    import subprocess

    port = int(os.environ.get("MASTER_PORT", "29500")) + 37
    if port > 65000:
        port -= 1000
    cmd = [sys.executable, "-m", "distributed_tensorflow_amd.parallel.xgmi_probe", "--rank", str(rank),
           "--world", str(world), "--device", str(local_rank), "--port", str(port), "--chunk", str(chunk),
           "--rows", "11000", "--steps", "150", "--budget", str(budget_s)]
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
    # The child makes a rendezvous of its own: nothing of the launcher's may leak into it (with
    # TORCHELASTIC_USE_AGENT_STORE set, rank 0 would not host the store it is asked to connect to).
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                  "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE",
                                                  "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE"):
            env.pop(k)

    def attempt(extra, port_shift, limit):
        cmd[cmd.index("--port") + 1] = str(port + port_shift)
        try:
            r = subprocess.run(cmd + extra, env=env, stdout=subprocess.PIPE, stderr=None, text=True, timeout=limit)
            text, note = r.stdout, f"trial exited {r.returncode}"
        except subprocess.TimeoutExpired as e:      # the stages that finished have already reported
            text = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            note = "trial timed out"
        for line in reversed(text.splitlines()):
            if line.startswith("{"):
                try:
                    return json.loads(line)
                except ValueError:
                    break
        return {"ok": False, "error": f"{note}: {text[-300:]}"}

    t0 = time.monotonic()
    res = attempt([], 0, budget_s + 45.0)
    res["uncached_grads"] = False
    # Replicas that ran to the end but disagree read a peer's gradient stale: every rank sees the same verdict
    # (it is computed from all ranks' checksums), so every rank repeats the trial with the slabs uncached too.
    left = budget_s - (time.monotonic() - t0)
    if not res.get("ok") and res.get("identical") is False and res.get("healthy") and left > 20.0:
        cmd[cmd.index("--budget") + 1] = str(left)
        res = attempt(["--uncached", "1", "--only", "xgmi"], 1, left + 30.0)
        res["uncached_grads"] = True
    res["trial_s"] = round(time.monotonic() - t0, 1)
    return res


def _child_async_rccl(rank: int, dev_index: int, world: int, steps: int, warmup: int, limit_s: float = 60.0) -> dict:
    """The two-sided asynchronous transport over RCCL, measured by a child process per rank that starts its HIP
    runtime with a larger pool of hardware queues (parallel/async_rccl_probe.py says why)."""
    import subprocess

    port = int(os.environ.get("MASTER_PORT", "29500")) + 61
    if port > 65000:
        port -= 1000
    cmd = [sys.executable, "-m", "distributed_tensorflow_amd.parallel.async_rccl_probe", "--rank", str(rank),
           "--world", str(world), "--device", str(dev_index), "--port", str(port), "--steps", str(steps),
           "--warmup", str(warmup)]
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                  "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE",
                                                  "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE"):
            env.pop(k)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, timeout=limit_s)
        text, note = r.stdout, f"child exited {r.returncode}"
    except subprocess.TimeoutExpired as e:
        text = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        note = f"child killed after {limit_s:.0f} s"
    for line in reversed(text.splitlines()):
        if line.startswith("{"):
            try:
                doc = json.loads(line)
                doc["measured_in"] = "child processes (larger hardware-queue pool), on the benchmark's data set"
                return doc
            except ValueError:
                break
    return {"error": f"{note}: {text[-300:]}"}


def _child_dense_dp(rank: int, dev_index: int, world: int, backend: str, limit_s: float) -> dict:
    """The data-parallel dense-stack stage in a child process per rank with a rendezvous of its own
    (parallel/dense_dp_probe.py): a crash or a hang in there costs the caller nothing but this stage."""
    import subprocess

    port = int(os.environ.get("MASTER_PORT", "29500")) + 83
    if port > 65000:
        port -= 1000
    cmd = [sys.executable, "-m", "distributed_tensorflow_amd.parallel.dense_dp_probe", "--rank", str(rank),
           "--world", str(world), "--device", str(dev_index), "--port", str(port), "--backend", backend]
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                  "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE",
                                                  "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE"):
            env.pop(k)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, timeout=limit_s)
        text, note = r.stdout, f"child exited {r.returncode}"
    except subprocess.TimeoutExpired as e:
        text = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        note = f"child killed after {limit_s:.0f} s"
    for line in reversed(text.splitlines()):
        if line.startswith("{"):
            try:
                doc = json.loads(line)
                doc["measured_in"] = "child processes with a rendezvous of their own"
                return doc
            except ValueError:
                break
    return {"error": f"{note}: {text[-300:]}"}


def _child_ingraph(n: int, steps: int, warmup: int, chunk: int, limit_s: float = 100.0, one_device: bool = False) -> dict:
    """In-graph replication measured by a client process of its own (``bench.py --mode ingraph``): one engine per
    process is how the engine is used and tested, and a rank that has built half a dozen trainers before is not that;
    whatever happens to the client stays in the client."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--mode", "ingraph", "--steps", str(steps),
           "--warmup", str(warmup), "--chunk", str(chunk)]
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                  "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE",
                                                  "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE"):
            env.pop(k)
    # the client itself, not a launcher with the client as ITS child (which a kill at the time limit would orphan); and
    # its own watchdog ends it should the kill not reach it
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", DTF_BENCH_DEADLINE_S=str(int(limit_s)))
    if one_device and int(env.get("GPU_MAX_HW_QUEUES", "4")) < 16:
        # towers that share a GPU (the rehearsal on a one-GPU machine) wait for each other inside their launches and
        # need a hardware queue each: the client asks the runtime for 16 (read when the runtime starts) and may then
        # run four towers instead of two
        env["GPU_MAX_HW_QUEUES"] = "16"
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, timeout=limit_s)
        text, note = r.stdout, f"client exited {r.returncode}"
    except subprocess.TimeoutExpired as e:
        text = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        note = f"client killed after {limit_s:.0f} s"
    for line in reversed(text.splitlines()):
        if line.startswith("{"):
            try:
                doc = json.loads(line)
            except ValueError:
                break
            rep = {k: doc[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup") if k in doc}
            rep["parallelism"] = doc.get("config", {}).get("parallelism")
            for k in ("towers", "replicas_identical", "healthy", "global_step", "host_calls_per_tower_per_chunk",
                      "test_accuracy_after_run", "test_loss_after_run", "by_exchange_ms_per_step"):
                if k in doc:
                    rep[k] = doc[k]
            rep["measured_in"] = "a client process of its own"
            return rep
    return {"error": f"{note}: {text[-300:]}"}


def _child_clusters(use_cpu: bool, limit_s: float = 60.0) -> dict:
    """The reference's between-graph recipes as it runs them (README.md:88-113, 150-176 there): the repository's
    ``tfdist_between.py`` / ``tfdist_between_sync.py`` as one process per task -- asynchronous 1 ps + 1 worker,
    1 ps + 2 workers and 2 ps + 2 workers (the variables dealt round-robin over the two shards), synchronous 1 ps +
    2 workers -- with all tasks on this GPU and the shards in the ps task's HBM
    (``--transport=ipc``; on the CPU, where the stage only runs when DTF_BENCH_CLUSTERS=1 asks for it: tcp and 1,100
    training rows).  Reports the slowest worker's fastest epoch
    after the first, from the scripts' own epoch clocks; recipes with several workers are run twice on a GPU and the
    faster run is reported (see the note in the loop)."""
    from .utils.local_cluster import run_local_cluster

    env = dict(os.environ)
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k)
    extra = []
    if use_cpu:
        env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
        extra = ["--train_size=1100"]
    out = {"what": "the scripts as separate tasks (one process each, all on this GPU), one sess.run per batch, "
                   + ("tcp shards on the host" if use_cpu else "shards in the ps task's HBM (HIP-IPC)")
                   + "; the slowest worker's fastest epoch after the first"}
    for key, script, n_ps, n_w in (("async_1ps_1w", "tfdist_between.py", 1, 1), ("async_1ps_2w", "tfdist_between.py", 1, 2),
                                   ("async_2ps_2w", "tfdist_between.py", 2, 2), ("sync_1ps_2w", "tfdist_between_sync.py", 1, 2)):
        # Client processes that time-share ONE GPU take turns when their hardware queues happen to land on one pipe of
        # it (decided when the queues are created: the same recipe then runs at half its speed from start to end, in
        # about one run in four with two workers).  A worker with a GPU of its own has no such neighbour, so a recipe
        # with several workers is run twice here and the faster run is the one reported; both are kept in the record.
        best, runs = None, []
        for _ in range(2 if (n_w > 1 and not use_cpu) else 1):
            try:
                r = run_local_cluster(script, n_ps, n_w, "tcp" if use_cpu else "ipc", epochs=4, extra=extra,
                                      timeout=limit_s, env=env)
            except Exception as e:  # noqa: BLE001
                runs.append({"error": f"{type(e).__name__}: {e}"})
                continue
            workers = [t for name, t in r["tasks"].items() if name.startswith("worker")]
            # per worker the fastest epoch after the first (as for tfsingle.py above), then the slowest worker
            steady = [min(t["epoch_times"][1:]) if len(t.get("epoch_times") or ()) > 1 else None for t in workers]
            if r["returncode"] != 0 or not steady or any(v is None for v in steady):
                runs.append({"error": "the cluster did not finish",
                             "tasks": {k: {kk: vv for kk, vv in t.items() if kk != "epoch_times"} for k, t in r["tasks"].items()}})
                continue
            rep = {"sec_per_epoch": round(max(steady), 5), "workers": n_w, "ps": n_ps,
                   "epoch_times": [t["epoch_times"] for t in workers],
                   "last_step": [t.get("last_step") for t in workers],
                   "final_accuracy": [t.get("final_accuracy") for t in workers]}
            runs.append({"sec_per_epoch": rep["sec_per_epoch"]})
            if best is None or rep["sec_per_epoch"] < best["sec_per_epoch"]:
                best = rep
        if best is None:
            out[key] = runs[-1] if runs else {"error": "not run"}
            continue
        if len(runs) > 1:
            best["runs_sec_per_epoch"] = [r.get("sec_per_epoch", r.get("error")) for r in runs]
        out[key] = best
    return out


def _child_per_batch_epoch(use_cpu: bool, limit_s: float = 90.0) -> dict:
    """The reference's LITERAL epoch (tfsingle.py:72-97): 550 x {next_batch, ``sess.run([train_op, cost, summary,
    global_step], feed_dict)``, ``add_summary``} + the full test-set evaluation, timed by the script itself as
    "Total Time".  Runs the repository's ``tfsingle.py`` unmodified in a process of its own for 3 epochs and reports the
    fastest of the epochs after the first (the first one holds kernel loading and the event file's creation)."""
    import re
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k)
    if use_cpu:
        env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    exact = None
    with tempfile.TemporaryDirectory() as logs:
        cmd = [sys.executable, os.path.join(root, "tfsingle.py"), "--epochs", "3", "--logs_path", logs]
        if use_cpu:
            cmd += ["--train_size", "5500"]
        env["DTF_EPOCH_TIMES_FILE"] = os.path.join(logs, "epoch_times.json")     # the same times, not rounded to 10 ms
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=limit_s,
                               cwd=logs)
            text = r.stdout
        except subprocess.TimeoutExpired as e:
            text = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        try:
            with open(env["DTF_EPOCH_TIMES_FILE"]) as f:
                exact = [float(v) for v in json.load(f)["epoch_times"]]
        except (OSError, ValueError, KeyError):
            exact = None
    times = [float(v) for v in re.findall(r"Total Time: ([\d.]+)s", text)]
    steps = [int(v) for v in re.findall(r"Batch:\s*\d+ of\s*(\d+)", text)]
    accs = re.findall(r"Test-Accuracy: ([\d.]+)", text)
    if len(times) < 2:
        return {"error": "tfsingle.py did not finish two epochs: " + text[-300:]}
    # (the script prints seconds with two decimals; the per-step figure comes from its AvgTime lines)
    avg = [float(v) for v in re.findall(r"AvgTime:\s*([\d.]+)ms", text)]
    steady = avg[6:] if len(avg) > 6 else avg
    printed = min(times[1:])
    if exact and len(exact) == len(times):
        times = exact
    return {"sec_per_epoch": round(min(times[1:]), 5), "sec_per_epoch_as_printed": printed,
            "first_epoch_s": round(times[0], 4), "epochs_run": len(times),
            "steps_per_epoch": steps[-1] if steps else None,
            "ms_per_batch_steady": round(sorted(steady)[len(steady) // 2], 4) if steady else None,
            "test_accuracy": float(accs[-1]) if accs else None,
            "what": "tfsingle.py unmodified: one sess.run with feed_dict and four fetches per batch, one event-file "
                    "record per batch, full test-set evaluation per epoch; the script's own 'Total Time'"}


def run_benchmark(args) -> Optional[dict]:
    import torch.distributed as dist

    from .data import DeviceDataset, read_data_sets
    from .parallel.replicas import build_replica_trainer

    t_begin = time.monotonic()
    rank, local_rank, world = _dist_env()
    n = int(args.gpus)
    mode = args.mode or ("single" if n == 1 else "auto")
    ingraph_only = mode == "ingraph"
    distributed = world > 1
    if distributed and world != n:
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}")
    if n > 1 and not distributed and not ingraph_only:
        raise SystemExit("--gpus N>1 needs one process per GPU: run `python bench.py --gpus N` without RANK / "
                         "WORLD_SIZE in the environment (it starts the ranks itself) or under torch.distributed.run")
    if mode == "async":     # "the asynchronous mode": one-sided shards on GPUs, two-sided send/recv elsewhere
        mode = "xgmi_async" if (torch.cuda.device_count() > 0 and not args.cpu
                                and os.environ.get("DTF_BENCH_SHARE_GPU") != "1" and not distributed) else "async_rccl"
    if not int(args.chunk):
        # one graph per epoch for the single-GPU step (8.20 vs 8.43 us/step with 50-step graphs: the fixed cost of
        # a replay is paid 11 times less often); the multi-rank modes keep their graphs short
        args.chunk = STEPS_PER_EPOCH if mode == "single" else 50
    wd = Watchdog(rank, float(os.environ.get("DTF_BENCH_DEADLINE_S", "900")))
    soft_budget = float(os.environ.get("DTF_BENCH_BUDGET_S", "170"))     # no NEW optional stage after this
    # device_count() does not create a GPU context, so the trial below still runs ahead of ours
    use_cpu = bool(args.cpu) or torch.cuda.device_count() == 0
    # Test hook: every rank on GPU 0 with a gloo rendezvous (RCCL refuses two ranks on one device), so that
    # the whole multi-rank flow can be rehearsed on a one-GPU machine.
    share_gpu = os.environ.get("DTF_BENCH_SHARE_GPU") == "1" and not use_cpu
    dev_index = 0 if share_gpu else local_rank
    trial = None
    if mode == "auto" and distributed and not use_cpu and os.environ.get("DTF_BENCH_XGMI", "1") != "0":
        wd.stage("peer-exchange trial (child processes)")
        trial = _trial_xgmi(rank, dev_index, world, int(args.chunk),
                            budget_s=float(os.environ.get("DTF_BENCH_TRIAL_S", "45")))
    use_cpu = use_cpu or not torch.cuda.is_available()
    if use_cpu:
        device = torch.device("cpu")
    else:
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    ranks_seen = 1
    if distributed:
        import datetime

        wd.stage("rendezvous")
        lib_backend = not (use_cpu or share_gpu)
        dist.init_process_group(backend="nccl" if lib_backend else "gloo", device_id=device if lib_backend else None,
                                timeout=datetime.timedelta(seconds=300))
        ones = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(ones)                           # the first collective: every rank the library sees adds 1
        ranks_seen = int(round(float(ones.item())))

    def agree(ok: bool) -> bool:
        """True only when ``ok`` on every rank (one all-reduce; every rank must call it at the same place)."""
        if not distributed:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def within_budget() -> bool:
        """Rank 0's clock decides for everybody whether another optional stage is started."""
        ok = time.monotonic() - t_begin < soft_budget
        if not distributed:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.broadcast(flag, src=0)
        return bool(int(flag.item()))

    trial_ok, trial_ms = {}, {}
    if trial is not None:
        keys = ("ok", "async_ok", "ok2", "ok2c", "ok2p")
        flag = torch.tensor([1 if trial.get(k) else 0 for k in keys], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        trial_ok = dict(zip(keys, (bool(int(v)) for v in flag.tolist())))
        if rank == 0 and not all(trial_ok.values()):
            print(f"[bench] peer-mapped trial verdicts {trial_ok}: {trial}", file=sys.stderr, flush=True)
        # the trial's own timings (maxima over its ranks, made identical here) rank the exchanges that passed
        tkeys = ("ms_per_step", "ms_per_step2", "ms_per_step2c", "ms_per_step2p")
        tms = torch.tensor([float(trial.get(k) or 1e9) for k in tkeys], dtype=torch.float64, device=device)
        dist.all_reduce(tms, op=dist.ReduceOp.MAX)
        trial_ms = dict(zip(("xgmi", "xgmi2", "xgmi2c", "xgmi2p"), (float(v) for v in tms.tolist())))

    # ---- resident data: every worker trains on its own independently shuffled copy (tfdist_between.py:92) ------
    wd.stage("data set")
    t_setup = time.perf_counter()
    sizes = (55000, 5000, 10000) if not use_cpu else (5500, 100, 1000)
    mnist = read_data_sets("MNIST_data", one_hot=True, seed=0, sizes=sizes)
    t_host = time.perf_counter()
    ds = DeviceDataset(mnist.train.images, mnist.train.labels, device, BATCH, seed=1000 + rank)
    test_x = torch.as_tensor(mnist.test.images).to(device)
    test_y = torch.as_tensor(mnist.test.labels).to(device)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    setup = {"host_data_s": round(t_host - t_setup, 3), "upload_and_first_shuffle_s": round(time.perf_counter() - t_host, 4)}
    steps, warmup = int(args.steps), int(args.warmup)
    bpe = ds.batches_per_epoch
    epoch_windows = (100,) * (bpe // 100) + ((bpe % 100,) if bpe % 100 else ())

    def build(m, **extra):
        # gloo collectives (the one-GPU rehearsal) cannot be captured; the peer-mapped exchange can
        graph = not args.no_graph and not (share_gpu and m not in PEER_MODES + ("xgmi_async",))
        if m == "xgmi":
            extra.setdefault("uncached_grads", bool(trial and trial.get("uncached_grads")))
        return build_replica_trainer(mode=m, device=device, dataset=ds, batch=BATCH, lr=LR, seed=1,
                                     rank=rank, world=world, chunk=int(args.chunk), use_graph=graph, **extra)

    def prepare(trainer, hint):
        # Capture the hipGraph chunks outside the timed region.  If capture fails on any rank
        # (e.g. an RCCL build that cannot be captured) every rank falls back to eager launches.
        ok = True
        try:
            trainer.prepare(steps_hint=hint)
        except Exception as e:  # noqa: BLE001
            ok = False
            print(f"[bench rank {rank}] graph capture failed ({type(e).__name__}: {e}); using eager launches",
                  file=sys.stderr, flush=True)
            if device.type == "cuda":
                # a capture that died may leave its side stream current: go back to the default stream
                torch.cuda.set_stream(torch.cuda.default_stream(device))
        if not agree(ok) and trainer.use_graph:
            trainer.use_graph = False
            trainer._graphs.clear()
            trainer.engine.reset_step_workspace(trainer.engine.parity)

    def retire(tr):
        if tr is None:
            return
        try:
            tr._graphs.clear()
            if hasattr(tr, "close"):
                _sync(device, distributed)
                tr.close()
        except Exception as e:  # noqa: BLE001 — a candidate on its way out must not take the run with it
            print(f"[bench rank {rank}] closing a trainer failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)

    def measure(cand: str, w: int, k: int, extra_hint=(), **extra):
        """(trainer, seconds for k steps [max over ranks], info) or (None, None, {"error": ...}); the verdict on
        every local failure is taken by ALL ranks together, after the local ``try`` is over."""
        tr, err = None, None
        t_b = time.perf_counter()
        try:
            tr = build(cand, **extra)
        except Exception as e:  # noqa: BLE001
            err = f"build: {type(e).__name__}: {e}"
        if not agree(err is None):
            retire(tr)
            return None, None, {"error": err or "failed on another rank"}
        prepare(tr, (w, k) + tuple(extra_hint))
        build_s = time.perf_counter() - t_b
        dev_ms = 0.0
        try:
            tr.run_steps(w)
            _aligned_start(device, distributed)
            # the K steps on the device's own clock as well: two events inside the bracket (the host number between the
            # barriers also holds the launch of the first graph from an idle device and the closing barrier)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if device.type == "cuda" else None
            if ev:
                ev[0].record()
            t0 = time.perf_counter()
            tr.run_steps(k)
            if ev:
                ev[1].record()
            _sync(device, distributed)
            dt = time.perf_counter() - t0
            if ev:
                dev_ms = float(ev[0].elapsed_time(ev[1]))
        except Exception as e:  # noqa: BLE001
            err, dt = f"run: {type(e).__name__}: {e}", 0.0
        if not agree(err is None):
            retire(tr)
            return None, None, {"error": err or "failed on another rank"}
        if distributed:
            t = torch.tensor([dt, dev_ms], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)        # identical decisions on every rank
            dt, dev_ms = float(t[0].item()), float(t[1].item())
        info = {"build_and_capture_s": round(build_s, 3)}
        if device.type == "cuda":
            info["ms_per_step_device"] = round(dev_ms / k, 5)
        if cand in PEER_MODES:
            # The replicas sum in rank order, so they must be bit-identical; anything else (or an exchange that
            # timed out, which skips its work from then on and would look FAST) voids the measurement.
            good = torch.tensor([1 if tr.healthy() else 0, tr.replica_checksum()], dtype=torch.int64, device=device)
            seen = [torch.zeros_like(good) for _ in range(world)] if distributed else [good]
            if distributed:
                dist.all_gather(seen, good)
            if not all(int(g[0]) == 1 and int(g[1]) == int(seen[0][1]) for g in seen):
                retire(tr)
                return None, None, {"error": "replicas disagree or an exchange timed out"}
        return tr, dt, info

    # ---- the synchronous (or single-process) headline -----------------------------------------------------------
    mode_err_pruned = {}
    if mode == "auto":
        # library collectives first: they are the safe baseline and give the run a result early
        if share_gpu:       # (the one-GPU rehearsal keeps its gloo group out of the per-step path when it can)
            library = () if any(trial_ok.get(k) for k in ("ok", "ok2", "ok2c", "ok2p")) else ("towers",)
        else:
            library = ("towers", "sync")
        cands = list(library)
        if not use_cpu:
            # of the peer-mapped exchanges that passed the trial, the fastest ones by the trial's timings (2 by default:
            # every candidate costs a build, a capture and a measurement on every rank)
            passed = [m for m, k in (("xgmi2p", "ok2p"), ("xgmi2c", "ok2c"), ("xgmi2", "ok2"), ("xgmi", "ok"))
                      if trial_ok.get(k)]
            passed.sort(key=lambda m: trial_ms.get(m, 1e9))
            keep = max(1, int(os.environ.get("DTF_BENCH_PEER_CANDS", "2")))
            cands += passed[:keep]
            for m in passed[keep:]:
                mode_err_pruned[m] = f"not measured: {trial_ms.get(m):.5f} ms/step in the trial, slower than {passed[:keep]}"
    elif ingraph_only:
        cands = []
    else:
        cands = [mode]
    mode_ms, mode_err, mode_dev_ms = {}, dict(mode_err_pruned), {}
    best = None                         # (mode, trainer, seconds, info)
    for i, cand in enumerate(cands):
        if best is not None and not within_budget():
            mode_err[cand] = "skipped: wall-time budget spent"
            continue
        if rank == 0 and len(cands) > 1:
            print(f"[bench] measuring {cand}", file=sys.stderr, flush=True)
        wd.stage(f"measuring {cand}")
        tr, dt, info = measure(cand, warmup, steps, extra_hint=epoch_windows)
        if tr is None:
            mode_ms[cand] = None
            mode_err[cand] = info["error"]
            if rank == 0:
                print(f"[bench] {cand} cannot be used here: {info['error']}", file=sys.stderr, flush=True)
            continue
        mode_ms[cand] = round(1000.0 * dt / steps, 5)
        if "ms_per_step_device" in info:
            mode_dev_ms[cand] = info["ms_per_step_device"]
        gs = tr.global_step_value()
        if best is None or dt < best[2]:
            if best is not None:
                retire(best[1])
            best = (cand, tr, dt, dict(info, global_step=gs))
        else:
            retire(tr)
        if rank == 0:       # from here on there is something to report whatever happens later
            wd.hold(_result(n, steps, warmup, best[2], best[1].describe(), mnist.source, use_cpu, device, ranks_seen,
                            {"global_step": best[3]["global_step"], "mode_ms_per_step": dict(mode_ms),
                             "note": "later stages did not finish"}))
    if not ingraph_only and best is None:
        raise SystemExit(f"no data-parallel mode could run on this node: {mode_err}")

    result_extra = {}
    if best is not None:
        mode, trainer, elapsed, info = best
        setup["build_and_capture_s"] = info["build_and_capture_s"]
        result_extra["global_step"] = info["global_step"]
        if "ms_per_step_device" in info:
            result_extra["ms_per_step_device"] = info["ms_per_step_device"]
        wd.stage("evaluation + like-for-like epoch")
        loss, acc = trainer.evaluate(test_x, test_y)
        last_cost = trainer.last_loss()
        # Like-for-like epoch time: one epoch of steps + a host fetch of the cost per 100 steps + the full
        # test-set evaluation, as the reference's "Total Time" does (tfdist_between.py:86-111).
        _sync(device, distributed)
        te = time.perf_counter()
        # (the cost of a window reaches the host while the next window runs, as in examples/mnist_mlp.py's loop)
        pending = None
        for k in epoch_windows:
            first = trainer.engine._host_cursor if hasattr(trainer, "engine") else 0
            trainer.run_steps(k)
            if hasattr(trainer, "fetch_window_async"):
                handle = trainer.fetch_window_async(first, k)
                if pending is not None:
                    pending.result()
                pending = handle
            else:
                trainer.last_loss()
        if pending is not None:
            last_cost = pending.result()[0]
        trainer.evaluate(test_x, test_y)
        _sync(device, distributed)
        epoch_s = time.perf_counter() - te
        if distributed:
            t = torch.tensor([epoch_s], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            epoch_s = float(t.item())
        result_extra.update(sec_per_epoch=round(epoch_s, 5), steps_per_epoch=bpe,
                            sec_per_epoch_baseline=1.3 if n == 1 else 5.35,
                            test_accuracy_after_run=round(acc, 4), test_loss_after_run=round(loss, 4),
                            last_cost=round(last_cost, 4))
        # cost of one reshuffle of the resident set (gather + Gram blocks), which the steady-state numbers include
        if device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(device)
            e0.record()
            for _ in range(3):
                ds.shuffle_epoch()
            e1.record()
            torch.cuda.synchronize(device)
            setup["epoch_refresh_ms"] = round(e0.elapsed_time(e1) / 3.0, 4)
        parallelism = trainer.describe()
        retire(trainer)
        if len(cands) > 1:
            result_extra["mode_ms_per_step"] = mode_ms
            if mode_dev_ms:
                result_extra["mode_ms_per_step_device"] = mode_dev_ms
        if mode_err:
            result_extra["mode_errors"] = mode_err
        if trial is not None:
            result_extra["peer_exchange_trial"] = {k: trial.get(k) for k in (
                "ok", "ms_per_step", "identical", "healthy", "uncached_grads", "async_ok", "async_fused_push",
                "async_ms_per_step_by_push", "async_ms_per_step", "async_loss", "ok2", "ms_per_step2", "ok2c",
                "ms_per_step2c", "ok2p", "ms_per_step2p", "error", "error2", "error2c", "error2p", "trial_s", "skipped") if k in trial}
        result_extra["setup_s"] = setup
        result_extra["wall_s_until_headline"] = round(time.monotonic() - t_begin, 1)
        if rank == 0:
            wd.hold(_result(n, steps, warmup, elapsed, parallelism, mnist.source, use_cpu, device, ranks_seen, result_extra))

    # ---- the asynchronous modes, on the same data set, in this process -------------------------------------------------
    async_reports = {}
    want_async = mode not in ASYNC_MODES and not ingraph_only and os.environ.get("DTF_BENCH_ASYNC", "1") != "0"
    if want_async and (args.mode in (None, "auto", "single")):
        a_cands = []
        if not use_cpu and (not distributed or trial_ok.get("async_ok")):
            a_cands.append("xgmi_async")
        if distributed and (use_cpu or share_gpu or os.environ.get("DTF_BENCH_ASYNC_RCCL", "1") != "0"):
            a_cands.append("async_rccl")
        for cand in a_cands:
            if not within_budget():
                async_reports[cand] = {"error": "skipped: wall-time budget spent"}
                continue
            wd.stage(f"measuring {cand}")
            # the two-sided transport is host-driven (eager point-to-point calls): a short run says what it costs
            a_w = min(warmup, bpe) if cand == "xgmi_async" else min(warmup, 20)
            a_k = min(steps, 4 * bpe) if cand == "xgmi_async" else min(steps, 200)
            if cand == "async_rccl" and not (use_cpu or share_gpu):
                _sync(device, distributed)
                rep = _child_async_rccl(rank, dev_index, world, a_k, a_w)
                rep.pop("ok", None)
                rep.pop("rank", None)
                async_reports[cand] = rep
                _sync(device, distributed)              # every rank's child is gone before anything else starts
                continue
            tr, dt, info = measure(cand, a_w, a_k)
            if tr is None:
                async_reports[cand] = {"error": info["error"]}
                continue
            try:
                _sync(device, distributed)                       # every worker's pushes have landed
                a_gs = tr.global_step_value()
                a_loss, a_acc = tr.evaluate(test_x, test_y)
                async_reports[cand] = {
                    "value": round(n * BATCH * a_k / dt, 1), "unit": "samples/s", "ms_per_step": round(1000.0 * dt / a_k, 5),
                    "ms_per_step_device": info.get("ms_per_step_device"),
                    "steps": a_k, "warmup": a_w, "global_step": a_gs, "every_push_counted": a_gs == n * (a_w + a_k),
                    "test_accuracy_after_run": round(a_acc, 4), "test_loss_after_run": round(a_loss, 4),
                    "measured_in": "this process, on the benchmark's data set", "parallelism": tr.describe()}
            except Exception as e:  # noqa: BLE001 — the extra measurement must never cost the headline one
                async_reports[cand] = {"error": f"{type(e).__name__}: {e}"}
            retire(tr)
            if rank == 0 and best is not None:
                result_extra["async_modes"] = dict(async_reports)
                wd.hold(_result(n, steps, warmup, elapsed, parallelism, mnist.source, use_cpu, device, ranks_seen, result_extra))
        good = {k: v for k, v in async_reports.items() if "value" in v and v.get("every_push_counted")}
        if good:
            result_extra["async_mode"] = dict(max(good.values(), key=lambda v: v["value"]))
        elif async_reports:
            result_extra["async_mode"] = next(iter(async_reports.values()))
        if async_reports:
            result_extra["async_modes"] = async_reports

    # ---- in-graph replication: ONE client drives a tower per GPU (rank 0; the other ranks wait) ---------------------
    want_ingraph = ingraph_only or (args.mode in (None, "auto") and n > 1 and not use_cpu
                                    and os.environ.get("DTF_BENCH_INGRAPH", "1") != "0")
    if want_ingraph and (ingraph_only or within_budget()):
        wd.stage("in-graph towers")
        if distributed:
            _sync(device, distributed)
        rep = None
        if rank == 0:
            try:
                if ingraph_only:
                    rep = _measure_ingraph(n, share_gpu or torch.cuda.device_count() < n, mnist, min(warmup, bpe), steps,
                                           int(args.chunk), test_x, test_y, use_cpu)
                elif os.environ.get("DTF_BENCH_INGRAPH_INPROC") == "1":
                    # in this very process, after everything this rank has built and retired before
                    rep = _measure_ingraph(n, share_gpu or torch.cuda.device_count() < n, mnist, min(warmup, bpe),
                                           min(steps, 4 * bpe), int(args.chunk), test_x, test_y, use_cpu)
                    rep["measured_in"] = "the benchmark's rank 0 process"
                else:
                    # (default: a client process of its own -- a fault on a node this code has never run on then
                    # costs this stage, not the line rank 0 is about to print)
                    rep = _child_ingraph(n, min(steps, 4 * bpe), min(warmup, bpe), int(args.chunk),
                                         one_device=share_gpu or torch.cuda.device_count() < n)
            except Exception as e:  # noqa: BLE001
                rep = {"error": f"{type(e).__name__}: {e}"}
            result_extra["ingraph_mode"] = rep
        if distributed:
            # the other ranks wait on the host (a store key), not inside a collective that occupies their GPU
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("dtf_bench_ingraph_done", "1")
            else:
                import datetime

                store.wait(["dtf_bench_ingraph_done"], datetime.timedelta(seconds=900))

    # ---- the same graph with wider / more layers on the dense-layer kernels (one GPU; never part of the headline) -------
    if (n == 1 and not use_cpu and not ingraph_only and args.mode in (None, "auto", "single")
            and os.environ.get("DTF_BENCH_DENSE", "1") != "0" and within_budget()):
        wd.stage("dense stacks")
        try:
            result_extra["dense_stacks"] = _measure_dense_stacks(device)
        except Exception as e:  # noqa: BLE001 — an extra measurement must never cost the headline one
            result_extra["dense_stacks"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- data-parallel steps of a wide stack: megabytes of gradients per step, bucketed all-reduce under the backward ----
    want_dense_dp = (distributed and not ingraph_only and args.mode in (None, "auto")
                     and os.environ.get("DTF_BENCH_DENSE", "1") != "0")
    if want_dense_dp and within_budget():
        if rank == 0 and best is not None:      # what is printed should this stage hang in a collective
            wd.hold(_result(n, steps, warmup, elapsed, parallelism, mnist.source, use_cpu, device, ranks_seen, result_extra))
        dp_limit = float(os.environ.get("DTF_BENCH_DENSE_DP_S", "75"))
        wd.stage("dense stack, data parallel", limit_s=dp_limit + 30.0)
        try:
            if use_cpu:         # (CPU rehearsal of the flow: in this process, over the gloo group at hand)
                from .parallel.dense_dp_probe import measure as _measure_dense_dp

                rep = _measure_dense_dp(device, rank, world, agree)
            else:
                _sync(device, distributed)
                rep = _child_dense_dp(rank, dev_index, world, "gloo" if share_gpu else "nccl", dp_limit)
                _sync(device, distributed)              # every rank's child is gone before anything else starts
        except Exception as e:  # noqa: BLE001 — an extra measurement must never cost the headline one
            rep = {"error": f"{type(e).__name__}: {e}"}
        result_extra["dense_dp"] = rep

    # ---- the reference's literal epoch: its script, one sess.run + one summary record per batch ------------------------
    if (n == 1 and not ingraph_only and args.mode in (None, "auto", "single")
            and os.environ.get("DTF_BENCH_PER_BATCH", "1") != "0" and within_budget()):
        wd.stage("per-batch tier (tfsingle.py, child process)", limit_s=120.0)
        try:
            rep = _child_per_batch_epoch(use_cpu)
        except Exception as e:  # noqa: BLE001 — an extra measurement must never cost the headline one
            rep = {"error": f"{type(e).__name__}: {e}"}
        result_extra["per_batch_tier"] = rep
        if "sec_per_epoch" in rep:
            result_extra["sec_per_epoch_per_batch_tier"] = rep["sec_per_epoch"]

    # ---- the reference's PS / worker recipes in their literal form: the scripts as separate tasks, one sess.run per batch
    if (n == 1 and not ingraph_only and args.mode in (None, "auto", "single")
            and os.environ.get("DTF_BENCH_CLUSTERS", "0" if use_cpu else "1") != "0" and within_budget()):
        wd.stage("per-batch PS/worker clusters (child processes)", limit_s=200.0)
        try:
            rep = _child_clusters(use_cpu)
        except Exception as e:  # noqa: BLE001 — an extra measurement must never cost the headline one
            rep = {"error": f"{type(e).__name__}: {e}"}
        result_extra["per_batch_clusters"] = rep
        for key, v in rep.items():
            if isinstance(v, dict) and "sec_per_epoch" in v:
                result_extra[f"sec_per_epoch_{key}_per_batch"] = v["sec_per_epoch"]

    result = None
    if rank == 0:
        if ingraph_only:
            rep = result_extra.get("ingraph_mode") or {}
            if "ms_per_step" not in rep:
                raise SystemExit(f"in-graph towers could not run: {rep}")
            result = _result(n, rep["steps"], rep["warmup"], rep["ms_per_step"] * rep["steps"] / 1000.0, rep["parallelism"],
                             mnist.source, use_cpu, device, ranks_seen,
                             dict({k: v for k, v in rep.items() if k not in ("value", "unit", "ms_per_step", "steps", "warmup", "parallelism")},
                                  setup_s=setup))
        else:
            result = _result(n, steps, warmup, elapsed, parallelism, mnist.source, use_cpu, device, ranks_seen, result_extra)
        result["wall_s"] = round(time.monotonic() - t_begin, 1)
    wd.stage("shutdown")
    if result is not None:
        result["stage_s"] = dict(wd.stage_s)
    wd.hold(result)
    if distributed:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    return wd.emit(result)


def _result(n, steps, warmup, elapsed, parallelism, source, use_cpu, device, ranks_seen, extra) -> dict:
    value = n * BATCH * steps / elapsed
    base = BASELINE_SINGLE if n == 1 else BASELINE_MULTI
    out = {
        "metric": "samples/sec (784-100-10 MLP, batch 100/worker, SGD) + sec/epoch",
        "value": round(value, 1),
        "unit": "samples/s",
        "n_gpus": n,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(1000.0 * elapsed / steps, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": round(value / base, 2),
        "dtype": "fp32",
        "data": f"synthetic MNIST-shaped ({source}), random-init N(0,1) weights",
        "config": {
            "model": "tfsingle/tfdist_between 2-layer FC MLP 784-100-10 sigmoid+softmax",
            "global_batch": n * BATCH,
            "seq_len": None,
            "parallelism": parallelism,
            "optimizer": "SGD lr=0.001",
            "device": "cpu" if use_cpu else torch.cuda.get_device_name(device),
        },
        "ranks_seen": ranks_seen,
        # every step of every measurement ran the hand-written fused kernels (CPU runs: the PyTorch oracle); the
        # graph executor's op-by-op fallback for unrecognised graphs is never part of this benchmark
        "generic_steps": 0,
        "multi_gpu_measured": bool(ranks_seen > 1 and not use_cpu and os.environ.get("DTF_BENCH_SHARE_GPU") != "1"),
        # what the K timed steps are, for reading ms_per_step at small K next to sec_per_epoch / steps_per_epoch
        "timed_region": ("run_steps(%d) between two synchronisations: hipGraph replays launched from an idle device "
                         "(launch and wake-up latency of about 20 us included)" % steps) if not use_cpu
                        else "run_steps(%d)" % steps,
    }
    out.update(extra)
    return out


def _measure_dense_stacks(device) -> dict:
    """Training steps of fully connected stacks other than the reference's network (models/dense.py on csrc/gemm.hip):
    fp32, synthetic data, hipGraph chunks; TFLOP/s counts the step's GEMMs (forward, input and weight gradients)."""
    from .models.dense import DenseStackEngine

    out = {"note": "not the reference's workload: the same graph with wider layers, every launch a hand-written kernel"}
    for dims, batch, steps in (((784, 1024, 1024, 10), 100, 160), ((784, 4096, 4096, 10), 4096, 24)):
        gen = torch.Generator().manual_seed(0)
        Ws = [(torch.randn(dims[i], dims[i + 1], generator=gen) / dims[i] ** 0.5).to(device) for i in range(len(dims) - 1)]
        bs = [torch.zeros(dims[i + 1], device=device) for i in range(len(dims) - 1)]
        nb = 8
        x = torch.rand(nb * batch, dims[0], generator=gen).to(device)
        y = torch.eye(dims[-1])[torch.randint(0, dims[-1], (nb * batch,), generator=gen)].to(device)
        eng = DenseStackEngine(Ws, bs, ["relu"] * (len(dims) - 2), batch=batch, lr=0.01)
        eng.bind_data(x, y)
        eng.run_chunk(nb)                      # captures the chunk graph
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps // nb):
            eng.run_chunk(nb)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / (steps // nb * nb)
        flop = 6.0 * batch * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)) - 2.0 * batch * dims[0] * dims[1]
        out["-".join(map(str, dims)) + f" batch {batch}"] = {
            "ms_per_step": round(1000.0 * dt, 5), "samples_per_s": round(batch / dt, 1),
            "gemm_tflops": round(flop / dt / 1e12, 1), "launches_per_step": eng.launches_per_step, "dtype": "fp32"}
        del eng, Ws, bs, x, y
    return out


def _measure_ingraph(n, one_device: bool, mnist, warmup: int, steps: int, chunk: int, test_x, test_y, use_cpu: bool) -> dict:
    """In-graph replication measured from ONE process: N towers (one per GPU; all on GPU 0 in the one-GPU
    rehearsal, limited to 2 towers there: the streams of one process share 4 hardware queues per GPU, and a
    third tower's spinning backward launch can be queued behind the launch it waits for -- or to 4 towers when the
    process was started with GPU_MAX_HW_QUEUES >= 16, as the benchmark's in-graph client is; six towers' spinning
    backward workgroups no longer fit the chip together and time out)."""
    from .parallel.towers import InGraphEngine

    if use_cpu:
        raise RuntimeError("the in-graph engine is a GPU path")
    towers = n if not one_device else min(n, 4 if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 16 else 2)
    devices = [torch.device("cuda", 0 if one_device else i) for i in range(towers)]
    out = {}
    for exchange in ("inbox", "rccl"):
        if exchange == "rccl" and one_device:
            continue
        eng = InGraphEngine(devices, mnist.train.images, mnist.train.labels, batch=BATCH, lr=LR, seed=1, chunk=chunk,
                            exchange=exchange)
        try:
            eng.prepare((warmup, steps))
            eng.run_steps(warmup)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.run_steps(steps)
            eng.synchronize()
            dt = time.perf_counter() - t0
            loss, acc = eng.evaluate(test_x.to(devices[0]), test_y.to(devices[0]))
            out[exchange] = {"value": round(towers * BATCH * steps / dt, 1), "unit": "samples/s",
                             "ms_per_step": round(1000.0 * dt / steps, 5), "steps": steps, "warmup": warmup,
                             "towers": towers, "replicas_identical": eng.replicas_identical(), "healthy": eng.healthy(),
                             "global_step": eng.global_step_value(), "host_calls_per_tower_per_chunk": eng.host_calls_per_chunk(),
                             "test_accuracy_after_run": round(acc, 4), "test_loss_after_run": round(loss, 4),
                             "parallelism": eng.describe()}
        except Exception as e:  # noqa: BLE001
            out[exchange] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            eng.close()
    good = {k: v for k, v in out.items() if "value" in v and v["replicas_identical"] and v["healthy"]}
    if not good:
        return {"error": "no in-graph exchange ran", "by_exchange": out}
    best = dict(max(good.values(), key=lambda v: v["value"]))
    best["by_exchange_ms_per_step"] = {k: v.get("ms_per_step", v.get("error")) for k, v in out.items()}
    return best
