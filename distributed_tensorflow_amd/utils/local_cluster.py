"""A whole localhost cluster of one of the between-graph scripts, every task a process of its own (PS first), exactly
like launching them by hand (README.md:88-113 of the reference: one terminal per ``--job_name=... --task_index=...``).

The cluster addresses travel through DTF_PS_SVRS / DTF_WORKER_SVRS (``settings.py`` reads them); every worker leaves its
epoch times as floats in a file (DTF_EPOCH_TIMES_FILE; stdout rounds them to 10 ms).  Used by ``bench/run_cluster.py``
and by ``bench.py``'s per-batch cluster stage.
"""
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time
from typing import Dict, List, Optional, Sequence

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def free_ports(n: int) -> List[int]:
    socks, ports = [], []
    for _ in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        socks.append(s)
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    return ports


def run_local_cluster(script: str, n_ps: int, n_workers: int, transport: str = "tcp", epochs: int = 3,
                      extra: Sequence[str] = (), timeout: float = 600.0, env: Optional[Dict[str, str]] = None) -> dict:
    """Runs ``script`` as ``n_ps`` parameter-server tasks and ``n_workers`` worker tasks until the workers are done (or
    ``timeout`` seconds have passed: what is left is killed).  Returns ``{"script", "ps", "workers", "transport",
    "epochs", "returncode", "tasks": {"worker0": {"returncode", "epoch_times", "first_epoch_s", "steady_epoch_s" (mean
    of the epochs after the first), "final_accuracy", "last_step"}, ...}}``; a task that failed carries the tail of
    its output."""
    ports = free_ports(n_ps + n_workers)
    base = dict(os.environ if env is None else env,
                DTF_PS_SVRS=",".join(f"localhost:{p}" for p in ports[:n_ps]),
                DTF_WORKER_SVRS=",".join(f"localhost:{p}" for p in ports[n_ps:]),
                PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    result = {"script": script, "ps": n_ps, "workers": n_workers, "transport": transport, "epochs": epochs, "tasks": {}}
    with tempfile.TemporaryDirectory(prefix="dtf_cluster_") as tmp:
        common = [f"--epochs={epochs}", f"--transport={transport}", f"--logs_path={os.path.join(tmp, 'logs')}", *extra]
        procs = []
        for job, n in (("ps", n_ps), ("worker", n_workers)):
            for i in range(n):
                cmd = [sys.executable, "-u", os.path.join(ROOT, script), f"--job_name={job}", f"--task_index={i}", *common]
                task_env = dict(base)
                if job == "worker":
                    task_env["DTF_EPOCH_TIMES_FILE"] = os.path.join(tmp, f"worker{i}.json")
                procs.append((f"{job}{i}", subprocess.Popen(cmd, cwd=ROOT, env=task_env, stdout=subprocess.PIPE,
                                                            stderr=subprocess.STDOUT, text=True)))
        t0 = time.time()
        rc = 0
        for name, p in procs:
            try:
                out, _ = p.communicate(timeout=max(1.0, timeout - (time.time() - t0)))
            except subprocess.TimeoutExpired:
                p.kill()
                out, _ = p.communicate()
                out += "\n<<TIMEOUT>>"
            rc = rc or p.returncode
            if name.startswith("worker"):
                times = [float(x) for x in re.findall(r"Total Time: ([\d.]+)s", out)]
                try:
                    with open(os.path.join(tmp, name + ".json")) as f:
                        exact = [float(t) for t in json.load(f)["epoch_times"]]
                    times = exact if len(exact) == len(times) else times
                except (OSError, ValueError, KeyError):
                    pass
                accs = [float(x) for x in re.findall(r"Test-Accuracy: ([\d.]+)", out)]
                steps = [int(x) for x in re.findall(r"Step: (\d+),", out)]
                result["tasks"][name] = {
                    "returncode": p.returncode,
                    "epoch_times": [round(t, 5) for t in times],
                    "first_epoch_s": times[0] if times else None,
                    "steady_epoch_s": round(sum(times[1:]) / len(times[1:]), 4) if len(times) > 1 else None,
                    "final_accuracy": accs[-1] if accs else None,
                    "last_step": steps[-1] if steps else None,
                }
                if p.returncode != 0:
                    result["tasks"][name]["tail"] = out[-1500:]
            elif p.returncode != 0:
                result["tasks"][name] = {"returncode": p.returncode, "tail": out[-1500:]}
    result["returncode"] = rc
    return result
