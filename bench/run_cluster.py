#!/usr/bin/env python
"""Launch a whole localhost cluster of one of the between-graph scripts and report sec/epoch per worker.

    python bench/run_cluster.py --script tfdist_between.py --ps 1 --workers 2 --transport ipc --epochs 3

Every task is its own process (PS first), exactly like launching them by hand; the cluster addresses are
passed through DTF_PS_SVRS / DTF_WORKER_SVRS.  Prints one JSON line with the steady-state epoch time
(mean of the epochs after the first), the first-epoch time and the final accuracy of every worker.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_ports(n):
    socks, ports = [], []
    for _ in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        socks.append(s)
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    return ports


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--script", default="tfdist_between.py")
    ap.add_argument("--ps", type=int, default=1)
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--transport", default="tcp")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    ports = free_ports(a.ps + a.workers)
    env = dict(os.environ,
               DTF_PS_SVRS=",".join(f"localhost:{p}" for p in ports[:a.ps]),
               DTF_WORKER_SVRS=",".join(f"localhost:{p}" for p in ports[a.ps:]),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    common = [f"--epochs={a.epochs}", f"--transport={a.transport}", "--logs_path=/tmp/dtf_bench_logs", *a.extra]
    procs = []
    import tempfile

    times_dir = tempfile.mkdtemp(prefix="dtf_epoch_times_")
    for job, n in (("ps", a.ps), ("worker", a.workers)):
        for i in range(n):
            cmd = [sys.executable, "-u", os.path.join(ROOT, a.script), f"--job_name={job}", f"--task_index={i}", *common]
            if job == "worker":       # the loop's own epoch times as floats (stdout rounds them to 10 ms)
                env = dict(env, DTF_EPOCH_TIMES_FILE=os.path.join(times_dir, f"worker{i}.json"))
            procs.append((f"{job}{i}", subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                                        stderr=subprocess.STDOUT, text=True)))
    t0 = time.time()
    result = {"script": a.script, "ps": a.ps, "workers": a.workers, "transport": a.transport, "epochs": a.epochs,
              "tasks": {}}
    rc = 0
    for name, p in procs:
        try:
            out, _ = p.communicate(timeout=max(1, a.timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n<<TIMEOUT>>"
        rc = rc or p.returncode
        if name.startswith("worker"):
            times = [float(x) for x in re.findall(r"Total Time: ([\d.]+)s", out)]
            try:
                with open(os.path.join(times_dir, name + ".json")) as f:
                    exact = [float(t) for t in json.load(f)["epoch_times"]]
                times = exact if len(exact) == len(times) else times
            except (OSError, ValueError, KeyError):
                pass
            accs = [float(x) for x in re.findall(r"Test-Accuracy: ([\d.]+)", out)]
            steps = [int(x) for x in re.findall(r"Step: (\d+),", out)]
            result["tasks"][name] = {
                "returncode": p.returncode,
                "first_epoch_s": times[0] if times else None,
                "steady_epoch_s": round(sum(times[1:]) / len(times[1:]), 4) if len(times) > 1 else None,
                "final_accuracy": accs[-1] if accs else None,
                "last_step": steps[-1] if steps else None,
            }
            if p.returncode != 0:
                result["tasks"][name]["tail"] = out[-1500:]
        elif p.returncode != 0:
            result["tasks"][name] = {"returncode": p.returncode, "tail": out[-1500:]}
    print(json.dumps(result))
    return rc


if __name__ == "__main__":
    sys.exit(main())
