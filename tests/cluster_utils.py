"""Helpers for fake clusters: every task on localhost with its own free port (README.md:29-30 technique)."""
import os
import socket
import subprocess
import sys

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


def cluster_dict(n_ps, n_workers):
    ports = free_ports(n_ps + n_workers)
    return {"ps": [f"localhost:{p}" for p in ports[:n_ps]], "worker": [f"localhost:{p}" for p in ports[n_ps:]]}


def launch_script(script, n_ps, n_workers, extra_args=(), timeout=240, env_extra=None, attempts=3):
    """Start every task of a cluster as its own process; returns {task: (returncode, output)}.

    Ports are probed and released before the tasks bind them, so another process can grab one in
    between; a run that failed to bind or to connect is repeated on fresh ports."""
    for attempt in range(attempts):
        out = _launch_once(script, n_ps, n_workers, extra_args, timeout, env_extra)
        text = "".join(o for _, o in out.values())
        if all(rc == 0 for rc, _ in out.values()) or attempt == attempts - 1:
            return out
        if "cannot bind task server" not in text and "cannot reach task server" not in text:
            return out
    return out


def _launch_once(script, n_ps, n_workers, extra_args, timeout, env_extra):
    c = cluster_dict(n_ps, n_workers)
    env = dict(os.environ)
    env.update({"DTF_PS_SVRS": ",".join(c["ps"]), "DTF_WORKER_SVRS": ",".join(c["worker"]),
                "OMP_NUM_THREADS": "2", "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    env.update(env_extra or {})
    procs = {}
    for job, n in (("ps", n_ps), ("worker", n_workers)):
        for i in range(n):
            cmd = [sys.executable, "-u", os.path.join(ROOT, script), f"--job_name={job}", f"--task_index={i}",
                   *extra_args]
            procs[f"{job}{i}"] = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                                  stderr=subprocess.STDOUT, text=True)
    out = {}
    try:
        for name, p in procs.items():
            try:
                o, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
                o += "\n<<TIMEOUT>>"
            out[name] = (p.returncode, o)
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
    return out
