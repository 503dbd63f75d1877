"""The reference's own scripts, unmodified, on this framework (skipped when the reference tree is absent).

Only a module-level constant (`epochs`) is overridden through the runner's --dtf-set and the cluster addresses
come from DTF_PS_SVRS / DTF_WORKER_SVRS; the script text itself is executed as it is.
"""
import os
import re
import subprocess
import sys

import pytest

from cluster_utils import ROOT, cluster_dict

REF = os.environ.get("DTF_REFERENCE_DIR", "/root/reference")
pytestmark = [pytest.mark.slow,
              pytest.mark.skipif(not os.path.exists(os.path.join(REF, "tfsingle.py")), reason="reference tree not mounted")]

STEP_RE = re.compile(r"Step: (\d+),  Epoch: +1,  Batch: +(\d+) of +(\d+),  Cost: ([\d.]+),  AvgTime: [\d.]+ms")


def _env(extra=None):
    env = dict(os.environ, DTF_TRAIN_SIZE="2200", OMP_NUM_THREADS="2",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.update(extra or {})
    return env


def _run(script, args=(), env=None, cwd=None, sets=()):
    cmd = [sys.executable, "-u", "-m", "distributed_tensorflow_amd.compat.run", os.path.join(REF, script),
           "--dtf-set", "epochs=1", *[a for s in sets for a in ("--dtf-set", s)], *args]
    return subprocess.Popen(cmd, cwd=cwd, env=env or _env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_reference_tfsingle_runs_unmodified(tmp_path):
    p = _run("tfsingle.py", cwd=str(tmp_path))
    out, _ = p.communicate(timeout=240)
    assert p.returncode == 0, out[-3000:]
    steps = STEP_RE.findall(out)
    assert [int(s[0]) for s in steps] == [22] and "Test-Accuracy:" in out and out.rstrip().endswith("Done")
    assert any(f.startswith("events.out.tfevents") for f in os.listdir(tmp_path / "logs"))


@pytest.mark.parametrize("script,sync", [("tfdist_between.py", False), ("tfdist_between_sync.py", True),
                                         ("tfdist.py", False)])
def test_reference_between_graph_scripts_run_unmodified(script, sync, tmp_path):
    """``tfdist.py`` is the reference's first between-graph script: the cluster addresses are module constants
    (tfdist.py:8-9: two machines of its authors) instead of ``settings.py``, the server gets no ConfigProto and the
    worker device names no GPU (tfdist.py:16,33); only those two address lists are replaced."""
    c = cluster_dict(1, 2)
    env = _env({"DTF_PS_SVRS": ",".join(c["ps"]), "DTF_WORKER_SVRS": ",".join(c["worker"])})
    sets = (f"ps_svrs={c['ps']!r}", f"worker_svrs={c['worker']!r}") if script == "tfdist.py" else ()
    procs = {"ps0": _run(script, ["--job_name=ps", "--task_index=0"], env, str(tmp_path), sets)}
    for i in range(2):
        procs[f"w{i}"] = _run(script, ["--job_name=worker", f"--task_index={i}"], env, str(tmp_path), sets)
    outs = {}
    try:
        for k, p in procs.items():
            outs[k], _ = p.communicate(timeout=240)
            assert p.returncode == 0, f"{k}:\n{outs[k][-3000:]}"
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
    assert "ps setting up ..." in outs["ps0"]
    finals = []
    for i in range(2):
        steps = STEP_RE.findall(outs[f"w{i}"])
        assert steps and outs[f"w{i}"].rstrip().endswith("Done")
        finals.append(int(steps[-1][0]))
    # 'Step' is the global step pulled BEFORE the worker's own update, plus one (tfdist_between.py:102): workers
    # whose last batches overlap pull the same value, so up to n_w - 1 updates may be missing from the largest print
    if sync:
        assert finals == [22, 22]
    else:
        assert 43 <= max(finals) <= 44
