"""Multi-process integration: the README experiment matrix on a fake localhost cluster, few steps each.

scenario 1: 1 PS + 1 worker async      scenario 2/3: 1 PS + 2 workers async
scenario 4: 1 PS + 2 workers sync      scenario 5: 2 PS + 2 workers async     scenario 6: 2 PS + 2 workers sync
"""
import re

import pytest

from cluster_utils import launch_script

ARGS = ["--epochs=1", "--train_size=1100", "--batch_size=100", "--freq=5", "--lr=0.01", "--nosummaries"]
STEP_RE = re.compile(r"Step: (\d+),  Epoch: +(\d+),  Batch: +(\d+) of +(\d+),  Cost: ([\d.]+),  AvgTime: [\d.]+ms")


def _check_worker(out):
    assert "worker setting up ..." in out and "Ready to go" in out and out.rstrip().endswith("Done"), out[-2000:]
    lines = STEP_RE.findall(out)
    assert len(lines) == 3                                   # 11 batches, freq 5 -> batches 5, 10, 11
    assert [int(l[2]) for l in lines] == [5, 10, 11] and all(int(l[3]) == 11 for l in lines)
    assert re.search(r"Test-Accuracy: \d\.\d\d\n", out) and re.search(r"Total Time: [\d.]+s\n", out)
    assert re.search(r"Final Cost: [\d.]+\n", out)
    return [int(l[0]) for l in lines], [float(l[4]) for l in lines]


@pytest.mark.slow
@pytest.mark.parametrize("script,n_ps,n_w", [
    ("tfdist_between.py", 1, 1),
    ("tfdist_between.py", 1, 2),
    ("tfdist_between_sync.py", 1, 2),
    ("tfdist_between.py", 2, 2),
    ("tfdist_between_sync.py", 2, 2),
])
def test_readme_scenario(script, n_ps, n_w, tmp_path):
    res = launch_script(script, n_ps, n_w, ARGS + [f"--logs_path={tmp_path}"])
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    for i in range(n_ps):
        assert "ps setting up ..." in res[f"ps{i}"][1]       # and the PS left join() by itself
    finals = []
    for i in range(n_w):
        steps, costs = _check_worker(res[f"worker{i}"][1])
        finals.append(steps[-1])
    if "sync" in script:
        assert finals == [11] * n_w                         # one update per N worker batches
    else:
        # every push is an update; "Step" is the value pulled before a worker's own update, plus one
        # (tfdist_between.py:102), so workers whose last batches overlap print the same number
        assert 11 * n_w - (n_w - 1) <= max(finals) <= 11 * n_w
    if n_ps == 2:
        out = res["worker0"][1]                             # log_device_placement shows the round-robin shards
        assert "weights/Variable: (VariableV2): /job:ps/task:1" in out
        assert "weights/Variable_1: (VariableV2): /job:ps/task:0" in out


@pytest.mark.slow
def test_sync_script_with_a_backup_worker(tmp_path):
    """tfdist_between_sync.py --replicas_to_aggregate=2 with three workers: SyncReplicasOptimizer with fewer gradients
    per update than replicas (the reference passes equal values, tfdist_between_sync.py:66-67): an update takes the
    first two gradients of a step, the third is dropped as stale, every worker gets a token -- one update per round."""
    res = launch_script("tfdist_between_sync.py", 1, 3, ARGS + ["--replicas_to_aggregate=2", f"--logs_path={tmp_path}"])
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    finals = [_check_worker(res[f"worker{i}"][1])[0][-1] for i in range(3)]
    # 11 rounds of three workers; a worker whose last gradient was the dropped one may print one step more or less
    assert all(10 <= f <= 12 for f in finals), finals


@pytest.mark.slow
def test_async_workers_on_the_device_resident_tier(tmp_path):
    """--resident: no per-batch feed; pull, gradient kernels and push per step against the PS shards."""
    res = launch_script("tfdist_between.py", 2, 2, ARGS[:-1] + ["--resident", f"--logs_path={tmp_path}"])
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    finals = []
    for i in range(2):
        out = res[f"worker{i}"][1]
        assert "between-graph async worker, device-resident" in out
        steps, costs = _check_worker(out)
        finals.append(steps[-1])
    assert 21 <= max(finals) <= 22     # 2 workers x 11 pushes, every one an update (the last two may overlap)
    import os
    assert any(f.startswith("events.out.tfevents.") for f in os.listdir(tmp_path))   # per-step summaries kept


@pytest.mark.slow
def test_single_process_script_both_tiers(tmp_path):
    import os
    import subprocess
    import sys

    from cluster_utils import ROOT

    for extra in ([], ["--resident"]):
        p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tfsingle.py"), *ARGS[:-1], f"--logs_path={tmp_path}",
                            *extra], cwd=ROOT, capture_output=True, text=True, timeout=240)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        steps, costs = _check_worker(p.stdout)
        assert steps == [5, 10, 11]
    assert any(f.startswith("events.out.tfevents.") for f in os.listdir(tmp_path))
