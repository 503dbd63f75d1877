"""bench.py prints one JSON line with the agreed fields, for one process and under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

from cluster_utils import free_ports

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
          "vs_baseline", "dtype", "data", "config"}


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


def test_single_process_defaults_to_one_gpu_and_reports_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu", "--steps", "30", "--warmup", "5"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _json_line(p.stdout)
    assert FIELDS <= set(r)
    assert r["n_gpus"] == 1 and r["steps"] == 30 and r["warmup"] == 5 and r["higher_is_better"] is True
    assert r["config"]["global_batch"] == 100 and r["scaling"] == "weak" and r["dtype"] == "fp32"
    assert r["value"] == pytest.approx(100 * 30 / (r["ms_per_step"] * 30 / 1000.0), rel=1e-3)
    assert r["global_step"] == 35                     # every timed and warm-up step applied its update
    assert r["ranks_seen"] == 1 and r["generic_steps"] == 0 and r["multi_gpu_measured"] is False
    # the reference's "Total Time" epoch is measured whatever --steps says (55 batches in the CPU-sized set)
    assert r["sec_per_epoch"] > 0 and r["steps_per_epoch"] == 55 and "upload_and_first_shuffle_s" in r["setup_s"]
    # wall seconds of every stage, so that a slow run can be read from its record
    assert {"data set", "measuring single", "evaluation + like-for-like epoch"} <= set(r["stage_s"])
    assert all(v >= 0 for v in r["stage_s"].values()) and sum(r["stage_s"].values()) <= r["wall_s"] + 1.0
    # the reference's literal epoch: the repository's tfsingle.py, one sess.run + one event record per batch
    pb = r["per_batch_tier"]
    assert r["sec_per_epoch_per_batch_tier"] == pb["sec_per_epoch"] > 0 and pb["steps_per_epoch"] == 55
    assert pb["epochs_run"] == 3 and pb["first_epoch_s"] > 0


def test_per_batch_cluster_stage_reports_the_reference_recipes():
    """DTF_BENCH_CLUSTERS=1 (the default on a GPU): the between-graph scripts as separate tasks, one sess.run per batch --
    asynchronous 1 ps + 1 worker, 1 ps + 2 workers and 2 ps + 2 workers, synchronous 1 ps + 2 workers -- next to the
    headline."""
    env = dict(os.environ, DTF_BENCH_CLUSTERS="1", DTF_BENCH_PER_BATCH="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu", "--steps", "10", "--warmup", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _json_line(p.stdout)
    c = r["per_batch_clusters"]
    for key, workers, steps in (("async_1ps_1w", 1, 4 * 11), ("async_1ps_2w", 2, 2 * 4 * 11), ("async_2ps_2w", 2, 2 * 4 * 11),
                                ("sync_1ps_2w", 2, 4 * 11)):
        assert "sec_per_epoch" in c[key], c[key]
        assert r[f"sec_per_epoch_{key}_per_batch"] == c[key]["sec_per_epoch"] > 0 and c[key]["workers"] == workers
        # asynchronous workers print the step they pulled before their own update: up to workers - 1 short of the total
        assert steps - (workers - 1) <= max(c[key]["last_step"]) <= steps
    assert "per-batch PS/worker clusters (child processes)" in r["stage_s"]


@pytest.mark.parametrize("mode", [None, "async"])
def test_two_ranks_over_gloo_report_the_aggregate(mode):
    port = free_ports(1)[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu",
           "--steps", "20", "--warmup", "4"] + (["--mode", mode] if mode else [])
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = _json_line(p.stdout)
    assert FIELDS <= set(r) and r["n_gpus"] == 2 and r["config"]["global_batch"] == 200
    assert r["value"] == pytest.approx(2 * 100 * 20 / (r["ms_per_step"] * 20 / 1000.0), rel=1e-3)
    assert r["ranks_seen"] == 2 and r["sec_per_epoch"] > 0 and "rendezvous" in r["stage_s"]
    if mode is None:      # auto: both synchronous exchanges ran the same measurement and one was kept
        assert set(r["mode_ms_per_step"]) == {"sync", "towers"}
        assert r["ms_per_step"] == min(r["mode_ms_per_step"].values())
        assert r["global_step"] == 24                 # one update per round of two worker batches
        a = r["async_mode"]                           # and the asynchronous mode next to it, on the same data set
        assert a["global_step"] == 2 * (a["steps"] + a["warmup"]) and a["every_push_counted"] is True
        assert "send/recv" in a["parallelism"] and a["value"] > 0
        d = r["dense_dp"]                             # a stack with bucketed gradient all-reduces, same two ranks
        assert d["replicas_identical"] is True and d["ranks"] == 2 and d["global_step"] == d["steps"] + d["warmup"]
        assert d["with_collectives"]["samples_per_s"] > 0 and d["compute_only"]["ms_per_step"] > 0
        assert d["allreduce_alone"]["bus_GB_per_s"] > 0 and d["allreduce_alone"]["backend"] == "gloo"
    else:                 # asynchronous parameter servers: every worker's gradient is its own update
        assert r["global_step"] == 2 * 24 and "send/recv" in r["config"]["parallelism"]


def test_plain_invocation_with_several_gpus_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torch.distributed.run): the launcher creates the rank processes itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu", "--steps", "20",
                        "--warmup", "4"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = _json_line(p.stdout)
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["config"]["global_batch"] == 200
    assert r["sec_per_epoch"] > 0 and r["async_mode"]["value"] > 0 and r["value"] > 0


def test_gpus_flag_must_match_the_world_size():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu", "--steps", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "one process per GPU" in (p.stderr + p.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_two_ranks_sharing_the_gpu_rehearse_the_peer_exchange_flow():
    """Trial in child processes -> probe of the candidates -> timed run -> replica agreement check.
    RCCL refuses two ranks on one device, so the rehearsal's rendezvous is gloo (DTF_BENCH_SHARE_GPU=1)."""
    port = free_ports(1)[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "600", "--warmup", "100"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=380,
                       env=dict(os.environ, DTF_BENCH_SHARE_GPU="1"))
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    doc = json.loads([line for line in p.stdout.splitlines() if line.startswith("{")][-1])
    assert doc["n_gpus"] == 2 and doc["config"]["global_batch"] == 200
    assert doc["peer_exchange_trial"]["ok"] is True and doc["peer_exchange_trial"]["identical"] is True
    # of the four peer-mapped exchanges that passed the trial, the two fastest by the trial's timings are measured
    peer = {"xgmi", "xgmi2", "xgmi2c", "xgmi2p"}
    assert set(doc["mode_ms_per_step"]) < peer and len(doc["mode_ms_per_step"]) == 2 and doc["ranks_seen"] == 2
    pruned = peer - set(doc["mode_ms_per_step"])
    assert all("not measured" in doc["mode_errors"][m] for m in pruned)
    assert set(doc["mode_ms_per_step_device"]) == set(doc["mode_ms_per_step"]) and doc["ms_per_step_device"] > 0
    assert doc["ms_per_step_device"] <= doc["ms_per_step"] * 1.05          # the device's clock inside the host's bracket
    assert {"peer-exchange trial (child processes)", "rendezvous", "data set", "in-graph towers"} <= set(doc["stage_s"])
    assert doc["peer_exchange_trial"]["ok2"] is True and doc["peer_exchange_trial"]["ok2c"] is True
    assert doc["peer_exchange_trial"]["ok2p"] is True
    assert "peer-mapped" in doc["config"]["parallelism"]          # whichever of the two exchanges was faster
    assert doc["global_step"] == 700 and doc["sec_per_epoch"] > 0 and doc["multi_gpu_measured"] is False
    a = doc["async_mode"]                                   # the asynchronous workers, measured next to it
    assert a["every_push_counted"] is True and a["global_step"] == 2 * (a["steps"] + a["warmup"]) and a["value"] > 0
    assert doc["peer_exchange_trial"]["async_ok"] is True
    assert set(doc["async_modes"]) == {"xgmi_async", "async_rccl"}
    assert "through peer mappings" in doc["async_modes"]["xgmi_async"]["parallelism"]
    # the main run's figure for the asynchronous workers on the device's clock is the trial's, not an artefact of the
    # bracket (two processes that share one GPU are co-scheduled more or less well from run to run: hence the factor)
    assert a["ms_per_step_device"] <= 2.5 * doc["peer_exchange_trial"]["async_ms_per_step"] + 0.005
    ig = doc["ingraph_mode"]                                # one client, towers sharing the GPU in this rehearsal
    assert ig["replicas_identical"] is True and ig["healthy"] is True and ig["host_calls_per_tower_per_chunk"] == 1
    d = doc["dense_dp"]                                     # wide stack, gradient buckets reduced on a side stream
    assert d["replicas_identical"] is True and d["global_step"] == d["steps"] + d["warmup"], d
    assert d["model"] == "784-4096-4096-10" and d["with_collectives"]["samples_per_s"] > 0
    assert d["allreduce_alone"]["ms"] > 0
    sb = d["small_batch"]                                   # launch-bound steps: eager against hipGraph chunks (RCCL only)
    assert sb["eager"]["replicas_identical"] and sb["graph_chunks"]["replicas_identical"] and sb["eager"]["ms_per_step"] > 0


@pytest.mark.parametrize("held", [True, False])
def test_watchdog_prints_the_held_result_when_a_stage_outlives_its_limit(held):
    """The safety net of the multi-rank benchmark: a stage stuck in a collective ends the process with the complete
    result that was registered before it (status 0), or with status 3 when there is none."""
    code = ("import time\n"
            "from distributed_tensorflow_amd.bench_support import Watchdog\n"
            "wd = Watchdog(0, 300.0)\n"
            + ("wd.hold({'metric': 'm', 'value': 1.5})\n" if held else "")
            + "wd.stage('optional stage', limit_s=0.5)\n"
            "time.sleep(30)\n"
            "print('not reached')\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert "not reached" not in p.stdout and "wall-time limit reached in stage 'optional stage'" in p.stderr
    assert 'File "<string>", line' in p.stderr                         # the main thread's stack is part of the diagnosis
    if held:
        assert p.returncode == 0
        got = _json_line(p.stdout)
        assert set(got.pop("stage_s")) == {"start", "optional stage"}
        assert got == {"metric": "m", "value": 1.5, "truncated_at_stage": "optional stage"}
    else:
        assert p.returncode == 3 and p.stdout.strip() == ""


def test_a_rank_that_cannot_set_up_the_dense_stage_costs_only_that_stage():
    """One rank fails before the stage's first collective: the ranks agree to skip it, everything else is reported."""
    port = free_ports(1)[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu",
           "--steps", "20", "--warmup", "4"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, DTF_BENCH_TEST_FAIL="dense_dp:1", DTF_BENCH_DENSE_DP_S="60"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = _json_line(p.stdout)
    assert r["dense_dp"] == {"error": "set-up failed on another rank"} and "truncated_at_stage" not in r
    assert r["ranks_seen"] == 2 and r["sec_per_epoch"] > 0 and r["async_mode"]["value"] > 0
