"""Collective (torch.distributed) replica trainers with gloo, world_size 2, on CPU."""
import json
import os
import subprocess
import sys

import pytest

from cluster_utils import ROOT, free_ports


def _run(mode, world=2, device="cpu", graph="eager"):
    port = free_ports(1)[0]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_worker.py"), mode, device, graph]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.slow
def test_sync_shards_and_towers_agree_and_replicas_stay_identical():
    sync = _run("sync")
    towers = _run("towers")
    assert sync["replicas_identical"] and towers["replicas_identical"]
    assert sync["global_step"] == 13 and towers["global_step"] == 13        # one update per N pushes
    # reduce-scatter + sharded apply + all-gather == all-reduce + replicated apply (mean gradient)
    assert abs(sync["checksum"] - towers["checksum"]) < 1e-3 * max(1.0, abs(sync["checksum"]))
    assert abs(sync["loss"] - towers["loss"]) < 1e-4


@pytest.mark.slow
def test_async_rule_counts_every_worker_gradient():
    res = _run("async_lockstep")
    assert res["replicas_identical"] and res["global_step"] == 26           # N updates per round


def test_async_send_recv_parameter_servers_count_every_push():
    """Two-sided asynchronous PS (gloo here, RCCL on GPUs): one communicator + service lane per (worker, shard)
    pair; every worker gradient is one update of every shard, global_step == sum of the workers' pushes."""
    res = _run("async_rccl")
    assert res["global_step"] == 26 and res["served"] == 26     # 13 own + 13 served for the peer, on every rank
    assert res["replicas_identical"]                            # after the final refresh from the shards
    assert "send/recv" in res["desc"] and res["loss"] == res["loss"]


def test_in_graph_towers_mean_gradient_and_identical_replicas():
    import numpy as np
    import torch

    from distributed_tensorflow_amd.ops import reference as ref
    from distributed_tensorflow_amd.parallel.towers import InGraphTowers

    rng = np.random.RandomState(0)
    images = rng.rand(400, 784).astype(np.float32)
    labels = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 400)]
    tw = InGraphTowers(["cpu", "cpu"], images, labels, batch=100, lr=0.05, seed=1)
    p = {k: v.clone() for k, v in tw.layout.mlp_views(tw.params[0]).items()}
    batches = [(ds.x_epoch[:100].clone(), ds.y_epoch[:100].clone()) for ds in tw.datasets]
    tw.step()
    assert tw.replicas_identical() and tw.global_step_value() == 1
    g = {k: torch.zeros_like(v) for k, v in p.items()}
    for xb, yb in batches:
        out = ref.mlp_grads(xb, yb, p["W1"], p["b1"], p["W2"], p["b2"])
        for k in g:
            g[k] += out["d" + k] / 2
    got = tw.layout.mlp_views(tw.params[1])
    for k in p:
        torch.testing.assert_close(got[k], p[k] - 0.05 * g[k], rtol=1e-5, atol=1e-6)
    tw.run_steps(5)                       # crosses the 4-batch epoch boundary (reshuffle)
    assert tw.replicas_identical() and tw.global_step_value() == 6


def test_window_fetch_on_the_cpu_path_returns_the_same_values_as_the_blocking_calls():
    import numpy as np
    import torch
    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(330, 784, generator=g), torch.eye(10)[torch.randint(0, 10, (330,), generator=g)]
    ds = DeviceDataset(x, y, "cpu", 100, seed=1)
    tr = build_replica_trainer(mode="single", device=torch.device("cpu"), dataset=ds, batch=100, lr=0.01, seed=2, chunk=2)
    first = tr.engine._host_cursor
    tr.run_steps(3)
    cost, losses, accs, gstep = tr.fetch_window_async(first, 3, want_history=True).result()
    want_l, want_a = tr.engine.history(first, 3)
    assert cost == tr.last_loss() and gstep == 3.0
    assert np.array_equal(losses.numpy(), want_l.numpy()) and np.array_equal(accs.numpy(), want_a.numpy())
