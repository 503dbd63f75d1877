"""The one-kernel gradient exchange over peer-mapped HBM, several replica processes sharing the GPU.

On one device the peers' slabs are reached through HIP-IPC mappings of the same HBM, which exercises
the handle exchange, the pointer table, the announce / wait protocol, the slot alternation and the
rank-ordered sum; the xGMI hop itself needs a multi-GPU node.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_replicas(world, steps, use_graph, uncached=0, timeout=150, mode="sync"):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_xgmi_worker.py"), str(r), str(world), str(port),
                               str(steps), str(int(use_graph)), str(int(uncached)), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    results = []
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
        results.append(json.loads([line for line in out.splitlines() if line.startswith("{")][-1]))
    return results


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,use_graph,uncached", [(2, False, 0), (2, True, 0), (4, True, 0), (3, True, 1)])
def test_replicas_stay_bit_identical_and_track_the_mean_gradient_reference(world, use_graph, uncached):
    steps = 7
    for r in _run_replicas(world, steps, use_graph, uncached):
        assert r["healthy"], r
        assert r["identical"] and r["same_sum"], r
        assert r["global_step"] == steps
        assert r["moved"] > 1e-3                                   # the run did train
        assert r["max_err"] <= 2e-5 * max(1.0, r["scale"]), r      # fp32 reference, different summation order
        assert "peer-mapped" in r["describe"]


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,use_graph", [(2, False), (3, True)])
def test_async_workers_update_shared_shards_one_sidedly(world, use_graph):
    steps = 8
    res = _run_replicas(world, steps, use_graph, mode="async")
    for r in res:
        assert r["finite"] and r["identical"], r              # after the last push everybody pulls the same shards
        assert r["global_step"] == world * steps, r           # every push counted, atomically
        assert r["moved"] > 1e-3 and r["loss_after"] < r["loss_before"], r
        assert "one-sided" in r["describe"]


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,use_graph", [(1, True), (2, False), (2, True), (3, True)])
def test_exchange_inside_the_backward_launch_gives_the_same_training(world, use_graph):
    """Two launches per step: every backward thread swaps its element with the same thread of the peers.
    (More than three replica processes do not fit on ONE GPU together: their spinning workgroups would wait for
    workgroups of a peer that cannot become resident.  One process per GPU has no such limit.)"""
    steps = 7
    for r in _run_replicas(world, steps, use_graph, uncached=1, mode="sync2"):
        assert r["healthy"], r
        assert r["identical"] and r["same_sum"], r
        assert r["global_step"] == steps
        assert r["moved"] > 1e-3
        assert r["max_err"] <= 2e-5 * max(1.0, r["scale"]), r
        assert "inside the backward launch" in r["describe"]
