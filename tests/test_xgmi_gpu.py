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
@pytest.mark.parametrize("world,use_graph", [(3, False), (4, False), (3, True)])
def test_backup_replicas_the_first_gradients_to_arrive_count_and_every_replica_applies_their_mean(world, use_graph):
    """SyncReplicasOptimizer(replicas_to_aggregate = N - 1, total_num_replicas = N) on the one-kernel exchange
    (tfdist_between_sync.py:66-67 passes equal values, the optimizer takes fewer): every exchange sums N - 1 of the N
    gradient slabs -- the arrival order decides, one ticket word, all replicas read the same verdict -- and every replica,
    the dropped one included, applies that mean.  Eager: the members of every step are read back and the CPU oracle
    averages exactly those; captured chunks: health, bit-identity and the step count."""
    steps = 7
    results = _run_replicas(world, steps, use_graph, mode="sync_backup")
    for r in results:
        assert r["healthy"], r
        assert r["identical"] and r["same_sum"], r
        assert r["global_step"] == steps and r["moved"] > 1e-3
        assert "first %d of %d" % (world - 1, world) in r["describe"]
        if not use_graph:
            assert r["same_members"] and len(r["members"]) == steps
            assert all(len(m) == world - 1 and len(set(m)) == world - 1 for m in r["members"]), r["members"]
            assert r["max_err"] <= 2e-5 * max(1.0, r["scale"]), r


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,use_graph,mode", [(2, False, "async"), (3, True, "async"), (4, True, "async"),
                                                  (2, True, "async_sep"), (3, True, "async_auto")])
def test_async_workers_update_shared_shards_one_sidedly(world, use_graph, mode):
    """``async``: the push is part of the backward launch (float atomics per element); ``async_sep``: a launch of
    its own; ``async_auto``: every worker measures the forms of its step against the live shards (with a learning rate
    of zero and no step count) and keeps the fastest -- which one that is does not change what is computed."""
    steps = 8
    res = _run_replicas(world, steps, use_graph, mode=mode)
    for r in res:
        assert r["finite"] and r["identical"], r              # after the last push everybody pulls the same shards
        assert r["global_step"] == world * steps, r           # every push counted, atomically
        assert r["moved"] > 1e-3 and r["loss_after"] < r["loss_before"], r
        if mode == "async_auto":
            assert "measured us/step: forward from the shards" in r["describe"]
        else:
            assert ("float atomics" if mode == "async" else "one-sided lock-free SGD push") in r["describe"]


@pytest.mark.gpu
def test_one_async_worker_trains_the_same_with_either_push():
    """With a single worker nothing is concurrent, so the push inside the backward launch and the push kernel
    must produce the same variables: ``shard + (-lr * g)`` either way."""
    import torch

    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(800, 784, generator=g)
    y = torch.eye(10)[torch.randint(0, 10, (800,), generator=g)]
    out = {}
    # "fold": the forward reads the variables from the shards (2 launches); "fused": pull launch + forward + pushing
    # backward (3); "separate": pull, forward, backward into a gradient slab, push kernel (4)
    for name, kw in (("fold", dict(fused_push=True, fold_pull=True)), ("fused", dict(fused_push=True, fold_pull=False)),
                     ("separate", dict(fused_push=False))):
        ds = DeviceDataset(x, y, dev, 100, seed=5)
        t = build_replica_trainer("xgmi_async", device=dev, dataset=ds, batch=100, lr=0.05, seed=1, rank=0, world=1,
                                  chunk=4, use_graph=name != "separate", **kw)
        assert t.fold_pull == (name == "fold")
        assert ("2 launches/step" in t.describe()) == (name == "fold")
        p0 = t.params.clone()
        t.prepare((12,))
        t.run_steps(12)
        torch.cuda.synchronize()
        assert t.global_step_value() == 12
        out[name] = t.params.clone()
        assert float((out[name] - p0)[4:].abs().max()) > 1e-3
        t.close()
    for name in ("fold", "fused"):
        assert torch.allclose(out[name], out["separate"], rtol=0, atol=2e-6), (name, float((out[name] - out["separate"]).abs().max()))


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,use_graph,mode", [(1, True, "sync2"), (2, False, "sync2"), (2, True, "sync2"),
                                                  (3, True, "sync2"), (1, True, "sync2c"), (2, False, "sync2c"),
                                                  (2, True, "sync2c"), (3, True, "sync2c"), (1, True, "sync2p"),
                                                  (2, False, "sync2p"), (2, True, "sync2p"), (3, True, "sync2p")])
def test_exchange_inside_the_backward_launch_gives_the_same_training(world, use_graph, mode):
    """Two launches per step: every backward thread swaps its element with the same thread of the peers
    (``sync2``: 8-byte {number, value} words; ``sync2c``: bare 4-byte values and a "not yet written" pattern;
    ``sync2p``: two phases through the owner of every element, which returns the new parameter value).
    (More than three replica processes do not fit on ONE GPU together: their spinning workgroups would wait for
    workgroups of a peer that cannot become resident.  One process per GPU has no such limit.)"""
    steps = 7
    for r in _run_replicas(world, steps, use_graph, uncached=1, mode=mode):
        assert r["healthy"], r
        assert r["identical"] and r["same_sum"], r
        assert r["global_step"] == steps
        assert r["moved"] > 1e-3
        assert r["max_err"] <= 2e-5 * max(1.0, r["scale"]), r
        assert "inside the backward launch" in r["describe"]
        assert ("two-phase" in r["describe"]) == (mode == "sync2p")


@pytest.mark.gpu
def test_one_launch_pull_over_shards_of_different_sizes():
    """``dtf_xgmi_pull_ranges``: slab = shard 0 ++ shard 1 ++ ..., an empty shard in the middle included."""
    import ctypes as C

    import torch

    from distributed_tensorflow_amd.ops import hip_backend as hb

    dev = torch.device("cuda", 0)
    sizes = [78404, 0, 1100, 12]
    shards = [torch.randn(max(n, 4), device=dev)[:n] for n in sizes]
    starts = [0]
    for n in sizes:
        starts.append(starts[-1] + n)
    local = torch.zeros(starts[-1], device=dev)
    ptrs = (C.c_uint64 * len(sizes))(*[t.data_ptr() if t.numel() else 0 for t in shards])
    st = (C.c_int64 * len(starts))(*starts)
    rc = hb.require().dtf_xgmi_pull_ranges(ptrs, st, len(sizes), local.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(local, torch.cat(shards))
    st_bad = (C.c_int64 * len(starts))(*[s + 2 if 0 < i < len(sizes) else s for i, s in enumerate(starts)])
    assert hb.require().dtf_xgmi_pull_ranges(ptrs, st_bad, len(sizes), local.data_ptr(), None) != 0   # unaligned range


@pytest.mark.gpu
def test_pushing_backward_matches_the_fp32_oracle():
    """Three steps of one asynchronous worker (push inside the backward launch) against plain PyTorch fp32:
    p <- p - lr * grad(batch) with the gradients of ``ops/reference.py``, and global_step counting the pushes."""
    import torch

    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.ops import reference as ref
    from distributed_tensorflow_amd.parallel.replicas import build_replica_trainer

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(400, 784, generator=g)
    y = torch.eye(10)[torch.randint(0, 10, (400,), generator=g)]
    ds = DeviceDataset(x, y, dev, 100, seed=3)
    lr = 0.05
    t = build_replica_trainer("xgmi_async", device=dev, dataset=ds, batch=100, lr=lr, seed=1, rank=0, world=1,
                              chunk=4, use_graph=False, fused_push=True)
    p = {k: v.clone() for k, v in t.layout.mlp_views(t.params).items()}
    for step in range(3):
        xb, yb = ds.x_epoch[100 * step:100 * (step + 1)], ds.y_epoch[100 * step:100 * (step + 1)]
        gr = ref.mlp_grads(xb, yb, p["W1"], p["b1"], p["W2"], p["b2"])
        for k in ("W1", "b1", "W2", "b2"):
            p[k] = p[k] - lr * gr["d" + k]
    t.run_steps(3)
    torch.cuda.synchronize()
    assert t.global_step_value() == 3               # (pulls the shards into t.params)
    got = t.layout.mlp_views(t.params)
    for k in ("W1", "b1", "W2", "b2"):
        err = float((got[k] - p[k]).abs().max())
        assert err <= 2e-5 * max(1.0, float(p[k].abs().max())), (k, err)
    t.close()


@pytest.mark.gpu
def test_stream_probe_tells_streams_on_one_hardware_queue_from_streams_that_run_side_by_side():
    import torch

    from distributed_tensorflow_amd.ops import hip_backend as hb

    a = torch.cuda.Stream()
    assert hb.streams_independent(a, a, timeout_us=2000) is False      # a kernel cannot overtake its own stream
    others = [torch.cuda.Stream() for _ in range(8)]
    assert any(hb.streams_independent(a, b, timeout_us=5000) for b in others)


@pytest.mark.gpu
@pytest.mark.timeout(200)
@pytest.mark.parametrize("towers,compact", [(2, False), (2, True), (4, False), (4, True)])
def test_in_graph_engine_towers_on_one_gpu_match_the_mean_gradient_oracle(towers, compact):
    """ONE client process, a stream and hipGraph chunks per tower, gradient mean inside the backward launches
    (plain pointers between the towers): bit-identical replicas that follow the CPU oracle of synchronous SGD
    on the mean of the towers' gradients, with one host call per tower per chunk.  (The streams of one process share
    4 hardware queues by default, and a third tower's spinning backward can end up queued behind a peer's: the client
    of four towers on ONE GPU asks the runtime for 16 queues.  Towers on GPUs of their own need no such thing.)"""
    from cluster_utils import ROOT

    env = dict(os.environ)
    if towers > 2:
        env["GPU_MAX_HW_QUEUES"] = "16"
    # a process of its own: one engine per client process is how the engine is used
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_ingraph_worker.py"), str(towers), str(int(compact)),
                          "5000"], cwd=ROOT, capture_output=True, text=True, timeout=180, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert res["healthy_5"], "an exchange between the towers timed out"
    assert res["step_5"] == 5 and res["identical_5"] and res["host_calls_per_chunk"] == 1
    assert res["oracle_error"] <= 1.0, res          # |got - oracle| <= 2e-5 + 2e-4 |oracle| everywhere
    assert res["healthy_14"] and res["identical_14"] and res["step_14"] == 14
    assert res["loss"] == res["loss"] and 0.0 <= res["acc"] <= 1.0


@pytest.mark.gpu
def test_the_single_process_all_reduce_binding_of_the_host_loop_towers_reaches_rccl():
    """``InGraphTowers`` with a tower per GPU sums the gradient slabs with ``torch.cuda.nccl.all_reduce`` (one process,
    a list of tensors, one per GPU).  One GPU here: a list of one tensor goes through the same binding into RCCL
    (communicator of one rank, the kernel launch) and must come back unchanged."""
    import torch
    import torch.cuda.nccl as nccl

    g = torch.arange(1000, dtype=torch.float32, device="cuda")
    assert nccl.is_available([g])
    nccl.all_reduce([g])
    torch.cuda.synchronize()
    assert torch.equal(g.cpu(), torch.arange(1000, dtype=torch.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards", [1, 2])
def test_gradient_step_that_reads_the_variables_from_shards_equals_the_plain_one(n_shards):
    """``MLPEngine.step_grad(shards=...)`` (a synchronous per-batch worker on mapped parameter-server shards: no pull in
    front of the step): with shards that hold the same values as the engine's own copy -- one shard, or whole variables
    dealt over two buffers as ``replica_device_setter`` deals them (tfdist_between.py:37-47) -- the gradients and the
    step's statistics are those of the plain gradient step and of the fp32 oracle within rounding (the engine's own copy of
    the variables holds NaNs here: it is not read)."""
    import torch

    from distributed_tensorflow_amd.models.mlp import MLPEngine
    from distributed_tensorflow_amd.ops import hip_backend as hb
    from distributed_tensorflow_amd.ops import reference as ref
    from distributed_tensorflow_amd.variables import ShardLayout, assign_shards, mlp_var_specs

    dev = torch.device("cuda", 0)
    specs = mlp_var_specs()
    assign_shards(specs, n_shards)
    layout = ShardLayout(specs, n_shards, mode="by_variable")
    slab = layout.new_slab(dev)
    layout.initialize(slab, 5)
    grads = [layout.new_slab(dev), layout.new_slab(dev)]
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(200, 784, generator=gen).to(dev)
    y = torch.eye(10)[torch.randint(0, 10, (200,), generator=gen)].to(dev)
    names = {"W1": "weights/Variable", "b1": "biases/Variable", "W2": "weights/Variable_1", "b2": "biases/Variable_1"}
    # the "parameter-server shards": buffers of their own with the shards' ranges of the slab
    shards = [slab[a:b].clone() for a, b in (layout.shard_range[k] for k in range(n_shards))]
    d = hb.DtfXgDesc()
    for k in range(n_shards):
        d.shard_start[k] = layout.shard_range[k][0]
        d.inbox[k] = shards[k].data_ptr()
    d.shard_start[n_shards] = layout.shard_range[n_shards - 1][1]
    d.state, d.n, d.rank, d.world, d.timeout_ms, d.step_index, d.compact = None, 0, 0, n_shards, 0, -1, 1
    d.offW1, d.offb1, d.offW2, d.offb2 = (layout.var_range[names[r]][0] for r in ("W1", "b1", "W2", "b2"))
    out = []
    for use_shards in (False, True):
        own = slab.clone() if not use_shards else torch.full_like(slab, float("nan"))   # (the own copy is not read)
        eng = MLPEngine({r: layout.view(own, n) for r, n in names.items()},
                        {r: layout.view(grads[use_shards], n) for r, n in names.items()}, batch=100, lr=0.1, act="sigmoid")
        eng.bind_data(x, y)
        eng.step_grad(1, 0.5, shards=d if use_shards else None)
        torch.cuda.synchronize()
        out.append((grads[use_shards].clone(), eng.last_loss()))
    # (the logits are summed with float atomics: two runs of one kernel agree within rounding, not bit for bit)
    assert bool(torch.isfinite(out[1][0]).all()) and abs(out[0][1] - out[1][1]) <= 1e-5 * max(1.0, abs(out[0][1]))
    p = layout.mlp_views(slab)
    want = ref.mlp_grads(x[100:200], y[100:200], p["W1"], p["b1"], p["W2"], p["b2"])
    for which in (0, 1):
        got = layout.mlp_views(out[which][0])
        for k in ("W1", "b1", "W2", "b2"):
            err = float((got[k] - 0.5 * want["d" + k]).abs().max())
            assert err <= 2e-5 * max(1.0, float(want["d" + k].abs().max())), (which, k, err)
