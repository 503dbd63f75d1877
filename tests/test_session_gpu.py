"""GPU: the TF-style Session takes the fused HIP path; replica trainers capture hipGraphs; RCCL world of one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.examples.mnist_mlp import build_model
from distributed_tensorflow_amd.ops import reference as ref

from cluster_utils import ROOT, cluster_dict, free_ports

pytestmark = pytest.mark.gpu


def _batch(seed, n=100):
    rng = np.random.RandomState(seed)
    return rng.rand(n, 784).astype(np.float32), np.eye(10, dtype=np.float32)[rng.randint(0, 10, n)]


def test_local_session_runs_fused_kernels_and_matches_reference():
    x, y = _batch(0)
    with tf.Graph().as_default():
        with tf.device("/gpu:0"):
            m = build_model(0.1)
        with tf.Session(config=tf.ConfigProto(allow_soft_placement=True)) as sess:
            assert sess.device.type == "cuda"
            sess.run(m.init_op)
            p = {k: sess.var_tensor(getattr(m, k)).double().cpu() for k in ("W1", "b1", "W2", "b2")}
            for _ in range(3):
                _, cost, summ, step = sess.run([m.train_op, m.cross_entropy, m.summary_op, m.global_step],
                                               {m.x: x, m.y_: y})
            assert sess.fused_steps == 3 and sess.generic_steps == 0 and step == 2.0
            for _ in range(3):
                out = ref.mlp_grads(torch.tensor(x).double(), torch.tensor(y).double(), p["W1"], p["b1"], p["W2"], p["b2"])
                for k in p:
                    p[k] = p[k] - 0.1 * out["d" + k]
            for k in p:
                torch.testing.assert_close(sess.var_tensor(getattr(m, k)).cpu(), p[k].float(), rtol=2e-4, atol=2e-5)
            assert abs(cost - float(out["loss"])) < 1e-3 and dict(summ.values)["cost"] == pytest.approx(cost)
            tx, ty = _batch(9, 10000)
            acc = sess.run(m.accuracy, {m.x: tx, m.y_: ty})
            la = ref.mlp_loss_acc(torch.tensor(tx).double(), torch.tensor(ty).double(), p["W1"], p["b1"], p["W2"], p["b2"])
            assert abs(acc - float(la[1])) < 2e-3


def test_ps_session_on_gpu_worker_with_host_shards():
    cluster = tf.train.ClusterSpec(cluster_dict(2, 1))
    cfg = tf.ConfigProto(allow_soft_placement=True, operation_timeout_in_ms=20000)
    servers = [tf.train.Server(cluster, "ps", i, config=cfg) for i in range(2)]
    w = tf.train.Server(cluster, "worker", 0, config=cfg)
    try:
        with tf.Graph().as_default() as g:
            with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:0/gpu:0", cluster=cluster)):
                m = build_model(0.05)
            sv = tf.train.Supervisor(graph=g, is_chief=True, global_step=m.global_step, init_op=m.init_op)
            with sv.prepare_or_wait_for_session(w.target) as sess:
                assert sess.device.type == "cuda"
                x, y = _batch(1)
                c0 = sess.run(m.cross_entropy, {m.x: x, m.y_: y})
                for _ in range(20):
                    _, c, step = sess.run([m.train_op, m.cross_entropy, m.global_step], {m.x: x, m.y_: y})
                assert c < c0 and step == 19.0 and sess.store.global_step() == 20 and sess.fused_steps == 20
            sv.stop()
    finally:
        for s in servers + [w]:
            s.stop()


@pytest.mark.parametrize("mode", ["sync", "towers", "async_lockstep"])
def test_rccl_world_of_one_with_graph_capture(mode):
    """The collective step (kernels + RCCL calls + sharded SGD) captured into hipGraphs; world size 1."""
    port = free_ports(1)[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), mode, "cuda", "graph"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["global_step"] == 13 and "hipGraph" in res["desc"] and res["loss"] == res["loss"]


@pytest.mark.parametrize("script,n_ps,n_w,board", [("tfdist_between.py", 1, 1, "1"), ("tfdist_between.py", 2, 2, "1"),
                                                   ("tfdist_between_sync.py", 1, 2, "1"),
                                                   ("tfdist_between_sync.py", 2, 3, "1"),
                                                   ("tfdist_between_sync.py", 1, 2, "0"),
                                                   ("tfdist_between_sync.py", 1, 2, "1-interpreter")])
def test_ipc_transport_gpu_resident_shards(script, n_ps, n_w, board, tmp_path):
    """PS and workers as separate processes sharing the GPU; shards live in the PS task's HBM (HIP-IPC).  The
    synchronous barrier runs over the shared-memory board of the node (board "1": the chief's common rounds as native
    code, ``dtf_sync_rounds``; "1-interpreter": every round in the interpreter) or over the task server's sockets."""
    import re

    from cluster_utils import launch_script

    args = ["--epochs=1", "--train_size=1100", "--freq=5", "--lr=0.01", "--nosummaries", "--transport=ipc",
            f"--logs_path={tmp_path}"]
    res = launch_script(script, n_ps, n_w, args, timeout=200, env_extra={"DTF_SYNC_BOARD": board[0], "DTF_SYNC_NATIVE_ROUND": "0" if board.endswith("interpreter") else "1"})
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    assert "resident on cuda" in res["ps0"][1]
    finals = []
    for i in range(n_w):
        out = res[f"worker{i}"][1]
        assert out.rstrip().endswith("Done")
        steps = [int(s) for s in re.findall(r"Step: (\d+),", out)]
        finals.append(steps[-1])
    # 'Step' is the global step pulled BEFORE the worker's own update, plus one (tfdist_between.py:102): workers
    # whose last batches overlap pull the same value, so up to n_w - 1 updates may be missing from the largest print
    if "sync" in script:
        assert finals == [11] * n_w
    else:
        assert 11 * n_w - (n_w - 1) <= max(finals) <= 11 * n_w
        # asynchronous workers on mapped shards: forward that reads the variables from the shards, backward that pushes
        # (two launches per sess.run, no pull, no separate apply; the step counter comes back from the push itself)
        for i in range(n_w):
            assert "11 on the fused kernels (11 of them pushed" in res[f"worker{i}"][1]


@pytest.mark.parametrize("n_ps,n_w", [(1, 1), (2, 2)])
def test_resident_async_workers_replay_hipgraphs_against_ipc_shards(n_ps, n_w, tmp_path):
    """Separate worker processes, each replaying captured chunks (pull copy, 2 launches, one-sided SGD push)
    against shards living in the ps tasks' HBM; every push is an update of the shared variables."""
    import re

    from cluster_utils import launch_script

    args = ["--epochs=2", "--train_size=2200", "--freq=11", "--lr=0.01", "--transport=ipc", "--resident",
            f"--logs_path={tmp_path}"]
    res = launch_script("tfdist_between.py", n_ps, n_w, args, timeout=200)
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    finals, costs = [], []
    for i in range(n_w):
        out = res[f"worker{i}"][1]
        assert "hipGraph chunks" in out, out[-2000:]
        assert "measured us/step: fused" in out, out[-2000:]     # both forms of the push were timed, one was kept
        assert out.rstrip().endswith("Done")
        finals.append([int(s) for s in re.findall(r"Step: (\d+),", out)][-1])
        costs.append([float(c) for c in re.findall(r"Cost: ([\d.]+)", out)])
    # 'Step' is the global step pulled BEFORE the worker's own update, plus one (tfdist_between.py:102), and a resident
    # worker prints the value its last WINDOW pulled: workers whose last windows run side by side (they usually do not:
    # one of them tends to finish first and the other then prints 88) each miss the other's updates still in flight.
    # Observed when they do overlap: 86 and 85 of 88.
    # (bound: every other worker may still have a whole window of 11 updates in flight when the last window pulls)
    assert 2 * 22 * n_w - 11 * (n_w - 1) <= max(finals) <= 2 * 22 * n_w
    assert all(c[-1] < c[0] for c in costs)                 # the shared model learns


@pytest.mark.parametrize("n_ps,n_w", [(2, 2)])
def test_resident_async_workers_of_a_deeper_stack_replay_hipgraphs_against_ipc_shards(n_ps, n_w, tmp_path):
    """tfdist_between.py --resident --hidden 64,32: chunks of pull -> dense-layer kernels -> one-sided SGD push, captured
    per worker process, against shards in the ps tasks' HBM."""
    import re

    from cluster_utils import launch_script

    args = ["--epochs=2", "--train_size=2200", "--freq=11", "--lr=0.05", "--transport=ipc", "--resident",
            "--hidden=64,32", "--activation=relu", f"--logs_path={tmp_path}"]
    res = launch_script("tfdist_between.py", n_ps, n_w, args, timeout=200)
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    finals, costs = [], []
    for i in range(n_w):
        out = res[f"worker{i}"][1]
        assert "dense stack 784-64-32-10" in out and "hipGraph chunks" in out, out[-2000:]
        assert out.rstrip().endswith("Done")
        finals.append([int(s) for s in re.findall(r"Step: (\d+),", out)][-1])
        costs.append([float(c) for c in re.findall(r"Cost: ([\d.]+)", out)])
    assert 2 * 22 * n_w - 11 * (n_w - 1) <= max(finals) <= 2 * 22 * n_w       # every push is an update (see above)
    assert all(c[-1] < c[0] for c in costs)


@pytest.mark.parametrize("n_ps,n_w,transport", [(1, 2, "ipc"), (2, 2, "tcp")])
def test_resident_sync_workers_exchange_inside_the_backward_launch(n_ps, n_w, transport, tmp_path):
    """tfdist_between_sync.py --resident: a replica per worker process, gradients swapped through peer-mapped
    inboxes (handles rendezvous over the ps control store), one update per n_w batches, identical workers."""
    import re

    from cluster_utils import launch_script

    args = ["--epochs=2", "--train_size=2200", "--freq=11", "--lr=0.01", f"--transport={transport}", "--resident",
            f"--logs_path={tmp_path}"]
    res = launch_script("tfdist_between_sync.py", n_ps, n_w, args, timeout=200)
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    steps, costs, accs = [], [], []
    for i in range(n_w):
        out = res[f"worker{i}"][1]
        assert "inside the backward launch" in out and "hipGraph chunks" in out, out[-2000:]
        assert out.rstrip().endswith("Done")
        steps.append([int(v) for v in re.findall(r"Step: (\d+),", out)])
        costs.append([float(c) for c in re.findall(r"Cost: ([\d.]+)", out)])
        accs.append(re.findall(r"Test-Accuracy: ([\d.]+)", out))
    assert steps[0] == steps[1] and steps[0][-1] == 2 * 22          # one update per n_w batches, in lock step
    assert accs[0] == accs[1] and len(accs[0]) == 2                 # bit-identical replicas evaluate identically
    assert all(c[-1] < c[0] for c in costs)


@pytest.mark.parametrize("n_ps,n_w,transport", [(1, 2, "ipc"), (2, 3, "tcp")])
def test_resident_sync_workers_of_a_deeper_stack_stay_identical(n_ps, n_w, transport, tmp_path):
    """tfdist_between_sync.py --resident --hidden 64,32: a replica per worker process, the stack's gradient launches and
    one exchange kernel per step (sum over every worker's slab through HIP-IPC mappings), one update per n_w batches."""
    import re

    from cluster_utils import launch_script

    args = ["--epochs=2", "--train_size=2200", "--freq=11", "--lr=0.05", f"--transport={transport}", "--resident",
            "--hidden=64,32", "--activation=relu", f"--logs_path={tmp_path}"]
    res = launch_script("tfdist_between_sync.py", n_ps, n_w, args, timeout=200)
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    steps, costs, accs = [], [], []
    for i in range(n_w):
        out = res[f"worker{i}"][1]
        assert "dense stack 784-64-32-10" in out and "one exchange kernel per step" in out and "hipGraph chunks" in out, out[-2000:]
        assert out.rstrip().endswith("Done")
        steps.append([int(v) for v in re.findall(r"Step: (\d+),", out)])
        costs.append([float(c) for c in re.findall(r"Cost: ([\d.]+)", out)])
        accs.append(re.findall(r"Test-Accuracy: ([\d.]+)", out))
    assert all(s == steps[0] for s in steps) and steps[0][-1] == 2 * 22     # one update per n_w batches, in lock step
    assert all(a == accs[0] for a in accs) and len(accs[0]) == 2            # bit-identical replicas evaluate identically
    assert all(c[-1] < c[0] for c in costs)


@pytest.mark.parametrize("hidden", [None, "64,32"])
def test_resident_sync_workers_with_a_backup_worker(hidden, tmp_path):
    """tfdist_between_sync.py --resident --replicas_to_aggregate=2 with three workers (SyncReplicasOptimizer with
    replicas_to_aggregate < total_num_replicas; the reference passes equal values, tfdist_between_sync.py:66-67): an
    update is the mean of the first two gradients to arrive, decided inside the exchange kernel; all three workers apply
    it, stay bit-identical and in lock step.  The reference's network and a deeper stack."""
    import re

    from cluster_utils import launch_script

    args = ["--epochs=2", "--train_size=2200", "--freq=11", "--lr=0.02", "--transport=ipc", "--resident",
            "--replicas_to_aggregate=2", f"--logs_path={tmp_path}"]
    if hidden:
        args += [f"--hidden={hidden}", "--activation=relu"]
    res = launch_script("tfdist_between_sync.py", 1, 3, args, timeout=200)
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    steps, costs, accs = [], [], []
    for i in range(3):
        out = res[f"worker{i}"][1]
        assert "first 2" in out and "hipGraph chunks" in out, out[-2000:]
        assert out.rstrip().endswith("Done")
        steps.append([int(v) for v in re.findall(r"Step: (\d+),", out)])
        costs.append([float(c) for c in re.findall(r"Cost: ([\d.]+)", out)])
        accs.append(re.findall(r"Test-Accuracy: ([\d.]+)", out))
    assert all(s == steps[0] for s in steps) and steps[0][-1] == 2 * 22     # one update per round of the three workers
    assert all(a == accs[0] for a in accs) and len(accs[0]) == 2            # bit-identical replicas evaluate identically
    assert all(c[-1] < c[0] for c in costs)


def test_checkpoints_during_a_resident_sync_run_save_the_replica_and_leave_it_alone(tmp_path):
    """tfdist_between_sync.py --resident --hidden 64,32 with a checkpoint every 0.2 s: while the trainer is attached
    the parameter-server shards still hold the initial values, so a save that pulled them would write a stale
    checkpoint AND set the chief's replica back (the workers would then differ).  The save must read the replica."""
    import re

    from cluster_utils import launch_script

    from distributed_tensorflow_amd.train.saver import latest_checkpoint, read_bundle

    logdir = tmp_path / "ckpt"
    args = ["--epochs=6", "--train_size=2200", "--freq=11", "--lr=0.05", "--transport=ipc", "--resident",
            "--hidden=64,32", "--activation=relu", f"--logs_path={tmp_path}", f"--logdir={logdir}",
            "--save_model_secs=0.2"]
    res = launch_script("tfdist_between_sync.py", 1, 2, args, timeout=200)
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    accs, costs = [], []
    for i in range(2):
        out = res[f"worker{i}"][1]
        assert out.rstrip().endswith("Done")
        accs.append(re.findall(r"Test-Accuracy: ([\d.]+)", out))
        costs.append([float(c) for c in re.findall(r"Cost: ([\d.]+)", out)])
    assert accs[0] == accs[1] and len(accs[0]) == 6          # the chief was never set back: replicas stayed identical
    assert all(c[-1] < c[0] for c in costs)
    last = latest_checkpoint(str(logdir))
    assert last is not None
    saved = read_bundle(last)
    step = int(round(float(saved["global_step"])))
    assert 0 < step <= 6 * 22                                 # a checkpoint of the TRAINED replica, not of the stale shards
    assert last.endswith(f"-{step}")


@pytest.mark.timeout(300)
def test_more_resident_sync_workers_than_one_gpu_can_keep_spinning_use_the_bounded_exchange(tmp_path):
    """1 ps + 4 synchronous workers, all on the one GPU (the reference's scenario 2 with more workers): their
    cross-replica backward launches could not all be resident at once, so the workers agree on the one-kernel
    exchange after the backward instead of spinning into the exchange's time limit."""
    import re

    from cluster_utils import launch_script

    args = ["--epochs=2", "--train_size=2200", "--freq=11", "--lr=0.01", "--transport=ipc", "--resident",
            f"--logs_path={tmp_path}"]
    res = launch_script("tfdist_between_sync.py", 1, 4, args, timeout=250)
    for name, (rc, out) in res.items():
        assert rc == 0, f"{name} exited {rc}:\n{out[-3000:]}"
    steps, accs = [], []
    for i in range(4):
        out = res[f"worker{i}"][1]
        assert "one bounded kernel after the backward" in out and out.rstrip().endswith("Done"), out[-2000:]
        steps.append([int(v) for v in re.findall(r"Step: (\d+),", out)])
        accs.append(re.findall(r"Test-Accuracy: ([\d.]+)", out))
    assert all(s == steps[0] for s in steps) and steps[0][-1] == 2 * 22      # one update per 4 batches, lock step
    assert all(a == accs[0] for a in accs) and len(accs[0]) == 2            # bit-identical replicas


def _single_trainer(double_buffer, pipelined, n=2200, batch=100, chunk=5):
    import numpy as np

    from distributed_tensorflow_amd.data import DeviceDataset
    from distributed_tensorflow_amd.parallel.replicas import SingleTrainer

    rng = np.random.RandomState(3)
    x = rng.rand(n, 784).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.randint(0, 10, n)]
    dev = torch.device("cuda")
    ds = DeviceDataset(x, y, dev, batch, seed=11, double_buffer=double_buffer)
    return SingleTrainer(device=dev, dataset=ds, batch=batch, lr=0.05, seed=1, chunk=chunk, use_graph=True,
                         pipelined=pipelined)


def test_one_launch_steps_equal_two_launch_steps_across_epochs():
    """22 batches per epoch, 60 steps: the pipelined trainer (Gram blocks refreshed at every reshuffle, odd and even
    chunks, chunk ends at epoch ends) follows the two-launch trainer."""
    a = _single_trainer(False, True)
    b = _single_trainer(False, False)
    assert a.pipelined and not b.pipelined and "1 launch/step" in a.describe()
    for n in (7, 22, 31):
        a.run_steps(n)
        b.run_steps(n)
    torch.cuda.synchronize()
    torch.testing.assert_close(a.params, b.params, rtol=5e-4, atol=5e-5)
    assert a.global_step_value() == b.global_step_value() == 60
    assert abs(a.last_loss() - b.last_loss()) < 1e-3 * max(1.0, abs(b.last_loss()))


def test_double_buffered_dataset_gives_the_same_run():
    """The side-stream prefetch of the next epoch's order (two buffers, per-buffer graphs) does not change the run."""
    a = _single_trainer(True, True)
    b = _single_trainer(False, True)
    assert a.ds.double_buffer and not b.ds.double_buffer
    a.prepare(steps_hint=(50, 16))
    for n in (50, 16):
        a.run_steps(n)
        b.run_steps(n)
    torch.cuda.synchronize()
    torch.testing.assert_close(a.params, b.params, rtol=1e-5, atol=1e-6)
    assert a.ds.epoch == b.ds.epoch == 2 and a.ds.active == 0


@pytest.mark.parametrize("towers", [2, 4])
def test_in_graph_script_drives_its_towers_with_one_graph_launch_per_tower_and_chunk(towers, tmp_path):
    """tfdist.py (in-graph replication): one client, two or four towers (sharing the GPU here; for four the script
    asks the runtime for a hardware queue per tower before its first GPU call), hipGraph chunks per tower, gradient
    mean inside the backward launches; same stdout contract as the other scripts."""
    import re
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tfdist.py"), f"--num_towers={towers}", "--epochs=2",
                        "--train_size=2200", "--freq=11", "--lr=0.01", f"--logs_path={tmp_path}"],
                       cwd=ROOT, capture_output=True, text=True, timeout=200, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    out = p.stdout
    assert f"in-graph: 1 client, {towers} towers" in out and "inside the towers' backward launches" in out
    assert f"tower_{towers - 1}: /job:localhost/replica:0/task:0/gpu:0" in out and out.rstrip().endswith("Done")
    steps = [int(v) for v in re.findall(r"Step: (\d+),", out)]
    costs = [float(c) for c in re.findall(r"Cost: ([\d.]+)", out)]
    assert steps[-1] == 2 * 22 and costs[-1] < costs[0]          # one averaged update per step of both towers
    assert len(re.findall(r"Test-Accuracy: ([\d.]+)", out)) == 2


@pytest.mark.parametrize("towers", [2, 4])
def test_in_graph_script_with_a_deeper_stack_on_the_gpu(towers, tmp_path):
    """tfdist.py --hidden 64,32 --num_towers 2 | 4 (towers sharing the GPU; four: the script asks for a hardware queue
    per tower): the dense-layer kernels per tower, one averaged update per step, the reference's progress lines."""
    import os
    import subprocess
    import sys

    from cluster_utils import ROOT

    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tfdist.py"), "--hidden", "64,32", "--activation", "relu",
                          "--num_towers", str(towers), "--lr", "0.05", "--epochs", "2", "--train_size", "2200", "--logs_path",
                          str(tmp_path)], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=200, env=dict(env, PYTHONPATH=ROOT))
    assert out.returncode == 0, out.stdout[-2000:]
    assert f"{towers} towers of a dense stack 784-64-32-10" in out.stdout
    assert f"tower_{towers - 1}: /job:localhost/replica:0/task:0/gpu:0" in out.stdout
    assert "Step: 44,  Epoch:  2,  Batch:  22 of  22" in out.stdout and out.stdout.rstrip().endswith("Done")
    accs = [float(line.split(":")[1]) for line in out.stdout.splitlines() if line.startswith("Test-Accuracy")]
    assert len(accs) == 2 and accs[-1] > 0.5
