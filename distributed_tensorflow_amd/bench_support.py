"""Implementation of ``bench.py`` (kept in the package so tests can exercise it on CPU)."""
from __future__ import annotations

import os
import sys
import time
from typing import Optional

import torch

BASELINE_SINGLE = 55000.0 / 1.3
BASELINE_MULTI = 3 * 55000.0 / 6.03
BATCH = 100
STEPS_PER_EPOCH = 550
LR = 0.001
PEER_MODES = ("xgmi", "xgmi2", "xgmi2c")      # synchronous exchanges over peer-mapped HBM (parallel/xgmi.py)


def _dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _sync(device, distributed: bool):
    import torch.distributed as dist

    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _trial_xgmi(rank: int, local_rank: int, world: int, chunk: int) -> dict:
    """Run the peer-mapped exchange once in a child process per rank (see parallel/xgmi_probe.py).

    Called before this process initialises the GPU.  Whatever happens to the child - a refused
    mapping, a timeout in the exchange, replicas that disagree, a crash - stays in the child."""
    import json
    import subprocess

    port = int(os.environ.get("MASTER_PORT", "29500")) + 37
    if port > 65000:
        port -= 1000
    cmd = [sys.executable, "-m", "distributed_tensorflow_amd.parallel.xgmi_probe", "--rank", str(rank),
           "--world", str(world), "--device", str(local_rank), "--port", str(port), "--chunk", str(chunk),
           "--rows", "55000", "--steps", str(STEPS_PER_EPOCH)]
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
    # The child makes a rendezvous of its own: nothing of the launcher's may leak into it (with
    # TORCHELASTIC_USE_AGENT_STORE set, rank 0 would not host the store it is asked to connect to).
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                  "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE",
                                                  "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE"):
            env.pop(k)

    def attempt(extra, port_shift):
        cmd[cmd.index("--port") + 1] = str(port + port_shift)
        try:
            r = subprocess.run(cmd + extra, env=env, stdout=subprocess.PIPE, stderr=None, text=True, timeout=180)
            text, note = r.stdout, f"trial exited {r.returncode}"
        except subprocess.TimeoutExpired as e:      # the stages that finished have already reported
            text = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            note = "trial timed out"
        for line in reversed(text.splitlines()):
            if line.startswith("{"):
                try:
                    return json.loads(line)
                except ValueError:
                    break
        return {"ok": False, "error": f"{note}: {text[-300:]}"}

    res = attempt([], 0)
    res["uncached_grads"] = False
    # Replicas that ran to the end but disagree read a peer's gradient stale: every rank sees the same verdict
    # (it is computed from all ranks' checksums), so every rank repeats the trial with the slabs uncached too.
    if not res.get("ok") and res.get("identical") is False and res.get("healthy"):
        res = attempt(["--uncached", "1"], 1)
        res["uncached_grads"] = True
    return res


def run_benchmark(args) -> Optional[dict]:
    import torch.distributed as dist

    from .data import DeviceDataset, read_data_sets
    from .parallel.replicas import build_replica_trainer

    rank, local_rank, world = _dist_env()
    n = int(args.gpus)
    distributed = world > 1
    if distributed and world != n:
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}")
    if n > 1 and not distributed:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    mode = args.mode or ("single" if n == 1 else "auto")
    if not int(args.chunk):
        # one graph per epoch for the single-GPU step (8.20 vs 8.43 us/step with 50-step graphs: the fixed cost of
        # a replay is paid 11 times less often); the multi-rank modes keep their graphs short
        args.chunk = STEPS_PER_EPOCH if mode == "single" else 50
    # device_count() does not create a GPU context, so the trial below still runs ahead of ours
    use_cpu = bool(args.cpu) or torch.cuda.device_count() == 0
    # Test hook: every rank on GPU 0 with a gloo rendezvous (RCCL refuses two ranks on one device), so that
    # the whole multi-rank flow can be rehearsed on a one-GPU machine.
    share_gpu = os.environ.get("DTF_BENCH_SHARE_GPU") == "1" and not use_cpu
    dev_index = 0 if share_gpu else local_rank
    trial = None
    if mode == "auto" and distributed and not use_cpu and os.environ.get("DTF_BENCH_XGMI", "1") != "0":
        trial = _trial_xgmi(rank, dev_index, world, int(args.chunk))
    use_cpu = use_cpu or not torch.cuda.is_available()
    if use_cpu:
        device = torch.device("cpu")
    else:
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    if distributed:
        lib_backend = not (use_cpu or share_gpu)
        dist.init_process_group(backend="nccl" if lib_backend else "gloo", device_id=device if lib_backend else None)
    xgmi_ok = xgmi2_ok = xgmi2c_ok = peer_async_ok = False
    if trial is not None:
        flag = torch.tensor([1 if trial.get(k) else 0 for k in ("ok", "async_ok", "ok2", "ok2c")], dtype=torch.int32,
                            device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        xgmi_ok, peer_async_ok, xgmi2_ok, xgmi2c_ok = (bool(int(v)) for v in flag.tolist())
        if rank == 0 and not xgmi_ok:
            print(f"[bench] peer-mapped exchange not used on this node: {trial}", file=sys.stderr, flush=True)

    sizes = (55000, 5000, 10000) if not use_cpu else (5500, 100, 1000)
    mnist = read_data_sets("MNIST_data", one_hot=True, seed=0, sizes=sizes)
    # Every worker trains on its own independently shuffled copy (tfdist_between.py:92).
    ds = DeviceDataset(mnist.train.images, mnist.train.labels, device, BATCH, seed=1000 + rank)
    steps, warmup = int(args.steps), int(args.warmup)

    def build(m):
        # gloo collectives (the one-GPU rehearsal) cannot be captured; the peer-mapped exchange can
        graph = not args.no_graph and not (share_gpu and m not in PEER_MODES + ("xgmi_async",))
        extra = {"uncached_grads": bool(trial and trial.get("uncached_grads"))} if m == "xgmi" else {}
        return build_replica_trainer(mode=m, device=device, dataset=ds, batch=BATCH, lr=LR, seed=1,
                                     rank=rank, world=world, chunk=int(args.chunk), use_graph=graph, **extra)

    def prepare(trainer, hint):
        # Capture the hipGraph chunks outside the timed region.  If capture fails on any rank
        # (e.g. an RCCL build that cannot be captured) every rank falls back to eager launches.
        ok = 1
        try:
            trainer.prepare(steps_hint=hint)
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"[bench rank {rank}] graph capture failed ({type(e).__name__}: {e}); using eager launches",
                  file=sys.stderr, flush=True)
            if device.type == "cuda":
                # a capture that died may leave its side stream current: go back to the default stream
                torch.cuda.set_stream(torch.cuda.default_stream(device))
        if distributed:
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok and trainer.use_graph:
            trainer.use_graph = False
            trainer._graphs.clear()
            trainer.engine.reset_step_workspace(trainer.engine.parity)

    def retire(tr):
        tr._graphs.clear()
        if hasattr(tr, "close"):
            _sync(device, distributed)
            tr.close()

    # Four synchronous data-parallel exchanges compute the same update (the mean gradient applied once):
    # PS shard per GPU (reduce-scatter -> sharded SGD -> all-gather) and replicated apply after one
    # all-reduce, both over RCCL, and two exchanges over peer-mapped HBM -- one kernel after the backward, or
    # inside the backward launch (parallel/xgmi.py; offered only after their trial run in child processes
    # succeeded on this node).  Which one is faster for a 318 KB
    # message is a property of the RCCL build and the xGMI topology, so "auto" times a short probe of each
    # (untimed, before the warm-up) and keeps the winner.
    probe_ms = None
    if mode == "auto":
        probe = max(2 * int(args.chunk), 100)
        probe_ms, best = {}, None
        # (the one-GPU rehearsal keeps its gloo group out of the per-step path whenever it can)
        library = ("sync", "towers") if not share_gpu else (() if xgmi_ok else ("towers",))
        for cand in ((("xgmi2c",) if xgmi2c_ok else ()) + (("xgmi2",) if xgmi2_ok else ())
                     + (("xgmi",) if xgmi_ok else ()) + library):
            if rank == 0:
                print(f"[bench] probing {cand}", file=sys.stderr, flush=True)
            try:
                t = build(cand)
                prepare(t, (probe, probe))
                t.run_steps(probe)
                _sync(device, distributed)
                tp = time.perf_counter()
                t.run_steps(probe)
                _sync(device, distributed)
                dt = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=device)
                if distributed:
                    dist.all_reduce(dt, op=dist.ReduceOp.MAX)     # identical decision on every rank
                if cand in PEER_MODES:
                    # an exchange that gave up on a peer skips its work from then on and would look FAST
                    sane = torch.tensor([1 if t.healthy() else 0], dtype=torch.int32, device=device)
                    if distributed:
                        dist.all_reduce(sane, op=dist.ReduceOp.MIN)
                    if not int(sane.item()):
                        retire(t)       # (every rank is here: the verdict was reduced over all of them)
                        raise RuntimeError("an exchange timed out during the probe")
            except Exception as e:  # noqa: BLE001 — a candidate that cannot run here is simply not chosen
                print(f"[bench rank {rank}] candidate {cand} failed ({type(e).__name__}: {e})", file=sys.stderr,
                      flush=True)
                probe_ms[cand] = None
                continue
            probe_ms[cand] = round(1000.0 * float(dt.item()) / probe, 5)
            if best is None or probe_ms[cand] < probe_ms[best[0]]:
                if best is not None:
                    retire(best[1])
                best = (cand, t)
            else:
                retire(t)
            del t
        if best is None:
            raise SystemExit("no data-parallel mode could run on this node")
        mode, trainer = best
        if mode not in PEER_MODES:
            fallback_mode = None
        else:       # should the peer-mapped run turn out inconsistent: the fastest library mode
            fallback_mode = min((m for m in probe_ms if m not in PEER_MODES and probe_ms[m] is not None),
                                key=lambda m: probe_ms[m], default=None)
    else:
        trainer = build(mode)
        fallback_mode = None

    def timed(tr):
        # warm-up, timed run, then the like-for-like epoch below (five 100-step windows and one of 50)
        prepare(tr, (warmup, steps) + (100,) * (STEPS_PER_EPOCH // 100) + ((STEPS_PER_EPOCH % 100,) if STEPS_PER_EPOCH % 100 else ()))
        tr.run_steps(warmup)
        _sync(device, distributed)
        t0 = time.perf_counter()
        tr.run_steps(steps)
        _sync(device, distributed)
        return time.perf_counter() - t0

    elapsed = timed(trainer)
    if mode in PEER_MODES:
        # The replicas sum in rank order, so they must be bit-identical; anything else (or an exchange that
        # timed out) voids the measurement and the run is repeated with the library collectives.
        good = torch.tensor([1 if trainer.healthy() else 0, trainer.replica_checksum()], dtype=torch.int64,
                            device=device)
        seen = [torch.zeros_like(good) for _ in range(world)] if distributed else [good]
        if distributed:
            dist.all_gather(seen, good)
        consistent = all(int(g[0]) == 1 and int(g[1]) == int(seen[0][1]) for g in seen)
        if not consistent and fallback_mode is not None:
            if rank == 0:
                print("[bench] replicas disagree after the peer-mapped run; repeating with " + fallback_mode,
                      file=sys.stderr, flush=True)
            retire(trainer)
            mode, trainer = fallback_mode, build(fallback_mode)
            elapsed = timed(trainer)
        elif not consistent:
            raise SystemExit("peer-mapped exchange: replicas disagree")

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The asynchronous mode next to the headline (synchronous / single) number: between-graph workers that
    # pull and push one-sidedly against GPU-resident shards (parallel/replicas.py::XgmiAsyncTrainer).  One
    # process needs no peer and measures it here; with several GPUs the number is the one the trial's child
    # processes measured (same kernels, same data shape), so that nothing after the headline run can cost it.
    async_report = None
    if distributed and trial is not None and peer_async_ok and trial.get("async_ms_per_step"):
        ms = float(trial["async_ms_per_step"])
        async_report = {"value": round(n * BATCH * 1000.0 / ms, 1), "unit": "samples/s", "ms_per_step": ms,
                        "steps": STEPS_PER_EPOCH, "global_step": trial.get("async_global_step"),
                        "every_push_counted": True, "measured_in": "trial child processes",
                        "parallelism": f"async PS dp{n}: shard per GPU in HBM, one-sided pull and lock-free SGD push "
                                       + ("inside the backward launch (float atomics), 3 launches/step, "
                                          if trial.get("async_fused_push", False) else "as a launch, 4 launches/step, ")
                                       + "through peer mappings, no barrier"}
    elif not use_cpu and not distributed and mode != "xgmi_async" and os.environ.get("DTF_BENCH_ASYNC", "1") != "0":
        try:
            a_steps = min(steps, 4 * STEPS_PER_EPOCH)
            ta = build("xgmi_async")
            prepare(ta, (warmup, a_steps))
            ta.run_steps(warmup)
            _sync(device, distributed)
            ta0 = time.perf_counter()
            ta.run_steps(a_steps)
            _sync(device, distributed)
            a_el = time.perf_counter() - ta0
            a_gs = ta.global_step_value()
            a_loss, a_acc = ta.evaluate(torch.as_tensor(mnist.test.images).to(device),
                                        torch.as_tensor(mnist.test.labels).to(device))
            async_report = {"value": round(n * BATCH * a_steps / a_el, 1), "unit": "samples/s",
                            "ms_per_step": round(1000.0 * a_el / a_steps, 5), "steps": a_steps,
                            "global_step": a_gs, "every_push_counted": a_gs == n * (warmup + a_steps),
                            "test_accuracy_after_run": round(a_acc, 4), "test_loss_after_run": round(a_loss, 4),
                            "parallelism": ta.describe()}
            retire(ta)
        except Exception as e:  # noqa: BLE001 — the extra measurement must never cost the headline one
            async_report = {"error": f"{type(e).__name__}: {e}"}

    test_x = torch.as_tensor(mnist.test.images).to(device)
    test_y = torch.as_tensor(mnist.test.labels).to(device)
    loss, acc = trainer.evaluate(test_x, test_y)
    last_cost = trainer.last_loss()
    global_step = trainer.global_step_value()

    # Like-for-like epoch time: 550 steps + per-100-step host fetch of the cost +
    # full test-set evaluation, as the reference's "Total Time" does (tfdist_between.py:86-111).
    epoch_s = None
    if steps >= STEPS_PER_EPOCH:
        _sync(device, distributed)
        te = time.perf_counter()
        done = 0
        while done < STEPS_PER_EPOCH:
            k = min(100, STEPS_PER_EPOCH - done)
            trainer.run_steps(k)
            trainer.last_loss()
            done += k
        trainer.evaluate(test_x, test_y)
        _sync(device, distributed)
        epoch_s = time.perf_counter() - te

    result = None
    if rank == 0:
        value = n * BATCH * steps / elapsed
        base = BASELINE_SINGLE if n == 1 else BASELINE_MULTI
        result = {
            "metric": "samples/sec (784-100-10 MLP, batch 100/worker, SGD)",
            "value": round(value, 1),
            "unit": "samples/s",
            "n_gpus": n,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(1000.0 * elapsed / steps, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": round(value / base, 2),
            "dtype": "fp32",
            "data": f"synthetic MNIST-shaped ({mnist.source}), random-init N(0,1) weights",
            "config": {
                "model": "tfsingle/tfdist_between 2-layer FC MLP 784-100-10 sigmoid+softmax",
                "global_batch": n * BATCH,
                "seq_len": None,
                "parallelism": trainer.describe(),
                "optimizer": "SGD lr=0.001",
                "device": "cpu" if use_cpu else torch.cuda.get_device_name(device),
            },
            "sec_per_epoch": None if epoch_s is None else round(epoch_s, 5),
            "sec_per_epoch_baseline": 1.3 if n == 1 else 5.35,
            "test_accuracy_after_run": round(acc, 4),
            "test_loss_after_run": round(loss, 4),
            "last_cost": round(last_cost, 4),
            "global_step": global_step,
        }
        if probe_ms is not None:
            result["mode_probe_ms_per_step"] = probe_ms
        if async_report is not None:
            result["async_mode"] = async_report
        if trial is not None:
            result["peer_exchange_trial"] = {k: trial.get(k) for k in ("ok", "ms_per_step", "identical", "healthy", "uncached_grads", "async_ok", "async_fused_push", "async_ms_per_step_by_push",
                                                       "async_ms_per_step", "async_loss", "ok2", "ms_per_step2", "ok2c", "ms_per_step2c",
                                                       "error", "error2", "error2c")
                                             if k in trial}
    if hasattr(trainer, "close"):
        _sync(device, distributed)
        trainer.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return result
