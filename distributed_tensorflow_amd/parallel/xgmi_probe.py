"""Trial run of the peer-mapped gradient exchange in a process of its own.

``bench.py`` (and anything else that wants the one-kernel exchange on a node it has never run on)
starts one of these per rank BEFORE it touches the GPU itself::

    python -m distributed_tensorflow_amd.parallel.xgmi_probe --rank R --world N --device D --port P

The replicas rendezvous over gloo on 127.0.0.1:P, map each other's buffers, train a few hundred
steps and then check what matters: no exchange timed out, every replica holds bit-identical
variables (the sums are taken in rank order, so anything else means a peer's gradient was read
stale or torn), and the variables moved.  The asynchronous mode (one-sided pulls and lock-free pushes on
shards in the peers' HBM) is tried the same way: finite variables, every push counted in ``global_step``, a
loss that went down.  One JSON line reports ``ok`` / ``async_ok`` and the times per step; the exit code is 0
only when ``ok``.  A crash, a hang (the caller's timeout) or a disagreement costs the
caller nothing but the fallback to the library collectives.
"""
from __future__ import annotations

import argparse
import json
import sys
import time


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--chunk", type=int, default=50)
    ap.add_argument("--rows", type=int, default=5500)
    ap.add_argument("--uncached", type=int, default=0)
    ap.add_argument("--budget", type=float, default=0.0,
                    help="seconds after which no further stage is started (0: no limit); rank 0's clock decides")
    ap.add_argument("--only", default="", help="run just this stage (xgmi)")
    args = ap.parse_args(argv)
    t_start = time.monotonic()

    import torch
    import torch.distributed as dist

    from ..data import DeviceDataset
    from .replicas import build_replica_trainer

    out = {"ok": False, "rank": args.rank, "world": args.world}

    def note(msg):
        print(f"[peer-exchange trial {args.rank}/{args.world}] {msg}", file=sys.stderr, flush=True)

    def go_on(stage: str) -> bool:
        """Collective: start another stage only while rank 0 is inside the budget (and the stage is wanted)."""
        if args.only:
            return False
        flag = torch.tensor([1 if (args.budget <= 0 or time.monotonic() - t_start < args.budget) else 0])
        dist.broadcast(flag, src=0)
        if not int(flag.item()):
            out.setdefault("skipped", []).append(stage)
        return bool(int(flag.item()))

    try:
        dist.init_process_group("gloo", rank=args.rank, world_size=args.world,
                                init_method=f"tcp://127.0.0.1:{args.port}")
        dev = torch.device("cuda", args.device)
        torch.cuda.set_device(dev)
        g = torch.Generator().manual_seed(4242 + args.rank)
        x = torch.rand(args.rows, 784, generator=g)
        y = torch.eye(10)[torch.randint(0, 10, (args.rows,), generator=g)]
        ds = DeviceDataset(x, y, dev, 100, seed=1000 + args.rank)
        t = build_replica_trainer("xgmi", device=dev, dataset=ds, batch=100, lr=0.001, seed=1, rank=args.rank,
                                  world=args.world, chunk=args.chunk, use_graph=True,
                                  uncached_grads=bool(args.uncached), timeout_ms=10000)
        note("buffers mapped")
        p0 = t.params.clone()
        t.prepare((args.steps, args.steps))
        note("graphs captured")
        dist.barrier()
        t.run_steps(args.steps)
        torch.cuda.synchronize(dev)
        note("warm-up steps done")
        dist.barrier()
        t0 = time.perf_counter()
        t.run_steps(args.steps)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        healthy = t.healthy()
        sums = [None] * args.world
        dist.all_gather_object(sums, (t.replica_checksum(), healthy, dt))
        moved = float((t.params - p0)[4:].abs().max())
        finite = bool(torch.isfinite(t.params).all())
        out.update(healthy=all(s[1] for s in sums), identical=len({s[0] for s in sums}) == 1, moved=moved,
                   finite=finite, ms_per_step=round(1000.0 * max(s[2] for s in sums) / args.steps, 5),
                   global_step=t.global_step_value(), exchanges=t.xg.exchanges_done())
        out["ok"] = bool(out["healthy"] and out["identical"] and finite and moved > 0.0
                         and out["global_step"] == 2 * args.steps)
        note(f"ok={out['ok']} {out['ms_per_step']} ms/step")
        print(json.dumps(out), flush=True)        # the caller reads the LAST line: later stages only add to it
        dist.barrier()
        t.close()
        # the exchange inside the backward launch (2 launches per step), with 8-byte {number, value} words and with
        # bare 4-byte values (half the bytes on a link), each judged by the same criteria
        for mode2, tag in (("xgmi2", "2"), ("xgmi2c", "2c"), ("xgmi2p", "2p")):
            if not go_on(mode2):
                continue
            try:
                t2 = build_replica_trainer(mode2, device=dev, dataset=ds, batch=100, lr=0.001, seed=1, rank=args.rank,
                                           world=args.world, chunk=args.chunk, use_graph=True, timeout_ms=10000)
                q0 = t2.params.clone()
                t2.prepare((args.steps, args.steps))
                dist.barrier()
                t2.run_steps(args.steps)
                torch.cuda.synchronize(dev)
                dist.barrier()
                t0 = time.perf_counter()
                t2.run_steps(args.steps)
                torch.cuda.synchronize(dev)
                dt2 = time.perf_counter() - t0
                sums2 = [None] * args.world
                dist.all_gather_object(sums2, (t2.replica_checksum(), t2.healthy(), dt2,
                                               bool(torch.isfinite(t2.params).all()),
                                               float((t2.params - q0)[4:].abs().max()) > 0.0,
                                               t2.global_step_value() == 2 * args.steps))
                out["ok" + tag] = bool(all(s[1] and s[3] and s[4] and s[5] for s in sums2)
                                       and len({s[0] for s in sums2}) == 1)
                out["ms_per_step" + tag] = round(1000.0 * max(s[2] for s in sums2) / args.steps, 5)
                note(f"in-backward exchange ({mode2}) ok={out['ok' + tag]} {out['ms_per_step' + tag]} ms/step")
                dist.barrier()
                t2.close()
            except Exception as e:  # noqa: BLE001 — collective calls above keep the ranks in step even when one fails here
                out["ok" + tag] = False
                out["error" + tag] = f"{type(e).__name__}: {e}"
            print(json.dumps(out), flush=True)
        # the asynchronous mode over the same kind of mappings: one-sided pulls and lock-free pushes, with the push
        # inside the backward launch (one float atomic per element through the mappings: a launch less, 64-byte
        # requests) and with the push as a launch of its own (whole vectors); the faster form that kept every push
        # is the one reported
        forms = {}
        import os

        # (the push as a launch of its own is the older form: tried only on request, it costs the trial a stage)
        for fused_push in ((True, False) if os.environ.get("DTF_TRIAL_ASYNC_BOTH") == "1" else (True,)):
            if not go_on("async, push " + ("in the backward" if fused_push else "as a launch")):
                continue
            ta = build_replica_trainer("xgmi_async", device=dev, dataset=ds, batch=100, lr=0.001, seed=1,
                                       rank=args.rank, world=args.world, chunk=args.chunk, use_graph=True,
                                       fused_push=fused_push)
            loss0, _ = ta.evaluate(ds.x_epoch[:2000], ds.y_epoch[:2000])
            ta.prepare((args.steps, args.steps))
            dist.barrier()
            ta.run_steps(args.steps)
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            ta.run_steps(args.steps)
            torch.cuda.synchronize(dev)
            dta = time.perf_counter() - t0
            dist.barrier()                                    # every worker's pushes have landed
            gs = ta.global_step_value()
            loss1, _ = ta.evaluate(ds.x_epoch[:2000], ds.y_epoch[:2000])
            a_fin = bool(torch.isfinite(ta.params).all())
            times = [None] * args.world
            dist.all_gather_object(times, (dta, a_fin and gs == 2 * args.steps * args.world and loss1 <= loss0))
            forms[fused_push] = dict(async_ok=all(x[1] for x in times), async_global_step=gs,
                                     async_fused_push=fused_push,
                                     async_ms_per_step=round(1000.0 * max(x[0] for x in times) / args.steps, 5),
                                     async_loss=[round(loss0, 4), round(loss1, 4)])
            note(f"async (push {'inside the backward' if fused_push else 'as a launch'}) "
                 f"ok={forms[fused_push]['async_ok']} {forms[fused_push]['async_ms_per_step']} ms/step")
            dist.barrier()
            ta.close()
        if forms:
            good = [f for f in forms.values() if f["async_ok"]] or list(forms.values())
            out.update(min(good, key=lambda f: f["async_ms_per_step"]))
            out["async_ms_per_step_by_push"] = {("in_backward" if k else "own_launch"): v["async_ms_per_step"]
                                                for k, v in forms.items()}
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 — every failure is an answer: "do not use this path here"
        out["error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 3


if __name__ == "__main__":
    sys.exit(main())
