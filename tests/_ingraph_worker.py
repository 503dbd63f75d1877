"""Body of the in-graph engine test, in a process of its own (one engine per client process is how the engine is
used; a pytest process that has created dozens of streams and engines before is not)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402
from distributed_tensorflow_amd.parallel.towers import InGraphEngine  # noqa: E402


def main():
    towers, compact = int(sys.argv[1]), sys.argv[2] == "1"
    timeout_ms = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    rng = np.random.RandomState(0)
    images = rng.rand(600, 784).astype(np.float32)
    labels = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 600)]
    eng = InGraphEngine(["cuda:0"] * towers, images, labels, batch=100, lr=0.05, seed=1, chunk=4, compact=compact,
                        timeout_ms=timeout_ms)
    res = {}
    try:
        p = {k: v.detach().cpu().clone() for k, v in eng.towers[0].layout.mlp_views(eng.towers[0].params).items()}
        sets = [(tr.ds.x_epoch.cpu().clone(), tr.ds.y_epoch.cpu().clone()) for tr in eng.towers]
        eng.prepare((5,))
        eng.run_steps(5)                       # chunks of 4 + 1, inside the first 6-batch epoch
        res["healthy_5"], res["step_5"], res["identical_5"] = eng.healthy(), eng.global_step_value(), eng.replicas_identical()
        res["host_calls_per_chunk"] = eng.host_calls_per_chunk()
        for step in range(5):
            g = {k: torch.zeros_like(v) for k, v in p.items()}
            for xs, ys in sets:
                out = ref.mlp_grads(xs[100 * step:100 * step + 100], ys[100 * step:100 * step + 100], p["W1"], p["b1"],
                                    p["W2"], p["b2"])
                for k in g:
                    g[k] += out["d" + k] / towers
            for k in p:
                p[k] -= 0.05 * g[k]
        got = eng.towers[-1].layout.mlp_views(eng.towers[-1].params)
        res["oracle_error"] = max(float(((got[k].cpu() - p[k]).abs() / (2e-5 + 2e-4 * p[k].abs())).max()) for k in p)
        eng.run_steps(9)                       # crosses the epoch boundary: every tower reshuffles its own set
        res["healthy_14"], res["step_14"], res["identical_14"] = eng.healthy(), eng.global_step_value(), eng.replicas_identical()
        loss, acc = eng.evaluate(torch.as_tensor(images).cuda(), torch.as_tensor(labels).cuda())
        res["loss"], res["acc"] = loss, acc
    finally:
        eng.close()
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
