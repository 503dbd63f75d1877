#!/usr/bin/env python
"""Launch a whole localhost cluster of one of the between-graph scripts and report sec/epoch per worker.

    python bench/run_cluster.py --script tfdist_between.py --ps 1 --workers 2 --transport ipc --epochs 3

Every task is its own process (PS first), exactly like launching them by hand
(``distributed_tensorflow_amd/utils/local_cluster.py``).  Prints one JSON line with the steady-state epoch time (mean of
the epochs after the first), the first-epoch time and the final accuracy of every worker.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils.local_cluster import run_local_cluster  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--script", default="tfdist_between.py")
    ap.add_argument("--ps", type=int, default=1)
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--transport", default="tcp")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    result = run_local_cluster(a.script, a.ps, a.workers, a.transport, a.epochs, a.extra, a.timeout)
    rc = result.pop("returncode")
    print(json.dumps(result))
    return rc


if __name__ == "__main__":
    sys.exit(main())
