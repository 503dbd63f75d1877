"""Summarise a rocprofv3 run that was taken with ``DTF_TRACE=roctx ... --kernel-trace --marker-trace``.

rocprofv3 writes a rocpd SQLite database (``<dir>/<host>/<pid>_results.db``).  This prints, per
framework phase (the message of each marker range), the number of ranges and their host time, and
next to it the kernels by total device time — the two halves of "where does a step go".

    python bench/summarize_markers.py gpurun_out/marker/vm/1116_results.db
"""
import json
import sqlite3
import sys


def main(path: str) -> None:
    db = sqlite3.connect(path)
    phases = {}
    for ext, start, end in db.execute("select extdata, start, end from regions where category like 'MARKER%'"):
        try:
            name = json.loads(ext).get("message", "?")
        except (TypeError, ValueError):
            name = "?"
        p = phases.setdefault(name, [0, 0, 0])
        p[0] += 1
        p[1] += end - start
        p[2] = max(p[2], end - start)
    print("marker ranges (host time)")
    print("%-22s %8s %12s %10s %10s" % ("phase", "count", "total ms", "mean us", "max us"))
    for name, (n, tot, mx) in sorted(phases.items(), key=lambda kv: -kv[1][1]):
        print("%-22s %8d %12.3f %10.1f %10.1f" % (name, n, tot / 1e6, tot / 1e3 / n, mx / 1e3))
    print()
    print("kernels (device time)")
    print("%-70s %8s %12s %10s %7s" % ("kernel", "calls", "total us", "mean us", "%"))
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage "
                                                   "from top_kernels limit 12"):
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print("%-70s %8d %12.1f %10.2f %7.2f" % (short[:70], calls, total, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1])
