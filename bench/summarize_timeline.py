"""Per-phase host time of the Chrome traces written with DTF_TRACE=timeline DTF_TRACE_FILE=dir/trace_%p.json.
Usage: python bench/summarize_timeline.py dir/trace_*.json"""
import collections
import json
import sys

for path in sys.argv[1:]:
    doc = json.load(open(path))
    events = doc["traceEvents"] if isinstance(doc, dict) else doc
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for e in events:
        if e.get("ph") == "X":
            a = acc[e["name"]]
            a[0] += 1
            a[1] += e["dur"]
            a[2] = max(a[2], e["dur"])
    print(path)
    print("  %-22s %8s %11s %9s %9s" % ("span", "count", "total ms", "mean us", "max us"))
    for name, (n, tot, mx) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("  %-22s %8d %11.2f %9.1f %9.1f" % (name, n, tot / 1000.0, tot / n, mx))
