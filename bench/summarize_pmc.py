"""Average rocprofv3 --pmc counter values per kernel from a counter_collection.csv (prints a small table)."""
import collections
import csv
import glob
import sys

files = [f for d in sys.argv[1:] for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = "step_pipe" if "step_pipe" in name else "fwd_tile8" if "fwd_tile8" in name else \
            "bwd_push" if ("bwd_tile16_xg" in name and "-1" in name) else \
            "bwd_tile16_xg" if "bwd_tile16_xg" in name else \
            "bwd_tile16" if "bwd_tile16" in name else "gather_rows" if "gather_rows" in name else \
            "range_pull" if "range_pull" in name else "shard_push" if "shard_push" in name else \
            "gemm_f32" if "gemm_f32" in name else "gather_gram" if "gather_gram" in name else "eval_fused" if "eval_fused" in name else "sum_apply" if "sum_apply" in name else None
        if short is None:
            continue
        k = (short, r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
for (kern, ctr), (tot, n) in sorted(acc.items()):
    print("%-12s %-28s launches %6d   mean per launch %14.1f" % (kern, ctr, n, tot / n))
