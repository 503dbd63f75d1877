#!/bin/bash
# Headline benchmark for several hipGraph chunk sizes: chunk, ms/step, samples/s, sec/epoch.
#   bash bench/chunk_sweep.sh
for c in 50 110 275 550; do
  DTF_BENCH_ASYNC=0 timeout -k 10 120 python bench.py --chunk $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['ms_per_step'], d['value'], d['sec_per_epoch'])" $c || exit 1
done
