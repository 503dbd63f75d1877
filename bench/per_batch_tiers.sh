#!/bin/bash
# bench/per_batch_tiers.sh OUT : sec/epoch of the per-batch tiers on one GPU: tfsingle.py, the asynchronous and the
# synchronous between-graph scripts (1 ps + 2 workers, ipc shards)
out=${1:-gpurun_out/per_batch_tiers.jsonl}
mkdir -p $(dirname $out); : > $out
echo "# tfsingle.py --epochs 4" >> $out
DTF_EPOCH_TIMES_FILE=/tmp/dtf_pb_single.json timeout -k 10 120 python -u tfsingle.py --epochs 4 --logs_path /tmp/dtf_pb_logs > /dev/null 2>&1 || exit 1; cat /tmp/dtf_pb_single.json >> $out; echo >> $out
run() { echo "# $*" >> $out; timeout -k 10 280 python -u bench/run_cluster.py --script $1 --ps $2 --workers $3 --transport ipc --epochs 4 >> $out 2>&1 || exit 1; }
run tfdist_between.py 1 2
run tfdist_between_sync.py 1 2
run tfdist_between_sync.py 2 2
python3 - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if not line.startswith("{"): print(line.strip()); continue
    d = json.loads(line)
    if "epoch_times" in d:
        print("epochs", [round(t, 4) for t in d["epoch_times"]]); continue
    print({k: (round(v["steady_epoch_s"], 4), v.get("final_accuracy"), v.get("last_step")) for k, v in d["tasks"].items() if "steady_epoch_s" in v})
PY
