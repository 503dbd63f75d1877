#!/bin/bash
# bench/sync_board_ab.sh OUT : sec/epoch of the synchronous per-batch tier (1 ps + 2 workers, ipc shards) with the barrier
# on the shared-memory board and on the task server's sockets, and the asynchronous tier for scale.
out=${1:-gpurun_out/r3/sync_board_ab.jsonl}
mkdir -p $(dirname $out); : > $out
run() { echo "# $*" >> $out; timeout -k 10 280 env "$1" python -u bench/run_cluster.py --script $2 --ps $3 --workers $4 --transport ipc --epochs 4 >> $out 2>&1 || exit 1; }
run DTF_SYNC_BOARD=1 tfdist_between_sync.py 1 2
run DTF_SYNC_BOARD=0 tfdist_between_sync.py 1 2
run DTF_SYNC_BOARD=1 tfdist_between_sync.py 2 2
run DTF_SYNC_BOARD=1 tfdist_between.py 1 2
python3 - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("#"): print(line.strip()); continue
    if not line.startswith("{"): continue
    d = json.loads(line)
    print({k: (round(v["steady_epoch_s"], 4), v.get("final_accuracy"), v.get("last_step")) for k, v in d["tasks"].items() if "steady_epoch_s" in v})
PY
