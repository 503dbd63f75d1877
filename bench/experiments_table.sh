#!/bin/bash
# bench/experiments_table.sh OUT : sec/epoch of every recipe of EXPERIMENTS.md that one GPU can run -- the scripts as
# separate tasks (bench/run_cluster.py), transports tcp / ipc, per batch and --resident -- one JSON line per recipe,
# then a table of the slowest worker's steady epoch (mean of the epochs after the first) and first epoch.
out=${1:-gpurun_out/experiments_table.jsonl}
mkdir -p $(dirname $out); : > $out
run() {   # label script ps workers transport [extra flags of the script...]
  local label=$1 script=$2 ps=$3 w=$4 tr=$5; shift 5
  echo "# $label" >> $out
  echo "[experiments] $label" >&2
  timeout -k 10 280 python -u bench/run_cluster.py --script $script --ps $ps --workers $w --transport $tr --epochs 4 -- "$@" >> $out 2>> $out.err || exit 1
}
run "1 async 1ps+1w tcp"           tfdist_between.py      1 1 tcp
run "1 async 1ps+1w ipc"           tfdist_between.py      1 1 ipc
run "1 async 1ps+1w ipc resident"  tfdist_between.py      1 1 ipc --resident
run "2 async 1ps+2w tcp"           tfdist_between.py      1 2 tcp
run "2 async 1ps+2w ipc"           tfdist_between.py      1 2 ipc
run "2 async 1ps+2w ipc resident"  tfdist_between.py      1 2 ipc --resident
run "4 sync 1ps+2w tcp"            tfdist_between_sync.py 1 2 tcp
run "4 sync 1ps+2w ipc"            tfdist_between_sync.py 1 2 ipc
run "4 sync 1ps+2w ipc resident"   tfdist_between_sync.py 1 2 ipc --resident
run "5 async 2ps+2w tcp"           tfdist_between.py      2 2 tcp
run "5 async 2ps+2w ipc"           tfdist_between.py      2 2 ipc
run "5 async 2ps+2w ipc resident"  tfdist_between.py      2 2 ipc --resident
run "6 sync 2ps+2w tcp"            tfdist_between_sync.py 2 2 tcp
run "6 sync 2ps+2w ipc"            tfdist_between_sync.py 2 2 ipc
run "6 sync 2ps+2w ipc resident"   tfdist_between_sync.py 2 2 ipc --resident
run "8 async 1ps+3w tcp"           tfdist_between.py      1 3 tcp
run "8 async 1ps+3w ipc"           tfdist_between.py      1 3 ipc
run "8 async 1ps+3w ipc resident"  tfdist_between.py      1 3 ipc --resident
python3 - "$out" <<'PY'
import json, sys
label = None
for line in open(sys.argv[1]):
    if line.startswith("# "):
        label = line[2:].strip(); continue
    if not line.startswith("{"): continue
    d = json.loads(line)
    w = [v for k, v in d["tasks"].items() if k.startswith("worker") and "steady_epoch_s" in v]
    print(f"{label:32s} steady {max(v['steady_epoch_s'] for v in w):.4f} s  fastest {max(min(v['epoch_times'][1:]) for v in w):.4f} s  "
          f"first {max(v['epoch_times'][0] for v in w):.3f} s  accuracy {[v.get('final_accuracy') for v in w]}  steps {[v.get('last_step') for v in w]}")
PY
