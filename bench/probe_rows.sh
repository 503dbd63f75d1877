#!/bin/bash
# bench/probe_rows.sh ROWS STEPS : the peer-exchange trial with 2 ranks on this GPU and a resident set of ROWS rows
rows=$1; steps=${2:-150}; port=$((29700 + RANDOM % 200))
for r in 0 1; do
  timeout -k 10 120 python -u -m distributed_tensorflow_amd.parallel.xgmi_probe --rank $r --world 2 --device 0 --port $port --rows $rows --steps $steps --chunk 50 > gpurun_out/r3/probe_rows_${rows}_$r.txt 2>&1 &
done
wait
tail -1 gpurun_out/r3/probe_rows_${rows}_0.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print($rows, {k:d.get(k) for k in ('ms_per_step','ms_per_step2','ms_per_step2c','ms_per_step2p','async_ms_per_step')})"
