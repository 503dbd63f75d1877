#!/bin/bash
# bench/ab_lib.sh : bench.py's single-GPU step with the in-tree library and with every variant library in
# bench/micro/bin/libdtf_hip_*.so, alternating, three times each
export DTF_BENCH_DENSE=0 DTF_BENCH_ASYNC=0 DTF_BENCH_PER_BATCH=0 DTF_BENCH_CLUSTERS=0
for i in 1 2 3; do
  for lib in "" bench/micro/bin/libdtf_hip_*.so; do
    if [ -n "$lib" ]; then [ -f "$lib" ] || continue; export DTF_HIP_LIB=$PWD/$lib; else unset DTF_HIP_LIB; fi
    timeout -k 10 120 python bench.py --gpus 1 --steps 5500 --warmup 550 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readlines()[-1]);print('${lib:-in-tree}',d['ms_per_step'],d.get('ms_per_step_device'),d.get('sec_per_epoch'))" || exit 1
  done
done
