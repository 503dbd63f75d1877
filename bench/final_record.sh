#!/bin/bash
# bench/final_record.sh OUT : the records kept under profiles/ at the end of a round: bench.py with its defaults and
# with the round driver's arguments, and a kernel trace of a shorter run
out=${1:-gpurun_out/final}
mkdir -p $out
timeout -k 10 300 python bench.py > $out/bench_1gpu.json 2> $out/bench_1gpu.err || exit 1
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_1gpu_20steps.json 2> $out/bench_1gpu_20steps.err || exit 1
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
# (the stages that run the scripts as child processes are left out of the traced run: every child would leave a trace of its own)
DTF_BENCH_PER_BATCH=0 DTF_BENCH_CLUSTERS=0 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gpus 1 --steps 2200 --warmup 550 > $out/trace.log 2>&1 || exit 1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
head -12 $out/kernel_stats.csv
