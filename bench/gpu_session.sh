set -o pipefail
mkdir -p gpurun_out/r2c
timeout -k 10 300 python bench.py --gpus 1 > gpurun_out/r2c/bench_1gpu.json 2> gpurun_out/r2c/bench_1gpu.err || exit 1
cat gpurun_out/r2c/bench_1gpu.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2c/bench_1gpu_s20.json 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2c/trace -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 2200 --warmup 550 > $GRAFT_REPO_ROOT/gpurun_out/r2c/trace.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && find gpurun_out/r2c/trace -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-180
