set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -6 || exit 1
DTF_BENCH_ASYNC=0 timeout -k 10 200 python bench.py --gpus 1 > gpurun_out/r2b/bench_1gpu_quick.json 2>gpurun_out/r2b/bench_1gpu_quick.err || exit 1
cat gpurun_out/r2b/bench_1gpu_quick.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2b/eval_trace -- python3 $GRAFT_REPO_ROOT/bench/eval_once.py > $GRAFT_REPO_ROOT/gpurun_out/r2b/eval_trace.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && find gpurun_out/r2b/eval_trace -name "*kernel_stats.csv" | head -1 | xargs head -8
