# One GPU session: the GPU test suite, the headline benchmark, a kernel trace of it.  Every GPU step has its own time
# limit and the script stops at the first failure.  Usage: gpurun --timeout 1200 -- 'bash bench/gpu_session.sh'
set -o pipefail
out=gpurun_out/session
mkdir -p $out
# the benchmark first: run right after the suite (dozens of processes just gone) it has come out 15 - 30 % low
timeout -k 10 300 python bench.py --gpus 1 > $out/bench_1gpu.json 2> $out/bench_1gpu.err || exit 1
python -c "import json; d=json.load(open('$out/bench_1gpu.json')); print(d['value'], d['ms_per_step'], d['sec_per_epoch'], d['vs_baseline'])"
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest_gpu.txt 2>&1; rc=$?; tail -4 $out/pytest_gpu.txt; [ $rc = 0 ] || { tail -80 $out/pytest_gpu.txt; exit 1; }
root=$PWD
cd /tmp && export TMPDIR=/tmp
export DTF_BENCH_DENSE=0        # the trace is of the headline workload: no dense-stack stage
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/bench.py --gpus 1 --steps 2200 --warmup 550 > $root/$out/trace.log 2>&1 || exit 1
cd $root && find $out/trace -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-160
