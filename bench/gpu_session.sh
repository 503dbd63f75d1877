set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
DTF_BENCH_ASYNC=0 timeout -k 10 200 python bench.py --gpus 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sec_per_epoch'], d['test_accuracy_after_run'])" || exit 1
timeout -k 10 60 python bench/probe_kernels.py bwd 2>&1 | grep -v amdgpu
timeout -k 10 60 python bench/probe_kernels.py fwd 2>&1 | grep -v amdgpu | head -9
