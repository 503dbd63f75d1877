set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "gram or dataset or drawing" 2>&1 | tail -3 || exit 1
for i in 1 2; do
DTF_BENCH_ASYNC=0 timeout -k 10 200 python bench.py --gpus 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sec_per_epoch'], d['setup_s']['epoch_refresh_ms'])" || exit 1
done
