set -o pipefail
for o in 0 1; do echo "== order $o"; DTF_PIPE_ORDER=$o timeout -k 10 120 python bench/probe_kernels.py pipe 2>&1 | tail -9 || exit 1; done
for o in 1 0 1 0; do
  DTF_PIPE_ORDER=$o DTF_BENCH_ASYNC=0 timeout -k 10 120 python bench.py --gpus 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('order=$o', d['ms_per_step'], d['sec_per_epoch'])"
done
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 250 --timeout-method thread 2>&1 | tail -3
