set -o pipefail
for i in 1 2 3; do
DTF_BENCH_ASYNC=0 timeout -k 10 200 python bench.py --gpus 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sec_per_epoch'], d['setup_s'])" || exit 1
done
timeout -k 10 300 python tfsingle.py --resident --epochs 5 --nosummaries 2>&1 | grep "Total Time\|Test-Acc" || exit 1
timeout -k 10 300 python tfsingle.py --resident --epochs 3 --logs_path /tmp/lg 2>&1 | grep "Total Time\|Test-Acc" || exit 1
timeout -k 10 600 python -u -m pytest tests/test_session_gpu.py tests/test_convergence_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
