set -o pipefail
for i in 1 2 3; do
DTF_BENCH_ASYNC=0 timeout -k 10 200 python bench.py --gpus 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sec_per_epoch'], d['test_accuracy_after_run'])" || exit 1
done
