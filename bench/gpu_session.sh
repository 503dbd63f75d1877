set -o pipefail
for na in 4 5 6; do
 echo "== NA $na"
 DTF_HIP_LIB=$PWD/bench/micro/bin/libdtf_hip_na$na.so DTF_PROBE_NA=$na timeout -k 10 120 python bench/probe_kernels.py pipe 2>&1 | tail -5 || exit 1
 for i in 1 2; do
 DTF_HIP_LIB=$PWD/bench/micro/bin/libdtf_hip_na$na.so DTF_BENCH_ASYNC=0 timeout -k 10 120 python bench.py --gpus 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('na=$na', d['ms_per_step'], d['sec_per_epoch'], d['test_accuracy_after_run'])" || exit 1
 done
done
DTF_BENCH_ASYNC=0 timeout -k 10 120 python bench.py --gpus 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('na=3', d['ms_per_step'], d['sec_per_epoch'])"
