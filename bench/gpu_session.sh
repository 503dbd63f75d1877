set -o pipefail
mkdir -p gpurun_out/r2d
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -6 || exit 1
DTF_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 > gpurun_out/r2d/rehearsal2.out 2> gpurun_out/r2d/rehearsal2.err || { tail -20 gpurun_out/r2d/rehearsal2.err; exit 1; }
grep '^{"metric"' gpurun_out/r2d/rehearsal2.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('mode_ms_per_step'), d.get('ranks_seen'), d.get('sec_per_epoch'), list(d.get('async_modes',{}).keys()), d.get('ingraph',{}).get('ms_per_step') if isinstance(d.get('ingraph'),dict) else d.get('ingraph'))"
