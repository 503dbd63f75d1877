set -o pipefail
mkdir -p gpurun_out/r2f
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4 || exit 1
timeout -k 10 300 python bench.py --gpus 1 > gpurun_out/r2f/bench_1gpu.json 2> gpurun_out/r2f/bench_1gpu.err || exit 1
python -c "import json; d=json.load(open('gpurun_out/r2f/bench_1gpu.json')); print(d['value'], d['ms_per_step'], d['sec_per_epoch'], d['vs_baseline'], d['async_mode']['value'])"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2f/bench_1gpu_s20.json 2>/dev/null || exit 1
python -c "import json; d=json.load(open('gpurun_out/r2f/bench_1gpu_s20.json')); print(d['value'], d['ms_per_step'], d['sec_per_epoch'])"
python -c "from __graft_entry__ import smoke; smoke(); print('smoke ok')"
