set -o pipefail
mkdir -p gpurun_out/r2/tl
rm -f gpurun_out/r2/tl/*
DTF_TRACE=timeline DTF_TRACE_FILE=$PWD/gpurun_out/r2/tl/trace_%p.json timeout -k 10 300 python bench/run_cluster.py --script tfdist_between_sync.py --ps 1 --workers 2 --transport ipc --epochs 4
python bench/summarize_timeline.py gpurun_out/r2/tl/trace_*.json
