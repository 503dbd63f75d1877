set -o pipefail
timeout -k 10 200 python tfsingle.py --epochs 8 --logs_path /tmp/lg1 2>&1 | grep "Total Time" | tail -4 | tr '\n' ' '; echo
timeout -k 10 200 python tfsingle.py --epochs 8 --nosummaries 2>&1 | grep "Total Time" | tail -4 | tr '\n' ' '; echo
timeout -k 10 300 python bench/run_cluster.py --script tfdist_between.py --ps 1 --workers 1 --transport ipc --epochs 8 2>&1 | tail -1 | cut -c1-400
timeout -k 10 300 python bench/run_cluster.py --script tfdist_between_sync.py --ps 1 --workers 2 --transport ipc --epochs 8 2>&1 | tail -1 | cut -c1-500
