# A deeper stack through real localhost clusters on the GPU (every task its own process, all sharing the GPU):
# asynchronous and synchronous parameter servers, tcp and ipc transports.  Usage: gpurun --timeout 600 -- 'bash bench/dense_cluster.sh'
set -o pipefail
out=gpurun_out/dense_cluster; mkdir -p $out
for cfg in "tfdist_between.py 1 2 ipc" "tfdist_between.py 2 2 tcp" "tfdist_between_sync.py 1 2 ipc" "tfdist_between_sync.py 2 2 tcp"; do
  set -- $cfg
  timeout -k 10 240 python bench/run_cluster.py --script $1 --ps $2 --workers $3 --transport $4 --epochs 3 --timeout 200 -- \
      --hidden=256,128 --activation=relu --lr=0.05 >> $out/results.jsonl 2>> $out/err.txt || { tail -30 $out/err.txt; exit 1; }
  tail -1 $out/results.jsonl | cut -c1-700
done
# the device-resident asynchronous worker of the same stack (hipGraph chunks of pull -> step -> push against the mapped shards)
for cfg in "1 1" "1 2" "2 2"; do
  set -- $cfg
  timeout -k 10 240 python bench/run_cluster.py --script tfdist_between.py --ps $1 --workers $2 --transport ipc --epochs 5 --timeout 200 -- \
      --hidden=256,128 --activation=relu --lr=0.05 --resident >> $out/results_resident.jsonl 2>> $out/err.txt || { tail -30 $out/err.txt; exit 1; }
  tail -1 $out/results_resident.jsonl | cut -c1-700
done
# synchronous replicas of the same stack, device-resident (gradient launches + one exchange kernel per step)
for cfg in "1 2" "2 2" "1 4"; do
  set -- $cfg
  timeout -k 10 240 python bench/run_cluster.py --script tfdist_between_sync.py --ps $1 --workers $2 --transport ipc --epochs 5 --timeout 200 -- \
      --hidden=256,128 --activation=relu --lr=0.05 --resident >> $out/results_resident.jsonl 2>> $out/err.txt || { tail -30 $out/err.txt; exit 1; }
  tail -1 $out/results_resident.jsonl | cut -c1-900
done
