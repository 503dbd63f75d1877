set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/final
mkdir -p $O
cd $R
timeout -k 10 200 python bench.py --gpus 1 > $O/bench_1gpu.json 2> $O/bench_1gpu.err; echo "bench1 rc=$?"
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_1gpu_driver_args.json 2>> $O/bench_1gpu.err; echo "bench1 short rc=$?"
DTF_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 600 --warmup 100 > $O/bench_rehearsal_2ranks_one_gpu.json 2> $O/share2.err; echo "share2 rc=$?"
DTF_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 8 --steps 20 --warmup 5 > $O/bench_rehearsal_8ranks_one_gpu.json 2> $O/share8.err; echo "share8 rc=$?"
DTF_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --mode ingraph --steps 1100 --warmup 100 > $O/bench_ingraph_2towers_one_gpu.json 2> $O/ingraph.err; echo "ingraph rc=$?"
for cfg in "tfdist_between.py 1 1 tcp" "tfdist_between.py 1 1 ipc" "tfdist_between.py 1 2 ipc" "tfdist_between.py 2 2 tcp" "tfdist_between.py 2 2 ipc" "tfdist_between.py 1 3 tcp" "tfdist_between.py 1 3 ipc" "tfdist_between_sync.py 1 2 tcp" "tfdist_between_sync.py 1 2 ipc" "tfdist_between_sync.py 2 2 ipc"; do
  set -- $cfg
  timeout -k 10 300 python bench/run_cluster.py --script $1 --ps $2 --workers $3 --transport $4 --epochs 6 >> $O/cluster_per_batch.jsonl 2>> $O/cluster.err
  echo "cluster $cfg rc=$?"
done
for cfg in "tfdist_between.py 1 1" "tfdist_between.py 1 2" "tfdist_between.py 2 2" "tfdist_between.py 1 3" "tfdist_between_sync.py 1 2" "tfdist_between_sync.py 2 2" "tfdist_between_sync.py 1 4"; do
  set -- $cfg
  timeout -k 10 300 python bench/run_cluster.py --script $1 --ps $2 --workers $3 --transport ipc --epochs 21 -- --resident >> $O/cluster_resident.jsonl 2>> $O/cluster.err
  echo "resident $cfg rc=$?"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench1 -- python3 $R/bench.py --gpus 1 --steps 2200 --warmup 550 > $O/kt_bench1.log 2>&1; echo "kt rc=$?"
f=$(find $O/kt_bench1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_1gpu_kernel_stats.csv && head -12 "$f"
