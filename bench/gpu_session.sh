set -o pipefail
mkdir -p gpurun_out/r2
DTF_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r2/share2.json 2> gpurun_out/r2/share2.err
echo "share2 rc=$?"; cat gpurun_out/r2/share2.json
DTF_BENCH_SHARE_GPU=1 timeout -k 10 600 python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/r2/share8.json 2> gpurun_out/r2/share8.err
echo "share8 rc=$?"; cat gpurun_out/r2/share8.json; tail -5 gpurun_out/r2/share8.err
