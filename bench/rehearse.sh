#!/bin/bash
# bench/rehearse.sh OUTDIR : the benchmark as the round driver runs it (1 GPU), then its multi-rank flow with the ranks
# sharing this GPU (gloo rendezvous), 2 and 8 ranks; every run under its own time limit, stops at the first failure.
out=${1:-gpurun_out/rehearsal}
mkdir -p $out
t0=$(date +%s)
timeout -k 10 300 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_1gpu_20.json 2> $out/bench_1gpu_20.err || { tail -5 $out/bench_1gpu_20.err; exit 1; }
echo "1 GPU, 20 steps: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
timeout -k 10 300 python -u bench.py > $out/bench_1gpu.json 2> $out/bench_1gpu.err || { tail -5 $out/bench_1gpu.err; exit 1; }
echo "1 GPU, defaults: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
for n in "$@"; do
  [ "$n" = "$out" ] && continue
  DTF_BENCH_SHARE_GPU=1 timeout -k 10 420 python -u bench.py --gpus $n --steps 20 --warmup 5 > $out/bench_${n}ranks_one_gpu.json 2> $out/bench_${n}ranks_one_gpu.err || { tail -20 $out/bench_${n}ranks_one_gpu.err; exit 1; }
  echo "$n ranks sharing the GPU: $(( $(date +%s) - t0 )) s"; t0=$(date +%s)
done
