# Dense-layer kernels: GEMM against the library, training steps of stacks, kernel trace of a deeper tfsingle run, counters.
# Usage: gpurun --timeout 900 -- 'bash bench/dense_session.sh'
set -o pipefail
root=$PWD; out=$root/gpurun_out/dense; mkdir -p $out
timeout -k 10 200 python -u bench/gemm_bench.py --tiles 0 > $out/gemm_vs_library.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench/dense_bench.py > $out/dense_steps.txt 2>&1 || exit 1
timeout -k 10 200 python bench/dense_dp_bench.py > $out/dense_dp_1rank.json 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tfsingle.py --resident --hidden 512,256 --activation relu --lr 0.05 --epochs 3 --logs_path /tmp/lg_dense > $out/tfsingle_512_256.txt 2>&1 || exit 1
cd $root
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/tfsingle_512_256_kernel_stats.csv
bash bench/pmc_gemm.sh 4096x4096x4096 NN NN_final > /dev/null && bash bench/pmc_gemm.sh 4096x4096x4096 NT NT_final > /dev/null || exit 1
cp gpurun_out/gemm/pmc_summary_NN_final.txt gpurun_out/gemm/pmc_summary_NT_final.txt $out/
head -12 $out/tfsingle_512_256_kernel_stats.csv | cut -c1-170
