# PMC counters of one GEMM shape: bash bench/pmc_gemm.sh <MxNxK> <layout> <tag>   (one rocprofv3 run per counter set)
set -o pipefail
shape=${1:-4096x4096x4096}; layout=${2:-NT}; tag=${3:-$layout}
root=$PWD; out=$root/gpurun_out/gemm; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc1_$tag -- python3 $root/bench/gemm_once.py $shape $layout 0 3 > $out/pmc1_$tag.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $out/pmc2_$tag -- python3 $root/bench/gemm_once.py $shape $layout 0 3 > $out/pmc2_$tag.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $out/pmc3_$tag -- python3 $root/bench/gemm_once.py $shape $layout 0 3 > $out/pmc3_$tag.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr -d $out/pmc4_$tag -- python3 $root/bench/gemm_once.py $shape $layout 0 3 > $out/pmc4_$tag.log 2>&1 || echo "counter set 4 not collected"
cd $root && python bench/summarize_pmc.py $out/pmc1_$tag $out/pmc2_$tag $out/pmc3_$tag $out/pmc4_$tag | tee $out/pmc_summary_$tag.txt
