set -o pipefail
root=$PWD; out=$root/gpurun_out/gemm; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc1 -- python3 $root/bench/gemm_once.py 4096x4096x4096 NT 128 3 > $out/pmc1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $out/pmc2 -- python3 $root/bench/gemm_once.py 4096x4096x4096 NT 128 3 > $out/pmc2.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $out/pmc3 -- python3 $root/bench/gemm_once.py 4096x4096x4096 NT 128 3 > $out/pmc3.log 2>&1 || exit 1
cd $root && python bench/summarize_pmc.py $out/pmc1 $out/pmc2 $out/pmc3 | tee $out/pmc_summary.txt
