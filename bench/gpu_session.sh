set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DTF_BENCH_ASYNC=0
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/pmcf1 -- python3 $R/bench.py --gpus 1 --steps 1100 --warmup 550 > $R/gpurun_out/pmcf1.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcf2 -- python3 $R/bench.py --gpus 1 --steps 1100 --warmup 550 > $R/gpurun_out/pmcf2.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/pmcf3 -- python3 $R/bench.py --gpus 1 --steps 1100 --warmup 550 > $R/gpurun_out/pmcf3.log 2>&1 || exit 1
cd $R && python bench/summarize_pmc.py gpurun_out/pmcf1 gpurun_out/pmcf2 gpurun_out/pmcf3
