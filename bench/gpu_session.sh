set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/r2/pmc_gram1 -- python3 $R/bench/gram_once.py > $R/gpurun_out/r2/pmc_gram1.log 2>&1
echo "pmc1 rc=$?"
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/r2/pmc_gram2 -- python3 $R/bench/gram_once.py > $R/gpurun_out/r2/pmc_gram2.log 2>&1
echo "pmc2 rc=$?"
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/r2/pmc_gram3 -- python3 $R/bench/gram_once.py 2 > $R/gpurun_out/r2/pmc_gram3.log 2>&1
echo "pmc3 rc=$?"
cd $R
python bench/summarize_pmc.py gpurun_out/r2/pmc_gram1 gpurun_out/r2/pmc_gram2
echo "--- products only"
python bench/summarize_pmc.py gpurun_out/r2/pmc_gram3
