set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_convergence_gpu.py -m gpu -x -q --timeout 250 --timeout-method thread 2>&1 | tail -4
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
for i in 1 2 3; do DTF_BENCH_ASYNC=0 timeout -k 10 120 python bench.py --gpus 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['sec_per_epoch'], d['setup_s'])"; done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/r2/pmc_pipe1 -- python3 $R/bench.py --gpus 1 --steps 1100 --warmup 550 > $R/gpurun_out/r2/pmc_pipe1.log 2>&1
echo "pmc rc=$?"
cd $R && python bench/summarize_pmc.py gpurun_out/r2/pmc_pipe1 | grep -E "step_pipe|gather_gram"
