#!/bin/bash
# bench/pmc_lds_step.sh OUTDIR : LDS bank-conflict counters of the one-launch step, whole and with only its forward /
# only its backward workgroups (DTF_PIPE_PARTS; results of the partial runs are wrong, their counters are what is read)
out=${1:-gpurun_out/r3/pmc_lds}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DTF_BENCH_DENSE=0 DTF_BENCH_ASYNC=0 DTF_BENCH_PER_BATCH=0 DTF_BENCH_CLUSTERS=0
for parts in 3 1 2; do
  DTF_PIPE_PARTS=$parts timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/parts$parts -- python3 bench.py --gpus 1 --steps 550 --warmup 550 > $out/parts$parts.txt 2>&1 || exit 1
  python3 bench/summarize_pmc.py $out/parts$parts > $out/parts$parts.summary.txt 2>&1
  echo "== parts $parts"; grep "step_pipe" $out/parts$parts.summary.txt
done
