#!/bin/bash
# bench/pmc_xg_modes.sh OUT : counters of the two-launch steps that every multi-replica mode runs (one replica here):
# xgmi2 (forward that reads W1 transposed + backward with the exchange) and xgmi_async (forward that reads the
# variables from the shards + backward that pushes)
out=${1:-gpurun_out/pmc_xg}
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
for mode in xgmi2 xgmi_async; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $out/a_$mode -- python3 bench/xg_one_mode.py $mode 1100 > $out/a_$mode.txt 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $out/b_$mode -- python3 bench/xg_one_mode.py $mode 1100 > $out/b_$mode.txt 2>&1 || exit 1
  python3 bench/summarize_pmc.py $out/a_$mode $out/b_$mode > $out/$mode.summary.txt 2>&1
  echo "== $mode"; grep "fwd_tile8\|bwd_tile16" $out/$mode.summary.txt
done
