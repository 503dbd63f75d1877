#!/bin/bash
# bench/pmc_final.sh OUT : three counter passes (8 SQ, 8 SQ, 4 TCC) over bench.py --gpus 1 --steps 1100 --warmup 550,
# averaged per launch of step_pipe_kernel, gather_gram_kernel and eval_fused_kernel (bench/summarize_pmc.py)
out=${1:-gpurun_out/pmc_final}
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
export DTF_BENCH_DENSE=0 DTF_BENCH_ASYNC=0 DTF_BENCH_PER_BATCH=0 DTF_BENCH_CLUSTERS=0
run() { timeout -s KILL 150 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $out/$1 -- python3 bench.py --gpus 1 --steps 1100 --warmup 550 > $out/$1.txt 2>&1 || exit 1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
run b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
run c "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
python3 bench/summarize_pmc.py $out/a $out/b $out/c > $out/summary.txt 2>&1
python3 - "$out" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
f = glob.glob(out + "/a/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/summary.txt", "a") as g:
    g.write("\nkernel durations in the first pass (ns, mean over launches; counters serialise the launches):\n")
    for k, v in acc.items():
        if any(s in k for s in ("step_pipe", "gather_gram", "eval_fused")):
            g.write(f"  {k[:70]:70s} launches {len(v):5d} mean {sum(v) / len(v):10.0f}\n")
PY
grep "step_pipe\|gather_gram\|eval_fused\|durations" $out/summary.txt
