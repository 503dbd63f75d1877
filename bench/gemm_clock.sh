set -o pipefail
root=$PWD; out=$root/gpurun_out/gemm; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for which in ours lib; do
  if [ $which = lib ]; then export DTF_GEMM_ONCE_LIBRARY=1; fi
  timeout -s KILL 90 rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES -d $out/clk_$which -- python3 $root/bench/gemm_once.py 4096x4096x4096 NT 0 4 > $out/clk_$which.log 2>&1 || exit 1
done
cd $root
python - <<'PY'
import csv, glob, collections
for which in ("ours", "lib"):
    d = f"gpurun_out/gemm/clk_{which}"
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:60])
    acc = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for did, c in acc.items():
        ns, name = dur.get(did, (0, "?"))
        if ns > 500000:
            print(which, name, "ns", ns, {k: int(v) for k, v in c.items()}, "GUI_ACTIVE/ns = GHz:", round(c.get("GRBM_GUI_ACTIVE", 0) / ns, 3))
PY
