#!/bin/bash
# dev: effective clock and matrix-pipe busy of the GEMM under ablation (EXCEL_DEV build): GRBM_GUI_ACTIVE/8 = kernel cycles, with the kernel trace's
# duration -> GHz; SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles) = busy share.  usage: bash tools_dev/gemm_clock.sh "0 2"
export TMPDIR=/tmp
REPO=$PWD
for d in $1; do
  OUT=$REPO/gpurun_out/gclk_$d; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && EXCEL_BF_DBG=$d rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -o g -- python $REPO/tools_dev/gemm_bench.py 25120 768 3072 20 > /dev/null 2> $OUT/err.txt)
  python - <<PY
import csv, glob, collections
cc = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if "gemm_bf16x3_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt)) if "gemm_bf16x3_kernel" in r["Kernel_Name"]]
cyc = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
busy = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * cyc)
d = sorted(dur)[len(dur)//2]
print("dbg=$d: median %.1f us, %.0f kcycles -> %.2f GHz, mfma busy %.3f (of 256 CUs; x 256/237 on the CUs that have a tile)" % (d / 1e3, cyc / 1e3, cyc / d, busy))
PY
done
