#!/bin/bash
# round 5: where the four-wave GEMM's k-loop loses against its own MFMA stream - cycles or clock?  PMC passes over the ablation arms
# (dev library, EXCEL_W4_DBG) on one layer shape.  usage: bash tools_dev/r05_w4_pmc.sh "0 8 10 15" ["M N K"]
export TMPDIR=/tmp
REPO=$PWD
SHAPE=${2:-"25120 2304 768"}
for d in $1; do
  OUT=$REPO/gpurun_out/w4pmc_$d; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && EXCEL_AB_LIB=$REPO/tools_dev/ab/dev.so EXCEL_W4_DBG=$d rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT -o a -- python $REPO/tools_dev/gemm_bench.py $SHAPE 10 bf16x3_split > /dev/null 2> $OUT/err_a.txt)
  (cd /tmp && EXCEL_AB_LIB=$REPO/tools_dev/ab/dev.so EXCEL_W4_DBG=$d rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT -o b -- python $REPO/tools_dev/gemm_bench.py $SHAPE 10 bf16x3_split > /dev/null 2> $OUT/err_b.txt)
  python - <<PY
import csv, glob, collections
def fold(tag):
    cc = [f for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True) if "/%s_" % tag in f or f.endswith("%s_counter_collection.csv" % tag)][0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if "gemm_w4" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
a, b = fold("a"), fold("b")
kt = [f for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True) if "a_kernel" in f][0]
dur = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt)) if "gemm_w4" in r["Kernel_Name"])
d = dur[len(dur)//2]
cyc = a["GRBM_GUI_ACTIVE"] / 8
wc = a["SQ_WAVE_CYCLES"]
print("dbg=$d: %.1f us  %.0f kcyc  %.2f GHz | mfma_busy %.3f | wave: wait %.2f stall %.2f (lds-issue %.2f) active %.2f | lds_busy %.3f confl %.3f | per WG-step: lds insts %.0f mfma %.0f vmem %.0f salu %.0f valu %.0f | totals: wave_cyc %.3g valu %.3g" % (
    d / 1e3, cyc / 1e3, cyc / d, a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_WAIT_INST_LDS"] / wc,
    a["SQ_ACTIVE_INST_ANY"] / wc, b["SQ_LDS_IDX_ACTIVE"] / (256 * b["GRBM_GUI_ACTIVE"] / 8), b["SQ_LDS_BANK_CONFLICT"] / max(b["SQ_LDS_IDX_ACTIVE"], 1),
    b["SQ_INSTS_LDS"] / 711 / 24 / 4, b["SQ_INSTS_MFMA"] / 711 / 24 / 4, b["SQ_INSTS_VMEM"] / 711 / 24 / 4, b["SQ_INSTS_SALU"] / 711 / 24 / 4, b["SQ_INSTS_VALU"] / 711 / 24 / 4, wc, b["SQ_INSTS_VALU"]))
PY
  find $OUT -name "*.csv" -delete
done
