#!/bin/bash
# dev: fabric bytes per launch of the bf16x3 GEMM per layer shape (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) next to the
# bytes the shape must move at least (A + W read once, C written once) -> profiles/r04_gemm_traffic.json   usage: bash tools_dev/gemm_traffic.sh
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out/gemm_traffic; rm -rf $OUT; mkdir -p $OUT
declare -A SH=( [qkv]="25120 2304 768 bf16x3_split" [proj]="25120 768 768 bf16x3" [fc1]="25120 3072 768 bf16x3_split" [fc2]="25120 768 3072 bf16x3" )
for name in qkv proj fc1 fc2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${name}_$ctr -o g -- python $REPO/tools_dev/gemm_bench.py ${SH[$name]% *} 10 ${SH[$name]##* } > $OUT/${name}_$ctr.txt 2> $OUT/${name}_$ctr.err)
  done
done
python - <<PY
import csv, glob, json
shapes = {"qkv": (25120, 2304, 768, 1), "proj": (25120, 768, 768, 0), "fc1": (25120, 3072, 768, 1), "fc2": (25120, 768, 3072, 0)}
out = {"_note": "bf16x3 GEMM alone (tools_dev/gemm_bench.py, 13 launches back to back on L2-cold operands of 77-309 MB): fabric bytes per launch = 2 x FETCH_SIZE (gfx950 correction) and WRITE_SIZE, against the algorithmic minimum A + W once, C once.  qkv / fc1 write the split output (same bytes as fp32).  W (2.4-9.4 MB) is read by all 8 XCDs; an A row tile lives in one XCD (xcd-aware tile order)."}
for name, (M, N, K, split) in shapes.items():
    rec = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (name, ctr), recursive=True)[0]
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_bf16x3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        rec[ctr] = sum(v) / len(v) * 1024
    a, w, c = M * K * 4, N * K * 4, M * N * 4
    rec = {"M": M, "N": N, "K": K, "fetch_bytes_per_launch": int(2 * rec["FETCH_SIZE"]), "write_bytes_per_launch": int(rec["WRITE_SIZE"]),
           "algorithmic_read_bytes": a + w, "algorithmic_write_bytes": c,
           "read_over_algorithmic": round(2 * rec["FETCH_SIZE"] / (a + w), 3), "write_over_algorithmic": round(rec["WRITE_SIZE"] / c, 3),
           "read_if_W_once_per_xcd": a + 8 * w, "time": open("$OUT/%s_FETCH_SIZE.txt" % name).read().strip()}
    out[name] = rec
print(json.dumps(out, indent=1))
PY
