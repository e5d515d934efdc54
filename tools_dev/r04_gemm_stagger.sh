#!/bin/bash
# round-4: GEMM epilogue stagger (dev library): EXCEL_BF_FIRST = short tiles dispatched first per XCD, EXCEL_BF_MIX1 = mixed heights on one-round shapes
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04i}; mkdir -p $OUT
export TMPDIR=/tmp
export EXCEL_AB_LIB=tools_dev/ab/dev.so
run_shapes() {
  for S in "25120 2304 768 bf16x3_split" "25120 768 768 bf16x3" "25120 3072 768 bf16x3_split" "25120 768 3072 bf16x3"; do
    set -- $S
    timeout 120 python tools_dev/gemm_bench.py $1 $2 $3 40 $4 2>/dev/null | sed "s/^/$ARM  /" | tee -a $OUT/stagger.txt
  done
}
for rep in 1 2; do
  ARM="base        " ; unset EXCEL_BF_FIRST EXCEL_BF_MIX1; run_shapes
  ARM="first8      " ; export EXCEL_BF_FIRST=8; run_shapes
  ARM="first16     " ; export EXCEL_BF_FIRST=16; run_shapes
  ARM="first24     " ; export EXCEL_BF_FIRST=24; run_shapes
  ARM="mix1        " ; unset EXCEL_BF_FIRST; export EXCEL_BF_MIX1=1; run_shapes
  ARM="mix1+first16" ; export EXCEL_BF_FIRST=16; run_shapes
done
for rep in 1 2; do
for ARM in base first16 mix1 both; do
  unset EXCEL_BF_FIRST EXCEL_BF_MIX1
  case $ARM in first16) export EXCEL_BF_FIRST=16;; mix1) export EXCEL_BF_MIX1=1;; both) export EXCEL_BF_FIRST=16 EXCEL_BF_MIX1=1;; esac
  timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>$OUT/err_$ARM.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('pipeline $ARM', 'gemm %.4f' % k['gemm_bf16x3'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'label_agree', d.get('verify', {}).get('label_agreement_mean'))" | tee -a $OUT/stagger.txt
done
done
