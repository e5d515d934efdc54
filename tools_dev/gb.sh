for shape in "25120 768 3072" "25120 768 768" "25120 2304 768" "25120 3072 768"; do
  for t in "" 320; do
    echo -n "tile=${t:-auto} "; if [ -z "$t" ]; then unset EXCEL_BF_TILE; else export EXCEL_BF_TILE=$t; fi; python tools_dev/gemm_bench.py $shape 30 bf16x3_split 2>&1 | tail -1
  done
done
