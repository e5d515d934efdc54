for M in 12560 16400; do for shape in "2304 768" "768 768" "3072 768" "768 3072"; do
  echo -n "default "; EXCEL_AB_LIB=tools_dev/ab/dev.so python tools_dev/gemm_bench.py $M $shape 30 bf16x3_split 2>&1 | tail -1
  echo -n "w4-320  "; EXCEL_AB_LIB=tools_dev/ab/dev.so EXCEL_BF_TILE=320 python tools_dev/gemm_bench.py $M $shape 30 bf16x3_split 2>&1 | tail -1
done; done
