for l in 0 20000 40000; do
  EXCEL_ROWPASS_LDS=$l timeout 120 python bench.py --cpu-images 0 --ragged-images 0 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('lds+$l', 'rowpass', k['attn_rowpass'], 'step', d['ms_per_step'])"
done
