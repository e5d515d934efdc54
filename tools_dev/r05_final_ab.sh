#!/bin/bash
# round 5, second half as a whole: base = head 77ce7b8 (start of the second half), head = the final sources; same box, alternating, with the power side-line
for i in 1 2 3; do for v in base head; do
  EXCEL_AB_LIB=tools_dev/ab/$v.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 --power-seconds 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; p=d.get('power') or {}
print('$v', ' '.join('%s %.3f' % (c, k.get(c, 0)) for c in 'gemm_bf16x3 attn_accum attn_rowpass par_iterate'.split()), 'step', d['ms_per_step'], 'img/s', d['value'], '| W', p.get('socket_power_w'), 'MHz', p.get('shader_clock_mhz'), 'J/step', p.get('joules_per_step'))"
done; done
