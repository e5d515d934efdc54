#!/bin/bash
# dev experiment: the step with 4-bit bf16 lo planes everywhere (tools_dev/lo_lite_build.sh; NOT a shipped mode) vs the shipped library: time, CAM
# difference against exact fp32 on the benchmark's weights (numerics_check), label agreement with the CPU port (verify)
for i in 1 2; do for v in base lite4; do
  EXCEL_AB_LIB=tools_dev/ab/$v.so timeout 400 python tools_dev/ab_bench.py --cpu-images 16 --ragged-images 0 --fp16w-steps 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; n=d.get('numerics_check') or {}; v=d.get('verify') or {}
print('$v', ' '.join('%s %.3f' % (c, k.get(c, 0)) for c in 'gemm_bf16x3 attn_accum attn_rowpass par_iterate'.split()), 'step', d['ms_per_step'], 'img/s', d['value'], '| CAM max-abs vs exact fp32', n.get('max_abs_diff'), '| label agreement mean / min', v.get('label_agreement_mean'), v.get('label_agreement_min'))"
done; done
