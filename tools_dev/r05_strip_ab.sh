#!/bin/bash
# round 5: strip kernel, W sweep on the flash row pass's row statistics (stripk = working tree) vs the exchange form (parh), same box; full GPU suite first
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
bash tools_dev/abn.sh "attn_accum attn_rowpass par_iterate gemm_bf16x3" 3 parh stripk
