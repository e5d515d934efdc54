#!/bin/bash
# dev experiment (NOT a shipped mode): build a library whose bf16 LO planes keep only the top KEEP of their 7 mantissa bits everywhere they are
# produced (weights, LayerNorm / GEMM / q|k|v / P / A_sum splits), to measure step time and CAM error of a "bf16x3 with short lo planes".
#   bash tools_dev/lo_lite_build.sh <KEEP> -> tools_dev/ab/lite<KEEP>.so ; the tree is restored afterwards
set -e
KEEP=${1:-4}
CS=excel_amd/csrc
python - "$KEEP" <<'PY'
import re, sys, glob
keep = int(sys.argv[1])
mask = (0xFFFF << (7 - keep)) & 0xFFFF
for f in glob.glob("excel_amd/csrc/*.hip") + ["excel_amd/csrc/common.h"]:
    if "train" in f or "decoder" in f: continue
    s = open(f).read(); o = s
    s = re.sub(r"split_hi\(([^()]*?(?:\[[^\]]*\])*[^()]*?) - \(float\)([A-Za-z_]\w*(?:\[[^\]]*\])?)\)", r"split_lo(\1 - (float)\2)", s)
    s = s.replace("split_hi(s[e] - hf)", "split_lo(s[e] - hf)")
    if f.endswith("common.h"):
        s = s.replace("// Two values at once -> packed planes", "#ifndef EXCEL_SPLIT_F16\n__device__ __forceinline__ split_t split_lo(float x) { const split_t r = (split_t)x; unsigned short b = __builtin_bit_cast(unsigned short, r); b &= (unsigned short)0x%04x; return __builtin_bit_cast(split_t, b); }\n#else\n__device__ __forceinline__ split_t split_lo(float x) { return split_hi(x); }\n#endif\n// Two values at once -> packed planes" % mask, 1)
        s = s.replace("    lo2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v - back, b2_));", "    lo2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v - back, b2_)) & 0x%04x%04xu;" % (mask, mask))
    if s != o:
        open(f, "w").write(s); print("patched", f, s.count("split_lo("))
PY
EXCEL_BUILD_FORCE=1 python -m excel_amd.build | tail -1
mkdir -p tools_dev/ab; cp $CS/libexcel_hip.so tools_dev/ab/lite$KEEP.so
git checkout -- $CS
EXCEL_BUILD_FORCE=1 python -m excel_amd.build | tail -1
echo "tools_dev/ab/lite$KEEP.so built; tree restored"
