#!/bin/bash
# dev: link a variant of the library for same-box A/B runs: the shipped objects with ONE translation unit recompiled under extra flags.
#   bash tools_dev/mkvariant.sh <name> <source.hip> [extra hipcc flags...]      -> tools_dev/ab/<name>.so   (base = a copy of the built library)
set -e
NAME=$1; SRC=$2; shift 2
CS=excel_amd/csrc
mkdir -p tools_dev/ab /tmp/mkv_$NAME
[ -f $CS/libexcel_hip.so ] || python -m excel_amd.build
if [ "$SRC" = "-" ]; then cp $CS/libexcel_hip.so tools_dev/ab/$NAME.so; echo "tools_dev/ab/$NAME.so = shipped library"; exit 0; fi
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -pragma-unroll-threshold=100000"
O=/tmp/mkv_$NAME/${SRC%.hip}.o
/opt/rocm/bin/hipcc $FL "$@" -Rpass-analysis=kernel-resource-usage -c $CS/$SRC -o $O 2> /tmp/mkv_$NAME/err.txt || { tail -20 /tmp/mkv_$NAME/err.txt; exit 1; }
grep -E "ScratchSize|VGPRs Spill" /tmp/mkv_$NAME/err.txt | grep -v ": 0 " | head -5 || true
OBJS=""
for o in $CS/*.o; do b=$(basename $o); if [ "$b" = "${SRC%.hip}.o" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS $o"; fi; done
if echo " gemm.hip gemm_bf16x3.hip gemm_w4.hip norm.hip attn.hip attn_strip.hip cam.hip " | grep -q " $SRC "; then
  O2=/tmp/mkv_$NAME/${SRC%.hip}_f16.o
  /opt/rocm/bin/hipcc $FL -DEXCEL_SPLIT_F16 "$@" -c $CS/$SRC -o $O2 2>> /tmp/mkv_$NAME/err.txt
  OBJS=$(echo "$OBJS" | sed "s#$CS/${SRC%.hip}_f16.o#$O2#")
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools_dev/ab/$NAME.so $OBJS
echo "tools_dev/ab/$NAME.so built ($SRC $*)"
