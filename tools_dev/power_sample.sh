#!/bin/bash
# dev: socket power / shader clock sampled by rocm-smi (every ~0.3 s) while the bench loop runs 600 steps, then while the whole-chip MFMA
# stream runs on random / zero operands (tools_dev/micro/mfma_power)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-r04t}; mkdir -p $OUT
sampler() { while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.25; done; }
sampler > $OUT/samples_bench.txt & SP=$!
python bench.py --steps 600 --warmup 5 --cpu-images 0 --ragged-images 0 --no-kernel-timing > $OUT/bench600.json 2>/dev/null
kill $SP
sampler > $OUT/samples_mfma.txt & SP=$!
./tools_dev/micro/mfma_power 100000 > $OUT/mfma_long.txt 2>&1 &
MP=$!; sleep 25; kill $MP; kill $SP
python - <<PY | tee $OUT/power.txt
import json, re
def top(f):
    rows = []
    for l in open(f):
        m = re.findall(r"\(?(\d+)Mhz\)?.*?(\d+\.\d+)", l)
        w = re.findall(r"(\d+\.\d+)\s*$", l.strip())
        c = re.findall(r"\((\d+)Mhz\)", l)
        if w and c: rows.append((float(w[0]), int(c[0])))
    rows.sort(reverse=True)
    return rows[:8], len(rows)
d = json.loads(open("$OUT/bench600.json").read().strip().splitlines()[-1])
print("bench 600 steps:", d["value"], "img/s", d["ms_per_step"], "ms/step")
print("bench loop, highest (W, sclk MHz) samples:", top("$OUT/samples_bench.txt"))
print("MFMA stream (zeros then smooth then random, 256-CU launches are the long ones), highest samples:", top("$OUT/samples_mfma.txt"))
PY
head -12 $OUT/mfma_long.txt
