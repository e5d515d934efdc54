#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (csv) into a per-kernel pipe-utilisation table.
Usage: summarize_busy.py busy1_counter_collection.csv busy2_counter_collection.csv > pipe_busy.txt

Per launch averages.  Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs (= 32 x MFMA count for
32x32x16 bf16); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE = kernel cycles x 8 XCDs.
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x kernel cycles)              share of matrix-pipe cycles issued
  valu_busy  = 4 x SQ_ACTIVE_INST_VALU / (1024 x kernel cycles)               share of SIMD cycles a VALU instruction was issuing
  lds_busy   = SQ_LDS_IDX_ACTIVE / (256 x kernel cycles)                      share of LDS cycles
  wait / stall / active = wave-state split of SQ_WAVE_CYCLES (s_waitcnt+barrier / issue stall / issuing)"""
import csv
import sys
from collections import defaultdict


def fold(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0][:52]
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: {c: v[0] / max(v[1], 1) for c, v in d.items()} | {"_n": max(v[1] for v in d.values())} for k, d in acc.items()}


def main(p1, p2):
    a, b = fold(p1), fold(p2)
    rows = []
    for k in a:
        if k not in b or "GRBM_GUI_ACTIVE" not in b[k]:
            continue
        cyc = b[k]["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0:
            continue
        wc = a[k].get("SQ_WAVE_CYCLES", 0.0) or 1.0
        rows.append((cyc * a[k]["_n"], k, a[k]["_n"], cyc, a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc),
                     4 * a[k].get("SQ_ACTIVE_INST_VALU", 0) / (1024 * cyc), b[k].get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc),
                     a[k].get("SQ_WAIT_ANY", 0) / wc, a[k].get("SQ_WAIT_INST_ANY", 0) / wc, a[k].get("SQ_ACTIVE_INST_ANY", 0) / wc,
                     b[k].get("SQ_INSTS_VALU", 0) / max(b[k].get("SQ_INSTS_MFMA", 0), 1) if b[k].get("SQ_INSTS_MFMA", 0) else 0.0,
                     b[k].get("SQ_LDS_BANK_CONFLICT", 0) / max(b[k].get("SQ_LDS_IDX_ACTIVE", 0), 1)))
    rows.sort(reverse=True)
    print(f"{'kernel':54s} {'launches':>8s} {'kcycles':>9s} {'mfma_busy':>9s} {'valu_busy':>9s} {'lds_busy':>8s} {'wait':>6s} {'stall':>6s} {'active':>6s} {'valu/mfma':>9s} {'lds_confl':>9s}")
    for _, k, n, cyc, mf, va, ld, w, st, ac, vm, lc in rows[:14]:
        print(f"{k:54s} {int(n):8d} {cyc / 1e3:9.1f} {mf:9.3f} {va:9.3f} {ld:8.3f} {w:6.2f} {st:6.2f} {ac:6.2f} {vm:9.1f} {lc:9.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
