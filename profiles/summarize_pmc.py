#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv output) into per-kernel fabric bytes per launch.
Usage: summarize_pmc.py fetch_counter_collection.csv write_counter_collection.csv [head] > hbm_traffic.json

bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB; FETCH_SIZE is doubled per
/opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read).  Infinity-Cache hits are counted,
so this is fabric traffic: an upper bound on HBM bytes."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\bexcel_(bf16|f16)::", lambda m: "" if m.group(1) == "bf16" else "f16::", name)   # (the split-type namespaces of round 4)
    return name.split("(")[0][:60]


def fold(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def source_stamp(head):
    """Which code the counters were collected on: git head (as given), date, and the sha of the kernel sources (bench.py compares it)."""
    import datetime
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import csrc_sha16
    return {"head": head, "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "csrc_sha16": csrc_sha16(),
            "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --cpu-images 0 --ragged-images 0"}


def main(fetch_csv, write_csv, head=None):
    fe, wr = fold(fetch_csv, "FETCH_SIZE"), fold(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fe) | set(wr)):
        n = max(fe[k][0], wr[k][0]) or 1
        f_kb, w_kb = fe[k][1] / (fe[k][0] or 1), wr[k][1] / (wr[k][0] or 1)
        out[k] = dict(launches=n, fetch_size_kb_avg=round(f_kb, 1), write_size_kb_avg=round(w_kb, 1),
                      bytes_per_launch=int((2 * f_kb + w_kb) * 1024))
    top = {}
    # (the bf16x3 GEMM category of bench.py = the four-wave kernel gemm_w4_kernel_* + the 8-wave gemm_bf16x3_kernel instances)
    for cat, prefix in (("gemm_bf16x3", ("gemm_bf16x3_kernel", "gemm_w4_kernel", "gemm_w4x2_kernel")), ("par_iterate", ("par_iterate",)), ("attn_rowpass", ("attn_rowpass",)),
                        ("attn_accum", ("attn_accum",)), ("attn_strip", ("attn_strip",)), ("par_iterate_guide", ("par_iterate_guide",))):
        ks = [v for k, v in out.items() if any(px in k for px in prefix)]
        n = sum(v["launches"] for v in ks)
        if n:        # launch-weighted mean over the template instances of one bench category
            top[cat + "_bytes_per_launch"] = int(sum(v["bytes_per_launch"] * v["launches"] for v in ks) / n)
    print(json.dumps({**top, "_source": source_stamp(head), "_note": __doc__.split("\n\n")[1].replace("\n", " "), "kernels": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
