#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) --kernel-trace result into the per-kernel stats table that
`--stats` prints: name, calls, total / average / min / max duration, share.  Usage: summarize_rocpd.py results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<72} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'share':>7}")
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) <= 72 else name[:69] + "..."
        print(f"{short:<72} {n:>6} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100.0 * tot / total:>6.2f}%")
    print(f"{'TOTAL':<72} {sum(r[1] for r in rows):>6} {total / 1e6:>10.3f}")
    # bench.py's profiling categories span several template instances: launch-weighted aggregates for direct comparison
    # with its roofline.avg_launch_ms / roofline_par_iterate.avg_launch_ms
    print()
    for cat, key in (("gemm_bf16x3 (w4 + 8-wave tiles)", ("gemm_bf16x3_kernel", "gemm_w4_kernel", "gemm_w4x2_kernel")), ("par_iterate", ("par_iterate",)),
                     ("attn_rowpass", ("attn_rowpass",)), ("attn_accum + attn_strip", ("attn_accum", "attn_strip"))):
        sel = [r for r in rows if any(k in r[0] for k in key)]
        n = sum(r[1] for r in sel)
        if n:
            tot = sum(r[2] for r in sel)
            print(f"category {cat:<28} calls {n:>6}  total_ms {tot / 1e6:>9.3f}  avg_us {tot / n / 1e3:>9.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
