#!/usr/bin/env python3
"""Host-side ceiling of BASELINE configs[3] at R ranks: R concurrent DECODE-ONLY processes, each running the harness's own decode pool
(datasets/loader.threaded_batches: Pillow JPEG + palette PNG in a thread pool, packed into ragged batches) over a synthetic on-disk VOC
tree, pinned to disjoint core sets like `infer_lam --cpu_affinity auto` would pin its ranks (or unpinned).  No GPU work: the aggregate
img/s printed here is what the host can feed, to be compared with R x the per-GPU rate of the pipeline.

    python tools_dev/decode_ceiling.py --procs 8 --threads 16 --images 512 --passes 3 [--no-affinity]
Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    from excel_amd.tools.infer_lam import pin_rank_to_cores
    cores = pin_rank_to_cores(args.rank, args.procs) if not args.no_affinity else None
    from excel_amd.datasets import voc
    from excel_amd.datasets.loader import threaded_batches
    ds = voc.VOC12SegDataset(root_dir=args.root, name_list_dir=args.lists, split="train", stage="val")
    order = [i for _ in range(args.passes) for i in range(args.rank, len(ds), args.procs)] or list(range(len(ds)))
    # wait for the common start time so the processes really overlap
    while time.time() < args.start_at:
        time.sleep(0.005)
    t0 = time.perf_counter()
    n = nbytes = 0
    for rb in threaded_batches(ds, order, args.batch, num_threads=args.threads):
        n += len(rb)
        nbytes += rb.images.numel()
    dt = time.perf_counter() - t0
    print(json.dumps({"rank": args.rank, "images": n, "seconds": round(dt, 3), "images_per_s": round(n / dt, 1),
                      "mbytes_decoded": round(nbytes / 1e6, 1), "cores": len(cores) if cores else None}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-affinity", action="store_true")
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--root")
    ap.add_argument("--lists")
    ap.add_argument("--start-at", type=float, default=0.0)
    args = ap.parse_args()
    if args.rank >= 0:
        return worker(args)
    from excel_amd.tools import synthetic
    tmp = tempfile.mkdtemp(prefix="excel_decode_")
    try:
        root, lists = os.path.join(tmp, "VOC2012"), os.path.join(tmp, "lists")
        synthetic.write_voc_tree(root, lists, args.images, seed=4321)
        start_at = time.time() + 6.0          # interpreter + imports of R processes
        cmd = [sys.executable, os.path.abspath(__file__), "--procs", str(args.procs), "--threads", str(args.threads), "--passes", str(args.passes),
               "--batch", str(args.batch), "--root", root, "--lists", lists, "--start-at", repr(start_at)] + (["--no-affinity"] if args.no_affinity else [])
        procs = [subprocess.Popen(cmd + ["--rank", str(r)], stdout=subprocess.PIPE, text=True) for r in range(args.procs)]
        recs = []
        for p in procs:
            out, _ = p.communicate(timeout=900)
            recs.append(json.loads(out.strip().splitlines()[-1]))
        wall = max(r["seconds"] for r in recs)
        total = sum(r["images"] for r in recs)
        print(json.dumps({"procs": args.procs, "threads_per_proc": args.threads, "affinity": not args.no_affinity, "host_cpus": os.cpu_count(),
                          "images_total": total, "wall_s": wall, "aggregate_images_per_s": round(total / wall, 1),
                          "per_proc_images_per_s": [r["images_per_s"] for r in recs], "cores_per_proc": recs[0]["cores"]}), flush=True)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
