"""CPU-side tests: the C-ABI library loads and exports every symbol include/excel_hip.h declares (no compute calls),
host logic (sharding, synthetic data, metric arithmetic, table), the oracle's OpenCV known-answer cases, and the
N>1 path (rank-strided shards + ONE all_gather of the confusion matrix) with world_size 2 on gloo."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "excel_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(excel_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from excel_amd import _lib
    names = _declared_functions()
    assert len(names) >= 25
    if not os.path.exists(_lib.LIB_PATH):
        from excel_amd import build
        build.build(verbose=False)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/excel_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in excel_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names
    lib = _lib.lib()
    assert lib.excel_abi_version() >= 1
    assert lib.excel_prof_num_categories() >= 10


def test_ragged_plan_is_a_host_function():
    """excel_ragged_plan builds the tile map of a ragged batch without touching a device (include/excel_hip.h, "ragged batches")."""
    from excel_amd import ops
    sizes = [(375, 500), (500, 333), (1, 1), (16, 64), (17, 65)]
    plan = ops.RaggedPlan(sizes, None)
    rec = plan.table_host[:8 * 6].reshape(6, 8)
    assert [tuple(r[:2]) for r in rec[:5]] == sizes
    tiles = [-(-w // 64) * -(-h // 16) for h, w in sizes]
    assert tiles == [24 * 8, 32 * 6, 1, 1, 4]
    assert list(rec[:, 3]) == list(np.concatenate([[0], np.cumsum(tiles)]))
    assert list(rec[:, 2]) == list(np.concatenate([[0], np.cumsum([h * ((w + 3) // 4 * 4) for h, w in sizes])]))
    assert list(rec[:, 4]) == list(np.concatenate([[0], np.cumsum([h * w for h, w in sizes])]))
    tile_img = plan.table_host[8 * 6:]
    assert len(tile_img) == sum(tiles) == plan.total_tiles
    assert np.array_equal(tile_img, np.repeat(np.arange(5), tiles))
    assert plan.info.max_plane_pix == 375 * 500
    with pytest.raises(RuntimeError, match="size"):
        ops.RaggedPlan([(0, 5)], None)


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: handing CPU tensors to an op raises instead of silently computing elsewhere."""
    from excel_amd import ops
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ops.VitHandle({}, 128, 1, 2, 16, 64, device="cpu")


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "excel_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"


def test_shard_indices_partition():
    from excel_amd.tools.infer_lam import shard_indices
    for n, R in ((10582, 8), (64, 3), (5, 8)):
        parts = [shard_indices(n, r, R) for r in range(R)]
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(n))
        assert np.array_equal(parts[1 % R][:3], np.arange(1 % R, n, R)[:3])


def test_synthetic_dataset_is_deterministic_and_voc_shaped():
    from excel_amd.tools import synthetic
    ds = synthetic.SyntheticSegDataset(50, (32, 48), seed=7)
    a, b = ds[13], ds[13]
    assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
    name, img, gt, cls = a
    assert img.shape == (3, 32, 48) and img.dtype == np.float32 and gt.dtype == np.uint8 and cls.shape == (20,)
    assert set(np.unique(gt)) <= set(range(21)) | {255}
    ks = [int(ds[i][3].sum()) for i in range(50)]
    assert 1 <= min(ks) and max(ks) <= 6
    sd = synthetic.make_vit_state_dict(dict(width=128, layers=2, heads=2, output_dim=64, input_resolution=64), seed=3)
    assert sd["transformer.resblocks.1.attn.in_proj_weight"].shape == (384, 128) and sd["proj"].shape == (128, 64)


def test_scores_from_hist_matches_oracle_and_table():
    from excel_amd.utils import evaluate
    from excel_amd.tools.infer_lam import format_scores_table, VOC_CLASSES
    rs = np.random.RandomState(0)
    hist = rs.randint(0, 1000, (21, 21)).astype(np.int64)
    hist[7] = 0            # a class absent from the ground truth
    a = evaluate.scores_from_hist(hist)
    b = oracle.evaluate.scores_from_hist(hist)
    assert a["miou"] == b["miou"] and a["pAcc"] == b["pAcc"] and a["mAcc"] == b["mAcc"]
    for k in ("iou", "precision", "recall", "confusion"):
        np.testing.assert_array_equal(np.array(list(a[k].values())), np.array(list(b[k].values())))
    tab = format_scores_table(a, VOC_CLASSES)
    assert "aeroplane" in tab and "average_metrics" in tab and "iou" in tab.splitlines()[1]


from _known_boxes import GRADED, KNOWN  # noqa: E402


@pytest.mark.parametrize("name", sorted(KNOWN))
def test_oracle_box_mask_known_answers(name):
    """Hand-derived from OpenCV's documented semantics (findContours outer contours are 8-connected; boundingRect is
    the tight box; the reference clamps x1/y1 to W-1/H-1 and fills end-exclusive, utils/affutils.py:46-51,:212)."""
    rows, exp = KNOWN[name]
    m = np.array([[float(c) for c in r] for r in rows], np.float32)
    e = np.array([[int(c) for c in r] for r in exp], np.float32)
    assert np.array_equal(oracle.aff.box_mask(m, 0.5), e)


@pytest.mark.parametrize("name", sorted(GRADED))
def test_oracle_box_mask_graded_known_answers(name):
    vals, thr, exp = GRADED[name]
    e = np.array([[int(c) for c in r] for r in exp], np.float32)
    assert np.array_equal(oracle.aff.box_mask(np.array(vals, np.float32), thr), e)


def test_oracle_scoremap2bbox_threshold_rule():
    # u8 truncation, thr = int(0.79 * max), strict >
    m = np.zeros((4, 4), np.float32)
    m[1, 1] = 1.0          # 255
    m[1, 2] = 0.7905       # trunc(201.58) = 201 ; thr = int(0.79*255) = 201 -> NOT above
    m[2, 1] = 0.7922       # trunc(202.01) = 202 -> above
    boxes, cnt = oracle.aff.scoremap2bbox(m, 0.79)
    assert cnt == 1 and boxes.tolist() == [[1, 1, 2, 3]]
    assert oracle.aff.scoremap2bbox(np.zeros((3, 3), np.float32), 0.79)[0].tolist() == [[0, 0, 0, 0]]


def test_cv2_resize_restatement_matches_half_pixel_bilinear():
    rs = np.random.RandomState(1)
    img = rs.rand(28, 28).astype(np.float32)
    a = oracle.interp.cv2_resize_linear(img, 500, 375)
    b = oracle.interp.bilinear_resize(img, 375, 500, align_corners=False)
    assert a.shape == (375, 500) and np.abs(a - b).max() < 1e-5      # double- vs float-computed source index


# ------------------------------------------------------------------ N > 1: world_size 2 on gloo
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, nc, q):
    import torch.distributed as dist
    from excel_amd.tools.infer_lam import gather_hists, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hist = torch.zeros((nc, nc), dtype=torch.int64)
    for i in shard_indices(n, rank, world):
        rs = np.random.RandomState(1000 + int(i))
        gt = rs.randint(0, nc, 500)
        gt[rs.rand(500) < 0.05] = 255
        pr = rs.randint(0, nc, 500)
        hist += torch.from_numpy(oracle.evaluate.fast_hist(gt, pr, nc))
    per_rank, total = gather_hists(hist)
    q.put((rank, per_rank.numpy(), total.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_confusion_allgather_world2():
    import torch.multiprocessing as mp
    n, nc, world = 11, 21, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nc, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = np.zeros((nc, nc), np.int64)
    for i in range(n):
        rs = np.random.RandomState(1000 + i)
        gt = rs.randint(0, nc, 500)
        gt[rs.rand(500) < 0.05] = 255
        pr = rs.randint(0, nc, 500)
        ref += oracle.evaluate.fast_hist(gt, pr, nc)
    for rank, per_rank, total in res:
        assert per_rank.shape == (world, nc, nc)
        assert np.array_equal(total, ref)                     # every rank holds the same aggregated matrix
        assert np.array_equal(per_rank.sum(0), ref)


def _ladder_worker(rank, world, port, q):
    """Rank 1 has an EMPTY shard (1 image, 2 ranks).  The stub model records what the ladder asked of it."""
    import torch.distributed as dist
    from types import SimpleNamespace
    from excel_amd.model.model_excel import ExCEL_model
    from excel_amd.tools import infer_lam
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class _Handle:
        mode = "bf16x3"
        def gemm_mode(self): return self.mode
        def set_gemm_mode(self, m): self.mode = m

    class _Model:                     # ExCEL_model.check_numerics itself, on a stub tower whose bf16x3 CAMs are 1e-2 off, f16x3 1e-5
        check_numerics = ExCEL_model.check_numerics
        def __init__(self):
            self.h = _Handle()
            self.encoder = SimpleNamespace(visual=SimpleNamespace(handle=lambda: self.h))
            self.calls = 0
        def forward(self, img):
            self.calls += 1
            off = {"f32": 0.0, "bf16x3": 1e-2, "f16x3": 1e-5}[self.h.mode]
            return None, None, img[:, :1].float() + off

    class _Set:
        def __len__(self): return 1
        def __getitem__(self, i): return "s", np.zeros((3, 8, 8), np.float32), None, None
        def batch(self, take): return None, np.zeros((len(take), 3, 8, 8), np.float32), None, None

    model, ds = _Model(), _Set()
    idx = infer_lam.shard_indices(len(ds), rank, world)
    args = SimpleNamespace(resize_size=8, gemm_check_tol=5e-4)
    out = infer_lam._gemm_self_check(model, ds, idx, args, torch.device("cpu"), world)
    q.put((rank, len(idx), model.calls, out["mode_after"], [m for m, _ in out["ladder"]], [float(d) for _, d in out["ladder"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_gemm_self_check_empty_shard_joins_collectives():
    """Round-4 advisor finding: a rank with an empty shard skipped the ladder's all-reduces and the other ranks hung.  Now every rank
    walks the same rungs (the empty one contributes 0) and lands on the same mode."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ladder_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, n0, calls0, after0, modes0, d0), (r1, n1, calls1, after1, modes1, d1) = res
    assert (n0, n1) == (1, 0) and calls0 == 3 and calls1 == 0          # exact + two rungs on rank 0, nothing on the empty rank
    assert after0 == after1 == "f16x3" and modes0 == modes1 == ["bf16x3", "f16x3"]
    assert d0 == d1 and abs(d0[0] - 1e-2) < 1e-6 and d0[1] < 5e-4


class _TinyRaggedSet:
    """(name, image u8 [h,w,3], label u8 [h,w], cls f32 [20]) with a different size per sample."""
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def max_k(self):
        return 2

    def __getitem__(self, i):
        rs = np.random.RandomState(500 + i)
        h, w = 5 + i % 7, 4 + (3 * i) % 5
        gt = rs.randint(0, 21, (h, w)).astype(np.uint8)
        gt[rs.rand(h, w) < 0.1] = 255
        cls = np.zeros(20, np.float32)
        cls[i % 20] = 1
        return f"s{i:03d}", rs.randint(0, 256, (h, w, 3)).astype(np.uint8), gt, cls


class _StubPipe:
    """Stands in for TrainingFreePipeline in the control-flow test: `labels` = a deterministic function of the image bytes."""
    device, smax = "cpu", 2

    def __init__(self):
        self.hist, self.batches = None, []

    def run_batch_ragged(self, images, plan, cls, gts, S=448, return_intermediates=False):
        assert images.numel() == 3 * plan.total_label_pix and gts.numel() == plan.total_label_pix and cls.shape[0] == plan.B
        pred = (images.view(-1, 3)[:, 0].to(torch.int64) % 21).numpy()
        self.hist += torch.from_numpy(oracle.evaluate.fast_hist(gts.numpy(), pred, 21))
        self.batches.append(plan.B)
        return torch.from_numpy(pred.astype(np.uint8))


def _validate_worker(rank, world, port, n, bs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from excel_amd.tools import infer_lam
    args = infer_lam.get_parser().parse_args(["--batch_size", str(bs), "--num_workers", "0", "--backend", "gloo"])
    pipe = _StubPipe()
    score, total = infer_lam.validate(args, dataset=_TinyRaggedSet(n), pipe=pipe)
    q.put((rank, pipe.batches, infer_lam.validate.last_per_rank.numpy(), total.numpy(), float(score["miou"])))
    dist.barrier()
    dist.destroy_process_group()


def test_validate_control_flow_world4_uneven_tail():
    """tools/infer_lam.validate's own control flow (rank-strided shards :166, ragged batches, last partial batch, ONE all_gather) on
    4 gloo ranks with an uneven tail - 46 = 4 * 11 + 2 samples, batch 5: ranks 0,1 run 12 images (5+5+2), ranks 2,3 run 11 (5+5+1),
    like 10 582 = 8 * 1 322 + 6 on the 8-GPU node - through a stub pipeline: every rank ends with the matrix of ALL samples."""
    import torch.multiprocessing as mp
    n, world, bs = 46, 4, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_validate_worker, args=(r, world, port, n, bs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in [q.get(timeout=180) for _ in range(world)]}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ds = _TinyRaggedSet(n)
    per_sample = []
    for i in range(n):
        _, img, gt, _ = ds[i]
        per_sample.append(oracle.evaluate.fast_hist(gt.flatten(), img.reshape(-1, 3)[:, 0].astype(np.int64) % 21, 21))
    ref = np.sum(per_sample, 0)
    for rank in range(world):
        _, batches, per_rank, total, miou = res[rank]
        assert batches == ([5, 5, 2] if rank < 2 else [5, 5, 1])
        assert np.array_equal(total, ref)
        for r in range(world):
            assert np.array_equal(per_rank[r], np.sum(per_sample[r::world], 0))
        assert abs(miou - oracle.evaluate.scores_from_hist(ref)["miou"]) < 1e-12


class _BenchStubPipe:
    """Stands in for TrainingFreePipeline under bench.main: the 'labels' of a batch are a function of its images, scored into `hist`."""

    def __init__(self):
        self.hist = torch.zeros(21, 21, dtype=torch.int64)
        self.calls = 0

    def reset(self):
        self.hist.zero_()

    def drain(self):
        pass

    def run_batch(self, imgs, cls, gts):
        pred = (imgs.view(imgs.shape[0], -1)[:, :gts[0].numel()].to(torch.int64) % 21).reshape(gts.shape)
        self.hist += torch.from_numpy(oracle.evaluate.fast_hist(gts.numpy().ravel(), pred.numpy().ravel(), 21))
        self.calls += 1
        return pred


def _bench_stub_batches(rank, world, B, n_batches=2):
    """Rank r's resident batches: images r, r+R, ... of a world*B*n_batches data set (bench.make_workload's sharding)."""
    from excel_amd.tools.infer_lam import shard_indices
    mine = shard_indices(world * B * n_batches, rank, world)
    out = []
    for i in range(n_batches):
        idx = mine[i * B:(i + 1) * B]
        imgs = torch.stack([torch.from_numpy(np.random.RandomState(900 + int(j)).randint(0, 256, (6, 7)).astype(np.int64)) for j in idx])
        gts = torch.stack([torch.from_numpy(np.random.RandomState(1900 + int(j)).randint(0, 21, (6, 7)).astype(np.uint8)) for j in idx])
        out.append((imgs, torch.zeros(B, 20), gts))
    return out, mine


def _bench_worker(rank, world, port, B, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import contextlib
    import io
    import torch.distributed as dist  # noqa: F401
    import bench
    pipe = _BenchStubPipe()

    def make_workload(args, r, w, device):
        assert (r, w) == (rank, world) and args.batch == B
        batches, _ = _bench_stub_batches(r, w, B)
        return pipe, batches, [np.ones(B), np.ones(B)], None

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", str(steps), "--warmup", "1", "--batch", str(B), "--no-kernel-timing",
                    "--cpu-images", "0", "--ragged-images", "0"],
                   hooks={"backend": "gloo", "device": "cpu", "make_workload": make_workload})
    q.put((rank, buf.getvalue(), pipe.calls, pipe.hist.numpy()))


def test_bench_main_world8_gloo_stub_pipeline():
    """bench.py's OWN multi-rank path (the command the driver's 8-GPU run executes) on 8 gloo ranks with a stand-in pipeline: sharding by
    rank, warm-up + reset, the timed loop, the ONE all-gather of the confusion matrices, the max-over-ranks time, exactly one JSON line from
    rank 0 whose `value` is the whole job's images over that time and whose per-rank masses show all 8 ranks."""
    import json
    import torch.multiprocessing as mp
    world, B, steps = 8, 3, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, B, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in [q.get(timeout=300) for _ in range(world)]}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lines = [l for l in res[0][1].splitlines() if l.strip()]
    assert len(lines) == 1 and all(not res[r][1].strip() for r in range(1, world))      # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["scaling"] == "weak" and out["steps"] == steps
    assert abs(out["value"] - world * B * steps / (out["ms_per_step"] * 1e-3 * steps)) < 1e-3 * out["value"] + 1e-3
    # every rank ran warm-up + steps batches of ITS shard; the gathered masses are the ranks' own scored pixels
    expect_total = np.zeros((21, 21), np.int64)
    for r in range(world):
        assert res[r][2] == 1 + steps
        batches, mine = _bench_stub_batches(r, world, B)
        assert list(mine) == list(range(r, world * B * 2, world))
        h = np.zeros((21, 21), np.int64)
        for i in range(steps):
            imgs, _, gts = batches[i % 2]
            pred = (imgs.view(B, -1).numpy() % 21).ravel()
            h += oracle.evaluate.fast_hist(gts.numpy().ravel(), pred, 21)
        assert np.array_equal(res[r][3], h)
        assert out["per_rank_hist_mass"][r] == int(h.sum())
        expect_total += h
    assert abs(out["miou_synthetic"] - round(float(oracle.evaluate.scores_from_hist(expect_total)["miou"]), 6)) < 1e-9


def test_bench_self_launch_argv():
    """`python bench.py --gpus N` launched bare re-executes itself as N ranks under torch.distributed.run on 127.0.0.1, with its own
    arguments passed through (the driver's multi-GPU command has exactly this shape)."""
    import sys
    import bench
    argv = bench.self_launch_argv(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29417)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29417"
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    port = int(bench.self_launch_argv([], 2)[bench.self_launch_argv([], 2).index("--master-port") + 1])
    assert 1024 < port < 65536
    # without a GPU the bare command refuses before it launches anything
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "2"])


def test_pin_rank_to_cores_disjoint_shares():
    """infer_lam.pin_rank_to_cores: local rank r of R takes the r-th contiguous share of the cores the process may run on; the shares are
    disjoint (8 ranks x (16 decode threads + launch thread) on one host, BASELINE configs[3])."""
    from excel_amd.tools.infer_lam import pin_rank_to_cores
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("needs >= 2 cores and sched_setaffinity")
    saved = os.sched_getaffinity(0)
    try:
        cores = sorted(saved)
        R = min(4, len(cores))
        shares = []
        for r in range(R):
            os.sched_setaffinity(0, saved)
            shares.append(pin_rank_to_cores(r, R))
            assert os.sched_getaffinity(0) == shares[-1]
        per = len(cores) // R
        assert all(len(sh) == per for sh in shares) and len(set().union(*shares)) == R * per
        assert shares[0] == set(cores[:per])
        os.sched_setaffinity(0, saved)
        assert pin_rank_to_cores(0, 1) is None and os.sched_getaffinity(0) == saved
    finally:
        os.sched_setaffinity(0, saved)


def test_default_decode_workers_follow_the_cpu_budget():
    """infer_lam: the default decode-pool size is this rank's share of the CPUs the process may really use (affinity, cgroup quota), 2..16."""
    from excel_amd.tools import infer_lam
    budget = infer_lam.host_cpu_budget()
    assert 1 <= budget <= (os.cpu_count() or 1)
    assert infer_lam.default_decode_workers(1) == max(2, min(16, budget))
    assert infer_lam.default_decode_workers(8) == max(2, min(16, budget // 8 - 1))
    assert infer_lam.get_parser().parse_args([]).num_workers == -1


def test_loaded_library_is_built_from_these_sources():
    """excel_build_id(): the id compiled into libexcel_hip.so equals the id of the sources in the tree (excel_amd/build.py:source_id)."""
    from excel_amd import _lib, build
    build.build(verbose=False)
    assert _lib.build_id() == build.source_id() and len(build.source_id(dev=False)) == 16


def test_ragged_loaders_pack_in_order():
    """datasets/loader: threaded_batches (decode thread pool, the harness default) and ragged_batches (DataLoader worker processes, the
    reference's mechanism) hand out the same RaggedBatches, in index order, last batch partial, images / labels packed back to back."""
    from excel_amd.datasets.loader import pack_samples, ragged_batches, threaded_batches
    ds = _TinyRaggedSet(23)
    order = list(range(22, -1, -1))
    ref = [pack_samples([ds[j] for j in order[s0:s0 + 5]]) for s0 in range(0, 23, 5)]
    for batches in (threaded_batches(ds, order, 5, num_threads=3, ahead=2), ragged_batches(ds, order, 5, num_workers=0, pin_memory=False)):
        got = list(batches)
        assert [len(b) for b in got] == [5, 5, 5, 5, 3]
        for a, b in zip(got, ref):
            assert a.names == b.names and np.array_equal(a.hw, b.hw)
            assert torch.equal(a.images, b.images) and torch.equal(a.labels, b.labels) and torch.equal(a.cls, b.cls)
    b0 = ref[0]
    assert b0.images.numel() == 3 * int((b0.hw[:, 0] * b0.hw[:, 1]).sum()) and b0.labels.numel() == int((b0.hw[:, 0] * b0.hw[:, 1]).sum())
    _, img, lab, _ = ds[22]
    assert np.array_equal(b0.images[: img.size].numpy().reshape(img.shape), img) and np.array_equal(b0.labels[: lab.size].numpy().reshape(lab.shape), lab)
    with pytest.raises(ValueError, match="one size"):
        pack_samples([("x", np.zeros((4, 5, 3), np.uint8), np.zeros((4, 6), np.uint8), np.zeros(20, np.float32))])


def test_gather_hists_single_process_identity():
    from excel_amd.tools.infer_lam import gather_hists
    h = torch.arange(9, dtype=torch.int64).reshape(3, 3)
    per_rank, total = gather_hists(h)
    assert per_rank.shape == (1, 3, 3) and torch.equal(total, h)


def test_on_disk_formats(tmp_path):
    """the VOC palette (known-answer entries of the published colormap), the logits record
    tools/infer_lam.py hands to its CRF stage, and the key lookup that stage applies."""
    from excel_amd.utils import imutils
    cm = imutils.colormap()
    assert cm.shape == (256, 3) and cm.dtype == np.uint8
    known = {0: (0, 0, 0), 1: (128, 0, 0), 2: (0, 128, 0), 3: (128, 128, 0), 4: (0, 0, 128), 8: (64, 0, 0), 15: (192, 128, 128),
             20: (0, 64, 128), 255: (224, 224, 192)}
    for k, rgb in known.items():
        assert tuple(int(v) for v in cm[k]) == rgb, k
    lab = np.array([[0, 1, 255], [20, 15, 2]], np.uint8)
    rgb = imutils.encode_cmap(lab)
    assert rgb.shape == (2, 3, 3) and tuple(rgb[0, 2]) == (224, 224, 192)
    rs = np.random.RandomState(0)
    lam = rs.rand(3, 5, 7).astype(np.float32)
    keys = np.array([4, 17], np.int64)
    p = imutils.save_logits(str(tmp_path / "logits"), "2007_000032", lam, keys)
    lam2, keys2 = imutils.load_logits(p)
    assert np.array_equal(lam, lam2) and np.array_equal(keys, keys2)
    d = np.load(p, allow_pickle=True).item()                       # exactly the reference reader's view (:203-206)
    assert set(d) == {"valid_lam", "keys_gt"}
    lab2 = imutils.crf_keys_to_labels(lam, keys)
    assert set(np.unique(lab2)) <= {0, 5, 18} and lab2.dtype == np.uint8
    png = imutils.save_label_png(str(tmp_path / "segs" / "x.png"), lab2)
    from PIL import Image
    back = np.asarray(Image.open(png))
    assert np.array_equal(back, imutils.encode_cmap(lab2))


def test_bpe_tokenizer_matches_reference_ids(golden):
    """Token ids minted with the reference's own tokenizer (tests/golden/make_goldens.py gold_text).  CLIP's merges file is
    not shipped with this repo: the test runs where one is available (EXCEL_BPE_VOCAB or the build container's reference)."""
    import os
    from excel_amd.clip import bpe
    path = os.environ.get("EXCEL_BPE_VOCAB") or "/root/reference/clip/bpe_simple_vocab_16e6.txt.gz"
    if not os.path.exists(path):
        pytest.skip("CLIP BPE merges file not available")
    g = golden("text_tiny.npz")
    tk = bpe.BPETokenizer(path)
    assert len(tk.encoder) == 49408 and tk.encoder[bpe.SOT] == 49406 and tk.encoder[bpe.EOT] == 49407
    ids = bpe.tokenize([str(t) for t in g["texts"]], tk)
    assert np.array_equal(ids, g["token_ids"])
    assert tk.decode(ids[0][1:list(ids[0]).index(49407)]).strip() == "a clean origami aeroplane ."
    with pytest.raises(RuntimeError):
        bpe.tokenize(["word " * 100], tk)
    assert bpe.tokenize(["word " * 100], tk, truncate=True)[0, -1] == 49407


def test_voc_on_disk_dataset(tmp_path):
    """datasets/voc.py input format: id list, JPEG image, palette PNG label, pickled one-hot dict."""
    from PIL import Image
    from excel_amd.datasets import voc
    from excel_amd.utils import imutils
    root, lists = tmp_path / "VOC2012", tmp_path / "lists"
    (root / "JPEGImages").mkdir(parents=True)
    (root / "SegmentationClassAug").mkdir()
    lists.mkdir()
    rs = np.random.RandomState(0)
    ids, onehot = ["2007_000033", "2007_000042"], {}
    pal = imutils.colormap().flatten().tolist()
    for k, name in enumerate(ids):
        img = rs.randint(0, 256, (20 + k, 24, 3)).astype(np.uint8)
        Image.fromarray(img).save(root / "JPEGImages" / (name + ".jpg"), quality=95)
        lab = rs.randint(0, 21, (20 + k, 24)).astype(np.uint8)
        lab[0, :3] = 255
        im = Image.fromarray(lab, mode="P")
        im.putpalette(pal)
        im.save(root / "SegmentationClassAug" / (name + ".png"))
        oh = np.zeros(20, np.float32)
        oh[[k, 7]] = 1
        onehot[name] = oh
        np.save(tmp_path / f"lab{k}.npy", lab)
    (lists / "val.txt").write_text("\n".join(ids) + "\n")
    np.save(lists / "cls_labels_onehot.npy", onehot)
    ds = voc.VOC12SegDataset(str(root), str(lists), split="val", stage="val")
    assert len(ds) == 2 and voc.class_list[15] == "person"
    name, image, label, cls = ds[1]
    assert name == ids[1] and image.shape == (21, 24, 3) and image.dtype == np.uint8
    assert np.array_equal(label, np.load(tmp_path / "lab1.npy")) and label[0, 0] == 255        # palette PNG -> index map
    assert np.array_equal(cls, onehot[ids[1]])
    assert np.array_equal(image, np.asarray(Image.open(root / "JPEGImages" / (ids[1] + ".jpg"))))
    names, imgs, gts, clss = ds.batch([0])
    assert imgs.shape == (1, 20, 24, 3) and gts.shape == (1, 20, 24) and clss.shape == (1, 20)
    with pytest.raises(ValueError):
        ds.batch([0, 1])
    t = voc.VOC12SegDataset(str(root), str(lists), split="val", stage="test")
    assert t[0][3].sum() == 0


# ------------------------------------------------------------------ training iteration: schedule + gradient averaging (N > 1 on gloo)
def test_poly_warmup_schedule_known_answers():
    """utils/optimizer.py PolyWarmupAdamW.step: warm-up lr_mult = 1 - (1 - s/W)(1 - ratio), then (1 - s/max)^power."""
    from excel_amd.scripts.train_voc import poly_warmup_lr
    base = 1e-3
    assert abs(poly_warmup_lr(base, 0, 50, 1000, 1e-6, 1) - 1e-9) < 1e-15
    assert abs(poly_warmup_lr(base, 1, 50, 1000, 1e-6, 1) - 2.000098e-05) < 1e-12          # value logged by the reference optimizer
    assert abs(poly_warmup_lr(base, 25, 50, 1000, 1e-6, 1) - base * (1 - 0.5 * (1 - 1e-6))) < 1e-15
    assert abs(poly_warmup_lr(base, 50, 50, 1000, 1e-6, 1) - base * 0.95) < 1e-15
    assert abs(poly_warmup_lr(base, 500, 50, 1000, 1e-6, 2) - base * 0.25) < 1e-15


def _grad_worker(rank, world, port, q):
    import torch.distributed as dist
    from excel_amd.scripts.train_voc import allreduce_mean_
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    allreduce_mean_(flat)
    q.put((rank, flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_mean_world2():
    """The one collective of a data-parallel training step: DistributedDataParallel's mean over the flat gradient buffer."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, flat in res:
        assert np.allclose(flat, np.arange(10, dtype=np.float32) * 1.5)


def test_coco_on_disk_dataset(tmp_path):
    """datasets/coco.py layout: train/val sub-directories, prefix-stripped label names, grey-scale JPEG replication."""
    from PIL import Image
    from excel_amd.datasets import coco
    root, lists = tmp_path / "COCO", tmp_path / "lists"
    for d in ("JPEGImages/val", "SegmentationClass/val"):
        (root / d).mkdir(parents=True)
    lists.mkdir()
    rs = np.random.RandomState(1)
    name = "COCO_val2014_000000000042"
    Image.fromarray(rs.randint(0, 256, (18, 22)).astype(np.uint8)).save(root / "JPEGImages/val" / (name + ".jpg"))      # grey-scale
    lab = rs.randint(0, 81, (18, 22)).astype(np.uint8)
    Image.fromarray(lab).save(root / "SegmentationClass/val" / "000000000042.png")
    oh = np.zeros(80, np.float32)
    oh[[0, 79]] = 1
    (lists / "val.txt").write_text(name + "\n")
    np.save(lists / "cls_labels_onehot.npy", {name: oh})
    ds = coco.CocoSegDataset(str(root), str(lists), split="val", stage="val")
    n, image, label, cls = ds[0]
    assert n == name and image.shape == (18, 22, 3) and np.array_equal(image[..., 0], image[..., 2])
    assert np.array_equal(label, lab) and np.array_equal(cls, oh)
    assert len(coco.class_list) == 81 and coco.class_list[1] == "person" and coco.class_list[80] == "toothbrush"


def test_infer_vit_config_from_state_dict():
    """clip/build_model.py:33-38: width / layers / patch / grid / output_dim read off the checkpoint tensors."""
    from excel_amd.clip.clip import infer_vit_config
    from oracle.vit import VitConfig, make_vit_weights
    cfg = VitConfig(width=128, layers=8, heads=2, patch=16, out_dim=64, input_resolution=64, n_surgery=5)
    got = infer_vit_config(make_vit_weights(cfg, seed=1))
    assert got == dict(width=128, layers=8, heads=2, patch=16, output_dim=64, input_resolution=64)


def test_infer_lam_resolves_real_inputs_or_refuses(tmp_path, monkeypatch):
    """tools/infer_lam.py:144-160 / model/model_excel.py:25-34: --model (a checkpoint path, or the model name under --clip_root) is what
    the tower is built from; the class + background prompts go through the checkpoint's text tower with CLIP's BPE; the decoder
    checkpoint is read only for --training_free false; real data without weights is refused (never scored with random weights)."""
    import torch
    from _clip_files import write_tiny_clip, MERGES
    from excel_amd.tools import infer_lam
    from excel_amd import clip as xclip
    monkeypatch.delenv("EXCEL_CLIP_ROOT", raising=False)
    monkeypatch.delenv("EXCEL_BPE_VOCAB", raising=False)
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    ckpt, bpe_path, full = write_tiny_clip(tmp_path)
    P = infer_lam.get_parser().parse_args
    # real data, no weights anywhere -> loud failure
    with pytest.raises(RuntimeError, match="Refusing to score real data"):
        infer_lam.resolve_model_inputs(P(["--data_folder", str(tmp_path)]))
    # explicit path
    kw = infer_lam.resolve_model_inputs(P(["--data_folder", str(tmp_path), "--model", ckpt, "--bpe_path", bpe_path]))
    assert np.array_equal(kw["state_dict"]["visual.conv1.weight"].numpy(), full["visual.conv1.weight"]) and "text_features" not in kw
    assert len(kw["tokenizer"].encoder) == 512 + len(MERGES) + 2
    ids = xclip.tokenize(["a clean origami the cat."], tokenizer=kw["tokenizer"])
    assert ids.shape == (1, 77) and int(ids[0, 0]) == 512 + len(MERGES) and int(ids.max()) == 512 + len(MERGES) + 1
    # model NAME resolved under --clip_root (where the reference's downloader would have stored ViT-B-16.pt)
    kw2 = infer_lam.resolve_model_inputs(P(["--data_folder", str(tmp_path), "--clip_root", str(tmp_path), "--bpe_path", bpe_path]))
    assert np.array_equal(kw2["state_dict"]["visual.proj"].numpy(), full["visual.proj"])
    # weights but no merges file -> the tokenizer's error, not a silent fallback
    with pytest.raises(FileNotFoundError):
        infer_lam.resolve_model_inputs(P(["--data_folder", str(tmp_path), "--model", ckpt]))
    # synthetic mode is explicit and only without a checkpoint
    kw3 = infer_lam.resolve_model_inputs(P(["--synthetic", "4"]))
    assert set(kw3) == {"state_dict", "text_features"} and kw3["text_features"].shape == (45, 512)
    # --training_free false: decoder checkpoint required, DDP prefixes stripped, encoder tensors left out
    with pytest.raises(RuntimeError, match="--model_path"):
        infer_lam.resolve_model_inputs(P(["--synthetic", "4", "--training_free", "false"]))
    dec = {"module.decoder.linear_pred.bias": torch.zeros(21), "module.decoder_fts_fuse.linear_fuse.bias": torch.ones(256),
           "module.encoder.visual.positional_embedding": torch.zeros(3, 3), "module.encoder.visual.proj": torch.zeros(2, 2)}
    torch.save(dec, tmp_path / "head.pth")
    kw4 = infer_lam.resolve_model_inputs(P(["--synthetic", "4", "--training_free", "false", "--model_path", str(tmp_path / "head.pth")]))
    assert sorted(kw4["decoder_state_dict"]) == ["decoder.linear_pred.bias", "decoder_fts_fuse.linear_fuse.bias"]


def test_checkpoint_torchscript_archive_branch(tmp_path, monkeypatch):
    """clip/clip.py:138-141: the published CLIP files are TorchScript archives; read_checkpoint takes their state_dict (every tensor
    of a scripted module with the CLIP key layout comes back bit-identical) and the harness resolves such a file like a plain one."""
    from _clip_files import write_tiny_clip_jit
    from excel_amd.clip import clip as xclip
    from excel_amd.tools import infer_lam
    monkeypatch.delenv("EXCEL_CLIP_ROOT", raising=False)
    path, full = write_tiny_clip_jit(tmp_path)
    import torch
    with pytest.raises(RuntimeError):
        torch.load(path, map_location="cpu", weights_only=True)           # it really is an archive, not a pickled dict
    sd = xclip.read_checkpoint(path)
    assert sorted(sd) == sorted(full)
    assert all(np.array_equal(sd[k].numpy(), np.asarray(full[k])) for k in full)
    kw = infer_lam.resolve_model_inputs(infer_lam.get_parser().parse_args(
        ["--data_folder", str(tmp_path), "--model", path, "--bpe_path", str(tmp_path / "bpe_tiny_vocab.txt.gz")]))
    assert np.array_equal(kw["state_dict"]["visual.conv1.weight"].numpy(), full["visual.conv1.weight"])


def test_clip_text_prompt_lists():
    """model/model_excel.py:31: 20 + 25 prompts for VOC, 80 + 23 for COCO (datasets/clip_text.py lists, shipped as data)."""
    from excel_amd.datasets import clip_text
    voc, coco = clip_text.text_prompts(21), clip_text.text_prompts(81)
    assert len(voc) == 45 and len(coco) == 103 and voc[0] == "aeroplane" and voc[14].startswith("person with clothes")
    assert voc[20:] == clip_text.BACKGROUND_CATEGORY and coco[80:] == clip_text.BACKGROUND_CATEGORY_COCO


def test_bench_power_sideline_without_rocm_smi(monkeypatch):
    """bench.py's power side-line is optional: without `rocm-smi` on the PATH it reports nothing and never touches the pipeline."""
    import shutil
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: None)
    assert bench.power_sideline(pipe=None, batch=None, seconds=0.1, B=32) is None


def test_product_reads_nothing_from_tests_dir():
    """Product data lives inside the package: no module of excel_amd (nor bench.py) names the tests/ directory as a data source."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"""["']tests["']|tests/golden|tests[\\/]""")
    bad = []
    for base, _, files in os.walk(os.path.join(root, "excel_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(base, f)).read()
                for i, line in enumerate(txt.splitlines(), 1):
                    code = line.split("#", 1)[0]
                    if pat.search(code) and "os.path.join" in code:
                        bad.append(f"{os.path.join(base, f)}:{i}")
    for i, line in enumerate(open(os.path.join(root, "bench.py")).read().splitlines(), 1):
        if pat.search(line.split("#", 1)[0]) and "os.path.join" in line:
            bad.append(f"bench.py:{i}")
    assert not bad, bad
    from excel_amd.model.load_attr import BANK_DIR, load_bank
    assert os.path.isdir(BANK_DIR) and "tests" not in os.path.relpath(BANK_DIR, root).split(os.sep)
    bank, flag = load_bank("pascal_voc", 112)
    assert tuple(bank.shape) == (512, 112)
