"""RCCL on the hardware: the path's ONE collective - the all-gather of the per-rank [nc,nc] int64 confusion matrices
(tools/infer_lam.py:133,164-167 in the reference: init_process_group + rank-strided shards) - sent through the real RCCL backend on the
GPU box.  The box has one GPU, so the group has one rank; the call path (torch.distributed "nccl" = RCCL, device tensors, the same
`gather_hists`) is the one N > 1 uses.  Each case runs in its own process: a process group is process-global state.
The N > 1 control flow itself is covered on CPU (gloo, world 2 / 4 / 8) in tests/test_host_cpu.py.
Round 6: `test_bench_and_harness_over_rccl[2]` ARMS ITSELF on a box with two or more GPUs (bench.py --gpus 2 launched bare, and the
infer_lam harness at world 2 against its own world-1 run) and is skipped with the reason on a one-GPU box; its world-1 twin runs the
same commands and the same checks everywhere, so the armed case cannot fail on plumbing the day a multi-GPU box runs `pytest -m gpu`."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    e = dict(os.environ)
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    return e


_ALLGATHER = r"""
import json, os, torch, torch.distributed as dist
from excel_amd.tools.infer_lam import gather_hists, shard_indices
from excel_amd import ops
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d", rank=0, world_size=1)
nc = %d
# a real device-side confusion matrix from the library's own kernel, then the collective
g = torch.Generator().manual_seed(5)
gt = torch.randint(0, nc, (100000,), generator=g, dtype=torch.uint8)
gt[torch.rand(100000, generator=g) < 0.02] = 255
pr = torch.randint(0, nc, (100000,), generator=g, dtype=torch.uint8)
hist = torch.zeros((nc, nc), dtype=torch.int64, device="cuda")
ops.confusion_accumulate(gt.cuda(), pr.cuda(), nc, hist)
per_rank, total = gather_hists(hist)
t = torch.tensor([1.25e-5], dtype=torch.float64, device="cuda")      # the numerics ladder's shared verdict (all-reduce MAX)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
keep = gt != 255
ref = torch.bincount(gt[keep].long() * nc + pr[keep].long(), minlength=nc * nc).reshape(nc, nc)
print(json.dumps({"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "shape": list(per_rank.shape),
                  "equal": bool((per_rank[0].cpu() == ref).all() and (total.cpu() == ref).all()), "mass": int(total.sum()),
                  "on_device": per_rank.is_cuda, "allreduce": float(t), "shard": [int(i) for i in shard_indices(5, 0, 1)]}))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("nc", [21, 81])
def test_rccl_allgather_of_confusion_matrix_one_rank(nc):
    """[21,21] (VOC, 3.5 KB) and [81,81] (COCO, 52 KB) int64 through dist.all_gather on the nccl (= RCCL) backend."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, "-c", _ALLGATHER % (_free_port(), nc)], capture_output=True, text=True, timeout=600,
                       env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])      # (librccl prints a version banner to stdout)
    print("rccl:", out)
    assert out["backend"] == "nccl" and out["rccl_ranks"] == 1 and out["shape"] == [1, nc, nc]
    assert out["equal"] and out["on_device"] and out["mass"] > 90000 and out["allreduce"] == 1.25e-5 and out["shard"] == [0, 1, 2, 3, 4]


def test_bench_under_torch_distributed_run_one_rank():
    """The driver's N > 1 launch line with --nproc-per-node 1: bench.py reads RANK / WORLD_SIZE / MASTER_* from the launcher, brings
    up RCCL, runs the sharded step and gathers the matrix through it; the JSON line says how many ranks the collective spanned."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--cpu-images", "0", "--ragged-images", "0", "--power-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    print("bench under torchrun:", {k: out[k] for k in ("value", "n_gpus", "rccl_ranks", "rccl", "per_rank_hist_mass")})
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["rccl"].startswith("all_gather")
    assert len(out["per_rank_hist_mass"]) == 1 and out["per_rank_hist_mass"][0] > 0 and out["value"] > 0


def _infer_lam(world, json_path, n_images=24, batch=4):
    """The harness on `n_images` seeded synthetic 448^2 samples, one rank per GPU under torch.distributed.run (world 1: the same launcher
    line with one rank) -> rank 0's JSON record."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "excel_amd.tools.infer_lam", "--synthetic", str(n_images), "--batch_size", str(batch),
           "--json_out", json_path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return json.load(open(json_path))


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("world", [1, 2])
def test_bench_and_harness_over_rccl(world, tmp_path):
    """(a) `python bench.py --gpus W` launched BARE (W = 2: it re-executes itself under torch.distributed.run, bench.self_launch_argv; the
    driver's own N > 1 line is the one-rank test above): the JSON line must say that the collective spanned W ranks and carry W non-zero
    per-rank histogram masses.  (b) tools/infer_lam.py:133,164-167: the harness at world W over RCCL - rank-strided shards, ONE all-gather of
    the [21,21] int64 confusion matrix - must gather exactly the matrix a one-rank run computes over the same list (every image scored
    once, by exactly one rank)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = torch.cuda.device_count()
    if n < world:
        pytest.skip(f"needs {world} GPUs for a {world}-rank RCCL group, this box has {n} (the multi-rank control flow is covered over gloo at "
                    "world 2 / 4 / 8 in tests/test_host_cpu.py, the RCCL call path with one rank above)")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--cpu-images", "0",
           "--ragged-images", "0", "--power-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("bench --gpus", world, {k: out[k] for k in ("value", "n_gpus", "rccl_ranks", "rccl", "per_rank_hist_mass")})
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["per_rank_hist_mass"]) == world and all(m > 0 for m in out["per_rank_hist_mass"])
    one = _infer_lam(1, str(tmp_path / "w1.json"))
    assert one["world"] == 1 and one["images_total"] == 24 and len(one["per_rank_hist_mass"]) == 1
    if world == 1:
        assert sum(map(sum, one["hist_total"])) == one["per_rank_hist_mass"][0] > 0
        return
    many = _infer_lam(world, str(tmp_path / f"w{world}.json"))
    assert many["world"] == world and many["rccl_ranks"] == world and many["images_total"] == 24
    assert len(many["per_rank_hist_mass"]) == world and all(m > 0 for m in many["per_rank_hist_mass"])
    assert sum(many["per_rank_hist_mass"]) == one["per_rank_hist_mass"][0]
    assert many["hist_total"] == one["hist_total"]              # the gathered matrix == the one-rank matrix of the same list, integer-exact
