"""RCCL smoke (single rank under torchrun): the collective calls bench.py / infer_lam.py make for N > 1."""
import os, torch, torch.distributed as dist
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dist.init_process_group(backend="nccl")
h = torch.arange(21 * 21, dtype=torch.int64, device="cuda").reshape(21, 21)
parts = [torch.empty_like(h) for _ in range(dist.get_world_size())]
dist.all_gather(parts, h)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", dist.get_world_size(), int(parts[0].sum()), float(t))
dist.destroy_process_group()
