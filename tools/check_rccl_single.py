"""Single-rank RCCL sanity check of the calls the data-parallel path makes (init with device_id, all_reduce on the flat
gradient buffer, broadcast, barrier) -- the 1-GPU boxes cannot run more than one nccl rank."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = torch.arange(1029670, dtype=torch.float32, device="cuda")
ref = g.clone()
dist.all_reduce(g, op=dist.ReduceOp.SUM)
dist.broadcast(g, src=0)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(g, ref)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rccl single-rank ok", float(t))
dist.destroy_process_group()
