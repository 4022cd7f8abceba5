"""Can RCCL run two ranks on ONE GPU (the only multi-rank NCCL test a 1-GPU box allows)?  torchrun --nproc-per-node 2 this file."""
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
r = dist.get_rank()
x = torch.full((4,), float(r + 1), device="cuda")
dist.all_reduce(x)
g = [torch.empty(3, device="cuda") for _ in range(2)]
dist.all_gather(g, torch.full((3,), float(r), device="cuda"))
dist.broadcast(x, 0)
torch.cuda.synchronize()
print("rank", r, x.tolist(), [t.tolist() for t in g], flush=True)
dist.destroy_process_group()
