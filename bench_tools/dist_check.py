import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/swift-homomorphic-encryption_amd")
import torch, torch.distributed as dist
from heamd import sharding
rank, local_rank, world = sharding.rank_and_world()
torch.cuda.set_device(local_rank)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
dist.barrier()
print("max_over_ranks", sharding.max_over_ranks(1.5, device="cuda"))
x = torch.arange(12, dtype=torch.int64, device="cuda").view(3, 2, 2)
out = torch.empty((world, 3, 2, 2), dtype=torch.int64, device="cuda")
dist.all_gather_into_tensor(out.view(-1), x.view(-1))
torch.cuda.synchronize()
print("gather ok", bool((out[0] == x).all()))
dist.destroy_process_group()
