"""Probe: can two ranks of torch.distributed's nccl (= RCCL) backend share ONE GPU on this box?  (The builder's lease has a single
MI355X; a positive answer lets the data-parallel path -- FlatAdam's one all-reduce -- execute for real.)  Run under `timeout`."""
import os
import sys
import torch
import torch.distributed as dist

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda:0'))
t = torch.full((1 << 20,), float(rank + 1), device='cuda:0')
dist.all_reduce(t)
torch.cuda.synchronize()
print(f'rank {rank}: all_reduce ok, value {t[0].item()} (expect {world * (world + 1) / 2})', flush=True)
dist.destroy_process_group()
