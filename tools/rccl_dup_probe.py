"""Does RCCL accept two ranks on one device?  (Run under torchrun with 2 processes on a 1-GPU box.)"""
import os

import torch
import torch.distributed as dist

rank = int(os.environ['RANK'])
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=rank, world_size=int(os.environ['WORLD_SIZE']), device_id=dev)
try:
    x = torch.ones(4, device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print('rank', rank, 'all_reduce ->', x.tolist())
except Exception as e:                                   # noqa: BLE001 - a probe: report whatever RCCL says
    print('rank', rank, 'refused:', str(e).splitlines()[0][:200])
finally:
    dist.destroy_process_group()
