"""Child process of tests/test_gpu_distributed.py: ONE rank of an RCCL ("nccl" backend) process group on cuda:0 — the GPU boxes
of this pool expose a single GPU, so this is the most the harness can run here: process-group init over RCCL, an all-reduce, a
broadcast, DDP-wrapped training steps (gradient all-reduce through RCCL buckets) with the fused flat optimiser."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import distributed, optim  # noqa: E402

out_path = sys.argv[1]
rank, local_rank, world = distributed.env_rank()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
torch.distributed.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)      # world size 1 is a valid group
seen = distributed.ranks_seen(dev)
t = torch.arange(8, dtype=torch.float32, device=dev)
torch.distributed.broadcast(t, src=0)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, bias=False), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Flatten(),
                          torch.nn.Linear(8 * 6 * 6, 4)).to(dev)
ddp = distributed.wrap_ddp(net, dev)
opt = optim.FusedAdamOneCycle(ddp, wd=0.01)
losses = []
for it in range(3):
    opt.zero_grad()
    loss = ddp(torch.randn(5, 3, 8, 8, device=dev)).pow(2).mean()
    loss.backward()
    opt.clip_grad_norm(10.0)
    opt.step()
    losses.append(float(loss))
json.dump({"backend": torch.distributed.get_backend(), "seen": seen, "ddp": type(ddp).__name__, "losses": losses,
           "slowest": distributed.max_over_ranks(1.5, dev)}, open(out_path, "w"))
distributed.barrier(dev)
distributed.finalize()
