"""Child process of tests/test_gpu_distributed.py: bench.py's multi-GPU training line (`train_step_ddp`, what `bench.py --gpus N`
times at N > 1: BASELINE.json configs[3], hvpr 3-class inside DistributedDataParallel over RCCL) on the ONE rank this pool has."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hvpr_amd import distributed  # noqa: E402

rank, local_rank, world = distributed.env_rank()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
torch.distributed.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)      # world size 1 is a valid group
line = bench.train_step_line(dev, "3class", int(sys.argv[2]), steps=2, warmup=1, ddp=True)
line["backend"] = torch.distributed.get_backend()
json.dump(line, open(sys.argv[1], "w"))
distributed.finalize()
