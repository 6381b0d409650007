"""Child process of tests/test_distributed_gloo.py::test_launch_local_*: one rank of a gloo group started by
hvpr_amd.distributed.launch_local (the launcher bench.py --gpus N and tools/bench_train.py --gpus N use over RCCL)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import distributed  # noqa: E402

out_dir, mode = sys.argv[1], sys.argv[2]
if mode == "fail_early" and int(os.environ["RANK"]) == 1:
    sys.exit(4)         # before the rendezvous: rank 0 would wait in init_process_group for ever
rank, local_rank, world = distributed.init("gloo")
if mode == "fail" and rank == 1:
    sys.exit(3)
dev = torch.device("cpu")
distributed.barrier(dev)
res = {"rank": rank, "local_rank": local_rank, "world": world, "seen": distributed.ranks_seen(dev),
       "times": distributed.gather_floats(10.0 + rank, dev), "slowest": distributed.max_over_ranks(10.0 + rank, dev),
       "master": os.environ["MASTER_ADDR"]}
json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
distributed.barrier(dev)
distributed.finalize()
