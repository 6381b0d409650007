"""GPU: the RCCL side of the multi-GPU harness, as far as a one-GPU box allows — a one-rank "nccl" process group started by
distributed.launch_local (the launcher of bench.py --gpus N / tools/bench_train.py --gpus N): group init over RCCL, all-reduce,
broadcast and DDP training steps with the fused flat optimiser.  The N > 1 logic is covered on CPU by the gloo tests."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_group_ddp_and_fused_optimizer(tmp_path):
    from hvpr_amd import distributed
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    out = tmp_path / "rank0.json"
    rc = distributed.launch_local(1, [worker, str(out)], timeout=600)
    assert rc == 0
    r = json.load(open(out))
    assert r["backend"] == "nccl" and r["seen"] == 1 and r["ddp"] == "DistributedDataParallel" and r["slowest"] == 1.5
    assert all(np.isfinite(r["losses"]))
