"""GPU: the RCCL side of the multi-GPU harness, as far as a one-GPU box allows — a one-rank "nccl" process group started by
distributed.launch_local (the launcher of bench.py --gpus N / tools/bench_train.py --gpus N): group init over RCCL, all-reduce,
broadcast and DDP training steps with the fused flat optimiser.  The N > 1 logic is covered on CPU by the gloo tests."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_group_real_detector_in_ddp_with_fused_optimizer(tmp_path):
    """Row a15 on the hardware this pool has: the real MixAnchor_Memory inside DistributedDataParallel over a one-rank RCCL group,
    FusedAdamOneCycle + the point-index prefetch, three steps — against the same steps without DDP.  The gradients stay in the
    optimiser's flat buffer under DDP (no re-pointing); DDP, plain and a rerun of plain agree bit for bit (losses and every parameter /
    buffer after three steps)."""
    from hvpr_amd import distributed
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    out = tmp_path / "rank0.json"
    rc = distributed.launch_local(1, [worker, str(out)], timeout=900)
    assert rc == 0
    r = json.load(open(out))
    print(r)
    assert r["backend"] == "nccl" and r["seen"] == 1 and r["ddp"] == "DistributedDataParallel" and r["plain"] == "MixAnchor_Memory"
    assert r["slowest"] == 1.5
    assert r["grads_in_flat_buffer_plain"] and r["grads_in_flat_buffer_ddp"]
    assert all(np.isfinite(r["losses_ddp"])) and len(r["losses_ddp"]) == 3
    # round 4: the scattering gradients are fixed-order sums (hvpr_segment_sum_rows_f32), nothing in the step is an atomic float add any
    # more: a rerun of the plain steps reproduces them BIT FOR BIT, and so do the steps inside DistributedDataParallel (one rank: the
    # all-reduce is the identity; the averaging by world size 1 and the bucket copies are exact)
    assert r["losses_rerun"] == r["losses_plain"], r
    assert r["losses_ddp"] == r["losses_plain"], r
    assert r["state_diff_rerun_vs_plain"] == 0.0 and r["state_diff_ddp_vs_plain"] == 0.0, r
    # the same three steps with distributed.convert_sync_batchnorm(model) (reference --sync_bn) inside DDP: every BatchNorm goes through
    # its all-reduce route (torch sums for the conv / point BatchNorms, the library's hook for the PFN layers and SpatialAttention:
    # per step 2 + 4 calls of the PFN kernels and 3 x 2 of the gates) — at world size 1 the same numbers up to fp32 round-off
    assert r["grads_in_flat_buffer_sync_bn"] and len(r["losses_sync_bn"]) == 3
    assert r["sync_bn_hook_calls"] == 3 * (6 + 6), r["sync_bn_hook_calls"]
    for a, b in zip(r["losses_sync_bn"], r["losses_plain"]):
        assert abs(a - b) <= 2e-4 * abs(b), (r["losses_sync_bn"], r["losses_plain"])
    # (no bar on the parameters themselves: Adam normalises every gradient, so a parameter whose exact gradient is zero — a convolution
    # bias in front of a batch-statistics BatchNorm — moves by +-lr on round-off alone; the losses of steps 2 and 3 are the check that
    # the updated models are the same model)


def test_bench_train_step_ddp_line_over_one_rank_rccl(tmp_path):
    """`bench.py --gpus N` (N > 1) also times the config-4 training step inside DistributedDataParallel, so that the driver's
    scaling run measures the RCCL gradient all-reduce (reference tools/train.py:143-145) without a code change.  Here: the same
    function over a one-rank RCCL group (all this pool can run), batch 2."""
    from hvpr_amd import distributed
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bench_ddp_worker.py")
    out = tmp_path / "ddp_line.json"
    rc = distributed.launch_local(1, [worker, str(out), "2"], timeout=900)
    assert rc == 0
    r = json.load(open(out))
    print(r)
    assert r["backend"] == "nccl" and r["ranks"] == 1 and r["parallelism"] == "dp1"
    assert 55.0 < r["allreduce_MB"] < 70.0                      # ~15.5 M fp32 gradients
    assert r["ms_per_step"] > 0 and r["frames_per_s_global"] > 0
    assert all(np.isfinite(r["loss_first_last"]))


def test_sync_batchnorm_path_on_one_rank_rccl(tmp_path):
    """SyncBatchNorm (reference tools/train.py:119-120) over the own BatchNorm kernels: conv -> bn_relu (statistics from the Winograd
    kernel's per-tile sums) -> fused SFM step, forward + backward, with the statistics all-reduced over a one-rank RCCL group against
    the per-rank path.  At world size 1 the two compute the same numbers by different routes (float64 torch sums + split backward
    vs the finalize kernel + fused backward): agreement to fp32 round-off.  The N > 1 arithmetic of the all-reduces is covered on
    CPU/gloo (test_sync_batchnorm_two_ranks_batch1_equal_one_process_batch2)."""
    from hvpr_amd import distributed
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_syncbn_worker.py")
    out = tmp_path / "syncbn.json"
    rc = distributed.launch_local(1, [worker, str(out)], timeout=600)
    assert rc == 0
    r = json.load(open(out))
    print(r)
    assert r["backend"] == "nccl" and r["world"] == 1
    for k, v in r["rel"].items():
        assert v <= 2e-6, (k, v)
    # SpatialAttention's BatchNorm went through the library's hook (hvpr_set_batchnorm_allreduce): one call forward, one backward; the
    # N > 1 arithmetic of the hooked kernels is covered by the emulated two-rank tests of tests/test_gpu_train_ops.py
    assert r["hook_calls"] == 2, r["hook_calls"]
