"""Child process of tests/test_gpu_distributed.py: ONE rank of an RCCL ("nccl" backend) process group on cuda:0 — the GPU boxes
of this pool expose a single GPU, so this is the most the harness can run here: process-group init over RCCL, an all-reduce, a
broadcast, and the REAL detector (MixAnchor_Memory, hvpr_car, batch 2: custom autograd.Functions, weights used 6-18x per forward,
multi-call BatchNorm buffers updated in place by the kernels, the point-index prefetch on a side stream) inside
DistributedDataParallel (reference: tools/train.py:143-145) with FusedAdamOneCycle on its flat buffers — three training steps,
compared with the same three steps without DDP."""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, distributed, optim, synthetic, synthetic_weights  # noqa: E402
from hvpr_amd.config import hvpr_car_cfg  # noqa: E402

out_path = sys.argv[1]
rank, local_rank, world = distributed.env_rank()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
torch.distributed.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)      # world size 1 is a valid group
seen = distributed.ranks_seen(dev)
t = torch.arange(8, dtype=torch.float32, device=dev)
torch.distributed.broadcast(t, src=0)

cfg = hvpr_car_cfg()
base = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
synthetic_weights.load_synthetic(base, seed=31, cls_bias=-4.595)
base = base.to(dev).train()
rng = np.random.default_rng(31)


def make_batch(seed0, B=2):
    frames = [synthetic.hvpr_frame(seed0 + b, shuffle=True) for b in range(B)]
    pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)])
    gt = np.zeros((B, 6, 8), np.float32)
    gt[..., 0] = rng.uniform(5, 42, (B, 6)); gt[..., 1] = rng.uniform(-15, 15, (B, 6)); gt[..., 2] = rng.uniform(-1.2, -0.8, (B, 6))
    gt[..., 3:6] = np.array([3.9, 1.6, 1.56], np.float32) * rng.uniform(0.9, 1.1, (B, 6, 3))
    gt[..., 6] = rng.uniform(-3, 3, (B, 6)); gt[..., 7] = 1
    return {"points": torch.from_numpy(pts).to(dev), "gt_boxes": torch.from_numpy(gt).to(dev), "batch_size": B}


batches = [make_batch(400 + 10 * i) for i in range(3)]


def run(wrap, sync_bn=False):
    model = copy.deepcopy(base)
    if sync_bn:          # tools/train.py:119-120 (--sync_bn): every train-mode BatchNorm of the model takes global-batch statistics
        model = distributed.convert_sync_batchnorm(model)
    m = distributed.wrap_ddp(model, dev) if wrap else model
    opt = optim.build_optimizer(m, cfg.OPTIMIZATION)
    assert isinstance(opt, optim.FusedAdamOneCycle)
    sched, _ = optim.build_scheduler(opt, total_iters_each_epoch=10, total_epochs=1, last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    losses, ptrs = [], []
    for it, b in enumerate(optim.prefetching(m, (dict(x) for x in batches))):
        loss, _ = optim.train_step(m, opt, sched, b, it, cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        losses.append(float(loss))
        ptrs.append([p.grad.data_ptr() for p in opt.params])
    torch.cuda.synchronize()
    own = [g.data_ptr() for g in opt._grad_views]
    return (type(m).__name__, losses, {k: v.detach().clone() for k, v in model.state_dict().items()},
            all(p == own for p in ptrs))


worst_keys = {}


def diff(a, b, tag=""):
    worst = 0.0
    for k in a:
        if a[k].dtype.is_floating_point:
            d = float((a[k] - b[k]).abs().max() / (b[k].abs().max() + 1e-12))
            if d > worst:
                worst, worst_keys[tag] = d, k
        else:
            assert torch.equal(a[k], b[k]), k
    return worst


name_p, loss_p, sd_p, flat_p = run(False)
name_d, loss_d, sd_d, flat_d = run(True)
name_q, loss_q, sd_q, _ = run(False)              # run-to-run: every sum of the step has a fixed order, so a rerun is bit-identical
# the whole detector with SyncBatchNorm inside DDP: at world size 1 the all-reduces are identities, the statistics come out of float64
# torch sums / the library's hook instead of the finalize kernels — same numbers up to fp32 round-off
from hvpr_amd import conv_train  # noqa: E402
name_s, loss_s, sd_s, flat_s = run(True, sync_bn=True)
hook_calls = conv_train._sync.get("hook_calls", 0)
conv_train.set_sync_batchnorm(None)
json.dump({"backend": torch.distributed.get_backend(), "seen": seen, "ddp": name_d, "plain": name_p, "losses_plain": loss_p, "losses_ddp": loss_d,
           "losses_rerun": loss_q, "state_diff_ddp_vs_plain": diff(sd_d, sd_p, "ddp"), "state_diff_rerun_vs_plain": diff(sd_q, sd_p, "rerun"),
           "grads_in_flat_buffer_plain": flat_p, "grads_in_flat_buffer_ddp": flat_d,
           "losses_sync_bn": loss_s, "state_diff_sync_bn_vs_plain": diff(sd_s, sd_p), "grads_in_flat_buffer_sync_bn": flat_s,
           "sync_bn_hook_calls": hook_calls, "worst_keys": worst_keys,
           "slowest": distributed.max_over_ranks(1.5, dev)}, open(out_path, "w"))
distributed.barrier(dev)
distributed.finalize()
