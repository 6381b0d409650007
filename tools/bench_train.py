"""Config 3 of SURVEY.md §8d: hvpr_car full training step (a1..a15: voxelize, point stream, VFE, memory, scatter, two-stream
BEV backbone, head, target assignment, losses, backward, clip, Adam-onecycle), B frames per GPU, synthetic frames with 8
random car boxes each.  One process per GPU; under torch.distributed.run the model is wrapped in DDP (RCCL all-reduce).

    python tools/bench_train.py --batch 16 --steps 10 --warmup 3
Prints ONE JSON line (steps/s and frames/s).  Not the headline bench (that is bench.py, config 2)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hvpr_amd import detector, distributed, optim, synthetic, synthetic_weights  # noqa: E402
from hvpr_amd.config import hvpr_3class_cfg, hvpr_car_cfg  # noqa: E402


SIZES = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], np.float32)


def gt_boxes(B, rng, per_frame=8, n_class=1):
    g = np.zeros((B, per_frame, 8), np.float32)
    cls = rng.integers(0, n_class, (B, per_frame))
    g[..., 0] = rng.uniform(3, 44, (B, per_frame)); g[..., 1] = rng.uniform(-17, 17, (B, per_frame))
    g[..., 2] = rng.uniform(-1.2, -0.8, (B, per_frame))
    g[..., 3:6] = SIZES[cls] * rng.uniform(0.9, 1.1, (B, per_frame, 3))
    g[..., 6] = rng.uniform(-np.pi, np.pi, (B, per_frame)); g[..., 7] = cls + 1
    return g


def make_batch(seed0, B, device, rng, n_class=1):
    frames = [synthetic.hvpr_frame(seed0 + b, shuffle=True) for b in range(B)]
    pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)])
    return {"points": torch.from_numpy(pts).to(device), "gt_boxes": torch.from_numpy(gt_boxes(B, rng, n_class=n_class)).to(device), "batch_size": B}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks of this node; without WORLD_SIZE in the env this process launches them")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cfg", choices=["car", "3class"], default="car", help="car = config 3, 3class = config 4 of SURVEY.md §8d")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark: MIOpen searches its convolution solvers once per shape")
    ap.add_argument("--channels-last", action="store_true", help="backbone + head weights in channels_last memory format")
    ap.add_argument("--no-prefetch", action="store_true", help="do not compute the next batch's point-stream indices beside the current step")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # launcher: one fresh child per GPU (tools/train.py:61-70)
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench_train.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        sys.exit(distributed.launch_local(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank, local_rank, world = distributed.env_rank()
    if torch.cuda.device_count() < max(world, args.gpus):
        raise SystemExit(f"bench_train.py: {max(world, args.gpus)} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench_train.py needs an MI355X")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed.init("nccl", device)
    cfg = hvpr_car_cfg() if args.cfg == "car" else hvpr_3class_cfg()
    model = detector.build_network(cfg.MODEL, len(cfg.CLASS_NAMES), detector.SyntheticDataset(cfg, training=True))
    synthetic_weights.load_synthetic(model, seed=0, cls_bias=-4.59511985013459)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    model = model.to(device)
    if args.channels_last:
        model.backbone_2d.to(memory_format=torch.channels_last)
        model.dense_head.to(memory_format=torch.channels_last)
    model = distributed.wrap_ddp(model, device)
    opt = optim.build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = optim.build_scheduler(opt, total_iters_each_epoch=args.steps + args.warmup, total_epochs=1, last_epoch=-1,
                                     optim_cfg=cfg.OPTIMIZATION)
    rng = np.random.default_rng(rank)
    pool = [make_batch(rank * 1000 + 100 * i, args.batch, device, rng, len(cfg.CLASS_NAMES)) for i in range(2)]
    losses = []
    # batches come one ahead (optim.prefetching): the generator enqueues the point-stream index kernels of batch i + 1 on a side
    # stream BEFORE it yields batch i.  The clock therefore starts right after step warmup - 1 (so that the index work of batch
    # warmup + 1 is enqueued inside the timed region) and stops after step warmup + steps - 1: `steps` steps and `steps` index
    # plans are inside, nothing is moved out of the timed region.  --no-prefetch: the plain loop.
    assert args.warmup >= 1
    n_all = args.warmup + args.steps
    feed = (dict(pool[i % 2]) for i in range(n_all + 1))
    batches = feed if args.no_prefetch else optim.prefetching(model, feed)
    t0 = 0.0
    for it, b in enumerate(batches):
        if it == n_all:
            break
        loss, _ = optim.train_step(model, opt, sched, b, it, cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        if it == args.warmup - 1:
            distributed.barrier(device)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if it >= args.warmup:
            losses.append(loss)
    distributed.barrier(device)
    torch.cuda.synchronize()
    dt = distributed.max_over_ranks(time.perf_counter() - t0, device)
    seen = distributed.ranks_seen(device)
    if rank == 0:
        print(json.dumps({"metric": "hvpr_car training steps/s (fwd+bwd+Adam-onecycle)", "value": round(args.steps / dt, 3),
                          "unit": "steps/s", "frames_per_s": round(world * args.batch * args.steps / dt, 2), "n_gpus": world, "rccl_ranks_seen": seen,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 1),
                          "dtype": "f32", "data": "synthetic", "scaling": "weak",
                          "config": {"workload": f"hvpr_{args.cfg}.yaml full train step a1..a15, batch={args.batch}/GPU, 8 GT boxes/frame",
                                     "parallelism": f"dp{world}" if world > 1 else "single"},
                          "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)],
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
    distributed.finalize()


if __name__ == "__main__":
    main()
