"""Which torch operators launch the small kernels of a training step: torch.profiler over one steady-state step, operators by call count."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_train import make_batch  # noqa: E402
from hvpr_amd import detector, optim, synthetic_weights  # noqa: E402
from hvpr_amd.config import hvpr_car_cfg  # noqa: E402

dev = torch.device("cuda", 0)
cfg = hvpr_car_cfg()
model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
synthetic_weights.load_synthetic(model, seed=0, cls_bias=-4.59511985013459)
model = model.to(dev)
opt = optim.build_optimizer(model, cfg.OPTIMIZATION)
sched, _ = optim.build_scheduler(opt, total_iters_each_epoch=20, total_epochs=1, last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
rng = np.random.default_rng(0)
pool = [make_batch(100 * i, 16, dev, rng, 1) for i in range(2)]
feed = optim.prefetching(model, (dict(pool[i % 2]) for i in range(6)))
for it, b in enumerate(feed):
    if it == 5:
        break
    if it == 4:
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=False) as prof:
            optim.train_step(model, opt, sched, b, it, cfg.OPTIMIZATION.GRAD_NORM_CLIP)
            torch.cuda.synchronize()
    else:
        optim.train_step(model, opt, sched, b, it, cfg.OPTIMIZATION.GRAD_NORM_CLIP)
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
print(f"{'operator':60s} {'calls':>6s} {'device ms':>10s} {'host ms':>9s}")
for e in rows[:45]:
    print(f"{e.key[:60]:60s} {e.count:6d} {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)) / 1e3:10.2f} {e.self_cpu_time_total / 1e3:9.2f}")
