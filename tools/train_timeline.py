"""Host against device clock of one steady-state training step (batch 16): when the HOST returns from forward / backward / optimiser
(the launches are asynchronous) and when the DEVICE reaches the same points (events).  A host that returns from the forward at
about the time the device finishes it is launch-bound there (the device waits for launches).
    python tools/train_timeline.py [--detail]   (--detail: the same per module of the forward)"""
import os
import sys
import time

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
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, time.perf_counter(), e))


if "--detail" in sys.argv:
    for name, mod in model.named_children():
        mod.register_forward_pre_hook(lambda m, a, n=name: mark("> " + n))
feed = optim.prefetching(model, (dict(pool[i % 2]) for i in range(8)))
for it, b in enumerate(feed):
    if it == 7:
        break
    rec = it == 5
    if rec:
        torch.cuda.synchronize()
        marks.clear()
        mark("start")
        if "--syncdebug" in sys.argv:
            torch.cuda.set_sync_debug_mode("warn")      # torch prints a warning (with the python line) for every synchronising call
    sched.step(it); model.train(); opt.zero_grad()
    ret, _, _ = model(b)
    loss = ret["loss"].mean()
    if rec: mark("forward")
    loss.backward()
    if rec: mark("backward")
    opt.clip_grad_norm(cfg.OPTIMIZATION.GRAD_NORM_CLIP); opt.step(); model.update_global_step()
    if rec:
        torch.cuda.set_sync_debug_mode("default")
        mark("optimiser")
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        out = list(marks)
t0, e0 = out[0][1], out[0][2]
print(f"{'point':28s} {'host ms':>9s} {'device ms':>10s}")
for name, t, e in out:
    print(f"{name:28s} {1e3 * (t - t0):9.1f} {e0.elapsed_time(e):10.1f}")
print(f"{'all done (synchronised)':28s} {1e3 * (t_end - t0):9.1f}")
