"""One training step under torch.profiler: the torch (non-hvpr) kernels by (op, input shapes), to find glue worth replacing."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, optim, synthetic_weights
from hvpr_amd.config import hvpr_car_cfg
from tools.bench_train import make_batch
dev = torch.device("cuda:0")
cfg = hvpr_car_cfg()
model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
synthetic_weights.load_synthetic(model, seed=0, cls_bias=-4.595)
model = model.to(dev)
opt = optim.build_optimizer(model, cfg.OPTIMIZATION)
sched, _ = optim.build_scheduler(opt, 10, 1, -1, cfg.OPTIMIZATION)
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
b = make_batch(0, B, dev, rng)
for it in range(2):
    optim.train_step(model, opt, sched, dict(b), it, 10)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    optim.train_step(model, opt, sched, dict(b), 2, 10)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t > 0 and e.key.startswith("aten::"):
        rows.append((t / 1e3, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {tot:.2f} ms")
for r in rows[:45]:
    print(f"{r[0]:8.3f} ms  x{r[1]:<4d} {r[2]:28s} {r[3]}")
