"""Child process of tests/test_gpu_distributed.py: SyncBatchNorm over the library's own BatchNorm kernels on ONE rank of an RCCL group
(all this pool can run): with the process group set (distributed.convert_sync_batchnorm) conv_train.bn_relu and the fused SFM step
take the all-reduce path — per-rank sums -> float64 all-reduce -> statistics, and the split backward (hvpr_bn_relu_bwd_sums_nhwc_f32 ->
all-reduce -> hvpr_bn_relu_bwd_apply_nhwc_f32) — and must agree with the per-rank path, which at world size 1 computes the same thing.
SpatialAttention's BatchNorm (inside one library call) goes through the library's hook: an RCCL all-reduce on the library's own device
doubles (forward and backward: two hook calls)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import conv_train as ct, distributed  # noqa: E402

out_path = sys.argv[1]
rank, local_rank, world = distributed.env_rank()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
torch.distributed.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
g = torch.Generator().manual_seed(11)
N, H, W, C = 2, 24, 40, 64
x0 = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).to(dev)
wconv = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
gate = torch.rand(N, H, W, 1, generator=g).to(dev)
wout = torch.randn(N, H, W, C, generator=g).to(dev)


def run(sync):
    ct.set_sync_batchnorm(True if sync else None)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
    bn2 = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(dev)
    x = x0.clone().requires_grad_(True)
    w = wconv.clone().requires_grad_(True)
    z, partials = ct.conv(x, w, 1, stats=True)
    y = ct.bn_relu(z, bn, partials=partials)                 # statistics from the Winograd kernel's per-tile sums
    y = ct.sfm_step(y, w, bn2, gate)                         # the fused SFM step (its own BatchNorm + gate + residual)
    # SpatialAttention's BatchNorm sits inside ONE library call: it reaches the all-reduce through hvpr_set_batchnorm_allreduce
    gw = (torch.linspace(-0.4, 0.4, 18).view(1, 2, 3, 3)).to(dev).requires_grad_(True)
    gb, gg, gbe = (torch.tensor([v], device=dev, requires_grad=True) for v in (0.3, 1.3, -0.2))
    sgate, smean, svar = ct.spatial_gate_train(y, gw, gb, gg, gbe, 1e-3)
    ((y * wout).sum() + (sgate * wout[..., :1]).sum()).backward()
    torch.cuda.synchronize()
    res = {"y": y.detach(), "dx": x.grad, "dw": w.grad, "dg": bn.weight.grad, "db": bn.bias.grad, "dg2": bn2.weight.grad,
           "rm": bn.running_mean.clone(), "rv": bn.running_var.clone(), "rv2": bn2.running_var.clone(),
           "sgate": sgate.detach(), "smean": smean.clone(), "svar": svar.clone(), "sdw": gw.grad, "sdgamma": gg.grad}
    ct.set_sync_batchnorm(None)
    return res


a, b = run(False), run(True)
rel = {k: float((a[k] - b[k]).norm() / a[k].norm().clamp_min(1e-30)) for k in a}
json.dump({"backend": torch.distributed.get_backend(), "world": world, "rel": rel, "hook_calls": ct._sync.get("hook_calls", 0)},
          open(out_path, "w"))
distributed.finalize()
