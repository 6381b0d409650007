"""bf16x3 convolution experiment: accuracy vs the exact fp32 kernel and time, on the hvpr_car 3x3 layer shapes."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import kernels
from hvpr_amd._lib import check, lib

DEV = "cuda:0"


def split_np(w):
    """fp32 array -> (hi, lo) uint16 arrays of bf16 bits, round to nearest even."""
    t = torch.from_numpy(np.ascontiguousarray(w, np.float32))
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi.view(torch.int16).numpy().view(np.uint16), lo.view(torch.int16).numpy().view(np.uint16)


def pack_w(weight, scale, cout_pad):
    """(Cout, Cin, 3, 3) -> [9, Cin/8, 2, cout_pad, 8] uint16."""
    cout, cin = weight.shape[:2]
    w = (weight * scale.view(-1, 1, 1, 1)).cpu().numpy()
    hi, lo = split_np(w)
    out = np.zeros((9, cin // 8, 2, cout_pad, 8), np.uint16)
    for k, part in enumerate((hi, lo)):
        p = part.transpose(2, 3, 1, 0).reshape(9, cin // 8, 8, cout)      # tap, chunk, ci, co
        out[:, :, k, :cout, :] = p.transpose(0, 1, 3, 2)
    return torch.from_numpy(out.view(np.int16)).to(DEV)


def split_dev(x):
    out = torch.empty_like(x)
    check(lib().hvpr_split_bf16_f32(x.data_ptr(), x.numel(), out.data_ptr(), kernels._stream()), "split")
    return out


def conv3(xs, N, H, W, cin, wp, bias, stride, cout, cout_pad, cfg, out_split=False):
    OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    out = torch.empty((N, OH, OW, cout), dtype=torch.float32, device=DEV)
    check(lib().hvpr_conv2d_nhwc_bf16x3(xs.data_ptr(), N, H, W, cin, wp.data_ptr(), bias.data_ptr(), stride, cout, cout_pad, 1, None,
                                        None, 0, out.data_ptr(), 1 if out_split else 0, cout, 0, cfg, kernels._stream()), "conv3")
    return out


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


SHAPES = [("L0 3x3 128->128 @248x296", 128, 128, 248, 296, 1), ("L1 3x3 256->256 @124x148", 256, 256, 124, 148, 1),
          ("L2 3x3 512->512 @62x74", 512, 512, 62, 74, 1), ("L1 3x3s2 128->256 @248x296", 128, 256, 248, 296, 2),
          ("small 16->32 @11x19", 16, 32, 11, 19, 1)]
for name, cin, cout, H, W, stride in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(1, H, W, cin, generator=g)).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(DEV)
    scale, shift = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    pc = kernels.pack_conv(w, scale, shift, stride=stride, tile_cfg=1)
    ref = kernels.conv2d_nhwc(x, pc)
    t_ref = timeit(lambda: kernels.conv2d_nhwc(x, pc, out=ref))
    ref64 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), stride=stride, padding=1).relu().permute(0, 2, 3, 1)
    cout_pad = (cout + 63) // 64 * 64
    wp = pack_w(w, scale, cout_pad)
    bias = torch.zeros(cout_pad, device=DEV)
    xs = split_dev(x)
    line = f"{name:28s} fp32 {t_ref:7.1f} us (err vs f64 {float((ref.cpu().double() - ref64).abs().max() / ref64.abs().max()):.1e})"
    for cfg in ((0, 1, 2) if stride == 1 else (0, 1)):
        y = conv3(xs, 1, H, W, cin, wp, bias, stride, cout, cout_pad, cfg)
        err = float((y.cpu().double() - ref64).abs().max() / ref64.abs().max())
        t = timeit(lambda: conv3(xs, 1, H, W, cin, wp, bias, stride, cout, cout_pad, cfg))
        fl = 2 * cin * cout * 9 * y.shape[1] * y.shape[2]
        line += f" | cfg{cfg}: {t:7.1f} us {fl / t / 1e6:6.1f} TF/s-eq err {err:.1e}"
    print(line, flush=True)
