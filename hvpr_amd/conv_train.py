"""Training-mode convolution + BatchNorm + ReLU of the BEV backbone on the library's own kernels (SURVEY.md §8a row a11;
reference: BaseBEVBackbone_Scale.forward in training mode, pcdet/models/backbones_2d/base_bev_backbone.py:228-279).

Activations are NHWC tensors (N, H, W, C) — what torch calls channels_last — end to end.  Three autograd.Functions:

  conv3x3 / conv1x1   forward  hvpr_conv2d_wino_nhwc_f32 (3x3 stride 1: Winograd F(2x2,3x3)) / hvpr_conv2d_nhwc_f32 (the rest);
                               fp32 matrix cores, raw output: no bias, no activation
                      dgrad    the same kernels on the flipped + transposed weights (stride 2: on the zero-upsampled gradient)
                      wgrad    hvpr_conv2d_wino_wgrad_nhwc_f32 (3x3 stride 1, Winograd domain) / hvpr_conv2d_wgrad_nhwc_f32;
                               split-K over pixel tiles, deterministic
  deconv (k == s)     forward  the 1x1 GEMM with s*s*Cout columns + pixel shuffle in the epilogue (ConvTranspose2d, :177-188)
                      backward 1x1 dgrad / wgrad on the space-to-depth view of the gradient
  bn_relu             train-mode BatchNorm (batch statistics, differentiated through) + ReLU: hvpr_bn_stats_nhwc_f32,
                      hvpr_bn_relu_fwd_nhwc_f32, hvpr_bn_relu_bwd_nhwc_f32; running statistics updated like nn.BatchNorm2d
                      (momentum, unbiased variance, one update per CALL — SURVEY.md B.5).
"""
import os

import torch

from . import kernels
from ._lib import check, lib

_ws_cache = {}

# ---------------------------------------------------------------------------------------------- SyncBatchNorm
# tools/train.py:119-120 turns every BatchNorm into torch.nn.SyncBatchNorm when --sync_bn is given (off by default).  The
# BatchNorms of this path do not run torch's kernels, so convert_sync_batchnorm would not synchronise anything; instead the
# statistics of bn_relu / sfm_step below — every BatchNorm2d of the two-stream backbone and head, the point stream's shared MLPs
# and the VFE scale stream — are all-reduced over a process group when one is set here: per-channel (sum x, sum x^2, count) in the
# forward, (sum dy, sum dy * xhat) in the backward, in float64, one all-reduce each (RCCL on the GPU; gloo runs the same code).
# The two BatchNorm1d inside the fused PFN kernels (csrc/vfe_train.hip) and SpatialAttention's one-channel BatchNorm
# (csrc/gate_train.hip) live inside single C-ABI calls; they reach the same all-reduce through the library's hook
# (hvpr_set_batchnorm_allreduce, include/hvpr_amd.h), which set_sync_batchnorm installs and removes together with the group.
_sync = {"group": None, "hook": None}


class _DeviceDoubles:
    """`n` float64 words at device address `ptr` for torch.as_tensor (the array interface torch reads device pointers through)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def device_doubles(ptr, n):
    return torch.as_tensor(_DeviceDoubles(ptr, n), device="cuda")


def _hook_allreduce(buf, n, stream, ctx):
    # called by the library between two of its launches, on the thread that made the call.  The collective is ordered against the
    # stream the ENTRY POINT was given (the kernels before and after the hook run there) — torch's current stream when the call
    # came through kernels._stream(), but a C caller may have passed any stream
    try:
        g = _sync["group"]
        if g is None or not torch.distributed.is_initialized():
            return 0
        if stream and int(stream) != torch.cuda.current_stream().cuda_stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream))):
                torch.distributed.all_reduce(device_doubles(buf, n), group=None if g is True else g)
        else:
            torch.distributed.all_reduce(device_doubles(buf, n), group=None if g is True else g)
        _sync["hook_calls"] = _sync.get("hook_calls", 0) + 1
        return 0
    except Exception:        # an exception must not cross the C frame: the entry point reports HVPR_ERR_LAUNCH
        import traceback
        traceback.print_exc()
        return 1


def set_sync_batchnorm(group=True):
    """group: a torch.distributed process group, True = the default group, None / False = off (per-rank statistics)."""
    from ._lib import ALLREDUCE_FN
    _sync["group"] = None if not group else group
    if _sync["group"] is None:
        if _sync["hook"] is not None:            # (the library is loaded: it was needed to set the hook)
            lib().hvpr_set_batchnorm_allreduce(None, None)
            _sync["hook"] = None
        return
    if torch.cuda.is_available():                # the hooked entry points are GPU kernels; a CPU (gloo) run never reaches them
        if _sync["hook"] is None:
            _sync["hook"] = ALLREDUCE_FN(_hook_allreduce)     # kept alive here for as long as the library holds the pointer
        lib().hvpr_set_batchnorm_allreduce(_sync["hook"], None)


def _sync_group():
    g = _sync["group"]
    if g is None or not torch.distributed.is_available() or not torch.distributed.is_initialized():
        return None
    return None if g is True else g, True


def any_rank_true(flag, device):
    """`flag` (python bool) OR-ed over the ranks of the SyncBatchNorm group; `flag` itself without one.  For conditions every rank
    must act on together because a collective follows (an empty batch on one rank)."""
    sg = _sync_group()
    if sg is None:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=sg[0])
    return bool(t.item() > 0)


def global_count(n, device):
    """The number of values a BatchNorm's statistics are taken over: `n` (python number) without SyncBatchNorm, else its sum over
    the ranks as a float64 scalar tensor (the unbiased-variance factor of the running statistics uses the global count)."""
    sg = _sync_group()
    if sg is None:
        return n
    t = torch.tensor([float(n)], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, group=sg[0])
    return t[0]


def sync_moments(sum_x, sum_x2, count, eps, group=None):
    """Global batch statistics from per-rank sums: all-reduce of [sum x | sum x^2 | count] in float64 -> (mean, biased variance,
    1 / sqrt(var + eps)) as float32 and the global count as a float64 scalar tensor.  Pure torch: CPU tensors with gloo, GPU
    tensors with RCCL."""
    C = sum_x.numel()
    packed = torch.cat([sum_x.double().view(-1), sum_x2.double().view(-1), torch.tensor([float(count)], dtype=torch.float64, device=sum_x.device)])
    torch.distributed.all_reduce(packed, group=group)
    n = packed[2 * C]
    mean = packed[:C] / n
    var = (packed[C:2 * C] / n - mean * mean).clamp_min(0.0)
    return mean.float(), var.float(), torch.rsqrt(var + float(eps)).float(), n


def sync_backward_sums(dgamma, dbeta, count, group=None):
    """(sum dy * xhat, sum dy) of the global batch, already divided by the global count (float32), from the per-rank sums."""
    C = dgamma.numel()
    packed = torch.cat([dgamma.double().view(-1), dbeta.double().view(-1)])
    torch.distributed.all_reduce(packed, group=group)
    packed = packed / count
    return packed[:C].float().contiguous(), packed[C:].float().contiguous()



def _workspace(nbytes, device):
    """One grow-only scratch buffer per device: every kernel that takes it consumes it before the next one is enqueued on the
    same stream (the training step runs on one stream)."""
    buf = _ws_cache.get(device)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[device] = buf
    return buf


def _tile_cfg(cout):
    """Workgroup tile of hvpr_conv2d_nhwc_f32 (0 = 128 px x 128 ch, 1 = 64 x 64, 2 = 128 x 64).  Measured on the batch-16 training
    step: 612 / 593 / 595 ms — the 64 x 64 tile the kernel was tuned on at batch 1 also wins here."""
    return 1


def _wino_groups():
    return 1          # 8 x 16 px x 64 channel Winograd workgroups (the only form with fused BatchNorm statistics)


def bn_statistics(z, eps, partials=None):
    """mean, biased variance, 1/sqrt(var + eps) per channel of z (N,H,W,C): from the per-tile sums the producing Winograd
    convolution left behind (conv(..., stats=True)) when it did, else by a pass over z."""
    C = z.shape[-1]
    P = z.numel() // C
    dev = z.device
    mean, var, invstd = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(3))
    if partials is not None and partials.numel() > 0:
        assert partials.shape[-1] == C and partials.shape[1] == 2
        check(lib().hvpr_bn_finalize_partials_f32(partials.data_ptr(), partials.shape[0], C, P, float(eps), mean.data_ptr(), var.data_ptr(),
                                                  invstd.data_ptr(), kernels._stream()), "hvpr_bn_finalize_partials_f32")
    else:
        ws = _workspace(lib().hvpr_bn_workspace_bytes(P, C), dev)
        check(lib().hvpr_bn_stats_nhwc_f32(kernels._ptr(z, torch.float32, "z"), P, C, float(eps), mean.data_ptr(), var.data_ptr(),
                                           invstd.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()), "hvpr_bn_stats_nhwc_f32")
    sg = _sync_group()
    if sg is None:
        return mean, var, invstd, P
    if partials is not None and partials.numel() > 0:
        sums = partials.double().sum(0)
        s1, s2 = sums[0], sums[1]
    else:
        s1 = mean.double() * P
        s2 = (var.double() + mean.double() ** 2) * P
    return sync_moments(s1, s2, P, eps, sg[0])


def _affine(mean, var, invstd, gamma, beta, count, running):
    """scale = gamma * invstd, shift = beta - mean * scale for the normalising kernel.  running = (running_mean, running_var,
    num_batches_tracked, momentum) of the nn.BatchNorm2d, or None: updated like the module does in train mode — in the same launch
    (hvpr_bn_train_affine_f32) when the moments are this rank's own; the SyncBatchNorm path keeps its torch arithmetic and updates
    them in _update_running.  Returns (scale, shift, running statistics are done)."""
    if _sync_group() is not None or torch.is_tensor(count):
        scale = (gamma.detach() * invstd).contiguous()
        return scale, (beta.detach() - mean * scale).contiguous(), False
    scale, shift = torch.empty_like(mean), torch.empty_like(mean)
    rm = rv = nbt = None
    m = mu = 0.0
    if running is not None and running[3] is not None:
        rm, rv, nbt, m = running
        mu = m * count / max(count - 1, 1)
    check(lib().hvpr_bn_train_affine_f32(mean.data_ptr(), var.data_ptr(), invstd.data_ptr(), mean.numel(), kernels._ptr(gamma.detach(), torch.float32, "gamma"),
                                         kernels._ptr(beta.detach(), torch.float32, "beta"), float(m), float(mu), kernels._ptr(rm, torch.float32, "running_mean"),
                                         kernels._ptr(rv, torch.float32, "running_var"), kernels._ptr(nbt, torch.int64, "num_batches_tracked"),
                                         scale.data_ptr(), shift.data_ptr(), kernels._stream()), "hvpr_bn_train_affine_f32")
    if rm is not None:      # the kernel wrote the buffers through raw pointers: let torch's version counters know (no launch)
        for t in (rm, rv, nbt):
            torch.autograd.graph.increment_version(t)
    return scale, shift, rm is not None


def _running_of(bn):
    """The running buffers of `bn` for _affine (None when it keeps none or averages cumulatively: _update_running then does it)."""
    if not bn.track_running_stats or bn.momentum is None or bn.running_mean is None:
        return None
    return (bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum))


def _bn_backward(dy, z, P, C, scale, shift, mean, invstd, relu, gate, dgate, count):
    """dz, d gamma, d beta of the train-mode BatchNorm (+ ReLU, + SFM gate) — one kernel pair per rank, or, with SyncBatchNorm,
    the two halves with the all-reduce of the two sums between them (count: the global count of the forward)."""
    dz = torch.empty_like(z)
    dgamma, dbeta = torch.empty_like(mean), torch.empty_like(mean)
    ws = _workspace(lib().hvpr_bn_workspace_bytes(P, C), z.device)
    sg = _sync_group()
    if sg is None or not torch.is_tensor(count):
        check(lib().hvpr_bn_relu_bwd_nhwc_f32(kernels._ptr(dy, torch.float32, "dy"), z.data_ptr(), P, C, scale.data_ptr(), shift.data_ptr(),
                                              mean.data_ptr(), invstd.data_ptr(), 1 if relu else 0, kernels._ptr(gate), kernels._ptr(dgate),
                                              dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(),
                                              kernels._stream()), "hvpr_bn_relu_bwd_nhwc_f32")
        return dz, dgamma, dbeta
    check(lib().hvpr_bn_relu_bwd_sums_nhwc_f32(kernels._ptr(dy, torch.float32, "dy"), z.data_ptr(), P, C, scale.data_ptr(), shift.data_ptr(),
                                               mean.data_ptr(), invstd.data_ptr(), 1 if relu else 0, kernels._ptr(gate), dgamma.data_ptr(),
                                               dbeta.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()), "hvpr_bn_relu_bwd_sums_nhwc_f32")
    dg_t, db_t = sync_backward_sums(dgamma, dbeta, count, sg[0])          # over the global batch, divided by its count
    check(lib().hvpr_bn_relu_bwd_apply_nhwc_f32(dy.data_ptr(), z.data_ptr(), P, C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                                invstd.data_ptr(), 1 if relu else 0, kernels._ptr(gate), kernels._ptr(dgate), dz.data_ptr(),
                                                dg_t.data_ptr(), db_t.data_ptr(), 1.0, kernels._stream()), "hvpr_bn_relu_bwd_apply_nhwc_f32")
    return dz, dgamma, dbeta          # parameter gradients: this rank's sums (DistributedDataParallel averages them)


_wino_pack_cache = {}
_weights_epoch = [0]
# HVPR_DEBUG_PACK_CACHE=1: every cache hit re-checks a checksum of the weight (a host sync per convolution: debugging only; the GPU
# test suite runs one training test this way)
_DEBUG_PACK_CACHE = os.environ.get("HVPR_DEBUG_PACK_CACHE", "0") == "1"


def weights_changed():
    """Parameters were written behind torch's back — through a raw pointer (the flat fused optimiser's kernel) or in place through
    `.data`, which has its own version counter: cached packed filters are stale.  Anything that writes weights that way calls this."""
    _weights_epoch[0] += 1
    _wino_pack_cache.clear()


def _packed(weight, kind, build):
    """build() — a packed image derived from `weight` alone — kept for as long as the parameter is not written.  Every convolution of
    the training step packed its filter first: 78 `k_wino_pack` launches per step (the weight-shared SFM layer six times forward and
    six times for the data gradient) and four torch launches per direct-kernel call.  An entry belongs to ONE tensor object (weak
    reference: an address can be handed to another tensor), one version and one data pointer (torch's in-place updates bump the
    version; the flat fused optimiser, which writes through a raw pointer, calls weights_changed())."""
    import weakref
    key = (id(weight), kind)
    state = (weight._version, weight.data_ptr())
    hit = _wino_pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == state:
        if _DEBUG_PACK_CACHE and float(weight.detach().double().sum()) != hit[3]:
            raise RuntimeError("hvpr_amd.conv_train: a parameter was written without a version bump (through `.data` or a raw pointer) and "
                               "without conv_train.weights_changed(): its cached packed filter is stale")
        return hit[2]
    pc = build()
    if len(_wino_pack_cache) > 1024:
        _wino_pack_cache.clear()
    # the entry goes when its tensor does (fuzz / gradcheck loops create thousands of short-lived weights: their packed copies would
    # otherwise stay until the size bound above)
    _wino_pack_cache[key] = (weakref.ref(weight, lambda _r, k=key: _wino_pack_cache.pop(k, None)), state, pc,
                             float(weight.detach().double().sum()) if _DEBUG_PACK_CACHE else None)
    return pc


def _pack_wino(weight, adjoint):
    return _packed(weight, ("wino", bool(adjoint)),
                   lambda: kernels.pack_conv_wino(weight.detach(), relu=False, px_groups=_wino_groups(), adjoint=adjoint))


def conv_fwd_raw(x, weight, stride=1, adjoint=False, stats=False):
    """x (N,H,W,Cin) -> conv(x, weight) (N,OH,OW,Cout), no bias / activation.  weight (Cout,Cin,k,k), k in {1,3}, pad (k-1)/2.
    stats: return (z, partials) — partials = the batch statistics' per-tile sums when the Winograd kernel produced them, else None.
    adjoint: weight is the (Cin', Cout', 3, 3) filter of the layer whose data gradient is wanted and x its output gradient.
    Stride-1 3x3: the Winograd kernel (packed on the device per call) unless HVPR_CONV_ALGO=direct."""
    if weight.shape[2] == 3 and stride == 1 and kernels.conv_algo() == "winograd" and weight.shape[1 if adjoint else 0] % 4 == 0 \
            and weight.shape[0 if adjoint else 1] % 8 == 0:
        g = _wino_groups()
        partials = None
        if stats and g == 1:
            N, H, W, _ = x.shape
            cout = weight.shape[1 if adjoint else 0]
            partials = torch.empty((lib().hvpr_conv2d_wino_stats_rows(N, H, W), 2, cout), dtype=torch.float32, device=x.device)
        z = kernels.conv2d_wino_nhwc(x, _pack_wino(weight, adjoint), bn_partials=partials)
        return (z, partials) if stats else z
    def build():
        w = weight.detach().permute(1, 0, 2, 3).flip(2, 3) if adjoint else weight       # (Cin, Cout, k, k): the adjoint kernel
        return kernels.pack_conv(w, None, None, stride=stride, relu=False, tile_cfg=_tile_cfg(w.shape[0]))
    pc = _packed(weight, ("direct", int(stride), bool(adjoint)), build)
    z = kernels.conv2d_nhwc(x, pc)
    return (z, None) if stats else z


def conv_wgrad(x, dz, taps, stride, cout, cin):
    N, H, W, _ = x.shape
    OH, OW = dz.shape[1], dz.shape[2]
    k = 3 if taps == 9 else 1
    dw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=x.device)
    # (the Winograd-domain kernel addresses inside an image with 32 bits: images of 2 GB and more go to the direct kernel)
    if taps == 9 and stride == 1 and kernels.conv_algo() == "winograd" and H * W * max(cin, cout) * 4 < 2 ** 31:
        nbytes = lib().hvpr_conv2d_wino_wgrad_workspace_bytes(N, H, W, cin, cout)
        ws = _workspace(nbytes, x.device)
        check(lib().hvpr_conv2d_wino_wgrad_nhwc_f32(kernels._ptr(x, torch.float32, "x"), N, H, W, cin, kernels._ptr(dz, torch.float32, "dz"),
                                                    cout, dw.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()),
              "hvpr_conv2d_wino_wgrad_nhwc_f32")
        return dw
    nbytes = lib().hvpr_conv2d_wgrad_workspace_bytes(N, OH, OW, cin, cout, taps, stride)
    ws = _workspace(nbytes, x.device)
    check(lib().hvpr_conv2d_wgrad_nhwc_f32(kernels._ptr(x, torch.float32, "x"), N, H, W, cin, kernels._ptr(dz, torch.float32, "dz"),
                                           cout, taps, stride, dw.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()),
          "hvpr_conv2d_wgrad_nhwc_f32")
    return dw


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, stats=False):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.stride = stride
        if not stats:
            return conv_fwd_raw(x, weight, stride)
        z, partials = conv_fwd_raw(x, weight, stride, stats=True)
        if partials is None:
            partials = z.new_empty((0, 2, z.shape[-1]))
        ctx.mark_non_differentiable(partials)
        ctx.set_materialize_grads(False)     # (their gradients would arrive as zero tensors: one allocation + fill each, per call)
        return z, partials

    @staticmethod
    def backward(ctx, dz, _dp=None):
        if dz is None:
            return None, None, None, None
        x, weight = ctx.saved_tensors
        dz = dz.contiguous()
        cout, cin, k, _ = weight.shape
        s = ctx.stride
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if s == 1:
                dx = conv_fwd_raw(dz, weight, 1, adjoint=True)
            elif k == 3 and cout % 8 == 0 and cin % 4 == 0:
                # y[o] = sum x[2 o + k - 1] w[k]  =>  dx[2a + py, 2b + px] = sum over the taps of that parity class of w[k]^T dz[...]:
                # the gather form on the direct kernel (a stride-1 convolution over the zero-upsampled dz multiplied four zeros in five)
                dx = kernels.conv2d_s2_dgrad_nhwc(dz, weight, x.shape[1], x.shape[2], pc=_packed(weight, "s2dgrad", lambda: kernels.pack_conv_s2_dgrad(weight)))
            else:            # ... = conv_stride1(zero-upsampled dz, flipped w)
                N, H, W, _ = x.shape
                OH, OW = dz.shape[1], dz.shape[2]
                up = torch.zeros((N, H, W, cout), dtype=torch.float32, device=dz.device)
                up[:, 0:2 * OH:2, 0:2 * OW:2] = dz
                dx = conv_fwd_raw(up, weight, 1, adjoint=True)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(x, dz, k * k, s, cout, cin)
        return dx, dw, None, None


class _Deconv(torch.autograd.Function):
    """ConvTranspose2d(kernel == stride == s, no bias): x (N,H,W,Cin), weight (Cin,Cout,s,s) -> (N,H*s,W*s,Cout)."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        cin, cout, s, _ = weight.shape
        def build():
            ones, zeros = torch.ones(cout, device=x.device), torch.zeros(cout, device=x.device)
            return kernels.pack_deconv(weight, ones, zeros, relu=False, tile_cfg=1 if s < 4 else 2)
        pc = _packed(weight, "deconv", build)
        ctx.save_for_backward(x, weight)
        return kernels.conv2d_nhwc(x, pc)

    @staticmethod
    def backward(ctx, dz):
        x, weight = ctx.saved_tensors
        cin, cout, s, _ = weight.shape
        N, H, W, _ = x.shape
        # space-to-depth: column (sy * s + sx) * Cout + co of pixel (y, x) = dz[y * s + sy][x * s + sx][co]
        cols = s * s * cout
        dzs = dz.reshape(N, H, s, W, s, cout).permute(0, 1, 3, 2, 4, 5).reshape(N, H, W, cols).contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            def build():
                g = weight.detach().permute(0, 2, 3, 1).reshape(cin, cols, 1, 1)     # dx[ci] = sum_col dzs[col] * g[ci][col]
                return kernels.pack_conv(g, None, None, stride=1, relu=False, tile_cfg=_tile_cfg(cin))
            dx = kernels.conv2d_nhwc(dzs, _packed(weight, "deconv_dgrad", build))
        if ctx.needs_input_grad[1]:
            dg = conv_wgrad(x, dzs, 1, 1, cols, cin)                              # (cols, Cin, 1, 1)
            dw = dg.reshape(s, s, cout, cin).permute(3, 2, 0, 1).contiguous()
        return dx, dw


class _BNReLU(torch.autograd.Function):
    """y = relu(BN_train(z)); with (gate (N,H,W,1), resid (N,H,W,C)): y = gate * relu(BN_train(z)) + resid — the SFM step
    x_att = attention(sfm(x_att), y) + x_att (base_bev_backbone.py:250-255) in the same two kernels."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps, relu, gate, resid, partials=None, running=None):
        z = z.contiguous()
        C = z.shape[-1]
        P = z.numel() // C
        dev = z.device
        mean, var, invstd, count = bn_statistics(z, eps, partials)
        scale, shift, ctx.running_done = _affine(mean, var, invstd, gamma, beta, count, running)
        y = torch.empty_like(z)
        if gate is not None:
            gate, resid = gate.detach().contiguous(), resid.detach().contiguous()
            assert gate.numel() == P and resid.shape == z.shape
        check(lib().hvpr_bn_relu_fwd_nhwc_f32(z.data_ptr(), P, C, scale.data_ptr(), shift.data_ptr(), 1 if relu else 0,
                                              kernels._ptr(gate, torch.float32, "gate"), kernels._ptr(resid, torch.float32, "resid"),
                                              y.data_ptr(), kernels._stream()), "hvpr_bn_relu_fwd_nhwc_f32")
        ctx.save_for_backward(z, scale, shift, mean, invstd, gate)
        ctx.relu, ctx.count = relu, count
        cnt = count if torch.is_tensor(count) else torch.tensor(float(count), dtype=torch.float64)      # (a HOST tensor: no copy to the device, no sync)
        ctx.mark_non_differentiable(mean, var, cnt)
        ctx.set_materialize_grads(False)     # (their gradients would arrive as zero tensors: one allocation + fill each, per call)
        return y, mean, var, cnt

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dc):
        if dy is None:
            return (None,) * 9
        z, scale, shift, mean, invstd, gate = ctx.saved_tensors
        dy = dy.contiguous()
        C = z.shape[-1]
        P = z.numel() // C
        dgate = torch.empty_like(gate) if gate is not None else None
        dz, dgamma, dbeta = _bn_backward(dy, z, P, C, scale, shift, mean, invstd, ctx.relu, gate, dgate, ctx.count)
        return dz, dgamma, dbeta, None, None, dgate, (dy if gate is not None else None), None, None


_ones_cache = {}


def _ones(shape, device):
    key = (tuple(shape), device)
    if key not in _ones_cache:
        _ones_cache[key] = torch.ones(shape, dtype=torch.float32, device=device)
    return _ones_cache[key]


class _SfmStep(torch.autograd.Function):
    """x_att' = gate * relu(BN_train(conv3x3(x_att))) + x_att (base_bev_backbone.py:250-255) as ONE autograd node: the backward adds
    the residual path's gradient in the epilogue of the data-gradient convolution (y = 1 * conv_adjoint(dz) + dy) instead of
    leaving a 3-tensor element-wise add of full activations to autograd."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, eps, gate, running=None):
        x = x.contiguous()
        z, partials = conv_fwd_raw(x, weight, 1, stats=True)
        C = z.shape[-1]
        P = z.numel() // C
        dev = z.device
        mean, var, invstd, count = bn_statistics(z, eps, partials)
        scale, shift, ctx.running_done = _affine(mean, var, invstd, gamma, beta, count, running)
        gate = gate.detach().contiguous()
        assert gate.numel() == P
        y = torch.empty_like(z)
        check(lib().hvpr_bn_relu_fwd_nhwc_f32(z.data_ptr(), P, C, scale.data_ptr(), shift.data_ptr(), 1, gate.data_ptr(), x.data_ptr(),
                                              y.data_ptr(), kernels._stream()), "hvpr_bn_relu_fwd_nhwc_f32")
        ctx.save_for_backward(x, weight, z, scale, shift, mean, invstd, gate)
        ctx.count = count
        cnt = count if torch.is_tensor(count) else torch.tensor(float(count), dtype=torch.float64)      # (a HOST tensor: no copy to the device, no sync)
        ctx.mark_non_differentiable(mean, var, cnt)
        ctx.set_materialize_grads(False)     # (their gradients would arrive as zero tensors: one allocation + fill each, per call)
        return y, mean, var, cnt

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dc):
        if dy is None:
            return (None,) * 7
        x, weight, z, scale, shift, mean, invstd, gate = ctx.saved_tensors
        dy = dy.contiguous()
        C = z.shape[-1]
        P = z.numel() // C
        dgate = torch.empty_like(gate)
        dz, dgamma, dbeta = _bn_backward(dy, z, P, C, scale, shift, mean, invstd, True, gate, dgate, ctx.count)
        cout, cin = weight.shape[0], weight.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if kernels.conv_algo() == "winograd" and cin % 4 == 0 and cout % 8 == 0:
                pc = _pack_wino(weight, True)
                dx = kernels.conv2d_wino_nhwc(dz, pc, gate=_ones(tuple(dz.shape[:3]), dz.device), resid=dy)    # + the residual path
            else:
                dx = conv_fwd_raw(dz, weight, 1, adjoint=True) + dy
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(x, dz, 9, 1, cout, cin)
        return dx, dw, dgamma, dbeta, None, dgate, None


def sfm_step(x, weight, bn, gate):
    """gate * relu(bn(conv3x3(x))) + x with train-mode `bn` (running statistics updated like nn.BatchNorm2d), one autograd node."""
    running = _running_of(bn) if _sync_group() is None else None        # then updated inside, by the launch that forms scale / shift
    y, mean, var, count = _SfmStep.apply(x, weight, bn.weight, bn.bias, bn.eps, gate, running)
    if running is None:
        _update_running(bn, mean, var, count if _sync_group() is not None else x.numel() // x.shape[-1])
    return y


class _BNReLUCat(torch.autograd.Function):
    """cat([relu(BN_train(z_j)) for j], dim=-1) as ONE node: every branch normalises straight into its channel slice of the result and
    takes its gradient straight out of the result's gradient (hvpr_bn_relu_fwd/bwd_slice_nhwc_f32) — base_bev_backbone.py:262-279 without
    the concatenation's copy (1.8 GB per stream at batch 16) and the three slice copies of its backward.
    apply(eps_0, running_0, ..., z_0, gamma_0, beta_0, z_1, ...): the first 2 k arguments are not tensors."""

    @staticmethod
    def forward(ctx, k, *args):
        meta, ten = args[:2 * k], args[2 * k:]
        zs = [ten[3 * j].contiguous() for j in range(k)]
        Cs = [z.shape[-1] for z in zs]
        P = zs[0].numel() // Cs[0]
        total = sum(Cs)
        out = torch.empty(tuple(zs[0].shape[:-1]) + (total,), dtype=torch.float32, device=zs[0].device)
        saved, off = [], 0
        for j in range(k):
            z, gamma, beta = zs[j], ten[3 * j + 1], ten[3 * j + 2]
            assert z.numel() // Cs[j] == P
            mean, var, invstd, count = bn_statistics(z, meta[2 * j])
            scale, shift, done = _affine(mean, var, invstd, gamma, beta, count, meta[2 * j + 1])
            assert done or meta[2 * j + 1] is None
            check(lib().hvpr_bn_relu_fwd_slice_nhwc_f32(z.data_ptr(), P, Cs[j], scale.data_ptr(), shift.data_ptr(), 1, out.data_ptr(), total, off,
                                                        kernels._stream()), "hvpr_bn_relu_fwd_slice_nhwc_f32")
            saved += [z, scale, shift, mean, invstd]
            off += Cs[j]
        ctx.save_for_backward(*saved)
        ctx.k, ctx.Cs, ctx.P = k, Cs, P
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        k, Cs, P = ctx.k, ctx.Cs, ctx.P
        total = sum(Cs)
        grads, off = [], 0
        for j in range(k):
            z, scale, shift, mean, invstd = ctx.saved_tensors[5 * j:5 * j + 5]
            dz = torch.empty_like(z)
            dgamma, dbeta = torch.empty_like(mean), torch.empty_like(mean)
            ws = _workspace(lib().hvpr_bn_workspace_bytes(P, Cs[j]), z.device)
            check(lib().hvpr_bn_relu_bwd_slice_nhwc_f32(dy.data_ptr(), total, off, z.data_ptr(), P, Cs[j], scale.data_ptr(), shift.data_ptr(),
                                                        mean.data_ptr(), invstd.data_ptr(), 1, dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), kernels._stream()), "hvpr_bn_relu_bwd_slice_nhwc_f32")
            grads += [dz, dgamma, dbeta]
            off += Cs[j]
        return (None,) + (None,) * (2 * k) + tuple(grads)


def bn_relu_cat(zs, bns):
    """torch.cat([bn_relu(z, bn) for z, bn in zip(zs, bns)], dim=-1) with train-mode BatchNorm2d modules, one autograd node, no copy of
    the parts (SyncBatchNorm or a BatchNorm that keeps no running statistics of the usual kind: the plain form)."""
    runs = [_running_of(bn) for bn in bns]
    if _sync_group() is not None or any(r is None for r in runs):
        return torch.cat([bn_relu(z, bn) for z, bn in zip(zs, bns)], dim=-1)
    meta, ten = [], []
    for z, bn, r in zip(zs, bns, runs):
        meta += [bn.eps, r]
        ten += [z, bn.weight, bn.bias]
    return _BNReLUCat.apply(len(zs), *meta, *ten)


def _update_running(bn, mean, var, n):
    """n: the number of values per channel the statistics were taken over — a python number, or (SyncBatchNorm: the global count) a
    scalar tensor; the unbiased-variance factor n / (n - 1) is then formed on the device."""
    if bn.track_running_stats:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            if torch.is_tensor(n):
                bn.running_var.mul_(1 - m).add_(var * (n / (n - 1).clamp_min(1.0)).to(var.dtype) * m)
            else:
                bn.running_var.mul_(1 - m).add_(var, alpha=m * n / max(n - 1, 1))


def conv(x, weight, stride=1, stats=False):
    """3x3 (pad 1) or 1x1 convolution without bias on NHWC activations, differentiable in x and weight.  stats: the output goes
    straight into bn_relu — returns (z, partials): the per-tile sums of the batch statistics the Winograd kernel left for it (an
    empty tensor when it did not), to be handed to bn_relu(..., partials=)."""
    return _Conv.apply(x, weight, int(stride), bool(stats))


def deconv(x, weight):
    return _Deconv.apply(x, weight)


def bn_relu(z, bn, relu=True, gate=None, resid=None, partials=None):
    """Train-mode nn.BatchNorm2d `bn` (its weight / bias / eps / momentum / running buffers) + optional ReLU on NHWC `z`; with
    gate (N,H,W,1) and resid (N,H,W,C): gate * relu(bn(z)) + resid, differentiable in all of them."""
    running = _running_of(bn) if _sync_group() is None else None        # then updated inside, by the launch that forms scale / shift
    y, mean, var, count = _BNReLU.apply(z, bn.weight, bn.bias, bn.eps, relu, gate, resid, partials, running)
    if running is None:
        _update_running(bn, mean, var, count if _sync_group() is not None else z.numel() // z.shape[-1])
    return y


class _GateTrain(torch.autograd.Function):
    """SpatialAttention with batch statistics on hvpr_spatial_gate_train_fwd/bwd_f32 (csrc/gate_train.hip): y (N,H,W,C) ->
    gate (N,H,W,1) = sigmoid(BN_train(conv3x3_{2->1}(cat[max_c y, mean_c y]) + bias)); spatial_attention.py:47-63."""

    @staticmethod
    def forward(ctx, y, weight, bias, gamma, beta, eps):
        y = y.contiguous()
        N, H, W, C = y.shape
        dev = y.device
        w18 = weight.detach().reshape(18).contiguous()
        b, g_, be = bias.detach().reshape(1).contiguous(), gamma.detach().reshape(1).contiguous(), beta.detach().reshape(1).contiguous()
        pooled = torch.empty((N, H, W, 2), dtype=torch.float32, device=dev)
        argmax = torch.empty((N, H, W), dtype=torch.int32, device=dev)
        a = torch.empty((N, H, W), dtype=torch.float32, device=dev)
        stats = torch.empty(3, dtype=torch.float32, device=dev)
        gate = torch.empty((N, H, W, 1), dtype=torch.float32, device=dev)
        ws = _workspace(lib().hvpr_spatial_gate_train_workspace_bytes(N, H, W), dev)
        check(lib().hvpr_spatial_gate_train_fwd_f32(kernels._ptr(y, torch.float32, "y"), N, H, W, C, w18.data_ptr(), b.data_ptr(), g_.data_ptr(),
                                                    be.data_ptr(), float(eps), pooled.data_ptr(), argmax.data_ptr(), a.data_ptr(),
                                                    stats.data_ptr(), gate.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()),
              "hvpr_spatial_gate_train_fwd_f32")
        ctx.save_for_backward(gate, a, stats, pooled, argmax, w18, g_)
        ctx.dims = (N, H, W, C)
        ctx.wshape = tuple(weight.shape)
        mean, var = stats[0:1], stats[1:2]
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)     # (their gradients would arrive as zero tensors: one allocation + fill each, per call)
        return gate, mean, var

    @staticmethod
    def backward(ctx, dgate, _dm, _dv):
        if dgate is None:
            return (None,) * 6
        gate, a, stats, pooled, argmax, w18, g_ = ctx.saved_tensors
        N, H, W, C = ctx.dims
        dev = gate.device
        dgate = dgate.contiguous()
        dy = torch.empty((N, H, W, C), dtype=torch.float32, device=dev)
        dw = torch.empty(18, dtype=torch.float32, device=dev)
        db, dgam, dbet = (torch.empty(1, dtype=torch.float32, device=dev) for _ in range(3))
        ws = _workspace(lib().hvpr_spatial_gate_train_workspace_bytes(N, H, W), dev)
        check(lib().hvpr_spatial_gate_train_bwd_f32(kernels._ptr(dgate, torch.float32, "dgate"), gate.data_ptr(), a.data_ptr(), stats.data_ptr(),
                                                    pooled.data_ptr(), argmax.data_ptr(), w18.data_ptr(), g_.data_ptr(), N, H, W, C, dy.data_ptr(),
                                                    dw.data_ptr(), db.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), ws.data_ptr(), ws.numel(),
                                                    kernels._stream()), "hvpr_spatial_gate_train_bwd_f32")
        return dy, dw.view(ctx.wshape), db, dgam, dbet, None


def spatial_gate_train(y, weight, bias, gamma, beta, eps):
    """-> (gate (N,H,W,1), batch mean (1,), biased batch variance (1,)) of SpatialAttention in training mode."""
    return _GateTrain.apply(y, weight, bias, gamma, beta, eps)


def update_running_repeated(bn, mean, var, n, times):
    """`times` running-statistics updates of nn.BatchNorm with the SAME batch statistics, in closed form:
    r <- (1 - m)^k r + (1 - (1 - m)^k) s; num_batches_tracked += k."""
    if not bn.track_running_stats or times < 1:
        return
    with torch.no_grad():
        if bn.momentum is None:      # cumulative average: every update is its own step
            for _ in range(times):
                _update_running(bn, mean, var, n)
            return
        keep = (1.0 - bn.momentum) ** times
        bn.num_batches_tracked += times
        bn.running_mean.mul_(keep).add_(mean.view_as(bn.running_mean), alpha=1.0 - keep)
        if torch.is_tensor(n):
            bn.running_var.mul_(keep).add_(var.view_as(bn.running_var) * (n / (n - 1).clamp_min(1.0)).to(var.dtype) * (1.0 - keep))
        else:
            bn.running_var.mul_(keep).add_(var.view_as(bn.running_var), alpha=(1.0 - keep) * n / max(n - 1, 1))
