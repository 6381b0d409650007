"""2-D BEV backbone with the reference's plugin interface (pcdet/models/backbones_2d/base_bev_backbone.py,
spatial_attention.py): same registry key, ctor kwargs and state-dict names.  The eval forward runs every convolution
on the fp32 matrix cores through hvpr_conv2d_nhwc_f32 with BatchNorm / ReLU / gate / residual fused in the epilogue,
activations kept NHWC end to end, the gate computed ONCE per level (it depends only on the scale stream,
base_bev_backbone.py:289-293) and each deconv writing straight into its slice of the 384-channel concat."""
import contextlib
import os

import torch
import torch.nn as nn

from . import kernels
from .folding import FoldCache, bn_scale_shift


class ConvLayer(nn.Module):
    """spatial_attention.py:9-45 — conv (+bias) + BatchNorm2d, no activation here."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, use_norm=True, activation=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.norm = nn.BatchNorm2d(out_channels, eps=1e-3, momentum=0.01) if use_norm else None
        self.relu = nn.ReLU() if activation else None


class SpatialAttention(nn.Module):
    """CBAM-style spatial gate — spatial_attention.py:51-63.  forward(x, w) = sigmoid(BN(conv(pool(w)))) * x."""

    def __init__(self):
        super().__init__()
        self.spatial = ConvLayer(2, 1, 3, stride=1, padding=1, use_norm=True, activation=False)

    def gate_params(self):
        s, t = bn_scale_shift(self.spatial.norm)
        return (self.spatial.conv.weight.detach().float().reshape(18).contiguous(), float(self.spatial.conv.bias.detach()),
                float(s), float(t))


def _conv_bn_relu(cin, cout, stride=1, zero_pad=False):
    if zero_pad:   # nn.ZeroPad2d(1) + conv(padding=0) — the form the reference uses for strided entries (:154-160)
        return [nn.ZeroPad2d(1), nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=0, bias=False),
                nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01), nn.ReLU()]
    return [nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01), nn.ReLU()]


# Workgroup tile of hvpr_conv2d_nhwc_f32 per layer group and level: 0 = 128 px x 128 ch, 1 = 64 px x 64 ch, 2 = 128 px x 64 ch.
# Measured on MI355X at batch 1 inside the whole frame (bench.py; the two streams of the backbone interact, so the per-layer
# micro-benchmarks tools/bench_conv.py / bench_conv1x1.py do not decide alone); the sweeps are in profiles/NOTES_r04.md.
_TILES = {"trunk": (1, 1, 1), "sfm": (1, 1, 1), "scale": (1, 1, 1), "deconv": (1, 1, 1)}   # deconv level 2: 128 x 64 won 0.2 % in round 1's 3-stage pipeline, 64 x 64 wins 0.7 % in the 4-stage one


def _tile_cfg(kind, level):
    return _TILES[kind][min(level, 2)]


def _wino_groups(level):
    """Pixel groups per workgroup of the Winograd kernel: 1 = 8 x 16 px x 64 channels on every level (16 x 16 px workgroups and
    32-channel tiles were measured slower inside the frame pipeline, profiles/NOTES_r04.md)."""
    return 1


class BaseBEVBackbone_Scale(nn.Module):
    """base_bev_backbone.py:116-315."""

    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums, strides, filters = list(model_cfg.LAYER_NUMS), list(model_cfg.LAYER_STRIDES), list(model_cfg.NUM_FILTERS)
        assert len(layer_nums) == len(strides) == len(filters)
        self.sfm_layer_nums = list(model_cfg.SFM_LAYER_NUMS)
        up_strides, up_filters = list(model_cfg.UPSAMPLE_STRIDES), list(model_cfg.NUM_UPSAMPLE_FILTERS)
        assert len(up_strides) == len(up_filters) == len(layer_nums), "HIP path: one deblock per level"
        scale_filters = list(model_cfg.NUM_SCALE_FILTERS)
        assert len(scale_filters) == len(strides)
        cin = [input_channels] + filters[:-1]
        cin_s = [input_channels // 4] + scale_filters[:-1]
        self.layer_strides, self.layer_nums, self.upsample_strides = strides, layer_nums, up_strides
        self.sfmblocks_down, self.sfmblocks_up = nn.ModuleList(), nn.ModuleList()
        self.scale_layers, self.blocks, self.deblocks = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(len(layer_nums)):
            layers = _conv_bn_relu(cin[i], filters[i], strides[i], zero_pad=True)
            for _ in range(layer_nums[i]):
                layers += _conv_bn_relu(filters[i], filters[i])
            self.blocks.append(nn.Sequential(*layers))
            self.sfmblocks_down.append(nn.Sequential(*_conv_bn_relu(filters[i], filters[i])))
            s = up_strides[i]
            assert s >= 1 and int(s) == s, "HIP path: integer upsample strides (ConvTranspose2d with kernel == stride)"
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(filters[i], up_filters[i], int(s), stride=int(s), bias=False),
                nn.BatchNorm2d(up_filters[i], eps=1e-3, momentum=0.01), nn.ReLU()))
            self.scale_layers.append(nn.Sequential(*_conv_bn_relu(cin_s[i], scale_filters[i], strides[i], zero_pad=True)))
        self.up_filters = up_filters
        self.num_bev_features = sum(up_filters)
        self.attention = SpatialAttention()
        self._fold = FoldCache()
        self._shape_key = None
        self._side = None
        self._sides = None
        # HIP streams of the eval forward: "1" none, "2" trunk + one branch stream, "4" trunk + one stream per level's branch
        self.n_streams = int(os.environ.get("HVPR_BEV_STREAMS", "4"))
        self.overlap_branches = self.n_streams != 1
        # "fp32": exact fp32 matrix-core kernel (default, the parity reference).  "bf16x3" / "bf16x6": trunk and SFM 3x3
        # convolutions on the bf16 matrix cores with operands split into 2 / 3 bf16 planes (kernels.conv2d_nhwc_bf3):
        # ~5e-6 relative error per layer / the fp32 kernel's own ~1.5e-6 (fp32 emulation)
        self.conv_precision = os.environ.get("HVPR_CONV_PRECISION", model_cfg.get("CONV_PRECISION", "fp32"))
        assert self.conv_precision in self.PRECISIONS

    PRECISIONS = {"fp32": 0, "bf16x3": 2, "bf16x6": 3}      # name -> bf16 planes per value

    def set_conv_precision(self, precision):
        assert precision in self.PRECISIONS
        self.conv_precision = precision
        self._fold.invalidate()

    def _side_stream(self, device):
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def _branch_streams(self, device, n):
        """One stream per level's attentive branch (n_streams == 4); a single shared one otherwise."""
        if self.n_streams < 4:
            return [self._side_stream(device)] * n
        if self._sides is None or len(self._sides) != n or self._sides[0].device != device:
            self._sides = [torch.cuda.Stream(device=device) for _ in range(n)]
        return self._sides

    def train(self, mode=True):
        self._fold.invalidate()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._fold.invalidate()
        return super()._load_from_state_dict(*a, **k)

    def _build_packed(self, H, W):
        packed = {"levels": [], "gate": self.attention.gate_params()}
        h, w = H, W
        for i in range(len(self.blocks)):
            s = self.layer_strides[i]
            h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
            blk = self.blocks[i]
            cout = blk[1].weight.shape[0]
            cfg = _tile_cfg("trunk", i)
            lv = {"convs": []}
            sc, sh = bn_scale_shift(blk[2])
            # stride-1 3x3 layers: Winograd F(2x2,3x3) kernel unless HVPR_CONV_ALGO=direct (kernels.pack_conv_auto)
            wg = _wino_groups(i)
            lv["convs"].append(kernels.pack_conv_auto(blk[1].weight, sc, sh, stride=s, tile_cfg=cfg, px_groups=wg))
            for k in range(self.layer_nums[i]):
                sc, sh = bn_scale_shift(blk[5 + 3 * k])
                lv["convs"].append(kernels.pack_conv_auto(blk[4 + 3 * k].weight, sc, sh, tile_cfg=cfg, px_groups=wg))
            sc, sh = bn_scale_shift(self.sfmblocks_down[i][1])
            lv["sfm"] = kernels.pack_conv_auto(self.sfmblocks_down[i][0].weight, sc, sh, tile_cfg=_tile_cfg("sfm", i), px_groups=wg)
            planes = self.PRECISIONS[self.conv_precision]
            if planes:
                c3 = 4                                # 64 px x 64 ch tiles, weights staged per kernel row (2-3 workgroups per CU)
                lv["convs3"] = []
                sc, sh = bn_scale_shift(blk[2])
                lv["convs3"].append(kernels.pack_conv_bf3(blk[1].weight, sc, sh, stride=s, tile_cfg=c3, planes=planes))
                for k in range(self.layer_nums[i]):
                    sc, sh = bn_scale_shift(blk[5 + 3 * k])
                    lv["convs3"].append(kernels.pack_conv_bf3(blk[4 + 3 * k].weight, sc, sh, tile_cfg=c3, planes=planes))
                sc, sh = bn_scale_shift(self.sfmblocks_down[i][1])
                lv["sfm3"] = kernels.pack_conv_bf3(self.sfmblocks_down[i][0].weight, sc, sh, tile_cfg=c3, planes=planes)
                de = self.deblocks[i]
                sc, sh = bn_scale_shift(de[1])
                # two planes: the deconvolution runs split as well; three planes: its six-product 1x1 is slower than the fp32 kernel
                lv["deconv3"] = kernels.pack_deconv_bf3(de[0].weight, sc, sh, planes=planes) \
                    if (planes == 2 and de[0].weight.shape[0] % 64 == 0) else None
            sl = self.scale_layers[i]
            sc, sh = bn_scale_shift(sl[2])
            lv["scale"] = kernels.pack_conv_auto(sl[1].weight, sc, sh, stride=s, tile_cfg=_tile_cfg("scale", i), px_groups=wg)
            de = self.deblocks[i]
            sc, sh = bn_scale_shift(de[1])
            us = int(self.upsample_strides[i])
            # the k = s = 4 deconvolution (K = 512, 2048 columns, 4.6 k pixels) is bound by what its tiles pull out of L2: the
            # 128 px x 64 ch tile halves the weight traffic per FLOP (121 -> 111 us alone; +0.2 % frames/s in an A/B of whole frames)
            dcfg = _tile_cfg("deconv", i)
            lv["deconv"] = kernels.pack_deconv(de[0].weight, sc, sh, tile_cfg=dcfg)
            packed["levels"].append(lv)
        return packed

    def _forward_train(self, data_dict):
        """Training forward, base_bev_backbone.py:228-279: the memory-fed and the point-fed canvases go through the SAME
        weights (two streams, shared scale stream); BatchNorm uses batch statistics and every call of a shared BN layer
        updates its running statistics.

        One code path: the library's own kernels through hvpr_amd.conv_train — forward / data-gradient convolutions on
        hvpr_conv2d_wino_nhwc_f32 / hvpr_conv2d_nhwc_f32, weight gradients on hvpr_conv2d_wino_wgrad_nhwc_f32 /
        hvpr_conv2d_wgrad_nhwc_f32, train-mode BatchNorm + ReLU on the hvpr_bn_* kernels, NHWC activations end to end.  CPU
        tensors and unsupported channel counts raise (the torch form of this forward is test infrastructure:
        tests/torch_forms.py)."""
        if not data_dict["spatial_features"].is_cuda:
            raise RuntimeError("hvpr_amd: BaseBEVBackbone_Scale's training forward needs GPU tensors (the HIP path has no CPU fallback)")
        from . import conv_train as ct

        def nhwc(t):
            return t.permute(0, 2, 3, 1).contiguous()          # no copy for the channels_last canvases of the scatter

        def cbr(seq, t, gate=None, resid=None):
            """Sequential of [ZeroPad2d(1)?, Conv2d 3x3 (no bias), BatchNorm2d, ReLU] groups (:154-169, :171-175, :200-209): the
            explicit zero pad + pad-0 conv of the strided entries is the same pad-1 convolution.  gate / resid: the SFM step
            gate * cbr(t) + resid folded into the last BatchNorm + ReLU."""
            mods, k = list(seq), 0
            while k < len(mods):
                if isinstance(mods[k], nn.ZeroPad2d):
                    k += 1
                conv, bn = mods[k], mods[k + 1]
                last = k + 3 >= len(mods)
                if last and gate is not None and resid is t and conv.stride[0] == 1 and conv.kernel_size[0] == 3:
                    t = ct.sfm_step(t, conv.weight, bn, gate)            # the SFM step as one autograd node
                else:
                    z, partials = ct.conv(t, conv.weight, conv.stride[0], stats=True)
                    t = ct.bn_relu(z, bn, gate=gate if last else None, resid=resid if last else None, partials=partials)
                k += 3
            return t

        def gate(y, uses):
            """SpatialAttention on the scale stream (spatial_attention.py:57-63) with batch statistics, forward and backward on
            hvpr_spatial_gate_train_*: ChannelPool -> conv3x3 2 -> 1 -> BatchNorm2d(1) -> sigmoid.  The reference calls the module
            once per SFM step and stream with the SAME input (:250-257): the values are identical every time, so the gate is
            computed once per level (its gradient accumulates over the uses through autograd) and the BatchNorm's running
            statistics receive the `uses` identical updates in closed form."""
            sp = self.attention.spatial
            g, mean, var = ct.spatial_gate_train(y, sp.conv.weight, sp.conv.bias, sp.norm.weight, sp.norm.bias, sp.norm.eps)
            ct.update_running_repeated(sp.norm, mean, var, ct.global_count(y.numel() // y.shape[-1], y.device), uses)
            return g                                                                       # (N,H,W,1)

        x, xp = nhwc(data_dict["spatial_features"]), nhwc(data_dict["spatial_features_point"])
        y = nhwc(data_dict["spatial_scale_features"])
        ups, ups_p = [], []
        for i in range(len(self.blocks)):
            x = cbr(self.blocks[i], x)
            xp = cbr(self.blocks[i], xp)
            y = cbr(self.scale_layers[i], y)
            xa, xpa = x, xp
            nsfm = self.sfm_layer_nums[i]
            if nsfm > 0:
                g = gate(y, 2 * nsfm)
                for _ in range(nsfm):
                    xa = cbr(self.sfmblocks_down[i], xa, gate=g, resid=xa)
                    xpa = cbr(self.sfmblocks_down[i], xpa, gate=g, resid=xpa)
            de = self.deblocks[i]
            ups.append(ct.deconv(xa, de[0].weight))
            ups_p.append(ct.deconv(xpa, de[0].weight))
        # BatchNorm + ReLU of the three branches write straight into their slices of the concatenation (:262-279).  The two streams share
        # the BatchNorm modules: statistics per stream, running statistics updated by the first stream's call and then the second's, as
        # the reference's two passes do
        bns = [de[1] for de in self.deblocks]
        data_dict["spatial_features_2d"] = ct.bn_relu_cat(ups, bns).permute(0, 3, 1, 2)          # (B, 384, H, W), channels_last
        data_dict["spatial_features_point_2d"] = ct.bn_relu_cat(ups_p, bns).permute(0, 3, 1, 2)
        return data_dict

    def forward(self, data_dict):
        """Eval forward.  data_dict["_bev_split"] = {"phase": "a" | "b", "level": L, "x": {i: ...}, "y": ..., "out": ...} runs it in
        two halves on caller-owned boundary buffers (the frame pipeline overlaps the halves of neighbouring frames): phase "a" =
        the whole trunk and the branches of the levels below L (writes x[i] = the trunk output of every level i >= L, y = the scale
        output of level L - 1, and its slices of out = the concat); phase "b" = the branches of the levels >= L (reads x, y;
        finishes out)."""
        if self.training:
            return self._forward_train(data_dict)
        split = data_dict.get("_bev_split")
        phase = split["phase"] if split is not None else None
        n_lv = len(self.blocks)
        L = split["level"] if split is not None else n_lv
        if phase == "b":
            x, y, out = None, split["y"], split["out"]
            B = y.shape[0]
            H, W = self._shape_key
        else:
            sp, sc = data_dict["spatial_features"], data_dict["spatial_scale_features"]
            x = sp.permute(0, 2, 3, 1).contiguous()       # no copy when the scatter produced channels_last
            y = sc.permute(0, 2, 3, 1).contiguous()
            B, H, W, _ = x.shape
            if self._shape_key != (H, W):
                self._fold.invalidate()
                self._shape_key = (H, W)
        P = self._fold.get(y.device, lambda: self._build_packed(H, W))
        gw, gb, gs, gt = P["gate"]
        gw = gw.to(y.device)
        us_all = [int(u) for u in self.upsample_strides]
        if phase is None:
            # output size of the concat: level-0 resolution after its own stride, times its upsample stride
            h0 = (H + 2 - 3) // self.layer_strides[0] + 1
            w0 = (W + 2 - 3) // self.layer_strides[0] + 1
            out = torch.empty((B, h0 * us_all[0], w0 * us_all[0], self.num_bev_features), dtype=torch.float32, device=x.device)
        elif phase == "a":
            out = split["out"]
        # HIP streams: the trunk (blocks of level i+1) does not depend on the attentive branch of level i (scale conv, gate, the
        # three weight-shared SFM steps and the deconv), and the branches depend on each other only through the scale stream
        # y_i = scale_i(y_{i-1}).  The trunk runs on the caller's stream, every branch on a stream of its own (waiting for its
        # x_i and, by event, for y_{i-1}): at batch 1 the upper levels have fewer tiles than the chip has workgroup slots.
        main = torch.cuda.current_stream()
        # phase "b" already runs on a forked stream of the caller's capture: forking again from it (a second level of stream
        # forks inside one hipGraph capture) crashed hipStreamEndCapture on this ROCm build, so its branches stay in line
        two_streams = self.overlap_branches and phase != "b"
        sides = self._branch_streams(y.device, n_lv) if two_streams else []
        coff = 0
        capturing = torch.cuda.is_current_stream_capturing()
        held = []     # tensors another stream reads: referenced until the join, so that the allocator of the producing stream
        #               cannot hand their memory out again while the other stream still reads them
        y_ready = None
        used = []
        planes = self.PRECISIONS[self.conv_precision]
        bf3 = planes > 0
        assert not (bf3 and phase is not None), "the split forward runs the fp32 kernels"
        if bf3:
            x = kernels.split_bf16(x, planes)  # the trunk runs in split-bf16 form from here on
        for i, lv in enumerate(P["levels"]):
            in_b = i >= L
            if phase == "b":
                if not in_b:
                    coff += self.up_filters[i]
                    continue
                x = split["x"][i]
            elif bf3:
                for pc in lv["convs3"]:
                    x = kernels.conv2d_nhwc_bf3(x, pc)
            else:
                for j, pc in enumerate(lv["convs"]):
                    x = kernels.conv2d_nhwc(x, pc, out=split["x"][i] if (phase == "a" and in_b and j == len(lv["convs"]) - 1) else None)
            if phase == "a" and in_b:
                coff += self.up_filters[i]
                continue                                   # this level's branch is phase "b"
            if two_streams:
                side = sides[i]
                used.append(side)
                side.wait_stream(main)          # x_i (and, at level 0, the scale stream's input) are ready for the branch
                if y_ready is not None and sides[i - 1] is not side:
                    side.wait_event(y_ready)    # y_{i-1} from the previous branch's stream
                held.append(x)
                ctx = torch.cuda.stream(side)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                y = kernels.conv2d_nhwc(y, lv["scale"], out=split["y"] if (phase == "a" and i == L - 1) else None)
                if two_streams:
                    y_ready = torch.cuda.Event()
                    y_ready.record(side)
                    held.append(y)
                gate = kernels.spatial_gate(y, gw, gb, gs, gt)
                x_att = x
                nsfm = self.sfm_layer_nums[i]
                for it in range(nsfm):
                    if bf3:     # the last step hands fp32 to the deconvolution only when that one runs on the fp32 kernel
                        x_att = kernels.conv2d_nhwc_bf3(x_att, lv["sfm3"], out_split=(it + 1 < nsfm or lv["deconv3"] is not None),
                                                        gate=gate, resid=x_att)
                    else:
                        x_att = kernels.conv2d_nhwc(x_att, lv["sfm"], gate=gate, resid=x_att)
                    held.append(x_att)
                held.append(gate)
                if bf3 and lv["deconv3"] is not None:
                    kernels.deconv_nhwc_bf3(x_att, lv["deconv3"], out, out_coff=coff)
                elif bf3 and nsfm == 0:
                    kernels.conv2d_nhwc(kernels.unsplit_bf16(x_att), lv["deconv"], out=out, out_coff=coff)
                else:
                    kernels.conv2d_nhwc(x_att, lv["deconv"], out=out, out_coff=coff)
                if two_streams and not capturing:     # eager mode: keep the caching allocator from recycling early
                    for t in (x, y, out):
                        t.record_stream(side)
                    if i + 1 < len(sides):
                        y.record_stream(sides[i + 1])
            coff += self.up_filters[i]
        if two_streams:
            for side in dict.fromkeys(used):          # only the streams this call forked (another phase may own the others)
                main.wait_stream(side)
        held.clear()
        if phase == "a":
            return data_dict
        data_dict["spatial_features_2d"] = out.permute(0, 3, 1, 2)   # (B, 384, H, W), channels_last
        return data_dict

    def split_buffers(self, batch_size, H, W, device, level=None):
        """Boundary buffers of the two-phase forward for canvases of (H, W): x[i] (trunk output of every level i >= level), y
        (scale output of level - 1), out (the concat).  level defaults to the last one; >= 1."""
        n_lv = len(self.blocks)
        if level is None:
            level = n_lv - 1
        assert 1 <= level <= n_lv - 1
        h, w, hw = H, W, []
        for s in self.layer_strides:
            h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
            hw.append((h, w))
        cy = self.scale_layers[level - 1][1].weight.shape[0]
        us0 = int(self.upsample_strides[0])
        mk = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)
        return {"level": level,
                "x": {i: mk(batch_size, hw[i][0], hw[i][1], self.blocks[i][1].weight.shape[0]) for i in range(level, n_lv)},
                "y": mk(batch_size, hw[level - 1][0], hw[level - 1][1], cy),
                "out": mk(batch_size, hw[0][0] * us0, hw[0][1] * us0, self.num_bev_features)}


__all__ = {
    "BaseBEVBackbone_Scale": BaseBEVBackbone_Scale,
}
