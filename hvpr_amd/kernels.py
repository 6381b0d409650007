"""Tensor-level wrappers over the C-ABI (include/hvpr_amd.h): torch is used for device memory and streams only.

Every wrapper validates its tensors (device, dtype, contiguity), passes raw pointers plus the current HIP
stream, and raises on a non-zero status.  There is no CPU path: CPU tensors are rejected.
"""
import os

import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t, dtype=None, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"hvpr_amd: {name} must live on the GPU (the HIP path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"hvpr_amd: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"hvpr_amd: {name} must be contiguous")
    return t.data_ptr()


# ------------------------------------------------------------------------------------------------ voxelizer
class VoxelizeWorkspace:
    """Persistent device workspace of the voxelizer (two cell maps + per-point scratch), sized for up to `batch`
    frames / `n_points` points per call."""

    def __init__(self, batch, n_points, grid, device):
        self.key = (int(batch), int(n_points), tuple(int(g) for g in grid), torch.device(device))
        nx, ny, nz = self.key[2]
        nbytes = lib().hvpr_voxelize_workspace_bytes(batch, n_points, nx, ny, nz)
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.reset()

    def reset(self):
        batch, n_points, (nx, ny, nz), _ = self.key
        check(lib().hvpr_voxelize_workspace_reset(self.buf.data_ptr(), self.buf.numel(), batch, n_points, nx, ny, nz,
                                                  _stream()), "hvpr_voxelize_workspace_reset")

    def status(self):
        """SYNCHRONISES the current stream; raises when a one-launch index kernel (encode_fwd(index_mode=1)) gave up a wait on this
        workspace since its last reset (that call and every later one reported zero pillars): reset() it."""
        batch, n_points, (nx, ny, nz), _ = self.key
        check(lib().hvpr_voxelize_workspace_status(self.buf.data_ptr(), self.buf.numel(), batch, n_points, nx, ny, nz,
                                                   _stream()), "hvpr_voxelize_workspace_status")


def voxelize(points, frame_offsets, batch, point_cloud_range, voxel_size, grid, max_points, max_voxels, workspace,
             xyz_col=0, n_feat=None, cap_mode=0, capacity=None):
    """points (N, stride) f32 cuda; frame_offsets (batch+1,) i32 cuda.

    Returns voxels (cap, P, n_feat) f32, coords (cap, 4) i32 [b,z,y,x], num_points (cap,) i32,
    voxel_offsets (batch+1,) i32 — rows past voxel_offsets[batch] are unspecified.
    """
    n, stride = points.shape
    n_feat = stride - xyz_col if n_feat is None else n_feat
    if capacity is None:
        capacity = min(n, batch * max_voxels)
    dev = points.device
    voxels = torch.empty((capacity, max_points, n_feat), dtype=torch.float32, device=dev)
    coords = torch.empty((capacity, 4), dtype=torch.int32, device=dev)
    num = torch.empty((capacity,), dtype=torch.int32, device=dev)
    offs = torch.empty((batch + 1,), dtype=torch.int32, device=dev)
    nx, ny, nz = [int(g) for g in grid]
    lo = [float(torch.tensor(v, dtype=torch.float32)) for v in point_cloud_range[:3]]
    vs = [float(torch.tensor(v, dtype=torch.float32)) for v in voxel_size]
    if workspace.key[0] < batch or workspace.key[1] < n:
        raise ValueError("voxelize workspace too small for this call")
    check(lib().hvpr_voxelize_f32(_ptr(points, torch.float32, "points"), n, stride, xyz_col, n_feat,
                                  _ptr(frame_offsets, torch.int32, "frame_offsets"), batch, lo[0], lo[1], lo[2],
                                  vs[0], vs[1], vs[2], nx, ny, nz, int(max_points), int(max_voxels), int(cap_mode),
                                  voxels.data_ptr(), coords.data_ptr(), num.data_ptr(), offs.data_ptr(), capacity,
                                  workspace.buf.data_ptr(), workspace.buf.numel(), workspace.key[0], workspace.key[1],
                                  _stream()), "hvpr_voxelize_f32")
    return voxels, coords, num, offs


# ------------------------------------------------------------------------------------------------ VFE
def pillar_vfe_fwd(voxels, num_points, coords, folded, voxel_size, offsets, m_device=None, want_mask=True):
    """folded: dict w0,b0,w1,b1,ws0,bs0,ws1,bs1 (f32 cuda, BN folded). Returns (pillar, scale, mask|None)."""
    M, P, C = voxels.shape
    if C != 4:
        raise ValueError("hvpr_pillar_vfe_fwd_f32 is built for 4 raw point features")
    dev = voxels.device
    pf = torch.empty((M, 64), dtype=torch.float32, device=dev)
    sf = torch.empty((M, 32), dtype=torch.float32, device=dev)
    mask = torch.empty((M, P, 1), dtype=torch.float32, device=dev) if want_mask else None
    check(lib().hvpr_pillar_vfe_fwd_f32(
        _ptr(voxels, torch.float32, "voxels"), _ptr(num_points, torch.int32, "voxel_num_points"),
        _ptr(coords, torch.int32, "voxel_coords"), M, P, _ptr(m_device, torch.int32, "m_device"),
        float(voxel_size[0]), float(voxel_size[1]), float(voxel_size[2]), float(offsets[0]), float(offsets[1]),
        float(offsets[2]), _ptr(folded["w0"], torch.float32), _ptr(folded["b0"], torch.float32),
        _ptr(folded["w1"], torch.float32), _ptr(folded["b1"], torch.float32), _ptr(folded["ws0"], torch.float32),
        _ptr(folded["bs0"], torch.float32), _ptr(folded["ws1"], torch.float32), _ptr(folded["bs1"], torch.float32),
        pf.data_ptr(), sf.data_ptr(), _ptr(mask), _stream()), "hvpr_pillar_vfe_fwd_f32")
    return pf, sf, mask


# ------------------------------------------------------------------------------------------------ memory + scatter
class PackedBank:
    """Memory bank with its streaming copy for the read-out kernel (hvpr_memory_bank_pack_f32): IEEE fp16 tiles in the matrix-core
    operand layout ([tile of 16 items][channel half][64 lanes] x 8 fp16, 1 KB contiguous per load instruction) + the 64
    channel maxima max_j |W_jc| that bound the pre-filter's rounding error.  The fp32 rows stay next to it: candidates are
    re-checked and the k selected rows are read in exact fp32."""

    def __init__(self, weight):
        w = weight.detach().float().contiguous()
        if w.dim() != 2 or w.shape[1] != 64:
            raise ValueError("PackedBank: (n_items, 64) memory bank expected")
        self.n_items = int(w.shape[0])
        self.shape = w.shape
        self.rows = w                 # row-major copy: the k selected rows are gathered from it
        self.data = torch.empty(lib().hvpr_memory_bank_packed_floats(self.n_items), dtype=torch.float32, device=w.device)
        check(lib().hvpr_memory_bank_pack_f32(_ptr(w, torch.float32, "memory.weight"), self.n_items, self.data.data_ptr(), _stream()),
              "hvpr_memory_bank_pack_f32")


def _bank_args(bank):
    """(row-major pointer, packed pointer, n_items) of a PackedBank; a plain (n_items, 64) tensor is packed here (callers that
    reuse a bank pack it once themselves)."""
    if not isinstance(bank, PackedBank):
        _ptr(bank, torch.float32, "memory.weight")
        bank = PackedBank(bank)
    return bank.rows.data_ptr(), bank.data.data_ptr(), bank.n_items


def memory_readout_fwd(f, bank, k, m_device=None, want_idx=False):
    M, C = f.shape
    out = torch.empty((M, C), dtype=torch.float32, device=f.device)
    idx = torch.empty((M, k), dtype=torch.int32, device=f.device) if want_idx else None
    check(lib().hvpr_memory_readout_fwd_f32(_ptr(f, torch.float32, "pillar_features"), M,
                                            _ptr(m_device, torch.int32, "m_device"), *_bank_args(bank), int(k),
                                            out.data_ptr(), _ptr(idx), _stream()), "hvpr_memory_readout_fwd_f32")
    return (out, idx) if want_idx else out


_SEG_CHUNK = 32      # edges per first-level sum of segment_sum_rows


def edges_by_destination(dst, n_dst):
    """The edges e -> dst[e] of a scatter, grouped by destination for hvpr_segment_sum_rows_f32.  Returns (order, chunk_ptr, dest_ptr):
    order i32 lists the edge ids destination by destination, inside a destination by ascending edge id (a STABLE sort: the summation
    order is a fixed function of the index tensor, so the scattering gradients are reproducible bit for bit, which float atomics are
    not).  A destination's run of edges is cut into chunks of _SEG_CHUNK: chunk_ptr i32 [n_chunks_max + 1] are the chunk boundaries
    inside `order`, dest_ptr i32 [n_dst + 1] the chunks of each destination — a point picked by 2000 pillars becomes 63 short sums on
    63 waves and one sum of 63 partials instead of one 2000-term loop on one wave.  Everything stays on the device: the chunk count
    is bounded by E / _SEG_CHUNK + min(n_dst, E), the unused tail of chunk_ptr is empty chunks.  Destinations outside [0, n_dst)
    are dropped (they sort to the two ends)."""
    dst = dst.reshape(-1).to(torch.int64)
    dev = dst.device
    E = dst.numel()
    n_max = E // _SEG_CHUNK + min(n_dst, E) + 1
    if E == 0:
        z = torch.zeros(1, dtype=torch.int32, device=dev)
        return torch.zeros(0, dtype=torch.int32, device=dev), z.expand(n_max + 1).contiguous(), z.expand(n_dst + 1).contiguous()
    order = torch.argsort(dst, stable=True)
    sdst = dst[order].contiguous()
    rowptr = torch.searchsorted(sdst, torch.arange(n_dst + 1, dtype=torch.int64, device=dev))   # edges of d: rowptr[d] .. rowptr[d+1]
    end_live = rowptr[n_dst:]                                                   # (1,)
    pos = torch.arange(E, dtype=torch.int64, device=dev)
    live = (sdst >= 0) & (sdst < n_dst)
    first = rowptr[sdst.clamp(0, n_dst - 1)]                                    # first edge of this edge's destination
    head = live & (((pos - first) % _SEG_CHUNK) == 0)                           # this edge opens a chunk
    csum = torch.cumsum(head.to(torch.int64), 0)
    cid, total = csum - 1, csum[-1:]                                            # chunk of every live edge; number of chunks
    chunk_ptr = end_live.expand(n_max + 1).clone()                              # chunks past the last one: empty, at the end
    chunk_ptr.scatter_(0, torch.where(head, cid, torch.full_like(cid, n_max)), torch.where(head, pos, end_live.expand(E)))
    # chunks of destination d: from the chunk its first edge opens (for an empty d: the next destination's) to the next one's
    dest_ptr = torch.where(rowptr < end_live, cid[rowptr.clamp(max=E - 1)], total.expand(n_dst + 1))
    return order.to(torch.int32), chunk_ptr.to(torch.int32), dest_ptr.to(torch.int32)


def segment_sum_rows(src, src_off, C, edge_row, edge_w, chunk_ptr, dest_ptr, n_dst):
    """dst (n_dst, C): dst[d] = sum over d's edges, in the order given, of edge_w[e] * src[edge_row[e], off : off + C] — two passes of
    hvpr_segment_sum_rows_f32: per chunk of <= _SEG_CHUNK edges, then per destination over its chunks (fixed order: reproducible)."""
    n_chunks = chunk_ptr.numel() - 1
    part = torch.empty((n_chunks, C), dtype=torch.float32, device=src.device)
    check(lib().hvpr_segment_sum_rows_f32(_ptr(src, torch.float32, "src"), src.shape[-1], int(src_off), int(C), _ptr(edge_row, torch.int32, "edge_row"),
                                          _ptr(edge_w, torch.float32, "edge_w"), _ptr(chunk_ptr, torch.int32, "chunk_ptr"), int(n_chunks),
                                          part.data_ptr(), int(C), _stream()), "hvpr_segment_sum_rows_f32")
    dst = torch.empty((n_dst, C), dtype=torch.float32, device=src.device)
    check(lib().hvpr_segment_sum_rows_f32(part.data_ptr(), int(C), 0, int(C), None, None, _ptr(dest_ptr, torch.int32, "dest_ptr"), int(n_dst),
                                          dst.data_ptr(), int(C), _stream()), "hvpr_segment_sum_rows_f32")
    return dst


def scatter_workspace(batch, nx, ny, device):
    """Idle cell map (-1 everywhere); every call returns it to idle."""
    n = lib().hvpr_scatter_workspace_bytes(batch, nx, ny) // 4
    return torch.full((n,), -1, dtype=torch.int32, device=device)


def scatter_bev_fwd(pillar, memory, scale, coords, batch, nx, ny, workspace, m_device=None):
    """Returns spatial (B, Cp+Cm, ny, nx) and spatial_scale (B, Cs, ny, nx) in channels_last memory format."""
    M, cp = pillar.shape
    cm = 0 if memory is None else memory.shape[1]
    cs = 0 if scale is None else scale.shape[1]
    dev = pillar.device
    spatial = torch.empty((batch, ny, nx, cp + cm), dtype=torch.float32, device=dev)
    spatial_scale = torch.empty((batch, ny, nx, cs), dtype=torch.float32, device=dev) if cs else None
    check(lib().hvpr_scatter_bev_fwd_f32(_ptr(pillar, torch.float32, "pillar_features"), cp, _ptr(memory, torch.float32),
                                         cm, _ptr(scale, torch.float32), cs, _ptr(coords, torch.int32, "voxel_coords"),
                                         M, _ptr(m_device, torch.int32), batch, nx, ny, spatial.data_ptr(),
                                         _ptr(spatial_scale), workspace.data_ptr(), workspace.numel() * 4, _stream()),
          "hvpr_scatter_bev_fwd_f32")
    spatial = spatial.permute(0, 3, 1, 2)
    if spatial_scale is not None:
        spatial_scale = spatial_scale.permute(0, 3, 1, 2)
    return spatial, spatial_scale


# ------------------------------------------------------------------------------------------------ split-bf16 convolutions
class PackedConvBf3:
    """3x3 conv weights split into `planes` bf16 planes, layout [9, Cin/8, planes, cout_pad, 8] (BatchNorm scale folded)."""

    def __init__(self, w, bias, cin, cout, cout_pad, stride, relu, tile_cfg, planes):
        self.w, self.bias, self.cin, self.cout, self.cout_pad = w, bias, cin, cout, cout_pad
        self.stride, self.relu, self.tile_cfg, self.planes = stride, relu, tile_cfg, planes


def _planes_of(x, planes):
    out, r = [], x
    for _ in range(planes):
        h = r.to(torch.bfloat16)
        out.append(h)
        r = r - h.float()
    return out


def pack_conv_bf3(weight, scale=None, shift=None, stride=1, relu=True, tile_cfg=0, planes=2):
    cout, cin, kh, kw = weight.shape
    assert kh == kw == 3 and cin % 16 == 0 and cout % 4 == 0 and planes in (2, 3)
    w = weight.detach().float()
    if scale is not None:
        w = w * scale.view(-1, 1, 1, 1)
    cout_pad = (cout + 63) // 64 * 64
    wp = torch.zeros((9, cin // 8, planes, cout_pad, 8), dtype=torch.bfloat16, device=w.device)
    for k, part in enumerate(_planes_of(w, planes)):       # (Cout, Cin, 3, 3) -> tap, chunk, ci, co -> tap, chunk, co, ci
        wp[:, :, k, :cout, :] = part.permute(2, 3, 1, 0).reshape(9, cin // 8, 8, cout).permute(0, 1, 3, 2)
    b = torch.zeros((cout_pad,), dtype=torch.float32, device=w.device)
    if shift is not None:
        b[:cout] = shift.detach().float()
    return PackedConvBf3(wp.contiguous(), b, cin, cout, cout_pad, stride, relu, tile_cfg, planes)


def pack_deconv_bf3(weight, scale, shift, relu=True, planes=2):
    """ConvTranspose2d weight (Cin, Cout, s, s), kernel == stride == s -> 1x1 GEMM with s*s*Cout columns, split planes."""
    cin, cout, s, s2 = weight.shape
    assert s == s2 and cin % 64 == 0 and cout % 4 == 0 and planes in (2, 3)
    w = weight.detach().float() * scale.view(1, -1, 1, 1)
    cols = s * s * cout
    cout_pad = (cols + 63) // 64 * 64
    g = w.permute(0, 2, 3, 1).reshape(cin, cols)                          # column = (ky*s + kx)*Cout + co
    wp = torch.zeros((1, cin // 8, planes, cout_pad, 8), dtype=torch.bfloat16, device=w.device)
    for k, part in enumerate(_planes_of(g, planes)):
        wp[0, :, k, :cols, :] = part.reshape(cin // 8, 8, cols).permute(0, 2, 1)
    b = torch.zeros((cout_pad,), dtype=torch.float32, device=w.device)
    b[:cols] = shift.detach().float().repeat(s * s)
    pc = PackedConvBf3(wp.contiguous(), b, cin, cout, cout_pad, 1, relu, 1, planes)
    pc.up = s
    return pc


def deconv_nhwc_bf3(xs, pc, out, out_coff=0):
    """xs split-bf16 NHWC (N,H,W,Cin/8,planes,8) -> fp32 into out[..., out_coff:out_coff+Cout] at pc.up x the resolution."""
    N, H, W, groups, planes, _ = xs.shape
    assert groups * 8 == pc.cin and planes == pc.planes and xs.is_contiguous()
    assert out.shape[:3] == (N, H * pc.up, W * pc.up) and out.is_contiguous() and out.dtype == torch.float32
    check(lib().hvpr_deconv_nhwc_bf16x3(xs.data_ptr(), N, H, W, pc.cin, pc.w.data_ptr(), pc.bias.data_ptr(), pc.cout, pc.cout_pad,
                                        pc.up, 1 if pc.relu else 0, out.data_ptr(), out.shape[-1], out_coff, planes, _stream()),
          "hvpr_deconv_nhwc_bf16x3")
    return out


def split_bf16(x, planes=2):
    """fp32 NHWC (..., C), C % 8 == 0 -> split-bf16 NHWC as a bf16 tensor (..., C/8, planes, 8)."""
    assert x.is_contiguous() and x.shape[-1] % 8 == 0 and x.dtype == torch.float32
    out = torch.empty((*x.shape[:-1], x.shape[-1] // 8, planes, 8), dtype=torch.bfloat16, device=x.device)
    check(lib().hvpr_split_bf16_f32(_ptr(x, torch.float32, "activations"), x.numel(), planes, out.data_ptr(), _stream()),
          "hvpr_split_bf16_f32")
    return out


def unsplit_bf16(xs):
    """split-bf16 NHWC (..., C/8, planes, 8) -> fp32 NHWC (..., C)."""
    assert xs.is_contiguous() and xs.dtype == torch.bfloat16 and xs.shape[-1] == 8
    planes = xs.shape[-2]
    out = torch.empty((*xs.shape[:-3], xs.shape[-3] * 8), dtype=torch.float32, device=xs.device)
    check(lib().hvpr_unsplit_bf16_f32(xs.data_ptr(), out.numel(), planes, out.data_ptr(), _stream()), "hvpr_unsplit_bf16_f32")
    return out


def conv2d_nhwc_bf3(xs, pc, out_split=True, gate=None, resid=None):
    """xs: split-bf16 NHWC (N,H,W,Cin/8,planes,8).  Returns split-bf16 NHWC (out_split) or fp32 NHWC (N,OH,OW,Cout)."""
    N, H, W, groups, planes, _ = xs.shape
    cin = groups * 8
    assert cin == pc.cin and planes == pc.planes and xs.is_contiguous()
    OH, OW = (H + 2 - 3) // pc.stride + 1, (W + 2 - 3) // pc.stride + 1
    if out_split:
        assert pc.cout % 8 == 0
        out = torch.empty((N, OH, OW, pc.cout // 8, planes, 8), dtype=torch.bfloat16, device=xs.device)
    else:
        out = torch.empty((N, OH, OW, pc.cout), dtype=torch.float32, device=xs.device)
    check(lib().hvpr_conv2d_nhwc_bf16x3(xs.data_ptr(), N, H, W, cin, pc.w.data_ptr(), pc.bias.data_ptr(), pc.stride, pc.cout,
                                        pc.cout_pad, 1 if pc.relu else 0, _ptr(gate, torch.float32, "gate"),
                                        None if resid is None else resid.data_ptr(), 0 if resid is None else resid.shape[3] * 8,
                                        out.data_ptr(), 1 if out_split else 0, pc.cout, 0, pc.tile_cfg, planes, _stream()),
          "hvpr_conv2d_nhwc_bf16x3")
    return out


def memory_scatter_fwd(pillar, scale, coords, bank, k, batch, nx, ny, workspace, m_device=None, out=None):
    """Fused a3+a4 (64+64+32 channels): returns memory_features (M,64), spatial (B,128,ny,nx), spatial_scale (B,32,ny,nx).
    `out` = (spatial, spatial_scale) as returned by an earlier call: write into those canvases instead of new ones."""
    M = pillar.shape[0]
    dev = pillar.device
    mem = torch.empty((M, 64), dtype=torch.float32, device=dev)
    if out is not None:
        spatial, spatial_scale = (o.permute(0, 2, 3, 1) for o in out)
        if spatial.shape != (batch, ny, nx, 128) or spatial_scale.shape != (batch, ny, nx, 32) or not spatial.is_contiguous() \
                or not spatial_scale.is_contiguous():
            raise ValueError("memory_scatter_fwd: `out` canvases do not match this call")
    else:
        spatial = torch.empty((batch, ny, nx, 128), dtype=torch.float32, device=dev)
        spatial_scale = torch.empty((batch, ny, nx, 32), dtype=torch.float32, device=dev)
    if pillar.shape[1] != 64 or scale.shape[1] != 32 or bank.shape[1] != 64:
        raise ValueError("memory_scatter_fwd is specialised for 64 pillar / 64 memory / 32 scale channels")
    check(lib().hvpr_memory_scatter_fwd_f32(_ptr(pillar, torch.float32, "pillar_features"), _ptr(scale, torch.float32, "scale"),
                                            _ptr(coords, torch.int32, "voxel_coords"), M, _ptr(m_device, torch.int32),
                                            *_bank_args(bank), int(k), batch, nx, ny,
                                            mem.data_ptr(), spatial.data_ptr(), spatial_scale.data_ptr(), workspace.data_ptr(),
                                            workspace.numel() * 4, _stream()), "hvpr_memory_scatter_fwd_f32")
    return mem, spatial.permute(0, 3, 1, 2), spatial_scale.permute(0, 3, 1, 2)


def frame_offsets(points, batch):
    """(batch+1,) i32 frame offsets of a collated (N, C) point tensor whose column 0 is the batch index (one launch)."""
    offs = torch.empty((batch + 1,), dtype=torch.int32, device=points.device)
    check(lib().hvpr_frame_offsets_f32(_ptr(points, torch.float32, "points"), points.shape[0], points.shape[1], int(batch),
                                       offs.data_ptr(), _stream()), "hvpr_frame_offsets_f32")
    return offs


def encode_fwd(points, frame_offsets, batch, point_cloud_range, voxel_size, grid, max_points, max_voxels, workspace, folded,
               offsets, bank, k, xyz_col=0, cap_mode=0, capacity=None, want_voxels=True, want_mask=True, out=None, state=None,
               index_mode=0):
    """a1..a4 fused (hvpr_encode_fwd_f32): raw points -> canvases in five launches (three with index_mode=1: the index phase as ONE
    launch, for a caller with a single encode lane per device — include/hvpr_amd.h), bit-identical to voxelize ->
    pillar_vfe_fwd -> memory_scatter_fwd.  Returns a dict with voxels|None, coords, num_points, voxel_offsets,
    pillar_features, pillar_scale_features, pillar_mask|None, memory_features, spatial (B,128,ny,nx), spatial_scale
    (B,32,ny,nx) (channels_last).  `out` = (spatial, spatial_scale) of an earlier call: write into those canvases.
    `state` (with `out`): the uint8 (B*ny*nx,) occupancy that travels with those canvases (see canvas_buffers): only the
    cells the previous call left non-zero are cleared instead of the whole 47 MB."""
    n, stride = points.shape
    n_feat = stride - xyz_col
    if capacity is None:
        capacity = min(n, batch * max_voxels)
    dev = points.device
    nx, ny, nz = [int(g) for g in grid]
    voxels = torch.empty((capacity, max_points, n_feat), dtype=torch.float32, device=dev) if want_voxels else None
    coords = torch.empty((capacity, 4), dtype=torch.int32, device=dev)
    num = torch.empty((capacity,), dtype=torch.int32, device=dev)
    offs = torch.empty((batch + 1,), dtype=torch.int32, device=dev)
    pf = torch.empty((capacity, 64), dtype=torch.float32, device=dev)
    sf = torch.empty((capacity, 32), dtype=torch.float32, device=dev)
    mem = torch.empty((capacity, 64), dtype=torch.float32, device=dev)
    mask = torch.empty((capacity, max_points, 1), dtype=torch.float32, device=dev) if want_mask else None
    if out is not None:
        spatial, spatial_scale = (o.permute(0, 2, 3, 1) for o in out)
        if spatial.shape != (batch, ny, nx, 128) or spatial_scale.shape != (batch, ny, nx, 32) or not spatial.is_contiguous() \
                or not spatial_scale.is_contiguous():
            raise ValueError("encode_fwd: `out` canvases do not match this call")
    else:
        spatial = torch.empty((batch, ny, nx, 128), dtype=torch.float32, device=dev)
        spatial_scale = torch.empty((batch, ny, nx, 32), dtype=torch.float32, device=dev)
    if bank.shape[1] != 64:
        raise ValueError("encode_fwd is specialised for 64 pillar / 64 memory / 32 scale channels")
    if state is not None and (out is None or state.dtype != torch.uint8 or state.numel() != batch * ny * nx or not state.is_contiguous()):
        raise ValueError("encode_fwd: `state` is the (B*ny*nx,) uint8 occupancy of the `out` canvases")
    lo = [float(torch.tensor(v, dtype=torch.float32)) for v in point_cloud_range[:3]]
    vs = [float(torch.tensor(v, dtype=torch.float32)) for v in voxel_size]
    if workspace.key[0] < batch or workspace.key[1] < n:
        raise ValueError("voxelize workspace too small for this call")
    check(lib().hvpr_encode_fwd_f32(
        _ptr(points, torch.float32, "points"), n, stride, xyz_col, n_feat, _ptr(frame_offsets, torch.int32, "frame_offsets"),
        batch, lo[0], lo[1], lo[2], vs[0], vs[1], vs[2], nx, ny, nz, int(max_points), int(max_voxels), int(cap_mode),
        float(offsets[0]), float(offsets[1]), float(offsets[2]),
        _ptr(folded["w0"], torch.float32), _ptr(folded["b0"], torch.float32), _ptr(folded["w1"], torch.float32),
        _ptr(folded["b1"], torch.float32), _ptr(folded["ws0"], torch.float32), _ptr(folded["bs0"], torch.float32),
        _ptr(folded["ws1"], torch.float32), _ptr(folded["bs1"], torch.float32), *_bank_args(bank), int(k), _ptr(voxels), coords.data_ptr(), num.data_ptr(), offs.data_ptr(), capacity, pf.data_ptr(),
        sf.data_ptr(), _ptr(mask), mem.data_ptr(), spatial.data_ptr(), spatial_scale.data_ptr(), _ptr(state), workspace.buf.data_ptr(),
        workspace.buf.numel(), workspace.key[0], workspace.key[1], int(index_mode), _stream()), "hvpr_encode_fwd_f32")
    return {"voxels": voxels, "coords": coords, "num_points": num, "voxel_offsets": offs, "pillar_features": pf,
            "pillar_scale_features": sf, "pillar_mask": mask, "memory_features": mem,
            "spatial": spatial.permute(0, 3, 1, 2), "spatial_scale": spatial_scale.permute(0, 3, 1, 2)}


def canvas_buffers(batch, nx, ny, device):
    """A persistent pair of zeroed canvases + their occupancy state for encode_fwd(out=..., state=...): (spatial (B,128,ny,nx),
    spatial_scale (B,32,ny,nx)) in channels_last memory format and the uint8 state.  Nobody else may write them."""
    sp = torch.zeros((batch, ny, nx, 128), dtype=torch.float32, device=device).permute(0, 3, 1, 2)
    sc = torch.zeros((batch, ny, nx, 32), dtype=torch.float32, device=device).permute(0, 3, 1, 2)
    return (sp, sc), torch.zeros((batch * ny * nx,), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------ convolutions
class PackedConv:
    """Weights of one conv layer in the kernel's layout [taps, Cin/8, 2, cout_pad, 4] with BatchNorm folded."""

    __slots__ = ("w", "bias", "taps", "stride", "cout", "cout_pad", "up", "cin", "relu", "tile_cfg")

    def __init__(self, w, bias, taps, stride, cout, cout_pad, up, cin, relu, tile_cfg):
        self.w, self.bias, self.taps, self.stride = w, bias, taps, stride
        self.cout, self.cout_pad, self.up, self.cin, self.relu, self.tile_cfg = cout, cout_pad, up, cin, relu, tile_cfg


def _tile_channels(tile_cfg):
    return 128 if tile_cfg == 0 else 64


def pack_conv(weight, scale=None, shift=None, stride=1, relu=True, tile_cfg=0):
    """weight (Cout, Cin, k, k) with k in {1,3}; scale/shift: folded BatchNorm (per Cout) or None / conv bias."""
    cout, cin, kh, kw = weight.shape
    assert kh == kw and kh in (1, 3) and cin % 8 == 0
    w = weight.detach().float()
    if scale is not None:
        w = w * scale.view(-1, 1, 1, 1)
    tc = _tile_channels(tile_cfg)
    cout_pad = (cout + tc - 1) // tc * tc
    taps = kh * kw
    # (Cout, Cin, kh, kw) -> (taps, Cin/8, 2 halves, Cout, 4)
    p = w.permute(2, 3, 1, 0).reshape(taps, cin // 8, 2, 4, cout).permute(0, 1, 2, 4, 3)
    wp = torch.zeros((taps, cin // 8, 2, cout_pad, 4), dtype=torch.float32, device=w.device)
    wp[:, :, :, :cout] = p
    b = torch.zeros((cout_pad,), dtype=torch.float32, device=w.device)
    if shift is not None:
        b[:cout] = shift.detach().float()
    return PackedConv(wp.contiguous(), b, taps, stride, cout, cout_pad, 1, cin, relu, tile_cfg)


_zero_bias = {}


class PackedConvWino:
    """Stride-1 3x3 conv weights in the Winograd kernel's stage image (hvpr_conv2d_wino_pack_f32) with BatchNorm folded."""

    __slots__ = ("w", "bias", "cout", "cout_pad", "cin", "relu", "px_groups")
    taps, stride, up = 9, 1, 1

    def __init__(self, w, bias, cout, cout_pad, cin, relu, px_groups):
        self.w, self.bias, self.cout, self.cout_pad, self.cin, self.relu, self.px_groups = w, bias, cout, cout_pad, cin, relu, px_groups


def pack_conv_wino(weight, scale=None, shift=None, relu=True, px_groups=1, adjoint=False):
    """weight (Cout, Cin, 3, 3) [adjoint: (Cin, Cout, 3, 3) of the layer whose data gradient is wanted]; scale / shift as in
    pack_conv.  Packs on the device (one small kernel), so the training step can repack every call."""
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    cout, cin = (w.shape[1], w.shape[0]) if adjoint else (w.shape[0], w.shape[1])
    assert w.shape[2:] == (3, 3) and cin % 8 == 0 and cout % 4 == 0
    cout_pad = (cout + 63) // 64 * 64
    wp = torch.empty((lib().hvpr_conv2d_wino_packed_floats(cin, cout),), dtype=torch.float32, device=w.device)
    sc = None if scale is None else scale.detach().float().contiguous()
    check(lib().hvpr_conv2d_wino_pack_f32(_ptr(w, torch.float32, "conv weight"), _ptr(sc, torch.float32, "scale"), cout, cin,
                                          1 if adjoint else 0, wp.data_ptr(), _stream()), "hvpr_conv2d_wino_pack_f32")
    if shift is None:                     # raw convolutions (training: packed every call) share one zero bias per size
        key = (cout_pad, w.device)
        if key not in _zero_bias:
            _zero_bias[key] = torch.zeros((cout_pad,), dtype=torch.float32, device=w.device)
        b = _zero_bias[key]
    else:
        b = torch.zeros((cout_pad,), dtype=torch.float32, device=w.device)
        b[:cout] = shift.detach().float()
    return PackedConvWino(wp, b, cout, cout_pad, cin, relu, px_groups)


def conv2d_wino_nhwc(x, pc, out=None, out_coff=0, gate=None, resid=None, bn_partials=None):
    """x (N,H,W,Cin) contiguous f32 -> (N,H,W,C): 3x3 / stride 1 / pad 1 by Winograd F(2x2,3x3); arguments as conv2d_nhwc.
    bn_partials: (hvpr_conv2d_wino_stats_rows(N,H,W), 2, C) f32 to receive the per-tile sums of the raw output (training)."""
    N, H, W, cin = x.shape
    assert cin == pc.cin
    if out is None:
        out = torch.empty((N, H, W, pc.cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (N, H, W) and out.is_contiguous()
    check(lib().hvpr_conv2d_wino_nhwc_f32(_ptr(x, torch.float32, "conv input"), N, H, W, cin, pc.w.data_ptr(), pc.bias.data_ptr(),
                                          pc.cout, 1 if pc.relu else 0, _ptr(gate, torch.float32, "gate"),
                                          _ptr(resid, torch.float32, "resid"), 0 if resid is None else resid.shape[-1],
                                          out.data_ptr(), out.shape[-1], int(out_coff), pc.px_groups,
                                          _ptr(bn_partials, torch.float32, "bn_partials"), _stream()), "hvpr_conv2d_wino_nhwc_f32")
    return out


def pack_deconv(weight, scale, shift, relu=True, tile_cfg=0):
    """ConvTranspose2d weight (Cin, Cout, s, s) with kernel == stride == s -> 1x1 GEMM with s*s*Cout columns."""
    cin, cout, s, s2 = weight.shape
    assert s == s2 and cin % 8 == 0
    w = weight.detach().float() * scale.view(1, -1, 1, 1)
    cols = s * s * cout
    tc = _tile_channels(tile_cfg)
    cout_pad = (cols + tc - 1) // tc * tc
    # column = (ky*s + kx)*Cout + co
    g = w.permute(0, 2, 3, 1).reshape(cin, cols)                       # (Cin, cols)
    p = g.reshape(cin // 8, 2, 4, cols).permute(0, 1, 3, 2).unsqueeze(0)     # (1, Cin/8, 2, cols, 4)
    wp = torch.zeros((1, cin // 8, 2, cout_pad, 4), dtype=torch.float32, device=w.device)
    wp[:, :, :, :cols] = p
    b = torch.zeros((cout_pad,), dtype=torch.float32, device=w.device)
    b[:cols] = shift.detach().float().repeat(s * s)
    return PackedConv(wp.contiguous(), b, 1, 1, cout, cout_pad, s, cin, relu, tile_cfg)


def conv_algo():
    """Algorithm of the stride-1 3x3 convolutions: "winograd" (hvpr_conv2d_wino_nhwc_f32, default) or "direct"
    (hvpr_conv2d_nhwc_f32; HVPR_CONV_ALGO=direct).  Every other shape always takes the direct kernel."""
    algo = os.environ.get("HVPR_CONV_ALGO", "winograd")
    if algo not in ("winograd", "direct"):
        raise ValueError(f"HVPR_CONV_ALGO must be 'winograd' or 'direct', got {algo!r}")
    return algo


def pack_conv_auto(weight, scale=None, shift=None, stride=1, relu=True, tile_cfg=1, px_groups=1):
    """pack_conv_wino for a 3x3 / stride-1 layer when conv_algo() is "winograd", pack_conv otherwise; conv2d_nhwc takes either."""
    if weight.shape[2] == 3 and stride == 1 and weight.shape[0] % 4 == 0 and weight.shape[1] % 8 == 0 and conv_algo() == "winograd":
        return pack_conv_wino(weight, scale, shift, relu=relu, px_groups=px_groups)
    return pack_conv(weight, scale, shift, stride=stride, relu=relu, tile_cfg=tile_cfg)


def conv2d_nhwc(x, pc, out=None, out_coff=0, gate=None, resid=None):
    """x (N,H,W,Cin) contiguous f32 -> (N,OH*up,OW*up,C) ; optional fused y = gate*y + resid (SFM step)."""
    if isinstance(pc, PackedConvWino):
        return conv2d_wino_nhwc(x, pc, out=out, out_coff=out_coff, gate=gate, resid=resid)
    N, H, W, cin = x.shape
    assert cin == pc.cin
    if pc.taps == 9:
        OH, OW = (H + 2 - 3) // pc.stride + 1, (W + 2 - 3) // pc.stride + 1
    else:
        OH, OW = H, W
    if out is None:
        out = torch.empty((N, OH * pc.up, OW * pc.up, pc.cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (N, OH * pc.up, OW * pc.up) and out.is_contiguous()
    check(lib().hvpr_conv2d_nhwc_f32(_ptr(x, torch.float32, "conv input"), N, H, W, cin, pc.w.data_ptr(),
                                     pc.bias.data_ptr(), pc.taps, pc.stride, pc.cout, pc.cout_pad, pc.up,
                                     1 if pc.relu else 0, _ptr(gate, torch.float32, "gate"),
                                     _ptr(resid, torch.float32, "resid"), 0 if resid is None else resid.shape[-1],
                                     out.data_ptr(), out.shape[-1], int(out_coff), pc.tile_cfg, _stream()),
          "hvpr_conv2d_nhwc_f32")
    return out


def pack_conv_s2_dgrad(weight):
    """The weight image conv2d_s2_dgrad_nhwc multiplies with: the filter with its channel axes swapped, taps not flipped."""
    return pack_conv(weight.detach().permute(1, 0, 2, 3), None, None, stride=1, relu=False, tile_cfg=1)


def conv2d_s2_dgrad_nhwc(dz, weight, H, W, pc=None):
    """Data gradient of a 3x3 stride-2 (pad 1) convolution with `weight` (Cout, Cin, 3, 3): dz (N,OH,OW,Cout) -> dx (N,H,W,Cin), gathered
    per output-pixel parity class (hvpr_conv2d_s2_dgrad_nhwc_f32: 2.25 taps per pixel instead of a stride-1 convolution's 9 over a
    zero-upsampled dz)."""
    N, OH, OW, cout = dz.shape
    cin = weight.shape[1]
    assert weight.shape[0] == cout and tuple(weight.shape[2:]) == (3, 3) and OH == (H + 2 - 3) // 2 + 1 and OW == (W + 2 - 3) // 2 + 1
    if pc is None:
        pc = pack_conv_s2_dgrad(weight)                                                                    # (Cin, Cout, 3, 3): taps not flipped
    dx = torch.empty((N, H, W, cin), dtype=torch.float32, device=dz.device)
    check(lib().hvpr_conv2d_s2_dgrad_nhwc_f32(_ptr(dz, torch.float32, "dz"), N, OH, OW, cout, pc.w.data_ptr(), pc.bias.data_ptr(), cin,
                                              pc.cout_pad, H, W, dx.data_ptr(), cin, 0, _stream()), "hvpr_conv2d_s2_dgrad_nhwc_f32")
    return dx


# ------------------------------------------------------------------------------------------------ gate / decode / top-k / NMS
def spatial_gate(y_nhwc, w18, conv_bias, bn_scale, bn_shift):
    N, H, W, C = y_nhwc.shape
    gate = torch.empty((N, H, W), dtype=torch.float32, device=y_nhwc.device)
    check(lib().hvpr_spatial_gate_f32(_ptr(y_nhwc, torch.float32, "scale features"), N, H, W, C,
                                      _ptr(w18, torch.float32, "gate conv weight"), float(conv_bias), float(bn_scale),
                                      float(bn_shift), gate.data_ptr(), _stream()), "hvpr_spatial_gate_f32")
    return gate


def head_decode(head_nhwc, n_anchor, n_class, n_dir_bins, x_shifts, y_shifts, anchor_table, dir_offset, dir_limit_offset,
                period, want_cls=True, want_scores=True, out=None):
    N, H, W, CH = head_nhwc.shape
    A = H * W * n_anchor
    dev = head_nhwc.device
    if out is not None:      # (cls, box, scores, labels) of an earlier call with the same shapes
        cls, box, scores, labels = out
        if box.shape != (N, A, 7) or not all(t is None or t.is_contiguous() for t in out):
            raise ValueError("head_decode: `out` tensors do not match this call")
    else:
        cls = torch.empty((N, A, n_class), dtype=torch.float32, device=dev) if want_cls else None
        box = torch.empty((N, A, 7), dtype=torch.float32, device=dev)
        scores = torch.empty((N, A), dtype=torch.float32, device=dev) if want_scores else None
        labels = torch.empty((N, A), dtype=torch.int32, device=dev) if want_scores else None
    check(lib().hvpr_head_decode_f32(_ptr(head_nhwc, torch.float32, "head output"), N, H, W, CH, n_anchor, n_class,
                                     n_dir_bins, _ptr(x_shifts, torch.float32), _ptr(y_shifts, torch.float32),
                                     _ptr(anchor_table, torch.float32), float(dir_offset), float(dir_limit_offset),
                                     float(period), _ptr(cls), box.data_ptr(), _ptr(scores), _ptr(labels), _stream()),
          "hvpr_head_decode_f32")
    return cls, box, scores, labels


class PostWorkspace:
    """Device scratch for score top-k + NMS of `batch` frames with `n_scores` anchors each."""

    def __init__(self, batch, n_scores, pre_max, device):
        self.batch, self.n_scores, self.pre_max = batch, n_scores, pre_max
        # zero-filled once: every hvpr_score_topk_f32 call leaves its counters and histogram zeroed again
        self.topk = torch.zeros(lib().hvpr_score_topk_workspace_bytes(batch, n_scores), dtype=torch.uint8, device=device)
        self.nms = torch.empty(lib().hvpr_nms_workspace_bytes(pre_max), dtype=torch.uint8, device=device)


def score_topk(scores, score_thresh, pre_max, ws, want_scores=True):
    """scores (B, A) -> order (B, pre_max) i32, sorted scores (B, pre_max), counts (B,) i32."""
    B, A = scores.shape
    dev = scores.device
    order = torch.empty((B, pre_max), dtype=torch.int32, device=dev)
    ss = torch.empty((B, pre_max), dtype=torch.float32, device=dev) if want_scores else None
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    use = score_thresh is not None
    check(lib().hvpr_score_topk_f32(_ptr(scores, torch.float32, "scores"), B, A, float(score_thresh) if use else 0.0,
                                    1 if use else 0, int(pre_max), order.data_ptr(), _ptr(ss), counts.data_ptr(),
                                    ws.topk.data_ptr(), ws.topk.numel(), _stream()), "hvpr_score_topk_f32")
    return order, ss, counts


def nms_bev(boxes, order, n_device, n_max, thresh, max_keep, ws_nms, map_through_order=True):
    """boxes (R, >=7) f32; order (n_max,) i32 or None.  Returns keep (max_keep,) i32, keep_count (1,) i32."""
    dev = boxes.device
    keep = torch.zeros((max(max_keep, 1),), dtype=torch.int32, device=dev)   # rows past keep_count stay valid ids
    kc = torch.empty((1,), dtype=torch.int32, device=dev)
    check(lib().hvpr_nms_bev_f32(_ptr(boxes, torch.float32, "boxes"), boxes.shape[1], _ptr(order, torch.int32),
                                 _ptr(n_device, torch.int32), int(n_max), float(thresh), int(max_keep),
                                 1 if map_through_order else 0, keep.data_ptr(), kc.data_ptr(), ws_nms.data_ptr(),
                                 ws_nms.numel(), _stream()), "hvpr_nms_bev_f32")
    return keep, kc


def gather_predictions(boxes, scores, labels, keep):
    """boxes (A,>=7), scores (A,), labels (A,) i32, keep (K,) i32 -> pred_boxes (K,7), pred_scores (K,), pred_labels (K,) i64,
    selected (K,) i64 in one launch."""
    K = keep.shape[0]
    dev = boxes.device
    ob = torch.empty((K, 7), dtype=torch.float32, device=dev)
    os_ = torch.empty((K,), dtype=torch.float32, device=dev)
    ol = torch.empty((K,), dtype=torch.int64, device=dev)
    osel = torch.empty((K,), dtype=torch.int64, device=dev)
    check(lib().hvpr_gather_predictions_f32(_ptr(boxes, torch.float32, "boxes"), boxes.shape[1], _ptr(scores, torch.float32, "scores"),
                                            _ptr(labels, torch.int32, "labels"), _ptr(keep, torch.int32, "keep"), K, ob.data_ptr(),
                                            os_.data_ptr(), ol.data_ptr(), osel.data_ptr(), _stream()), "hvpr_gather_predictions_f32")
    return ob, os_, ol, osel


def boxes_pairwise(a, b, mode):
    """mode 0: BEV overlap area, 1: BEV IoU, 2: 3D IoU.  a (N,7), b (M,7) -> (N,M)."""
    a = a[:, :7].contiguous()
    b = b[:, :7].contiguous()
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(lib().hvpr_boxes_pairwise_f32(_ptr(a, torch.float32, "boxes_a"), a.shape[0], _ptr(b, torch.float32, "boxes_b"),
                                        b.shape[0], int(mode), out.data_ptr(), _stream()), "hvpr_boxes_pairwise_f32")
    return out
