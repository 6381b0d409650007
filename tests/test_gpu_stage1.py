"""GPU parity: voxelizer (bit-exact), pillar VFE, memory read-out, scatter — HIP through the C-ABI vs the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from detparams import det_tensor
from hvpr_amd import kernels, synthetic
from oracle import hvpr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VS, RNG = synthetic.HVPR_VOXEL, synthetic.HVPR_RANGE
GRID = [296, 248, 1]


def _gpu_voxelize(frames, max_points, max_voxels, voxel_size=VS, rng=RNG, grid=GRID, cap_mode=0):
    pts = np.concatenate(frames, 0) if len(frames) else np.zeros((0, 4), np.float32)
    offs = np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)
    B = len(frames)
    ws = kernels.VoxelizeWorkspace(B, max(len(pts), 1), grid, DEV)
    out = []
    for _ in range(2):   # second call proves the workspace came back to idle
        v, c, n, vo = kernels.voxelize(torch.from_numpy(pts).to(DEV), torch.from_numpy(offs).to(DEV), B, rng, voxel_size,
                                       grid, max_points, max_voxels, ws, cap_mode=cap_mode)
        torch.cuda.synchronize()
        out.append((v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy(), vo.cpu().numpy()))
    tot = int(out[0][3][-1])
    np.testing.assert_array_equal(out[0][3], out[1][3])
    for a, b in zip(out[0][:3], out[1][:3]):
        np.testing.assert_array_equal(a[:tot], b[:tot])
    return out[0]


def _check_voxelize(frames, max_points, max_voxels, mode="v2", **kw):
    v, c, n, vo = _gpu_voxelize(frames, max_points, max_voxels, cap_mode=1 if mode == "v1" else 0, **kw)
    start = 0
    for b, f in enumerate(frames):
        rv, rc, rn = O.voxelize(f, kw.get("voxel_size", VS), kw.get("rng", RNG), max_points, max_voxels, mode=mode)
        m = rv.shape[0]
        assert vo[b] == start and vo[b + 1] == start + m, (b, vo, start, m)
        np.testing.assert_array_equal(c[start:start + m, 0], b)
        np.testing.assert_array_equal(c[start:start + m, 1:], rc)          # voxel order + zyx coords bit-exact
        np.testing.assert_array_equal(n[start:start + m], rn)
        np.testing.assert_array_equal(v[start:start + m], rv)              # point order inside voxels bit-exact
        start += m
    return v, c, n, vo


def test_voxelize_kitti_like():
    _check_voxelize([synthetic.hvpr_frame(0)], 32, 40000)
    _check_voxelize([synthetic.hvpr_frame(1, shuffle=True)], 32, 16000)


def test_voxelize_batch_ragged_and_empty():
    f = [synthetic.hvpr_frame(2)[:5000], np.zeros((0, 4), np.float32), synthetic.hvpr_frame(3)[:777],
         synthetic.hvpr_frame(4, shuffle=True)]
    _check_voxelize(f, 32, 40000)
    _check_voxelize([np.zeros((0, 4), np.float32), synthetic.hvpr_frame(5)[:100]], 32, 40000)
    _check_voxelize([synthetic.hvpr_frame(5)[:100], np.zeros((0, 4), np.float32)], 32, 40000)


def test_voxelize_multi_frame_batches():
    """ragged batches of 2-4 frames (scan tiles straddle frame borders), an empty frame in the middle, caps in both modes."""
    two = [synthetic.hvpr_frame(20), synthetic.hvpr_frame(21, shuffle=True)]
    _check_voxelize(two, 32, 40000)
    _check_voxelize([two[0], two[1][:9000]], 32, 3000)
    _check_voxelize([two[0], two[1][:9000]], 32, 3000, mode="v1")
    three = [synthetic.hvpr_frame(22), np.zeros((0, 4), np.float32), synthetic.hvpr_frame(23, shuffle=True),
             synthetic.hvpr_frame(24)[:5]]
    _check_voxelize(three, 32, 40000)
    _check_voxelize(three, 32, 3000, mode="v1")
    _check_voxelize([synthetic.hvpr_frame(25)[:2048]], 32, 40000)      # exactly one scan tile
    _check_voxelize([synthetic.hvpr_frame(25)[:2049]], 32, 40000)


def test_voxelize_edges():
    # points exactly on / outside the range borders, z outside, a single point, duplicates
    p = np.array([[0.0, -19.84, -2.5, 0.1], [47.36, 0, 0, 0.2], [47.3599, 19.8399, 0.4999, 0.3], [10, 0, 0.5, 0.4],
                  [10, 0, -2.6, 0.5], [-0.001, 0, 0, 0.6], [10, 19.84, 0, 0.7], [5.0, 5.0, -1.0, 0.8],
                  [5.0, 5.0, -1.0, 0.9], [0.16, -19.84 + 0.16, 0, 1.0]], np.float32)
    _check_voxelize([p], 32, 40000)
    _check_voxelize([p[7:8]], 32, 40000)
    # dense cell: 500 points in one pillar, more than 64 -> exercises the chunked selection
    rng = np.random.default_rng(3)
    q = np.concatenate([rng.uniform([8.0, 0.0, -2, 0], [8.15, 0.15, 0, 1], (500, 4)),
                        synthetic.uniform_frame(9, 2000, RNG)]).astype(np.float32)
    q = q[rng.permutation(len(q))]
    _check_voxelize([q], 32, 40000)
    _check_voxelize([q], 5, 40000)


def test_voxelize_cap_v2_and_v1():
    f = synthetic.uniform_frame(11, 16384, RNG)
    for cap in (100, 4000, 14000):
        _check_voxelize([f, f[::-1].copy()], 32, cap, mode="v2")
        _check_voxelize([f, f[::-1].copy()], 32, cap, mode="v1")


def test_voxelize_dense_nuscenes_scale():
    rng = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    vs = [0.2, 0.2, 8.0]
    f = [synthetic.uniform_frame(20 + i, 200000, rng, n_feat=5) for i in range(2)]
    pts = np.concatenate(f, 0)
    offs = np.array([0, 200000, 400000], np.int32)
    ws = kernels.VoxelizeWorkspace(2, len(pts), [512, 512, 1], DEV)
    v, c, n, vo = kernels.voxelize(torch.from_numpy(pts).to(DEV), torch.from_numpy(offs).to(DEV), 2, rng, vs,
                                   [512, 512, 1], 20, 60000, ws)
    v, c, n, vo = v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy(), vo.cpu().numpy()
    for b in range(2):
        rv, rc, rn = O.voxelize(f[b], vs, rng, 20, 60000)
        assert rv.shape[0] == 60000 and vo[b + 1] - vo[b] == 60000   # the cap is hit
        sl = slice(vo[b], vo[b + 1])
        np.testing.assert_array_equal(c[sl, 1:], rc)
        np.testing.assert_array_equal(n[sl], rn)
        np.testing.assert_array_equal(v[sl], rv)


def test_voxelize_matches_reference_loop_fixture_g13(golden_dir):
    """hvpr_voxelize_f32(cap_mode=1) against fixture G13 — the reference's own in-tree voxel index loop (tools/vis.py:9-60,
    executed as plain Python by tests/golden/make_golden.py): cell -> voxel-id order, border handling, the stop point at the
    cap and the per-voxel counts, bit for bit; also through the fused encode entry point."""
    z = np.load(os.path.join(golden_dir, "g13_voxel_index.npz"))
    for tag in ("nocap", "cap", "dense", "densecap", "cap1"):
        pts, cap = z[tag + "_points"], int(z[tag + "_max_voxels"])
        cells, counts = z[tag + "_cells_zyx"], z[tag + "_counts"]
        v, c, n, vo = _gpu_voxelize([pts], 32, cap, cap_mode=1)
        m = int(vo[-1])
        assert m == len(cells), (tag, m, len(cells))
        np.testing.assert_array_equal(c[:m, 1:], cells)
        np.testing.assert_array_equal(n[:m], np.minimum(counts, 32))
        if m < cap:      # cap not reached: the V2 (`continue`) build must give the same voxels
            v2, c2, n2, vo2 = _gpu_voxelize([pts], 32, cap, cap_mode=0)
            np.testing.assert_array_equal(c2[:m, 1:], cells)
            np.testing.assert_array_equal(n2[:m], np.minimum(counts, 32))


# ---------------------------------------------------------------------------------------------- VFE
def _fold(lin_w, bn_w, bn_b, mean, var, eps=1e-3):
    s = bn_w / np.sqrt(var + eps)
    return (lin_w * s[:, None]).astype(np.float32), (bn_b - mean * s).astype(np.float32)


def _folded_from(params):
    f = {}
    for key, (lin, bn) in {"0": ("pfn_layers.0.linear.weight", "pfn_layers.0.norm"),
                           "1": ("pfn_layers.1.linear.weight", "pfn_layers.1.norm"),
                           "s0": ("pfn_scale_layers.0.0.weight", "pfn_scale_layers.0.1"),
                           "s1": ("pfn_scale_layers.1.0.weight", "pfn_scale_layers.1.1")}.items():
        w, b = _fold(params[lin], params[bn + ".weight"], params[bn + ".bias"], params[bn + ".running_mean"],
                     params[bn + ".running_var"])
        f["w" + key], f["b" + key] = torch.from_numpy(w).to(DEV).contiguous(), torch.from_numpy(b).to(DEV)
    return f


def _vfe_params(seed):
    shapes = {"pfn_layers.0.linear.weight": (16, 10), "pfn_layers.1.linear.weight": (64, 32),
              "pfn_scale_layers.0.0.weight": (16, 5), "pfn_scale_layers.1.0.weight": (32, 16)}
    for bn, c in (("pfn_layers.0.norm", 16), ("pfn_layers.1.norm", 64), ("pfn_scale_layers.0.1", 16), ("pfn_scale_layers.1.1", 32)):
        for k in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{bn}.{k}"] = (c,)
    return {k: det_tensor(k, s, seed) for k, s in shapes.items()}


def _run_vfe(vox, num, coords, params):
    offs = [VS[0] / 2 + RNG[0], VS[1] / 2 + RNG[1], VS[2] / 2 + RNG[2]]
    pf, sf, mask = kernels.pillar_vfe_fwd(torch.from_numpy(vox).to(DEV), torch.from_numpy(num.astype(np.int32)).to(DEV),
                                          torch.from_numpy(coords.astype(np.int32)).to(DEV), _folded_from(params), VS, offs)
    torch.cuda.synchronize()
    return pf.cpu().numpy(), sf.cpu().numpy(), mask.cpu().numpy()


def test_vfe_golden_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_vfe.npz"))
    params = {k[6:]: z[k] for k in z.files if k.startswith("param.")}
    pf, sf, mask = _run_vfe(z["voxels"], z["voxel_num_points"], z["voxel_coords"], params)
    # tolerance: north_star's 1e-3 relative (fp32); observed error is ~1e-6
    np.testing.assert_allclose(pf, z["eval_pillar_features"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(sf, z["eval_pillar_scale_features"], rtol=1e-3, atol=1e-4)
    np.testing.assert_array_equal(mask, z["eval_pillar_mask"])


@pytest.mark.parametrize("seed,M", [(0, None), (1, 1), (2, 3)])
def test_vfe_vs_oracle_on_voxelizer_output(seed, M):
    v, c, n = O.voxelize(synthetic.hvpr_frame(seed), VS, RNG, 32, 40000)
    if M is not None:
        v, c, n = v[:M], c[:M], n[:M]
    coords = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    params = _vfe_params(30 + seed)
    tp = {k: torch.from_numpy(x) for k, x in params.items()}
    rpf, rsf, rmask = O.pillar_vfe_scale(v, n.astype(np.float32), coords.astype(np.float32), tp, VS, RNG)
    pf, sf, mask = _run_vfe(v, n, coords, params)
    np.testing.assert_allclose(pf, rpf.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(sf, rsf.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_array_equal(mask, rmask.numpy())


# ---------------------------------------------------------------------------------------------- memory + scatter
def _check_memory(f, W, k=20):
    out, idx = kernels.memory_readout_fwd(torch.from_numpy(f).to(DEV), torch.from_numpy(W).to(DEV), k, want_idx=True)
    torch.cuda.synchronize()
    rout, ridx, logits = O.memory_readout_eval(f, W, k)
    out, idx, logits = out.cpu().numpy(), idx.cpu().numpy(), logits.numpy()
    # index sets must agree except where the k-th / (k+1)-th logits are within fp32 summation noise
    srt = np.sort(logits, 1)[:, ::-1]
    gap = srt[:, k - 1] - srt[:, k] if logits.shape[1] > k else np.full(len(f), 1.0)
    same = (np.sort(idx, 1) == np.sort(ridx.numpy(), 1)).all(1)
    assert (same | (gap < 1e-5 * np.abs(srt[:, 0]).clip(1e-3))).all()
    ok = same
    np.testing.assert_allclose(out[ok], rout.numpy()[ok], rtol=1e-3, atol=1e-5)
    assert ok.mean() > 0.99
    return out


def test_memory_readout_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "g2_memory_eval.npz"))
    W = det_tensor(str(z["W_name"]), (2000, 64), int(z["W_seed"]))
    out = _check_memory(z["f"], W)
    np.testing.assert_allclose(out, z["output"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("M", [1, 15, 16, 17, 1000])
def test_memory_readout_sizes(M):
    rng = np.random.default_rng(M)
    f = np.maximum(rng.normal(0, 1, (M, 64)), 0).astype(np.float32)
    _check_memory(f, det_tensor("memory.weight", (2000, 64), 5))


def test_memory_readout_ties_take_lowest_indices():
    W = det_tensor("memory.weight", (2000, 64), 6)
    f = np.zeros((3, 64), np.float32)             # all logits tie at 0 -> defined rule: lowest item indices
    out, idx = kernels.memory_readout_fwd(torch.from_numpy(f).to(DEV), torch.from_numpy(W).to(DEV), 20, want_idx=True)
    np.testing.assert_array_equal(np.sort(idx.cpu().numpy(), 1), np.tile(np.arange(20), (3, 1)))
    np.testing.assert_allclose(out.cpu().numpy(), np.tile(W[:20].mean(0), (3, 1)), rtol=1e-5, atol=1e-6)


def test_memory_readout_prefilter_edge_cases():
    """The fp16 pre-filter must never change the exact fp32 top-k: (a) 50 nearly identical bank rows put far more than k items
    inside the filter's error band (candidates 33..64: second exact pass), (b) 300 of them overflow the 64-candidate list
    (exact slow path over all items), (c) features beyond the fp16 range switch the filter off, (d) tiny features and tiny
    bank entries (fp16 subnormal range)."""
    rng = np.random.default_rng(77)
    f = np.maximum(rng.normal(0, 1, (40, 64)), 0).astype(np.float32)
    W = det_tensor("memory.weight", (2000, 64), 8).copy()
    best = int(np.argmax(f[:1] @ W.T))
    for n_close in (50, 300):
        Wc = W.copy()
        ids = rng.choice(2000, n_close, replace=False)
        Wc[ids] = W[best] * (1 + 3e-4 * rng.normal(0, 1, (n_close, 1))) + 1e-4 * rng.normal(0, 1, (n_close, 64))
        _check_memory(f, Wc.astype(np.float32))
    # (the logits themselves keep their usual scale in every case: the oracle takes its top-k on softmax scores, which
    # saturate for huge logits and collapse for tiny ones — memory_module.py:65-66)
    _check_memory((f * 1e-6).astype(np.float32), (W * 1e6).astype(np.float32))        # bank beyond the fp16 range
    _check_memory((f * 1e6).astype(np.float32), (W * 1e-6).astype(np.float32))        # features beyond it, bank in the subnormal range
    ft = f.copy(); ft[:, ::2] *= 1e-6                                                   # half of the channels tiny
    W0 = W.copy(); W0[::3, 1::2] *= 1e-5
    _check_memory(ft.astype(np.float32), W0.astype(np.float32))


def test_scatter_golden_and_idle_workspace(golden_dir):
    z = np.load(os.path.join(golden_dir, "g3_scatter_eval.npz"))
    W = det_tensor(str(z["W_name"]), (2000, 64), int(z["W_seed"]))
    nx, ny, B = int(z["nx"]), int(z["ny"]), int(z["batch_size"])
    pf = torch.from_numpy(z["pillar_features"]).to(DEV)
    sf = torch.from_numpy(z["pillar_scale_features"]).to(DEV)
    coords = torch.from_numpy(z["voxel_coords"].astype(np.int32)).to(DEV)
    mem = kernels.memory_readout_fwd(pf, torch.from_numpy(W).to(DEV), 20)
    ws = kernels.scatter_workspace(B, nx, ny, DEV)
    for _ in range(2):
        sp, sc = kernels.scatter_bev_fwd(pf, mem, sf, coords, B, nx, ny, ws)
        assert sp.shape == (B, 128, ny, nx) and sp.is_contiguous(memory_format=torch.channels_last)
        np.testing.assert_allclose(sp.cpu().numpy(), z["spatial_features"], rtol=1e-3, atol=1e-5)
        np.testing.assert_array_equal(sc.cpu().numpy(), z["spatial_scale_features"])
        assert (ws == -1).all()


def test_scatter_full_grid_vs_oracle():
    v, c, n = O.voxelize(synthetic.hvpr_frame(7), VS, RNG, 32, 40000)
    M = len(c)
    rng = np.random.default_rng(0)
    pf = rng.normal(0, 1, (2 * M, 64)).astype(np.float32)
    mo = rng.normal(0, 1, (2 * M, 64)).astype(np.float32)
    sf = rng.normal(0, 1, (2 * M, 32)).astype(np.float32)
    coords = np.concatenate([np.concatenate([np.full((M, 1), b, np.int32), c], 1) for b in range(2)])
    rsp, rsc = O.scatter_eval(pf, mo, sf, coords.astype(np.float32), 2, 296, 248)
    ws = kernels.scatter_workspace(2, 296, 248, DEV)
    sp, sc = kernels.scatter_bev_fwd(*(torch.from_numpy(a).to(DEV) for a in (pf, mo, sf, coords)), 2, 296, 248, ws)
    np.testing.assert_array_equal(sp.cpu().numpy(), rsp.numpy())
    np.testing.assert_array_equal(sc.cpu().numpy(), rsc.numpy())


def test_fused_memory_scatter_equals_the_two_calls():
    v, c, n = O.voxelize(synthetic.hvpr_frame(9), VS, RNG, 32, 40000)
    M = len(c)
    rng = np.random.default_rng(1)
    pf = torch.from_numpy(rng.normal(0, 1, (2 * M, 64)).astype(np.float32)).to(DEV)
    sf = torch.from_numpy(rng.normal(0, 1, (2 * M, 32)).astype(np.float32)).to(DEV)
    W = torch.from_numpy(rng.uniform(-0.125, 0.125, (2000, 64)).astype(np.float32)).to(DEV)
    coords = torch.from_numpy(np.concatenate([np.concatenate([np.full((M, 1), b, np.int32), c], 1) for b in range(2)])).to(DEV)
    ws = kernels.scatter_workspace(2, 296, 248, DEV)
    mem = kernels.memory_readout_fwd(pf, W, 20)
    sp, sc = kernels.scatter_bev_fwd(pf, mem, sf, coords, 2, 296, 248, ws)
    for md in (None, torch.tensor([2 * M - 100], dtype=torch.int32, device=DEV)):
        mem2, sp2, sc2 = kernels.memory_scatter_fwd(pf, sf, coords, W, 20, 2, 296, 248, ws, m_device=md)
        assert (ws == -1).all()
        if md is None:
            assert torch.equal(mem2, mem) and torch.equal(sp2, sp) and torch.equal(sc2, sc)
        else:      # only the first m_device pillars exist
            spx, scx = kernels.scatter_bev_fwd(pf, mem, sf, coords, 2, 296, 248, ws, m_device=md)
            assert torch.equal(sp2, spx) and torch.equal(sc2, scx)
    # no pillars at all: zero canvases
    e = torch.empty((0, 64), device=DEV)
    _, sp0, sc0 = kernels.memory_scatter_fwd(e, torch.empty((0, 32), device=DEV), torch.empty((0, 4), dtype=torch.int32, device=DEV),
                                             W, 20, 1, 296, 248, kernels.scatter_workspace(1, 296, 248, DEV))
    assert float(sp0.abs().sum()) == 0.0 and float(sc0.abs().sum()) == 0.0


# ---------------------------------------------------------------------------------------------- a1..a4 fused
def _encode_both(frames, max_voxels, cap_mode=0, seed=40, RNG=RNG, VS=VS, GRID=GRID, P=32):
    """hvpr_encode_fwd_f32 against the three separate C-ABI calls on the same inputs (points as (N,5) [b,x,y,z,r])."""
    pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)], 0)
    offs = torch.from_numpy(np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)).to(DEV)
    B = len(frames)
    tp = torch.from_numpy(pts).to(DEV)
    folded = _folded_from(_vfe_params(seed))
    vfe_off = [VS[0] / 2 + RNG[0], VS[1] / 2 + RNG[1], VS[2] / 2 + RNG[2]]
    W = torch.from_numpy(np.random.default_rng(seed).uniform(-0.125, 0.125, (2000, 64)).astype(np.float32)).to(DEV)
    ws = kernels.VoxelizeWorkspace(B, max(len(pts), 1), GRID, DEV)
    v, c, n, vo = kernels.voxelize(tp, offs, B, RNG, VS, GRID, P, max_voxels, ws, xyz_col=1, cap_mode=cap_mode)
    md = vo[B:B + 1]
    pf, sf, mask = kernels.pillar_vfe_fwd(v, n, c, folded, VS, vfe_off, m_device=md)
    mem, sp, sc = kernels.memory_scatter_fwd(pf, sf, c, W, 20, B, GRID[0], GRID[1], kernels.scatter_workspace(B, GRID[0], GRID[1], DEV),
                                             m_device=md)
    outs = []
    # both forms of the index phase (index_mode 1: one launch, 0: three), each twice: the second call proves the voxelizer workspace
    # came back to idle and stale canvases are cleared
    for index_mode in (1, 1, 0, 0):
        r = kernels.encode_fwd(tp, offs, B, RNG, VS, GRID, P, max_voxels, ws, folded, vfe_off, W, 20, xyz_col=1, cap_mode=cap_mode,
                               out=None if not outs else (outs[0]["spatial"], outs[0]["spatial_scale"]), index_mode=index_mode)
        torch.cuda.synchronize()
        outs.append(r)
        m = int(vo[B])
        assert torch.equal(r["voxel_offsets"], vo)
        for key, ref in (("voxels", v), ("coords", c), ("num_points", n), ("pillar_features", pf), ("pillar_scale_features", sf),
                         ("pillar_mask", mask), ("memory_features", mem)):
            assert torch.equal(r[key][:m], ref[:m]), key
        assert torch.equal(r["spatial"], sp) and torch.equal(r["spatial_scale"], sc)
    return m


def test_encode_fused_equals_the_three_calls():
    assert _encode_both([synthetic.hvpr_frame(0)], 40000) > 3000
    _encode_both([synthetic.hvpr_frame(1, shuffle=True)], 16000)
    # ragged batch with an empty frame, then frames whose pillars hold far more than 32 points (64-lane selection path)
    _encode_both([synthetic.hvpr_frame(2)[:5000], np.zeros((0, 4), np.float32), synthetic.hvpr_frame(3)[:777],
                  synthetic.hvpr_frame(4, shuffle=True)], 40000)
    dense = synthetic.hvpr_frame(5).copy()
    dense[:3000, 0] = 10.0 + 0.3 * np.random.default_rng(0).random(3000).astype(np.float32)     # ~2 x 1 cells x 1500 points
    dense[:3000, 1] = 0.05
    _encode_both([dense, synthetic.hvpr_frame(6)[:1]], 40000)


@pytest.mark.parametrize("cap_mode", [0, 1])
def test_encode_fused_with_the_voxel_cap(cap_mode):
    f = synthetic.uniform_frame(11, 16384, RNG)
    for cap in (100, 4000):
        assert _encode_both([f, f[::-1].copy()], cap, cap_mode=cap_mode) == 2 * cap


@pytest.mark.parametrize("case", ["one_frame", "batch", "large_batch"])
def test_encode_fused_capacity_below_the_voxel_count_truncates(case):
    """A C caller whose output buffers hold fewer rows than there are voxels gets the first `capacity` rows and canvases without
    the dropped pillars (include/hvpr_amd.h: 'capacity'), on every path of k_vfe_gather: one frame (first / later passes on
    different waves), a small batch, and a batch large enough for the persistent window walk (no role split); with a 1500-point
    cell (the pass of its own) among the dropped ones.  ADVICE r4: the packed pass used to wait for a dropped pillar for ever."""
    dense = synthetic.hvpr_frame(5).copy()
    dense[-3000:, 0] = 10.0 + 0.3 * np.random.default_rng(0).random(3000).astype(np.float32)
    dense[-3000:, 1] = 0.05
    frames = {"one_frame": [dense], "batch": [synthetic.hvpr_frame(2)[:5000], np.zeros((0, 4), np.float32), dense[-6000:]],
              "large_batch": [synthetic.hvpr_frame(7), dense, synthetic.hvpr_frame(8, shuffle=True)]}[case]
    pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)], 0)
    offs = torch.from_numpy(np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)).to(DEV)
    B, tp = len(frames), torch.from_numpy(pts).to(DEV)
    folded = _folded_from(_vfe_params(3))
    vfe_off = [VS[0] / 2 + RNG[0], VS[1] / 2 + RNG[1], VS[2] / 2 + RNG[2]]
    W = torch.from_numpy(np.random.default_rng(3).uniform(-0.125, 0.125, (2000, 64)).astype(np.float32)).to(DEV)
    ws = kernels.VoxelizeWorkspace(B, len(pts), GRID, DEV)
    full = kernels.encode_fwd(tp, offs, B, RNG, VS, GRID, 32, 40000, ws, folded, vfe_off, W, 20, xyz_col=1)
    torch.cuda.synchronize()
    m = int(full["voxel_offsets"][B])
    for cap in (m - 1, m // 2, 7, 1):
        r = kernels.encode_fwd(tp, offs, B, RNG, VS, GRID, 32, 40000, ws, folded, vfe_off, W, 20, xyz_col=1, capacity=cap)
        torch.cuda.synchronize()
        for key in ("voxels", "coords", "num_points", "pillar_features", "pillar_scale_features", "pillar_mask", "memory_features"):
            assert r[key].shape[0] == cap and torch.equal(r[key], full[key][:cap]), (cap, key)
        sp, sc = full["spatial"].clone(), full["spatial_scale"].clone()
        c = full["coords"][cap:m].long()
        sp[c[:, 0], :, c[:, 2], c[:, 3]] = 0
        sc[c[:, 0], :, c[:, 2], c[:, 3]] = 0
        assert torch.equal(r["spatial"], sp) and torch.equal(r["spatial_scale"], sc), cap
    again = kernels.encode_fwd(tp, offs, B, RNG, VS, GRID, 32, 40000, ws, folded, vfe_off, W, 20, xyz_col=1)   # workspace idle again
    torch.cuda.synchronize()
    assert torch.equal(again["voxel_offsets"], full["voxel_offsets"])
    assert all(torch.equal(again[k][:m], full[k][:m]) for k in ("coords", "num_points", "pillar_features", "memory_features"))
    assert torch.equal(again["spatial"], full["spatial"]) and torch.equal(again["spatial_scale"], full["spatial_scale"])


def test_encode_fused_dense_scene_config5():
    """SURVEY.md §8d config 5 shape: 512 x 512 grid, 20 points per voxel, 200 k uniform points per frame, the 60 k voxel
    cap hit in both frames (several pillars per wave, batch > 1 row lookup, P < 32)."""
    rng = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    f = [synthetic.uniform_frame(80 + i, 200000, rng) for i in range(2)]
    assert _encode_both(f, 60000, RNG=rng, VS=[0.2, 0.2, 8.0], GRID=[512, 512, 1], P=20) == 120000


def test_packed_bank_equals_row_major_and_follows_the_weight():
    """The read-out streams a packed copy of the bank (kernels.PackedBank): same results as the row-major bank, also for a
    bank whose length is not a multiple of 16, and the module's cached copy follows weight updates."""
    from hvpr_amd.map_to_bev import MemoryUnit_Agg
    rng = np.random.default_rng(5)
    f = torch.from_numpy(rng.normal(0, 1, (300, 64)).astype(np.float32)).to(DEV)
    for n_items in (2000, 1999, 37):
        W = torch.from_numpy(rng.uniform(-0.125, 0.125, (n_items, 64)).astype(np.float32)).to(DEV)
        a, ia = kernels.memory_readout_fwd(f, W, 20, want_idx=True)
        b, ib = kernels.memory_readout_fwd(f, kernels.PackedBank(W), 20, want_idx=True)
        assert torch.equal(a, b) and torch.equal(ia.sort(dim=1)[0], ib.sort(dim=1)[0])
    mem = MemoryUnit_Agg(2000, 64).to(DEV).eval()
    out0 = mem(f, 20)["output"]
    assert torch.equal(out0, kernels.memory_readout_fwd(f, mem.weight.detach(), 20))
    with torch.no_grad():
        mem.weight.mul_(-1.5)
    out1 = mem(f, 20)["output"]
    assert torch.equal(out1, kernels.memory_readout_fwd(f, mem.weight.detach(), 20)) and not torch.equal(out0, out1)


def test_encode_fused_persistent_canvases_sparse_clear():
    """canvas_state: the same pair of canvases across different frames (and a frame with fewer pillars, an empty frame, a cap)
    always equals the dense result of a fresh call, and the state equals the occupancy."""
    frames_seq = [[synthetic.hvpr_frame(0), synthetic.hvpr_frame(1)[:4000]], [synthetic.hvpr_frame(2)[:300], synthetic.hvpr_frame(3)],
                  [np.zeros((0, 4), np.float32), synthetic.hvpr_frame(4)[:50]], [synthetic.hvpr_frame(0), synthetic.hvpr_frame(1)[:4000]]]
    folded = _folded_from(_vfe_params(41))
    vfe_off = [VS[0] / 2 + RNG[0], VS[1] / 2 + RNG[1], VS[2] / 2 + RNG[2]]
    W = torch.from_numpy(np.random.default_rng(41).uniform(-0.125, 0.125, (2000, 64)).astype(np.float32)).to(DEV)
    canv, state = kernels.canvas_buffers(2, GRID[0], GRID[1], DEV)
    ws = kernels.VoxelizeWorkspace(2, 40000, GRID, DEV)
    for step, frames in enumerate(frames_seq):
        cap = 40000 if step != 1 else 2000
        pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)], 0)
        offs = torch.from_numpy(np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)).to(DEV)
        tp = torch.from_numpy(pts).to(DEV)
        ref = kernels.encode_fwd(tp, offs, 2, RNG, VS, GRID, 32, cap, ws, folded, vfe_off, W, 20, xyz_col=1)
        got = kernels.encode_fwd(tp, offs, 2, RNG, VS, GRID, 32, cap, ws, folded, vfe_off, W, 20, xyz_col=1, out=canv, state=state)
        torch.cuda.synchronize()
        assert got["spatial"].data_ptr() == canv[0].data_ptr()
        assert torch.equal(got["spatial"], ref["spatial"]) and torch.equal(got["spatial_scale"], ref["spatial_scale"]), step
        m = int(ref["voxel_offsets"][-1])
        c = ref["coords"][:m].long()
        occ = torch.zeros(2 * GRID[0] * GRID[1], dtype=torch.uint8, device=DEV)
        occ[(c[:, 0] * GRID[1] + c[:, 2]) * GRID[0] + c[:, 3]] = 1
        assert torch.equal(state, occ), step


@pytest.mark.parametrize("P", [5, 20, 32])
def test_encode_fused_stress_shapes(P):
    """Odd sizes around the scan-tile and wave boundaries, very dense cells (> 512 points: the chunked-selection fallback of the
    fused gather), few points per voxel allowed (P < 32), several frames, both cap modes — fused == the three calls, and the
    voxelizer == the oracle."""
    rng = np.random.default_rng(100 + P)
    base = synthetic.hvpr_frame(7)
    for n_pts in (1, 63, 64, 65, 511, 512, 513, 1025, 6000):
        f0 = base[:n_pts].copy()
        _encode_both([f0], 40000, P=P)
        _check_voxelize([f0], P, 40000)
    heavy = base[:9000].copy()                                   # 700 points in one cell, 3000 in another, the rest spread
    heavy[:700, 0] = 20.0 + 0.1 * rng.random(700).astype(np.float32); heavy[:700, 1] = 1.0
    heavy[700:3700, 0] = 30.0 + 0.1 * rng.random(3000).astype(np.float32); heavy[700:3700, 1] = -3.0
    rng.shuffle(heavy)
    for cap_mode in (0, 1):
        _encode_both([heavy, base[:100], heavy[::-1].copy()], 500, cap_mode=cap_mode, P=P)
        _encode_both([heavy], 40000, cap_mode=cap_mode, P=P)
    _check_voxelize([heavy, base[:100]], P, 40000)


@pytest.mark.parametrize("n_total", [1, 63, 511, 512, 513, 1023, 1025, 8191, 16383, 16385, 32767, 32768, 32769, 40000])
def test_encode_fused_sizes_around_the_index_kernel_boundaries(n_total):
    """The index phase changes form with the point count (512-point owners up to 16 384 points, 1024-point owners up to 32 768, three
    launches with K1's slots beyond) and its owners' last tiles are ragged: the fused entry point against the separate calls, bit
    for bit, at the sizes where something switches — as one frame and split unevenly over three (one of them empty)."""
    rng = np.random.default_rng(n_total)
    base = np.concatenate([synthetic.hvpr_frame(30 + i, shuffle=True) for i in range(3)])
    pts = base[rng.permutation(len(base))[:n_total]]
    _encode_both([pts], 40000)
    if n_total >= 3:
        a, b = sorted(rng.choice(np.arange(1, n_total), 2, replace=False).tolist())
        _encode_both([pts[:a], np.zeros((0, 4), np.float32), pts[a:b], pts[b:]], 40000, cap_mode=1)


def test_encode_fused_index_phase_under_load_many_times():
    """The one-launch index phase hands data between workgroups through atomics and (when its owners share an XCD) that XCD's L2,
    without a fence: check every word of its results 300 times over while another stream streams through the caches and keeps
    every CU busy (hand-offs that are only right on an idle chip fail under uneven load), alternating two frames so that no
    result can come from the previous call."""
    frames = [synthetic.hvpr_frame(21, shuffle=True), synthetic.hvpr_frame(22)[:9000]]
    folded = _folded_from(_vfe_params(5))
    vfe_off = [VS[0] / 2 + RNG[0], VS[1] / 2 + RNG[1], VS[2] / 2 + RNG[2]]
    W = torch.from_numpy(np.random.default_rng(5).uniform(-0.125, 0.125, (2000, 64)).astype(np.float32)).to(DEV)
    one = torch.tensor([0], dtype=torch.int32, device=DEV)
    pts, offs, ref = [], [], []
    ws = kernels.VoxelizeWorkspace(1, 16384, GRID, DEV)
    for f in frames:
        p = torch.from_numpy(np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)).to(DEV)
        o = torch.cat([one, torch.tensor([len(f)], dtype=torch.int32, device=DEV)])
        r = kernels.encode_fwd(p, o, 1, RNG, VS, GRID, 32, 40000, ws, folded, vfe_off, W, 20, xyz_col=1, index_mode=0)
        torch.cuda.synchronize()
        pts.append(p); offs.append(o); ref.append({k: v.clone() for k, v in r.items() if v is not None})
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)       # 256 MB: through every L2 and the MALL
    stop = 300
    with torch.cuda.stream(side):
        for _ in range(60):
            torch.mm(a, a)
            big.add_(1.0)
    bad = 0
    for it in range(stop):
        j = it & 1
        r = kernels.encode_fwd(pts[j], offs[j], 1, RNG, VS, GRID, 32, 40000, ws, folded, vfe_off, W, 20, xyz_col=1, index_mode=1)
        m = int(ref[j]["voxel_offsets"][1])
        ok = torch.equal(r["voxel_offsets"], ref[j]["voxel_offsets"])
        for k in ("coords", "num_points", "voxels", "pillar_features", "memory_features"):
            ok = ok and torch.equal(r[k][:m], ref[j][k][:m])
        ok = ok and torch.equal(r["spatial"], ref[j]["spatial"]) and torch.equal(r["spatial_scale"], ref[j]["spatial_scale"])
        bad += 0 if ok else 1
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of {stop} encodes differ"
    ws.status()


def _encode_lane(seed, n_pts, index_mode):
    """Everything ONE caller of hvpr_encode_fwd_f32 owns: points, workspace, weights; and the reference result (three launches)."""
    f = synthetic.hvpr_frame(seed, shuffle=True)[:n_pts]
    p = torch.from_numpy(np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)).to(DEV)
    o = torch.tensor([0, len(f)], dtype=torch.int32, device=DEV)
    folded = _folded_from(_vfe_params(seed))
    vfe_off = [VS[0] / 2 + RNG[0], VS[1] / 2 + RNG[1], VS[2] / 2 + RNG[2]]
    W = torch.from_numpy(np.random.default_rng(seed).uniform(-0.125, 0.125, (2000, 64)).astype(np.float32)).to(DEV)
    ws = kernels.VoxelizeWorkspace(1, 16384, GRID, DEV)
    call = lambda mode=index_mode: kernels.encode_fwd(p, o, 1, RNG, VS, GRID, 32, 40000, ws, folded, vfe_off, W, 20, xyz_col=1, index_mode=mode)
    ref = {k: v.clone() for k, v in call(0).items() if v is not None}
    torch.cuda.synchronize()
    return call, ref, ws


def _same(r, ref):
    m = int(ref["voxel_offsets"][1])
    ok = torch.equal(r["voxel_offsets"], ref["voxel_offsets"])
    for k in ("coords", "num_points", "pillar_features", "memory_features"):
        ok = ok and torch.equal(r[k][:m], ref[k][:m])
    return ok and torch.equal(r["spatial"], ref["spatial"]) and torch.equal(r["spatial_scale"], ref["spatial_scale"])


@pytest.mark.parametrize("index_mode", [0, 1])
def test_concurrent_encode_calls_on_four_streams_complete(index_mode):
    """the entry points are re-entrant.  Four callers, each with its own stream, workspace and buffers, keep
    hvpr_encode_fwd_f32 calls in flight at the same time, 200 rounds, while a fifth stream keeps every compute unit busy — all of
    them must complete and give the bits a lone call gives.  index_mode 0 has no workgroup waiting for another; with index_mode 1
    the library takes the one-launch index kernel only when no other one is in flight on the device (same stream as the previous
    one, or that one finished) and the three launches otherwise, so the kernel's owners never wait for compute units held by
    another launch's owners (include/hvpr_amd.h; round 5 documented "at most two in flight" instead of enforcing it)."""
    lanes = [_encode_lane(60 + i, 16384 - 1500 * i, index_mode) for i in range(4)]
    streams = [torch.cuda.Stream() for _ in lanes]
    busy = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(busy):
        for _ in range(40):
            torch.mm(a, a)
    bad = 0
    for it in range(200):
        rs = []
        for (call, ref, ws), st in zip(lanes, streams):
            with torch.cuda.stream(st):
                rs.append(call())
        if it % 20 == 19:
            for st in streams:
                st.synchronize()
            bad += sum(0 if _same(r, ref) else 1 for r, (_, ref, _) in zip(rs, lanes))
    torch.cuda.synchronize()              # "must complete": a hang ends the test through pytest-timeout / the box's limit
    assert bad == 0
    for _, _, ws in lanes:
        ws.status()


def test_one_launch_index_kernel_refuses_a_poisoned_workspace_until_reset():
    """Every wait of the one-launch index kernel is bounded; an owner that gives one up raises a sticky error word in the workspace,
    the launch reports zero pillars, and so does every later one-launch call until the workspace is reset (a launch that could not
    complete must neither hang the device nor hand out partial results).  The give-up itself needs a 2 s stall, so the word is
    raised by hand here (the last carved field of the workspace: 72 ints in a 512-byte slot, error word = int 4)."""
    call, ref, ws = _encode_lane(70, 12000, 1)
    assert _same(call(), ref)
    ws.status()
    words = ws.buf[ws.buf.numel() - 512:].view(torch.int32)
    assert int(words[:72].abs().sum()) == 0                   # census, exit counter, flags: idle after the calls above
    words[4] = 1
    r = call()
    torch.cuda.synchronize()
    assert int(r["voxel_offsets"].abs().sum()) == 0
    with pytest.raises(RuntimeError, match="gave up a wait"):
        ws.status()
    assert int(words[:4].abs().sum()) == 0 and int(words[5:72].abs().sum()) == 0    # the barrier words were not touched
    ws.reset()
    ws.status()
    assert _same(call(), ref)


@pytest.mark.parametrize("env", [{"HVPR_INDEX_FUSED": "0"}, {"HVPR_INDEX_FUSED": "1", "HVPR_INDEX_AGENT": "1"}], ids=["three_launches", "one_launch_device_scope"])
def test_encode_other_index_forms_in_a_child_process(env):
    """The index phase of hvpr_encode_fwd_f32 has two forms — K1 / K2 / K3 as three launches (what more than 32 768 points take;
    forced here for the small cases too) and the one-launch kernel, whose hand-offs stay in one XCD's L2 when its owners share
    that XCD and are device-scope otherwise (forced here: the placement-independent path must give the same bits).  The switch
    is read once per process, hence the child."""
    if os.environ.get("HVPR_INDEX_CHILD"):
        pytest.skip("already the child")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "encode_fused_equals or voxel_cap or capacity or stress or persistent or under_load or boundaries"],
                       env={**os.environ, **env, "HVPR_INDEX_CHILD": "1"}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
