"""The algebra behind csrc/conv_wino.hip and csrc/conv_wino_wgrad.hip, checked in numpy (float64) on the CPU: the F(2x2, 3x3)
transforms reproduce a 3x3 / stride-1 correlation (what nn.Conv2d computes, base_bev_backbone.py:154-175), the adjoint pack
gives the data gradient, and Gt (sum_blocks (A dY At) . (Bt d B)) G gives the weight gradient — including the sign convention
the weight-gradient kernel uses (last row / column of A left positive in the operands, fixed up by the reducer)."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def corr2d(x, g):
    """x (H, W), g (3, 3), pad 1, stride 1."""
    H, W = x.shape
    xp = np.pad(x, 1)
    return np.array([[np.sum(xp[i:i + 3, j:j + 3] * g) for j in range(W)] for i in range(H)])


def wino_blocks(x):
    """4x4 input patches of the 2x2 output blocks of an even-sized image (pad 1)."""
    H, W = x.shape
    xp = np.pad(x, 1)
    return {(by, bx): xp[2 * by:2 * by + 4, 2 * bx:2 * bx + 4] for by in range(H // 2) for bx in range(W // 2)}


def test_forward_transform_equals_correlation():
    rng = np.random.default_rng(0)
    x, g = rng.standard_normal((8, 6)), rng.standard_normal((3, 3))
    U = G @ g @ G.T
    y = np.zeros_like(x)
    for (by, bx), d in wino_blocks(x).items():
        y[2 * by:2 * by + 2, 2 * bx:2 * bx + 2] = AT @ (U * (BT @ d @ BT.T)) @ AT.T
    np.testing.assert_allclose(y, corr2d(x, g), rtol=0, atol=1e-12)


def test_rows_of_bt_have_two_non_zeros_as_the_kernel_assumes():
    """Wave a reads two patch rows (r0, r1) and forms t = d[r0] + s * d[r1]: a=0: d0 - d2, a=1: d1 + d2, a=2: d2 - d1, a=3: d1 - d3."""
    sel = [(0, 2, -1.0), (1, 2, 1.0), (2, 1, -1.0), (1, 3, -1.0)]
    for a, (r0, r1, s) in enumerate(sel):
        row = np.zeros(4)
        row[r0] += 1.0
        row[r1] += s
        np.testing.assert_array_equal(row, BT[a])
    # columns of V from t: (t0 - t2, t1 + t2, t2 - t1, t1 - t3)
    t = np.arange(4, dtype=np.float64) + 1
    np.testing.assert_array_equal(BT @ t, [t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]])


def test_adjoint_pack_is_the_data_gradient():
    """d(sum(y * dy))/dx of y = corr(x, g) equals corr(dy, flip(g)) — hvpr_conv2d_wino_pack_f32(adjoint) packs w'[o][i][u][v] =
    w[i][o][2-u][2-v]."""
    rng = np.random.default_rng(1)
    x, g, dy = rng.standard_normal((6, 6)), rng.standard_normal((3, 3)), rng.standard_normal((6, 6))
    eps, num = 1e-6, np.zeros_like(x)
    for i in range(6):
        for j in range(6):
            e = np.zeros_like(x)
            e[i, j] = eps
            num[i, j] = np.sum((corr2d(x + e, g) - corr2d(x - e, g)) * dy) / (2 * eps)
    np.testing.assert_allclose(corr2d(dy, g[::-1, ::-1]), num, atol=1e-6)


def test_weight_gradient_in_the_winograd_domain():
    rng = np.random.default_rng(2)
    x, dy = rng.standard_normal((8, 8)), rng.standard_normal((8, 8))
    want = np.zeros((3, 3))
    xp = np.pad(x, 1)
    for u in range(3):
        for v in range(3):
            want[u, v] = np.sum(xp[u:u + 8, v:v + 8] * dy)
    A = AT.T
    dU = np.zeros((4, 4))
    dU_kernel = np.zeros((4, 4))          # the kernel's operands: last row / column of A taken positive
    A_pos = np.abs(A) * np.array([[1, 1], [1, 1], [1, -1], [1, 1]])     # only A[2][1] = -1 stays (a = 2: dY0 - dY1)
    for (by, bx), d in wino_blocks(x).items():
        dyb = dy[2 * by:2 * by + 2, 2 * bx:2 * bx + 2]
        V = BT @ d @ BT.T
        dU += (A @ dyb @ A.T) * V
        dU_kernel += (A_pos @ dyb @ A_pos.T) * V
    np.testing.assert_allclose(G.T @ dU @ G, want, atol=1e-10)
    sign = np.ones((4, 4))
    sign[3, :] *= -1
    sign[:, 3] *= -1                       # k_wgrad_wino_reduce: negative when exactly one of (a == 3), (b == 3)
    np.testing.assert_allclose(G.T @ (dU_kernel * sign) @ G, want, atol=1e-10)


def test_f4x4_rounding_error_in_fp32_against_f2x2():
    """The evidence behind DESIGN.md §4.2c ("next factor"): Winograd F(4x4, 3x3) with fp32 transforms and fp32 accumulation over 128
    input channels sits ~15x above F(2x2, 3x3) in rounding error (5.5e-6 vs 3.8e-7 of the output scale at 128 channels; 1.1e-5 ..
    1.3e-5 at 256 / 512) — inside a 2e-5 per-layer budget, not inside F(2x2)'s 1e-6 class."""
    rng = np.random.default_rng(0)
    mats = {
        2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
            np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
            np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)),
        4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                      [0, 4, 0, -5, 0, 1]], np.float64),
            np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], np.float64),
            np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)),
    }
    C, K, H = 128, 16, 8
    x = np.maximum(rng.normal(0, 1, (C, H, H)), 0).astype(np.float32)
    w = (rng.normal(0, 1, (K, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    xp64 = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    ref = sum(np.einsum("kc,chw->khw", w[:, :, u, v].astype(np.float64), xp64[:, u:u + H, v:v + H]) for u in range(3) for v in range(3))
    err = {}
    for m, (bt, g, at) in mats.items():
        t = m + 2
        U = np.einsum("ab,kcbd,ed->kcae", g, w.astype(np.float64), g).astype(np.float32)     # filter transform in double, rounded once
        btf, atf = bt.astype(np.float32), at.astype(np.float32)
        xp = np.pad(x, ((0, 0), (1, 1 + t), (1, 1 + t)))
        out = np.zeros((K, H, H), np.float32)
        for by in range(0, H, m):
            for bx in range(0, H, m):
                V = np.einsum("ab,cbd,ed->cae", btf, xp[:, by:by + t, bx:bx + t], btf).astype(np.float32)
                M = np.einsum("kcae,cae->kae", U, V).astype(np.float32)
                Y = np.einsum("ab,kbd,ed->kae", atf, M, atf).astype(np.float32)
                hh, ww = min(m, H - by), min(m, H - bx)
                out[:, by:by + hh, bx:bx + ww] = Y[:, :hh, :ww]
        err[m] = float(np.abs(out - ref).max() / np.abs(ref).max())
    assert err[2] < 2e-6 and err[4] < 2e-5, err            # both algebraically exact ...
    assert err[4] > 4 * err[2], err                        # ... and F(4x4) pays for its 1/24 .. 8 transform constants
