"""Lane-level numpy model of the depthwise-taps-on-the-matrix-pipe formulation (cosypose_amd/csrc/kernels_wave.hip, kernels_smx.hip) against a plain
depthwise convolution with 'same' zero padding (MBConvBlock._depthwise_conv, cosypose/models/efficientnet.py:80-83).

The model executes what the kernels do with registers and lanes -- v_mfma_f32_4x4x4's operand layout (measured: profiles/exp/mfma_depthwise.hip), the Toeplitz
fragments as the host packs them (wave_pack_params / small_mx_pack_params), the quad_perm moves and selects that build the halo operands -- so an index slip in
the formulation shows up here, on the CPU, without a kernel.  The device kernels themselves are held by tests/test_gpu_parity.py
(test_fused_kernels_vs_storage_emulation, block kinds 5 and 6)."""
import numpy as np
import pytest

LANES = np.arange(64)
CB, JQ = LANES >> 2, LANES & 3          # small-MFMA roles: block (= channel of the 16-channel tile), lane in the block
OFF = [[0, 1, 2, 3], [-2, -1, 4, 5]]     # pixel offsets (from the quad's first pixel) of operand 0 (the quad) and operand 1 (its two-pixel halos)


def quad_perm(v, sel):
    """lane l <- lane (l & ~3) + sel[l & 3]"""
    return v[(LANES & ~3) + np.array(sel)[LANES & 3]]


def mma4(a, b, c):
    """v_mfma_f32_4x4x4_16B: per block of 4 lanes D[i][j] = sum_k A[lane 4b+i][k] * B[lane 4b+j][k]; D lands in lane 4b+j, register i"""
    d = c.copy()
    for blk in range(16):
        A, B = a[4 * blk:4 * blk + 4], b[4 * blk:4 * blk + 4]        # [i][k], [j][k]
        d[4 * blk:4 * blk + 4] += (A @ B.T).T                        # [j][i]
    return d


def toeplitz(w, ks):
    """the host-packed fragments: A[ky][operand][lane][k] = w[ky][OFF[operand][k] - i + LO][channel of the lane], i = lane & 3"""
    lo = (ks - 1) // 2
    A = np.zeros((ks, 2, 64, 4))
    for ky in range(ks):
        for m in range(2):
            for l in range(64):
                for q in range(4):
                    kx = OFF[m][q] - (l & 3) + lo
                    if 0 <= kx < ks:
                        A[ky, m, l, q] = w[ky, kx, CB[l]]
    return A


def depthwise_same(E, w):
    ks = w.shape[0]
    lo = (ks - 1) // 2
    H, W, C = E.shape
    Ep = np.zeros((H + 2 * lo, W + 2 * lo, C))
    Ep[lo:lo + H, lo:lo + W] = E
    out = np.zeros_like(E)
    for ky in range(ks):
        for kx in range(ks):
            out += w[ky, kx] * Ep[ky:ky + H, kx:kx + W]
    return out


@pytest.mark.parametrize('ks', [3, 5])
@pytest.mark.parametrize('ppl', [1, 2, 4])
def test_wave_kernel_segments(ks, ppl):
    """rows of PPL 16-pixel segments, lane (channel, quad j) holds the pixels 16 q + 4 j .. + 3 of segment q; input-stationary over the rows"""
    rng = np.random.RandomState(ks * 10 + ppl)
    H, W, lo = 9, 16 * ppl, (ks - 1) // 2
    E, w = rng.randn(H, W, 16), rng.randn(ks, ks, 16)
    A = toeplitz(w, ks)
    out = np.zeros((H, W, 16))
    acc = np.zeros((H + 2 * lo, ppl, 64, 4))                       # accumulators of the output rows (the kernel keeps KS of them open)
    for iy in range(H):
        seg = np.zeros((ppl, 64, 4))
        for q in range(ppl):
            for l in range(64):
                seg[q, l] = E[iy, 16 * q + 4 * JQ[l]:16 * q + 4 * JQ[l] + 4, CB[l]]
        for q in range(ppl):
            lo_d, hi_d = seg[q][:, 0:2], seg[q][:, 2:4]
            hp, ln = quad_perm(hi_d, [0, 0, 1, 2]), quad_perm(lo_d, [1, 2, 3, 3])
            hp0 = quad_perm(seg[q - 1][:, 2:4], [3, 0, 1, 2]) if q > 0 else np.zeros((64, 2))          # the neighbouring segment's last / first quad
            ln3 = quad_perm(seg[q + 1][:, 0:2], [1, 2, 3, 0]) if q < ppl - 1 else np.zeros((64, 2))
            hp = np.where((JQ == 0)[:, None], hp0, hp)
            ln = np.where((JQ == 3)[:, None], ln3, ln)
            w2 = np.concatenate([hp, ln], axis=1)
            for ky in range(ks):                                   # input row iy is tap row ky of output row iy + lo - ky
                oy = iy + lo - ky
                acc[oy + lo, q] = mma4(A[ky, 0], seg[q], acc[oy + lo, q])
                acc[oy + lo, q] = mma4(A[ky, 1], w2, acc[oy + lo, q])
    for oy in range(H):
        for q in range(ppl):
            for l in range(64):
                out[oy, 16 * q + 4 * JQ[l]:16 * q + 4 * JQ[l] + 4, CB[l]] = acc[oy + lo, q, l]
    assert np.abs(out - depthwise_same(E, w)).max() < 1e-12


@pytest.mark.parametrize('ks', [3, 5])
def test_small_map_kernel(ks):
    """8x8 maps: segment = two map rows, quad j = 2 * (row of the segment) + (half of the row); wave w produces output segment w from the input
    segments w - 1, w, w + 1 (zero segments above and below the map), odd row offsets with the row halves of the lanes swapped"""
    rng = np.random.RandomState(ks)
    lo = (ks - 1) // 2
    E, w = rng.randn(8, 8, 16), rng.randn(ks, ks, 16)
    A = toeplitz(w, ks)
    Eh = np.zeros((6, 64, 4))
    for s in range(4):
        for l in range(64):
            Eh[s + 1, l] = E[2 * s + (JQ[l] >> 1), 4 * (JQ[l] & 1):4 * (JQ[l] & 1) + 4, CB[l]]
    row0, half0 = JQ < 2, (JQ & 1) == 0
    out = np.zeros((8, 8, 16))
    for wave in range(4):
        S = [Eh[wave + q] for q in range(3)]
        Sw = [quad_perm(s, [2, 3, 0, 1]) for s in S]
        acc = np.zeros((64, 4))
        for ky in range(ks):
            d = ky - lo
            op = {-2: S[0], 0: S[1], 2: S[2]}.get(d)
            if d == -1:
                op = np.where(row0[:, None], Sw[0], Sw[1])
            elif d == 1:
                op = np.where(row0[:, None], Sw[1], Sw[2])
            hp = np.where(half0[:, None], 0.0, quad_perm(op[:, 2:4], [0, 0, 2, 2]))
            ln = np.where(half0[:, None], quad_perm(op[:, 0:2], [1, 1, 3, 3]), 0.0)
            acc = mma4(A[ky, 0], op, acc)
            acc = mma4(A[ky, 1], np.concatenate([hp, ln], axis=1), acc)
        for l in range(64):
            out[2 * wave + (JQ[l] >> 1), 4 * (JQ[l] & 1):4 * (JQ[l] & 1) + 4, CB[l]] = acc[l]
    assert np.abs(out - depthwise_same(E, w)).max() < 1e-12


def test_identity_mfma_transposes_lane_and_register_axes():
    """v_mfma_f32_16x16x16 against the identity: A operand lane (row r = lane & 15, k-group lane >> 4) holding Data[r][4 kq .. 4 kq + 3] -> C lane (column n, row
    group) holding Data[4 rq .. 4 rq + 3][n]: (channel, 4 pixels) becomes (pixel, 4 channels), the layout of the 8-byte D store"""
    data = np.random.RandomState(0).randn(16, 16)                  # [channel][pixel]
    a = np.zeros((64, 4)); ident = np.zeros((64, 4))
    for l in range(64):
        a[l] = data[l & 15, 4 * (l >> 4):4 * (l >> 4) + 4]
        ident[l] = [(4 * (l >> 4) + e) == (l & 15) for e in range(4)]
    Amat, Bmat = np.zeros((16, 16)), np.zeros((16, 16))            # A[r][k], B[k][n] from the fragment layouts
    for l in range(64):
        Amat[l & 15, 4 * (l >> 4):4 * (l >> 4) + 4] = a[l]
        Bmat[4 * (l >> 4):4 * (l >> 4) + 4, l & 15] = ident[l]
    C = Amat @ Bmat
    for l in range(64):                                            # C lane (n = lane & 15, rq = lane >> 4) holds C[4 rq + e][n]
        got = C[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
        assert np.array_equal(got, data[4 * (l >> 4):4 * (l >> 4) + 4, l & 15])
