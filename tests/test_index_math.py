"""CPU restatements of the index arithmetic the HIP convolution kernels rely on, checked against torch's own
convolutions in float64. They pin the formulas (not the kernels — those are compared on the GPU in test_hip_conv.py):

  * the incremental im2col walk of the weight-gradient kernel (csrc/dir_conv_wgrad.hip: WgWalk / wg_advance and the
    constants dir_conv_wgrad() derives) against a direct (n, ho, wo) decode, for ordinary and degenerate geometries;
  * the decomposition of a 3x3 / stride-2 / pad-1 data gradient into four stride-1 convolutions by output-pixel parity,
    with the class-packed weight layout of conv_prep_rot_index (csrc/dir_conv.hip), against autograd;
  * the compact stride-2 1x1 gradient added at the even pixels (dir_conv_dgrad_join) against autograd;
  * the 24-element row windows and packed weights of the stem kernels (csrc/dir_stem.hip), forward and weight gradient;
  * the LDS image of the all-taps 3x3 weight gradient (csrc/dir_conv_wgrad3.hip): DMA piece -> (row, physical chunk) with the
    half-swap swizzle on the SOURCE side, the transposing-read lane mapping (as pinned on the GPU by dir_probe_tr16), the per-lane
    patch slot of a pixel, taps as row offsets, operand k order and the C/D layout — an emulated workgroup reproduces dW;
  * the patch-staged 3x3 kernel's addressing (csrc/dir_conv.hip, conv3x3_patch_kernel): the tap row is an immediate under the
    XOR swizzle because the row pitch is a multiple of 16 rows; fragment address = base ^ (kk << 5);
  * the split-K reductions of the weight gradients (csrc/dir_conv_wgrad.hip): the per-layer kernel and the one-launch-for-all-layers kernel
    (round 5) visit every element exactly once and add its partials in the SAME order — emulated in float32, bit-equal.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------
def _walk_consts(H, W, Cin, R, S, stride, pad):
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    dw, dh, dn = 64 % Wo, (64 // Wo) % Ho, 64 // (Wo * Ho)
    return dict(Ho=Ho, Wo=Wo, WoS=Wo * stride, HoS=Ho * stride, c1=(stride * W - Wo * stride) * Cin * 2,
                c2=(H * W - Ho * stride * W) * Cin * 2, dws64=dw * stride, dhs64=dh * stride, dn64=dn,
                c64=(dn * H * W + dh * stride * W + dw * stride) * Cin * 2, cstep=stride * Cin * 2)


def _advance(st, c, stride, dws, dhs, dn, cbase):                 # wg_advance
    n, hs, ws, off = st
    ws += dws
    carry1 = ws >= c["WoS"]
    if carry1:
        ws -= c["WoS"]
    hs += dhs + (stride if carry1 else 0)
    carry2 = hs >= c["HoS"]
    if carry2:
        hs -= c["HoS"]
    return (n + dn + (1 if carry2 else 0), hs, ws, off + cbase + (c["c1"] if carry1 else 0) + (c["c2"] if carry2 else 0))


def _direct(m, H, W, Cin, stride, c, trp, tsp, coff):
    n, ho, wo = m // (c["Ho"] * c["Wo"]), (m // c["Wo"]) % c["Ho"], m % c["Wo"]
    hs, ws = ho * stride, wo * stride
    return (n, hs, ws, (((n * H + hs + trp) * W + ws + tsp) * Cin + coff) * 2)


@pytest.mark.parametrize("geo", [(56, 56, 3, 3, 1, 1), (56, 56, 3, 3, 2, 1), (14, 14, 3, 3, 1, 1), (7, 7, 3, 3, 1, 1), (9, 13, 3, 3, 1, 1),
                                 (5, 3, 3, 3, 2, 1), (112, 112, 3, 3, 1, 1), (224, 224, 3, 3, 2, 1), (1, 1, 1, 1, 1, 0), (56, 56, 1, 1, 2, 0),
                                 (3, 200, 3, 3, 1, 1), (200, 3, 3, 3, 1, 1), (8, 12, 3, 3, 2, 1), (2, 2, 3, 3, 2, 1)])
def test_incremental_im2col_walk_equals_direct_decode(geo):
    H, W, R, S, stride, pad = geo
    Cin = 64
    c = _walk_consts(H, W, Cin, R, S, stride, pad)
    rows = 3 * c["Ho"] * c["Wo"] + 7
    for tr in range(R):
        for ts in range(S):
            trp, tsp = tr - pad, ts - pad
            for start in (0, 4, 200):                            # first row of a thread: any multiple of 4
                st, m = _direct(start, H, W, Cin, stride, c, trp, tsp, 8), start
                for _ in range(rows // 64 + 2):                  # K-steps
                    w = st
                    for j in range(4):                           # the thread's 4 consecutive rows
                        assert w == _direct(m + j, H, W, Cin, stride, c, trp, tsp, 8), (geo, tr, ts, m + j)
                        if j < 3:
                            w = _advance(w, c, stride, stride, 0, 0, c["cstep"])
                    st, m = _advance(st, c, stride, c["dws64"], c["dhs64"], c["dn64"], c["c64"]), m + 64


# ---------------------------------------------------------------------------------------------------------------
def _rot_index_s2(ci, tap, co, Cin, Cout):                        # conv_prep_rot_index, rot_mode 1
    r, s = tap // 3, tap % 3
    a, b, dr, ds = int(r != 1), int(s != 1), int(r == 0), int(s == 0)
    taps, t = (1 + a) * (1 + b), dr * (1 + b) + ds
    base = (5 if b else 3) if a else (1 if b else 0)
    return base * Cin * Cout + (ci * taps + t) * Cout + co


@pytest.mark.parametrize("shape", [(2, 5, 8, 12, 7), (1, 3, 2, 2, 4), (2, 4, 6, 6, 3)])
def test_stride2_dgrad_parity_classes_and_weight_packing(shape):
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64)
    y = F.conv2d(x, wt, None, 2, 1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    ho, wo = h // 2, w // 2
    # class-packed weights exactly as the prep kernel places them
    packed = np.zeros(cin * 9 * cout)
    wn = wt.numpy()
    for co in range(cout):
        for ci in range(cin):
            for tap in range(9):
                packed[_rot_index_s2(ci, tap, co, cin, cout)] = wn[co, ci, tap // 3, tap % 3]
    dyn = np.zeros((n, ho + 1, wo + 1, cout))                     # zero beyond the bottom / right edge
    dyn[:, :ho, :wo] = dy.permute(0, 2, 3, 1).numpy()
    dx = np.zeros((n, h, w, cin))
    tap_base = [0, 1, 3, 5]
    for a in (0, 1):
        for b in (0, 1):
            taps = (1 + a) * (1 + b)
            wc = packed[tap_base[a * 2 + b] * cin * cout:][:cin * taps * cout].reshape(cin, 1 + a, 1 + b, cout)
            acc = np.zeros((n, ho, wo, cin))
            for dr in range(1 + a):
                for ds in range(1 + b):
                    acc += np.einsum("nhwo,io->nhwi", dyn[:, dr:dr + ho, ds:ds + wo], wc[:, dr, ds])
            dx[:, a::2, b::2] = acc                                # written exactly once
    np.testing.assert_allclose(dx, x.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-12, atol=1e-12)


def test_compact_stride2_1x1_gradient_added_at_even_pixels():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 6, 8, 10, generator=g, dtype=torch.float64, requires_grad=True)
    w1 = torch.randn(4, 6, 1, 1, generator=g, dtype=torch.float64)
    wd = torch.randn(5, 6, 1, 1, generator=g, dtype=torch.float64)
    y1, yd = F.conv2d(x, w1), F.conv2d(x, wd, None, 2)
    dy1 = torch.randn(y1.shape, generator=g, dtype=torch.float64)
    dyd = torch.randn(yd.shape, generator=g, dtype=torch.float64)
    (y1 * dy1).sum().backward(retain_graph=True)
    (yd * dyd).sum().backward()
    compact = torch.einsum("nohw,oi->nihw", dyd, wd[:, :, 0, 0])  # plain 1x1 GEMM on dY of the strided conv
    dx = torch.einsum("nohw,oi->nihw", dy1, w1[:, :, 0, 0])
    dx[:, :, ::2, ::2] += compact
    torch.testing.assert_close(dx, x.grad, rtol=1e-12, atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------
ROWE, COL0 = 800, 16                                              # staged row image: column 0 at element 16


def _stem_rows(x_nhwc, n, ho):
    H, W = x_nhwc.shape[1], x_nhwc.shape[2]
    img = np.zeros((7, ROWE))
    for r in range(7):
        hi = 2 * ho - 3 + r
        if 0 <= hi < H:
            img[r, COL0:COL0 + 3 * W] = x_nhwc[n, hi].reshape(-1)
    return img


@pytest.mark.parametrize("shape", [(2, 16, 24), (1, 9, 8)])
def test_stem_row_windows_and_packed_weights(shape):
    n, H, W = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 3, H, W, generator=g, dtype=torch.float64)
    wt = torch.randn(64, 3, 7, 7, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, wt, None, 2, 3)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    Ho, Wo = y.shape[2], y.shape[3]
    xn = x.permute(0, 2, 3, 1).numpy()
    wn = wt.detach().permute(0, 2, 3, 1).numpy()                  # [co][r][s][c]
    wp = np.zeros((64, 176))                                      # stem_prep_weights_kernel
    for r in range(7):
        for s in range(7):
            for c in range(3):
                wp[:, r * 24 + (s + 1) * 3 + c] = wn[:, r, s, c]
    yk = np.zeros((n, Ho, Wo, 64))
    D = np.zeros((168, 64))                                       # weight-gradient tile [(r, t)][co]
    dyn = dy.permute(0, 2, 3, 1).numpy()
    for b in range(n):
        for ho in range(Ho):
            img = _stem_rows(xn, b, ho)
            for wo in range(Wo):
                win = np.concatenate([img[r, 4 + 6 * wo:4 + 6 * wo + 24] for r in range(7)])   # 168 window elements
                yk[b, ho, wo] = wp[:, :168] @ win
                D += np.outer(win, dyn[b, ho, wo])
    np.testing.assert_allclose(yk, y.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-12, atol=1e-12)
    dw = np.zeros((64, 7, 7, 3))                                  # stem_wgrad_reduce_kernel's mapping
    for r in range(7):
        for sc in range(21):
            dw[:, r, sc // 3, sc % 3] = D[r * 24 + 3 + sc]
    np.testing.assert_allclose(dw, wt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-11, atol=1e-11)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [7, 13, 14, 28, 56, 112, 127, 224, 1000])
def test_float_reciprocal_division_is_exact_below_2_to_24(d):
    """The kernels decode m -> (n, ho, wo) with a float32 multiply by 1/d, a truncation and ONE correction step
    (csrc/dir_conv.hip, dir_conv_wgrad.hip: 'exact for m < 2^24', enforced by the C-ABI's M < 2^24 check). Checked here at
    every multiple of d and its two neighbours, where a wrong rounding would show, plus random m."""
    def decode(m):
        inv = np.float32(1.0) / np.float32(d)
        q = (m.astype(np.float32) * inv).astype(np.int64)
        r = m - q * d
        lo = r < 0
        q, r = np.where(lo, q - 1, q), np.where(lo, r + d, r)
        hi = r >= d
        return np.where(hi, q + 1, q), np.where(hi, r - d, r)
    k = np.arange(0, (1 << 24) // d + 1, dtype=np.int64)
    ms = [k * d + delta for delta in (-1, 0, 1)] + [np.random.default_rng(d).integers(0, 1 << 24, 100000)]
    for m in ms:
        m = m[(m >= 0) & (m < (1 << 24))]
        q, r = decode(m)
        assert np.array_equal(q, m // d) and np.array_equal(r, m % d)


# ---------------------------------------------------------------------------------------------------------------
# All-taps 3x3 weight gradient (csrc/dir_conv_wgrad3.hip), restated lane by lane for one workgroup and one chunk
W3_GEOM = {56: dict(RB=2, P=60), 28: dict(RB=4, P=32), 14: dict(RB=7, P=16), 7: dict(RB=7, P=12)}


def _tr_read(lds, addr):
    """ds_read_b64_tr_b16 of one wavefront: addr[64] byte addresses -> [64][4] 16-bit values. Within each 16-lane group lane i
    supplies the address of row (i >> 2), columns 4 (i & 3) .. + 3 and receives column i, rows 0 .. 3."""
    out = np.zeros((64, 4), dtype=lds.dtype)
    for l in range(64):
        g, i = l >> 4, l & 15
        for j in range(4):
            src = 16 * g + 4 * j + (i >> 2)
            out[l, j] = lds[addr[src] // 2 + (i & 3)]
    return out


@pytest.mark.parametrize("wi", [56, 28, 14, 7])
def test_all_taps_wgrad_lds_image_and_transposing_reads(wi):
    g = W3_GEOM[wi]
    RB, P, H = g["RB"], g["P"], wi
    KPIX = RB * wi
    NK = (KPIX + 15) // 16
    DY_PIECES = NK * 2
    XP = (RB + 2) * P
    X_PIECES = (XP + 7) // 8
    PIECES = DY_PIECES + X_PIECES
    DY_BYTES = DY_PIECES * 1024
    rng = np.random.default_rng(wi)
    C = 64
    n_img = 2
    # small integers: every product and sum below is exact in float32, so the emulation can be compared with == 
    x = rng.integers(-3, 4, (n_img, H, wi, C)).astype(np.float32)
    dy = rng.integers(-3, 4, (n_img, H, wi, C)).astype(np.float32)
    acc = np.zeros((9, 64, 64), np.float64)                                  # [tap][co][ci] for the (co block 0, ci block 0) workgroup
    for n in range(n_img):
        for h0 in range(0, H, RB):
            lds = np.zeros(PIECES * 512, np.float32)                         # one stage as 16-bit elements (values stored as floats)
            # ---- DMA: piece q, lane -> LDS element q*512 + lane*8 .. + 7; source chunk swizzled on bit 1 of the row
            for q in range(PIECES):
                for lane in range(64):
                    lrow, pc8 = lane >> 3, lane & 7
                    lchunk = pc8 ^ (((lane >> 4) & 1) << 2)
                    vals = np.zeros(8, np.float32)
                    if q < DY_PIECES:
                        k = q * 8 + lrow
                        if k < KPIX:
                            vals = dy[n, h0 + k // wi, k % wi, lchunk * 8:lchunk * 8 + 8]
                    else:
                        slot = (q - DY_PIECES) * 8 + lrow
                        pr, pc = slot // P, slot % P
                        hi = h0 - 1 + pr
                        if slot < XP and 1 <= pc <= wi and 0 <= hi < H:
                            vals = x[n, hi, pc - 1, lchunk * 8:lchunk * 8 + 8]
                    lds[q * 512 + lane * 8:q * 512 + lane * 8 + 8] = vals
            # ---- every wavefront (co half wm, ci half wn, 16-pixel steps of parity kpar)
            for wave in range(8):
                kpar, wm, wn = wave >> 2, (wave >> 1) & 1, wave & 1
                lane = np.arange(64)
                hf, r4 = lane >> 5, (lane >> 2) & 3
                lane_c = ((lane >> 4) & 1) * 32 + (lane & 3) * 8
                dyl = (8 * hf + r4) * 128 + ((wm ^ (r4 >> 1)) << 6) + lane_c + kpar * 2048
                xl = DY_BYTES + lane_c
                for tk in range((NK + 1) // 2):
                    kk = kpar + 2 * tk
                    if kk >= NK:
                        continue
                    a = np.concatenate([_tr_read(lds, dyl + tk * 4096), _tr_read(lds, dyl + tk * 4096 + 512)], 1)   # [lane][8 k]
                    pp0 = []
                    for j in range(2):
                        k = 16 * kk + 8 * hf + 4 * j + r4
                        pp0.append(np.where(k < KPIX, (k // wi) * P + k % wi, 0))
                    for s_ in range(3):
                        q0, q1 = pp0[0] + s_, pp0[1] + s_
                        x0 = xl + (q0 << 7) + ((((q0 >> 1) & 1) ^ wn) << 6)
                        x1 = xl + (q1 << 7) + ((((q1 >> 1) & 1) ^ wn) << 6)
                        for r_ in range(3):
                            b = np.concatenate([_tr_read(lds, x0 + r_ * P * 128), _tr_read(lds, x1 + r_ * P * 128)], 1)
                            # v_mfma_f32_32x32x16: lane l supplies row / column (l & 31), k = 8 (l >> 5) + e
                            A = np.zeros((32, 16)); B = np.zeros((32, 16))
                            for l in range(64):
                                A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a[l]
                                B[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = b[l]
                            acc[r_ * 3 + s_, wm * 32:wm * 32 + 32, wn * 32:wn * 32 + 32] += A @ B.T
    ref = torch.nn.grad.conv2d_weight(torch.as_tensor(x).permute(0, 3, 1, 2).double(), (C, C, 3, 3),
                                      torch.as_tensor(dy).permute(0, 3, 1, 2).double(), padding=1).numpy()      # [co][ci][r][s]
    got = acc.reshape(3, 3, 64, 64).transpose(2, 3, 0, 1)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("wi,P", [(56, 64), (28, 32), (14, 16)])
def test_patch_kernel_tap_rows_are_immediates_under_the_swizzle(wi, P):
    """conv3x3_patch_kernel: physical 16-B chunk of logical chunk c in patch row pp is c ^ ((pp >> 1) & 7). A tap (r, s) reads row
    pp0 + s + r * P: with P a multiple of 16 the swizzle term depends on pp0 + s only, so r * P * 128 is an address immediate; and
    row * 128 + (((kk * 2 + fhalf) ^ z) << 4) == (row * 128 + ((fhalf ^ z) << 4)) ^ (kk << 5) — one base register per fragment."""
    assert P % 16 == 0 and P >= wi + 2
    for pp0 in range(0, 4 * P):
        for s_ in range(3):
            z = ((pp0 + s_) >> 1) & 7
            for r_ in range(3):
                assert (((pp0 + s_ + r_ * P) >> 1) & 7) == z
    for row in range(256):
        z = (row >> 1) & 7
        for fhalf in range(2):
            base = row * 128 + ((fhalf ^ z) << 4)
            for kk in range(4):
                assert row * 128 + (((kk * 2 + fhalf) ^ z) << 4) == base ^ (kk << 5)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layers", [[(64 * 64, 7), (512 * 9 * 512 // 64, 16)], [(4096, 1), (64 * 3 * 64, 33), (2048 * 64, 2)]])
def test_batched_splitk_reduction_adds_in_the_per_layer_kernels_order(layers):
    """conv_wgrad_reduce_kernel: block = 16 float4 columns x 16 split lanes; lane sl adds splits sl, sl + 16, ... in order, the lanes are then
    added in lane order. conv_wgrad_reduce_batched_kernel walks the same column groups with a grid-stride loop (gridDim.x = 512), one
    blockIdx.y per layer. Emulated here in float32: every element once, identical sums."""
    RD_COLS, RD_LANES, GRID_X = 16, 16, 512
    rng = np.random.default_rng(0)

    def column(part, splits, i):                                          # one float4 element's sum, as both kernels form it
        lanes = []
        for sl in range(RD_LANES):
            s = np.float32(0.0)
            for k in range(sl, splits, RD_LANES):
                s = np.float32(s + part[k, i])
            lanes.append(s)
        t = lanes[0]
        for k in range(1, RD_LANES):
            t = np.float32(t + lanes[k])
        return t
    for n, splits in layers:
        assert n % 4 == 0
        part = rng.normal(0, 1, (splits, n)).astype(np.float32)
        # per-layer kernel: grid = ceil(n / 4 / 16) blocks, block b owns float4 columns 16 b .. 16 b + 15
        visited_single = np.zeros(n // 4, np.int32)
        for b in range((n // 4 + RD_COLS - 1) // RD_COLS):
            for col in range(RD_COLS):
                i4 = b * RD_COLS + col
                if i4 * 4 < n:
                    visited_single[i4] += 1
        # batched kernel: the same groups, strided over GRID_X blocks
        visited_batched = np.zeros(n // 4, np.int32)
        groups = (n // 4 + RD_COLS - 1) // RD_COLS
        for bx in range(min(GRID_X, groups + 3)):
            gi = bx
            while gi < groups:
                for col in range(RD_COLS):
                    i4 = gi * RD_COLS + col
                    if i4 * 4 < n:
                        visited_batched[i4] += 1
                gi += GRID_X
        assert np.all(visited_single == 1) and np.all(visited_batched == 1)
        # the order of additions per element does not depend on which kernel visits it: spot-check the arithmetic against a float64 sum
        for i in rng.integers(0, n, 8):
            v = column(part, splits, int(i))
            assert abs(float(v) - float(part[:, i].astype(np.float64).sum())) <= 1e-5 * max(1.0, float(np.abs(part[:, i]).sum()))
