// Exact-float32 convolutions for the PARITY MODE of the ResNet-50 stack (amp_dtype=None): forward, data gradient and weight
// gradient of any nn.Conv2d of imdb-wiki-dir/resnet.py:44-49,79,112-116 (7x7/2 stem, 1x1, 3x3, stride 1 or 2) on NHWC float32
// activations and [Cout][R][S][Cin] float32 weights, as implicit GEMMs on v_mfma_f32_32x32x2_f32.
//
// Why it exists: north_star's "training loss within 1e-5 relative" has to be shown on hand-written kernels, not on a library
// convolution. The f32-input MFMA is bit-for-bit a k-ordered fmaf chain (one rounding per product, fp32 accumulate,
// MI355X_MICROARCH.md "Matrix cores"), i.e. the same arithmetic class as the reference's fp32 cuDNN / CPU convolutions, at the
// f32 vector rate (157 TFLOP/s peak = 1/16 of bf16 MFMA). The product path stays the bf16 kernels of dir_conv.hip; this file
// is what `resnet50` runs on when it is asked for float32, so that the SAME autograd graph (fused BatchNorm nodes, joins,
// FDS / loss tail) can be compared with the reference at float32 accuracy.
//
// One tile shape for all three GEMMs: workgroup 256 threads = 4 wavefronts, output tile 64 x 64 (each wavefront one 32 x 32
// MFMA tile), K-step 16 (8 MFMAs per wavefront per step). Operands are gathered element-wise (4-byte loads, index decode per
// K-step) into k-major LDS tiles [16][68] so that fragment reads are 32 consecutive floats per half-wave (conflict free);
// the next K-step's 8 elements per thread are fetched into registers while the current one is multiplied.
//   forward : M = N*Ho*Wo, Ncol = Cout,     K = R*S*Cin    A gathers x (zero outside the image), B = w rows
//   dgrad   : M = N*H*W,   Ncol = Cin,      K = R*S*Cout   A gathers dy at (hi + pad - r) / stride when divisible, B = w^T
//   wgrad   : M = Cout,    Ncol = R*S*Cin,  K = N*Ho*Wo    A = dy^T, B gathers x; split-K over blockIdx.z into float32
//             partial tiles, summed in a fixed order by a second kernel (deterministic, no atomics)
#include "dir_common.h"
#include "dir_conv_shared.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float cf_f32x16;
constexpr int CF_BM = 64, CF_BN = 64, CF_BK = 16, CF_LD = 68;

struct ConvF32P {
    const float* a;            // forward: x,  dgrad: dy,  wgrad: dy
    const float* b;            // forward: w,  dgrad: w,   wgrad: x
    float* out;                // forward: y,  dgrad: dx,  wgrad: partial tiles [splits][M][Ncol] (or dw when splits == 1)
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
    int M, Ncol, K;
    int klen;                  // K range per blockIdx.z (multiple of CF_BK)
    // fused store epilogue of the data gradient (same semantics as dir_conv_fwd_fused / dir_conv_dgrad_join of the bf16 path):
    const float* addend;       // [M][Ncol] added to the result (gradient accumulation of a fan-out)
    const float* addend2;      // COMPACT [N][H/2][W/2][Ncol], added at the even (h, w) pixels only (stride-2 1x1 sibling)
    const float* mask;         // [M][Ncol]: result zeroed where !(mask > 0) (ReLU backward of the tensor the gradient belongs to)
    float* stats;              // forward tile kernel only: per 64-row slab of the output and channel, (sum, sum of squares) [slabs][2][Ncol] — the
                               // statistics partials the following BatchNorm consumes (dir_bn_fwd_train_partials): no statistics pass over y
};

// 8 MFMA 32x32x2 on one staged K-step. A fragment: lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31].
__device__ __forceinline__ void cf_mfma_step(const float* __restrict__ As, const float* __restrict__ Bs, int wm, int wn, int lane,
                                             cf_f32x16& acc) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int kk = 0; kk < CF_BK / 2; ++kk) {
        const float a = As[(2 * kk + half) * CF_LD + wm * 32 + col];
        const float b = Bs[(2 * kk + half) * CF_LD + wn * 32 + col];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
__device__ __forceinline__ void cf_store_tile(const cf_f32x16& acc, float* __restrict__ out, int m0, int n0, int wm, int wn, int lane,
                                              int M, int Ncol) {
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= Ncol) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M) out[(size_t)row * Ncol + col] = acc[e];
    }
}

// Data-gradient store with the fused epilogue: rows are pixels (n, h, w) of dx [N][H][W][Ncol].
__device__ __forceinline__ void cf_store_tile_fused(const cf_f32x16& acc, const ConvF32P& p, int m0, int n0, int wm, int wn, int lane) {
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= p.Ncol) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row >= p.M) continue;
        const size_t o = (size_t)row * p.Ncol + col;
        float v = acc[e];
        if (p.addend) v += p.addend[o];
        if (p.addend2) {
            const int w = row % p.W, q = row / p.W, h = q % p.H, n = q / p.H;
            if (!((h | w) & 1)) v += p.addend2[(((size_t)n * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.Ncol + col];
        }
        if (p.mask && !(p.mask[o] > 0.0f)) v = 0.0f;
        p.out[o] = v;
    }
}

// MODE 0 = forward, 1 = data gradient. Loader: thread t fetches k = k0 + (t & 15) of rows / columns (t >> 4) + 16 i.
template <int MODE>
__global__ void __launch_bounds__(DIR_TPB) conv_f32_kfast_kernel(ConvF32P p) {
    __shared__ float As[CF_BK * CF_LD];
    __shared__ float Bs[CF_BK * CF_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * CF_BM, n0 = blockIdx.y * CF_BN;
    const int kq = t & 15, rq = t >> 4;
    // rows of the A operand owned by this thread: (image, y, x) of the output pixel (forward) / input pixel (dgrad)
    int r_n[4], r_y[4], r_x[4];
    bool r_ok[4];
    const int PH = MODE == 0 ? p.Ho : p.H, PW = MODE == 0 ? p.Wo : p.W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + rq + 16 * i;
        r_ok[i] = m < p.M;
        const int mm = r_ok[i] ? m : 0;
        r_x[i] = mm % PW; const int q = mm / PW; r_y[i] = q % PH; r_n[i] = q / PH;
    }
    const int CK = MODE == 0 ? p.Cin : p.Cout;                  // channel extent of the K axis (innermost of k)
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
        const int k = k0 + kq;
        const bool kok = k < p.K;
        const int kk = kok ? k : 0;
        const int c = kk % CK, tap = kk / CK, r = tap / p.S, s = tap - r * p.S;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = 0.0f;
            if (kok && r_ok[i]) {
                if (MODE == 0) {
                    const int hi = r_y[i] * p.stride - p.pad + r, wi = r_x[i] * p.stride - p.pad + s;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        v = p.a[(((size_t)r_n[i] * p.H + hi) * p.W + wi) * p.Cin + c];
                } else {
                    const int th = r_y[i] + p.pad - r, tw = r_x[i] + p.pad - s;
                    if (th >= 0 && tw >= 0) {
                        const int ho = th / p.stride, wo = tw / p.stride;
                        if (ho * p.stride == th && wo * p.stride == tw && ho < p.Ho && wo < p.Wo)
                            v = p.a[(((size_t)r_n[i] * p.Ho + ho) * p.Wo + wo) * p.Cout + c];
                    }
                }
            }
            ra[i] = v;
            const int n = n0 + rq + 16 * i;
            float w = 0.0f;
            if (kok && n < p.Ncol) {
                if (MODE == 0) w = p.b[(size_t)n * p.K + kk];                                        // w[co = n][r][s][ci]
                else w = p.b[(((size_t)c * p.R + r) * p.S + s) * p.Cin + n];                         // w[co = c][r][s][ci = n]
            }
            rb[i] = w;
        }
    };
    cf_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    fetch(0);
    for (int k0 = 0; k0 < p.K; k0 += CF_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[kq * CF_LD + rq + 16 * i] = ra[i]; Bs[kq * CF_LD + rq + 16 * i] = rb[i]; }
        __syncthreads();
        if (k0 + CF_BK < p.K) fetch(k0 + CF_BK);
        cf_mfma_step(As, Bs, wm, wn, lane, acc);
        __syncthreads();
    }
    if (MODE == 1 && (p.addend || p.addend2 || p.mask)) cf_store_tile_fused(acc, p, m0, n0, wm, wn, lane);
    else cf_store_tile(acc, p.out, m0, n0, wm, wn, lane, p.M, p.Ncol);
}

// Weight gradient. Loader: thread t fetches column c = t & 63 of the A tile (output channel m0 + c) and of the B tile
// (weight element n0 + c = (r, s, ci)) for the four pixels k = k0 + (t >> 6) + 4 i: both reads are contiguous along c.
__global__ void __launch_bounds__(DIR_TPB) conv_f32_wgrad_kernel(ConvF32P p) {
    __shared__ float As[CF_BK * CF_LD];
    __shared__ float Bs[CF_BK * CF_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * CF_BM, n0 = blockIdx.y * CF_BN;
    const int c = t & 63, kq = t >> 6;
    const int kbeg = blockIdx.z * p.klen;
    const int kend = (kbeg + p.klen < p.K) ? kbeg + p.klen : p.K;
    const int co = m0 + c;
    const bool co_ok = co < p.M;
    const int n = n0 + c;
    const bool n_ok = n < p.Ncol;
    const int nn = n_ok ? n : 0;
    const int ci = nn % p.Cin, tap = nn / p.Cin, r = tap / p.S, s = tap - r * p.S;
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kq + 4 * i;
            float a = 0.0f, b = 0.0f;
            if (k < kend) {
                if (co_ok) a = p.a[(size_t)k * p.Cout + co];
                if (n_ok) {
                    const int wo = k % p.Wo, q = k / p.Wo, ho = q % p.Ho, img = q / p.Ho;
                    const int hi = ho * p.stride - p.pad + r, wi = wo * p.stride - p.pad + s;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        b = p.b[(((size_t)img * p.H + hi) * p.W + wi) * p.Cin + ci];
                }
            }
            ra[i] = a; rb[i] = b;
        }
    };
    cf_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += CF_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[(kq + 4 * i) * CF_LD + c] = ra[i]; Bs[(kq + 4 * i) * CF_LD + c] = rb[i]; }
        __syncthreads();
        if (k0 + CF_BK < kend) fetch(k0 + CF_BK);
        cf_mfma_step(As, Bs, wm, wn, lane, acc);
        __syncthreads();
    }
    cf_store_tile(acc, p.out + (size_t)blockIdx.z * p.M * p.Ncol, m0, n0, wm, wn, lane, p.M, p.Ncol);
}

__global__ void __launch_bounds__(DIR_TPB) conv_f32_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                        size_t n, int splits) {
    for (size_t i = (size_t)blockIdx.x * DIR_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * DIR_TPB) {
        float s = part[i];
        for (int z = 1; z < splits; ++z) s += part[(size_t)z * n + i];          // fixed order: bit-reproducible
        dw[i] = s;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Tile kernels (round 6). The gather kernels above run at a quarter of the float32 MFMA rate (4-byte gathers with an index decode per
// element, one 32 x 32 tile per wavefront, one LDS stage): 160 ms per B = 256 training step, slower than the vendor library's float32
// step. Same arithmetic — v_mfma_f32_32x32x2_f32 over k in the same order (k = (r, s, c), one K-step = 16 channels of one filter tap),
// so forward and data gradient are BIT-IDENTICAL to the gather kernels — on a 128 x 128 (x 64, 64 x 128) workgroup tile, 64 x 64 per
// wavefront, operands staged global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, out-of-image / out-of-range lanes
// read beyond the buffer: the hardware writes zeros), two LDS stages, one barrier per K-step. A K-step is 16 KB of operands per 2048
// MFMA cycles of a wavefront (8 B/clk/CU): the loop is bound by the matrix pipe, not by operand delivery like the bf16 kernels'.
// An operand tile is staged in the order memory has it:
//   K-contiguous (x / dy rows of forward and data gradient, forward weights [Cout][K]): rows of 64 B = 16 k, the four 16-B chunks of a
//     row XOR-swizzled with (row >> 2) & 3 on the DMA's source side; a lane's fragment read is ONE ds_read_b128 per two MFMAs (its row,
//     four consecutive k; the half-wave picks k = 2 j + (lane >> 5)): conflict-free for every 16-lane group of the instruction;
//   k-major (data-gradient weights w[co][r][s][ci] read as B[k = (r, s, co)][n = ci], both weight-gradient operands dy[pixel][co],
//     x[pixel + tap][ci]): 16 k-rows of T floats, fragment reads are 32 consecutive floats per half-wave (ds_read_b32).
// Needs the K axis in whole 16-channel steps (Cin resp. Cout % 16 == 0) and 16-byte columns; the 7x7 stem (Cin = 3) and odd shapes
// stay on the gather kernels.
constexpr int FT_BK = 16, FT_ROWB = FT_BK * 4;
constexpr int FT_OOB = (int)0x80000000;
constexpr int FT_SPLIT_STAGES = 2;      // LDS stages of the split-bf16 arithmetics. The loop below is a ring with counted s_waitcnt (FT_SPLIT_STAGES - 1 K-steps in flight
                                        // across the barrier); measured at 2 / 3 / 4 stages: 33.8 / 35.9 / 38.3 ms over the step's layers (x2) - the loop is not bound by
                                        // the DMA's latency, so two stages = the most workgroups per CU (profiles/r06_f32_split_arith.txt)
constexpr int FT_ABLATE = 0;            // measurement builds only (tools/ablate_f32.py patches this line): 1 = no DMA, 2 = no epilogue stores, 3 = no MFMA, 4 = no barriers / DMA waits, 5 = (split arithmetics) no split VALU work
enum { FT_FWD = 0, FT_DGRAD = 1, FT_WGRAD = 2 };

struct ConvF32T {
    ConvF32P c;
    int ntn, nblocks;          // N tiles, workgroups per K split
    int kbytes_a, kbytes_b;    // buffer extents in bytes
    float inv_wo, inv_ho;      // reciprocals for the pixel decode of the weight gradient's K axis
    int cls;                   // stride-2 data gradient by PARITY CLASS (blockIdx.y = 2 a + b): the rows of a tile are the pixels (2 i + a, 2 j + b) of
                               // one class, c.M = pixels per class; only the filter taps r = (a + pad) mod 2, + 2, ... reach them, so the K loop
                               // visits those taps only (the element-gather kernel multiplies the other 3/4 of its products by zero)
};


// ARITH (round 6): 0 = v_mfma_f32_32x32x2_f32 on the float32 operands (exact: the parity arithmetic); 3 / 2 = SPLIT-bf16 arithmetic on the bf16 matrix
// pipe: every float32 operand element is split in registers into three (two) bf16 terms hi + mid (+ lo) — each split exact: the residual of a
// round-to-nearest bf16 is representable in float32 — and a K-step of 16 becomes ONE v_mfma_f32_32x32x16_bf16 per term pair kept: hi*hi, hi*mid,
// mid*hi, hi*lo, lo*hi, mid*mid (ARITH 3: the dropped pairs are below 2^-24 of the product: float32-GRADE results, not bit-equal ones) or hi*hi,
// hi*lo, lo*hi (ARITH 2, terms hi + lo: 16 significand bits). 6 (3) x 32 matrix clocks per 32 x 32 x 16 block instead of 8 x 64; the price is
// ~44 (24) VALU operations per 8-element fragment for the splits. Same staging, same epilogue, same statistics rows as the exact kernel.
typedef __attribute__((ext_vector_type(8))) float cf_f32x8;
typedef __attribute__((ext_vector_type(4))) float cf_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 cf_bf16x8;
// Cost of the split per element (measured forms, profiles/r06_f32_split_arith.txt): a v_cvt_pk_bf16_f32 rounds two elements to nearest, the residual needs
// the rounded value back as float32 (shift / mask of the packed pair) and one exact subtraction. THREE terms: the two leading terms are TRUNCATED instead
// (upper 16 bits: one v_perm_b32 packs a pair, v_and + v_sub leave the exact residual) and only the last term is rounded — the error of the sum of the
// terms is that of the last rounding, 2^-24 of the element, unbiased; 0.5 instead of 1.5 conversions per element. TWO terms: both rounded to nearest
// (truncating the leading term doubles the error, 2^-16 instead of 2^-17, for 3 % of the time).
constexpr int FT_X2_WAVES = 4;         // wavefronts per SIMD the two-term kernels are compiled for (their natural allocation is 130-147 registers = 3)
constexpr int FT_WAVES(int arith, int mode) { return arith == 2 ? (mode == 2 ? 3 : FT_X2_WAVES) : (arith == 3 ? 3 : 4); }   // (x2 weight gradient, k-major operands: 4 would spill)
constexpr bool FT_SPLIT_RNE_HI(int nt) { return nt == 2; }
template <int NT>
__device__ __forceinline__ void ft_split(cf_f32x8 v, cf_bf16x8 (&t)[NT]) {
    typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#pragma unroll
    for (int s = 0; s + 1 < NT; ++s) {
        u32x4 pk;
        u32x8 hb;
        if (FT_SPLIT_RNE_HI(NT)) {
            pk = __builtin_bit_cast(u32x4, __builtin_convertvector(v, cf_bf16x8));
#pragma unroll
            for (int j = 0; j < 4; ++j) { hb[2 * j] = pk[j] << 16; hb[2 * j + 1] = pk[j] & 0xffff0000u; }
        } else {
            const u32x8 b = __builtin_bit_cast(u32x8, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) pk[j] = __builtin_amdgcn_perm(b[2 * j + 1], b[2 * j], 0x07060302u);
            hb = b & 0xffff0000u;
        }
        t[s] = __builtin_bit_cast(cf_bf16x8, pk);
        v -= __builtin_bit_cast(cf_f32x8, hb);                               // exact
    }
    t[NT - 1] = __builtin_convertvector(v, cf_bf16x8);                       // v_cvt_pk_bf16_f32: round to nearest even
}

template <int TM, int TN, int MODE, int ARITH = 0>
__global__ void __launch_bounds__(DIR_TPB) __attribute__((amdgpu_waves_per_eu(FT_WAVES(ARITH, MODE)))) conv_f32_tile_kernel(ConvF32T q) {
    const ConvF32P& p = q.c;
    constexpr bool AKM = MODE == FT_WGRAD, BKM = MODE != FT_FWD;
    constexpr int WGM = (TM == 128 && TN == 64) ? 4 : 2, WGN = 4 / WGM;
    constexpr int WM = TM / WGM, WN = TN / WGN, MI = WM / 32, NI = WN / 32;
    constexpr int NA = TM / 64, NB = TN / 64;                       // 1 KB DMA pieces per wavefront and K-step
    constexpr int A_BYTES = TM * FT_ROWB, STAGE = (TM + TN) * FT_ROWB;
    constexpr int NS = ARITH ? FT_SPLIT_STAGES : 2;                 // LDS stages: two for the exact arithmetic (bound by the matrix pipe), a ring for the split ones
    constexpr int LDS_BYTES = (NS * STAGE > 64 * TN * 4) ? NS * STAGE : 64 * TN * 4;   // the K-loop stages; the epilogue's 64-row slab
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    int lin;
    {
        const int b = blockIdx.x, n8 = q.nblocks / 8, r8 = q.nblocks % 8, xcd = b % 8, i = b / 8;
        lin = (xcd < r8 ? xcd * (n8 + 1) : r8 * (n8 + 1) + (xcd - r8) * n8) + i;
    }
    const int mt = lin / q.ntn, nt = lin - mt * q.ntn;
    const int m0 = mt * TM, n0 = nt * TN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int fi = lane & 31, fh = lane >> 5;
    const int cls_a = (MODE == FT_DGRAD && q.cls) ? (int)(blockIdx.y >> 1) : 0, cls_b = (MODE == FT_DGRAD && q.cls) ? (int)(blockIdx.y & 1) : 0;
    const int kbeg = MODE == FT_WGRAD ? (int)blockIdx.y * p.klen : 0;
    const int kend = MODE == FT_WGRAD ? ((kbeg + p.klen < p.K) ? kbeg + p.klen : p.K) : p.K;
    // The LDS-DMA is inline assembly (cp_dma16) with an explicit s_waitcnt before each barrier: with the compiler-visible builtin and a
    // run-time stage index the compiler drains the pending DMA (vmcnt(0)) in front of the first fragment read of every K-step — it cannot
    // tell the two stages apart — which exposes the whole L2 round trip per step (measured: 100 instead of 125 TFLOP/s).
    const cp_u32x4 rs_a = cp_rsrc(p.a, (uint32_t)q.kbytes_a);
    const cp_u32x4 rs_b = cp_rsrc(p.b, (uint32_t)q.kbytes_b);
    typedef __attribute__((address_space(3))) unsigned char* ft_lds_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(ft_lds_t)smem;

    // ---- loader state. K-contiguous tiles: piece pa = rows 16 pa .. 16 pa + 15, lane -> row (lane >> 2), physical chunk lane & 3
    // = logical chunk (lane & 3) ^ (lane >> 4). k-major tiles: piece = floats 256 pa .. of the [16][T] image, lane -> 4 consecutive columns.
    const int rlc = ((lane & 3) ^ (lane >> 4)) * 4;                 // first k (float index inside the K-step) of this lane's chunk
    int a_img[NA], a_y[NA], a_x[NA];                                // forward / data gradient: the pixel of the lane's A row (a_img < 0: no row)
    int a_kr[NA], a_col[NA];                                        // weight gradient: k-row and column of the lane's chunk
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int pa = wave + 4 * i;
        if (!AKM) {
            const int m = m0 + 16 * pa + (lane >> 2);
            const bool cl = MODE == FT_DGRAD && q.cls;
            const int PH = MODE == FT_FWD ? p.Ho : (cl ? p.H >> 1 : p.H), PW = MODE == FT_FWD ? p.Wo : (cl ? p.W >> 1 : p.W);
            a_img[i] = -1; a_y[i] = a_x[i] = 0;
            if (m < p.M) {
                a_x[i] = m % PW; const int qq = m / PW; a_y[i] = qq % PH; a_img[i] = qq / PH;
                if (cl) { a_y[i] = 2 * a_y[i] + cls_a; a_x[i] = 2 * a_x[i] + cls_b; }
            }
            a_kr[i] = a_col[i] = 0;
        } else {
            const int f = 256 * pa + 4 * lane;
            a_kr[i] = f / TM; a_col[i] = m0 + f % TM;
            a_img[i] = a_y[i] = a_x[i] = 0;
        }
    }
    int b_row[NB], b_kr[NB], b_col[NB], b_r[NB], b_s[NB], b_ci[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pb = wave + 4 * i;
        b_row[i] = b_kr[i] = b_col[i] = b_r[i] = b_s[i] = b_ci[i] = 0;
        if (!BKM) {
            const int n = n0 + 16 * pb + (lane >> 2);
            b_row[i] = n < p.Ncol ? n : -1;
        } else {
            const int f = 256 * pb + 4 * lane;
            b_kr[i] = f / TN; b_col[i] = n0 + f % TN;
            if (MODE == FT_WGRAD) {
                const int nn = b_col[i] < p.Ncol ? b_col[i] : 0;
                const int tap = nn / p.Cin;
                b_ci[i] = nn - tap * p.Cin; b_r[i] = tap / p.S; b_s[i] = tap - b_r[i] * p.S;
            }
        }
    }
    const int CK = MODE == FT_FWD ? p.Cin : p.Cout;                 // channel extent of the K axis of forward / data gradient
    // cursor of the next K-step to issue: channel offset and filter tap (parity-class data gradient: first tap and tap step per axis)
    const int tstep = (MODE == FT_DGRAD && q.cls) ? 2 : 1;
    const int r0 = (MODE == FT_DGRAD && q.cls) ? ((cls_a + p.pad) & 1) : 0, s0 = (MODE == FT_DGRAD && q.cls) ? ((cls_b + p.pad) & 1) : 0;
    int ld_c = 0, ld_r = r0, ld_s = s0;
    int nsteps;                                                     // K-steps of this workgroup
    if (MODE == FT_WGRAD) nsteps = kbeg < kend ? (kend - kbeg + FT_BK - 1) / FT_BK : 0;
    else nsteps = (r0 >= p.R || s0 >= p.S) ? 0 : ((p.R - r0 + tstep - 1) / tstep) * ((p.S - s0 + tstep - 1) / tstep) * (CK / FT_BK);

    auto issue = [&](int stage, int k0) {
        if (FT_ABLATE == 1) return;
        const uint32_t sa = lds0 + (uint32_t)(stage * STAGE), sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pa = wave + 4 * i;
            int off = FT_OOB;
            if (MODE == FT_FWD) {
                const int hi = a_y[i] * p.stride - p.pad + ld_r, wi = a_x[i] * p.stride - p.pad + ld_s;
                if (a_img[i] >= 0 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    off = ((((a_img[i] * p.H + hi) * p.W + wi) * p.Cin) + ld_c + rlc) * 4;
            } else if (MODE == FT_DGRAD) {
                const int th = a_y[i] + p.pad - ld_r, tw = a_x[i] + p.pad - ld_s;
                if (a_img[i] >= 0 && th >= 0 && tw >= 0) {
                    int ho, wo; bool ok;
                    if (p.stride == 1) { ho = th; wo = tw; ok = true; }
                    else if (p.stride == 2) { ho = th >> 1; wo = tw >> 1; ok = !((th | tw) & 1); }
                    else { ho = th / p.stride; wo = tw / p.stride; ok = ho * p.stride == th && wo * p.stride == tw; }
                    if (ok && ho < p.Ho && wo < p.Wo) off = ((((a_img[i] * p.Ho + ho) * p.Wo + wo) * p.Cout) + ld_c + rlc) * 4;
                }
            } else {
                const int k = k0 + a_kr[i];
                if (k < kend && a_col[i] < p.M) off = (k * p.Cout + a_col[i]) * 4;
            }
            cp_dma16(rs_a, sa + pa * 1024, off, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pb = wave + 4 * i;
            int off = FT_OOB;
            if (MODE == FT_FWD) {
                if (b_row[i] >= 0) off = (b_row[i] * p.K + k0 + rlc) * 4;
            } else if (MODE == FT_DGRAD) {
                if (b_col[i] < p.Ncol) off = ((((ld_c + b_kr[i]) * p.R + ld_r) * p.S + ld_s) * p.Cin + b_col[i]) * 4;
            } else {
                const int k = k0 + b_kr[i];
                if (k < kend && b_col[i] < p.Ncol) {
                    int q1 = (int)((float)k * q.inv_wo), wo = k - q1 * p.Wo;
                    if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
                    int img = (int)((float)q1 * q.inv_ho), ho = q1 - img * p.Ho;
                    if (ho < 0) { --img; ho += p.Ho; } else if (ho >= p.Ho) { ++img; ho -= p.Ho; }
                    const int hi = ho * p.stride - p.pad + b_r[i], wi = wo * p.stride - p.pad + b_s[i];
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) off = ((((img * p.H + hi) * p.W + wi) * p.Cin) + b_ci[i]) * 4;
                }
            }
            cp_dma16(rs_b, sb + pb * 1024, off, 0);
        }
        if (MODE != FT_WGRAD) {                                    // next K-step: 16 more channels, then the next filter tap
            ld_c += FT_BK;
            if (ld_c == CK) { ld_c = 0; ld_s += tstep; if (ld_s >= p.S) { ld_s = s0; ld_r += tstep; } }
        }
    };

    // ---- fragment read offsets (bytes inside a stage)
    uint32_t afo[MI], bfo[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * WM + mi * 32 + fi;
        afo[mi] = AKM ? (uint32_t)(fh * TM * 4 + row * 4) : (uint32_t)(row * FT_ROWB);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * WN + ni * 32 + fi;
        bfo[ni] = A_BYTES + (BKM ? (uint32_t)(fh * TN * 4 + row * 4) : (uint32_t)(row * FT_ROWB));
    }
    const uint32_t aswz = (uint32_t)((fi >> 2) & 3), bswz = aswz;    // (row >> 2) & 3 of a fragment row = (fi >> 2) & 3: tile rows start at multiples of 32

    cf_f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    // One K-step of MFMAs. The fragments of chunk c + 1 (four k) are requested BEFORE the eight MFMAs of chunk c are issued: a wavefront issues
    // in order, so reads placed after its MFMAs would wait for all of them and then expose the LDS latency once per chunk (the compiler's own
    // schedule of the straightforward loop; SQ counters: matrix pipe 59-67 % busy with 2-3 wavefronts per SIMD).
    auto mfma_step = [&](int stage) {
        const unsigned char* sbase = smem + stage * STAGE;
        float4 ra[2][MI], rb[2][NI];
        float ka[2][MI][2], kb[2][NI][2];
        auto load = [&](int c, int buf) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (!AKM) ra[buf][mi] = *reinterpret_cast<const float4*>(sbase + afo[mi] + (((uint32_t)c ^ aswz) << 4));
                else {
                    ka[buf][mi][0] = *reinterpret_cast<const float*>(sbase + afo[mi] + (4 * c) * (TM * 4));
                    ka[buf][mi][1] = *reinterpret_cast<const float*>(sbase + afo[mi] + (4 * c + 2) * (TM * 4));
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                if (!BKM) rb[buf][ni] = *reinterpret_cast<const float4*>(sbase + bfo[ni] + (((uint32_t)c ^ bswz) << 4));
                else {
                    kb[buf][ni][0] = *reinterpret_cast<const float*>(sbase + bfo[ni] + (4 * c) * (TN * 4));
                    kb[buf][ni][1] = *reinterpret_cast<const float*>(sbase + bfo[ni] + (4 * c + 2) * (TN * 4));
                }
            }
        };
        load(0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {                               // four chunks of four k
            const int cb = c & 1;
            if (c < 3) load(c + 1, cb ^ 1);
            __builtin_amdgcn_sched_barrier(0);                      // (the scheduler otherwise sinks the reads below the MFMAs again)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {                        // MFMA j = 2 c + jj multiplies k = 2 j (lanes 0-31) and 2 j + 1 (lanes 32-63)
                float a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    a[mi] = AKM ? ka[cb][mi][jj] : (jj == 0 ? (fh ? ra[cb][mi].y : ra[cb][mi].x) : (fh ? ra[cb][mi].w : ra[cb][mi].z));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    b[ni] = BKM ? kb[cb][ni][jj] : (jj == 0 ? (fh ? rb[cb][ni].y : rb[cb][ni].x) : (fh ? rb[cb][ni].w : rb[cb][ni].z));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        if (FT_ABLATE == 3) asm volatile("" :: "v"(a[mi]), "v"(b[ni]));
                        else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    };

    // Split-bf16 K-step (ARITH 2 / 3): lane (fi, fh) takes the EIGHT k = 8 fh .. 8 fh + 7 of its row (K-contiguous tiles: two ds_read_b128, the
    // chunks 2 fh and 2 fh + 1; k-major tiles: eight ds_read_b32) as element i of the bf16 fragments — the same k <-> element map for both operands.
    // Term pairs in ascending magnitude, the four accumulator tiles interleaved (no two consecutive MFMAs on one accumulator).
    auto mfma_step_split = [&](int stage) {
        constexpr int NT = ARITH == 3 ? 3 : 2;
        const unsigned char* sbase = smem + stage * STAGE;
        cf_bf16x8 at[MI][NT], bt[NI][NT];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            cf_f32x8 v;
            if (!AKM) {
                const float4 lo4 = *reinterpret_cast<const float4*>(sbase + afo[mi] + (((uint32_t)(2 * fh) ^ aswz) << 4));
                const float4 hi4 = *reinterpret_cast<const float4*>(sbase + afo[mi] + (((uint32_t)(2 * fh + 1) ^ aswz) << 4));
                v = (cf_f32x8){lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            } else {
                // afo = fh * TM * 4 + row * 4 (the exact kernel's k = 2 j + fh rows): here rows k = 8 fh + i
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float*>(sbase + afo[mi] + (7 * fh + i) * (TM * 4));
            }
            if (FT_ABLATE == 5) { for (int s_ = 0; s_ < NT; ++s_) at[mi][s_] = __builtin_bit_cast(cf_bf16x8, (cf_f32x4){v[s_], v[s_ + 1], v[s_ + 2], v[s_ + 3]}); }
            else ft_split<NT>(v, at[mi]);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            cf_f32x8 v;
            if (!BKM) {
                const float4 lo4 = *reinterpret_cast<const float4*>(sbase + bfo[ni] + (((uint32_t)(2 * fh) ^ bswz) << 4));
                const float4 hi4 = *reinterpret_cast<const float4*>(sbase + bfo[ni] + (((uint32_t)(2 * fh + 1) ^ bswz) << 4));
                v = (cf_f32x8){lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float*>(sbase + bfo[ni] + (7 * fh + i) * (TN * 4));
            }
            if (FT_ABLATE == 5) { for (int s_ = 0; s_ < NT; ++s_) bt[ni][s_] = __builtin_bit_cast(cf_bf16x8, (cf_f32x4){v[s_], v[s_ + 1], v[s_ + 2], v[s_ + 3]}); }
            else ft_split<NT>(v, bt[ni]);
        }
        // (term of A, term of B), smallest products first; index NT - 1 = lo, 0 = hi
        constexpr int NP = ARITH == 3 ? 6 : 3;
        constexpr int PA3[6] = {2, 0, 1, 1, 0, 0}, PB3[6] = {0, 2, 1, 0, 1, 0};
        constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
            const int ta = ARITH == 3 ? PA3[pi] : PA2[pi], tb = ARITH == 3 ? PB3[pi] : PB2[pi];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    if (FT_ABLATE == 3) asm volatile("" :: "v"(at[mi][ta]), "v"(bt[ni][tb]));
                    else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[mi][ta], bt[ni][tb], acc[mi][ni], 0, 0, 0);
        }
    };

    int stage = 0;
    if (ARITH == 0) {
        if (nsteps > 0) {
            issue(0, kbeg);
            cp_dma_wait();
            __syncthreads();
            for (int it = 0; it < nsteps; ++it) {
                if (it + 1 < nsteps) issue(stage ^ 1, kbeg + (it + 1) * FT_BK);    // next K-step in flight under this one's MFMAs
                mfma_step(stage);
                if (FT_ABLATE != 4) {
                    cp_dma_wait();                                      // this wavefront's pieces of the next stage have landed ...
                    __syncthreads();                                    // ... everyone's have, and everyone is done reading this stage
                }
                stage ^= 1;
            }
        }
    } else {
        // ring of NS stages, NS - 1 K-steps in flight: the wavefront's pieces land in issue order, so "all but the youngest (NS - 2) K-steps' pieces"
        // = s_waitcnt vmcnt((NS - 2) * (NA + NB)); the barrier then says everyone's pieces of this stage have landed AND everyone has finished reading the
        // stage of the previous K-step, which is the one the next issue overwrites
        constexpr int PF = NS - 1, PIECES = NA + NB;
        for (int s_ = 0; s_ < PF && s_ < nsteps; ++s_) issue(s_, kbeg + s_ * FT_BK);
        int nxt = PF % NS;                                              // stage of the next issue
        for (int it = 0; it < nsteps; ++it) {
            const int younger = nsteps - 1 - it;                        // K-steps issued after this one (capped at PF - 1 by the ring)
            if (FT_ABLATE != 4) {
                if (younger >= PF - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PF - 1) * PIECES) : "memory");
                else if (PF >= 3 && younger == PF - 2 && PF - 2 > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PF >= 3 ? PF - 2 : 0) * PIECES) : "memory");
                else cp_dma_wait();
                __syncthreads();
            }
            if (it + PF < nsteps) { issue(nxt, kbeg + (it + PF) * FT_BK); nxt = nxt + 1 == NS ? 0 : nxt + 1; }
            mfma_step_split(stage);
            stage = stage + 1 == NS ? 0 : stage + 1;
        }
        __syncthreads();                                                // the epilogue reuses the stages' LDS
    }

    // ---- store. C/D layout of the 32 x 32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5): a lane holds ONE column, so
    // storing from the accumulators is 4 bytes per lane (and the fused addend / mask reads likewise). The tile goes through the (now free) K-loop
    // LDS instead, 64 rows at a time ([64][TN] floats: the two stages' bytes at 128 x 128): written column-per-lane, read back row-wise as float4 —
    // 16-byte global accesses, whole 512-byte rows per half-wavefront, a quarter of the memory instructions.
    float* out = MODE == FT_WGRAD ? p.out + (size_t)blockIdx.y * p.M * p.Ncol : p.out;
    if (FT_ABLATE == 2) {
        float s_ = 0.0f;
        for (int mi = 0; mi < MI; ++mi) for (int ni = 0; ni < NI; ++ni) for (int e = 0; e < 16; ++e) s_ += acc[mi][ni][e];
        if (s_ == 1234.5f) out[t] = 1.0f;
        return;
    }
    const bool fused = MODE == FT_DGRAD && (p.addend || p.addend2 || p.mask);
    auto map_row = [&](int row) {                                   // parity-class data gradient: class row (img, i, j) -> pixel (2 i + a, 2 j + b)
        if (MODE == FT_DGRAD && q.cls) {
            const int w2 = p.W >> 1, h2 = p.H >> 1, j = row % w2, qq = row / w2, i = qq % h2, img = qq / h2;
            row = (img * p.H + 2 * i + cls_a) * p.W + 2 * j + cls_b;
        }
        return row;
    };
    if ((p.Ncol & 3) == 0) {
        float* slab = reinterpret_cast<float*>(smem);
        constexpr int C4 = TN / 4, RPI = DIR_TPB / C4;              // float4 per row, rows per pass of the 256 threads
        const int c4 = t % C4, r4 = t / C4;
        const int col = n0 + 4 * c4;
#pragma unroll
        for (int h = 0; h < TM / 64; ++h) {
            if ((wm * WM) / 64 == h) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            slab[(wm * WM - 64 * h + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * TN + wn * WN + ni * 32 + fi] = acc[mi][ni][e];
            }
            __syncthreads();
            if (MODE == FT_FWD && p.stats && t < TN && n0 + t < p.Ncol) {        // BatchNorm statistics of this slab: one thread per column, rows in order
                float s1 = 0.0f, s2 = 0.0f;                                       // (rows past M hold exact zeros: their A rows were zero-filled)
#pragma unroll 8
                for (int r = 0; r < 64; ++r) { const float v = slab[r * TN + t]; s1 += v; s2 += v * v; }
                float* so = p.stats + ((size_t)(mt * (TM / 64) + h) * 2) * p.Ncol + n0 + t;
                so[0] = s1; so[p.Ncol] = s2;
            }
#pragma unroll
            for (int it = 0; it < 64 / RPI; ++it) {
                const int lr = it * RPI + r4;
                int row = m0 + 64 * h + lr;
                if (row < p.M && col < p.Ncol) {
                    row = map_row(row);
                    const size_t o = (size_t)row * p.Ncol + col;
                    float4 v = *reinterpret_cast<const float4*>(slab + lr * TN + 4 * c4);
                    if (fused) {
                        if (p.addend) { const float4 a4 = *reinterpret_cast<const float4*>(p.addend + o); v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w; }
                        if (p.addend2) {
                            const int w = row % p.W, qq = row / p.W, hh = qq % p.H, n = qq / p.H;
                            if (!((hh | w) & 1)) {
                                const float4 a4 = *reinterpret_cast<const float4*>(p.addend2 + (((size_t)n * (p.H >> 1) + (hh >> 1)) * (p.W >> 1) + (w >> 1)) * p.Ncol + col);
                                v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
                            }
                        }
                        if (p.mask) {
                            const float4 m4 = *reinterpret_cast<const float4*>(p.mask + o);
                            if (!(m4.x > 0.0f)) v.x = 0.0f;
                            if (!(m4.y > 0.0f)) v.y = 0.0f;
                            if (!(m4.z > 0.0f)) v.z = 0.0f;
                            if (!(m4.w > 0.0f)) v.w = 0.0f;
                        }
                    }
                    *reinterpret_cast<float4*>(out + o) = v;
                }
            }
            if (h + 1 < TM / 64) __syncthreads();
        }
        return;
    }
    // (column counts that are not multiples of 4: from the accumulators)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = n0 + wn * WN + ni * 32 + fi;
        if (col >= p.Ncol) continue;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int row = m0 + wm * WM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (row >= p.M) continue;
                row = map_row(row);
                const size_t o = (size_t)row * p.Ncol + col;
                float v = acc[mi][ni][e];
                if (fused) {
                    if (p.addend) v += p.addend[o];
                    if (p.addend2) {
                        const int w = row % p.W, qq = row / p.W, hh = qq % p.H, n = qq / p.H;
                        if (!((hh | w) & 1)) v += p.addend2[(((size_t)n * (p.H >> 1) + (hh >> 1)) * (p.W >> 1) + (w >> 1)) * p.Ncol + col];
                    }
                    if (p.mask && !(p.mask[o] > 0.0f)) v = 0.0f;
                }
                out[o] = v;
            }
    }
}

// (A form of the forward kernel with a dedicated LOADER wavefront — wavefront 4 issues all DMA pieces, wavefronts 0-3 only MFMAs — was built,
// bit-identical, and measured 3-40 % SLOWER on every ResNet-50 layer (profiles/r06_f32_loader_wave_negative.txt) and not kept. The phase
// ablation that motivated it: profiles/r06_f32_tile_phase_ablation.txt — time ~ MFMA-only + 0.7 x DMA-only, unchanged without
// barriers / DMA waits.)
// Is the tile kernel applicable? (whole 16-channel K-steps, 16-byte columns, 32-bit byte offsets; the weight gradient's pixel decode is
// exact below 2^24 pixels)
bool ft_ok(int mode, const ConvF32P& p, size_t a_elems, size_t b_elems) {
    if (a_elems * 4 >= (1ull << 31) || b_elems * 4 >= (1ull << 31)) return false;
    if (mode == FT_FWD) return p.Cin % FT_BK == 0;
    if (mode == FT_DGRAD) return p.Cout % FT_BK == 0 && p.Cin % 4 == 0;
    return p.Cout % 4 == 0 && p.Cin % 4 == 0 && p.K < (1 << 24);
}

// split-K of the weight gradient on TM x TN tiles: ~FT_WGRAD_WGS workgroups, at least 16 K-steps per split
constexpr int FT_WGRAD_WGS = 768;
int ft_wgrad_splits(int M, int Ncol, int K, int TM, int TN, int* klen) {
    const long long tiles = (long long)dir_cdiv(M, TM) * dir_cdiv(Ncol, TN);
    long long splits = (FT_WGRAD_WGS + tiles - 1) / tiles;
    const long long max_splits = (K + 16 * FT_BK - 1) / (16 * FT_BK);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long len = (K + splits - 1) / splits;
    len = (len + FT_BK - 1) / FT_BK * FT_BK;
    *klen = (int)len;
    return (int)((K + len - 1) / len);
}
void ft_wgrad_tile(const ConvF32P& p, int* TM, int* TN) { *TM = p.M >= 128 ? 128 : 64; *TN = p.Ncol >= 128 ? 128 : 64; }

// arith: 0 = exact float32 MFMA, 3 / 2 = split-bf16 with three / two terms per operand (the kernel's ARITH)
template <int TM, int TN, int MODE>
int ft_launch(ConvF32T q, int splits, hipStream_t s, int arith = 0) {
    q.ntn = dir_cdiv(q.c.Ncol, TN);
    q.nblocks = dir_cdiv(q.c.M, TM) * q.ntn;
    if (arith == 3) hipLaunchKernelGGL((conv_f32_tile_kernel<TM, TN, MODE, 3>), dim3(q.nblocks, splits), dim3(DIR_TPB), 0, s, q);
    else if (arith == 2) hipLaunchKernelGGL((conv_f32_tile_kernel<TM, TN, MODE, 2>), dim3(q.nblocks, splits), dim3(DIR_TPB), 0, s, q);
    else hipLaunchKernelGGL((conv_f32_tile_kernel<TM, TN, MODE, 0>), dim3(q.nblocks, splits), dim3(DIR_TPB), 0, s, q);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
// `variant` of the C-ABI -> (kernel family, arithmetic): the split forms are tile kernels
inline int ft_arith(int variant) { return variant == DIR_CONV_F32_TILE_X3 ? 3 : (variant == DIR_CONV_F32_TILE_X2 ? 2 : 0); }
inline bool ft_forced_tile(int variant) { return variant >= DIR_CONV_F32_TILE; }

int cf_check(const void* a, const void* b, const void* out, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
             int* Ho, int* Wo) {
    DIR_RETURN_IF(!a || !b || !out, DIR_EINVAL);
    DIR_RETURN_IF(N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0, DIR_EINVAL);
    *Ho = (H + 2 * pad - R) / stride + 1; *Wo = (W + 2 * pad - S) / stride + 1;
    DIR_RETURN_IF(*Ho <= 0 || *Wo <= 0, DIR_EINVAL);
    const long long lim = 1ll << 31;
    DIR_RETURN_IF((long long)N * H * W >= lim || (long long)N * *Ho * *Wo >= lim || (long long)R * S * Cin >= lim ||
                  (long long)R * S * Cout >= lim, DIR_EUNSUPPORTED);
    return DIR_OK;
}

int cf_wgrad_splits(int M, int Ncol, int K, int* klen) {
    const long long tiles = (long long)dir_cdiv(M, CF_BM) * dir_cdiv(Ncol, CF_BN);
    long long splits = (2048 + tiles - 1) / tiles;                              // ~8 workgroups per CU
    const long long max_splits = (K + 16 * CF_BK - 1) / (16 * CF_BK);          // >= 16 K-steps per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long len = (K + splits - 1) / splits;
    len = (len + CF_BK - 1) / CF_BK * CF_BK;
    *klen = (int)len;
    return (int)((K + len - 1) / len);
}

}  // namespace

// variant: 0 = the product's choice (tile kernel where applicable), DIR_CONV_F32_GATHER = the element-gather kernels, DIR_CONV_F32_TILE =
// the tile kernels (DIR_EUNSUPPORTED where they do not apply)
static ConvF32T ft_params(const ConvF32P& p, size_t a_elems, size_t b_elems) {
    ConvF32T q;
    q.c = p; q.ntn = q.nblocks = 0;
    q.kbytes_a = (int)(a_elems * 4); q.kbytes_b = (int)(b_elems * 4);
    q.inv_wo = 1.0f / (float)p.Wo; q.inv_ho = 1.0f / (float)p.Ho;
    q.cls = 0;
    return q;
}

// rows of the statistics list of dir_conv_f32_fwd_stats: one per 64-row slab of the 128-row tiles
extern "C" size_t dir_conv_f32_stats_rows(int N, int Ho, int Wo) {
    const long long M = (long long)N * Ho * Wo;
    return M > 0 ? (size_t)((M + 127) / 128) * 2 : 0;
}

static int cf_fwd_impl(const float* x, const float* w, float* y, float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout, int R, int S,
                       int stride, int pad, int variant, dir_stream_t stream) {
    int Ho, Wo;
    const int rc = cf_check(x, w, y, N, H, W, Cin, Cout, R, S, stride, pad, &Ho, &Wo);
    if (rc != DIR_OK) return rc;
    DIR_RETURN_IF(variant < 0 || variant > DIR_CONV_F32_TILE_X2, DIR_EINVAL);
    ConvF32P p;
    p.a = x; p.b = w; p.out = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = N * Ho * Wo; p.Ncol = Cout; p.K = R * S * Cin; p.klen = p.K;
    p.addend = p.addend2 = p.mask = nullptr; p.stats = nullptr;
    const size_t xe = (size_t)N * H * W * Cin, we = (size_t)Cout * p.K;
    const bool tile_ok = dir_aligned16(x) && dir_aligned16(w) && ft_ok(FT_FWD, p, xe, we);
    DIR_RETURN_IF(ft_forced_tile(variant) && !tile_ok, DIR_EUNSUPPORTED);
    if (stats) {                                                        // fused BatchNorm statistics: tile kernel + its LDS store loop only
        DIR_RETURN_IF(!tile_ok || variant == DIR_CONV_F32_GATHER || (Cout & 3), DIR_EUNSUPPORTED);
        DIR_RETURN_IF((size_t)stats_rows != dir_conv_f32_stats_rows(N, Ho, Wo), DIR_EINVAL);
        p.stats = stats;
    }
    if (tile_ok && variant != DIR_CONV_F32_GATHER) {
        const ConvF32T q = ft_params(p, xe, we);
        return p.Ncol > 64 ? ft_launch<128, 128, FT_FWD>(q, 1, dir_s(stream), ft_arith(variant)) : ft_launch<128, 64, FT_FWD>(q, 1, dir_s(stream), ft_arith(variant));
    }
    DIR_RETURN_IF(dir_cdiv(p.Ncol, CF_BN) > 65535, DIR_EUNSUPPORTED);
    hipLaunchKernelGGL((conv_f32_kfast_kernel<0>), dim3(dir_cdiv(p.M, CF_BM), dir_cdiv(p.Ncol, CF_BN)), dim3(DIR_TPB), 0, dir_s(stream), p);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_conv_f32_fwd_variant(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                                        int stride, int pad, int variant, dir_stream_t stream) {
    return cf_fwd_impl(x, w, y, nullptr, 0, N, H, W, Cin, Cout, R, S, stride, pad, variant, stream);
}

extern "C" int dir_conv_f32_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                                int stride, int pad, dir_stream_t stream) {
    return cf_fwd_impl(x, w, y, nullptr, 0, N, H, W, Cin, Cout, R, S, stride, pad, 0, stream);
}

// forward + the per-channel (sum, sum of squares) partials of y for the BatchNorm that follows ([stats_rows][2][Cout] f32, stats_rows =
// dir_conv_f32_stats_rows): tile kernel geometries with Cout % 4 == 0 only (DIR_EUNSUPPORTED otherwise: the caller lets the BatchNorm count)
extern "C" int dir_conv_f32_fwd_stats(const float* x, const float* w, float* y, float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout,
                                      int R, int S, int stride, int pad, dir_stream_t stream) {
    DIR_RETURN_IF(!stats, DIR_EINVAL);
    return cf_fwd_impl(x, w, y, stats, stats_rows, N, H, W, Cin, Cout, R, S, stride, pad, 0, stream);
}

extern "C" int dir_conv_f32_fwd_stats_variant(const float* x, const float* w, float* y, float* stats, int stats_rows, int N, int H, int W, int Cin,
                                              int Cout, int R, int S, int stride, int pad, int variant, dir_stream_t stream) {
    DIR_RETURN_IF(!stats, DIR_EINVAL);
    return cf_fwd_impl(x, w, y, stats, stats_rows, N, H, W, Cin, Cout, R, S, stride, pad, variant, stream);
}

extern "C" int dir_conv_f32_dgrad_variant(const float* dy, const float* w, const float* addend, const float* addend_s2,
                                          const float* relu_mask, float* dx, int N, int H, int W, int Cin, int Cout, int R, int S,
                                          int stride, int pad, int variant, dir_stream_t stream) {
    int Ho, Wo;
    const int rc = cf_check(dy, w, dx, N, H, W, Cin, Cout, R, S, stride, pad, &Ho, &Wo);
    if (rc != DIR_OK) return rc;
    DIR_RETURN_IF(variant < 0 || variant > DIR_CONV_F32_TILE_X2, DIR_EINVAL);
    DIR_RETURN_IF(addend_s2 && ((H | W) & 1), DIR_EUNSUPPORTED);
    ConvF32P p;
    p.a = dy; p.b = w; p.out = dx;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = N * H * W; p.Ncol = Cin; p.K = R * S * Cout; p.klen = p.K;
    p.addend = addend; p.addend2 = addend_s2; p.mask = relu_mask; p.stats = nullptr;
    const size_t ye = (size_t)N * Ho * Wo * Cout, we = (size_t)Cout * R * S * Cin;
    const bool tile_ok = dir_aligned16(dy) && dir_aligned16(w) && ft_ok(FT_DGRAD, p, ye, we);
    DIR_RETURN_IF(ft_forced_tile(variant) && !tile_ok, DIR_EUNSUPPORTED);
    if (tile_ok && variant != DIR_CONV_F32_GATHER) {
        ConvF32T q = ft_params(p, ye, we);
        int zdim = 1;
        if (stride == 2 && !((H | W) & 1)) { q.cls = 1; q.c.M = N * (H >> 1) * (W >> 1); zdim = 4; }     // four parity classes, only their own filter taps
        return p.Ncol > 64 ? ft_launch<128, 128, FT_DGRAD>(q, zdim, dir_s(stream), ft_arith(variant)) : ft_launch<128, 64, FT_DGRAD>(q, zdim, dir_s(stream), ft_arith(variant));
    }
    DIR_RETURN_IF(dir_cdiv(p.Ncol, CF_BN) > 65535, DIR_EUNSUPPORTED);
    hipLaunchKernelGGL((conv_f32_kfast_kernel<1>), dim3(dir_cdiv(p.M, CF_BM), dir_cdiv(p.Ncol, CF_BN)), dim3(DIR_TPB), 0, dir_s(stream), p);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_conv_f32_dgrad_fused(const float* dy, const float* w, const float* addend, const float* addend_s2,
                                        const float* relu_mask, float* dx, int N, int H, int W, int Cin, int Cout, int R, int S,
                                        int stride, int pad, dir_stream_t stream) {
    return dir_conv_f32_dgrad_variant(dy, w, addend, addend_s2, relu_mask, dx, N, H, W, Cin, Cout, R, S, stride, pad, 0, stream);
}

extern "C" int dir_conv_f32_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int R, int S,
                                  int stride, int pad, dir_stream_t stream) {
    return dir_conv_f32_dgrad_variant(dy, w, nullptr, nullptr, nullptr, dx, N, H, W, Cin, Cout, R, S, stride, pad, 0, stream);
}

// geometry of the weight-gradient GEMM and which kernel `variant` resolves to (true = tile kernel)
static bool cf_wgrad_plan(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int variant, ConvF32P* p, int* splits,
                          int* TM, int* TN) {
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    p->N = N; p->H = H; p->W = W; p->Cin = Cin; p->Ho = Ho; p->Wo = Wo; p->Cout = Cout; p->R = R; p->S = S; p->stride = stride; p->pad = pad;
    p->M = Cout; p->Ncol = R * S * Cin; p->K = N * Ho * Wo;
    p->addend = p->addend2 = p->mask = nullptr; p->stats = nullptr;
    const bool tile_ok = ft_ok(FT_WGRAD, *p, (size_t)p->K * Cout, (size_t)N * H * W * Cin);
    const bool tile = tile_ok && variant != DIR_CONV_F32_GATHER;
    if (tile) { ft_wgrad_tile(*p, TM, TN); *splits = ft_wgrad_splits(p->M, p->Ncol, p->K, *TM, *TN, &p->klen); }
    else *splits = cf_wgrad_splits(p->M, p->Ncol, p->K, &p->klen);
    return tile;
}

// (sized for whichever kernel a variant may choose: the larger of the two split counts)
extern "C" size_t dir_conv_f32_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return 0;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || (long long)N * Ho * Wo >= (1ll << 31)) return 0;
    ConvF32P p;
    int s_tile = 0, s_gather = 0, TM, TN;
    (void)cf_wgrad_plan(N, H, W, Cin, Cout, R, S, stride, pad, DIR_CONV_F32_GATHER, &p, &s_gather, &TM, &TN);
    if (cf_wgrad_plan(N, H, W, Cin, Cout, R, S, stride, pad, 0, &p, &s_tile, &TM, &TN) && s_tile > s_gather) s_gather = s_tile;
    return dir_align_up((size_t)s_gather * Cout * R * S * Cin * sizeof(float), 256);
}

extern "C" int dir_conv_f32_wgrad_variant(const float* dy, const float* x, float* dw, int N, int H, int W, int Cin, int Cout, int R, int S,
                                          int stride, int pad, void* workspace, size_t workspace_bytes, int variant, dir_stream_t stream) {
    int Ho, Wo;
    const int rc = cf_check(dy, x, dw, N, H, W, Cin, Cout, R, S, stride, pad, &Ho, &Wo);
    if (rc != DIR_OK) return rc;
    DIR_RETURN_IF(variant < 0 || variant > DIR_CONV_F32_TILE_X2, DIR_EINVAL);
    ConvF32P p;
    int splits, TM = 64, TN = 64;
    bool tile = cf_wgrad_plan(N, H, W, Cin, Cout, R, S, stride, pad, variant, &p, &splits, &TM, &TN);
    p.a = dy; p.b = x;
    if (tile && !(dir_aligned16(dy) && dir_aligned16(x))) {
        DIR_RETURN_IF(ft_forced_tile(variant), DIR_EUNSUPPORTED);
        tile = cf_wgrad_plan(N, H, W, Cin, Cout, R, S, stride, pad, DIR_CONV_F32_GATHER, &p, &splits, &TM, &TN);
        p.a = dy; p.b = x;
    }
    DIR_RETURN_IF(ft_forced_tile(variant) && !tile, DIR_EUNSUPPORTED);
    const size_t need = dir_align_up((size_t)splits * Cout * R * S * Cin * sizeof(float), 256);
    DIR_RETURN_IF(splits > 1 && (!workspace || workspace_bytes < need), DIR_EWORKSPACE);
    DIR_RETURN_IF(dir_cdiv(p.Ncol, CF_BN) > 65535 || splits > 65535, DIR_EUNSUPPORTED);
    p.out = splits > 1 ? static_cast<float*>(workspace) : dw;
    if (tile) {
        const ConvF32T q = ft_params(p, (size_t)p.K * Cout, (size_t)N * H * W * Cin);
        int rc2;
        const int ar = ft_arith(variant);
        if (TM == 128) rc2 = TN == 128 ? ft_launch<128, 128, FT_WGRAD>(q, splits, dir_s(stream), ar) : ft_launch<128, 64, FT_WGRAD>(q, splits, dir_s(stream), ar);
        else rc2 = TN == 128 ? ft_launch<64, 128, FT_WGRAD>(q, splits, dir_s(stream), ar) : ft_launch<64, 64, FT_WGRAD>(q, splits, dir_s(stream), ar);
        if (rc2 != DIR_OK) return rc2;
    } else {
        hipLaunchKernelGGL(conv_f32_wgrad_kernel, dim3(dir_cdiv(p.M, CF_BM), dir_cdiv(p.Ncol, CF_BN), splits), dim3(DIR_TPB), 0, dir_s(stream), p);
        DIR_LAUNCH_CHECK();
    }
    if (splits > 1) {
        const size_t n = (size_t)p.M * p.Ncol;
        int grid = dir_cdiv((long long)n, DIR_TPB); if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(conv_f32_wgrad_reduce_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const float*>(workspace), dw, n, splits);
        DIR_LAUNCH_CHECK();
    }
    return DIR_OK;
}

extern "C" int dir_conv_f32_wgrad(const float* dy, const float* x, float* dw, int N, int H, int W, int Cin, int Cout, int R, int S,
                                  int stride, int pad, void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    return dir_conv_f32_wgrad_variant(dy, x, dw, N, H, W, Cin, Cout, R, S, stride, pad, workspace, workspace_bytes, 0, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// float32 pools of the parity mode (resnet.py:82,131 MaxPool2d(3, 2, 1); :85,136 AvgPool2d(7) on the 7x7 map), NHWC.
namespace {
__global__ void __launch_bounds__(DIR_TPB)
maxpool_f32_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int c = (int)(i % C); long long q = i / C;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho); const int n = (int)(q / Ho);
        float best = -INFINITY; int bi = 0;
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float v = x[(((size_t)n * H + hi) * W + wi) * C + c];
                if (v > best || v != v) { best = v; bi = r * 3 + s; }            // first maximum wins, NaN propagates (torch)
            }
        }
        y[i] = best; idx[i] = (uint8_t)bi;
    }
}

__global__ void __launch_bounds__(DIR_TPB)
maxpool_f32_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int c = (int)(i % C); long long q = i / C;
        const int w = (int)(q % W); q /= W;
        const int h = (int)(q % H); const int n = (int)(q / H);
        float acc = 0.0f;
        for (int a = 0; a < 2; ++a) {
            const int ho = (h + 1) / 2 - a, r = h - (2 * ho - 1);
            if (ho < 0 || ho >= Ho || r < 0 || r > 2) continue;
            for (int b = 0; b < 2; ++b) {
                const int wo = (w + 1) / 2 - b, s = w - (2 * wo - 1);
                if (wo < 0 || wo >= Wo || s < 0 || s > 2) continue;
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
                if (idx[o] == (uint8_t)(r * 3 + s)) acc += dy[o];
            }
        }
        dx[i] = acc;
    }
}

// y[n][c] = (sum over the HW pixels in order) / HW, like AvgPool2d's sequential window sum followed by one division
__global__ void __launch_bounds__(DIR_TPB)
avgpool_f32_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const long long total = (long long)N * C;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int c = (int)(i % C); const int n = (int)(i / C);
        float s = 0.0f;
        for (int q = 0; q < HW; ++q) s += x[((size_t)n * HW + q) * C + c];
        y[i] = s / (float)HW;
    }
}

__global__ void __launch_bounds__(DIR_TPB)
avgpool_f32_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int HW, int C) {
    const long long total = (long long)N * HW * C;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int c = (int)(i % C); const int n = (int)(i / ((long long)HW * C));
        dx[i] = dy[(size_t)n * C + c] / (float)HW;
    }
}
}  // namespace

static int cf_grid(long long total) { long long g = (total + DIR_TPB - 1) / DIR_TPB; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

extern "C" int dir_maxpool3x3s2_f32_fwd(const float* x, float* y, void* argmax, int N, int H, int W, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0, DIR_EINVAL);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_f32_fwd_kernel, dim3(cf_grid((long long)N * Ho * Wo * C)), dim3(DIR_TPB), 0, dir_s(stream), x, y,
                       static_cast<uint8_t*>(argmax), N, H, W, C, Ho, Wo);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_maxpool3x3s2_f32_bwd(const float* dy, const void* argmax, float* dx, int N, int H, int W, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !dx || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0, DIR_EINVAL);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_f32_bwd_kernel, dim3(cf_grid((long long)N * H * W * C)), dim3(DIR_TPB), 0, dir_s(stream), dy,
                       static_cast<const uint8_t*>(argmax), dx, N, H, W, C, Ho, Wo);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_avgpool_f32_fwd(const float* x, float* y, int N, int HW, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || N <= 0 || HW <= 0 || C <= 0, DIR_EINVAL);
    hipLaunchKernelGGL(avgpool_f32_fwd_kernel, dim3(cf_grid((long long)N * C)), dim3(DIR_TPB), 0, dir_s(stream), x, y, N, HW, C);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_avgpool_f32_bwd(const float* dy, float* dx, int N, int HW, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !dx || N <= 0 || HW <= 0 || C <= 0, DIR_EINVAL);
    hipLaunchKernelGGL(avgpool_f32_bwd_kernel, dim3(cf_grid((long long)N * HW * C)), dim3(DIR_TPB), 0, dir_s(stream), dy, dx, N, HW, C);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
