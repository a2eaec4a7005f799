// MFMA implicit-GEMM convolution for NHWC bf16 activations on MI355X / gfx950 (CDNA4).
//
// Replaces the nn.Conv2d layers of imdb-wiki-dir/resnet.py:41-70,79,112-116 (cuDNN in the reference): the only
// genuinely dense contraction of the hot path, so the only place MFMA is used.
//
//   Y[m, co] = sum_{r,s,ci} X[n, ho*stride - pad + r, wo*stride - pad + s, ci] * Wt[co, r, s, ci],   m = (n, ho, wo)
//
// GEMM view: M = N*Ho*Wo rows, N = Cout columns, K = R*S*Cin with Cin % 64 == 0, so one 64-wide K-step never
// straddles a filter tap and the A-operand loader is "row pointer + bounds predicate" (zero fill at the borders).
// Both operands are K-contiguous in memory (NHWC activations, [Cout][R][S][Cin] weights = torch channels_last), which
// is exactly the 16-bytes-per-lane fragment of v_mfma_f32_32x32x16_bf16 — no transposes anywhere.
//
// Workgroup = 256 threads = 4 wavefronts (one per SIMD), tile 128 x BN x 64 (BN = 128 or 64):
//   global -> registers (16 B per lane, next K-step prefetched while the current one is multiplied)
//   registers -> LDS, rows of 128 B, 16-B chunks XOR-swizzled with (row >> 1) & 7 so that both the 8-lane
//   ds_write_b128 groups and the four 16-lane ds_read_b128 groups are bank-conflict free
//   LDS -> MFMA fragments -> 32x32x16 bf16 MFMA, fp32 accumulators (2x2 or 1x2 tiles of 32x32 per wavefront)
//   epilogue: bf16 rounding, optional per-channel (sum, sum of squares) partials of the rounded outputs for the
//   following BatchNorm (no separate statistics pass over Y), LDS transpose staging, 16-B coalesced row stores.
// Workgroup ids are remapped so that the N-tiles of one M-tile run on the same XCD (A tile re-reads hit that L2).
#include <cstdlib>
#include "dir_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef const __attribute__((address_space(1))) u32x4* gvec_ptr;        // explicit global address space (no flat loads)
__device__ __forceinline__ uint4 cv_gload(const uint16_t* base, ptrdiff_t elem_off) {
    const u32x4 v = *reinterpret_cast<gvec_ptr>(reinterpret_cast<uintptr_t>(base + elem_off));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

struct ConvP {
    const uint16_t* x; const uint16_t* w; uint16_t* y; float* stats;
    const uint16_t* addend;   // optional [M][Cout] bf16 added to the rounded result (fused gradient accumulation)
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
    int M, KT, cpk, ntn, nblocks;
    int simple;               // 1x1, stride 1, pad 0: row m of the GEMM is row m of x (no index arithmetic at all)
    float inv_wo, inv_ho;     // reciprocals for the (n, ho, wo) decode of the general case
    int nbuf;                 // LDS stages of the K loop: 2 = prefetched tile written while the current one is read, 1 = extra barrier
};

constexpr int CV_BM = 128, CV_BK = 64, CV_ROWB = CV_BK * 2;      // 128-byte LDS rows

__device__ __forceinline__ uint32_t cv_f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <int BN>
__global__ void __launch_bounds__(DIR_TPB)
conv_igemm_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_BYTES = CV_BM * CV_ROWB;              // 16 KB
    constexpr int B_BYTES = BN * CV_ROWB;
    constexpr int BROWS = BN / 32;                        // B rows per loader thread
    constexpr int MI = (BN == 128) ? 2 : 1;               // 32x32 tiles per wavefront along M
    constexpr int NI = 2;                                 //                      ... along N
    constexpr int WM = MI * 32;
    unsigned char* As = smem;
    unsigned char* Bs = smem + p.nbuf * A_BYTES;

    // ---- workgroup -> (m tile, n tile), XCD-aware and bijective
    int lin;
    {
        const int b = blockIdx.x, q = p.nblocks / 8, r = p.nblocks % 8, xcd = b % 8, i = b / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int mt = lin / p.ntn, nt = lin - mt * p.ntn;
    const int m0 = mt * CV_BM, n0 = nt * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave;
    const int wn = (BN == 128) ? (wave & 1) : 0;

    // ---- loader coordinates: thread loads the 16-B chunk (t & 7) of rows (t >> 3) + 32 i.
    // Everything position dependent is computed ONCE: per row a signed element offset of its (hi0, wi0) pixel and a
    // bit mask of the filter taps that fall inside the image; per K-step only a wave-uniform offset is added.
    const int lrow = t >> 3, lchunk = t & 7;
    int aoff[4];
    uint32_t amask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lrow + 32 * i;
        aoff[i] = 0; amask[i] = 0;
        if (m < p.M) {
            if (p.simple) { aoff[i] = m * p.Cin + lchunk * 8; amask[i] = 1u; continue; }
            // (n, ho, wo) from m: float reciprocal + one correction step instead of integer divisions (exact for m < 2^24)
            int q1 = (int)((float)m * p.inv_wo), wo = m - q1 * p.Wo;
            if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
            int n = (int)((float)q1 * p.inv_ho), ho = q1 - n * p.Ho;
            if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            aoff[i] = ((n * p.H + hi0) * p.W + wi0) * p.Cin + lchunk * 8;
            for (int r = 0; r < p.R; ++r)
                for (int s2 = 0; s2 < p.S; ++s2)
                    if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + s2) < (unsigned)p.W) amask[i] |= 1u << (r * p.S + s2);
        }
    }
    const size_t K = (size_t)p.KT * CV_BK;
    const uint16_t* wrow[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) wrow[i] = p.w + (size_t)(n0 + lrow + 32 * i) * K + lchunk * 8;

    // K-step cursor (wave-uniform): filter tap and 64-channel block of the NEXT tile to fetch
    int ld_tap = 0, ld_c = 0, ld_r = 0, ld_s = 0;
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;              // named registers: no private-memory arrays
    rb2 = rb3 = make_uint4(0, 0, 0, 0);

#define CV_LOAD_TILE()                                                                                          \
    {                                                                                                           \
        const int koff = (ld_r * p.W + ld_s) * p.Cin + ld_c * CV_BK;                                            \
        const uint32_t bit = 1u << ld_tap;                                                                      \
        /* out-of-image taps read a dummy in-bounds address (offset 0) and are zeroed after: no divergence */   \
        const bool v0 = amask[0] & bit, v1 = amask[1] & bit, v2 = amask[2] & bit, v3 = amask[3] & bit;          \
        ra0 = cv_gload(p.x, v0 ? (ptrdiff_t)(aoff[0] + koff) : 0);                                                                \
        ra1 = cv_gload(p.x, v1 ? (ptrdiff_t)(aoff[1] + koff) : 0);                                                                \
        ra2 = cv_gload(p.x, v2 ? (ptrdiff_t)(aoff[2] + koff) : 0);                                                                \
        ra3 = cv_gload(p.x, v3 ? (ptrdiff_t)(aoff[3] + koff) : 0);                                                                \
        rb0 = cv_gload(wrow[0], 0); wrow[0] += CV_BK;                                                   \
        rb1 = cv_gload(wrow[1], 0); wrow[1] += CV_BK;                                                   \
        if (BROWS == 4) {                                                                                       \
            rb2 = cv_gload(wrow[BROWS - 2], 0); wrow[BROWS - 2] += CV_BK;                               \
            rb3 = cv_gload(wrow[BROWS - 1], 0); wrow[BROWS - 1] += CV_BK;                               \
        }                                                                                                       \
        const uint32_t k0 = v0 ? ~0u : 0u, k1 = v1 ? ~0u : 0u, k2 = v2 ? ~0u : 0u, k3 = v3 ? ~0u : 0u;          \
        ra0.x &= k0; ra0.y &= k0; ra0.z &= k0; ra0.w &= k0; ra1.x &= k1; ra1.y &= k1; ra1.z &= k1; ra1.w &= k1; \
        ra2.x &= k2; ra2.y &= k2; ra2.z &= k2; ra2.w &= k2; ra3.x &= k3; ra3.y &= k3; ra3.z &= k3; ra3.w &= k3; \
        if (++ld_c == p.cpk) { ld_c = 0; ++ld_tap; if (++ld_s == p.S) { ld_s = 0; ++ld_r; } }                   \
    }
#define CV_ST(base, bytes, row, v) *reinterpret_cast<uint4*>((base) + (bytes) + (row) * CV_ROWB + ((lchunk ^ (((row) >> 1) & 7)) << 4)) = (v)
#define CV_STORE_TILE(buf)                                                                                      \
    {                                                                                                           \
        CV_ST(As, (buf) * A_BYTES, lrow, ra0); CV_ST(As, (buf) * A_BYTES, lrow + 32, ra1);                      \
        CV_ST(As, (buf) * A_BYTES, lrow + 64, ra2); CV_ST(As, (buf) * A_BYTES, lrow + 96, ra3);                 \
        CV_ST(Bs, (buf) * B_BYTES, lrow, rb0); CV_ST(Bs, (buf) * B_BYTES, lrow + 32, rb1);                      \
        if (BROWS == 4) { CV_ST(Bs, (buf) * B_BYTES, lrow + 64, rb2); CV_ST(Bs, (buf) * B_BYTES, lrow + 96, rb3); } \
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    CV_LOAD_TILE();
    CV_STORE_TILE(0);
    __syncthreads();
    const int frow = lane & 31, fhalf = lane >> 5;
    const bool dbuf = p.nbuf == 2;
    for (int kt = 0; kt < p.KT; ++kt) {
        const int buf = dbuf ? (kt & 1) : 0;
        const bool more = kt + 1 < p.KT;
        if (more) CV_LOAD_TILE();                              // global loads in flight during the MFMAs
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[MI], b[NI];
            const int chunk = kk * 2 + fhalf;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * WM + mi * 32 + frow;
                a[mi] = *reinterpret_cast<const bf16x8*>(As + buf * A_BYTES + row * CV_ROWB + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wn * 64 + ni * 32 + frow;
                b[ni] = *reinterpret_cast<const bf16x8*>(Bs + buf * B_BYTES + row * CV_ROWB + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (!dbuf) __syncthreads();                          // single stage: everyone is done reading before the overwrite
        if (more) CV_STORE_TILE(dbuf ? (buf ^ 1) : 0);
        __syncthreads();
    }
#undef CV_LOAD_TILE
#undef CV_STORE_TILE
#undef CV_ST

    // ---- epilogue. C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
    // Neighbouring lanes hold neighbouring columns, so lane pairs swap one value per register pair through DPP
    // (quad_perm [1,0,3,2]) and every lane stores packed bf16x2 dwords: even lanes the even-numbered rows of the pair,
    // odd lanes the odd ones. Staging rows are padded by 64 B so the two rows of a pair land on disjoint banks.
    constexpr int CS_STRIDE = BN * 2 + 64;                      // bytes per staging row
    unsigned char* Cs = smem;                                   // [128][CS_STRIDE] (<= 40 KB, the K-loop buffers are free now)
    float* Ss = reinterpret_cast<float*>(smem + CV_BM * CS_STRIDE);   // [4 waves][2][64] column partials
    float csum[NI], csq[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { csum[ni] = 0.0f; csq[ni] = 0.0f; }
    const bool odd = lane & 1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const uint32_t h0 = cv_f2bf(acc[mi][ni][e]), h1 = cv_f2bf(acc[mi][ni][e + 1]);
                const float f0 = __uint_as_float(h0 << 16), f1 = __uint_as_float(h1 << 16);   // statistics of what is stored
                csum[ni] += f0 + f1; csq[ni] += f0 * f0 + f1 * f1;
                const uint32_t send = odd ? h0 : h1;             // what the partner lane needs
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);
                const uint32_t packed = odd ? (recv | (h1 << 16)) : (h0 | (recv << 16));
                const int row = wm * WM + mi * 32 + ((e + (odd ? 1 : 0)) & 3) + 8 * (e >> 2) + 4 * fhalf;
                const int col = wn * 64 + ni * 32 + (frow & ~1);
                *reinterpret_cast<uint32_t*>(Cs + row * CS_STRIDE + col * 2) = packed;
            }
    if (p.stats) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            csum[ni] += __shfl_xor(csum[ni], 32, DIR_WAVE);
            csq[ni] += __shfl_xor(csq[ni], 32, DIR_WAVE);
            if (fhalf == 0) { Ss[(wave * 2 + 0) * 64 + ni * 32 + frow] = csum[ni]; Ss[(wave * 2 + 1) * 64 + ni * 32 + frow] = csq[ni]; }
        }
    }
    __syncthreads();
    if (p.stats && t < 2 * BN) {                                // one thread per (which, column)
        const int which = t / BN, col = t - which * BN;
        float s = 0.0f;
        if (BN == 128) { const int w0 = col >> 6; s = Ss[((w0) * 2 + which) * 64 + (col & 63)] + Ss[((w0 + 2) * 2 + which) * 64 + (col & 63)]; }
        else { s = Ss[(0 * 2 + which) * 64 + col] + Ss[(1 * 2 + which) * 64 + col] + Ss[(2 * 2 + which) * 64 + col] + Ss[(3 * 2 + which) * 64 + col]; }
        p.stats[((size_t)mt * 2 + which) * p.Cout + n0 + col] = s;
    }
    constexpr int CPR = BN / 8;                                 // 16-B chunks per C row
#pragma unroll
    for (int i = 0; i < (CV_BM * CPR) / DIR_TPB; ++i) {
        const int q = t + DIR_TPB * i, row = q / CPR, ch = q - row * CPR;
        if (m0 + row < p.M) {
            uint4 c = *reinterpret_cast<const uint4*>(Cs + row * CS_STRIDE + ch * 16);
            const size_t go = (size_t)(m0 + row) * p.Cout + n0 + ch * 8;
            if (p.addend) {                                     // y = bf16(bf16(conv) + addend), like an eager add kernel
                const uint4 a = *reinterpret_cast<const uint4*>(p.addend + go);
                uint32_t cw[4] = {c.x, c.y, c.z, c.w};
                const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = __uint_as_float(cw[q] << 16) + __uint_as_float(aw[q] << 16);
                    const float hi = __uint_as_float(cw[q] & 0xffff0000u) + __uint_as_float(aw[q] & 0xffff0000u);
                    cw[q] = cv_f2bf(lo) | (cv_f2bf(hi) << 16);
                }
                c = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            }
            *reinterpret_cast<uint4*>(p.y + go) = c;
        }
    }
}

}  // namespace

extern "C" size_t dir_conv_stats_rows(int N, int Ho, int Wo) {
    const long long M = (long long)N * Ho * Wo;
    return (size_t)((M + CV_BM - 1) / CV_BM);
}

extern "C" int dir_conv_fwd_add(const void* x, const void* w, const void* addend, void* y, float* stats, int N, int H,
                                int W, int Cin, int Cout, int R, int S, int stride, int pad, dir_stream_t stream);

extern "C" int dir_conv_fwd(const void* x, const void* w, void* y, float* stats, int N, int H, int W, int Cin,
                            int Cout, int R, int S, int stride, int pad, dir_stream_t stream) {
    return dir_conv_fwd_add(x, w, nullptr, y, stats, N, H, W, Cin, Cout, R, S, stride, pad, stream);
}

extern "C" int dir_conv_fwd_add(const void* x, const void* w, const void* addend, void* y, float* stats, int N, int H,
                                int W, int Cin, int Cout, int R, int S, int stride, int pad, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !w || !y, DIR_EINVAL);
    DIR_RETURN_IF(addend && (!dir_aligned16(addend) || stats), DIR_EINVAL);     // statistics are of the conv result alone
    DIR_RETURN_IF(N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0, DIR_EINVAL);
    DIR_RETURN_IF(Cin % CV_BK != 0 || Cout % 64 != 0, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(w) || !dir_aligned16(y), DIR_EINVAL);
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    DIR_RETURN_IF(Ho <= 0 || Wo <= 0, DIR_EINVAL);
    const long long M = (long long)N * Ho * Wo;
    DIR_RETURN_IF(M >= (1ll << 24) || (long long)N * H * W * Cin >= (1ll << 31) || M * Cout >= (1ll << 31) || R * S > 32, DIR_EUNSUPPORTED);
    ConvP p;
    p.x = static_cast<const uint16_t*>(x); p.w = static_cast<const uint16_t*>(w); p.y = static_cast<uint16_t*>(y);
    p.stats = stats;
    p.addend = static_cast<const uint16_t*>(addend);
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = (int)M; p.cpk = Cin / CV_BK; p.KT = R * S * p.cpk;
    p.simple = (R == 1 && S == 1 && stride == 1 && pad == 0) ? 1 : 0;
    p.inv_wo = 1.0f / (float)Wo; p.inv_ho = 1.0f / (float)Ho;
    const int mtiles = (int)((M + CV_BM - 1) / CV_BM);
    const bool wide = (Cout % 128 == 0);
    p.ntn = wide ? Cout / 128 : Cout / 64;
    p.nblocks = mtiles * p.ntn;
    hipStream_t s = dir_s(stream);
    // LDS stages: 2 (64 KB, 2 workgroups per CU) for long K loops, 1 (43 KB, 3 per CU) when the loop is short and the
    // layer is bound by memory latency rather than by MFMA issue. DIR_CONV_NBUF=1|2 overrides (experiments).
    static const int force_nbuf = []() { const char* e = getenv("DIR_CONV_NBUF"); return e ? atoi(e) : 0; }();
    static const int nbuf_kt = []() { const char* e = getenv("DIR_CONV_NBUF_KT"); return e ? atoi(e) : 18; }();
    p.nbuf = force_nbuf ? force_nbuf : (p.KT <= nbuf_kt ? 1 : 2);
    if (wide) {
        const int stage = CV_BM * (128 * 2 + 64) + 2048;            // epilogue staging + column partials
        const int loop = p.nbuf * (CV_BM * CV_ROWB + 128 * CV_ROWB);
        const int lds = loop > stage ? loop : stage;
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<128>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 65536), true);
        (void)once;
        hipLaunchKernelGGL(conv_igemm_kernel<128>, dim3(p.nblocks), dim3(DIR_TPB), lds, s, p);
    } else {
        const int stage = CV_BM * (64 * 2 + 64) + 2048;
        const int loop = p.nbuf * (CV_BM * CV_ROWB + 64 * CV_ROWB);
        const int lds = loop > stage ? loop : stage;
        hipLaunchKernelGGL(conv_igemm_kernel<64>, dim3(p.nblocks), dim3(DIR_TPB), lds, s, p);
    }
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight preparation: one launch per conv layer and optimizer step turns the float32 master weight
// [Cout][R][S][Cin] into the two bf16 operands the MFMA kernels consume — w16 (same layout, forward / wgrad shape)
// and, optionally, w16_rot [Cin][R][S][Cout] with the taps rotated by 180 degrees (the data-gradient convolution's
// weight) — instead of a cast + flip + permute + copy chain of library kernels.
namespace {
__global__ void __launch_bounds__(DIR_TPB)
conv_prep_weights_kernel(const float* __restrict__ w, int Cout, int RS, int Cin, uint16_t* __restrict__ w16,
                         uint16_t* __restrict__ w16_rot) {
    const size_t n = (size_t)Cout * RS * Cin;
    for (size_t i = (size_t)blockIdx.x * DIR_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * DIR_TPB) {
        const uint16_t h = (uint16_t)cv_f2bf(w[i]);
        w16[i] = h;
        if (w16_rot) {
            const int ci = (int)(i % Cin); const size_t t1 = i / Cin; const int tap = (int)(t1 % RS); const int co = (int)(t1 / RS);
            w16_rot[((size_t)ci * RS + (RS - 1 - tap)) * Cout + co] = h;
        }
    }
}
}  // namespace

extern "C" int dir_conv_prep_weights(const float* w, int Cout, int R, int S, int Cin, void* w16, void* w16_rot,
                                     dir_stream_t stream) {
    DIR_RETURN_IF(!w || !w16 || Cout <= 0 || R <= 0 || S <= 0 || Cin <= 0, DIR_EINVAL);
    const size_t n = (size_t)Cout * R * S * Cin;
    int grid = dir_cdiv((long long)n, DIR_TPB); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(conv_prep_weights_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), w, Cout, R * S, Cin,
                       static_cast<uint16_t*>(w16), static_cast<uint16_t*>(w16_rot));
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
