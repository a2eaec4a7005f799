// Stem convolution 7x7 / stride 2 / pad 3, 3 -> 64 channels (imdb-wiki-dir/resnet.py:79,129) for NHWC bf16 images on
// MI355X / gfx950, as an MFMA GEMM without an im2col buffer.
//
// With 3 input channels the implicit-GEMM K axis (r, s, c) has runs of only 3 contiguous elements per tap, but for a
// fixed filter row r the 7 taps x 3 channels of one output pixel are ONE contiguous run of 21 input elements starting at
// column 2 wo - 3. Widening the window by one (zero-weight) pixel on the left makes that run 24 elements = three
// 8-element MFMA operand groups that start at byte 8 + 12 wo of the staged input row: 4-byte aligned, so the A fragment
// is two ds_read2_b32 straight out of the raw row image — no gather, no per-tap masking (the row image carries its own
// zero padding). K = 7 rows x 24 = 168, padded to 176 = 11 MFMA steps of 16 with zero weights.
//
// Workgroup (256 threads, 4 wavefronts) = one output row (n, ho): up to 128 pixels x 64 channels; persistent over rows
// so the packed weights (22 KB) are staged in LDS once. Per row: 7 input rows -> LDS (16-B chunks, coalesced), 11 x
// (A: 4 dwords, B: 2 ds_read_b128, 2 MFMA 32x32x16) per wavefront, epilogue as in dir_conv.hip (v_cvt_pk_bf16_f32,
// b16 LDS staging, 16-B row stores). The per-channel sums of the ROUNDED outputs and their squares are accumulated across
// the workgroup's rows and written once: stats[workgroup][2][64], the partial list dir_bn_prepare_train consumes.
#include "dir_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int ST_CIN = 3, ST_COUT = 64, ST_R = 7;
constexpr int ST_ROWE = 800;                       // bf16 elements per staged input row (col 0 at element 16)
constexpr int ST_ROWB = ST_ROWE * 2;
constexpr int ST_KP = 176;                         // padded K: 7 x 24 = 168 real, 8 zero
constexpr int ST_WSTRIDE = 368;                    // bytes per weight row in LDS: 92 dwords, conflict-free ds_read_b128 over 16 rows
constexpr int ST_IN_BYTES = ST_R * ST_ROWB;        // 11 200
constexpr int ST_W_BYTES = ST_COUT * ST_WSTRIDE;   // 23 552
constexpr int ST_CS_STRIDE = ST_COUT * 2 + 16;     // staging row: 128 B + pad (rows r, r+4 on disjoint banks)
constexpr int ST_CS_BYTES = 128 * ST_CS_STRIDE;    // 18 432
constexpr int ST_LDS = ST_IN_BYTES + ST_W_BYTES + ST_CS_BYTES;   // 53 184 B: 3 workgroups per CU
constexpr int ST_MAX_BLOCKS = 768;

struct StemP { const uint16_t* x; const uint16_t* wp; uint16_t* y; float* stats; int N, H, W, Ho, Wo, rows; };

__global__ void __launch_bounds__(DIR_TPB)
stem_conv_kernel(StemP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* in = smem;
    unsigned char* ws = smem + ST_IN_BYTES;
    unsigned char* cs = ws + ST_W_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, frow = lane & 31, fhalf = lane >> 5;

    // packed weights [64][176] bf16 -> LDS rows of 368 B; zero the whole input image once (its pads stay zero)
    for (int i = t; i < ST_COUT * (ST_KP / 8); i += DIR_TPB) {
        const int row = i / (ST_KP / 8), ch = i - row * (ST_KP / 8);
        *reinterpret_cast<u32x4*>(ws + row * ST_WSTRIDE + ch * 16) = *reinterpret_cast<const u32x4*>(p.wp + row * ST_KP + ch * 8);
    }
    for (int i = t; i < ST_IN_BYTES / 16; i += DIR_TPB) *reinterpret_cast<u32x4*>(in + i * 16) = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();

    const int pix = wave * 32 + frow;                               // A-operand row of this lane
    const int chunks = p.W * ST_CIN * 2 / 16;                       // 16-B chunks per input row (W % 8 == 0)
    // store-loop roles: 8 chunks of 16 B per output pixel
    const int srow = t >> 3, sch = t & 7;
    float ss0[8], ss1[8];                                           // per-channel sums of this thread's 8 channels over all its pixels
#pragma unroll
    for (int j = 0; j < 8; ++j) { ss0[j] = 0.0f; ss1[j] = 0.0f; }

    for (int row = blockIdx.x; row < p.rows; row += gridDim.x) {
        const int n = row / p.Ho, ho = row - n * p.Ho;
        // ---- the 7 input rows of this output row -> LDS (rows outside the image: zeros)
        for (int i = t; i < ST_R * chunks; i += DIR_TPB) {
            const int r = i / chunks, j = i - r * chunks;
            const int hi = 2 * ho - 3 + r;
            u32x4 v = {0u, 0u, 0u, 0u};
            if ((unsigned)hi < (unsigned)p.H) v = *reinterpret_cast<const u32x4*>(p.x + ((size_t)(n * p.H + hi) * p.W) * ST_CIN + j * 8);
            *reinterpret_cast<u32x4*>(in + r * ST_ROWB + 32 + j * 16) = v;
        }
        __syncthreads();

        // ---- 11 K-steps: lanes 0-31 take operand group 2 ks, lanes 32-63 group 2 ks + 1; group q = (filter row q / 3,
        // elements 8 (q % 3) .. + 8 of the 24-element window). Group 21 has zero weights (any finite A will do).
        f32x16 acc[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < ST_KP / 16; ++ks) {
            const int q = 2 * ks + fhalf;
            const int r = q / 3 > ST_R - 1 ? ST_R - 1 : q / 3, tq = q - 3 * (q / 3);
            const unsigned char* ap = in + r * ST_ROWB + (4 + 6 * pix + 8 * tq) * 2;
            u32x4 av;
            av[0] = *reinterpret_cast<const uint32_t*>(ap); av[1] = *reinterpret_cast<const uint32_t*>(ap + 4);
            av[2] = *reinterpret_cast<const uint32_t*>(ap + 8); av[3] = *reinterpret_cast<const uint32_t*>(ap + 12);
            const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(ws + (ni * 32 + frow) * ST_WSTRIDE + q * 16);
                acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ni], 0, 0, 0);
            }
        }

        // ---- epilogue: C/D layout col = lane & 31 (channel), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (pixel in the wave's 32)
        unsigned char* cbase = cs + (wave * 32 + 4 * fhalf) * ST_CS_STRIDE + frow * 2;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 v = {acc[ni][e], acc[ni][e + 1]};
                const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                unsigned char* d = cbase + ((e & 3) + 8 * (e >> 2)) * ST_CS_STRIDE + ni * 64;
                *reinterpret_cast<uint16_t*>(d) = (uint16_t)pk;
                *reinterpret_cast<uint16_t*>(d + ST_CS_STRIDE) = (uint16_t)(pk >> 16);
            }
        __syncthreads();
        uint16_t* yrow = p.y + (size_t)row * p.Wo * ST_COUT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                               // 128 pixels x 8 chunks / 256 threads
            const int px = srow + 32 * i;
            if (px < p.Wo) {
                const u32x4 c = *reinterpret_cast<const u32x4*>(cs + px * ST_CS_STRIDE + sch * 16);
                __builtin_nontemporal_store(c, reinterpret_cast<u32x4*>(yrow + (size_t)px * ST_COUT + sch * 8));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float f0 = __uint_as_float(c[q] << 16), f1 = __uint_as_float(c[q] & 0xffff0000u);
                    ss0[2 * q] += f0; ss1[2 * q] += f0 * f0;
                    ss0[2 * q + 1] += f1; ss1[2 * q + 1] += f1 * f1;
                }
            }
        }
        // (the next row's staging writes `in`, last read before the barrier above; `cs` is rewritten only after the next
        //  row's first barrier, which every thread reaches after finishing this store loop)
    }

    if (p.stats) {                                                  // 32 pixel lanes per channel group -> one row of partials
        __syncthreads();
        float* red = reinterpret_cast<float*>(cs);                  // [2][32][64] floats = 16 KB
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[(0 * 32 + srow) * ST_COUT + sch * 8 + j] = ss0[j];
            red[(1 * 32 + srow) * ST_COUT + sch * 8 + j] = ss1[j];
        }
        __syncthreads();
        if (t < 2 * ST_COUT) {
            const int which = t / ST_COUT, c = t - which * ST_COUT;
            float s = 0.0f;
            for (int r = 0; r < 32; ++r) s += red[(which * 32 + r) * ST_COUT + c];      // fixed order: deterministic
            p.stats[((size_t)blockIdx.x * 2 + which) * ST_COUT + c] = s;
        }
    }
}

// w [64][7][7][3] f32 (channels_last [64, 3, 7, 7]) -> wp [64][176] bf16: k = r * 24 + (s + 1) * 3 + c, zeros elsewhere
__global__ void __launch_bounds__(DIR_TPB)
stem_prep_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp) {
    const int i = blockIdx.x * DIR_TPB + threadIdx.x;
    if (i >= ST_COUT * ST_KP) return;
    const int co = i / ST_KP, k = i - co * ST_KP;
    const int r = k / 24, tt = k - 24 * r;
    float v = 0.0f;
    if (r < ST_R && tt >= 3) { const int s = tt / 3 - 1, c = tt - 3 * (tt / 3); v = w[((co * ST_R + r) * ST_R + s) * ST_CIN + c]; }
    const f32x2 pr = {v, 0.0f};
    wp[i] = (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2)) & 0xffffu);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the stem: dW[co][r][s][c] = sum over (n, ho, wo) of dY[n, ho, wo, co] * X[n, 2 ho - 3 + r, 2 wo - 3 + s, c].
// Per output row the contraction index is wo, and with the 24-element windows of the forward kernel the gradient of
// filter row r is D[(r, t)][co] = sum_wo in_r[6 wo + 4 + t] * dY[wo][co]: a GEMM with M = 7 x 24 = 168 window positions
// (6 MFMA tiles of 32), N = 64 channels, K = 128 pixels per row. The dY row is transposed on the way into LDS (4 pixels x 8
// channels per thread, v_perm_b32, 8-byte stores) so that its MFMA operand is one ds_read_b128; the image operand has
// a 12-byte stride between consecutive pixels and is gathered with eight 16-bit LDS reads per fragment. Workgroups are
// persistent over output rows and keep their accumulators in registers; each writes ONE partial [168][64] float tile,
// summed in workgroup order by a second kernel that also drops the zero-weight window column and emits
// [64][7][7][3]: deterministic, no atomics.
constexpr int SW_DROWB = 272;                      // bytes per channel row of the transposed dY tile: 128 pixels x 2 B + pad
constexpr int SW_D_BYTES = ST_COUT * SW_DROWB;     // 17 408
constexpr int SW_LDS = ST_IN_BYTES + SW_D_BYTES;   // 28 608 B
constexpr int SW_M = ST_R * 24;                    // 168 window positions
constexpr int SW_MAX_BLOCKS = 1024;

struct StemWgP { const uint16_t* dy; const uint16_t* x; float* part; int N, H, W, Ho, Wo, rows; };

__global__ void __launch_bounds__(DIR_TPB)
stem_wgrad_kernel(StemWgP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* in = smem;
    unsigned char* dt = smem + ST_IN_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, frow = lane & 31, fhalf = lane >> 5;
    for (int i = t; i < ST_IN_BYTES / 16; i += DIR_TPB) *reinterpret_cast<u32x4*>(in + i * 16) = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();
    const int chunks = p.W * ST_CIN * 2 / 16;
    const int pg = t >> 3, cg = t & 7;                              // dY staging: pixels 4 pg .. + 3, channels 8 cg .. + 7
    // M tiles of this wavefront: waves 0, 1 own tiles (w, w + 4), waves 2, 3 own tile w alone (6 tiles of 32 rows)
    const int nmt = wave < 2 ? 2 : 1;
    int abase[2];                                                   // byte offset of this lane's A row: filter row + window element
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = (wave + 4 * j) * 32 + frow;
        const int r = m / 24 > ST_R - 1 ? ST_R - 1 : m / 24, tt = m - 24 * (m / 24);     // rows >= 168: junk, never stored
        abase[j] = r * ST_ROWB + (4 + tt) * 2;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][ni][e] = 0.0f;

    for (int row = blockIdx.x; row < p.rows; row += gridDim.x) {
        const int n = row / p.Ho, ho = row - n * p.Ho;
        for (int i = t; i < ST_R * chunks; i += DIR_TPB) {          // the 7 input rows, as in the forward kernel
            const int r = i / chunks, j = i - r * chunks;
            const int hi = 2 * ho - 3 + r;
            u32x4 v = {0u, 0u, 0u, 0u};
            if ((unsigned)hi < (unsigned)p.H) v = *reinterpret_cast<const u32x4*>(p.x + ((size_t)(n * p.H + hi) * p.W) * ST_CIN + j * 8);
            *reinterpret_cast<u32x4*>(in + r * ST_ROWB + 32 + j * 16) = v;
        }
        {                                                           // dY row -> [channel][pixel] (4 x 8 transposes in registers)
            const uint16_t* src = p.dy + ((size_t)row * p.Wo + 4 * pg) * ST_COUT + cg * 8;
            u32x4 r0 = {0u, 0u, 0u, 0u}, r1 = r0, r2 = r0, r3 = r0;
            if (4 * pg + 0 < p.Wo) r0 = *reinterpret_cast<const u32x4*>(src);
            if (4 * pg + 1 < p.Wo) r1 = *reinterpret_cast<const u32x4*>(src + ST_COUT);
            if (4 * pg + 2 < p.Wo) r2 = *reinterpret_cast<const u32x4*>(src + 2 * ST_COUT);
            if (4 * pg + 3 < p.Wo) r3 = *reinterpret_cast<const u32x4*>(src + 3 * ST_COUT);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // v_perm_b32(src0 = high dword, src1 = low dword): 0x05040100 -> lo16(src1) | lo16(src0) << 16
                const uint32_t lo01 = __builtin_amdgcn_perm(r1[q], r0[q], 0x05040100u), lo23 = __builtin_amdgcn_perm(r3[q], r2[q], 0x05040100u);
                const uint32_t hi01 = __builtin_amdgcn_perm(r1[q], r0[q], 0x07060302u), hi23 = __builtin_amdgcn_perm(r3[q], r2[q], 0x07060302u);
                *reinterpret_cast<uint2*>(dt + (cg * 8 + 2 * q) * SW_DROWB + pg * 8) = make_uint2(lo01, lo23);
                *reinterpret_cast<uint2*>(dt + (cg * 8 + 2 * q + 1) * SW_DROWB + pg * 8) = make_uint2(hi01, hi23);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {                            // 16 pixels per step: this lane's 8 are wo0 .. wo0 + 7
            const int wo0 = ks * 16 + 8 * fhalf;
            bf16x8 b[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(dt + (ni * 32 + frow) * SW_DROWB + wo0 * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j < nmt) {
                    const unsigned char* ap = in + abase[j] + 12 * wo0;
                    u32x4 av;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        av[q] = (uint32_t)*reinterpret_cast<const uint16_t*>(ap + 24 * q) |
                                ((uint32_t)*reinterpret_cast<const uint16_t*>(ap + 24 * q + 12) << 16);
                    const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[j][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[ni], acc[j][ni], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                            // everyone is done with `in` / `dt` before the next row lands
    }

    // partial tile of this workgroup: part[block][168][64]; C/D: col = lane & 31 (channel), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    float* out = p.part + (size_t)blockIdx.x * SW_M * ST_COUT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (j < nmt)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = (wave + 4 * j) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
                    if (m < SW_M) out[m * ST_COUT + ni * 32 + frow] = acc[j][ni][e];
                }
}

// dW[co][r][s][c] = sum over workgroups of part[b][r * 24 + (s + 1) * 3 + c][co]. One workgroup per filter element k =
// (r, s, c): 64 channels x 4 lanes, lane l adds workgroups l, l + 4, ... in four interleaved chains (independent loads in
// flight), the lane sums are combined in lane order: a fixed summation tree, bit-reproducible.
__global__ void __launch_bounds__(DIR_TPB)
stem_wgrad_reduce_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ dw) {
    __shared__ float sh[4][ST_COUT];
    const int k = blockIdx.x, co = threadIdx.x & (ST_COUT - 1), l = threadIdx.x / ST_COUT;
    const int r = k / 21, sc = k - 21 * r;                          // sc = s * 3 + c
    const int m = r * 24 + 3 + sc;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = l;
    for (; b + 12 < nblocks; b += 16) {
        s0 += part[((size_t)(b + 0) * SW_M + m) * ST_COUT + co];
        s1 += part[((size_t)(b + 4) * SW_M + m) * ST_COUT + co];
        s2 += part[((size_t)(b + 8) * SW_M + m) * ST_COUT + co];
        s3 += part[((size_t)(b + 12) * SW_M + m) * ST_COUT + co];
    }
    for (; b < nblocks; b += 4) s0 += part[((size_t)b * SW_M + m) * ST_COUT + co];
    sh[l][co] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (l == 0) dw[(size_t)co * (ST_R * ST_R * ST_CIN) + k] = (sh[0][co] + sh[1][co]) + (sh[2][co] + sh[3][co]);
}

int stem_wgrad_grid(int N, int Ho) {
    const long long rows = (long long)N * Ho;
    return (int)(rows < SW_MAX_BLOCKS ? rows : SW_MAX_BLOCKS);
}

int stem_grid(int N, int Ho) {
    const long long rows = (long long)N * Ho;
    return (int)(rows < ST_MAX_BLOCKS ? rows : ST_MAX_BLOCKS);
}

}  // namespace

extern "C" size_t dir_stem_conv_stats_rows(int N, int H) {
    if (N <= 0 || H <= 0) return 0;
    return (size_t)stem_grid(N, (H + 6 - 7) / 2 + 1);
}

extern "C" int dir_stem_conv_prep_weights(const float* w, void* wpack, dir_stream_t stream) {
    DIR_RETURN_IF(!w || !wpack, DIR_EINVAL);
    hipLaunchKernelGGL(stem_prep_weights_kernel, dim3(dir_cdiv(ST_COUT * ST_KP, DIR_TPB)), dim3(DIR_TPB), 0, dir_s(stream), w,
                       static_cast<uint16_t*>(wpack));
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_stem_conv_fwd(const void* x, const void* wpack, void* y, float* stats, int N, int H, int W,
                                 dir_stream_t stream) {
    DIR_RETURN_IF(!x || !wpack || !y || N <= 0 || H <= 0 || W <= 0, DIR_EINVAL);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(wpack) || !dir_aligned16(y), DIR_EINVAL);
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    // one workgroup per output row: Wo <= 128 pixels; 16-B aligned input rows: W % 8 == 0; the row image holds W <= 256
    DIR_RETURN_IF(Wo > 128 || W % 8 != 0 || 16 + 3 * W > ST_ROWE, DIR_EUNSUPPORTED);
    DIR_RETURN_IF((long long)N * H * W * ST_CIN >= (1ll << 31) || (long long)N * Ho * Wo * ST_COUT >= (1ll << 31), DIR_EUNSUPPORTED);
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS), true);
    (void)once;
    StemP p;
    p.x = static_cast<const uint16_t*>(x); p.wp = static_cast<const uint16_t*>(wpack); p.y = static_cast<uint16_t*>(y);
    p.stats = stats; p.N = N; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.rows = N * Ho;
    hipLaunchKernelGGL(stem_conv_kernel, dim3(stem_grid(N, Ho)), dim3(DIR_TPB), ST_LDS, dir_s(stream), p);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" size_t dir_stem_conv_wgrad_workspace(int N, int H) {
    if (N <= 0 || H <= 0) return 0;
    return dir_align_up(sizeof(float) * (size_t)stem_wgrad_grid(N, (H + 6 - 7) / 2 + 1) * SW_M * ST_COUT, 256);
}

extern "C" int dir_stem_conv_wgrad(const void* dy, const void* x, float* dw, int N, int H, int W, void* workspace,
                                   size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !x || !dw || !workspace || N <= 0 || H <= 0 || W <= 0, DIR_EINVAL);
    DIR_RETURN_IF(!dir_aligned16(dy) || !dir_aligned16(x) || !dir_aligned16(dw) || !dir_aligned16(workspace), DIR_EINVAL);
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    DIR_RETURN_IF(Wo > 128 || W % 8 != 0 || 16 + 3 * W > ST_ROWE, DIR_EUNSUPPORTED);
    DIR_RETURN_IF((long long)N * H * W * ST_CIN >= (1ll << 31) || (long long)N * Ho * Wo * ST_COUT >= (1ll << 31), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(workspace_bytes < dir_stem_conv_wgrad_workspace(N, H), DIR_EWORKSPACE);
    StemWgP p;
    p.dy = static_cast<const uint16_t*>(dy); p.x = static_cast<const uint16_t*>(x); p.part = static_cast<float*>(workspace);
    p.N = N; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.rows = N * Ho;
    const int grid = stem_wgrad_grid(N, Ho);
    hipStream_t s = dir_s(stream);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(DIR_TPB), SW_LDS, s, p);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(ST_R * ST_R * ST_CIN), dim3(DIR_TPB), 0, s, p.part, grid, dw);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

