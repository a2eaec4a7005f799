// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions (conv2 of every Bottleneck, imdb-wiki-dir/resnet.py:46-47), all
// nine filter taps in ONE pass over dY and X (gfx950).
//
//   dW[co, r, s, ci] = sum_{n, i, j} dY[n, i, j, co] * X[n, i + r - 1, j + s - 1, ci]
//
// The per-tap kernel of dir_conv_wgrad.hip runs one workgroup per (tap, tile, K range): every tap re-reads its dY rows and a
// shifted copy of the X rows, so the L2 -> LDS traffic of a layer is 9 x (|dY| + |X|) (1.85 GB for 64 -> 64 at 56 x 56, batch
// 256) and that traffic, not the 59 GFLOP, is what its ~100-200 us are made of. Here a workgroup owns a 64 x 64 (co x ci) block
// of ALL nine taps (nine 32 x 32 accumulator tiles per wavefront) and walks over pixel chunks:
//   chunk   = RB whole output rows of one image (RB * W <= 112 pixels, padded to a multiple of 16 with zero dY rows);
//   LDS     = the chunk's dY rows [pixel][64 co] and its X patch [(RB + 2) x P pixels][64 ci] with the zero border of the
//             padding materialised (P = W + 2 rounded up to a multiple of 4), both in their NATURAL layout (channels contiguous),
//             filled by LDS-DMA (buffer_load ... lds: no staging registers, no transposing stores); two stages, the next chunk
//             in flight while the current one is multiplied;
//   operands = ds_read_b64_tr_b16: the hardware-transposing LDS read hands each lane the 4 consecutive PIXELS (the K axis of
//             this GEMM) of its channel from four 128-byte pixel rows, so a filter tap is nothing but a row offset of the X
//             reads: (r * P + s) * 128 bytes, an instruction immediate for r and three precomputed addresses for s;
//   MFMA    = v_mfma_f32_32x32x16_bf16, per 16 pixels: 2 dY reads + 18 X reads + 9 MFMAs per wavefront.
// 512 threads = 8 wavefronts: (co half, ci half) x (even / odd 16-pixel steps); the two K halves are added through LDS at the
// end, so a workgroup emits ONE [64][9][64] float32 partial. Split-K over chunks, partials summed in split order by
// conv_wgrad_reduce_kernel (deterministic, no atomics).
// LDS image: 128-byte pixel rows; the two 64-byte halves of a row are swapped when bit 1 of the row index is set, so the four
// rows of a transposing read fall into four different 16-bank groups (applied on the DMA source side: the DMA destination is
// lane-linear).
#include "dir_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 w3_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 w3_bf16x4;
typedef __attribute__((ext_vector_type(16))) float w3_f32x16;

struct Wg3P {
    const uint16_t* dy; const uint16_t* x; float* part;
    int N, Cin, Cout;
    int nco, nci, nsplit, units, ups;      // units = N * chunks-per-image, ups = units per split
};

constexpr int W3_TPB = 512;
constexpr int W3_OOB = (int)0x80000000;

template <int WI> struct W3Geom {
    static constexpr int H = WI;
    static constexpr int RB = (WI == 56) ? 2 : (WI == 28) ? 4 : 7;          // output rows per chunk
    static constexpr int CPI = H / RB;                                       // chunks per image
    static constexpr int P = (WI == 56) ? 60 : (WI == 28) ? 32 : (WI == 14) ? 16 : 12;   // patch row pitch in pixels, % 4 == 0
    static constexpr int KPIX = RB * WI;                                     // real pixels per chunk
    static constexpr int NK = (KPIX + 15) / 16;                              // 16-pixel MFMA steps
    static constexpr int NKH = (NK + 1) / 2;                                 // ... per K-parity wavefront
    static constexpr int DY_PIECES = NK * 2;                                 // 1 KB DMA pieces (8 pixel rows of 128 B)
    static constexpr int XP = (RB + 2) * P;                                  // patch slots
    static constexpr int X_PIECES = (XP + 7) / 8;
    static constexpr int PIECES = DY_PIECES + X_PIECES;
    static constexpr int PPW = (PIECES + 7) / 8;                             // pieces per wavefront
    static constexpr int DY_BYTES = DY_PIECES * 1024;
    static constexpr int STAGE = PIECES * 1024;
    static_assert(H % RB == 0 && P % 4 == 0 && P >= WI + 2, "chunk geometry");
};
constexpr int W3_RED_BYTES = 4 * 144 * 64 * 4;                               // K-parity reduction: 4 wavefronts x 144 floats x 64 lanes

typedef __attribute__((ext_vector_type(4))) uint32_t w3_u32x4;
// Raw buffer descriptor (stride 0, 32-bit offsets, out-of-range reads return zero): the words __builtin_amdgcn_make_buffer_rsrc builds
__device__ __forceinline__ w3_u32x4 w3_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    w3_u32x4 r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return r;
}
// One LDS-DMA piece: 64 lanes x 16 B from (descriptor, per-lane byte offset) to the 1 KB of LDS at byte address lds_addr
// (wave-uniform; becomes M0), lane l landing at lds_addr + 16 l; out-of-range lanes write zeros. Issued as inline assembly ON
// PURPOSE: hipcc orders every LDS read behind a pending LDS-DMA it knows about (s_waitcnt vmcnt(0) in front of the transposing
// reads, which have no memory operand to disambiguate), i.e. it would drain the next chunk's loads before the current chunk's
// first MFMA. Hidden from its bookkeeping, the DMA runs under the MFMAs and is waited for explicitly (w3_dma_wait) before the
// barrier that publishes the stage. M0 is saved and restored inside the statement (the compiler owns it).
__device__ __forceinline__ void w3_dma16(w3_u32x4 rs, uint32_t lds_addr, int voffset) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voffset), "s"(rs) : "memory");
}
__device__ __forceinline__ void w3_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ds_read_b64_tr_b16: within each 16-lane group, lane i supplies the address of 4 consecutive bf16 of matrix row (i >> 2),
// columns 4 (i & 3) .. + 3, and receives column i of the 4 x 16 block: rows 0 .. 3 (tests/test_hip_conv_wgrad3.py pins this)
__device__ __forceinline__ w3_bf16x4 w3_tr(const unsigned char* p) {
    typedef __attribute__((address_space(3))) w3_bf16x4* lds_v4_t;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)p);
}
__device__ __forceinline__ w3_bf16x8 w3_cat(w3_bf16x4 a, w3_bf16x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int WI>
__global__ void __launch_bounds__(W3_TPB) __attribute__((amdgpu_waves_per_eu(2)))
conv_wgrad3_kernel(Wg3P p) {
    using G = W3Geom<WI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kpar = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int hf = lane >> 5;

    int b = blockIdx.x;
    const int cib = b % p.nci; b /= p.nci;
    const int cob = b % p.nco; b /= p.nco;
    const int split = b;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int u0 = split * p.ups;
    int u1 = u0 + p.ups; if (u1 > p.units) u1 = p.units;

    const w3_u32x4 rs_dy = w3_rsrc(p.dy, (uint32_t)(p.N * G::H * WI) * (uint32_t)p.Cout * 2u);
    const w3_u32x4 rs_x = w3_rsrc(p.x, (uint32_t)(p.N * G::H * WI) * (uint32_t)p.Cin * 2u);
    typedef __attribute__((address_space(3))) unsigned char* w3_lds_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(w3_lds_t)smem;            // LDS byte address of the dynamic region

    // ---- DMA roles: piece q = wave + 8 i covers LDS bytes [q KB, (q + 1) KB) of a stage = 8 pixel rows; the lane writes
    // physical 16-byte chunk (lane & 7) of row (lane >> 3), which holds LOGICAL chunk (lane & 7) ^ 4 * ((row >> 1) & 1)
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((lane >> 4) & 1) << 2);
    int rel[G::PPW];                    // byte offset relative to the chunk's base (dY: its first pixel; X: pixel (h0 - 1, -1)), or OOB
    int edge[G::PPW];                   // X pieces: 1 = top halo row, 2 = bottom halo row (outside the image for the first / last chunk)
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) {
        const int q = wave + 8 * i;
        rel[i] = W3_OOB; edge[i] = 0;
        if (q < G::DY_PIECES) {
            const int k = q * 8 + lrow;
            if (k < G::KPIX) rel[i] = (k * p.Cout + lchunk * 8) * 2;
        } else if (q < G::PIECES) {
            const int slot = (q - G::DY_PIECES) * 8 + lrow;
            const int pr = slot / G::P, pc = slot - pr * G::P;
            if (slot < G::XP && pc >= 1 && pc <= WI) {
                rel[i] = ((pr * WI + pc) * p.Cin + lchunk * 8) * 2;
                edge[i] = pr == 0 ? 1 : (pr == G::RB + 1 ? 2 : 0);
            }
        }
    }

    // ---- operand read addresses (bytes inside a stage)
    const int r4 = (lane >> 2) & 3;                              // the pixel row of a 4 x 16 block this lane addresses
    const uint32_t lane_c = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    // dY: pixel k = 16 kk + 8 hf + 4 j + r4, channels of co half wm; (k >> 1) & 1 == (r4 >> 1)
    const uint32_t dyl = (uint32_t)((8 * hf + r4) * 128 + ((wm ^ (r4 >> 1)) << 6)) + lane_c + (uint32_t)kpar * 2048u;
    // X: patch slot of pixel k at tap (0, 0) (the patch starts one row above and one column left of the chunk)
    int pp0[G::NKH][2];
#pragma unroll
    for (int tk = 0; tk < G::NKH; ++tk)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = 16 * (kpar + 2 * tk) + 8 * hf + 4 * j + r4;
            const int i = k / WI, w = k - i * WI;
            pp0[tk][j] = k < G::KPIX ? i * G::P + w : 0;         // padded K rows: dY is zero there, any finite X will do
        }
    const uint32_t xl = (uint32_t)G::DY_BYTES + lane_c;

    w3_f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.0f;

#define W3_ISSUE(u, stage)                                                                                      \
    {                                                                                                           \
        const int n_ = (u) / G::CPI, c_ = (u) - n_ * G::CPI, h0_ = c_ * G::RB;                                  \
        const int dyb = ((n_ * G::H + h0_) * WI * p.Cout + co0) * 2;                                            \
        const int xb = (((n_ * G::H + h0_ - 1) * WI - 1) * p.Cin + ci0) * 2;                                    \
        const int dead = (h0_ == 0 ? 1 : 0) | (h0_ + G::RB == G::H ? 2 : 0);                                    \
        _Pragma("unroll")                                                                                       \
        for (int i = 0; i < G::PPW; ++i) {                                                                      \
            const int q = wave + 8 * i;                                                                         \
            if (q < G::PIECES) {                                                                                \
                const uint32_t dst = lds0 + (uint32_t)((stage) * G::STAGE + q * 1024);                          \
                if (q < G::DY_PIECES) w3_dma16(rs_dy, dst, rel[i] == W3_OOB ? W3_OOB : dyb + rel[i]);           \
                else w3_dma16(rs_x, dst, (rel[i] == W3_OOB || (edge[i] & dead)) ? W3_OOB : xb + rel[i]);        \
            }                                                                                                   \
        }                                                                                                       \
    }

    if (u0 < u1) {
        W3_ISSUE(u0, 0);
        w3_dma_wait();
        __syncthreads();
        for (int u = u0; u < u1; ++u) {
            const int stage = (u - u0) & 1;
            if (u + 1 < u1) W3_ISSUE(u + 1, stage ^ 1);          // next chunk in flight during the MFMAs
            const unsigned char* sb = smem + stage * G::STAGE;
#pragma unroll
            for (int tk = 0; tk < G::NKH; ++tk) {
                if (kpar + 2 * tk < G::NK) {
                    const w3_bf16x8 a = w3_cat(w3_tr(sb + dyl + tk * 4096), w3_tr(sb + dyl + tk * 4096 + 512));
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const uint32_t q0 = (uint32_t)(pp0[tk][0] + s), q1 = (uint32_t)(pp0[tk][1] + s);
                        const unsigned char* x0 = sb + xl + (q0 << 7) + ((((q0 >> 1) & 1u) ^ (uint32_t)wn) << 6);
                        const unsigned char* x1 = sb + xl + (q1 << 7) + ((((q1 >> 1) & 1u) ^ (uint32_t)wn) << 6);
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const w3_bf16x8 bx = w3_cat(w3_tr(x0 + r * G::P * 128), w3_tr(x1 + r * G::P * 128));
                            acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bx, acc[r * 3 + s], 0, 0, 0);
                        }
                    }
                }
            }
            w3_dma_wait();                                       // this wavefront's pieces of the next chunk have landed ...
            __syncthreads();                                     // ... everyone's have, and everyone is done reading `stage`
        }
    }
#undef W3_ISSUE

    // ---- the odd-step wavefronts hand their sums to the even-step ones through LDS (the stages are free: barrier above)
    float* red = reinterpret_cast<float*>(smem);
    const int q2 = wave & 3;
    if (kpar == 1) {
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[((q2 * 144) + a * 16 + e) * 64 + lane] = acc[a][e];
    }
    __syncthreads();
    if (kpar == 0) {
        // partial[split][co][tap][ci] (fp32). C/D: col = lane & 31 -> ci, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) -> co
        const int rsc = 9 * p.Cin;
        float* out = p.part + ((size_t)split * p.Cout + co0 + wm * 32) * rsc + ci0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = acc[a][e] + red[((q2 * 144) + a * 16 + e) * 64 + lane];
                out[(size_t)((e & 3) + 8 * (e >> 2) + 4 * hf) * rsc + a * p.Cin] = v;
            }
    }
}

struct W3Plan { int nco, nci, nsplit, units, ups; size_t ws_bytes; };

bool w3_shape_ok(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
    if (R != 3 || S != 3 || stride != 1 || pad != 1 || H != W) return false;
    if (W != 56 && W != 28 && W != 14 && W != 7) return false;
    if (N <= 0 || Cin % 64 || Cout % 64) return false;
    const long long M = (long long)N * H * W;
    return M * Cin < (1ll << 30) && M * Cout < (1ll << 30);                 // 32-bit byte offsets
}

W3Plan w3_plan(int N, int W, int Cin, int Cout) {
    W3Plan pl;
    pl.nco = Cout / 64; pl.nci = Cin / 64;
    const int cpi = W == 56 ? 28 : W == 28 ? 7 : W == 14 ? 2 : 1;
    pl.units = N * cpi;
    // one 8-wavefront workgroup per CU (147 KB of LDS for the final reduction): 256 partials of [64][9][64] floats per layer
    int nsplit = 256 / (pl.nco * pl.nci);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > pl.units) nsplit = pl.units;
    pl.ups = (pl.units + nsplit - 1) / nsplit;
    pl.nsplit = (pl.units + pl.ups - 1) / pl.ups;
    pl.ws_bytes = dir_align_up(sizeof(float) * (size_t)pl.nsplit * Cout * 9 * Cin, 256);
    return pl;
}

template <int WI>
void w3_launch(const Wg3P& p, hipStream_t s) {
    using G = W3Geom<WI>;
    constexpr int lds = 2 * G::STAGE > W3_RED_BYTES ? 2 * G::STAGE : W3_RED_BYTES;
    DIR_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad3_kernel<WI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((conv_wgrad3_kernel<WI>), dim3(p.nco * p.nci * p.nsplit), dim3(W3_TPB), lds, s, p);
}

}  // namespace

// (dir_conv_wgrad.hip) sums the split partials in split order
extern "C" int dir_conv_wgrad_reduce_splits(const float* part, int splits, size_t n, float* dw, dir_stream_t stream);

extern "C" size_t dir_conv_wgrad3x3_workspace(int N, int H, int W, int Cin, int Cout) {
    if (!w3_shape_ok(N, H, W, Cin, Cout, 3, 3, 1, 1)) return 0;
    return w3_plan(N, W, Cin, Cout).ws_bytes;
}

static int wgrad3_impl(const void* dy, const void* x, float* dw, int* splits_out, int N, int H, int W, int Cin, int Cout,
                       void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !x || !workspace, DIR_EINVAL);
    DIR_RETURN_IF(!w3_shape_ok(N, H, W, Cin, Cout, 3, 3, 1, 1), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(dy) || !dir_aligned16(x) || (dw && !dir_aligned16(dw)) || (reinterpret_cast<uintptr_t>(workspace) & 255u), DIR_EINVAL);
    const W3Plan pl = w3_plan(N, W, Cin, Cout);
    DIR_RETURN_IF(workspace_bytes < pl.ws_bytes, DIR_EWORKSPACE);
    Wg3P p;
    p.dy = static_cast<const uint16_t*>(dy); p.x = static_cast<const uint16_t*>(x); p.part = static_cast<float*>(workspace);
    p.N = N; p.Cin = Cin; p.Cout = Cout;
    p.nco = pl.nco; p.nci = pl.nci; p.nsplit = pl.nsplit; p.units = pl.units; p.ups = pl.ups;
    hipStream_t s = dir_s(stream);
    if (W == 56) w3_launch<56>(p, s);
    else if (W == 28) w3_launch<28>(p, s);
    else if (W == 14) w3_launch<14>(p, s);
    else w3_launch<7>(p, s);
    DIR_LAUNCH_CHECK();
    if (splits_out) *splits_out = pl.nsplit;
    if (!dw) return DIR_OK;                                              // partials [nsplit][Cout * 9 * Cin] stay in the workspace
    return dir_conv_wgrad_reduce_splits(p.part, pl.nsplit, (size_t)Cout * 9 * Cin, dw, stream);
}

extern "C" int dir_conv_wgrad3x3(const void* dy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout,
                                 void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!dw, DIR_EINVAL);
    return wgrad3_impl(dy, x, dw, nullptr, N, H, W, Cin, Cout, workspace, workspace_bytes, stream);
}

extern "C" int dir_conv_wgrad3x3_partials(const void* dy, const void* x, int* splits, int N, int H, int W, int Cin, int Cout,
                                          void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!splits, DIR_EINVAL);
    return wgrad3_impl(dy, x, nullptr, splits, N, H, W, Cin, Cout, workspace, workspace_bytes, stream);
}
