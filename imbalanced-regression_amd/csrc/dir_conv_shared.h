// Shared pieces of the MFMA implicit-GEMM convolution kernels (dir_conv.hip): launch parameters, tile constants, bf16 packing, accumulator staging, inline-asm LDS-DMA.
#pragma once
#include "dir_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
struct ConvP {
    const uint16_t* x; const uint16_t* w; uint16_t* y; float* stats;
    const uint16_t* addend;   // optional [M][Cout] bf16 added to the rounded result (fused gradient accumulation)
    const uint16_t* mask;     // optional [M][Cout] bf16: result zeroed where !(mask > 0) (fused ReLU backward)
    const uint16_t* addend2;  // optional COMPACT addend [N][Ho/2][Wo/2][Cout] bf16, added at the even (ho, wo) only: the data
                              // gradient of a 1x1 stride-2 convolution of the same input, never scattered to full size
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
    int M, KT, cpk, ntn, nblocks;
    int simple;               // 1x1, stride 1, pad 0: row m of the GEMM is row m of x (no index arithmetic at all)
    float inv_wo, inv_ho;     // reciprocals for the (n, ho, wo) decode of the general case
    int o2, o_a, o_b, OH, OW; // o2: output row (n, i, j) is stored at pixel (2 i + o_a, 2 j + o_b) of an [N][OH][OW][Cout] tensor
                              // (one parity class of a stride-2 data gradient); else rows are stored densely
    int nbuf;                 // LDS stages of the K loop: 2 = prefetched tile written while the current one is read, 1 = extra barrier
    // BatchNorm-backward reduction fused into a data-gradient store loop: this launch's result is the gradient of the OUTPUT of a
    // BatchNorm whose input is bnx (same [rows][Cout] geometry as y); `stats` then receives the per-tile partials (sum g, sum g*bnx)
    // of dir_bn_bwd's first pass. With bn_gamma: that BatchNorm is followed by a ReLU whose mask (bnx * a + b > 0, the forward's own
    // decision) is applied to g for the sums only (the stored gradient stays unmasked: the BatchNorm's apply pass masks it again).
    const uint8_t* mask_bits; // the same ReLU mask as one bit per element ([M][Cout / 8] bytes, dir_bn_*_bits), instead of `mask`
    const uint16_t* bnx;
    const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_rstd;
};
struct ConvBn { const void* x; const float* gamma; const float* beta; const float* mean; const float* rstd; const void* mask_bits; };

constexpr int CV_BM = 128, CV_BK = 64, CV_ROWB = CV_BK * 2;      // 128-byte LDS rows
constexpr int CV_DMA_MIN_KT = 32;                                // shortest K loop (64-wide steps) that takes the LDS-DMA variant: measured +3...+20 % from 32 steps up, mixed at 16, slower below (profiles/r02_conv_variants.txt)
constexpr int CV_OOB = (int)0x80000000;                         // buffer-load offset beyond any tensor: the load returns zeros

__device__ __forceinline__ uint32_t cv_f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t cv_u32x4;
// Cache policy `sc1 nt` of a buffer store (aux bits of the raw buffer intrinsics on gfx940+: 1 = sc0, 2 = nt, 16 = sc1). Kept as a
// compiler-visible intrinsic: an inline-asm store hides its 128-bit data registers from the hazard recognizer (overwritten one
// instruction later -> corrupted rows, found as NaNs in a BatchNorm's running variance).
constexpr int CV_AUX_SC1_NT = 18;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// two floats -> packed bf16x2 (round to nearest even, quiet NaN): one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t cv_pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// Accumulators -> bf16 -> LDS staging tile [128 pixels][CS_STRIDE]. The K loops feed the MFMA with the operands SWAPPED
// (weights as the "A" matrix, pixels as "B"), so an accumulator tile is D'[channel][pixel]: lane l holds pixel (l & 31) and, in
// registers 4 q .. 4 q + 3, the four CONSECUTIVE channels 8 q + 4 (l >> 5) + {0..3}. Two v_cvt_pk_bf16_f32 make them one 8-byte
// ds_write_b64: 16 LDS stores per lane for a 64 x 64 wavefront tile instead of the 128 ds_write_b16 of the pixel-major
// orientation, whose 2048 LDS cycles per workgroup tile (= four K-steps of MFMA time) were the whole cost of the 1-4 step
// K loops of the 1x1 layers. Row stride 272 / 144 B: consecutive pixels shift 4 banks -> 2-way on the write, rows stay 16-B
// aligned for the row reads of the store loop.
template <int MI, int NI, int CS_STRIDE>
__device__ __forceinline__ void cv_stage_acc(const f32x16 (&acc)[MI][NI], unsigned char* cbase) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t lo = cv_pack_bf16(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1]);
                const uint32_t hi = cv_pack_bf16(acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
                *reinterpret_cast<uint2*>(cbase + mi * 32 * CS_STRIDE + (ni * 32 + 8 * q) * 2) = make_uint2(lo, hi);
            }
}


typedef __attribute__((ext_vector_type(4))) uint32_t cp_u32x4;
__device__ __forceinline__ cp_u32x4 cp_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    cp_u32x4 r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return r;
}
__device__ __forceinline__ void cp_dma16(cp_u32x4 rs, uint32_t lds_addr, int voffset, int soffset) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voffset), "s"(rs), "s"(soffset) : "memory");
}
__device__ __forceinline__ void cp_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
