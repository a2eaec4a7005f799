// MFMA implicit-GEMM convolution for NHWC bf16 activations on MI355X / gfx950 (CDNA4).
//
// Replaces the nn.Conv2d layers of imdb-wiki-dir/resnet.py:41-70,79,112-116 (cuDNN in the reference): the only
// genuinely dense contraction of the hot path, so the only place MFMA is used.
//
//   Y[m, co] = sum_{r,s,ci} X[n, ho*stride - pad + r, wo*stride - pad + s, ci] * Wt[co, r, s, ci],   m = (n, ho, wo)
//
// GEMM view: M = N*Ho*Wo rows, N = Cout columns, K = R*S*Cin with Cin % 64 == 0, so one 64-wide K-step never
// straddles a filter tap and the A-operand loader is "per-row byte offset + wave-uniform tap offset" into a raw buffer
// load; taps outside the image are sent out of the buffer's range, where the hardware returns zeros.
// Both operands are K-contiguous in memory (NHWC activations, [Cout][R][S][Cin] weights = torch channels_last), which
// is exactly the 16-bytes-per-lane fragment of v_mfma_f32_32x32x16_bf16 — no transposes anywhere.
//
// Workgroup = 256 threads = 4 wavefronts (one per SIMD), tile 128 x BN x 64 (BN = 128 or 64):
//   global -> registers (buffer_load_dwordx4, 16 B per lane, next K-step(s) prefetched while the current one is multiplied)
//   registers -> LDS, rows of 128 B, 16-B chunks XOR-swizzled with (row >> 1) & 7 so that both the 8-lane
//   ds_write_b128 groups and the four 16-lane ds_read_b128 groups are bank-conflict free
//   LDS -> MFMA fragments -> 32x32x16 bf16 MFMA, fp32 accumulators (2x2 or 1x2 tiles of 32x32 per wavefront)
//   epilogue: v_cvt_pk_bf16_f32 + 16-bit LDS stores into a staging tile, optional per-channel (sum, sum of squares)
//   partials of the rounded outputs for the following BatchNorm (no separate statistics pass over Y), 16-B coalesced
//   row stores with optional fused "+ addend", "+ compact stride-2 addend" and ReLU-backward mask (data-gradient use).
// All LDS offsets are computed once and pinned in registers, the LDS stage is a template literal: a K-step is 16 MFMAs,
// 24 LDS and 8 buffer instructions and ~16 VALU instructions.
// Workgroup ids are remapped so that the N-tiles of one M-tile run on the same XCD (A tile re-reads hit that L2).
#include <cstdlib>
#include <type_traits>
#include "dir_common.h"
#include "dir_conv_shared.h"
#include "dir_conv_epilogue.h"

namespace {


// PF = prefetch distance of the global loads in K-steps. The K loop is bound by load latency, not by MFMA or LDS
// throughput: a 32 KB K-tile takes > 1 us to arrive under load while its 16 MFMAs per wavefront take 0.2 us, so the
// rate is (bytes in flight per CU) / latency. PF = 2 keeps two K-tiles per workgroup in flight in two register sets
// (p, q) for 32 more VGPRs (2 instead of 3 wavefronts per SIMD, which the 64 KB two-stage LDS image allows anyway).
template <int BN, int PF, int NBUF, bool LEAN = false>
__global__ void __launch_bounds__(DIR_TPB) __attribute__((amdgpu_waves_per_eu((PF == 2 || (NBUF == 2 && BN == 128)) ? 2 : (LEAN && NBUF == 1 ? 4 : 3))))   // (two 32 KB stages: LDS admits 2 workgroups per CU anyway)
conv_igemm_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_BYTES = CV_BM * CV_ROWB;              // 16 KB
    constexpr int B_BYTES = BN * CV_ROWB;
    constexpr int BROWS = BN / 32;                        // B rows per loader thread
    constexpr int MI = (BN == 128) ? 2 : 1;               // 32x32 tiles per wavefront along M
    constexpr int NI = 2;                                 //                      ... along N
    constexpr int WM = MI * 32;
    constexpr bool dbuf = NBUF == 2;                      // LDS stages: 2 = next tile written while the current one is read
    unsigned char* As = smem;
    unsigned char* Bs = smem + NBUF * A_BYTES;

    // ---- workgroup -> (m tile, n tile), XCD-aware and bijective: the hardware deals workgroup b to XCD b % 8, so the
    // linear tile space is cut into 8 contiguous chunks and the N-tiles of one M-tile run on the same XCD (the A-tile
    // re-reads hit that L2)
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
    const int frow = lane & 31, fhalf = lane >> 5;

    // ---- loader coordinates: thread loads the 16-B chunk (t & 7) of rows (t >> 3) + 32 i.
    // Everything position dependent is computed ONCE: per row the signed byte offset of its (hi0, wi0) pixel and a bit
    // mask of the filter taps that fall inside the image; per K-step a wave-uniform offset is added and out-of-image
    // taps are sent out of the buffer's range (the buffer load then returns zeros: no masking instructions).
    const int lrow = t >> 3, lchunk = t & 7;
    int aoff[4];
    uint32_t amask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lrow + 32 * i;
        aoff[i] = 0; amask[i] = 0;
        if (m < p.M) {
            if (p.simple) { aoff[i] = (m * p.Cin + lchunk * 8) * 2; amask[i] = 1u; continue; }
            // (n, ho, wo) from m: float reciprocal + one correction step instead of integer divisions (exact for m < 2^24)
            int q1 = (int)((float)m * p.inv_wo), wo = m - q1 * p.Wo;
            if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
            int n = (int)((float)q1 * p.inv_ho), ho = q1 - n * p.Ho;
            if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            aoff[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + lchunk * 8) * 2;
            for (int r = 0; r < p.R; ++r)
                for (int s2 = 0; s2 < p.S; ++s2)
                    if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + s2) < (unsigned)p.W) amask[i] |= 1u << (r * p.S + s2);
        }
    }
    const int K = p.KT * CV_BK;
    int woff[BROWS];                                       // byte offset of this thread's chunk in weight row n0 + lrow + 32 i
#pragma unroll
    for (int i = 0; i < BROWS; ++i) woff[i] = ((n0 + lrow + 32 * i) * K + lchunk * 8) * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), (short)0,
                                                                           (int)((unsigned)(p.N * p.H * p.W) * (unsigned)p.Cin * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), (short)0,
                                                                           (int)((unsigned)p.Cout * (unsigned)K * 2u), 0x00020000);
    // LDS byte offsets, pinned in registers: one for the loader's stores (rows + 32 i are immediates; the swizzle of
    // row lrow + 32 i equals that of lrow) and one per fragment read of a K-step
    uint32_t st_off = lrow * CV_ROWB + ((lchunk ^ ((lrow >> 1) & 7)) << 4);
    asm volatile("" : "+v"(st_off));
    // fragment read offsets: row * 128 + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) << 4) = base ^ (kk << 5) with ONE base per 32-row
    // fragment (kk * 2 and fhalf occupy different bits, and the XOR part stays below the 128-byte row pitch): 4 registers
    // instead of 16, one v_xor per read
    uint32_t afb[MI], bfb[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * WM + mi * 32 + frow;
        afb[mi] = row * CV_ROWB + ((fhalf ^ ((row >> 1) & 7)) << 4);
        asm volatile("" : "+v"(afb[mi]));
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * 64 + ni * 32 + frow;
        bfb[ni] = row * CV_ROWB + ((fhalf ^ ((row >> 1) & 7)) << 4);
        asm volatile("" : "+v"(bfb[ni]));
    }

    // K-step cursor (wave-uniform): filter tap and 64-channel block of the NEXT tile to fetch, and its index
    int ld_tap = 0, ld_c = 0, ld_r = 0, ld_s = 0, ld_k = 0;
    // named registers (no private-memory arrays): set p, and set q for PF = 2
    u32x4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;
    pa0 = pa1 = pa2 = pa3 = pb0 = pb1 = pb2 = pb3 = (u32x4){0u, 0u, 0u, 0u};
    qa0 = qa1 = qa2 = qa3 = qb0 = qb1 = qb2 = qb3 = (u32x4){0u, 0u, 0u, 0u};

#define CV_BL(rs, vo, so) __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0)
#define CV_LOAD_TILE(SET)                                                                                       \
    {                                                                                                           \
        const int koff = ((ld_r * p.W + ld_s) * p.Cin + ld_c * CV_BK) * 2;                                      \
        const uint32_t bit = 1u << ld_tap;                                                                      \
        SET##a0 = CV_BL(rs_x, (amask[0] & bit) ? aoff[0] + koff : CV_OOB, 0);                                   \
        SET##a1 = CV_BL(rs_x, (amask[1] & bit) ? aoff[1] + koff : CV_OOB, 0);                                   \
        SET##a2 = CV_BL(rs_x, (amask[2] & bit) ? aoff[2] + koff : CV_OOB, 0);                                   \
        SET##a3 = CV_BL(rs_x, (amask[3] & bit) ? aoff[3] + koff : CV_OOB, 0);                                   \
        const int wso = ld_k * CV_BK * 2;                          /* wave-uniform: the K-step inside the weight rows */ \
        SET##b0 = CV_BL(rs_w, woff[0], wso);                                                                    \
        SET##b1 = CV_BL(rs_w, woff[1], wso);                                                                    \
        if (BROWS == 4) { SET##b2 = CV_BL(rs_w, woff[BROWS - 2], wso); SET##b3 = CV_BL(rs_w, woff[BROWS - 1], wso); } \
        ++ld_k;                                                                                                 \
        if (++ld_c == p.cpk) { ld_c = 0; ++ld_tap; if (++ld_s == p.S) { ld_s = 0; ++ld_r; } }                   \
    }
#define CV_ST(base, bytes, i, v) *reinterpret_cast<u32x4*>((base) + (bytes) + (i) * 32 * CV_ROWB + st_off) = (v)
#define CV_STORE_TILE(buf, SET)                                                                                 \
    {                                                                                                           \
        CV_ST(As, (buf) * A_BYTES, 0, SET##a0); CV_ST(As, (buf) * A_BYTES, 1, SET##a1);                         \
        CV_ST(As, (buf) * A_BYTES, 2, SET##a2); CV_ST(As, (buf) * A_BYTES, 3, SET##a3);                         \
        CV_ST(Bs, (buf) * B_BYTES, 0, SET##b0); CV_ST(Bs, (buf) * B_BYTES, 1, SET##b1);                         \
        if (BROWS == 4) { CV_ST(Bs, (buf) * B_BYTES, 2, SET##b2); CV_ST(Bs, (buf) * B_BYTES, 3, SET##b3); }     \
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    // one K-step of MFMAs on LDS stage `buf` (a literal: the stage folds into the instructions' immediate offsets):
    // 4 x (4 fragment reads, 4 (or 2) MFMA 32x32x16)
#define CV_MFMA_STEP(buf)                                                                                       \
    {                                                                                                           \
        _Pragma("unroll")                                                                                       \
        for (int kk = 0; kk < 4; ++kk) {                                                                        \
            bf16x8 a[MI], b[NI];                                                                                \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(As + (buf) * A_BYTES + (afb[mi] ^ (kk << 5))); \
            _Pragma("unroll")                                                                                   \
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bs + (buf) * B_BYTES + (bfb[ni] ^ (kk << 5))); \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi)                                                                     \
                _Pragma("unroll")                                                                               \
                for (int ni = 0; ni < NI; ++ni)                                                                 \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);  /* D'[channel][pixel] */  \
        }                                                                                                       \
    }

    if (PF == 1 && !dbuf) {
        CV_LOAD_TILE(p);
        CV_STORE_TILE(0, p);
        __syncthreads();
        for (int kt = 0; kt < p.KT; ++kt) {
            const bool more = kt + 1 < p.KT;
            if (more) CV_LOAD_TILE(p);                          // global loads in flight during the MFMAs
            CV_MFMA_STEP(0);
            if (more) {
                __syncthreads();                                // single stage: everyone is done reading before the overwrite
                CV_STORE_TILE(0, p);
            }
            __syncthreads();
        }
    } else if (PF == 1) {
        // two K-steps per trip (stages 0 and 1 are literals), whole pairs only, the odd last step peeled off after the loop: a
        // `break` between the halves gives the loop two exits and the compiler then copies all 64 accumulators every trip
        CV_LOAD_TILE(p);
        CV_STORE_TILE(0, p);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 <= p.KT; kt += 2) {
            CV_LOAD_TILE(p);                                    // tile kt + 1
            CV_MFMA_STEP(0);                                    // tile kt
            CV_STORE_TILE(1, p);
            __syncthreads();
            if (kt + 2 < p.KT) CV_LOAD_TILE(p);                 // tile kt + 2
            CV_MFMA_STEP(1);                                    // tile kt + 1
            if (kt + 2 < p.KT) CV_STORE_TILE(0, p);
            __syncthreads();
        }
        if (kt < p.KT) CV_MFMA_STEP(0);                         // odd K-step count: the last tile sits in stage 0
    } else {
        // PF = 2 (two stages): tiles kt+1 (set q) and kt+2 (set p) are in flight while tile kt is multiplied; a set is
        // re-issued as soon as it has been written to LDS, i.e. two K-steps before it is needed again
        CV_LOAD_TILE(p);                                        // tile 0
        if (p.KT > 1) CV_LOAD_TILE(q);                          // tile 1
        CV_STORE_TILE(0, p);
        if (p.KT > 2) CV_LOAD_TILE(p);                          // tile 2
        __syncthreads();
        int kt = 0;
        for (; kt + 2 <= p.KT; kt += 2) {
            CV_MFMA_STEP(0);                                    // tile kt
            CV_STORE_TILE(1, q);                                // tile kt + 1
            if (kt + 3 < p.KT) CV_LOAD_TILE(q);                 // tile kt + 3
            __syncthreads();
            CV_MFMA_STEP(1);                                    // tile kt + 1
            if (kt + 2 < p.KT) {
                CV_STORE_TILE(0, p);                            // tile kt + 2
                if (kt + 4 < p.KT) CV_LOAD_TILE(p);             // tile kt + 4
            }
            __syncthreads();
        }
        if (kt < p.KT) CV_MFMA_STEP(0);
    }
    __syncthreads();                                            // (every wavefront is done with the K-loop buffers)
#undef CV_MFMA_STEP
#undef CV_LOAD_TILE
#undef CV_STORE_TILE
#undef CV_ST
#undef CV_BL

    cv_epilogue<BN, LEAN>(p, acc, smem, t, m0, n0, mt);
}


// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the K loop: global -> LDS directly (buffer_load_dwordx4 ... lds), no staging registers and no
// ds_write pass. The register-staged loop above spends more LDS-pipe cycles writing a K-tile (32 x ds_write_b128 = ~416
// cycles per workgroup and K-step) than its MFMAs take to issue on a SIMD (16 x 32 = 512) once the 256 cycles of fragment
// reads are added: the LDS pipe, not the matrix pipe, bounds it. A DMA piece is one wave-instruction = 1 KB = 8 LDS rows
// of 128 B: lane l lands at row (l >> 3), physical 16-B chunk (l & 7) of its piece, so the XOR swizzle of the LDS image
// ((row >> 1) & 7, same image as above: the fragment reads are unchanged) is applied to the lane's SOURCE address instead.
// Out-of-image taps still go out of the buffer's range: the hardware writes zeros into LDS for those lanes.
// Two LDS stages, tile kt+1 in flight while tile kt is multiplied, one barrier per K-step; 32 fewer VGPRs than the
// register-staged kernel.
// one DMA piece: 64 lanes x 16 B from (buffer, per-lane voffset + wave-uniform soffset) to the 1 KB of LDS at `lds`
// (wave-uniform; it becomes M0). Out-of-range lanes write zeros.
__device__ __forceinline__ void cv_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voffset, int soffset) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds, 16, voffset, soffset, 0, 0);
}

// NST = LDS stages: 2 = tile kt+1 in flight while tile kt is multiplied (long K loops); 1 = one 32 KB stage, so that with the LEAN
// epilogue (no fused operand) a workgroup needs < 40 KB of LDS and <= 128 registers: FOUR workgroups per CU for the short-K forward
// launches, whose throughput follows the number of resident workgroups (profiles/r02_conv_occupancy_sensitivity.txt).
template <int BN, int NST = 2, bool LEAN = false>
__global__ void __launch_bounds__(DIR_TPB) __attribute__((amdgpu_waves_per_eu(NST == 1 ? 4 : 2)))
conv_igemm_dma_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_BYTES = CV_BM * CV_ROWB;              // 16 KB
    constexpr int B_BYTES = BN * CV_ROWB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int BI = BN / 32;                           // B pieces per wave
    constexpr int MI = (BN == 128) ? 2 : 1;
    constexpr int NI = 2;
    constexpr int WM = MI * 32;
    int lin;
    {
        const int b = blockIdx.x, q = p.nblocks / 8, r = p.nblocks % 8, xcd = b % 8, i = b / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int mt = lin / p.ntn, nt = lin - mt * p.ntn;
    const int m0 = mt * CV_BM, n0 = nt * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // wave-uniform by construction; tell the compiler (LDS base -> M0)
    const int wm = (BN == 128) ? (wave >> 1) : wave;
    const int wn = (BN == 128) ? (wave & 1) : 0;
    const int frow = lane & 31, fhalf = lane >> 5;

    // ---- loader: piece i of this wave = rows wave*32 + 8 i + (lane >> 3) of the A tile (B: wave*8*BI + 8 i + (lane >> 3));
    // the lane's LDS slot is physical chunk (lane & 7) of that row = logical chunk (lane & 7) ^ ((row >> 1) & 7), and
    // (row >> 1) & 7 = (lane >> 4) | ((i & 1) << 2) for both tiles
    const int lr = lane >> 3, lc = lane & 7;
    int aoff[4];
    uint32_t amask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chunk = lc ^ ((lane >> 4) | ((i & 1) << 2));
        const int m = m0 + wave * 32 + 8 * i + lr;
        aoff[i] = 0; amask[i] = 0;
        if (m < p.M) {
            if (p.simple) { aoff[i] = (m * p.Cin + chunk * 8) * 2; amask[i] = 1u; continue; }
            int q1 = (int)((float)m * p.inv_wo), wo = m - q1 * p.Wo;
            if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
            int n = (int)((float)q1 * p.inv_ho), ho = q1 - n * p.Ho;
            if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            aoff[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + chunk * 8) * 2;
            for (int r = 0; r < p.R; ++r)
                for (int s2 = 0; s2 < p.S; ++s2)
                    if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + s2) < (unsigned)p.W) amask[i] |= 1u << (r * p.S + s2);
        }
    }
    const int K = p.KT * CV_BK;
    int woff[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int chunk = lc ^ ((lane >> 4) | ((i & 1) << 2));
        woff[i] = ((n0 + wave * 8 * BI + 8 * i + lr) * K + chunk * 8) * 2;
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), (short)0,
                                                                           (int)((unsigned)(p.N * p.H * p.W) * (unsigned)p.Cin * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), (short)0,
                                                                           (int)((unsigned)p.Cout * (unsigned)K * 2u), 0x00020000);
    uint32_t af[MI][4], bf[NI][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = wm * WM + mi * 32 + frow;
            af[mi][kk] = row * CV_ROWB + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) << 4);
            asm volatile("" : "+v"(af[mi][kk]));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int row = wn * 64 + ni * 32 + frow;
            bf[ni][kk] = A_BYTES + row * CV_ROWB + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) << 4);
            asm volatile("" : "+v"(bf[ni][kk]));
        }
    }
    int ld_tap = 0, ld_c = 0, ld_r = 0, ld_s = 0, ld_k = 0;

#define CV_DMA(rs, ldsoff, vo, so) cv_dma16(rs, smem + (ldsoff), vo, so)
#define CV_ISSUE_TILE(stage)                                                                                    \
    {                                                                                                           \
        const int koff = ((ld_r * p.W + ld_s) * p.Cin + ld_c * CV_BK) * 2;                                      \
        const uint32_t bit = 1u << ld_tap;                                                                      \
        const int abase = (stage) * STAGE + wave * 4096;                                                        \
        CV_DMA(rs_x, abase + 0 * 1024, (amask[0] & bit) ? aoff[0] + koff : CV_OOB, 0);                          \
        CV_DMA(rs_x, abase + 1 * 1024, (amask[1] & bit) ? aoff[1] + koff : CV_OOB, 0);                          \
        CV_DMA(rs_x, abase + 2 * 1024, (amask[2] & bit) ? aoff[2] + koff : CV_OOB, 0);                          \
        CV_DMA(rs_x, abase + 3 * 1024, (amask[3] & bit) ? aoff[3] + koff : CV_OOB, 0);                          \
        const int wso = ld_k * CV_BK * 2;                                                                       \
        const int bbase = (stage) * STAGE + A_BYTES + wave * (BI * 1024);                                       \
        CV_DMA(rs_w, bbase + 0 * 1024, woff[0], wso);                                                           \
        CV_DMA(rs_w, bbase + 1 * 1024, woff[1], wso);                                                           \
        if (BI == 4) { CV_DMA(rs_w, bbase + 2 * 1024, woff[BI - 2], wso); CV_DMA(rs_w, bbase + 3 * 1024, woff[BI - 1], wso); } \
        ++ld_k;                                                                                                 \
        if (++ld_c == p.cpk) { ld_c = 0; ++ld_tap; if (++ld_s == p.S) { ld_s = 0; ++ld_r; } }                   \
    }
#define CV_MFMA_STEP(stage)                                                                                     \
    {                                                                                                           \
        _Pragma("unroll")                                                                                       \
        for (int kk = 0; kk < 4; ++kk) {                                                                        \
            bf16x8 a[MI], b[NI];                                                                                \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(smem + (stage) * STAGE + af[mi][kk]); \
            _Pragma("unroll")                                                                                   \
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(smem + (stage) * STAGE + bf[ni][kk]); \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi)                                                                     \
                _Pragma("unroll")                                                                               \
                for (int ni = 0; ni < NI; ++ni)                                                                 \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);  /* D'[channel][pixel] */  \
        }                                                                                                       \
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    if (NST == 1) {
        for (int kt = 0; kt < p.KT; ++kt) {
            CV_ISSUE_TILE(0);
            __syncthreads();                                        // (drains the DMA: vmcnt(0) before the barrier)
            CV_MFMA_STEP(0);
            __syncthreads();
        }
    } else {
    CV_ISSUE_TILE(0);
        __syncthreads();                                            // (drains the DMA: vmcnt(0) before the barrier)
        // Two K-steps per trip (stages 0 and 1 are literals), whole pairs only and the odd last step peeled off AFTER the loop:
        // a `break` between the two halves gives the loop two exits and makes the compiler carry a second copy of the 64
        // accumulators through v_accvgpr_read / v_accvgpr_write (138 extra VALU per 32 MFMAs, each waiting for its MFMA).
        int kt = 0;
        for (; kt + 2 <= p.KT; kt += 2) {
            CV_ISSUE_TILE(1);                                       // tile kt + 1
            CV_MFMA_STEP(0);                                        // tile kt
            __syncthreads();
            if (kt + 2 < p.KT) CV_ISSUE_TILE(0);                    // tile kt + 2
            CV_MFMA_STEP(1);                                        // tile kt + 1
            __syncthreads();
        }
        if (kt < p.KT) {                                            // odd K-step count: the last tile sits in stage 0
            CV_MFMA_STEP(0);
            __syncthreads();
        }
    }
#undef CV_MFMA_STEP
#undef CV_ISSUE_TILE
#undef CV_DMA
    cv_epilogue<BN, LEAN, (NST == 1 && !LEAN) ? 4 : 2>(p, acc, smem, t, m0, n0, mt);
}


// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 CU tile, 16 wavefronts (round 3). The counters say the K loops above are bound by the LDS-staging instructions themselves
// (one 1 KB `buffer_load ... lds` piece per ~50 clocks and CU at 93 % L2 hits, profiles/r03_conv_l2_counters.txt), so the lever is
// staged bytes per FLOP: here ONE workgroup of 1024 threads owns a 256 x 256 output tile and its 16 wavefronts — each with the
// proven 64 x 64 wave tile, four per SIMD as with four independent 128 x 128 workgroups — share one 64 KB K-step stage
// (A 256 rows + B 256 rows): half the pieces per FLOP (4 per wavefront and K-step instead of 8). Two stages (128 KB), one barrier per
// K-step. The epilogue is the 128 x 128 one, run by the four 4-wavefront groups on their quadrants (four staging tiles = 152 KB of the
// LDS the stages no longer need); statistics rows stay one per 128 pixels. Needs Cout % 256 == 0 and M % 256 == 0.
// The template: TM x TN tile (multiples of 128) on (TM / 64) x (TN / 64) wavefronts, NST LDS stages. <256, 256, 2> is the form described
// above; <256, 128, 1> (8 wavefronts, one 48 KB stage, two workgroups per CU = the same 16 wavefronts per CU) is the short-K variant:
// 6 instead of 8 pieces per wavefront and K-step.
constexpr int CVB_EPI = CV_BM * (128 * 2 + 16) + 4 * 2 * 128 * 4;                  // 38 912 B per 128 x 128 quadrant
template <int TM, int TN, int NST> struct CvbGeom {
    static constexpr int NW = (TM / 64) * (TN / 64), THREADS = 64 * NW, NQ = (TM / 128) * (TN / 128), QN = TN / 128;
    static constexpr int A_BYTES = TM * CV_ROWB, STAGE = (TM + TN) * CV_ROWB;
    static constexpr int APW = TM / 8 / NW, BPW = TN / 8 / NW;                     // 1 KB pieces per wavefront and K-step (A, B)
    static constexpr int LDS = (NQ * CVB_EPI > NST * STAGE) ? NQ * CVB_EPI : NST * STAGE;
    static_assert(APW >= 1 && BPW >= 1 && (TM / NW) % 16 == 0 && (TN / NW) % 16 == 0, "a wavefront loads whole 16-row groups (the swizzle phase)");
};
template <int TM, int TN, int NST, bool LEAN>
__global__ void __launch_bounds__((CvbGeom<TM, TN, NST>::THREADS)) __attribute__((amdgpu_waves_per_eu(4)))
conv_igemm_big_kernel(ConvP p, int ntn2, int nblocks2) {
    using G = CvbGeom<TM, TN, NST>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int MI = 2, NI = 2;
    int lin;
    {
        const int b = blockIdx.x, q = nblocks2 / 8, r = nblocks2 % 8, xcd = b % 8, i = b / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int mt2 = lin / ntn2, nt2 = lin - mt2 * ntn2;
    const int m0 = mt2 * TM, n0 = nt2 * TN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, wl = wave & 3, gm = grp / G::QN, gn = grp - gm * G::QN, wm = wl >> 1, wn = wl & 1;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int lr = lane >> 3, lc = lane & 7;
    int aoff[G::APW];
    uint32_t amask[G::APW];
#pragma unroll
    for (int i = 0; i < G::APW; ++i) {
        const int chunk = lc ^ ((lane >> 4) | ((i & 1) << 2));
        const int m = m0 + wave * (TM / G::NW) + 8 * i + lr;
        aoff[i] = 0; amask[i] = 0;
        if (m < p.M) {
            if (p.simple) { aoff[i] = (m * p.Cin + chunk * 8) * 2; amask[i] = 1u; continue; }
            int q1 = (int)((float)m * p.inv_wo), wo = m - q1 * p.Wo;
            if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
            int n = (int)((float)q1 * p.inv_ho), ho = q1 - n * p.Ho;
            if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            aoff[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + chunk * 8) * 2;
            for (int r = 0; r < p.R; ++r)
                for (int s2 = 0; s2 < p.S; ++s2)
                    if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + s2) < (unsigned)p.W) amask[i] |= 1u << (r * p.S + s2);
        }
    }
    const int K = p.KT * CV_BK;
    int woff[G::BPW];
#pragma unroll
    for (int i = 0; i < G::BPW; ++i) {
        const int chunk = lc ^ ((lane >> 4) | ((i & 1) << 2));
        woff[i] = ((n0 + wave * (TN / G::NW) + 8 * i + lr) * K + chunk * 8) * 2;
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), (short)0,
                                                                           (int)((unsigned)(p.N * p.H * p.W) * (unsigned)p.Cin * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), (short)0,
                                                                           (int)((unsigned)p.Cout * (unsigned)K * 2u), 0x00020000);
    uint32_t af[MI][4], bf[NI][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = gm * 128 + wm * 64 + mi * 32 + frow;
            af[mi][kk] = row * CV_ROWB + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) << 4);
            asm volatile("" : "+v"(af[mi][kk]));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int row = gn * 128 + wn * 64 + ni * 32 + frow;
            bf[ni][kk] = G::A_BYTES + row * CV_ROWB + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) << 4);
            asm volatile("" : "+v"(bf[ni][kk]));
        }
    }
    int ld_tap = 0, ld_c = 0, ld_r = 0, ld_s = 0, ld_k = 0;
#define CVB_ISSUE(stage)                                                                                        \
    {                                                                                                           \
        const int koff = ((ld_r * p.W + ld_s) * p.Cin + ld_c * CV_BK) * 2;                                      \
        const uint32_t bit = 1u << ld_tap;                                                                      \
        const int abase = (stage) * G::STAGE + wave * (G::APW * 1024);                                          \
        _Pragma("unroll")                                                                                       \
        for (int i = 0; i < G::APW; ++i) cv_dma16(rs_x, smem + abase + i * 1024, (amask[i] & bit) ? aoff[i] + koff : CV_OOB, 0); \
        const int wso = ld_k * CV_BK * 2;                                                                       \
        const int bbase = (stage) * G::STAGE + G::A_BYTES + wave * (G::BPW * 1024);                             \
        _Pragma("unroll")                                                                                       \
        for (int i = 0; i < G::BPW; ++i) cv_dma16(rs_w, smem + bbase + i * 1024, woff[i], wso);                 \
        ++ld_k;                                                                                                 \
        if (++ld_c == p.cpk) { ld_c = 0; ++ld_tap; if (++ld_s == p.S) { ld_s = 0; ++ld_r; } }                   \
    }
#define CVB_MFMA(stage)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll")                                                                                       \
        for (int kk = 0; kk < 4; ++kk) {                                                                        \
            bf16x8 a[MI], b[NI];                                                                                \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(smem + (stage) * G::STAGE + af[mi][kk]); \
            _Pragma("unroll")                                                                                   \
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(smem + (stage) * G::STAGE + bf[ni][kk]); \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi)                                                                     \
                _Pragma("unroll")                                                                               \
                for (int ni = 0; ni < NI; ++ni)                                                                 \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);  \
        }                                                                                                       \
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
    if (NST == 1) {
        for (int kt = 0; kt < p.KT; ++kt) {
            CVB_ISSUE(0);
            __syncthreads();
            CVB_MFMA(0);
            __syncthreads();
        }
    } else {
        CVB_ISSUE(0);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 <= p.KT; kt += 2) {
            CVB_ISSUE(1);
            CVB_MFMA(0);
            __syncthreads();
            if (kt + 2 < p.KT) CVB_ISSUE(0);
            CVB_MFMA(1);
            __syncthreads();
        }
        if (kt < p.KT) {
            CVB_MFMA(0);
            __syncthreads();
        }
    }
#undef CVB_MFMA
#undef CVB_ISSUE
    cv_epilogue<128, LEAN, LEAN ? 2 : 4>(p, acc, smem + grp * CVB_EPI, t & 255, m0 + gm * 128, n0 + gn * 128, mt2 * (TM / 128) + gm);
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 on 56^2, 28^2, 14^2 maps (conv2 of the Bottlenecks and, with rotated weights, its data gradient):
// the K loops above fetch the A tile once per filter tap — nine shifted copies of the same pixels — and their time is that
// L2 -> LDS traffic (~12 TB/s chip-wide), not the MFMAs. Here the M tile is a chunk of RB whole image rows (<= 112 pixels of
// the 128-row MFMA tile; the padded rows are computed on garbage-free dummy pixels and never stored) and, per 64-channel
// block, its input PATCH ((RB + 2) rows x P pixels, zero border materialised, P = row pitch rounded up to a multiple of 16) is
// DMA'd into LDS ONCE; the nine taps then read their A fragments from the patch at a row offset of (r * P + s) * 128 bytes,
// and only the 64-wide weight slices stream per tap. A traffic / 9: total L2 -> LDS bytes x0.48 (64 channels) ... x0.74 (256).
// LDS image of the patch: 128-byte pixel rows, 16-byte chunks XOR-swizzled with (row >> 1) & 7 like the A tile above (applied
// on the DMA source side); r * P is a multiple of 16 rows, so the tap row is an instruction immediate and only the three tap
// columns need their own addresses. The DMA is inline assembly (see dir_conv_wgrad3.hip: the compiler would otherwise drain
// the pending LDS-DMA in front of LDS reads it cannot disambiguate) with an explicit s_waitcnt before each barrier.
template <int WI> struct CpGeom {
    static constexpr int RB = (WI == 56) ? 2 : (WI == 28) ? 4 : 7;
    static constexpr int CPI = WI / RB;
    static constexpr int P = (WI == 56) ? 64 : (WI == 28) ? 32 : 16;
    static constexpr int KPIX = RB * WI;
    static constexpr int PROWS = (RB + 2) * P;                    // patch rows (a multiple of 8)
    static constexpr int PPIECES = PROWS / 8;
    static constexpr int PPW = (PPIECES + 3) / 4;                 // patch pieces per wavefront
    static constexpr int PATCH = PROWS * CV_ROWB;
    static_assert(WI % RB == 0 && P % 16 == 0 && P >= WI + 2 && PROWS % 8 == 0, "chunk geometry");
};

// NST = 2: two patch stages (Cin > 64) and two weight stages, the next slice in flight under the MFMAs: 56-80 KB of LDS, two
// workgroups per CU. NST = 1: one stage each (<= 40 KB, <= 128 registers): FOUR workgroups per CU, no overlap inside a workgroup.
template <int WI, int BN, int NST>
__global__ void __launch_bounds__(DIR_TPB) __attribute__((amdgpu_waves_per_eu(NST == 1 ? 4 : 2)))
conv3x3_patch_kernel(ConvP p) {
    using G = CpGeom<WI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int B_BYTES = BN * CV_ROWB;
    constexpr int BI = BN / 32;                           // B pieces per wavefront and K-step
    constexpr int MI = (BN == 128) ? 2 : 1;
    constexpr int NI = 2;
    constexpr int WM = MI * 32;
    // LDS: [patch stage 0][patch stage 1 (only used when Cin > 64)][B stage 0][B stage 1]
    const int npatch = (NST == 2 && p.cpk > 1) ? 2 : 1;
    const uint32_t b_base = (uint32_t)(npatch * G::PATCH);
    int lin;
    {
        const int b = blockIdx.x, q = p.nblocks / 8, r = p.nblocks % 8, xcd = b % 8, i = b / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int mt = lin / p.ntn, nt = lin - mt * p.ntn;    // mt = chunk index
    const int n_img = mt / G::CPI, h0 = (mt - n_img * G::CPI) * G::RB;
    const int m0 = (n_img * WI + h0) * WI, n0 = nt * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = (BN == 128) ? (wave >> 1) : wave;
    const int wn = (BN == 128) ? (wave & 1) : 0;
    const int frow = lane & 31, fhalf = lane >> 5;
    typedef __attribute__((address_space(3))) unsigned char* cp_lds_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(cp_lds_t)smem;

    const cp_u32x4 rs_x = cp_rsrc(p.x, (uint32_t)(p.N * p.H * p.W) * (uint32_t)p.Cin * 2u);
    const int K = p.KT * CV_BK;
    const cp_u32x4 rs_w = cp_rsrc(p.w, (uint32_t)p.Cout * (uint32_t)K * 2u);

    // ---- patch DMA roles: piece q = wave + 4 i = patch rows 8 q .. 8 q + 7; the lane fills physical chunk (lane & 7) of row
    // 8 q + (lane >> 3) with LOGICAL chunk (lane & 7) ^ ((row >> 1) & 7)
    const int lr = lane >> 3, lc = lane & 7;
    int prel[G::PPW];                                     // byte offset of the lane's source chunk relative to pixel (h0 - 1, -1) of
                                                          // the image, channel block 0; CV_OOB = zero border / unused slot
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) {
        const int q = wave + 4 * i;
        prel[i] = CV_OOB;
        if (q < G::PPIECES) {
            const int row = q * 8 + lr;
            const int pr = row / G::P, pc = row - pr * G::P;
            const int hi = h0 - 1 + pr;
            if (pc >= 1 && pc <= WI && hi >= 0 && hi < WI) prel[i] = ((pr * WI + pc) * p.Cin + (lc ^ ((row >> 1) & 7)) * 8) * 2;
        }
    }
    const int xbase = (((n_img * WI + h0 - 1) * WI - 1) * p.Cin) * 2;      // (signed) byte offset of pixel (h0 - 1, -1)
    // ---- weight DMA roles (as conv_igemm_dma_kernel): piece i of this wave = rows wave*8*BI + 8 i + lr of the B tile
    int woff[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int chunk = lc ^ ((lane >> 4) | ((i & 1) << 2));
        woff[i] = ((n0 + wave * 8 * BI + 8 * i + lr) * K + chunk * 8) * 2;
    }
    // ---- fragment reads. A: tile row (pixel) k -> patch row pp0 at tap (0, 0); per tap column s the row pp0 + s and its swizzle
    uint32_t arow[MI][3], az[MI][3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int k = wm * WM + mi * 32 + frow;
        const int i = k / WI, w = k - i * WI;
        const int pp0 = k < G::KPIX ? i * G::P + w : 0;   // padded tile rows read the patch's (finite) first pixels; never stored
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
            arow[mi][s2] = (uint32_t)(pp0 + s2) * CV_ROWB;
            az[mi][s2] = (uint32_t)(((pp0 + s2) >> 1) & 7);
        }
    }
    uint32_t bf[NI][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int row = wn * 64 + ni * 32 + frow;
            bf[ni][kk] = row * CV_ROWB + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) << 4);
            asm volatile("" : "+v"(bf[ni][kk]));
        }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

#define CP_ISSUE_PATCH(cb, ps)                                                                                  \
    {                                                                                                           \
        const int cbo = xbase + (cb) * CV_BK * 2;                                                               \
        _Pragma("unroll")                                                                                       \
        for (int i = 0; i < G::PPW; ++i) {                                                                      \
            const int q = wave + 4 * i;                                                                         \
            if (q < G::PPIECES) cp_dma16(rs_x, lds0 + (uint32_t)((ps) * G::PATCH + q * 1024), prel[i] == CV_OOB ? CV_OOB : cbo + prel[i], 0); \
        }                                                                                                       \
    }
    // K-step (tap, cb) of the weights = columns (tap * cpk + cb) * 64 .. + 63 of the [Cout][R*S*Cin] matrix
#define CP_ISSUE_B(tap, cb, bs)                                                                                 \
    {                                                                                                           \
        const int wso = ((tap) * p.cpk + (cb)) * CV_BK * 2;                                                     \
        const uint32_t bb = lds0 + b_base + (uint32_t)((bs) * B_BYTES + wave * (BI * 1024));                    \
        _Pragma("unroll")                                                                                       \
        for (int i = 0; i < BI; ++i) cp_dma16(rs_w, bb + i * 1024, woff[i], wso);                               \
    }
#define CP_MFMA_STEP(r_, s_, ps, bs)                                                                            \
    {                                                                                                           \
        const unsigned char* ab = smem + (ps) * G::PATCH + (r_) * (G::P * CV_ROWB);                             \
        const unsigned char* bbs = smem + b_base + (bs) * B_BYTES;                                              \
        _Pragma("unroll")                                                                                       \
        for (int kk = 0; kk < 4; ++kk) {                                                                        \
            bf16x8 a[MI], b[NI];                                                                                \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi)                                                                     \
                a[mi] = *reinterpret_cast<const bf16x8*>(ab + arow[mi][s_] + ((((uint32_t)(kk * 2 + fhalf)) ^ az[mi][s_]) << 4)); \
            _Pragma("unroll")                                                                                   \
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(bbs + bf[ni][kk]);         \
            _Pragma("unroll")                                                                                   \
            for (int mi = 0; mi < MI; ++mi)                                                                     \
                _Pragma("unroll")                                                                               \
                for (int ni = 0; ni < NI; ++ni)                                                                 \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);  \
        }                                                                                                       \
    }

    if (NST == 1) {
        for (int cb = 0; cb < p.cpk; ++cb) {
            CP_ISSUE_PATCH(cb, 0);                                  // (everyone is past the previous block's last MFMA: barrier below)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                CP_ISSUE_B(tap, cb, 0);
                cp_dma_wait();
                __syncthreads();
                CP_MFMA_STEP(tap / 3, tap % 3, 0, 0);
                __syncthreads();
            }
        }
    } else {
    CP_ISSUE_PATCH(0, 0);
        CP_ISSUE_B(0, 0, 0);
        cp_dma_wait();
        __syncthreads();
        int bs = 0;
        for (int cb = 0; cb < p.cpk; ++cb) {
            const int ps = cb & 1;
            if (cb + 1 < p.cpk) CP_ISSUE_PATCH(cb + 1, ps ^ 1);       // next channel block's patch lands during this block's nine taps
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // next weight slice in flight during the MFMAs (the first slice of the next channel block after tap 8)
                if (tap < 8) { CP_ISSUE_B(tap + 1, cb, bs ^ 1); }
                else if (cb + 1 < p.cpk) { CP_ISSUE_B(0, cb + 1, bs ^ 1); }
                CP_MFMA_STEP(tap / 3, tap % 3, ps, bs);
                cp_dma_wait();
                __syncthreads();
                bs ^= 1;
            }
        }
    }
#undef CP_MFMA_STEP
#undef CP_ISSUE_B
#undef CP_ISSUE_PATCH
    ConvP pe = p;
    pe.M = m0 + G::KPIX;                                          // rows past the chunk's pixels are padding: not stored, not counted
    cv_epilogue<BN, false, NST == 1 ? 4 : 2>(pe, acc, smem, t, m0, n0, mt);
}

template <int WI> constexpr int cp_chunks_per_image() { return CpGeom<WI>::CPI; }

}  // namespace

extern "C" size_t dir_conv_stats_rows(int N, int Ho, int Wo) {
    const long long M = (long long)N * Ho * Wo;
    return (size_t)((M + CV_BM - 1) / CV_BM);
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel selection. ONE function decides which kernel a launch takes and how that kernel tiles M (= how many rows the per-tile
// statistics / BatchNorm-partial list of the launch has); dir_conv_plan_rows (what the host sizes `stats` with) and the launcher
// both call it, and the launcher refuses a `stats_rows` that is not the plan's. No process-wide switches: what used to be
// dir_conv_set_* is the explicit `variant` argument, and the thresholds are constants (measured: profiles/r03_conv_big_tiles.txt).
//   variant 0 (DIR_CONV_AUTO)   the product heuristic below
//   variant 1 / 2               128 x 128 (x 64) tiles, register-staged / LDS-DMA K loop
//   variant 3                   patch-staged 3x3 (3x3 / stride 1 / pad 1 on square 56, 28, 14 maps only)
//   variant 5                   256 x 256 CU tile (Cout % 256 == 0, M % 256 == 0)
// Heuristic: the 256 x 256 CU tile from CVB_MIN_KT K-steps and CVB_MIN_TILES tiles (it loses on short K loops — the lone workgroup's
// prologue / epilogue are exposed — and on the 7^2 layers' 98 tiles); else the patch-staged kernel for its shapes (M tiles = chunks of
// whole image rows); else 128-row tiles: LDS-DMA from CV_DMA_MIN_KT K-steps, single-stage LDS-DMA at four workgroups per CU for
// 128-wide launches of <= 18 steps, register-staged otherwise.
constexpr int CVB_MIN_KT = 16, CVB_MIN_TILES = 150;
enum { CK_UNSUPPORTED = -1, CK_TILE = 0, CK_PATCH3 = 1, CK_BIG = 2 };
struct ConvPlan { int kind; int cpw; size_t rows; };

static int cp_chunks(int W) { return W == 56 ? 28 : W == 28 ? 7 : 2; }
static bool cp_geometry(int H, int W, int R, int S, int stride, int pad) {
    return R == 3 && S == 3 && stride == 1 && pad == 1 && H == W && (W == 56 || W == 28 || W == 14);
}
static bool conv_big_geometry(long long M, int Cout, int RS) { return Cout % 256 == 0 && M % 256 == 0 && RS <= 9; }

// (N, H, W) input map, output map (Ho, Wo); two_addends: the launch carries BOTH fused addends (the big kernel's epilogue takes one);
// cls: one parity class of a stride-2 data gradient (always 128-row tiles)
static ConvPlan conv_plan(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, bool two_addends,
                          bool cls, int variant) {
    const long long M = (long long)N * Ho * Wo;
    const size_t rows128 = (size_t)((M + CV_BM - 1) / CV_BM);
    if (variant == DIR_CONV_BIG)
        return conv_big_geometry(M, Cout, R * S) ? ConvPlan{CK_BIG, 0, rows128} : ConvPlan{CK_UNSUPPORTED, 0, 0};
    if (variant == DIR_CONV_PATCH3)
        return (!cls && cp_geometry(H, W, R, S, stride, pad)) ? ConvPlan{CK_PATCH3, W, (size_t)N * cp_chunks(W)} : ConvPlan{CK_UNSUPPORTED, 0, 0};
    if (variant == DIR_CONV_TILE_REG || variant == DIR_CONV_TILE_DMA) return ConvPlan{CK_TILE, 0, rows128};
    if (variant != DIR_CONV_AUTO) return ConvPlan{CK_UNSUPPORTED, 0, 0};
    if (!two_addends && conv_big_geometry(M, Cout, R * S) && R * S * (Cin / CV_BK) >= CVB_MIN_KT && (M / 256) * (Cout / 256) >= CVB_MIN_TILES)
        return ConvPlan{CK_BIG, 0, rows128};
    if (!cls && cp_geometry(H, W, R, S, stride, pad)) return ConvPlan{CK_PATCH3, W, (size_t)N * cp_chunks(W)};
    return ConvPlan{CK_TILE, 0, rows128};
}

// Rows of the per-tile statistics / BatchNorm-partial list of ONE launch with this geometry through `variant` (0 = the product
// heuristic): what `stats` must hold, and the `stats_rows` the launch entry points check. 0 = invalid geometry / variant not applicable.
extern "C" size_t dir_conv_plan_rows(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int two_addends, int variant) {
    if (N <= 0 || H <= 0 || W <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0 || Cin <= 0 || Cout <= 0) return 0;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const ConvPlan pl = conv_plan(N, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo, two_addends != 0, false, variant);
    return pl.kind == CK_UNSUPPORTED ? 0 : pl.rows;
}

static int conv_launch_ex(const void* x, const void* w, const void* addend, const void* addend_s2, const void* relu_mask, void* y,
                          float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                          int cls_a, int cls_b, int variant, dir_stream_t stream, const ConvBn* bn = nullptr);

extern "C" int dir_conv_fwd(const void* x, const void* w, void* y, float* stats, int stats_rows, int N, int H, int W, int Cin,
                            int Cout, int R, int S, int stride, int pad, dir_stream_t stream) {
    return conv_launch_ex(x, w, nullptr, nullptr, nullptr, y, stats, stats_rows, N, H, W, Cin, Cout, R, S, stride, pad, -1, 0, DIR_CONV_AUTO, stream);
}

extern "C" int dir_conv_fwd_fused(const void* x, const void* w, const void* addend, const void* relu_mask, void* y,
                                  float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  dir_stream_t stream) {
    return conv_launch_ex(x, w, addend, nullptr, relu_mask, y, stats, stats_rows, N, H, W, Cin, Cout, R, S, stride, pad, -1, 0, DIR_CONV_AUTO, stream);
}

// The same convolution with the kernel forced (tests and A/B measurements; DIR_EUNSUPPORTED when the geometry is not that kernel's).
extern "C" int dir_conv_fwd_variant(const void* x, const void* w, void* y, float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout,
                                    int R, int S, int stride, int pad, int variant, dir_stream_t stream) {
    return conv_launch_ex(x, w, nullptr, nullptr, nullptr, y, stats, stats_rows, N, H, W, Cin, Cout, R, S, stride, pad, -1, 0, variant, stream);
}

extern "C" int dir_conv_dgrad_join(const void* x, const void* w, const void* addend, const void* addend_s2,
                                   const void* relu_mask, void* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                                   int pad, dir_stream_t stream) {
    return conv_launch_ex(x, w, addend, addend_s2, relu_mask, y, nullptr, 0, N, H, W, Cin, Cout, R, S, 1, pad, -1, 0, DIR_CONV_AUTO, stream);
}

static int conv_dgrad_s2_impl(const void* dy, const void* wcls, void* dx, int N, int Ho, int Wo, int Cy, int Cx,
                              const ConvBn* bn, float* stats, int stats_rows, int variant, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !wcls || !dx, DIR_EINVAL);
    // classes (a, b) in the order (0,0) (0,1) (1,0) (1,1): 1, 2, 2, 4 filter taps, packed back to back as [Cx][taps][Cy]
    static const int tap_base[4] = {0, 1, 3, 5};
    const size_t rows = dir_conv_stats_rows(N, Ho, Wo);              // partial rows per class (fused BatchNorm-backward sums)
    DIR_RETURN_IF(stats && (size_t)stats_rows != 4 * rows, DIR_EINVAL);
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const uint16_t* wc = static_cast<const uint16_t*>(wcls) + (size_t)tap_base[a * 2 + b] * Cx * Cy;
            float* st = stats ? stats + (size_t)(a * 2 + b) * rows * 2 * Cx : nullptr;
            const int rc = conv_launch_ex(dy, wc, nullptr, nullptr, nullptr, dx, st, (int)rows, N, Ho, Wo, Cy, Cx, 1 + a, 1 + b, 1, 0, a, b, variant, stream, bn);
            if (rc != DIR_OK) return rc;
        }
    return DIR_OK;
}

extern "C" int dir_conv_dgrad_s2(const void* dy, const void* wcls, void* dx, int N, int Ho, int Wo, int Cy, int Cx,
                                 dir_stream_t stream) {
    return conv_dgrad_s2_impl(dy, wcls, dx, N, Ho, Wo, Cy, Cx, nullptr, nullptr, 0, DIR_CONV_AUTO, stream);
}

// The data gradients above with the FIRST pass of the BatchNorm backward that consumes them fused into the store loop
// (dir_bn_bwd_partials is the rest): bn_x = the input of that BatchNorm ([N, H', W', Cout] like the result), stats =
// [stats_rows][2][Cout] floats, stats_rows = dir_conv_plan_rows(...) (4 x dir_conv_stats_rows(N, Ho, Wo) for the stride-2 form: one
// block of rows per parity class). bn_gamma / bn_beta non-null: the BatchNorm is followed by a ReLU without residual; its mask is
// recomputed for the sums.
extern "C" int dir_conv_dgrad_bnstats(const void* x, const void* w, const void* addend, const void* addend_s2,
                                      const void* relu_mask, void* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                                      int pad, const void* bn_x, const float* bn_gamma, const float* bn_beta,
                                      const float* bn_mean, const float* bn_rstd, float* stats, int stats_rows, dir_stream_t stream) {
    DIR_RETURN_IF(!bn_x || !stats || !dir_aligned16(bn_x) || (bn_gamma && (!bn_beta || !bn_mean || !bn_rstd)), DIR_EINVAL);
    const ConvBn bn{bn_x, bn_gamma, bn_beta, bn_mean, bn_rstd, nullptr};
    return conv_launch_ex(x, w, addend, addend_s2, relu_mask, y, stats, stats_rows, N, H, W, Cin, Cout, R, S, 1, pad, -1, 0, DIR_CONV_AUTO, stream, &bn);
}

// The general stride-1 data gradient: dir_conv_dgrad_join (+ dir_conv_dgrad_bnstats when bn_x != NULL) with the ReLU mask given
// either as the tensor itself (relu_mask) or as the bit mask dir_bn_fwd_train_bits / dir_bn_apply_bits emitted
// (relu_mask_bits, [N*H*W][Cout / 8] bytes): 1/16 of the bytes for the same decision. At most one of the two. `variant` as in
// dir_conv_fwd_variant (0 = the product heuristic).
extern "C" int dir_conv_dgrad_ex(const void* x, const void* w, const void* addend, const void* addend_s2, const void* relu_mask,
                                 const void* relu_mask_bits, void* y, int N, int H, int W, int Cin, int Cout, int R, int S, int pad,
                                 const void* bn_x, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                                 const float* bn_rstd, float* stats, int stats_rows, int variant, dir_stream_t stream) {
    DIR_RETURN_IF((bn_x == nullptr) != (stats == nullptr) || (bn_x && !dir_aligned16(bn_x)), DIR_EINVAL);
    DIR_RETURN_IF(bn_gamma && (!bn_x || !bn_beta || !bn_mean || !bn_rstd), DIR_EINVAL);
    const ConvBn bn{bn_x, bn_gamma, bn_beta, bn_mean, bn_rstd, relu_mask_bits};
    return conv_launch_ex(x, w, addend, addend_s2, relu_mask, y, stats, stats_rows, N, H, W, Cin, Cout, R, S, 1, pad, -1, 0, variant, stream, &bn);
}

extern "C" int dir_conv_dgrad_s2_bnstats(const void* dy, const void* wcls, void* dx, int N, int Ho, int Wo, int Cy, int Cx,
                                         const void* bn_x, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                                         const float* bn_rstd, float* stats, int stats_rows, dir_stream_t stream) {
    DIR_RETURN_IF(!bn_x || !stats || !dir_aligned16(bn_x) || (bn_gamma && (!bn_beta || !bn_mean || !bn_rstd)), DIR_EINVAL);
    const ConvBn bn{bn_x, bn_gamma, bn_beta, bn_mean, bn_rstd, nullptr};
    return conv_dgrad_s2_impl(dy, wcls, dx, N, Ho, Wo, Cy, Cx, &bn, stats, stats_rows, DIR_CONV_AUTO, stream);
}

// The stride-2 3x3 data gradient with everything explicit (tests: the parity-class launches through a forced kernel).
extern "C" int dir_conv_dgrad_s2_ex(const void* dy, const void* wcls, void* dx, int N, int Ho, int Wo, int Cy, int Cx,
                                    const void* bn_x, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                                    const float* bn_rstd, float* stats, int stats_rows, int variant, dir_stream_t stream) {
    DIR_RETURN_IF((bn_x == nullptr) != (stats == nullptr) || (bn_x && !dir_aligned16(bn_x)), DIR_EINVAL);
    DIR_RETURN_IF(bn_gamma && (!bn_x || !bn_beta || !bn_mean || !bn_rstd), DIR_EINVAL);
    const ConvBn bn{bn_x, bn_gamma, bn_beta, bn_mean, bn_rstd, nullptr};
    return conv_dgrad_s2_impl(dy, wcls, dx, N, Ho, Wo, Cy, Cx, bn_x ? &bn : nullptr, stats, stats_rows, variant, stream);
}

// cls_a >= 0: parity class (cls_a, cls_b) of a stride-2 data gradient: x = dY [N, H, W, Cin], kernel (1 + a) x (1 + b)
// anchored top-left (zero beyond the bottom / right edge), output grid H x W stored at pixels (2 i + a, 2 j + b) of
// y [N, 2 H, 2 W, Cout]
static int conv_launch_ex(const void* x, const void* w, const void* addend, const void* addend_s2, const void* relu_mask, void* y,
                          float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                          int cls_a, int cls_b, int variant, dir_stream_t stream, const ConvBn* bn) {
    DIR_RETURN_IF(!x || !w || !y, DIR_EINVAL);
    const bool fwd_stats = stats && !(bn && bn->x);                              // forward statistics are of the conv result alone
    DIR_RETURN_IF(addend_s2 && (!dir_aligned16(addend_s2) || fwd_stats), DIR_EINVAL);
    DIR_RETURN_IF(addend && (!dir_aligned16(addend) || fwd_stats), DIR_EINVAL);
    DIR_RETURN_IF(relu_mask && (!dir_aligned16(relu_mask) || fwd_stats), DIR_EINVAL);
    DIR_RETURN_IF(bn && ((stats != nullptr) != (bn->x != nullptr)), DIR_EINVAL);
    DIR_RETURN_IF(bn && bn->mask_bits && relu_mask, DIR_EINVAL);
    DIR_RETURN_IF(N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0, DIR_EINVAL);
    DIR_RETURN_IF(Cin % CV_BK != 0 || Cout % 64 != 0, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(w) || !dir_aligned16(y), DIR_EINVAL);
    const bool cls = cls_a >= 0;
    DIR_RETURN_IF(cls && (stride != 1 || pad != 0 || addend || addend_s2 || relu_mask || fwd_stats), DIR_EINVAL);
    const int Ho = cls ? H : (H + 2 * pad - R) / stride + 1, Wo = cls ? W : (W + 2 * pad - S) / stride + 1;
    DIR_RETURN_IF(Ho <= 0 || Wo <= 0, DIR_EINVAL);
    DIR_RETURN_IF(addend_s2 && ((Ho | Wo) & 1), DIR_EUNSUPPORTED);
    const long long M = (long long)N * Ho * Wo;
    DIR_RETURN_IF(M >= (1ll << 24) || (long long)N * H * W * Cin >= (1ll << 30) || M * Cout >= (1ll << 31) || R * S > 32, DIR_EUNSUPPORTED);   // 32-bit byte offsets into x
    // which kernel, and how it tiles M: the caller's statistics list must have exactly the rows that kernel writes
    const ConvPlan plan = conv_plan(N, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo, addend && addend_s2, cls, variant);
    DIR_RETURN_IF(plan.kind == CK_UNSUPPORTED, (variant < 0 || variant > DIR_CONV_BIG || variant == 4) ? DIR_EINVAL : DIR_EUNSUPPORTED);
    DIR_RETURN_IF(stats && (size_t)stats_rows != plan.rows, DIR_EINVAL);
    ConvP p;
    p.x = static_cast<const uint16_t*>(x); p.w = static_cast<const uint16_t*>(w); p.y = static_cast<uint16_t*>(y);
    p.stats = stats;
    p.addend = static_cast<const uint16_t*>(addend);
    p.mask = static_cast<const uint16_t*>(relu_mask);
    p.addend2 = static_cast<const uint16_t*>(addend_s2);
    p.mask_bits = bn ? static_cast<const uint8_t*>(bn->mask_bits) : nullptr;
    p.bnx = bn ? static_cast<const uint16_t*>(bn->x) : nullptr;
    p.bn_gamma = bn ? bn->gamma : nullptr; p.bn_beta = bn ? bn->beta : nullptr;
    p.bn_mean = bn ? bn->mean : nullptr; p.bn_rstd = bn ? bn->rstd : nullptr;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = (int)M; p.cpk = Cin / CV_BK; p.KT = R * S * p.cpk;
    p.simple = (R == 1 && S == 1 && stride == 1 && pad == 0) ? 1 : 0;
    p.inv_wo = 1.0f / (float)Wo; p.inv_ho = 1.0f / (float)Ho;
    p.o2 = cls ? 1 : 0; p.o_a = cls ? cls_a : 0; p.o_b = cls ? cls_b : 0; p.OH = 2 * H; p.OW = 2 * W;
    DIR_RETURN_IF(cls && (long long)N * 4 * H * W * Cout >= (1ll << 31), DIR_EUNSUPPORTED);
    const int mtiles = (int)((M + CV_BM - 1) / CV_BM);
    const bool wide = (Cout % 128 == 0);
    p.ntn = wide ? Cout / 128 : Cout / 64;
    p.nblocks = mtiles * p.ntn;
    hipStream_t s = dir_s(stream);
    const int tile_n = wide ? 128 : 64;
    const int stage = CV_BM * (tile_n * 2 + 16) + 4 * 2 * tile_n * 4;      // epilogue staging + column partials
    if (plan.kind == CK_PATCH3) {
        // patch-staged 3x3: M tiles = chunks of whole image rows; one LDS stage (126 registers, 40 KB: four workgroups per CU)
        const int cpw = plan.cpw;
        p.nblocks = N * cp_chunks(cpw) * p.ntn;
        const int prows = cpw == 56 ? 256 : cpw == 28 ? 192 : 144;
        const int loop3 = prows * CV_ROWB + tile_n * CV_ROWB;
        const int lds3 = loop3 > stage ? loop3 : stage;
#define CP_LAUNCH(W_, BN_) hipLaunchKernelGGL((conv3x3_patch_kernel<W_, BN_, 1>), dim3(p.nblocks), dim3(DIR_TPB), lds3, s, p)
        if (cpw == 56) { if (wide) CP_LAUNCH(56, 128); else CP_LAUNCH(56, 64); }
        else if (cpw == 28) { if (wide) CP_LAUNCH(28, 128); else CP_LAUNCH(28, 64); }
        else { if (wide) CP_LAUNCH(14, 128); else CP_LAUNCH(14, 64); }
#undef CP_LAUNCH
        DIR_LAUNCH_CHECK();
        return DIR_OK;
    }
    if (plan.kind == CK_BIG) {
        // 256 x 256 CU tile on 16 wavefronts (half the LDS-DMA pieces per FLOP)
        using G = CvbGeom<256, 256, 2>;
        const bool leanb = !p.addend && !p.addend2 && !p.mask && !p.mask_bits && !p.bnx && !p.o2;
        DIR_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_big_kernel<256, 256, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
                            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_big_kernel<256, 256, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        const int ntn2 = Cout / 256, nb2 = (int)(M / 256) * ntn2;
        if (leanb) hipLaunchKernelGGL((conv_igemm_big_kernel<256, 256, 2, true>), dim3(nb2), dim3(G::THREADS), G::LDS, s, p, ntn2, nb2);
        else hipLaunchKernelGGL((conv_igemm_big_kernel<256, 256, 2, false>), dim3(nb2), dim3(G::THREADS), G::LDS, s, p, ntn2, nb2);
        DIR_LAUNCH_CHECK();
        return DIR_OK;
    }
    // ---- 128-row tiles. K-loop form: 2 = LDS-DMA (two 32/24 KB stages, no staging registers): loops of >= CV_DMA_MIN_KT steps, where
    // the LDS pipe bounds the register-staged loop. 1 = register-staged: LDS stages 2 (64 KB, 2 workgroups per CU) for long K loops,
    // 1 (43 KB, 3 per CU, one extra barrier) when the loop is short and the layer is bound by memory latency; prefetch distance 2
    // K-tiles once the loop is long enough to use them.
    const bool dma = variant == DIR_CONV_TILE_DMA || (variant == DIR_CONV_AUTO && p.KT >= CV_DMA_MIN_KT);
    if (dma) {
        const int loop2 = 2 * (CV_BM * CV_ROWB + tile_n * CV_ROWB);
        const int lds2 = loop2 > stage ? loop2 : stage;
        DIR_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_dma_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_dma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        if (wide) hipLaunchKernelGGL((conv_igemm_dma_kernel<128>), dim3(p.nblocks), dim3(DIR_TPB), lds2, s, p);
        else hipLaunchKernelGGL((conv_igemm_dma_kernel<64>), dim3(p.nblocks), dim3(DIR_TPB), lds2, s, p);
        DIR_LAUNCH_CHECK();
        return DIR_OK;
    }
    p.nbuf = p.KT <= 18 ? 1 : 2;
    const int pf = p.KT >= 36 ? 2 : 1;
    if (pf == 2) p.nbuf = 2;                                        // the two-tile prefetch is written for two LDS stages
    const int loop2 = p.nbuf * (CV_BM * CV_ROWB + tile_n * CV_ROWB);
    const int lds2 = loop2 > stage ? loop2 : stage;
    DIR_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<128, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<128, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
#define CV_LAUNCH(BN_, PF_, NB_) hipLaunchKernelGGL((conv_igemm_kernel<BN_, PF_, NB_>), dim3(p.nblocks), dim3(DIR_TPB), lds2, s, p)
    // lean = no fused operand (the plain forward): the single-stage kernel then fits four workgroups per CU
    const bool lean = !p.addend && !p.addend2 && !p.mask && !p.mask_bits && !p.bnx && !p.o2;
    // 128-wide tiles, short K loops (<= 18 steps): the single-stage LDS-DMA kernel, FOUR workgroups per CU (114 registers, 39 KB
    // of LDS) instead of three — throughput of these launches follows the resident workgroups (profiles/r02_conv_occupancy_sensitivity.txt;
    // A/B per train step -0.8 ms, per epoch-tail forward -0.7 ms). The 64-wide register-staged kernel already runs four per CU
    // (the DMA form measured slower there).
    const bool single = wide && p.nbuf == 1 && pf == 1 && variant == DIR_CONV_AUTO;
    if (single && lean) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 1, true>), dim3(p.nblocks), dim3(DIR_TPB), lds2, s, p);
    else if (single) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 1, false>), dim3(p.nblocks), dim3(DIR_TPB), lds2, s, p);
    else if (!wide && lean && p.nbuf == 1 && pf == 1) hipLaunchKernelGGL((conv_igemm_kernel<64, 1, 1, true>), dim3(p.nblocks), dim3(DIR_TPB), lds2, s, p);
    else if (wide) { if (pf == 2) CV_LAUNCH(128, 2, 2); else if (p.nbuf == 2) CV_LAUNCH(128, 1, 2); else CV_LAUNCH(128, 1, 1); }
    else           { if (pf == 2) CV_LAUNCH(64, 2, 2);  else if (p.nbuf == 2) CV_LAUNCH(64, 1, 2);  else CV_LAUNCH(64, 1, 1); }
#undef CV_LAUNCH
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight preparation: one launch per conv layer and optimizer step turns the float32 master weight
// [Cout][R][S][Cin] into the two bf16 operands the MFMA kernels consume — w16 (same layout, forward / wgrad shape)
// and, optionally, w16_rot [Cin][R][S][Cout] with the taps rotated by 180 degrees (the data-gradient convolution's
// weight) — instead of a cast + flip + permute + copy chain of library kernels.
namespace {
// One layer: w [Cout][RS][Cin] f32 -> w16 (same layout, bf16) and optionally w16_rot [Cin][RS][Cout] with the taps
// reversed. Cin, Cout % 64 == 0: the (co, ci) transpose of each tap goes through a 64x64 LDS tile, so both the float
// reads (256 B rows) and the two bf16 writes (128 B rows) are coalesced. Tiles are strided over gridDim.x.
// rot_mode 0: w16_rot = [Cin][R][S][Cout] with the taps rotated by 180 degrees (stride-1 data gradient).
// rot_mode 1 (3x3 only): w16_rot = the four parity-class weights of the STRIDE-2 data gradient, packed back to back:
//   class (a, b) = (r != 1, s != 1) holds the (1 + a) x (1 + b) taps that reach output pixels (2 i + a, 2 j + b), as
//   [Cin][(1 + a)(1 + b)][Cout] with tap (dr, ds) = (r == 0, s == 0) (dY row i + dr, column j + ds); class bases at
//   0, 1, 3, 5 taps.
__device__ __forceinline__ size_t conv_prep_rot_index(int rot_mode, int ci, int tap, int co, int RS, int Cin, int Cout) {
    if (rot_mode == 0) return ((size_t)ci * RS + (RS - 1 - tap)) * Cout + co;
    const int r = tap / 3, s = tap - 3 * r;
    const int a = r != 1, b = s != 1, dr = r == 0, ds = s == 0;
    const int taps = (1 + a) * (1 + b), t = dr * (1 + b) + ds;
    const int base = a ? (b ? 5 : 3) : (b ? 1 : 0);
    return (size_t)base * Cin * Cout + ((size_t)ci * taps + t) * Cout + co;
}

// Adam (torch.optim.Adam's single-tensor arithmetic, no amsgrad): one element; returns the new parameter value
struct AdamH { float step_size, beta2, one_minus_beta1, one_minus_beta2, eps, weight_decay, bias_correction2_sqrt; };   // (host: float64, then rounded, like torch's Python scalars)
__device__ __forceinline__ float adam_update(float p, float g, float* __restrict__ m, float* __restrict__ v, size_t i, const AdamH& h) {
    if (h.weight_decay != 0.0f) g = g + h.weight_decay * p;                      // grad.add(param, alpha=weight_decay)
    const float m0 = m[i], v0 = v[i];
    const float m1 = m0 + h.one_minus_beta1 * (g - m0);                           // exp_avg.lerp_(grad, 1 - beta1)
    const float v1 = v0 * h.beta2 + h.one_minus_beta2 * (g * g);                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    m[i] = m1; v[i] = v1;
    const float denom = sqrtf(v1) / h.bias_correction2_sqrt + h.eps;              // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    return p - h.step_size * (m1 / denom);                                        // param.addcdiv_(exp_avg, denom, value=-step_size), step_size = lr / bias_correction1
}

// torch.optim.SGD's single-tensor arithmetic (sgd.py: _single_tensor_sgd), one element; returns the new parameter value. The `add(x, alpha=a)`
// steps are written as fmaf, which is how torch's element-wise kernels evaluate `a + alpha * b`.
struct SgdH { float lr, momentum, one_minus_dampening, weight_decay; int nesterov, first; };
__device__ __forceinline__ float sgd_update(float p, float g, float* __restrict__ buf, size_t i, const SgdH& h) {
    if (h.weight_decay != 0.0f) g = __fmaf_rn(h.weight_decay, p, g);                 // grad = grad.add(param, alpha=weight_decay)
    if (h.momentum != 0.0f) {
        const float b = h.first ? g : __fmaf_rn(h.one_minus_dampening, g, buf[i] * h.momentum);   // buf = clone(grad) | buf.mul_(momentum).add_(grad, alpha=1 - dampening)
        buf[i] = b;
        g = h.nesterov ? __fmaf_rn(h.momentum, b, g) : b;                             // grad = grad.add(buf, alpha=momentum) | buf
    }
    return __fmaf_rn(-h.lr, g, p);                                                    // param.add_(grad, alpha=-lr)
}
// what the weight preparation pass does to the master weight first: nothing, an Adam step, an SGD step
struct OptNone { __device__ __forceinline__ float operator()(float w, size_t) const { return w; } static constexpr bool UPDATES = false; };
struct OptAdam { const float* g; float* m; float* v; AdamH h; static constexpr bool UPDATES = true;
                 __device__ __forceinline__ float operator()(float w, size_t i) const { return adam_update(w, g[i], m, v, i, h); } };
struct OptSgd { const float* g; float* buf; SgdH h; static constexpr bool UPDATES = true;
                __device__ __forceinline__ float operator()(float w, size_t i) const { return sgd_update(w, g[i], buf, i, h); } };

// OPT::UPDATES: w is first UPDATED in place from (g, m, v) — the optimizer step — and the bf16 operands are made from the new value in the
// same pass (dir_adam_step): one read of the master weight for the optimizer and the two layout conversions together.
template <typename OPT = OptNone>
__device__ __forceinline__ void conv_prep_body(float* __restrict__ w, int Cout, int RS, int Cin,
                                               uint16_t* __restrict__ w16, uint16_t* __restrict__ w16_rot, int rot_mode,
                                               const OPT opt = OPT{}) {
    __shared__ uint16_t tile[64][66];
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;           // 4 rows of 64 per pass
    if ((Cout & 63) || (Cin & 63)) {                               // generic fallback (not used by ResNet-50's layers)
        const size_t n = (size_t)Cout * RS * Cin;
        for (size_t i = (size_t)blockIdx.x * DIR_TPB + t; i < n; i += (size_t)gridDim.x * DIR_TPB) {
            float wv = w[i];
            if (OPT::UPDATES) { wv = opt(wv, i); w[i] = wv; }
            const uint16_t h = (uint16_t)cv_f2bf(wv);
            w16[i] = h;
            if (w16_rot) {
                const int ci = (int)(i % Cin); const size_t t1 = i / Cin; const int tap = (int)(t1 % RS); const int co = (int)(t1 / RS);
                w16_rot[conv_prep_rot_index(rot_mode, ci, tap, co, RS, Cin, Cout)] = h;
            }
        }
        return;
    }
    const int tci = Cin >> 6, tco = Cout >> 6, ntiles = tci * tco * RS;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int ci0 = (tl % tci) << 6; const int t1 = tl / tci; const int tap = t1 % RS; const int co0 = (t1 / RS) << 6;
#pragma unroll 4
        for (int r = ty; r < 64; r += 4) {                         // row = output channel, 64 consecutive input channels
            const size_t i = ((size_t)(co0 + r) * RS + tap) * Cin + ci0 + tx;
            float wv = w[i];
            if (OPT::UPDATES) { wv = opt(wv, i); w[i] = wv; }
            const uint16_t hb = (uint16_t)cv_f2bf(wv);
            w16[i] = hb;
            tile[r][tx] = hb;
        }
        if (w16_rot) {
            __syncthreads();
#pragma unroll 4
            for (int r = ty; r < 64; r += 4)                       // row = input channel, 64 consecutive output channels
                w16_rot[conv_prep_rot_index(rot_mode, ci0 + r, tap, co0 + tx, RS, Cin, Cout)] = tile[tx][r];
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(DIR_TPB)
conv_prep_weights_kernel(const float* __restrict__ w, int Cout, int RS, int Cin, uint16_t* __restrict__ w16,
                         uint16_t* __restrict__ w16_rot, int rot_mode) {
    conv_prep_body(const_cast<float*>(w), Cout, RS, Cin, w16, w16_rot, rot_mode);
}

// blockIdx.y = layer; the layer's row of the table holds its pointers and extents
__global__ void __launch_bounds__(DIR_TPB)
conv_prep_weights_batched_kernel(const long long* __restrict__ table) {
    const long long* e = table + (size_t)blockIdx.y * 7;
    conv_prep_body(reinterpret_cast<float*>(e[0]), (int)e[3], (int)e[4], (int)e[5],
                   reinterpret_cast<uint16_t*>(e[1]), reinterpret_cast<uint16_t*>(e[2]), (int)e[6]);
}
}  // namespace

namespace {
// One Adam step for EVERY parameter tensor of the network in one launch (blockIdx.y = tensor; row of the table = 12 int64:
// param, grad, exp_avg, exp_avg_sq, numel, w16, w16_rot, Cout, RS, Cin, rot_mode, 0). Rows with w16 != 0 are convolution weights:
// their bf16 operands for the next forward / data gradient are rewritten from the updated value in the same pass.
__global__ void __launch_bounds__(DIR_TPB)
adam_step_kernel(const long long* __restrict__ table, AdamH h) {
    const long long* e = table + (size_t)blockIdx.y * 12;
    float* w = reinterpret_cast<float*>(e[0]);
    const float* g = reinterpret_cast<const float*>(e[1]);
    float* m = reinterpret_cast<float*>(e[2]);
    float* v = reinterpret_cast<float*>(e[3]);
    if (e[5]) {
        conv_prep_body<OptAdam>(w, (int)e[7], (int)e[8], (int)e[9], reinterpret_cast<uint16_t*>(e[5]), reinterpret_cast<uint16_t*>(e[6]), (int)e[10], OptAdam{g, m, v, h});
        return;
    }
    const size_t n = (size_t)e[4];
    for (size_t i = (size_t)blockIdx.x * DIR_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * DIR_TPB)
        w[i] = adam_update(w[i], g[i], m, v, i, h);
}

// The same launch shape for torch.optim.SGD (rows: param, grad, momentum_buffer (0 = no momentum), 0, numel, w16, w16_rot, Cout, RS, Cin, rot_mode, 0)
__global__ void __launch_bounds__(DIR_TPB)
sgd_step_kernel(const long long* __restrict__ table, SgdH h) {
    const long long* e = table + (size_t)blockIdx.y * 12;
    float* w = reinterpret_cast<float*>(e[0]);
    const float* g = reinterpret_cast<const float*>(e[1]);
    float* buf = reinterpret_cast<float*>(e[2]);
    if (e[5]) {
        conv_prep_body<OptSgd>(w, (int)e[7], (int)e[8], (int)e[9], reinterpret_cast<uint16_t*>(e[5]), reinterpret_cast<uint16_t*>(e[6]), (int)e[10], OptSgd{g, buf, h});
        return;
    }
    const size_t n = (size_t)e[4];
    for (size_t i = (size_t)blockIdx.x * DIR_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * DIR_TPB)
        w[i] = sgd_update(w[i], g[i], buf, i, h);
}
}  // namespace

// torch.optim.SGD.step() (train.py:163-164 builds it for --optimizer sgd) for all parameters in ONE launch, fused with the bf16 weight
// preparation like dir_adam_step. table: device [ntensors][12] int64 (see sgd_step_kernel). first != 0: the momentum buffers do not
// hold a value yet (torch creates them as clone(grad) at the first step); momentum == 0: no buffer is touched (entries may be 0).
extern "C" int dir_sgd_step(const void* table, int ntensors, double lr, double momentum, double dampening, double weight_decay, int nesterov,
                            int first, dir_stream_t stream) {
    DIR_RETURN_IF(!table || ntensors <= 0 || ntensors > 65535, DIR_EINVAL);
    DIR_RETURN_IF(nesterov && (momentum <= 0.0 || dampening != 0.0), DIR_EINVAL);       // torch: "Nesterov momentum requires a momentum and zero dampening"
    SgdH h;
    h.lr = (float)lr; h.momentum = (float)momentum; h.one_minus_dampening = (float)(1.0 - dampening); h.weight_decay = (float)weight_decay;
    h.nesterov = nesterov ? 1 : 0; h.first = first ? 1 : 0;
    hipLaunchKernelGGL(sgd_step_kernel, dim3(128, ntensors), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const long long*>(table), h);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// torch.optim.Adam.step() (train.py:161-162 builds the optimizer, :259-260 steps it) for all parameters in ONE launch, fused with
// the bf16 weight preparation of the convolution layers (dir_conv_prep_weights_batched). table: device [ntensors][12] int64 (see
// adam_step_kernel); step >= 1 is the step count AFTER the increment (bias corrections 1 - beta^step, computed on the host in
// float64 like torch's non-capturable path). float32 parameters, gradients and state, all dense.
extern "C" int dir_adam_step(const void* table, int ntensors, double lr, double beta1, double beta2, double eps, double weight_decay,
                             long long step, dir_stream_t stream) {
    DIR_RETURN_IF(!table || ntensors <= 0 || ntensors > 65535 || step < 1, DIR_EINVAL);
    AdamH h;
    h.beta2 = (float)beta2; h.one_minus_beta1 = (float)(1.0 - beta1); h.one_minus_beta2 = (float)(1.0 - beta2);
    h.eps = (float)eps; h.weight_decay = (float)weight_decay;
    h.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    h.bias_correction2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adam_step_kernel, dim3(128, ntensors), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const long long*>(table), h);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_conv_prep_weights_batched(const void* table, int nlayers, dir_stream_t stream) {
    DIR_RETURN_IF(!table || nlayers <= 0 || nlayers > 65535, DIR_EINVAL);
    hipLaunchKernelGGL(conv_prep_weights_batched_kernel, dim3(128, nlayers), dim3(DIR_TPB), 0, dir_s(stream),
                       static_cast<const long long*>(table));
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_conv_prep_weights_ex(const float* w, int Cout, int R, int S, int Cin, void* w16, void* w16_rot,
                                        int rot_mode, dir_stream_t stream);
extern "C" int dir_conv_prep_weights(const float* w, int Cout, int R, int S, int Cin, void* w16, void* w16_rot,
                                     dir_stream_t stream) {
    return dir_conv_prep_weights_ex(w, Cout, R, S, Cin, w16, w16_rot, 0, stream);
}

extern "C" int dir_conv_prep_weights_ex(const float* w, int Cout, int R, int S, int Cin, void* w16, void* w16_rot,
                                        int rot_mode, dir_stream_t stream) {
    DIR_RETURN_IF(!w || !w16 || Cout <= 0 || R <= 0 || S <= 0 || Cin <= 0, DIR_EINVAL);
    DIR_RETURN_IF(rot_mode < 0 || rot_mode > 1 || (rot_mode == 1 && (R != 3 || S != 3)), DIR_EINVAL);
    const size_t n = (size_t)Cout * R * S * Cin;
    int grid = dir_cdiv((long long)n, DIR_TPB); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(conv_prep_weights_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), w, Cout, R * S, Cin,
                       static_cast<uint16_t*>(w16), static_cast<uint16_t*>(w16_rot), rot_mode);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
