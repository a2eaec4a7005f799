// Weight gradient of the NHWC bf16 convolution as an MFMA GEMM with the batch*pixel axis as K (gfx950).
//
//   dW[co, r, s, ci] = sum_{m = (n, ho, wo)} dY[m, co] * X[n, ho*stride - pad + r, wo*stride - pad + s, ci]
//
// Replaces the weight-gradient half of the autograd of nn.Conv2d in imdb-wiki-dir/resnet.py (cuDNN in the
// reference; MIOpen's atomic split-K kernels + zero-fill + cast passes in round 1 of this repo).
//
// GEMM view per filter tap: D[co][ci] (TM x TN tile) with K = M = N*Ho*Wo. BOTH operands are stored with K as the
// SLOW axis (rows m, channels contiguous), while an MFMA lane needs 8 consecutive k of one row. So the global ->
// LDS staging transposes on the fly: a thread fetches the same 16-byte channel group (8 channels) of 4
// consecutive rows m, permutes the 4x8 block in registers (v_perm_b32) into 8 x (4 consecutive k) and issues
// eight 8-byte ds_write_b64 into a [channel][k] tile — after that the tiles look exactly like the forward
// kernel's ([row][64 k], 128-byte rows, 16-byte chunks XOR-swizzled) and the fragment reads are plain
// ds_read_b128. Swizzle f(row) = ((row >> 1) ^ (row >> 4)) & 7: conflict-free fragment reads, 2-way on the
// transposing writes (the minimum for 16 lanes x 8 B landing on 8 chunk slots).
// Split-K: K is cut into `splits` contiguous ranges (one workgroup each, per tile and tap); partial tiles go to
// a float32 workspace [split][Cout][R*S][Cin] and are summed in split order by a second kernel: deterministic,
// no atomics, no zero-fill pass.
#include <cstdlib>
#include "dir_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct WgP {
    const uint16_t* dy; const uint16_t* x; float* part;
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
    int M, RS, tiles_m, tiles_n, splits, ksteps_total, ksteps_per_split;
    float inv_wo, inv_ho;
    int simple;               // 1x1, stride 1, pad 0: gathered row == m
    // incremental im2col walk (see WgWalk): extents in input-pixel units and byte corrections of the running offset
    int WoS, HoS;             // Wo * stride, Ho * stride
    int c1, c2;               // added when the walk wraps to the next image row / the next image
    int dws64, dhs64, dn64, c64;   // one K-step = 64 rows further
    int cstep;                // one row further
};

constexpr int WG_BK = 64, WG_ROWB = 128;
constexpr int WG_OOB = (int)0x80000000;            // buffer-load offset beyond any tensor: the load returns zeros

__device__ __forceinline__ int wg_swz(int row, int chunk) { return chunk ^ (((row >> 1) ^ (row >> 4)) & 7); }

// Position of one im2col row m = (n, ho, wo) for a fixed filter tap, kept incrementally: hs = ho * stride,
// ws = wo * stride and `off` = byte offset of X[n, hs + tr - pad, ws + ts - pad, ci0 + 8 cb]. Moving d rows further is
// mixed-radix addition with two carries; the offset moves by a constant plus a constant per carry — no division
// and no multiplication per row (the decode by reciprocals happens once per workgroup).
struct WgWalk { int n, hs, ws, off; };
__device__ __forceinline__ void wg_advance(WgWalk& w, const WgP& p, int dws, int dhs, int dn, int cbase) {
    w.ws += dws;
    const bool c1 = w.ws >= p.WoS;
    w.ws -= c1 ? p.WoS : 0;
    w.hs += dhs + (c1 ? p.stride : 0);
    const bool c2 = w.hs >= p.HoS;
    w.hs -= c2 ? p.HoS : 0;
    w.n += dn + (c2 ? 1 : 0);
    w.off += cbase + (c1 ? p.c1 : 0) + (c2 ? p.c2 : 0);
}
// 16 B of X for the row at `w` (zeros outside the image or past the last row: the buffer load is sent out of range)
__device__ __forceinline__ u32x4 wg_gather(__amdgpu_buffer_rsrc_t rx, const WgWalk& w, const WgP& p, int trp, int tsp) {
    const bool ok = (unsigned)(w.hs + trp) < (unsigned)p.H && (unsigned)(w.ws + tsp) < (unsigned)p.W && w.n < p.N;
    return __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? w.off : WG_OOB, 0, 0);
}

// 4 rows x 8 channels (4 x 16 B) -> 8 channels x 4 k, written as 8-byte pieces into the [channel][k] tile at the
// per-thread byte offsets o[0..7] (rows ch0 + 0..7).
__device__ __forceinline__ void wg_transpose_store(unsigned char* tile, const uint32_t (&o)[8], const u32x4& r0, const u32x4& r1,
                                                   const u32x4& r2, const u32x4& r3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // v_perm_b32(src0 = high dword, src1 = low dword): 0x05040100 -> lo16(src1) | lo16(src0) << 16
        const uint32_t lo01 = __builtin_amdgcn_perm(r1[q], r0[q], 0x05040100u), lo23 = __builtin_amdgcn_perm(r3[q], r2[q], 0x05040100u);
        const uint32_t hi01 = __builtin_amdgcn_perm(r1[q], r0[q], 0x07060302u), hi23 = __builtin_amdgcn_perm(r3[q], r2[q], 0x07060302u);
        *reinterpret_cast<uint2*>(tile + o[2 * q]) = make_uint2(lo01, lo23);
        *reinterpret_cast<uint2*>(tile + o[2 * q + 1]) = make_uint2(hi01, hi23);
    }
}

// PF = prefetch distance of the global loads in K-steps (2 for the 128 x 128 tile, which runs two workgroups per CU anyway: 248
// registers; the narrower tiles would drop from three to two workgroups per CU with the second register set)
// SIMPLE = 1x1 / stride 1 / pad 0 (row m of X is row m of the GEMM): a compile-time switch, because a run-time branch around the
// loads makes the compiler's wait counts conservative again.
template <int TM, int TN, bool SIMPLE, int PF = (TM == 128 && TN == 128) ? 2 : 1>
__global__ void __launch_bounds__(DIR_TPB) __attribute__((amdgpu_waves_per_eu(2)))
conv_wgrad_kernel(WgP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_BYTES = TM * WG_ROWB, B_BYTES = TN * WG_ROWB;
    constexpr int MI = TM / 64, NI = TN / 64;                     // 32x32 MFMA tiles per wavefront (2x2 wavefronts)
    constexpr int CA = TM / 8, CB = TN / 8;                       // 16-B chunk columns of the global tiles
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * A_BYTES;

    // Workgroup order: filter tap fastest, then ci tile, co tile, K range — workgroups that re-read the same dY rows
    // (all taps / ci tiles of one (co tile, K range)) and overlapping X rows (neighbouring taps) are adjacent, and
    // the XCD remap keeps adjacent ids on one XCD so those re-reads hit that XCD's L2.
    int b;
    {
        const int nb = gridDim.x, id = blockIdx.x, q = nb / 8, r = nb % 8, xcd = id % 8, i = id / 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int tap = b % p.RS; b /= p.RS;
    const int tn = b % p.tiles_n; b /= p.tiles_n;
    const int tm = b % p.tiles_m; b /= p.tiles_m;
    const int split = b;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int trp = tr - p.pad, tsp = ts - p.pad;
    const int co0 = tm * TM, ci0 = tn * TN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fhalf = lane >> 5;

    const int ks0 = split * p.ksteps_per_split;
    int ks1 = ks0 + p.ksteps_per_split; if (ks1 > p.ksteps_total) ks1 = p.ksteps_total;

    // loader roles: A chunk column ca / row group ga (threads < 2*TM), B chunk column cb / row group gb (threads < 2*TN)
    const bool doA = (2 * TM >= DIR_TPB) || t < 2 * TM, doB = (2 * TN >= DIR_TPB) || t < 2 * TN;   // (compile-time true for 128-wide tiles)
    const int ca = t % CA, ga = t / CA, cb = t % CB, gb = t / CB;

    // ---- everything that depends only on the thread is computed once and pinned in registers: the K loop then issues
    // loads, permutes, LDS accesses and MFMAs with (almost) no address arithmetic. (Left to itself the compiler
    // re-derives the swizzled LDS addresses every K-step: ~450 VALU instructions per 16 MFMAs, i.e. VALU bound.)
    uint32_t aw[8], bw[8];                                        // transposing stores
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ra = ca * 8 + q, rb = cb * 8 + q;
        aw[q] = ra * WG_ROWB + (wg_swz(ra, ga >> 1) << 4) + (ga & 1) * 8;
        bw[q] = rb * WG_ROWB + (wg_swz(rb, gb >> 1) << 4) + (gb & 1) * 8;
        asm volatile("" : "+v"(aw[q])); asm volatile("" : "+v"(bw[q]));
    }
    uint32_t af[MI][4], bf[NI][4];                                // fragment reads
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = wm * (TM / 2) + mi * 32 + frow;
            af[mi][kk] = row * WG_ROWB + (wg_swz(row, kk * 2 + fhalf) << 4);
            asm volatile("" : "+v"(af[mi][kk]));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int row = wn * (TN / 2) + ni * 32 + frow;
            bf[ni][kk] = row * WG_ROWB + (wg_swz(row, kk * 2 + fhalf) << 4);
            asm volatile("" : "+v"(bf[ni][kk]));
        }
    }
    // global byte offsets inside one K-step's 64 rows (the K-step itself is a wave-uniform soffset)
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dy), (short)0, (int)((unsigned)p.M * (unsigned)p.Cout * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), (short)0, (int)((unsigned)(p.N * p.H * p.W) * (unsigned)p.Cin * 2u), 0x00020000);
    int voa[4], vob[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        voa[j] = ((4 * ga + j) * p.Cout + co0 + ca * 8) * 2;
        vob[j] = ((4 * gb + j) * p.Cin + ci0 + cb * 8) * 2;       // simple (1x1, stride 1) case: row m of X is row m of the GEMM
        asm volatile("" : "+v"(voa[j])); asm volatile("" : "+v"(vob[j]));
    }
    WgWalk walk = {0, 0, 0, 0};
    if (!SIMPLE && doB) {                                       // decode the first row of this thread once
        const int mm = ks0 * WG_BK + 4 * gb;
        int q1 = (int)((float)mm * p.inv_wo); int wo = mm - q1 * p.Wo;
        if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
        int n = (int)((float)q1 * p.inv_ho); int ho = q1 - n * p.Ho;
        if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }
        walk.n = n; walk.hs = ho * p.stride; walk.ws = wo * p.stride;
        walk.off = (((n * p.H + walk.hs + trp) * p.W + walk.ws + tsp) * p.Cin + ci0 + cb * 8) * 2;
    }

    // Two register sets (P, Q): the global loads of K-step k + 2 are issued at the start of step k — a K-step's loads have two
    // MFMA steps to arrive instead of one (the loop is bound by that latency, ~1.5-2 us under load against 0.25-0.7 us per step).
    // Loads of steps beyond the workgroup's K range go out of range (zeros, no memory access) instead of being branched around:
    // the wait counter is in issue order, and only with unconditional loads can the compiler wait for "all but the newest 8".
    u32x4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;
    pa0 = pa1 = pa2 = pa3 = pb0 = pb1 = pb2 = pb3 = qa0 = qa1 = qa2 = qa3 = qb0 = qb1 = qb2 = qb3 = (u32x4){0u, 0u, 0u, 0u};

#define WG_BL(rs, vo, so) __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0)
#define WG_LOAD(ks, S)                                                                                            \
    {                                                                                                             \
        const bool live = (ks) < ks1;                                                                             \
        const int mrem = live ? p.M - (ks) * WG_BK : 0;           /* rows left from this K-step's first row */    \
        const bool full = mrem >= WG_BK;                                                                          \
        if (doA) {                                                                                                \
            const int so = live ? (ks) * WG_BK * p.Cout * 2 : 0;                                                  \
            S##a0 = WG_BL(rs_dy, (full || 4 * ga + 0 < mrem) ? voa[0] : WG_OOB, so);                              \
            S##a1 = WG_BL(rs_dy, (full || 4 * ga + 1 < mrem) ? voa[1] : WG_OOB, so);                              \
            S##a2 = WG_BL(rs_dy, (full || 4 * ga + 2 < mrem) ? voa[2] : WG_OOB, so);                              \
            S##a3 = WG_BL(rs_dy, (full || 4 * ga + 3 < mrem) ? voa[3] : WG_OOB, so);                              \
        }                                                                                                         \
        if (doB) {                                                                                                \
            if (SIMPLE) {                                                                                         \
                const int so = live ? (ks) * WG_BK * p.Cin * 2 : 0;                                               \
                S##b0 = WG_BL(rs_x, (full || 4 * gb + 0 < mrem) ? vob[0] : WG_OOB, so);                           \
                S##b1 = WG_BL(rs_x, (full || 4 * gb + 1 < mrem) ? vob[1] : WG_OOB, so);                           \
                S##b2 = WG_BL(rs_x, (full || 4 * gb + 2 < mrem) ? vob[2] : WG_OOB, so);                           \
                S##b3 = WG_BL(rs_x, (full || 4 * gb + 3 < mrem) ? vob[3] : WG_OOB, so);                           \
            } else {                                                                                              \
                WgWalk w = walk;                                                                                  \
                if (!live) w.n = p.N;                             /* (wg_gather sends n >= N out of range) */     \
                S##b0 = wg_gather(rs_x, w, p, trp, tsp); wg_advance(w, p, p.stride, 0, 0, p.cstep);               \
                S##b1 = wg_gather(rs_x, w, p, trp, tsp); wg_advance(w, p, p.stride, 0, 0, p.cstep);               \
                S##b2 = wg_gather(rs_x, w, p, trp, tsp); wg_advance(w, p, p.stride, 0, 0, p.cstep);               \
                S##b3 = wg_gather(rs_x, w, p, trp, tsp);                                                          \
                wg_advance(walk, p, p.dws64, p.dhs64, p.dn64, p.c64);                                             \
            }                                                                                                     \
        }                                                                                                         \
    }
#define WG_STORE(buf, S)                                                                                          \
    {                                                                                                             \
        if (doA) wg_transpose_store(As + (buf) * A_BYTES, aw, S##a0, S##a1, S##a2, S##a3);                        \
        if (doB) wg_transpose_store(Bs + (buf) * B_BYTES, bw, S##b0, S##b1, S##b2, S##b3);                        \
    }
    // one K-step on LDS stage `buf` (a literal: the stage offset folds into the instructions' immediate offsets)
#define WG_MFMA_STEP(buf)                                                                                         \
    {                                                                                                             \
        _Pragma("unroll")                                                                                         \
        for (int kk = 0; kk < 4; ++kk) {                                                                          \
            bf16x8 a[MI], bb[NI];                                                                                 \
            _Pragma("unroll")                                                                                     \
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(As + (buf) * A_BYTES + af[mi][kk]); \
            _Pragma("unroll")                                                                                     \
            for (int ni = 0; ni < NI; ++ni) bb[ni] = *reinterpret_cast<const bf16x8*>(Bs + (buf) * B_BYTES + bf[ni][kk]); \
            _Pragma("unroll")                                                                                     \
            for (int mi = 0; mi < MI; ++mi)                                                                       \
                _Pragma("unroll")                                                                                 \
                for (int ni = 0; ni < NI; ++ni)                                                                   \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);   \
        }                                                                                                         \
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    // two K-steps per trip (stages 0 and 1 and the register sets are literals), whole pairs only, the odd last step peeled off
    // after the loop: a `break` between the halves gives the loop two exits, and the compiler then copies all accumulators
    // (32 v_mov_b64 per 16 MFMAs, each waiting for its MFMA chain) on every trip
    if (ks0 < ks1) {
        int ks = ks0;
        if constexpr (PF == 2) {
            WG_LOAD(ks0, p);
            WG_LOAD(ks0 + 1, q);
            WG_STORE(0, p);
            __syncthreads();
            for (; ks + 2 <= ks1; ks += 2) {
                WG_LOAD(ks + 2, p);                              // stage 0 = tile ks, set q = tile ks + 1 (in flight since the last trip)
                __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise sinks these loads below the stores of set q,
                WG_MFMA_STEP(0);                                 //  which then wait for them: found in the ISA)
                WG_STORE(1, q);
                __syncthreads();
                WG_LOAD(ks + 3, q);
                __builtin_amdgcn_sched_barrier(0);
                WG_MFMA_STEP(1);
                WG_STORE(0, p);                                  // (tile ks + 2, or zeros past the range: never multiplied)
                __syncthreads();
            }
        } else {
            WG_LOAD(ks0, p);
            WG_STORE(0, p);
            __syncthreads();
            for (; ks + 2 <= ks1; ks += 2) {
                WG_LOAD(ks + 1, q);                              // global loads in flight during the MFMAs
                WG_MFMA_STEP(0);
                WG_STORE(1, q);
                __syncthreads();
                WG_LOAD(ks + 2, p);
                WG_MFMA_STEP(1);
                WG_STORE(0, p);
                __syncthreads();
            }
        }
        if (ks < ks1) WG_MFMA_STEP(0);                           // odd K-step count: the last tile sits in stage 0
    }
#undef WG_BL
#undef WG_LOAD
#undef WG_STORE
#undef WG_MFMA_STEP

    // partial[split][co][tap][ci] (fp32). C/D: col = lane & 31 -> ci, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) -> co.
    // Address = wave-uniform pointer (split, tile, register index) + one per-thread 32-bit offset.
    const int rsc = p.RS * p.Cin;
    float* out = p.part + (size_t)split * p.Cout * rsc + (size_t)co0 * rsc + (size_t)tap * p.Cin + ci0;
    const uint32_t toff = (uint32_t)((wm * (TM / 2) + 4 * fhalf) * rsc + wn * (TN / 2) + frow);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* ob = out + (size_t)(mi * 32 + (e & 3) + 8 * (e >> 2)) * rsc + ni * 32;
                ob[toff] = acc[mi][ni][e];
            }
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 / stride-1 weight gradient, register-lean form (round 4): dW[co][ci] = sum_m dY[m][co] * X[m][ci] with both operands staged in
// LDS in their NATURAL [pixel][channel] layout by LDS-DMA (buffer_load ... lds, no staging registers, no v_perm, no ds_write pass)
// and the MFMA fragments — 8 consecutive pixels (the K axis) of one channel per lane — read with the hardware-transposing
// ds_read_b64_tr_b16 (two per operand and 16-pixel step), as the all-taps 3x3 kernel does (dir_conv_wgrad3.hip). The transposing
// kernel above needs 248 registers (two workgroups per CU) and every removed phase left its 1x1 launches on 14^2 / 7^2 maps at ~45 us:
// 24 K-steps x a barrier that two wavefronts per SIMD cannot hide (profiles/r03_small_kernel_experiments.txt). This one holds 64
// accumulators + 16 fragment registers + a dozen addresses: <= 128 registers, one 32 KB stage (NST = 1: FOUR workgroups per CU, whose
// phases overlap each other) or two stages (NST = 2: two per CU, the next K-step's DMA in flight during the MFMAs).
// Tile 128 (co) x 128 (ci), 2 x 2 wavefronts of 64 x 64; K-step = 64 pixels; LDS rows of 256 B (128 channels), the 64-byte quarter
// of a row XORed with (pixel & 3) — on the DMA's SOURCE side, the destination is lane-linear — so that the four rows of a
// transposing read fall into four different 16-bank groups.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) __bf16 wg_bf16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t wg_u32x4;
__device__ __forceinline__ wg_bf16x4 wg_tr(const unsigned char* p) {
    typedef __attribute__((address_space(3))) wg_bf16x4* lds_v4_t;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)p);
}
__device__ __forceinline__ bf16x8 wg_cat(wg_bf16x4 a, wg_bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ wg_u32x4 wg_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    wg_u32x4 r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return r;
}
// one LDS-DMA piece: 64 lanes x 16 B -> 1 KB at lds_addr (wave-uniform), sources voffset (per lane) + soffset (wave-uniform).
// Inline assembly on purpose: the compiler would otherwise drain the DMA (s_waitcnt vmcnt(0)) in front of every LDS read it cannot
// prove disjoint from it.
__device__ __forceinline__ void wg_dma16(wg_u32x4 rs, uint32_t lds_addr, int voffset, int soffset) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voffset), "s"(rs), "s"(soffset) : "memory");
}

constexpr int W1_T = 128, W1_ROWB = W1_T * 2, W1_OP = WG_BK * W1_ROWB, W1_STAGE = 2 * W1_OP;      // 256-B rows, 16 KB per operand, 32 KB per stage

template <int NST>
__global__ void __launch_bounds__(DIR_TPB) __attribute__((amdgpu_waves_per_eu(NST == 1 ? 4 : 2)))
conv_wgrad1_dma_kernel(WgP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int b;
    {
        const int nb = gridDim.x, id = blockIdx.x, q = nb / 8, r = nb % 8, xcd = id % 8, i = id / 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int tn = b % p.tiles_n; b /= p.tiles_n;
    const int tm = b % p.tiles_m; b /= p.tiles_m;
    const int split = b;
    const int co0 = tm * W1_T, ci0 = tn * W1_T;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int ks0 = split * p.ksteps_per_split;
    int ks1 = ks0 + p.ksteps_per_split; if (ks1 > p.ksteps_total) ks1 = p.ksteps_total;

    // ---- DMA roles: wavefront w stages rows 16 w .. 16 w + 15 of both operands, 4 pieces of 4 rows each; lane l of a piece lands at
    // row (l >> 4), 16-byte chunk (l & 15) and fetches the chunk whose 64-byte quarter is XORed with the row's low two bits
    const wg_u32x4 rs_dy = wg_rsrc(p.dy, (uint32_t)p.M * (uint32_t)p.Cout * 2u), rs_x = wg_rsrc(p.x, (uint32_t)p.M * (uint32_t)p.Cin * 2u);
    const int lrow = 16 * wave + (lane >> 4);                                  // + 4 j for piece j
    const int lchunk = (lane & 15) ^ (((lane >> 4) & 3) << 2);                 // (row & 3) == (lane >> 4) & 3: 16 w + 4 j are multiples of 4
    const int vo_a = (lrow * p.Cout + co0) * 2 + lchunk * 16, vo_b = (lrow * p.Cin + ci0) * 2 + lchunk * 16;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;                            // (LDS addresses are 32-bit offsets)
    const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)(16 * wave) * W1_ROWB)), lb = la + W1_OP;   // wave-uniform: SGPRs (M0)
    const int rowa4 = 4 * p.Cout * 2, rowb4 = 4 * p.Cin * 2;                   // four rows further (the next piece)

    // ---- fragment reads: within a 16-lane group, lane i addresses pixel row (i >> 2), channels 4 (i & 3) .. + 3 of the group's 16
    // channels and receives channel i, four consecutive pixels; groups 0 / 1 = channels 0-15 / 16-31 of the 32-wide MFMA tile,
    // groups 2 / 3 the same for pixels 8-15 of the 16-pixel step
    const int gi = lane & 15, grp = lane >> 4;
    const int prow = (grp >> 1) * 8 + (gi >> 2);                               // + 4 h + 16 kk as instruction immediates
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ca = wm * 64 + i * 32 + (grp & 1) * 16 + 4 * (gi & 3), cb = wn * 64 + i * 32 + (grp & 1) * 16 + 4 * (gi & 3);
        fa[i] = (uint32_t)(prow * W1_ROWB + ((ca * 2) ^ ((prow & 3) << 6)));
        fb[i] = (uint32_t)(W1_OP + prow * W1_ROWB + ((cb * 2) ^ ((prow & 3) << 6)));
        asm volatile("" : "+v"(fa[i])); asm volatile("" : "+v"(fb[i]));
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

#define W1_ISSUE(ks, stage)                                                                                               \
    {                                                                                                                     \
        const int mrem = p.M - (ks) * WG_BK;                      /* rows left from this K-step's first row (> 0) */      \
        const int soa = (ks) * WG_BK * p.Cout * 2, sob = (ks) * WG_BK * p.Cin * 2;                                        \
        _Pragma("unroll")                                                                                                 \
        for (int j = 0; j < 4; ++j) {                                                                                     \
            const bool ok = lrow + 4 * j < mrem;                  /* rows past M: out of range -> zeros, no access */     \
            wg_dma16(rs_dy, la + (stage) * W1_STAGE + j * 4 * W1_ROWB, ok ? vo_a + j * rowa4 : WG_OOB, soa);              \
            wg_dma16(rs_x, lb + (stage) * W1_STAGE + j * 4 * W1_ROWB, ok ? vo_b + j * rowb4 : WG_OOB, sob);               \
        }                                                                                                                 \
    }
#define W1_MFMA(stage)                                                                                                    \
    {                                                                                                                     \
        const unsigned char* sb = smem + (stage) * W1_STAGE;                                                              \
        _Pragma("unroll")                                                                                                 \
        for (int kk = 0; kk < 4; ++kk) {                                                                                  \
            bf16x8 a[2], bb[2];                                                                                           \
            _Pragma("unroll")                                                                                             \
            for (int i = 0; i < 2; ++i) {                                                                                 \
                a[i] = wg_cat(wg_tr(sb + fa[i] + kk * 16 * W1_ROWB), wg_tr(sb + fa[i] + kk * 16 * W1_ROWB + 4 * W1_ROWB)); \
                bb[i] = wg_cat(wg_tr(sb + fb[i] + kk * 16 * W1_ROWB), wg_tr(sb + fb[i] + kk * 16 * W1_ROWB + 4 * W1_ROWB)); \
            }                                                                                                             \
            _Pragma("unroll")                                                                                             \
            for (int mi = 0; mi < 2; ++mi)                                                                                \
                _Pragma("unroll")                                                                                         \
                for (int ni = 0; ni < 2; ++ni)                                                                            \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);           \
        }                                                                                                                 \
    }
    if constexpr (NST == 1) {
        for (int ks = ks0; ks < ks1; ++ks) {
            W1_ISSUE(ks, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                      // everyone's pieces have landed
            W1_MFMA(0);
            __syncthreads();                                      // everyone is done reading the stage
        }
    } else {
        if (ks0 < ks1) {
            W1_ISSUE(ks0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int ks = ks0;
            for (; ks + 2 <= ks1; ks += 2) {                      // stages are literals: two K-steps per trip
                W1_ISSUE(ks + 1, 1);                              // lands during this step's MFMAs
                W1_MFMA(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (ks + 2 < ks1) W1_ISSUE(ks + 2, 0);
                W1_MFMA(1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (ks < ks1) W1_MFMA(0);                             // odd count: the last tile sits in stage 0
        }
    }
#undef W1_ISSUE
#undef W1_MFMA

    // partial[split][co][ci] (fp32). C/D: col = lane & 31 -> ci, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) -> co (as conv_wgrad_kernel)
    const int rsc = p.Cin;
    float* out = p.part + (size_t)split * p.Cout * rsc + (size_t)co0 * rsc + ci0;
    const uint32_t toff = (uint32_t)((wm * 64 + 4 * fhalf) * rsc + wn * 64 + frow);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* ob = out + (size_t)(mi * 32 + (e & 3) + 8 * (e >> 2)) * rsc + ni * 32;
                ob[toff] = acc[mi][ni][e];
            }
}

// Sum the split partials in split order. 256 threads = 16 float4 columns x 16 split lanes: lane j adds splits
// j, j+16, ... (independent loads in flight), then the 16 lane sums are added in lane order through LDS —
// a fixed summation tree, so the result is bit-reproducible, and a 1024-way split costs 64 loads per thread
// instead of a 1024-long dependent chain.
constexpr int RD_COLS = 16, RD_LANES = DIR_TPB / RD_COLS;
__global__ void __launch_bounds__(DIR_TPB)
conv_wgrad_reduce_kernel(const float* __restrict__ part, int splits, size_t n, float* __restrict__ dw) {
    __shared__ float4 sh[RD_LANES][RD_COLS];
    const int col = threadIdx.x % RD_COLS, sl = threadIdx.x / RD_COLS;
    const size_t i = ((size_t)blockIdx.x * RD_COLS + col) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
#pragma unroll 4
        for (int k = sl; k < splits; k += RD_LANES) {
            const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * n + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[sl][col] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        float4 t = sh[0][col];
#pragma unroll
        for (int k = 1; k < RD_LANES; ++k) { const float4 v = sh[k][col]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(dw + i) = t;
    }
}

// The same reduction for MANY layers in one launch (blockIdx.y = layer; row of the table = 4 int64: partials, splits, n, dw): the 52 per-layer
// launches of a backward pass (5-8 us each, 0.4 ms per step) become one at its end. Every element is summed exactly as above (split lane sl takes
// splits sl, sl + 16, ...; the lanes are combined in lane order): bit-identical to the per-layer kernel.
__global__ void __launch_bounds__(DIR_TPB)
conv_wgrad_reduce_batched_kernel(const long long* __restrict__ table) {
    __shared__ float4 sh[RD_LANES][RD_COLS];
    const long long* e = table + (size_t)blockIdx.y * 4;
    const float* __restrict__ part = reinterpret_cast<const float*>(e[0]);
    const int splits = (int)e[1];
    const size_t n = (size_t)e[2];
    float* __restrict__ dw = reinterpret_cast<float*>(e[3]);
    const int col = threadIdx.x % RD_COLS, sl = threadIdx.x / RD_COLS;
    const size_t groups = (n / 4 + RD_COLS - 1) / RD_COLS;
    for (size_t gi = blockIdx.x; gi < groups; gi += gridDim.x) {     // (block-uniform trip count: the barriers below are safe)
        const size_t i = (gi * RD_COLS + col) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
#pragma unroll 4
            for (int k = sl; k < splits; k += RD_LANES) {
                const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * n + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        sh[sl][col] = s;
        __syncthreads();
        if (sl == 0 && i < n) {
            float4 t = sh[0][col];
#pragma unroll
            for (int k = 1; k < RD_LANES; ++k) { const float4 v = sh[k][col]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            *reinterpret_cast<float4*>(dw + i) = t;
        }
        __syncthreads();
    }
}

struct WgPlan { int tm, tn, tiles_m, tiles_n, splits, ksteps_total, ksteps_per_split; size_t ws_bytes; int dma1; };

// dma1: the register-lean 1x1 form (conv_wgrad1_dma_kernel): 0 = the transposing kernel, 1 = one stage / four workgroups per CU,
// 2 = two stages / two per CU. W1_FORM is the product's choice for the 1x1 / stride-1 layers with 128-multiples of channels.
constexpr int W1_FORM = 2;
WgPlan wg_plan(long long M, int Cin, int Cout, int RS, int dma1 = 0) {
    WgPlan pl;
    pl.dma1 = dma1;
    pl.tm = (Cout % 128 == 0) ? 128 : 64;
    pl.tn = (Cin % 128 == 0) ? 128 : 64;
    pl.tiles_m = Cout / pl.tm; pl.tiles_n = Cin / pl.tn;
    pl.ksteps_total = (int)((M + WG_BK - 1) / WG_BK);
    const int tiles = pl.tiles_m * pl.tiles_n * RS;
    // Split-K so that the grid fills the chip a whole number of times: `slots` workgroups are resident at once (2 per CU
    // for the 128x128 tile, 3 otherwise: registers / LDS), and a grid slightly above a multiple of that costs a whole
    // extra round. One round measured best (4.99 -> 4.40 ms per ResNet-50 step at batch 256), unless that leaves fewer than
    // 4 K ranges per tile (512-channel 3x3 layers): then two.
    const int slots = 256 * (dma1 == 1 ? 4 : (pl.tm == 128 && pl.tn == 128) ? 2 : 3);
    const int rounds = slots / tiles >= 4 ? 1 : 2;
    int splits = rounds * slots / tiles;
    int max_splits = pl.ksteps_total / 4; if (max_splits < 1) max_splits = 1;      // >= 4 K-steps per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    pl.ksteps_per_split = (pl.ksteps_total + splits - 1) / splits;
    pl.splits = (pl.ksteps_total + pl.ksteps_per_split - 1) / pl.ksteps_per_split;
    pl.ws_bytes = dir_align_up(sizeof(float) * (size_t)pl.splits * Cout * RS * Cin, 256);
    return pl;
}

template <int TM, int TN>
void wg_launch(const WgP& p, int nblocks, hipStream_t s) {
    constexpr int lds = 2 * (TM + TN) * WG_ROWB;
    if (p.simple) hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN, true>), dim3(nblocks), dim3(DIR_TPB), lds, s, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN, false>), dim3(nblocks), dim3(DIR_TPB), lds, s, p);
}

}  // namespace

// (also used by dir_conv_wgrad3.hip) dw[i] = sum over splits, in split order, of part[split][i]; n % 4 == 0
extern "C" int dir_conv_wgrad_reduce_splits(const float* part, int splits, size_t n, float* dw, dir_stream_t stream) {
    DIR_RETURN_IF(!part || !dw || splits <= 0 || n == 0 || (n & 3), DIR_EINVAL);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(dir_cdiv((long long)n / 4, RD_COLS)), dim3(DIR_TPB), 0, dir_s(stream), part, splits, n, dw);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// form (DIR_WGRAD_*): 0 = the product's choice, 1 = the transposing kernel, 2 / 3 = the LDS-DMA + transposing-read 1x1 kernel with one /
// two stages. Returns the plan's dma1 code, or -1 when the forced form does not take the geometry.
static int wg_dma1_form(int Cin, int Cout, int R, int S, int stride, int pad, int form) {
    const bool ok = R == 1 && S == 1 && stride == 1 && pad == 0 && Cin % 128 == 0 && Cout % 128 == 0;
    if (form == DIR_WGRAD_AUTO) return ok ? W1_FORM : 0;
    if (form == DIR_WGRAD_TRANSPOSE) return 0;
    if (form == DIR_WGRAD_DMA1 || form == DIR_WGRAD_DMA2) return ok ? form - 1 : -1;
    return -1;
}

extern "C" size_t dir_conv_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int form) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin % 64 || Cout % 64 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return 0;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const int dma1 = wg_dma1_form(Cin, Cout, R, S, stride, pad, form);
    if (dma1 < 0) return 0;
    return wg_plan((long long)N * Ho * Wo, Cin, Cout, R * S, dma1).ws_bytes;
}

// table: device [nlayers][4] int64 = (partials [splits][n] float32, splits, n (% 4 == 0), dw [n] float32) per layer
extern "C" int dir_conv_wgrad_reduce_batched(const void* table, int nlayers, dir_stream_t stream) {
    DIR_RETURN_IF(!table || nlayers <= 0 || nlayers > 65535, DIR_EINVAL);
    hipLaunchKernelGGL(conv_wgrad_reduce_batched_kernel, dim3(512, nlayers), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const long long*>(table));
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// dw == nullptr: the split-K partials stay in `workspace` ([splits][Cout * R * S * Cin] float32 from its start) for a later reduction
static int wgrad_impl(const void* dy, const void* x, float* dw, int* splits_out, int N, int H, int W, int Cin, int Cout,
                      int R, int S, int stride, int pad, int form, void* workspace, size_t workspace_bytes,
                      dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !x || !workspace, DIR_EINVAL);
    DIR_RETURN_IF(form < DIR_WGRAD_AUTO || form > DIR_WGRAD_DMA2, DIR_EINVAL);
    DIR_RETURN_IF(N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0, DIR_EINVAL);
    DIR_RETURN_IF(Cin % 64 != 0 || Cout % 64 != 0, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(dy) || !dir_aligned16(x) || (dw && !dir_aligned16(dw)) || (reinterpret_cast<uintptr_t>(workspace) & 255u), DIR_EINVAL);
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    DIR_RETURN_IF(Ho <= 0 || Wo <= 0, DIR_EINVAL);
    const long long M = (long long)N * Ho * Wo;
    DIR_RETURN_IF(M >= (1ll << 24) || (long long)N * H * W * Cin >= (1ll << 30) || M * Cout >= (1ll << 30), DIR_EUNSUPPORTED);   // 32-bit byte offsets
    const int dma1 = wg_dma1_form(Cin, Cout, R, S, stride, pad, form);
    DIR_RETURN_IF(dma1 < 0, DIR_EUNSUPPORTED);
    const WgPlan pl = wg_plan(M, Cin, Cout, R * S, dma1);
    DIR_RETURN_IF(workspace_bytes < pl.ws_bytes, DIR_EWORKSPACE);
    WgP p;
    p.dy = static_cast<const uint16_t*>(dy); p.x = static_cast<const uint16_t*>(x); p.part = static_cast<float*>(workspace);
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = (int)M; p.RS = R * S; p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.splits = pl.splits;
    p.ksteps_total = pl.ksteps_total; p.ksteps_per_split = pl.ksteps_per_split;
    p.inv_wo = 1.0f / (float)Wo; p.inv_ho = 1.0f / (float)Ho;
    p.simple = (R == 1 && S == 1 && stride == 1 && pad == 0) ? 1 : 0;
    {   // constants of the incremental im2col walk (bytes of bf16 X)
        const int dw = WG_BK % Wo, dh = (WG_BK / Wo) % Ho, dn = WG_BK / (Wo * Ho);
        p.WoS = Wo * stride; p.HoS = Ho * stride;
        p.c1 = (stride * W - Wo * stride) * Cin * 2;
        p.c2 = (H * W - Ho * stride * W) * Cin * 2;
        p.dws64 = dw * stride; p.dhs64 = dh * stride; p.dn64 = dn;
        p.c64 = (dn * H * W + dh * stride * W + dw * stride) * Cin * 2;
        p.cstep = stride * Cin * 2;
    }
    const int nblocks = pl.splits * pl.tiles_m * pl.tiles_n * p.RS;
    hipStream_t s = dir_s(stream);
    if (pl.dma1 == 1) hipLaunchKernelGGL(conv_wgrad1_dma_kernel<1>, dim3(nblocks), dim3(DIR_TPB), W1_STAGE, s, p);
    else if (pl.dma1 == 2) {
        DIR_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad1_dma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W1_STAGE));
        hipLaunchKernelGGL(conv_wgrad1_dma_kernel<2>, dim3(nblocks), dim3(DIR_TPB), 2 * W1_STAGE, s, p);
    }
    else if (pl.tm == 128 && pl.tn == 128) wg_launch<128, 128>(p, nblocks, s);
    else if (pl.tm == 128) wg_launch<128, 64>(p, nblocks, s);
    else if (pl.tn == 128) wg_launch<64, 128>(p, nblocks, s);
    else wg_launch<64, 64>(p, nblocks, s);
    DIR_LAUNCH_CHECK();
    if (splits_out) *splits_out = pl.splits;
    if (!dw) return DIR_OK;
    const size_t n = (size_t)Cout * p.RS * Cin;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(dir_cdiv((long long)n / 4, RD_COLS)), dim3(DIR_TPB), 0, s, p.part, pl.splits, n, dw);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_conv_wgrad(const void* dy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout,
                              int R, int S, int stride, int pad, int form, void* workspace, size_t workspace_bytes,
                              dir_stream_t stream) {
    DIR_RETURN_IF(!dw, DIR_EINVAL);
    return wgrad_impl(dy, x, dw, nullptr, N, H, W, Cin, Cout, R, S, stride, pad, form, workspace, workspace_bytes, stream);
}

extern "C" int dir_conv_wgrad_partials(const void* dy, const void* x, int* splits, int N, int H, int W, int Cin, int Cout,
                                       int R, int S, int stride, int pad, int form, void* workspace, size_t workspace_bytes,
                                       dir_stream_t stream) {
    DIR_RETURN_IF(!splits, DIR_EINVAL);
    return wgrad_impl(dy, x, nullptr, splits, N, H, W, Cin, Cout, R, S, stride, pad, form, workspace, workspace_bytes, stream);
}
