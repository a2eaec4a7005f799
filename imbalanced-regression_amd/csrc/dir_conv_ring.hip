// Persistent ring-pipelined MFMA implicit-GEMM convolution for MI355X / gfx950: the K loop of dir_conv.hip restructured so that
// load, MFMA and store phases OVERLAP inside one workgroup instead of adding up (profiles/r02_conv_phase_ablation.txt).
// Replaces nn.Conv2d forward / stride-1 data gradient of imdb-wiki-dir/resnet.py:46-51 for the 128-wide layers.
//
//   * ONE workgroup of 512 threads = 8 wavefronts per CU (two per SIMD), persistent: it walks its XCD's range of 128 x 128 output
//     tiles, so tile i's store drain and tile i + 1's loads and MFMAs are in flight together.
//   * Fixed wave ROLES, because gfx950 counts loads and stores of a wavefront in ONE in-order counter (a wait for a load also waits
//     for every older store, and a store is acknowledged microseconds after issue when HBM writes are saturated):
//       wavefronts 0-3 "loaders": issue the LDS-DMA (buffer_load_dwordx4 ... lds, inline asm: 1 KB per wave-instruction) of every
//         K-step into a ring of NSLOT 32 KB stages, NSLOT - 1 steps ahead, across tile boundaries, and wait for them with
//         COUNTED vmcnt — their counter never holds a store;
//       wavefronts 4-7 "storers": run the previous tile's whole epilogue (bf16 staging tile -> fused operands -> 16-byte row
//         stores -> BatchNorm partial sums) in slices spread over the next tile's first K-steps — their counter never gates a
//         K-step: "matrix beside memory" on every SIMD.
//     All eight multiply: a wavefront owns a 64 x 32 sub-tile (two 32x32x16 bf16 MFMA accumulators).
//   * One raw s_barrier per K-step (it does not drain LDS-DMA) + one per tile (staging tile hand-over).
//   * LDS: NSLOT x (A 128 rows + B 128 rows of 128 B, XOR-swizzled on the DMA source side like dir_conv.hip) + the bf16
//     staging tile [128][272 B] + 4 KB of column partials.
// Accumulation order per output element (K-steps in order, four 16-wide MFMAs per step) and the order of the statistics sums are
// those of dir_conv.hip's kernels: results and BatchNorm partials are bit-identical to them.
#include <type_traits>
#include "dir_conv_shared.h"

namespace {

constexpr int RG_THREADS = 512;
constexpr int RG_BN = 128;
constexpr int RG_A_BYTES = CV_BM * CV_ROWB;                 // 16 KB
constexpr int RG_SLOT = 2 * RG_A_BYTES;                     // A + B stage: 32 KB
constexpr int RG_CS_STRIDE = RG_BN * 2 + 16;                // 272-byte staging rows (as cv_epilogue<128>)
constexpr int RG_CS_BYTES = CV_BM * RG_CS_STRIDE;           // 34 816
constexpr int RG_SS_BYTES = 4 * 2 * RG_BN * 4;              // 4 KB

__device__ __forceinline__ void rg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS-DMA piece (64 lanes x 16 B -> 1 KB of LDS at the wave-uniform address `lds_addr`), inline asm so that the compiler neither
// counts it nor drains it. The s_nop covers the M0 write and an SGPR operand a scalar instruction has just written.
__device__ __forceinline__ void rg_dma16(cp_u32x4 rs, uint32_t lds_addr, int voffset, int soffset) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 3\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voffset), "s"(rs), "s"(soffset) : "memory");
}
template <int N> __device__ __forceinline__ void rg_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// KTB: K-steps per tile as the kernel's control-flow shape: 1 (KT == 1), 2 (KT == 2), 4 (KT >= 4) — the storers' four epilogue
// slices ride on the first min(KT, 4) K-steps as straight-line code, so that the compiler's own counted waits for the
// storers' operand loads stay exact (a `switch` inside a loop would degrade them to vmcnt(0) = a store round trip per slice).
template <int NSLOT, int KTB, bool LEAN>
__global__ void __launch_bounds__(RG_THREADS)
conv_ring_kernel(ConvP p, int ntiles, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Cs = smem + NSLOT * RG_SLOT;
    float* Ss = reinterpret_cast<float*>(smem + NSLOT * RG_SLOT + RG_CS_BYTES);
    typedef __attribute__((address_space(3))) unsigned char* lds_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_t)smem;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool loader = wave < 4;
    const int gw = wave & 3;                                    // index inside the role group

    // ---- this workgroup's tiles: the linear tile space (N tiles of one M tile adjacent) is cut into 8 contiguous chunks, one per
    // XCD (the hardware deals workgroup b to XCD b % 8); the workgroups of an XCD take its tiles round-robin, so at any moment
    // they work on neighbouring tiles: the A rows the N tiles share and the weights are L2 hits.
    const int nx = gridDim.x >> 3;                              // workgroups per XCD (grid is a multiple of 8)
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int t_first = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + wi;
    const int t_count = tq + (xcd < tr ? 1 : 0);
    const int my_tiles = t_count > wi ? (t_count - wi + nx - 1) / nx : 0;
    if (my_tiles == 0) return;
    const int KT = p.KT;

    // ---- MFMA roles: wavefront tile 64 (pixels) x 32 (channels)
    const int wm = wave & 1, wn = wave >> 1;
    const int frow = lane & 31, fhalf = lane >> 5;
    uint32_t af[2][4], bfo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int row = wm * 64 + mi * 32 + frow;
            af[mi][kk] = row * CV_ROWB + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) << 4);
        }
        const int row = wn * 32 + frow;
        bfo[kk] = RG_A_BYTES + row * CV_ROWB + (((kk * 2 + fhalf) ^ ((row >> 1) & 7)) << 4);
    }
    f32x16 acc[2][1];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][0][e] = 0.0f;

    // ---- loader state (wavefronts 0-3; as conv_igemm_dma_kernel: piece i of wave gw = rows gw*32 + 8 i + (lane >> 3) of the
    // A tile and of the B tile, the lane's LDS slot = physical chunk lane & 7 = logical chunk (lane & 7) ^ ((row >> 1) & 7))
    const int lr = lane >> 3, lc = lane & 7;
    const int K = KT * CV_BK;
    const cp_u32x4 rs_x = cp_rsrc(p.x, (uint32_t)(p.N * p.H * p.W) * (uint32_t)p.Cin * 2u);
    const cp_u32x4 rs_w = cp_rsrc(p.w, (uint32_t)p.Cout * (uint32_t)K * 2u);
    int aoff[4], woff[4];
    uint32_t amask[4];
    int ld_j = 0, ld_k = 0, ld_c = 0, ld_tap = 0, ld_r = 0, ld_s = 0;
    int issued = 0, issue_slot = 0;
    const int total = my_tiles * KT;
    auto ld_setup = [&](int j) {
        const int lin = t_first + j * nx;
        const int mt = lin / p.ntn, nt = lin - mt * p.ntn;
        const int m0 = mt * CV_BM, n0 = nt * RG_BN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chunk = lc ^ ((lane >> 4) | ((i & 1) << 2));
            const int m = m0 + gw * 32 + 8 * i + lr;
            aoff[i] = 0; amask[i] = 0;
            if (m < p.M) {
                if (p.simple) { aoff[i] = (m * p.Cin + chunk * 8) * 2; amask[i] = 1u; }
                else {
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
            woff[i] = ((n0 + gw * 32 + 8 * i + lr) * K + chunk * 8) * 2;
        }
    };
    auto ld_issue = [&]() {                                     // the next stage of the (tile, K-step) stream -> ring slot issue_slot
        const int koff = ((ld_r * p.W + ld_s) * p.Cin + ld_c * CV_BK) * 2;
        const uint32_t bit = 1u << ld_tap;
        const uint32_t abase = lds0 + (uint32_t)(issue_slot * RG_SLOT + gw * 4096);
        rg_dma16(rs_x, abase + 0 * 1024, (amask[0] & bit) ? aoff[0] + koff : CV_OOB, 0);
        rg_dma16(rs_x, abase + 1 * 1024, (amask[1] & bit) ? aoff[1] + koff : CV_OOB, 0);
        rg_dma16(rs_x, abase + 2 * 1024, (amask[2] & bit) ? aoff[2] + koff : CV_OOB, 0);
        rg_dma16(rs_x, abase + 3 * 1024, (amask[3] & bit) ? aoff[3] + koff : CV_OOB, 0);
        const int wso = ld_k * CV_BK * 2;
        const uint32_t bbase = abase + RG_A_BYTES;
        rg_dma16(rs_w, bbase + 0 * 1024, woff[0], wso);
        rg_dma16(rs_w, bbase + 1 * 1024, woff[1], wso);
        rg_dma16(rs_w, bbase + 2 * 1024, woff[2], wso);
        rg_dma16(rs_w, bbase + 3 * 1024, woff[3], wso);
        ++issued;
        if (++issue_slot == NSLOT) issue_slot = 0;
        if (++ld_k == KT) {
            ld_k = 0; ld_c = 0; ld_tap = 0; ld_r = 0; ld_s = 0;
            if (++ld_j < my_tiles) ld_setup(ld_j);
        } else if (++ld_c == p.cpk) { ld_c = 0; ++ld_tap; if (++ld_s == p.S) { ld_s = 0; ++ld_r; } }
    };

    // ---- storer state (wavefronts 4-7): thread (srow, sch) of the 256 owns the 16-byte chunk sch of rows srow + 16 i, i < 8
    const int st = t & 255;
    const int srow = st >> 4, sch = st & 15;
    const unsigned char* cs = Cs + srow * RG_CS_STRIDE + sch * 16;
    const uint32_t ybytes = p.o2 ? (uint32_t)p.N * (uint32_t)p.OH * (uint32_t)p.OW * (uint32_t)p.Cout * 2u : (uint32_t)p.M * (uint32_t)p.Cout * 2u;
    const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, (int)ybytes, 0x00020000);
    const uint16_t* addp = p.addend ? p.addend : p.addend2;
    const uint32_t addbytes = p.addend ? ybytes : (uint32_t)p.N * (uint32_t)(p.Ho >> 1) * (uint32_t)(p.Wo >> 1) * (uint32_t)p.Cout * 2u;
    const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(addp), (short)0, addp ? (int)addbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bnx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.bnx), (short)0, p.bnx ? (int)ybytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.mask_bits), (short)0, p.mask_bits ? (int)(ybytes >> 4) : 0, 0x00020000);
    const bool fwd_stats = p.stats && !p.bnx;
    const bool decode = p.o2 || p.addend2;
    uint32_t f_ob[8], f_oa[8], f_bits[8];
    cv_u32x4 f_add[8], f_bnx[8];
    float maf[8], mbf[8], ssum[8], ssq[8];
    int e_mt = 0, e_n0 = 0;                                     // tile whose epilogue is pending (statistics row / column base)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; ssq[j] = 0.0f; maf[j] = 0.0f; mbf[j] = 0.0f; f_ob[j] = 0u; f_oa[j] = 0u; f_bits[j] = 0u; }

    // offsets (and, fused launches, operand loads: all eight rows in flight at once) of tile (mt, n0)
    auto epi_issue = [&](int mt, int n0) {
        const int m0 = mt * CV_BM;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + srow + 16 * i;
            uint32_t ob = (uint32_t)(((size_t)m * p.Cout + n0 + sch * 8) * 2u), oa = ob;
            if constexpr (!LEAN) {
                if (decode) {
                    int q1 = (int)((float)m * p.inv_wo), wo = m - q1 * p.Wo;
                    if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
                    int n = (int)((float)q1 * p.inv_ho), ho = q1 - n * p.Ho;
                    if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }
                    const bool valid = m < p.M;
                    if (p.o2) {
                        ob = valid ? (uint32_t)((((size_t)n * p.OH + 2 * ho + p.o_a) * p.OW + 2 * wo + p.o_b) * p.Cout + n0 + sch * 8) * 2u : (uint32_t)CV_OOB;
                        oa = ob;
                    }
                    if (p.addend2)
                        oa = (valid && !((ho | wo) & 1)) ? (uint32_t)(((size_t)(n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + n0 + sch * 8) * 2u
                                                         : (uint32_t)CV_OOB;
                }
            }
            f_ob[i] = ob; f_oa[i] = oa;
            if constexpr (!LEAN) {
                f_add[i] = __builtin_amdgcn_raw_buffer_load_b128(r_add, (int)oa, 0, 2);
                f_bits[i] = __builtin_amdgcn_raw_buffer_load_b8(r_bits, (int)(ob >> 4), 0, 0);
                f_bnx[i] = __builtin_amdgcn_raw_buffer_load_b128(r_bnx, (int)ob, 0, 2);
            }
        }
        if constexpr (!LEAN) {
            if (p.bnx && p.bn_gamma) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {                   // same expression as dir_bn.hip's bn_mask_coef (= the forward's coefficients)
                    const int ch = n0 + sch * 8 + j;
                    const double gm = (double)p.bn_gamma[ch], rs = (double)p.bn_rstd[ch];
                    maf[j] = (float)(gm * rs);
                    mbf[j] = (float)((double)p.bn_beta[ch] - (double)p.bn_mean[ch] * gm * rs);
                }
            }
        }
    };
    // rows 2 hh, 2 hh + 1 of the pending tile: staging tile -> fused arithmetic -> 16-byte row stores (+ partial sums)
    auto epi_slice = [&](auto hh_c) {
        constexpr int hh = decltype(hh_c)::value;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            constexpr int dummy = 0; (void)dummy;
            const int i = hh * 2 + ii;
            const cv_u32x4 cc = *reinterpret_cast<const cv_u32x4*>(cs + i * 16 * RG_CS_STRIDE);
            uint32_t cw[4] = {cc.x, cc.y, cc.z, cc.w};
            if (LEAN ? (p.stats != nullptr) : fwd_stats) {
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2) {
                    const float f0 = __uint_as_float(cw[q2] << 16), f1 = __uint_as_float(cw[q2] & 0xffff0000u);
                    ssum[2 * q2] += f0; ssq[2 * q2] += f0 * f0; ssum[2 * q2 + 1] += f1; ssq[2 * q2 + 1] += f1 * f1;
                }
            }
            if constexpr (LEAN) {
                __builtin_amdgcn_raw_buffer_store_b128(cv_u32x4{cw[0], cw[1], cw[2], cw[3]}, r_y, (int)f_ob[i], 0, CV_AUX_SC1_NT);
            } else {
                if (addp) {                                     // y = bf16(bf16(conv) + addend), like an eager add kernel
                    const uint32_t aw[4] = {f_add[i].x, f_add[i].y, f_add[i].z, f_add[i].w};
                    const bool has = p.addend || f_oa[i] != (uint32_t)CV_OOB;
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const uint32_t sum = cv_pack_bf16(__uint_as_float(cw[q2] << 16) + __uint_as_float(aw[q2] << 16),
                                                          __uint_as_float(cw[q2] & 0xffff0000u) + __uint_as_float(aw[q2] & 0xffff0000u));
                        cw[q2] = has ? sum : cw[q2];
                    }
                }
                if (p.mask_bits) {                              // ReLU backward, from the forward's bit per element
                    const uint32_t bb = f_bits[i];
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        if (!(bb & (1u << (2 * q2)))) cw[q2] &= 0xffff0000u;
                        if (!(bb & (2u << (2 * q2)))) cw[q2] &= 0x0000ffffu;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(cv_u32x4{cw[0], cw[1], cw[2], cw[3]}, r_y, (int)f_ob[i], 0, 2);
                if (p.bnx) {
                    const uint32_t xw[4] = {f_bnx[i].x, f_bnx[i].y, f_bnx[i].z, f_bnx[i].w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        float g0 = __uint_as_float(cw[q2] << 16), g1 = __uint_as_float(cw[q2] & 0xffff0000u);
                        const float x0 = __uint_as_float(xw[q2] << 16), x1 = __uint_as_float(xw[q2] & 0xffff0000u);
                        if (p.bn_gamma) {
                            if (!(x0 * maf[2 * q2] + mbf[2 * q2] > 0.0f)) g0 = 0.0f;
                            if (!(x1 * maf[2 * q2 + 1] + mbf[2 * q2 + 1] > 0.0f)) g1 = 0.0f;
                        }
                        ssum[2 * q2] += g0; ssq[2 * q2] += g0 * x0; ssum[2 * q2 + 1] += g1; ssq[2 * q2 + 1] += g1 * x1;
                    }
                }
            }
        }
        if constexpr (hh == 3) {                                // column partials of this wavefront -> Ss (order of cv_epilogue_stats)
            if (p.stats) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int o = 16; o < DIR_WAVE; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, DIR_WAVE); ssq[j] += __shfl_xor(ssq[j], o, DIR_WAVE); }
                }
                if (lane < 16) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { Ss[(gw * 2 + 0) * RG_BN + lane * 8 + j] = ssum[j]; Ss[(gw * 2 + 1) * RG_BN + lane * 8 + j] = ssq[j]; }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; ssq[j] = 0.0f; }
        }
    };
    auto epi_stats_out = [&]() {                                // after a barrier behind slice 3: one thread per (which, column)
        if (p.stats) {
            const int which = st >> 7, col = st & 127;
            const float v = (Ss[(0 * 2 + which) * RG_BN + col] + Ss[(1 * 2 + which) * RG_BN + col]) + (Ss[(2 * 2 + which) * RG_BN + col] + Ss[(3 * 2 + which) * RG_BN + col]);
            p.stats[((size_t)e_mt * 2 + which) * p.Cout + e_n0 + col] = v;
        }
    };

    // ---- K-step machinery
    int g = 0, use_slot = 0;                                    // consumer step counter / its ring slot
    auto step_sync = [&]() {                                    // stage g has landed for everybody; the slot of stage g - 1 is free
        if (loader) {
            const int newer = issued - g - 1;                   // stages issued behind stage g: 8 DMA pieces each
            if (NSLOT >= 4 && newer >= 2) rg_vmwait<16>();
            else if (newer >= 1) rg_vmwait<8>();
            else rg_vmwait<0>();
        }
        rg_barrier();
        if (loader && issued < total) { if (!(dbg & 2)) ld_issue(); else { ++issued; } }
    };
    auto step_mfma = [&]() {
        const unsigned char* sb = smem + use_slot * RG_SLOT;
        if (!(dbg & 1))
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sb + af[0][kk]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sb + af[1][kk]);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(sb + bfo[kk]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a0, acc[0][0], 0, 0, 0);   // D'[channel][pixel]
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a1, acc[1][0], 0, 0, 0);
        }
        ++g;
        if (++use_slot == NSLOT) use_slot = 0;
    };
    typedef std::integral_constant<int, 0> S0; typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2; typedef std::integral_constant<int, 3> S3;

    // ---- prologue: the first NSLOT - 1 stages
    if (loader) {
        ld_setup(0);
#pragma unroll
        for (int s = 0; s < NSLOT - 1; ++s)
            if (issued < total) ld_issue();
    }

    for (int j = 0; j < my_tiles; ++j) {
        const bool pend = !loader && j > 0;                     // storers: the previous tile's epilogue rides on this tile's K-steps
        if constexpr (KTB == 1) {
            step_sync();
            if (pend) { epi_slice(S0{}); epi_slice(S1{}); epi_slice(S2{}); epi_slice(S3{}); }
            step_mfma();
        } else if constexpr (KTB == 2) {
            step_sync();
            if (pend) { epi_slice(S0{}); epi_slice(S1{}); }
            step_mfma();
            step_sync();
            if (pend) { epi_slice(S2{}); epi_slice(S3{}); }
            step_mfma();
        } else {
            step_sync();
            if (pend) epi_slice(S0{});
            step_mfma();
            step_sync();
            if (pend) epi_slice(S1{});
            step_mfma();
            step_sync();
            if (pend) epi_slice(S2{});
            step_mfma();
            step_sync();
            if (pend) epi_slice(S3{});
            step_mfma();
            for (int k = 4; k < KT; ++k) { step_sync(); step_mfma(); }
        }
        // ---- tile end: this tile's operand loads go out (they fly during the hand-over), the previous tile's statistics leave,
        // the accumulators become the staging tile
        const int lin = t_first + j * nx;
        const int mt = lin / p.ntn, n0 = (lin - mt * p.ntn) * RG_BN;
        if (!loader) epi_issue(mt, n0);
        rg_barrier();                                           // every slice of the previous tile has read the staging tile / written Ss
        if (pend) epi_stats_out();
        e_mt = mt; e_n0 = n0;
        cv_stage_acc<2, 1, RG_CS_STRIDE>(acc, Cs + (wm * 64 + frow) * RG_CS_STRIDE + (wn * 32 + 4 * fhalf) * 2);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][0][e] = 0.0f;
    }
    // ---- drain: the last tile's epilogue
    rg_barrier();
    if (!loader) { epi_slice(S0{}); epi_slice(S1{}); epi_slice(S2{}); epi_slice(S3{}); }
    rg_barrier();
    if (!loader) epi_stats_out();
}

int rg_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        n &= ~7;
        if (n < 8) n = 8;
    }
    return n;
}

}  // namespace

// Which launches the ring kernel takes: 128-wide output tiles, K loops of 1, 2 or >= 4 steps, no tensor ReLU mask (the bit mask
// is taken), not both shortcut-gradient streams at once.
bool conv_ring_takes(const ConvP& p) {
    if (p.Cout % RG_BN != 0) return false;
    if (p.KT == 3 || p.KT < 1) return false;
    if (p.mask || (p.addend && p.addend2)) return false;
    return true;
}

int conv_ring_launch(const ConvP& p_in, int dbg, hipStream_t s) {
    if (!conv_ring_takes(p_in)) return -1;
    ConvP p = p_in;
    p.ntn = p.Cout / RG_BN;
    const int mtiles = (p.M + CV_BM - 1) / CV_BM;
    const int ntiles = mtiles * p.ntn;
    int grid = rg_num_cus();
    if (ntiles < grid) grid = (ntiles + 7) & ~7;
    const bool lean = !p.addend && !p.addend2 && !p.mask && !p.mask_bits && !p.bnx && !p.o2;
    const int ktb = p.KT == 1 ? 1 : (p.KT == 2 ? 2 : 4);
    constexpr int NS = 3;
    constexpr int lds = NS * RG_SLOT + RG_CS_BYTES + RG_SS_BYTES;
#define RG_GO(KTB_, LEAN_)                                                                                                     \
    {                                                                                                                         \
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ring_kernel<NS, KTB_, LEAN_>),       \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);                \
        (void)once;                                                                                                           \
        hipLaunchKernelGGL((conv_ring_kernel<NS, KTB_, LEAN_>), dim3(grid), dim3(RG_THREADS), lds, s, p, ntiles, dbg);             \
    }
    if (lean) { if (ktb == 1) RG_GO(1, true) else if (ktb == 2) RG_GO(2, true) else RG_GO(4, true) }
    else      { if (ktb == 1) RG_GO(1, false) else if (ktb == 2) RG_GO(2, false) else RG_GO(4, false) }
#undef RG_GO
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
