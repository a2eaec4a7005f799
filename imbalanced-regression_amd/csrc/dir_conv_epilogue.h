// Epilogue of the MFMA implicit-GEMM convolution kernels (dir_conv.hip): bf16 staging tile -> 16-byte row stores with the
// fused statistics / addend / ReLU-mask / BatchNorm-backward options. One definition, so that every kernel variant stores and sums in the same order
// (outputs and partial-sum lists are bit-identical across variants: tools/check_conv_variants.py).
#pragma once
#include "dir_common.h"
#include "dir_conv_shared.h"

// Epilogue shared by the K-loop variants: accumulators -> bf16 staging tile in LDS (the K-loop buffers are free: the caller
// has passed a barrier after its last fragment read) -> 16-B row stores with the fused statistics / addend / mask options.
// The global operands of the store loop (shortcut gradient — dense or compact stride-2 —, ReLU mask as tensor or bits, BatchNorm
// input for the fused backward sums) are fetched in batches of rows, one batch AHEAD of the arithmetic and stores that consume
// them and the first batch before the barrier that publishes the staging tile (see cv_epilogue). Written as load-then-use inside
// the loop, each of them costs a full memory round trip (the compiler keeps `s_waitcnt vmcnt(0)` right behind every load: found
// in the ISA), 8 x 3 serial round trips per workgroup in the data-gradient launches.
// Column partials of a tile's statistics: over the lanes / wavefronts that share a channel chunk, in a fixed order -> stats[mt]
template <int BN>
__device__ __forceinline__ void cv_epilogue_stats(const ConvP& p, float (&ssum)[8], float (&ssq)[8], float* Ss, int t, int n0, int mt) {
    constexpr int CPR = BN / 8;
    const int lane = t & 63, wave = t >> 6;
    if (p.stats) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int o = CPR; o < DIR_WAVE; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, DIR_WAVE); ssq[j] += __shfl_xor(ssq[j], o, DIR_WAVE); }
        }
        if (lane < CPR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { Ss[(wave * 2 + 0) * BN + lane * 8 + j] = ssum[j]; Ss[(wave * 2 + 1) * BN + lane * 8 + j] = ssq[j]; }
        }
        __syncthreads();
        if (t < 2 * BN) {                                       // one thread per (which, column)
            const int which = t / BN, col = t - which * BN;
            const float v = (Ss[(0 * 2 + which) * BN + col] + Ss[(1 * 2 + which) * BN + col]) + (Ss[(2 * 2 + which) * BN + col] + Ss[(3 * 2 + which) * BN + col]);
            p.stats[((size_t)mt * 2 + which) * p.Cout + n0 + col] = v;
        }
    }
}

// LEAN: the launch has no fused operand (no addend / compact addend / ReLU mask / BatchNorm-backward sums, dense rows): the plain
// forward. Those code paths then do not exist in the kernel, which brings its register count under 128 = a fourth wavefront per SIMD.
// cv_epilogue_staged: everything after the accumulators have been written to the staging tile Cs = smem [128 pixels][BN * 2 + 16 bytes] (by
// cv_stage_acc, whatever the caller's wavefront tiling) and BEFORE the barrier that publishes it: t = 0 .. 255 (the 256 threads that own the tile).
// Barriers executed: one (publishing the tile) + one more when p.stats is given (cv_epilogue_stats) — a caller whose workgroup has further
// wavefronts must execute the same number.
template <int BN, bool LEAN = false, int NBATCH = 2>
__device__ __forceinline__ void cv_epilogue_staged(const ConvP& p, unsigned char* smem, int t, int m0, int n0, int mt) {
    constexpr int CS_STRIDE = BN * 2 + 16;                      // bytes per staging row
    unsigned char* Cs = smem;                                   // [128][CS_STRIDE] (<= 34 KB)
    float* Ss = reinterpret_cast<float*>(smem + CV_BM * CS_STRIDE);   // [4 waves][2][BN] column partials (<= 4 KB)
    constexpr int CPR = BN / 8;                                 // 16-B chunks per C row
    constexpr int RPI = DIR_TPB / CPR;                          // rows per pass of the workgroup
    constexpr int NIT = CV_BM / RPI;                            // passes = rows per thread (8 or 4)
    const int srow = t / CPR, sch = t - srow * CPR;
    const unsigned char* cs = Cs + srow * CS_STRIDE + sch * 16;
    const size_t go0 = (size_t)(m0 + srow) * p.Cout + n0 + sch * 8;
    const size_t gstep = (size_t)RPI * p.Cout;
    const bool full = m0 + CV_BM <= p.M;
    if constexpr (LEAN) {
        const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, (int)((uint32_t)p.M * (uint32_t)p.Cout * 2u), 0x00020000);
        __syncthreads();
        float ssum[8], ssq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; ssq[j] = 0.0f; }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (full || m0 + srow + i * RPI < p.M) {
                const uint4 c = *reinterpret_cast<const uint4*>(cs + i * RPI * CS_STRIDE);
                if (p.stats) {
                    const uint32_t sw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const float f0 = __uint_as_float(sw[q2] << 16), f1 = __uint_as_float(sw[q2] & 0xffff0000u);
                        ssum[2 * q2] += f0; ssq[2 * q2] += f0 * f0; ssum[2 * q2 + 1] += f1; ssq[2 * q2 + 1] += f1 * f1;
                    }
                }
                // `sc1 nt`: written through to the memory side (where the 256 MB Infinity Cache keeps it for the BatchNorm pass that reads it
                // next) and streamed past the L2, whose lines are better spent on the operands the neighbouring tiles share. Same-box A/B,
                // train step / epoch-tail forward: plain store -> `nt` -0.18 / -0.16 ms, `nt` -> `sc1 nt` another -1.17 / -0.37 ms.
                __builtin_amdgcn_raw_buffer_store_b128(cv_u32x4{c.x, c.y, c.z, c.w}, r_y, (int)(((uint32_t)go0 + (uint32_t)i * (uint32_t)gstep) * 2u), 0, CV_AUX_SC1_NT);
            }
        }
        cv_epilogue_stats<BN>(p, ssum, ssq, Ss, t, n0, mt);
        return;
    }
    const bool fwd_stats = p.stats && !p.bnx;
    const bool decode = p.o2 || p.addend2;                      // rows need their (n, ho, wo)

    // fused BatchNorm-backward partials: (sum g, sum g * bnx) of the gradient as stored, optionally under the recomputed ReLU mask
    float maf[8], mbf[8];
    auto mask_coefficients = [&]() {
        if (p.bnx && p.bn_gamma) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {                       // same expression as dir_bn.hip's bn_mask_coef (= the forward's coefficients)
                const int ch = n0 + sch * 8 + j;
                const double gm = (double)p.bn_gamma[ch], rs = (double)p.bn_rstd[ch];
                maf[j] = (float)(gm * rs);
                mbf[j] = (float)((double)p.bn_beta[ch] - (double)p.bn_mean[ch] * gm * rs);
            }
        }
    };
    // BatchNorm statistics of the ROUNDED outputs (what the following BatchNorm reads): this thread's 8 channels over the rows
    // it stores, then over the lanes / wavefronts that share the channel chunk, in a fixed order
    float ssum[8], ssq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; ssq[j] = 0.0f; }

    constexpr int HALF = NIT / NBATCH;                          // rows per batch (NBATCH = 4: fewer registers, for the 4-per-CU variant)
    // Every launch but the one with BOTH shortcut-gradient streams: the thread's rows in batches, software-pipelined — batch b + 1's
    // operand loads are issued BEFORE batch b's arithmetic and stores — and free of control flow around the memory instructions: loads and stores are buffer
    // instructions, the descriptor of an operand the launch does not have is EMPTY (out-of-range lanes read zeros / store nothing,
    // without a memory access) and the descriptors end at row M, which also drops the rows of a ragged tile. The wait counter is
    // in issue order, so the compiler can then wait for "everything but the newest N" and N covers the stores: nothing in the
    // epilogue waits for a store to be acknowledged. With a conditional store or load in between it has to assume the shortest
    // path and emits `s_waitcnt vmcnt(0)` — in the general loop below that is one store round trip per row, eight in a row per
    // workgroup tile (found in the ISA; same-box A/B: -0.19 ms per train step for the dense launches, -0.07 ms more for the scattered ones).
    if (!(p.addend && p.addend2)) {
        // y-shaped operands span the whole result tensor (the scattered rows of a stride-2 class launch index it like y itself)
        const uint32_t ybytes = p.o2 ? (uint32_t)p.N * (uint32_t)p.OH * (uint32_t)p.OW * (uint32_t)p.Cout * 2u : (uint32_t)p.M * (uint32_t)p.Cout * 2u;
        const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, (int)ybytes, 0x00020000);
        // ONE shortcut-gradient stream: the dense one (addend, same offsets as y) or the compact stride-2 one (addend2, own offsets)
        const uint16_t* addp = p.addend ? p.addend : p.addend2;
        const uint32_t addbytes = p.addend ? ybytes : (uint32_t)p.N * (uint32_t)(p.Ho >> 1) * (uint32_t)(p.Wo >> 1) * (uint32_t)p.Cout * 2u;
        const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(addp), (short)0, addp ? (int)addbytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_mask = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.mask), (short)0, p.mask ? (int)ybytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_bnx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.bnx), (short)0, p.bnx ? (int)ybytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_bits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.mask_bits), (short)0, p.mask_bits ? (int)(ybytes >> 4) : 0, 0x00020000);
        const uint32_t gob = (uint32_t)go0 * 2u, gsb = (uint32_t)gstep * 2u;     // byte offsets of the thread's first row chunk / row step
        // (aux 2 on the 16-byte operand loads = non-temporal: each is read once by this launch; A/B -0.05 ms per train step. The same
        // hint on the K loop's A-operand DMA costs +0.7 ms: the N tiles of a row block share those lines through the L2.)
        cv_u32x4 f_add[2][HALF], f_mask[2][HALF], f_bnx[2][HALF];
        uint32_t f_bits[2][HALF], f_ob[2][HALF], f_oa[2][HALF];     // f_ob: byte offset of the row's chunk in y, f_oa: in the addend stream
#define CV_EPI_LOAD(set, hh_)                                                                                   \
        _Pragma("unroll")                                                                                       \
        for (int ii = 0; ii < HALF; ++ii) {                                                                     \
            uint32_t ob = gob + (uint32_t)((hh_) * HALF + ii) * gsb, oa = ob;                                   \
            if (decode) {                                         /* (arithmetic only: no memory instruction inside) */ \
                const int m = m0 + srow + ((hh_) * HALF + ii) * RPI;                                            \
                int q1 = (int)((float)m * p.inv_wo), wo = m - q1 * p.Wo;                                        \
                if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }                    \
                int n = (int)((float)q1 * p.inv_ho), ho = q1 - n * p.Ho;                                        \
                if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }                      \
                const bool valid = full || m < p.M;                                                             \
                if (p.o2) {                                        /* parity class of a stride-2 data gradient: scattered rows */ \
                    ob = valid ? (uint32_t)((((size_t)n * p.OH + 2 * ho + p.o_a) * p.OW + 2 * wo + p.o_b) * p.Cout + n0 + sch * 8) * 2u : (uint32_t)CV_OOB; \
                    oa = ob;                                                                                    \
                }                                                                                               \
                if (p.addend2)                                     /* rows at even (ho, wo) also receive compact[n, ho/2, wo/2, :] */ \
                    oa = (valid && !((ho | wo) & 1)) ? (uint32_t)(((size_t)(n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + n0 + sch * 8) * 2u \
                                                     : (uint32_t)CV_OOB;                                         \
            }                                                                                                   \
            f_ob[set][ii] = ob; f_oa[set][ii] = oa;                                                             \
            f_add[set][ii] = __builtin_amdgcn_raw_buffer_load_b128(r_add, (int)oa, 0, 2);                       \
            f_mask[set][ii] = __builtin_amdgcn_raw_buffer_load_b128(r_mask, (int)ob, 0, 2);                     \
            f_bits[set][ii] = __builtin_amdgcn_raw_buffer_load_b8(r_bits, (int)(ob >> 4), 0, 0);                \
            f_bnx[set][ii] = __builtin_amdgcn_raw_buffer_load_b128(r_bnx, (int)ob, 0, 2);                       \
        }
        CV_EPI_LOAD(0, 0);
        mask_coefficients();                                    // (its loads travel together with the first batch's)
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < NBATCH; ++hh) {
            const int set = hh & 1;
            if (hh + 1 < NBATCH) { CV_EPI_LOAD(set ^ 1, hh + 1); }
#pragma unroll
            for (int ii = 0; ii < HALF; ++ii) {
                const int i = hh * HALF + ii;
                const cv_u32x4 cc = *reinterpret_cast<const cv_u32x4*>(cs + i * RPI * CS_STRIDE);
                uint32_t cw[4] = {cc.x, cc.y, cc.z, cc.w};
                if (!full && !(m0 + srow + i * RPI < p.M)) cw[0] = cw[1] = cw[2] = cw[3] = 0u;   // (not stored; zeros in the sums)
                if (fwd_stats) {
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const float f0 = __uint_as_float(cw[q2] << 16), f1 = __uint_as_float(cw[q2] & 0xffff0000u);
                        ssum[2 * q2] += f0; ssq[2 * q2] += f0 * f0; ssum[2 * q2 + 1] += f1; ssq[2 * q2 + 1] += f1 * f1;
                    }
                }
                if (addp) {                                     // y = bf16(bf16(conv) + addend), like an eager add kernel
                    const uint32_t aw[4] = {f_add[set][ii].x, f_add[set][ii].y, f_add[set][ii].z, f_add[set][ii].w};
                    const bool has = p.addend || f_oa[set][ii] != (uint32_t)CV_OOB;   // (compact stream: even pixels only; the others keep their bits)
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const uint32_t sum = cv_pack_bf16(__uint_as_float(cw[q2] << 16) + __uint_as_float(aw[q2] << 16),
                                                          __uint_as_float(cw[q2] & 0xffff0000u) + __uint_as_float(aw[q2] & 0xffff0000u));
                        cw[q2] = has ? sum : cw[q2];
                    }
                }
                if (p.mask) {                                   // ReLU backward of the tensor this gradient belongs to
                    const uint32_t kw[4] = {f_mask[set][ii].x, f_mask[set][ii].y, f_mask[set][ii].z, f_mask[set][ii].w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        if (!(__uint_as_float(kw[q2] << 16) > 0.0f)) cw[q2] &= 0xffff0000u;
                        if (!(__uint_as_float(kw[q2] & 0xffff0000u) > 0.0f)) cw[q2] &= 0x0000ffffu;
                    }
                }
                if (p.mask_bits) {                              // the same decision, from the forward's bit per element
                    const uint32_t bb = f_bits[set][ii];
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        if (!(bb & (1u << (2 * q2)))) cw[q2] &= 0xffff0000u;
                        if (!(bb & (2u << (2 * q2)))) cw[q2] &= 0x0000ffffu;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(cv_u32x4{cw[0], cw[1], cw[2], cw[3]}, r_y, (int)f_ob[set][ii], 0, 2);   // (aux 2 = non-temporal, as above: -0.14 ms)
                if (p.bnx) {
                    const uint32_t xw[4] = {f_bnx[set][ii].x, f_bnx[set][ii].y, f_bnx[set][ii].z, f_bnx[set][ii].w};
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
#undef CV_EPI_LOAD
        cv_epilogue_stats<BN>(p, ssum, ssq, Ss, t, n0, mt);
        return;
    }

    mask_coefficients();
    // The general loop (only when a dense AND a compact shortcut gradient are given — no layer of ResNet-50 does): the thread's rows in
    // batches, (operand loads of a batch, all in flight together) -> (its arithmetic and stores). The first batch's loads are issued
    // before the barrier that publishes the staging tile. (All rows at once would need 100+ registers.)
#pragma unroll
    for (int hh = 0; hh < NBATCH; ++hh) {
        uint32_t orow[HALF];                                    // element offset of the row's chunk in y (and in bnx / addend / mask)
        uint32_t o2row[HALF];                                   // ... of its compact stride-2 addend, or ~0u
        uint4 v_add[HALF], v_mask[HALF], v_bnx[HALF];
        uint32_t v_bits[HALF];
#pragma unroll
        for (int ii = 0; ii < HALF; ++ii) {
            const int i = hh * HALF + ii;
            orow[ii] = (uint32_t)(go0 + (size_t)i * gstep);
            o2row[ii] = ~0u;
            v_add[ii] = v_mask[ii] = v_bnx[ii] = make_uint4(0u, 0u, 0u, 0u);
            v_bits[ii] = 0xffu;
            if (full || m0 + srow + i * RPI < p.M) {
                if (decode) {
                    const int m = m0 + srow + i * RPI;
                    int q1 = (int)((float)m * p.inv_wo), wo = m - q1 * p.Wo;
                    if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
                    int n = (int)((float)q1 * p.inv_ho), ho = q1 - n * p.Ho;
                    if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }
                    if (p.o2)                                    // parity class of a stride-2 data gradient: scattered rows
                        orow[ii] = (uint32_t)((((size_t)n * p.OH + 2 * ho + p.o_a) * p.OW + 2 * wo + p.o_b) * p.Cout + n0 + sch * 8);
                    if (p.addend2 && !((ho | wo) & 1))           // rows at even (ho, wo) also receive compact[n, ho/2, wo/2, :]
                        o2row[ii] = (uint32_t)(((size_t)(n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + n0 + sch * 8);
                }
                if (p.addend) v_add[ii] = *reinterpret_cast<const uint4*>(p.addend + orow[ii]);
                if (p.mask) v_mask[ii] = *reinterpret_cast<const uint4*>(p.mask + orow[ii]);
                if (p.mask_bits) v_bits[ii] = p.mask_bits[orow[ii] >> 3];
                if (p.bnx) v_bnx[ii] = *reinterpret_cast<const uint4*>(p.bnx + orow[ii]);
            }
        }
        if (hh == 0) __syncthreads();
#pragma unroll
        for (int ii = 0; ii < HALF; ++ii) {
            const int i = hh * HALF + ii;
            if (full || m0 + srow + i * RPI < p.M) {
                uint4 c = *reinterpret_cast<const uint4*>(cs + i * RPI * CS_STRIDE);
                if (fwd_stats) {
                    const uint32_t sw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const float f0 = __uint_as_float(sw[q2] << 16), f1 = __uint_as_float(sw[q2] & 0xffff0000u);
                        ssum[2 * q2] += f0; ssq[2 * q2] += f0 * f0; ssum[2 * q2 + 1] += f1; ssq[2 * q2 + 1] += f1 * f1;
                    }
                }
                if (p.addend) {                                 // y = bf16(bf16(conv) + addend), like an eager add kernel
                    uint32_t cw[4] = {c.x, c.y, c.z, c.w};
                    const uint32_t aw[4] = {v_add[ii].x, v_add[ii].y, v_add[ii].z, v_add[ii].w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2)
                        cw[q2] = cv_pack_bf16(__uint_as_float(cw[q2] << 16) + __uint_as_float(aw[q2] << 16),
                                              __uint_as_float(cw[q2] & 0xffff0000u) + __uint_as_float(aw[q2] & 0xffff0000u));
                    c = make_uint4(cw[0], cw[1], cw[2], cw[3]);
                }
                if (p.addend2 && o2row[ii] != ~0u) {            // (three launches per step: loaded here, not ahead)
                    const uint4 a = *reinterpret_cast<const uint4*>(p.addend2 + o2row[ii]);
                    uint32_t cw[4] = {c.x, c.y, c.z, c.w};
                    const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2)
                        cw[q2] = cv_pack_bf16(__uint_as_float(cw[q2] << 16) + __uint_as_float(aw[q2] << 16),
                                              __uint_as_float(cw[q2] & 0xffff0000u) + __uint_as_float(aw[q2] & 0xffff0000u));
                    c = make_uint4(cw[0], cw[1], cw[2], cw[3]);
                }
                if (p.mask) {                                   // ReLU backward of the tensor this gradient belongs to
                    const uint32_t kw[4] = {v_mask[ii].x, v_mask[ii].y, v_mask[ii].z, v_mask[ii].w};
                    uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        if (!(__uint_as_float(kw[q2] << 16) > 0.0f)) cw[q2] &= 0xffff0000u;
                        if (!(__uint_as_float(kw[q2] & 0xffff0000u) > 0.0f)) cw[q2] &= 0x0000ffffu;
                    }
                    c = make_uint4(cw[0], cw[1], cw[2], cw[3]);
                }
                if (p.mask_bits) {                              // the same decision, from the forward's bit per element
                    uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        if (!(v_bits[ii] & (1u << (2 * q2)))) cw[q2] &= 0xffff0000u;
                        if (!(v_bits[ii] & (2u << (2 * q2)))) cw[q2] &= 0x0000ffffu;
                    }
                    c = make_uint4(cw[0], cw[1], cw[2], cw[3]);
                }
                __builtin_nontemporal_store(cv_u32x4{c.x, c.y, c.z, c.w}, reinterpret_cast<cv_u32x4*>(p.y + orow[ii]));
                if (p.bnx) {
                    const uint32_t gw[4] = {c.x, c.y, c.z, c.w};
                    const uint32_t xw[4] = {v_bnx[ii].x, v_bnx[ii].y, v_bnx[ii].z, v_bnx[ii].w};
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        float g0 = __uint_as_float(gw[q2] << 16), g1 = __uint_as_float(gw[q2] & 0xffff0000u);
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
    }
    cv_epilogue_stats<BN>(p, ssum, ssq, Ss, t, n0, mt);
}

template <int BN, bool LEAN = false, int NBATCH = 2>
__device__ __forceinline__ void cv_epilogue(const ConvP& p, const f32x16 (&acc)[(BN == 128) ? 2 : 1][2], unsigned char* smem, int t, int m0,
                                            int n0, int mt) {
    constexpr int MI = (BN == 128) ? 2 : 1;
    constexpr int NI = 2;
    constexpr int WM = MI * 32;
    const int lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave;
    const int wn = (BN == 128) ? (wave & 1) : 0;
    const int frow = lane & 31, fhalf = lane >> 5;
    constexpr int CS_STRIDE = BN * 2 + 16;
    cv_stage_acc<MI, NI, CS_STRIDE>(acc, smem + (wm * WM + frow) * CS_STRIDE + (wn * 64 + 4 * fhalf) * 2);
    cv_epilogue_staged<BN, LEAN, NBATCH>(p, smem, t, m0, n0, mt);
}
