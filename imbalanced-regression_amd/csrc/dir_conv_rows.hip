// Row-resident 1x1 convolution for the short-K expanding layers (imdb-wiki-dir/resnet.py:44-51: conv3 of a Bottleneck, the 1x1 shortcut,
// and — with the rotated weights — the data gradient of conv1): Cin = 64 / 128 / 256, Cout a multiple of 128, M = N*Ho*Wo rows.
//
// The 128 x 128 tile kernels (dir_conv.hip) stage BOTH operands of every tile through L2 -> LDS: for 256 -> 1024 at 14^2 that is 401 MB
// for 129 MB of tensors (every one of the 392 row tiles re-stages each 64 KB weight slice, every one of the 8 column tiles re-stages each
// A tile), and a workgroup's load, MFMA and store phases add up instead of overlapping: its wavefronts wait for the K-step's DMA
// (s_waitcnt vmcnt), and on gfx950 that counter is in issue order over loads AND stores, so a wavefront that has stored a tile cannot wait for
// a later load without waiting ~3 us for the stores' acknowledgement as well (profiles/r02_conv_phase_ablation.txt).
//
// Here the A operand never goes through LDS and is read ONCE: a workgroup owns 128 rows (pixels) and walks over its column tiles; each of its
// four compute wavefronts keeps the MFMA fragments of its 32 rows x K in registers (16 B per lane and 16-wide k-step: K = 256 -> 64 VGPRs),
// loaded straight from global memory in fragment shape. Only the weights stream through LDS, as a continuous ring of 16 KB K-step slices
// (128 channels x 64 k, the B-tile image of conv_igemm_dma_kernel) issued by TWO more wavefronts that do nothing else: they own the
// LDS-DMA queue (their vmcnt counts only DMA pieces, so their waits are exact), the compute wavefronts never wait on the memory counter at
// all — their stores drain while they multiply the next tiles — and the two meet at one s_barrier per K-step.
// LDS: two persistent ring slots (next tile's K-steps 0 and 1, prefetched across the epilogue) + the 128 x 128 staging tile of the shared
// epilogue, which doubles as slots 2 and 3 while a K = 256 tile is multiplied = 70 KB: two workgroups per CU (<= 168 registers, 3 waves per SIMD).
// Staged bytes for 256 -> 1024 at 14^2: 205 MB of weights (L2 hits) + 26 MB of A, instead of 401 MB.
// Same MFMA, same operand order, same k order per output element and the shared epilogue (dir_conv_epilogue.h): outputs and statistics rows are
// BIT-IDENTICAL to the tile kernels (tools/check_conv_variants.py rows).
#include "dir_common.h"
#include "dir_conv_shared.h"
#include "dir_conv_epilogue.h"

namespace {

constexpr int WR_NC = 8, WR_NL = 2;                                   // compute wavefronts (two 128-row tiles x 4), loader wavefronts
constexpr int WR_WAVES = WR_NC + WR_NL, WR_THREADS = 64 * WR_WAVES;  // 10 wavefronts, ONE workgroup per CU (see the residency note above)
constexpr int WR_BMW = 256;                                          // rows per workgroup
constexpr int WR_SLOT = 128 * CV_ROWB;                               // 16 KB: 128 output channels x 64 k (128-byte rows, XOR-swizzled chunks)
constexpr int WR_NS = 4;                                             // ring slots
constexpr int WR_EPI = CV_BM * (128 * 2 + 16) + 4 * 2 * 128 * 4;      // 38 912 B: staging tile + column partials of one 128 x 128 tile
constexpr int WR_LDS = WR_NS * WR_SLOT + 2 * WR_EPI;                 // 65 536 + 77 824 = 143 360
constexpr int WR_PPL = 16 / WR_NL;                                   // 1 KB pieces of a slice per loader

#define WR_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
#define WR_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// One loader's share of a slice: WR_PPL consecutive pieces, LDS destination m0 = lds .. + 1 KB per piece, source = descriptor + voffset (even / odd
// piece: the swizzle phase) + soffset .. + `step` bytes per piece. ONE statement: M0 is written and read inside it (the compiler does not preserve
// M0 around inline assembly and the loader branch has no compiler-generated user of it), two scalar adds per piece instead of the
// save / set / restore sequence of cp_dma16.
__device__ __forceinline__ void wr_issue_pieces(cp_u32x4 rs, uint32_t lds, int v_even, int v_odd, int soff, int step) {
    static_assert(WR_PPL == 8, "the statement below issues eight pieces");
    asm volatile(
        "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %2, %4, %0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %0, %0, %5\n\t"
        "buffer_load_dwordx4 %3, %4, %0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %0, %0, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %0, %0, %5\n\t"
        "buffer_load_dwordx4 %3, %4, %0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %0, %0, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %0, %0, %5\n\t"
        "buffer_load_dwordx4 %3, %4, %0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %0, %0, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %0, %0, %5\n\t"
        "buffer_load_dwordx4 %3, %4, %0 offen lds"
        : "+s"(soff) : "s"(lds), "v"(v_even), "v"(v_odd), "s"(rs), "s"(step) : "memory", "scc");
}
// wait until at most min(n, rem) slices issued by this loader are still in flight (n = 3, 2: the steady state; rem: what the stream has left)
__device__ __forceinline__ void wr_wait_slices(int n, int rem) {
    const int k = rem < n ? rem : n;
    if (k >= 3) WR_WAIT(3 * WR_PPL); else if (k == 2) WR_WAIT(2 * WR_PPL); else if (k == 1) WR_WAIT(WR_PPL); else WR_WAIT(0);
}

template <int KT, bool LEAN>
__global__ void __launch_bounds__(WR_THREADS) __attribute__((amdgpu_waves_per_eu(3)))
conv1x1_rows_kernel(ConvP p, int ct, int ncg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K = KT * CV_BK;
    int lin;
    {
        const int b = blockIdx.x, q = p.nblocks / 8, r = p.nblocks % 8, xcd = b % 8, i = b / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int rb = lin / ncg, cg = lin - rb * ncg;
    const int m0 = rb * WR_BMW, nt0 = cg * ct;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int total = ct * KT;                                       // slices of this workgroup's weight stream; slice s = (tile s / KT, K-step s % KT) lives in slot s % 4

    if (wave >= WR_NC) {
        // ---- loaders: the weight stream of this workgroup's column tiles, one 16 KB slice = 16 pieces of 1 KB (8 rows x 128 B) per K-step, loader
        // l issues pieces 8 l .. 8 l + 7. Piece i = rows 8 i + (lane >> 3) of the slice; the lane fills physical chunk (lane & 7) with LOGICAL chunk
        // (lane & 7) ^ ((row >> 1) & 7), (row >> 1) & 7 = (lane >> 4) | ((i & 1) << 2) — the B-tile image of conv_igemm_dma_kernel.
        typedef __attribute__((address_space(3))) unsigned char* wr_lds_t;
        const int l = wave - WR_NC;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(wr_lds_t)smem + (uint32_t)(l * WR_PPL * 1024);
        const cp_u32x4 rs_w = cp_rsrc(p.w, (uint32_t)p.Cout * (uint32_t)K * 2u);
        const int lr = lane >> 3, lc = lane & 7;
        const int wv0 = (lr * K + ((lc ^ (lr >> 1)) * 8)) * 2, wv1 = (lr * K + ((lc ^ ((lr >> 1) | 4)) * 8)) * 2;
        const int soff0 = (nt0 * 128 + l * WR_PPL * 8) * K * 2;
        // slice s: column tile s / KT, K-step s % KT -> byte offset (tile * 128 * K + kstep * 64) * 2 into the [Cout][K] weights
#define WR_ISSUE(s_) { const int s__ = (s_); if (s__ < total) { const int c__ = s__ / KT, k__ = s__ - c__ * KT;                                    \
                       wr_issue_pieces(rs_w, lds0 + (uint32_t)((s__ & (WR_NS - 1)) * WR_SLOT), wv0, wv1, soff0 + (c__ * 128 * K + k__ * CV_BK) * 2, 8 * K * 2); } }
        // Barriers per tile (the compute wavefronts execute the same sequence): "K-step ks has landed" for ks = 0 .. KT - 1 (which also says K-step
        // ks - 1 is consumed), "last K-step consumed", then the epilogue's own. Slice s + 4 is issued as soon as slice s is known to be consumed, so
        // before the barrier of slice s (K-step ks) the slices up to s + 3 (ks = 0) / s + 2 (ks > 0) are in flight: the vmcnt of the wait.
        WR_ISSUE(0); WR_ISSUE(1); WR_ISSUE(2); WR_ISSUE(3);
        int s = 0;
        for (int c = 0; c < ct; ++c) {
#pragma unroll
            for (int ks = 0; ks < KT; ++ks, ++s) {
                wr_wait_slices(ks == 0 ? 3 : 2, total - 1 - s);
                WR_BAR();                                           // slice s has landed; slice s - 1 (same tile) is consumed
                if (ks > 0) WR_ISSUE(s + 3);
            }
            WR_BAR();                                               // the tile's last slice is consumed
            WR_ISSUE(s + 3);
            WR_BAR();                                               // the epilogue's barriers: staging tiles published ...
            if (p.stats) WR_BAR();                                  // ... and the column partials of their statistics (cv_epilogue_stats)
        }
#undef WR_ISSUE
        return;
    }

    // ---- compute wavefronts: group g = wave >> 2 owns the 128-row tile m0 + 128 g, wavefront wl = wave & 3 its rows 32 wl .. + 31; all K of those rows in
    // registers as MFMA fragments (lane: row lane & 31, k = 16 s + 8 (lane >> 5) .. + 7), read ONCE, straight from global memory
    const int g = wave >> 2, wl = wave & 3;
    const int frow = lane & 31, fhalf = lane >> 5;
    bf16x8 a[KT * 4];
    {
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), (short)0,
                                                                               (int)((unsigned)(p.N * p.H * p.W) * (unsigned)p.Cin * 2u), 0x00020000);
        const int m = m0 + wave * 32 + frow;
        int aoff = CV_OOB;
        if (m < p.M) {
            if (p.simple) aoff = m * K * 2;
            else {                                                   // 1x1 / pad 0 / stride s: row (n, ho, wo) reads pixel (ho s, wo s)
                int q1 = (int)((float)m * p.inv_wo), wo = m - q1 * p.Wo;
                if (wo < 0) { --q1; wo += p.Wo; } else if (wo >= p.Wo) { ++q1; wo -= p.Wo; }
                int n = (int)((float)q1 * p.inv_ho), ho = q1 - n * p.Ho;
                if (ho < 0) { --n; ho += p.Ho; } else if (ho >= p.Ho) { ++n; ho -= p.Ho; }
                aoff = ((n * p.H + ho * p.stride) * p.W + wo * p.stride) * K * 2;
            }
            aoff += fhalf * 16;
        }
#pragma unroll
        for (int s = 0; s < KT * 4; ++s) {
            const cv_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, aoff, s * 32, 0);
            a[s] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // B fragments: channel row ni * 32 + frow of the slice (ni * 4096 bytes further: the swizzle phase (row >> 1) & 7 does not depend on ni)
    uint32_t bfa[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        bfa[kk] = frow * CV_ROWB + (((kk * 2 + fhalf) ^ ((frow >> 1) & 7)) << 4);
        asm volatile("" : "+v"(bfa[kk]));
    }
    constexpr int CS_STRIDE = 128 * 2 + 16;
    unsigned char* const Cs = smem + WR_NS * WR_SLOT + g * WR_EPI;
    const int tg = t & 255;

    for (int c = 0; c < ct; ++c) {
        f32x16 acc[1][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][ni][e] = 0.0f;
        const int s0 = (c * KT) & (WR_NS - 1);                      // slot of the tile's first slice (KT = 4: always 0)
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            WR_BAR();                                               // this K-step's slice has landed
            const unsigned char* slot = smem + ((s0 + ks) & (WR_NS - 1)) * WR_SLOT;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 b[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(slot + ni * 4096 + bfa[kk]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[ks * 4 + kk], acc[0][ni], 0, 0, 0);   // D'[channel][pixel]
            }
        }
        WR_BAR();                                                   // every wavefront is past its last fragment read of the tile
        cv_stage_acc<1, 4, CS_STRIDE>(acc, Cs + (wl * 32 + frow) * CS_STRIDE + (4 * fhalf) * 2);
        cv_epilogue_staged<128, LEAN, LEAN ? 2 : 8, true>(p, Cs, tg, m0 + g * CV_BM, (nt0 + c) * 128, rb * 2 + g);
    }
}

}  // namespace

// Geometry the kernel takes: 1x1, pad 0, stride 1 or 2, Cin in {64, 128, 256}, Cout % 128 == 0 with at least two column tiles, whole 256-row blocks.
bool conv_rows_geometry(long long M, int Cin, int Cout, int R, int S, int stride, int pad) {
    return R == 1 && S == 1 && pad == 0 && (stride == 1 || stride == 2) && (Cin == 64 || Cin == 128 || Cin == 256) && Cout % 128 == 0 && Cout >= 256 &&
           M % WR_BMW == 0;
}

// p: as conv_launch_ex fills it (M, KT, simple, stats, fused operands ...). Sets the grid fields itself.
int conv_rows_launch(ConvP p, bool lean, hipStream_t s) {
    const int mtiles = p.M / WR_BMW, ntn = p.Cout / 128;
    // column groups: one workgroup walks over ALL column tiles of its rows (A is read once) unless that leaves the chip short of workgroups
    int ncg = 1;
    while (mtiles * ncg < 192 && ntn % (ncg * 2) == 0 && ntn / (ncg * 2) >= 2) ncg *= 2;
    const int ct = ntn / ncg;
    p.ntn = ntn;
    p.nblocks = mtiles * ncg;
#define WR_ATTR(KT_, L_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_rows_kernel<KT_, L_>), hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS)
    DIR_ONCE_PER_DEVICE(WR_ATTR(1, true); WR_ATTR(1, false); WR_ATTR(2, true); WR_ATTR(2, false); WR_ATTR(4, true); WR_ATTR(4, false));
#undef WR_ATTR
#define WR_LAUNCH(KT_, L_) hipLaunchKernelGGL((conv1x1_rows_kernel<KT_, L_>), dim3(p.nblocks), dim3(WR_THREADS), WR_LDS, s, p, ct, ncg)
    if (p.KT == 4) { if (lean) WR_LAUNCH(4, true); else WR_LAUNCH(4, false); }
    else if (p.KT == 2) { if (lean) WR_LAUNCH(2, true); else WR_LAUNCH(2, false); }
    else if (p.KT == 1) { if (lean) WR_LAUNCH(1, true); else WR_LAUNCH(1, false); }
    else return DIR_EUNSUPPORTED;
#undef WR_LAUNCH
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
