// FDS kernels for MI355X / gfx950 (CDNA4).  Hand-written HIP; compiled with -ffp-contract=off.
// Reference behaviour replaced: imdb-wiki-dir/fds.py (all) + utils.py:97-107; see include/dir_hip.h.
//
// Every kernel here is HBM/launch bound (no dense contraction), so the design rules are the
// streaming ones: 16-byte-per-lane coalesced rows, row-uniform table indices in SGPRs,
// float64 register accumulators, wavefront-shuffle / ballot based grouping, no atomics on the
// data path (bit-reproducible results), grids sized >> 256 CUs.
#include "dir_common.h"

// =============================================================================================
// K1: labels -> bins
// =============================================================================================
__global__ void __launch_bounds__(DIR_TPB)
fds_label_flags_kernel(const float* __restrict__ labels, int n, float lo, float hi, uint32_t* flags) {
    uint32_t f = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        f |= dir_label_flag_bits(labels[i], lo, hi);
    f = dir_wave_or(f);
    if ((threadIdx.x & (DIR_WAVE - 1)) == 0 && f) atomicOr(flags, f);
}

__global__ void __launch_bounds__(DIR_TPB)
fds_assign_bins_kernel(const float* __restrict__ labels, int n, float lo, float hi,
                       const uint32_t* __restrict__ flags, int32_t* __restrict__ bins) {
    const uint32_t f = *flags;
    const bool has_lo = f & DIR_FLAG_HAS_LO, has_hi = f & DIR_FLAG_HAS_HI;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        bins[i] = dir_bin_of(labels[i], lo, hi, has_lo, has_hi);
}

static int check_buckets(int bucket_start, int bucket_num) {
    return (bucket_num - bucket_start) >= 1 ? DIR_OK : DIR_EINVAL;
}

extern "C" int dir_fds_label_flags(const float* labels, int n, int bucket_start, int bucket_num,
                                   uint32_t* flags, dir_stream_t stream) {
    DIR_RETURN_IF(!labels || !flags || n < 0, DIR_EINVAL);
    DIR_RETURN_IF(check_buckets(bucket_start, bucket_num), DIR_EINVAL);
    if (n == 0) return DIR_OK;
    int grid = dir_cdiv(n, DIR_TPB); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(fds_label_flags_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream),
                       labels, n, (float)bucket_start, (float)(bucket_num - 1), flags);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_fds_assign_bins(const float* labels, int n, int bucket_start, int bucket_num,
                                   const uint32_t* flags, int32_t* bins, dir_stream_t stream) {
    DIR_RETURN_IF(!labels || !flags || !bins || n < 0, DIR_EINVAL);
    DIR_RETURN_IF(check_buckets(bucket_start, bucket_num), DIR_EINVAL);
    if (n == 0) return DIR_OK;
    int grid = dir_cdiv(n, DIR_TPB); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(fds_assign_bins_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream),
                       labels, n, (float)bucket_start, (float)(bucket_num - 1), flags, bins);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_fds_bin_index(const float* labels, int n, int bucket_start, int bucket_num,
                                 int32_t* bins, uint32_t* flags, dir_stream_t stream) {
    DIR_RETURN_IF(!flags, DIR_EINVAL);
    hipError_t e = hipMemsetAsync(flags, 0, sizeof(uint32_t), dir_s(stream));
    if (e != hipSuccess) return (int)e;
    int rc = dir_fds_label_flags(labels, n, bucket_start, bucket_num, flags, stream);
    if (rc) return rc;
    return dir_fds_assign_bins(labels, n, bucket_start, bucket_num, flags, bins, stream);
}

// =============================================================================================
// K2: per-bin (count, mean, M2) in one streaming pass
// =============================================================================================
// Stage G1-G3: stable counting sort of the row indices by bin (rows with bin < 0 dropped).
//   tile  = GROUP_TILE consecutive rows handled by ONE wavefront (so ranks inside a tile come from
//           ballots in program order: stable and deterministic without atomics on the data path);
// Stage P : every "piece" (<= PIECE_ROWS sorted rows of one bin) x (column tile) is reduced in
//           float64 registers around the shift K = first row of the bin;
// Stage C : pieces of a bin are combined in index order -> (count, mean, M2).
#define GROUP_TILE 256        // rows per grouping wavefront: N / 256 wavefronts keep all 256 CUs busy at N = 191 509
#define PIECE_ROWS 256        // rows per piece: 2 x fewer float64 partials to write and re-read than 128 (52 -> 26 MB at N = 191 509)
#define PIECE_UNROLL 8

struct ScatterWs {            // carved from the caller's workspace (all 256-B aligned)
    int32_t* tile_hist;       // [ntiles][nb]   counts, then exclusive prefix over tiles
    int32_t* totals;          // [nb]           rows per bin
    int32_t* offsets;         // [nb+1]         start of each bin in perm
    int32_t* bin_piece0;      // [nb+1]         first piece of each bin
    int32_t* npieces;         // [1]
    int32_t* perm;            // [n]            row indices grouped by bin
    int32_t* piece_bin;       // [maxpieces]
    int32_t* piece_p0;        // [maxpieces]
    int32_t* piece_p1;        // [maxpieces]
    double*  partials;        // [maxpieces][2][C]
    int ntiles, maxpieces;
    size_t bytes;
};

static ScatterWs carve_ws(void* base, int n, int C, int nb) {
    ScatterWs w;
    w.ntiles = dir_cdiv(n > 0 ? n : 1, GROUP_TILE);
    w.maxpieces = dir_cdiv(n > 0 ? n : 1, PIECE_ROWS) + nb;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = dir_align_up(off + bytes, 256); return o; };
    char* b = static_cast<char*>(base);
    size_t o_hist = take(sizeof(int32_t) * (size_t)w.ntiles * nb);
    size_t o_tot = take(sizeof(int32_t) * nb);
    size_t o_off = take(sizeof(int32_t) * (nb + 1));
    size_t o_bp = take(sizeof(int32_t) * (nb + 1));
    size_t o_np = take(sizeof(int32_t));
    size_t o_perm = take(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    size_t o_pb = take(sizeof(int32_t) * (size_t)w.maxpieces);
    size_t o_p0 = take(sizeof(int32_t) * (size_t)w.maxpieces);
    size_t o_p1 = take(sizeof(int32_t) * (size_t)w.maxpieces);
    size_t o_part = take(sizeof(double) * (size_t)w.maxpieces * 2 * C);
    w.tile_hist = (int32_t*)(b + o_hist); w.totals = (int32_t*)(b + o_tot); w.offsets = (int32_t*)(b + o_off);
    w.bin_piece0 = (int32_t*)(b + o_bp); w.npieces = (int32_t*)(b + o_np);
    w.perm = (int32_t*)(b + o_perm); w.piece_bin = (int32_t*)(b + o_pb);
    w.piece_p0 = (int32_t*)(b + o_p0); w.piece_p1 = (int32_t*)(b + o_p1);
    w.partials = (double*)(b + o_part);
    w.bytes = off;
    return w;
}

extern "C" size_t dir_fds_scatter_stats_workspace(int n, int C, int nb) {
    if (n < 0 || C <= 0 || nb <= 0) return 0;
    return carve_ws(nullptr, n, C, nb).bytes;
}

// G1: one wavefront per tile; LDS histogram.
__global__ void __launch_bounds__(DIR_WAVE)
fds_group_hist_kernel(const int32_t* __restrict__ bins, int n, int nb, int32_t* __restrict__ tile_hist) {
    extern __shared__ __attribute__((aligned(16))) int32_t hist[];
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int b = lane; b < nb; b += DIR_WAVE) hist[b] = 0;
    __syncthreads();
    const int r0 = tile * GROUP_TILE;
    for (int r = r0 + lane; r < min(n, r0 + GROUP_TILE); r += DIR_WAVE) {
        const int b = bins[r];
        if (b >= 0 && b < nb) atomicAdd(&hist[b], 1);        // LDS integer atomic: order-free
    }
    __syncthreads();
    for (int b = lane; b < nb; b += DIR_WAVE) tile_hist[(size_t)tile * nb + b] = hist[b];
}

// Exclusive prefix of one value per thread over the 256 threads of a workgroup (wavefront shuffles + one LDS hop), plus the workgroup total.
__device__ __forceinline__ int fds_block_excl_scan(int v, int* total) {
    __shared__ int32_t wsum[DIR_TPB / DIR_WAVE];
    const int t = threadIdx.x, lane = t & (DIR_WAVE - 1), wid = t / DIR_WAVE;
    int incl = v;
#pragma unroll
    for (int o = 1; o < DIR_WAVE; o <<= 1) { const int u = __shfl_up(incl, o, DIR_WAVE); if (lane >= o) incl += u; }
    __syncthreads();                                       // (wsum of a previous call is no longer being read)
    if (lane == DIR_WAVE - 1) wsum[wid] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < DIR_TPB / DIR_WAVE; ++k) { const int w = wsum[k]; if (k < wid) woff += w; tot += w; }
    *total = tot;
    return woff + incl - v;
}

// G2a: one workgroup per bin: exclusive prefix of that bin's counts over the tiles, total per bin. Round 6: a thread owns GS_PER CONSECUTIVE
// tiles — all its loads are independent (one L2 round trip instead of a dependent load -> scan -> store chain per 256 tiles: 10.9 -> ~3 us at
// 2 166 tiles), prefix inside the thread, ONE workgroup scan over the thread totals.
#define GS_PER 16
__global__ void __launch_bounds__(DIR_TPB)
fds_group_scan_tiles_kernel(int32_t* __restrict__ tile_hist, int ntiles, int nb, int32_t* __restrict__ totals) {
    const int b = blockIdx.x, t = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < ntiles; base += DIR_TPB * GS_PER) {
        int v[GS_PER];
        const int t0 = base + t * GS_PER;
#pragma unroll
        for (int u = 0; u < GS_PER; ++u) v[u] = (t0 + u < ntiles) ? tile_hist[(size_t)(t0 + u) * nb + b] : 0;
        int sum = 0;
#pragma unroll
        for (int u = 0; u < GS_PER; ++u) { const int c = v[u]; v[u] = sum; sum += c; }
        int tot;
        const int off = carry + fds_block_excl_scan(sum, &tot);
#pragma unroll
        for (int u = 0; u < GS_PER; ++u) if (t0 + u < ntiles) tile_hist[(size_t)(t0 + u) * nb + b] = off + v[u];
        carry += tot;
    }
    if (t == 0) totals[b] = carry;
}

// G2b: single workgroup: offsets over bins (workgroup scan over the bins, 256 at a time), piece table.
__global__ void __launch_bounds__(DIR_TPB)
fds_group_scan_kernel(const int32_t* __restrict__ totals, int nb, int maxpieces,
                      int32_t* __restrict__ offsets, int32_t* __restrict__ bin_piece0,
                      int32_t* __restrict__ npieces, int32_t* __restrict__ piece_bin,
                      int32_t* __restrict__ piece_p0, int32_t* __restrict__ piece_p1) {
    const int t = threadIdx.x;
    int co = 0, cp = 0;                                    // running totals of rows / pieces over the bins before this chunk
    for (int base = 0; base < nb; base += DIR_TPB) {
        const int b = base + t;
        const int c = b < nb ? totals[b] : 0;
        const int np = (c + PIECE_ROWS - 1) / PIECE_ROWS;
        int to, tp;
        const int o0 = co + fds_block_excl_scan(c, &to);
        int k = cp + fds_block_excl_scan(np, &tp);
        if (b < nb) {
            offsets[b] = o0; bin_piece0[b] = k;
            const int o1 = o0 + c;
            for (int p0 = o0; p0 < o1; p0 += PIECE_ROWS, ++k) {
                if (k < maxpieces) { piece_bin[k] = b; piece_p0[k] = p0; piece_p1[k] = min(p0 + PIECE_ROWS, o1); }
            }
        }
        co += to; cp += tp;
    }
    if (t == 0) { offsets[nb] = co; bin_piece0[nb] = cp; *npieces = cp; }
}

// G3: one wavefront per tile; stable placement. For each chunk of 64 rows the lanes that share a bin are found with one
// ballot per BIT of the bin index ("match-any": 7 ballots for 100 bins, no loop over the distinct bins of the chunk, no
// barrier), the lowest such lane bumps the bin's cursor once (LDS atomic of ONE lane per bin and chunk, chunks in program
// order: the result is the stable order, independent of timing) and every lane stores its row at base + (number of lower
// lanes of its bin). Round 1 looped over the distinct bins of a chunk with two barriers each: 22-27 us, latency bound.
__global__ void __launch_bounds__(DIR_WAVE)
fds_group_place_kernel(const int32_t* __restrict__ bins, int n, int nb, int nbits,
                       const int32_t* __restrict__ tile_prefix, const int32_t* __restrict__ offsets,
                       int32_t* __restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) int32_t cursor[];
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int b = lane; b < nb; b += DIR_WAVE) cursor[b] = offsets[b] + tile_prefix[(size_t)tile * nb + b];
    __syncthreads();
    const int r0 = tile * GROUP_TILE;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int c = 0; c < GROUP_TILE; c += DIR_WAVE) {
        const int r = r0 + c + lane;
        int b = (r < n) ? bins[r] : -1;
        if (b >= nb) b = -1;
        const bool valid = b >= 0;
        unsigned long long mask = __ballot(valid);
        for (int k = 0; k < nbits; ++k) {
            const bool bit = valid && ((b >> k) & 1);
            const unsigned long long m = __ballot(bit);
            mask &= bit ? m : ~m;
        }
        if (valid) {                                            // mask = the valid lanes of this chunk with my bin
            const int leader = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&cursor[b], __popcll(mask));
            base = __shfl(base, leader, DIR_WAVE);
            perm[base + __popcll(mask & lt)] = r;
        }
        if (r0 + c + DIR_WAVE >= n) break;
    }
}

// P: piece x column-tile partial sums in float64 registers. VEC = 4 (16 B per lane) or 1.
// non-temporal 16-byte load / store: feature rows are read once per pass and are far larger than the L2
typedef __attribute__((ext_vector_type(4))) float fds_f32x4;
__device__ __forceinline__ float4 fds_ldnt(const float* p) { const fds_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const fds_f32x4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float fds_ldnt1(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void fds_stnt(float* p, const float4& v) { __builtin_nontemporal_store(fds_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<fds_f32x4*>(p)); }
template <int VEC> struct FVec;
template <> struct FVec<4> { using T = float4; static __device__ __forceinline__ float4 ldnt(const float* p) { return fds_ldnt(p); } };
template <> struct FVec<1> { using T = float; static __device__ __forceinline__ float ldnt(const float* p) { return fds_ldnt1(p); } };
template <int VEC> __device__ __forceinline__ void fvec_get(const typename FVec<VEC>::T& v, float (&o)[VEC]);
template <> __device__ __forceinline__ void fvec_get<4>(const float4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void fvec_get<1>(const float& v, float (&o)[1]) { o[0] = v; }

template <int VEC>
__global__ void __launch_bounds__(DIR_TPB)
fds_piece_sums_kernel(const float* __restrict__ feats, int C,
                      const int32_t* __restrict__ perm, const int32_t* __restrict__ offsets,
                      const int32_t* __restrict__ npieces, const int32_t* __restrict__ piece_bin,
                      const int32_t* __restrict__ piece_p0, const int32_t* __restrict__ piece_p1,
                      double* __restrict__ partials) {
    using V = typename FVec<VEC>::T;
    const int k = blockIdx.x;
    if (k >= *npieces) return;
    const int col = (blockIdx.y * DIR_TPB + threadIdx.x) * VEC;
    if (col >= C) return;
    const int b = piece_bin[k], p0 = piece_p0[k], p1 = piece_p1[k];      // block-uniform (SGPRs)
    float kf[VEC];
    fvec_get<VEC>(*reinterpret_cast<const V*>(feats + (size_t)perm[offsets[b]] * C + col), kf);
    double kd[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { kd[j] = (double)kf[j]; s1[j] = 0.0; s2[j] = 0.0; }
    int p = p0;
    for (; p + PIECE_UNROLL <= p1; p += PIECE_UNROLL) {
        V v[PIECE_UNROLL];
#pragma unroll
        for (int u = 0; u < PIECE_UNROLL; ++u)              // 8 independent 16-B loads in flight per lane
            v[u] = FVec<VEC>::ldnt(feats + (size_t)perm[p + u] * C + col);
#pragma unroll
        for (int u = 0; u < PIECE_UNROLL; ++u) {
            float x[VEC]; fvec_get<VEC>(v[u], x);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const double d = (double)x[j] - kd[j]; s1[j] += d; s2[j] += d * d; }
        }
    }
    for (; p < p1; ++p) {
        float x[VEC]; fvec_get<VEC>(FVec<VEC>::ldnt(feats + (size_t)perm[p] * C + col), x);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const double d = (double)x[j] - kd[j]; s1[j] += d; s2[j] += d * d; }
    }
    double* o1 = partials + ((size_t)k * 2 + 0) * C + col;
    double* o2 = partials + ((size_t)k * 2 + 1) * C + col;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { o1[j] = s1[j]; o2[j] = s2[j]; }
}


// Narrow feature rows (C <= 512, e.g. the dense NYUD2 variant's C = 128): one row occupies only C/4 lanes, so a
// workgroup walks RL = 256 / (C/4) rows of the piece at a time (row lanes) and the lane partials are combined in LDS
// in lane order (deterministic) before the piece partial is written.
__global__ void __launch_bounds__(DIR_TPB)
fds_piece_sums_narrow_kernel(const float* __restrict__ feats, int C, int tpr,
                             const int32_t* __restrict__ perm, const int32_t* __restrict__ offsets,
                             const int32_t* __restrict__ npieces, const int32_t* __restrict__ piece_bin,
                             const int32_t* __restrict__ piece_p0, const int32_t* __restrict__ piece_p1,
                             double* __restrict__ partials) {
    __shared__ double sh[2][DIR_TPB * 4];
    const int k = blockIdx.x;
    if (k >= *npieces) return;
    const int rl = DIR_TPB / tpr;                       // row lanes
    const int t = threadIdx.x, tc = t % tpr, tl = t / tpr;
    const int col = tc * 4;
    const int b = piece_bin[k], p0 = piece_p0[k], p1 = piece_p1[k];
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (tl < rl) {
        const float4 kv = *reinterpret_cast<const float4*>(feats + (size_t)perm[offsets[b]] * C + col);
        const double kd[4] = {(double)kv.x, (double)kv.y, (double)kv.z, (double)kv.w};
        int p = p0 + tl;
        for (; p + 3 * rl < p1; p += 4 * rl) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = fds_ldnt(feats + (size_t)perm[p + u * rl] * C + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double d0 = (double)v[u].x - kd[0], d1 = (double)v[u].y - kd[1], d2 = (double)v[u].z - kd[2], d3 = (double)v[u].w - kd[3];
                s1[0] += d0; s2[0] += d0 * d0; s1[1] += d1; s2[1] += d1 * d1; s1[2] += d2; s2[2] += d2 * d2; s1[3] += d3; s2[3] += d3 * d3;
            }
        }
        for (; p < p1; p += rl) {
            const float4 v = fds_ldnt(feats + (size_t)perm[p] * C + col);
            const double d0 = (double)v.x - kd[0], d1 = (double)v.y - kd[1], d2 = (double)v.z - kd[2], d3 = (double)v.w - kd[3];
            s1[0] += d0; s2[0] += d0 * d0; s1[1] += d1; s2[1] += d1 * d1; s1[2] += d2; s2[2] += d2 * d2; s1[3] += d3; s2[3] += d3 * d3;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[0][t * 4 + j] = s1[j]; sh[1][t * 4 + j] = s2[j]; }
    __syncthreads();
    if (tl == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a = 0.0, q = 0.0;
            for (int l = 0; l < rl; ++l) { a += sh[0][(l * tpr + tc) * 4 + j]; q += sh[1][(l * tpr + tc) * 4 + j]; }
            partials[((size_t)k * 2 + 0) * C + col + j] = a;
            partials[((size_t)k * 2 + 1) * C + col + j] = q;
        }
    }
}

// C: combine the pieces of each bin in index order.
__global__ void __launch_bounds__(DIR_TPB)
fds_combine_kernel(const float* __restrict__ feats, int C, int nb,
                   const int32_t* __restrict__ perm, const int32_t* __restrict__ offsets,
                   const int32_t* __restrict__ bin_piece0, const double* __restrict__ partials,
                   double* __restrict__ count, double* __restrict__ mean, double* __restrict__ m2) {
    const int b = blockIdx.x;
    const int col = blockIdx.y * DIR_TPB + threadIdx.x;
    const int o0 = offsets[b], cnt = offsets[b + 1] - o0;
    if (col == 0) count[b] = (double)cnt;
    if (col >= C) return;
    const size_t o = (size_t)b * C + col;
    if (cnt == 0) { mean[o] = 0.0; m2[o] = 0.0; return; }
    double s1 = 0.0, s2 = 0.0;
    const int k1 = bin_piece0[b + 1];
    int k = bin_piece0[b];
    for (; k + 4 <= k1; k += 4) {                           // 8 independent loads in flight; summed in piece order
        double a[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = partials[((size_t)(k + u) * 2 + 0) * C + col]; q[u] = partials[((size_t)(k + u) * 2 + 1) * C + col]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += a[u]; s2 += q[u]; }
    }
    for (; k < k1; ++k) {
        s1 += partials[((size_t)k * 2 + 0) * C + col];
        s2 += partials[((size_t)k * 2 + 1) * C + col];
    }
    const double n = (double)cnt;
    const double kd = (double)feats[(size_t)perm[o0] * C + col];
    mean[o] = kd + s1 / n;
    const double v = s2 - s1 * s1 / n;
    m2[o] = v > 0.0 ? v : 0.0;
}

extern "C" int dir_fds_scatter_stats(const void* feats, int dtype, const int32_t* bins, int n, int C, int nb,
                                     double* count, double* mean, double* m2,
                                     void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!bins && n > 0, DIR_EINVAL);
    DIR_RETURN_IF(!count || !mean || !m2 || n < 0 || C <= 0 || nb <= 0, DIR_EINVAL);
    DIR_RETURN_IF(!feats && n > 0, DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32, DIR_EUNSUPPORTED);
    hipStream_t s = dir_s(stream);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(double) * nb, s);
        if (e == hipSuccess) e = hipMemsetAsync(mean, 0, sizeof(double) * (size_t)nb * C, s);
        if (e == hipSuccess) e = hipMemsetAsync(m2, 0, sizeof(double) * (size_t)nb * C, s);
        return (int)e;
    }
    DIR_RETURN_IF(!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u), DIR_EINVAL);
    ScatterWs w = carve_ws(workspace, n, C, nb);
    DIR_RETURN_IF(workspace_bytes < w.bytes, DIR_EWORKSPACE);
    const float* f = static_cast<const float*>(feats);
    const size_t lds_nb = sizeof(int32_t) * (size_t)nb;
    DIR_RETURN_IF(lds_nb > 64 * 1024, DIR_EUNSUPPORTED);                          // (one LDS counter per bin in the grouping wavefronts: nb <= 16 384)

    hipLaunchKernelGGL(fds_group_hist_kernel, dim3(w.ntiles), dim3(DIR_WAVE), lds_nb, s, bins, n, nb, w.tile_hist);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(fds_group_scan_tiles_kernel, dim3(nb), dim3(DIR_TPB), 0, s, w.tile_hist, w.ntiles, nb, w.totals);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(fds_group_scan_kernel, dim3(1), dim3(DIR_TPB), 0, s,
                       w.totals, nb, w.maxpieces, w.offsets, w.bin_piece0, w.npieces,
                       w.piece_bin, w.piece_p0, w.piece_p1);
    DIR_LAUNCH_CHECK();
    int nbits = 0;
    while ((1 << nbits) < nb) ++nbits;
    hipLaunchKernelGGL(fds_group_place_kernel, dim3(w.ntiles), dim3(DIR_WAVE), lds_nb, s,
                       bins, n, nb, nbits, w.tile_hist, w.offsets, w.perm);
    DIR_LAUNCH_CHECK();
    const bool vec4 = (C % 4 == 0) && dir_aligned16(feats);
    const int tpr = C / 4;
    if (vec4 && tpr <= 128 && DIR_TPB % tpr == 0) {
        hipLaunchKernelGGL(fds_piece_sums_narrow_kernel, dim3(w.maxpieces), dim3(DIR_TPB), 0, s, f, C, tpr, w.perm, w.offsets,
                           w.npieces, w.piece_bin, w.piece_p0, w.piece_p1, w.partials);
    } else if (vec4) {
        dim3 grid(w.maxpieces, dir_cdiv(C, DIR_TPB * 4));
        hipLaunchKernelGGL(fds_piece_sums_kernel<4>, grid, dim3(DIR_TPB), 0, s, f, C, w.perm, w.offsets,
                           w.npieces, w.piece_bin, w.piece_p0, w.piece_p1, w.partials);
    } else {
        dim3 grid(w.maxpieces, dir_cdiv(C, DIR_TPB));
        hipLaunchKernelGGL(fds_piece_sums_kernel<1>, grid, dim3(DIR_TPB), 0, s, f, C, w.perm, w.offsets,
                           w.npieces, w.piece_bin, w.piece_p0, w.piece_p1, w.partials);
    }
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(fds_combine_kernel, dim3(nb, dir_cdiv(C, DIR_TPB)), dim3(DIR_TPB), 0, s,
                       f, C, nb, w.perm, w.offsets, w.bin_piece0, w.partials, count, mean, m2);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// =============================================================================================
// K3: momentum update of the running tables
// =============================================================================================
__global__ void __launch_bounds__(DIR_TPB)
fds_finalize_tables_kernel(const double* __restrict__ count, const double* __restrict__ mean,
                           const double* __restrict__ m2, int C, int factor_mode, double momentum,
                           float* __restrict__ running_mean, float* __restrict__ running_var,
                           const float* __restrict__ tracked) {
    const int b = blockIdx.x;
    const double n = count[b];
    if (n <= 0.0) return;                                   // bin unseen this epoch: keep (fds.py loop skips it)
    double factor;
    if (factor_mode == DIR_FACTOR_ZERO) factor = 0.0;
    else if (factor_mode == DIR_FACTOR_MOMENTUM) factor = momentum;
    else factor = 1.0 - n / (double)(tracked[b] + (float)n);    // fds.py:105-106 with the bumped counter
    const float a = (float)(1.0 - factor), f = (float)factor;
    for (int col = blockIdx.y * DIR_TPB + threadIdx.x; col < C; col += gridDim.y * DIR_TPB) {
        const size_t o = (size_t)b * C + col;
        const float cm = (float)mean[o];
        const float cv = (n == 1.0) ? 0.0f : (float)(m2[o] / (n - 1.0));   // fds.py:102
        running_mean[o] = a * cm + f * running_mean[o];
        running_var[o] = a * cv + f * running_var[o];
    }
}

__global__ void fds_finalize_tracked_kernel(const double* __restrict__ count, int nb, float* __restrict__ tracked) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb && count[b] > 0.0) tracked[b] = tracked[b] + (float)count[b];   // fds.py:104
}

extern "C" int dir_fds_finalize_update(const double* count, const double* mean, const double* m2, int nb, int C,
                                       int factor_mode, double momentum,
                                       float* running_mean, float* running_var, float* num_samples_tracked,
                                       dir_stream_t stream) {
    DIR_RETURN_IF(!count || !mean || !m2 || !running_mean || !running_var || !num_samples_tracked, DIR_EINVAL);
    DIR_RETURN_IF(nb <= 0 || C <= 0 || factor_mode < 0 || factor_mode > 2, DIR_EINVAL);
    int gy = dir_cdiv(C, DIR_TPB); if (gy > 64) gy = 64;
    hipLaunchKernelGGL(fds_finalize_tables_kernel, dim3(nb, gy), dim3(DIR_TPB), 0, dir_s(stream),
                       count, mean, m2, C, factor_mode, momentum, running_mean, running_var, num_samples_tracked);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(fds_finalize_tracked_kernel, dim3(dir_cdiv(nb, DIR_TPB)), dim3(DIR_TPB), 0, dir_s(stream),
                       count, nb, num_samples_tracked);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// K3 for NON-INTEGER labels (fds.py:91-111 iterates torch.unique(labels): ONE blend per distinct label VALUE, in ascending order, into
// bin int(value - bucket_start)). The statistics arrive per value GROUP (count / mean / m2 [U], groups sorted by value, so the groups of a bin are
// the contiguous range bin_ptr[b] .. bin_ptr[b + 1]); a thread owns one (bin, column) and applies the bin's blends one after the other, with the
// running num_samples_tracked the count-based factor is made from (fds.py:104-106) advancing group by group.
__global__ void __launch_bounds__(DIR_TPB)
fds_finalize_groups_tables_kernel(const double* __restrict__ count, const double* __restrict__ mean, const double* __restrict__ m2, int C,
                                  const int32_t* __restrict__ bin_ptr, int factor_mode, double momentum,
                                  float* __restrict__ running_mean, float* __restrict__ running_var, const float* __restrict__ tracked) {
    const int b = blockIdx.x;
    const int g0 = bin_ptr[b], g1 = bin_ptr[b + 1];
    if (g0 >= g1) return;
    for (int col = blockIdx.y * DIR_TPB + threadIdx.x; col < C; col += gridDim.y * DIR_TPB) {
        const size_t ob = (size_t)b * C + col;
        float rm = running_mean[ob], rv = running_var[ob], tr = tracked[b];
        for (int g = g0; g < g1; ++g) {
            const double n = count[g];
            if (n <= 0.0) continue;
            tr = tr + (float)n;                                            // fds.py:104 (float32 buffer)
            double factor;
            if (factor_mode == DIR_FACTOR_ZERO) factor = 0.0;
            else if (factor_mode == DIR_FACTOR_MOMENTUM) factor = momentum;
            else factor = 1.0 - n / (double)tr;                            // fds.py:105-106
            const float a = (float)(1.0 - factor), f = (float)factor;
            const size_t og = (size_t)g * C + col;
            const float cm = (float)mean[og];
            const float cv = (n == 1.0) ? 0.0f : (float)(m2[og] / (n - 1.0));   // fds.py:102
            rm = a * cm + f * rm;
            rv = a * cv + f * rv;
        }
        running_mean[ob] = rm;
        running_var[ob] = rv;
    }
}

__global__ void fds_finalize_groups_tracked_kernel(const double* __restrict__ count, const int32_t* __restrict__ bin_ptr, int nb,
                                                   float* __restrict__ tracked) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    float tr = tracked[b];
    for (int g = bin_ptr[b]; g < bin_ptr[b + 1]; ++g)
        if (count[g] > 0.0) tr = tr + (float)count[g];
    tracked[b] = tr;
}

extern "C" int dir_fds_finalize_update_groups(const double* count, const double* mean, const double* m2, int ngroups, int C,
                                              const int32_t* bin_ptr, int nb, int factor_mode, double momentum,
                                              float* running_mean, float* running_var, float* num_samples_tracked, dir_stream_t stream) {
    DIR_RETURN_IF(!count || !mean || !m2 || !bin_ptr || !running_mean || !running_var || !num_samples_tracked, DIR_EINVAL);
    DIR_RETURN_IF(ngroups <= 0 || nb <= 0 || C <= 0 || factor_mode < 0 || factor_mode > 2, DIR_EINVAL);
    int gy = dir_cdiv(C, DIR_TPB); if (gy > 64) gy = 64;
    hipLaunchKernelGGL(fds_finalize_groups_tables_kernel, dim3(nb, gy), dim3(DIR_TPB), 0, dir_s(stream),
                       count, mean, m2, C, bin_ptr, factor_mode, momentum, running_mean, running_var, num_samples_tracked);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(fds_finalize_groups_tracked_kernel, dim3(dir_cdiv(nb, DIR_TPB)), dim3(DIR_TPB), 0, dir_s(stream),
                       count, bin_ptr, nb, num_samples_tracked);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// =============================================================================================
// K4: smoothing across bins (reflect padding), mean and var tables in one launch
// =============================================================================================
__device__ __forceinline__ int reflect_idx(int i, int nb) {
    if (i < 0) i = -i;
    if (i >= nb) i = 2 * (nb - 1) - i;
    return i;
}

__global__ void __launch_bounds__(DIR_TPB)
fds_smooth_bins_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                       const float* __restrict__ window, int ks, int nb, int C,
                       float* __restrict__ smean, float* __restrict__ svar) {
    const int b = blockIdx.x;
    const float* __restrict__ in = blockIdx.z ? var : mean;
    float* __restrict__ out = blockIdx.z ? svar : smean;
    const int h = ks / 2;
    for (int col = blockIdx.y * DIR_TPB + threadIdx.x; col < C; col += gridDim.y * DIR_TPB) {
        float acc = 0.0f;
        for (int k = 0; k < ks; ++k) {
            const int r = reflect_idx(b + k - h, nb);        // block-uniform
            acc = acc + window[k] * in[(size_t)r * C + col];
        }
        out[(size_t)b * C + col] = acc;
    }
}

extern "C" int dir_fds_smooth_bins(const float* mean, const float* var, const float* window, int ks, int nb, int C,
                                   float* smoothed_mean, float* smoothed_var, dir_stream_t stream) {
    DIR_RETURN_IF(!mean || !var || !window || !smoothed_mean || !smoothed_var, DIR_EINVAL);
    DIR_RETURN_IF(nb <= 0 || C <= 0 || ks <= 0 || (ks & 1) == 0 || ks / 2 >= nb, DIR_EINVAL);
    DIR_RETURN_IF(smoothed_mean == mean || smoothed_var == var, DIR_EINVAL);
    int gy = dir_cdiv(C, DIR_TPB); if (gy > 64) gy = 64;
    hipLaunchKernelGGL(fds_smooth_bins_kernel, dim3(nb, gy, 2), dim3(DIR_TPB), 0, dir_s(stream),
                       mean, var, window, ks, nb, C, smoothed_mean, smoothed_var);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// =============================================================================================
// K5a: multiplier table
// =============================================================================================
__global__ void __launch_bounds__(DIR_TPB)
fds_prepare_scale_kernel(const float* __restrict__ v1, const float* __restrict__ v2, int C,
                         float clip_min, float clip_max, float* __restrict__ scale) {
    __shared__ double wsum[DIR_TPB / DIR_WAVE];
    const int b = blockIdx.x;
    const float* r1 = v1 + (size_t)b * C;
    const float* r2 = v2 + (size_t)b * C;
    double s = 0.0;
    for (int c = threadIdx.x; c < C; c += DIR_TPB) s += (double)r1[c];
    s = dir_wave_sum(s);
    if ((threadIdx.x & (DIR_WAVE - 1)) == 0) wsum[threadIdx.x / DIR_WAVE] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < DIR_TPB / DIR_WAVE; ++i) tot += wsum[i];
    const bool row_identity = tot < 1e-10;                  // utils.py:98-99
    for (int c = threadIdx.x; c < C; c += DIR_TPB) {
        const float a = r1[c];
        float out;
        if (row_identity || a == 0.0f) out = -1.0f;          // utils.py:100-104
        else {
            float q = r2[c] / a;
            q = (q < clip_min) ? clip_min : ((q > clip_max) ? clip_max : q);   // NaN falls through
            out = sqrtf(q);
        }
        scale[(size_t)b * C + c] = out;
    }
}

extern "C" int dir_fds_prepare_scale(const float* v1, const float* v2, int nb, int C, float clip_min, float clip_max,
                                     float* scale, dir_stream_t stream) {
    DIR_RETURN_IF(!v1 || !v2 || !scale || nb <= 0 || C <= 0, DIR_EINVAL);
    hipLaunchKernelGGL(fds_prepare_scale_kernel, dim3(nb), dim3(DIR_TPB), 0, dir_s(stream),
                       v1, v2, C, clip_min, clip_max, scale);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// =============================================================================================
// K5 / K6: calibration forward (in place) and backward; fused label->bin variant for a batch
// =============================================================================================
__device__ __forceinline__ float calib1(float x, float m1, float s, float m2) {
    return (s < 0.0f) ? x : (x - m1) * s + m2;               // utils.py:107, contraction off
}

template <int VEC, bool FUSED>
__global__ void __launch_bounds__(DIR_TPB)
fds_calibrate_fwd_kernel(float* __restrict__ x, const int32_t* __restrict__ bins_in,
                         const float* __restrict__ labels, int B, int C, float lo, float hi,
                         const float* __restrict__ m1, const float* __restrict__ scale,
                         const float* __restrict__ m2, int32_t* __restrict__ bins_out) {
    const int row = blockIdx.x;
    int bin;
    if (FUSED) {
        // every workgroup rescans the (small) label vector for the two presence flags (A.3)
        int has_lo = 0, has_hi = 0;
        for (int i = threadIdx.x; i < B; i += DIR_TPB) {
            const float l = labels[i];
            has_lo |= (l == lo); has_hi |= (l == hi);
        }
        has_lo = __syncthreads_or(has_lo);                  // predicate OR (returns a boolean, not a bit-OR)
        has_hi = __syncthreads_or(has_hi);
        bin = dir_bin_of(labels[row], lo, hi, has_lo != 0, has_hi != 0);
        if (blockIdx.y == 0 && threadIdx.x == 0) bins_out[row] = bin;
    } else {
        bin = bins_in[row];
    }
    if (bin < 0) return;
    const size_t xo = (size_t)row * C, to = (size_t)bin * C;
    if (VEC == 4) {
        const int col = (blockIdx.y * DIR_TPB + threadIdx.x) * 4;
        if (col >= C) return;
        // the row is streamed (non-temporal: read once, written once) so that the three [Nb, C] tables stay L2 resident
        typedef __attribute__((ext_vector_type(4))) float f32x4_t;
        f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(x + xo + col));
        const float4 a = *reinterpret_cast<const float4*>(m1 + to + col);
        const float4 s = *reinterpret_cast<const float4*>(scale + to + col);
        const float4 c = *reinterpret_cast<const float4*>(m2 + to + col);
        v.x = calib1(v.x, a.x, s.x, c.x); v.y = calib1(v.y, a.y, s.y, c.y);
        v.z = calib1(v.z, a.z, s.z, c.z); v.w = calib1(v.w, a.w, s.w, c.w);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(x + xo + col));
    } else {
        const int col = blockIdx.y * DIR_TPB + threadIdx.x;
        if (col >= C) return;
        x[xo + col] = calib1(x[xo + col], m1[to + col], scale[to + col], m2[to + col]);
    }
}

template <int VEC>
__global__ void __launch_bounds__(DIR_TPB)
fds_calibrate_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                         const int32_t* __restrict__ bins, int C, const float* __restrict__ scale) {
    const int row = blockIdx.x;
    const int bin = bins[row];
    const size_t xo = (size_t)row * C, to = (size_t)(bin < 0 ? 0 : bin) * C;
    if (VEC == 4) {
        const int col = (blockIdx.y * DIR_TPB + threadIdx.x) * 4;
        if (col >= C) return;
        float4 g = fds_ldnt(dy + xo + col);
        if (bin >= 0) {
            const float4 s = *reinterpret_cast<const float4*>(scale + to + col);
            g.x = s.x < 0.0f ? g.x : g.x * s.x; g.y = s.y < 0.0f ? g.y : g.y * s.y;
            g.z = s.z < 0.0f ? g.z : g.z * s.z; g.w = s.w < 0.0f ? g.w : g.w * s.w;
        }
        fds_stnt(dx + xo + col, g);
    } else {
        const int col = blockIdx.y * DIR_TPB + threadIdx.x;
        if (col >= C) return;
        float g = dy[xo + col];
        if (bin >= 0) { const float s = scale[to + col]; g = s < 0.0f ? g : g * s; }
        dx[xo + col] = g;
    }
}

// Narrow rows (C <= 512): several rows per workgroup, per-thread bin lookup (dense NYUD2 features: every pixel has
// its own depth bucket). BWD = false: x = (x - m1) * s + m2 in place; BWD = true: dx = dy * s.
template <bool BWD>
__global__ void __launch_bounds__(DIR_TPB)
fds_calibrate_narrow_kernel(float* __restrict__ x, const float* __restrict__ dy, const int32_t* __restrict__ bins,
                            long long B, int C, int tpr, const float* __restrict__ m1, const float* __restrict__ scale,
                            const float* __restrict__ m2) {
    const int rpb = DIR_TPB / tpr;
    const int t = threadIdx.x, tc = t % tpr, tl = t / tpr;
    if (tl >= rpb) return;
    const int col = tc * 4;
    for (long long row = (long long)blockIdx.x * rpb + tl; row < B; row += (long long)gridDim.x * rpb) {
        const int bin = bins[row];
        const size_t xo = (size_t)row * C + col;
        if (BWD) {
            float4 g = fds_ldnt(dy + xo);
            if (bin >= 0) {
                const float4 sc = *reinterpret_cast<const float4*>(scale + (size_t)bin * C + col);
                g.x = sc.x < 0.0f ? g.x : g.x * sc.x; g.y = sc.y < 0.0f ? g.y : g.y * sc.y;
                g.z = sc.z < 0.0f ? g.z : g.z * sc.z; g.w = sc.w < 0.0f ? g.w : g.w * sc.w;
            }
            fds_stnt(x + xo, g);
        } else {
            if (bin < 0) continue;
            const size_t to = (size_t)bin * C + col;
            float4 v = fds_ldnt(x + xo);
            const float4 a = *reinterpret_cast<const float4*>(m1 + to);
            const float4 sc = *reinterpret_cast<const float4*>(scale + to);
            const float4 c = *reinterpret_cast<const float4*>(m2 + to);
            v.x = calib1(v.x, a.x, sc.x, c.x); v.y = calib1(v.y, a.y, sc.y, c.y);
            v.z = calib1(v.z, a.z, sc.z, c.z); v.w = calib1(v.w, a.w, sc.w, c.w);
            fds_stnt(x + xo, v);
        }
    }
}

static bool vec4_ok(int C, const void* a, const void* b, const void* c, const void* d) {
    return (C % 4 == 0) && dir_aligned16(a) && dir_aligned16(b) && dir_aligned16(c) && dir_aligned16(d);
}

extern "C" int dir_fds_calibrate_fwd(void* x_inout, int dtype, const int32_t* bins, int B, int C,
                                     const float* m1, const float* scale, const float* m2, dir_stream_t stream) {
    DIR_RETURN_IF(B < 0 || C <= 0, DIR_EINVAL);
    if (B == 0) return DIR_OK;
    DIR_RETURN_IF(!x_inout || !bins || !m1 || !scale || !m2, DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32, DIR_EUNSUPPORTED);
    float* x = static_cast<float*>(x_inout);
    if (vec4_ok(C, x, m1, scale, m2) && C / 4 <= 128 && DIR_TPB % (C / 4) == 0) {
        const int tpr = C / 4, rpb = DIR_TPB / tpr;
        int grid = dir_cdiv(B, rpb); if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(fds_calibrate_narrow_kernel<false>, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), x, nullptr, bins,
                           (long long)B, C, tpr, m1, scale, m2);
    } else if (vec4_ok(C, x, m1, scale, m2)) {
        hipLaunchKernelGGL((fds_calibrate_fwd_kernel<4, false>), dim3(B, dir_cdiv(C, DIR_TPB * 4)), dim3(DIR_TPB), 0,
                           dir_s(stream), x, bins, nullptr, B, C, 0.f, 0.f, m1, scale, m2, nullptr);
    } else {
        hipLaunchKernelGGL((fds_calibrate_fwd_kernel<1, false>), dim3(B, dir_cdiv(C, DIR_TPB)), dim3(DIR_TPB), 0,
                           dir_s(stream), x, bins, nullptr, B, C, 0.f, 0.f, m1, scale, m2, nullptr);
    }
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_fds_calibrate_bwd(const void* dy, void* dx, int dtype, const int32_t* bins, int B, int C,
                                     const float* scale, dir_stream_t stream) {
    DIR_RETURN_IF(B < 0 || C <= 0, DIR_EINVAL);
    if (B == 0) return DIR_OK;
    DIR_RETURN_IF(!dy || !dx || !bins || !scale, DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32, DIR_EUNSUPPORTED);
    const float* g = static_cast<const float*>(dy);
    float* o = static_cast<float*>(dx);
    if (vec4_ok(C, g, o, scale, scale) && C / 4 <= 128 && DIR_TPB % (C / 4) == 0) {
        const int tpr = C / 4, rpb = DIR_TPB / tpr;
        int grid = dir_cdiv(B, rpb); if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(fds_calibrate_narrow_kernel<true>, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), o, g, bins,
                           (long long)B, C, tpr, nullptr, scale, nullptr);
    } else if (vec4_ok(C, g, o, scale, scale)) {
        hipLaunchKernelGGL(fds_calibrate_bwd_kernel<4>, dim3(B, dir_cdiv(C, DIR_TPB * 4)), dim3(DIR_TPB), 0,
                           dir_s(stream), g, o, bins, C, scale);
    } else {
        hipLaunchKernelGGL(fds_calibrate_bwd_kernel<1>, dim3(B, dir_cdiv(C, DIR_TPB)), dim3(DIR_TPB), 0,
                           dir_s(stream), g, o, bins, C, scale);
    }
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// ---------------------------------------------------------------------------------------------
// Calibration with the bin-statistic tables STAGED IN LDS (large batches: the per-row kernel above re-reads its three
// table rows through the L2 for every feature row — 4 x the HBM bytes at B = 65 536 — and stops at 5.0 TB/s).
// A workgroup owns a TILE of TW = 128 columns and keeps that tile of ALL nb rows of the three tables resident
// (3 x nb x 128 floats = 150 KB at nb = 100: one workgroup per CU, 1024 threads), then streams its range of feature rows in
// their natural order: 32 lanes cover the 512-byte slice of a row, every lane keeps four rows in flight (non-temporal 16-byte
// loads / stores), and the table values come from LDS with the row's bin as the index — conflict free (16 lanes of a
// ds_read_b128 group read 256 contiguous bytes). No sort, no atomics, the arithmetic of calib1 unchanged: bit-identical results.
// Replaces imdb-wiki-dir/fds.py:120-143 / utils.py:97-107 for B >= FDS_LDS_MIN_ROWS and the NHWC form of the NYUD2 map.
// ---------------------------------------------------------------------------------------------
#define FDS_LDS_TPB 1024
#define FDS_LDS_MIN_ROWS 4096
template <int TW, int TPB = FDS_LDS_TPB>
__global__ void __launch_bounds__(TPB)
fds_calibrate_lds_kernel(float* __restrict__ x, const int32_t* __restrict__ bins, long long B, int C, int nb,
                         const float* __restrict__ m1, const float* __restrict__ scale, const float* __restrict__ m2,
                         long long rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float fds_tab[];      // [3][nb][TW]
    constexpr int TPR = TW / 4;                                          // lanes per row slice
    const int t = threadIdx.x;
    const int col0 = blockIdx.x * TW;
    {
        const float* src[3] = {m1, scale, m2};
        const int per = nb * TPR;
        for (int i = t; i < 3 * per; i += TPB) {
            const int which = i / per, rem = i - which * per, b = rem / TPR, c4 = rem - b * TPR;
            float4 v = make_float4(0.f, -1.f, 0.f, 0.f);
            if (col0 + c4 * 4 < C) v = *reinterpret_cast<const float4*>(src[which] + (size_t)b * C + col0 + c4 * 4);
            *reinterpret_cast<float4*>(fds_tab + ((size_t)which * nb + b) * TW + c4 * 4) = v;
        }
    }
    __syncthreads();
    const int tc = t % TPR, tl = t / TPR;
    constexpr int RPP = TPB / TPR;                               // rows per pass of the workgroup
    const int col = col0 + tc * 4;
    if (col >= C) return;
    const long long r0 = (long long)blockIdx.y * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < B) ? r0 + rows_per_wg : B;
    const float4* t1 = reinterpret_cast<const float4*>(fds_tab) + tc;
    const float4* ts = reinterpret_cast<const float4*>(fds_tab + (size_t)nb * TW) + tc;
    const float4* t2 = reinterpret_cast<const float4*>(fds_tab + (size_t)2 * nb * TW) + tc;
    for (long long row = r0 + tl; row < r1; row += 4 * RPP) {
        int bn[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = row + (long long)u * RPP;
            bn[u] = (r < r1) ? bins[r] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = row + (long long)u * RPP;
            v[u] = (bn[u] >= 0) ? fds_ldnt(x + (size_t)r * C + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (bn[u] < 0) continue;
            const long long r = row + (long long)u * RPP;
            const float4 a = t1[bn[u] * TPR], sc = ts[bn[u] * TPR], c = t2[bn[u] * TPR];
            float4 o;
            o.x = calib1(v[u].x, a.x, sc.x, c.x); o.y = calib1(v[u].y, a.y, sc.y, c.y);
            o.z = calib1(v[u].z, a.z, sc.z, c.z); o.w = calib1(v[u].w, a.w, sc.w, c.w);
            fds_stnt(x + (size_t)r * C + col, o);
        }
    }
}

static int fds_calibrate_lds_launch(float* x, const int32_t* bins, long long B, int C, int nb, const float* m1, const float* scale,
                                    const float* m2, hipStream_t s) {
    // widest column tile whose three table slices fit the 160 KB of a CU (64 KB when that allows two workgroups per CU)
    int tw = 128;
    while (tw > 16 && (size_t)3 * nb * tw * 4 > 160 * 1024) tw >>= 1;
    if ((size_t)3 * nb * tw * 4 > 160 * 1024) return -1;
    const int nct = dir_cdiv(C, tw);
    int ny = dir_cdiv(512, nct);
    const long long min_rows = 4 * (FDS_LDS_TPB / (tw / 4));
    if ((long long)ny * min_rows > B) ny = (int)((B + min_rows - 1) / min_rows);
    if (ny < 1) ny = 1;
    const long long rows_per_wg = (B + ny - 1) / ny;
    const size_t lds = (size_t)3 * nb * tw * 4;
#define FDS_LDS_GO(TW_)                                                                                                        \
    {                                                                                                                          \
        DIR_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(fds_calibrate_lds_kernel<TW_>),            \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                \
        hipLaunchKernelGGL(fds_calibrate_lds_kernel<TW_>, dim3(nct, ny), dim3(FDS_LDS_TPB), lds, s, x, bins, B, C, nb, m1, scale, m2, rows_per_wg); \
    }
    if (tw == 128) FDS_LDS_GO(128) else if (tw == 64) FDS_LDS_GO(64) else if (tw == 32) FDS_LDS_GO(32) else FDS_LDS_GO(16)
#undef FDS_LDS_GO
    return 0;
}

// Forward calibration of B rows with the tables staged in LDS; nb = rows of the tables (bins are in [0, nb) or < 0 = row untouched).
// Falls back to dir_fds_calibrate_fwd when the layout does not allow 16-byte accesses or nb is too large for a CU's LDS.
extern "C" int dir_fds_calibrate_fwd_lds(void* x_inout, int dtype, const int32_t* bins, long long B, int C, int nb,
                                         const float* m1, const float* scale, const float* m2, dir_stream_t stream) {
    DIR_RETURN_IF(B < 0 || C <= 0 || nb <= 0, DIR_EINVAL);
    if (B == 0) return DIR_OK;
    DIR_RETURN_IF(!x_inout || !bins || !m1 || !scale || !m2, DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32, DIR_EUNSUPPORTED);
    float* x = static_cast<float*>(x_inout);
    if (vec4_ok(C, x, m1, scale, m2) && fds_calibrate_lds_launch(x, bins, B, C, nb, m1, scale, m2, dir_s(stream)) == 0) {
        DIR_LAUNCH_CHECK();
        return DIR_OK;
    }
    DIR_RETURN_IF(B > 0x7fffffffll, DIR_EUNSUPPORTED);
    return dir_fds_calibrate_fwd(x_inout, dtype, bins, (int)B, C, m1, scale, m2, stream);
}

// ---------------------------------------------------------------------------------------------
// NYUD2-DIR dense variant on the network's own NCHW feature map (nyud2-dir/models/fds.py:128-149 permutes it to [B*H*W, C],
// calibrates per pixel and permutes back: two 284 MB copies around the pass). Here the map is read and written in place of
// those copies: a lane owns FOUR consecutive pixels of a channel plane (16-byte coalesced accesses along W), walks the C
// channels, and looks the (bin, channel) table values up in LDS, where all three tables live TRANSPOSED as [C][nbp] so that the
// lanes of a wavefront — same channel, neighbouring pixels, nearby bins — hit different banks (or broadcast).
// BWD: dx = dy * scale (scale < 0: untouched column) with only the multiplier table staged.
// ---------------------------------------------------------------------------------------------
// Channel split (round 4): one workgroup covers 4096 pixels, so the NYUD2 map [32, 128, 114, 152] is only 136 pixel blocks — half the CUs
// idle, and the 147 KB of tables allow one workgroup per CU. blockIdx.y picks a channel GROUP of CG = C / gridDim.y channels: the
// workgroup stages only its group's table slices (73 KB at two groups: two workgroups per CU) and walks only those channel planes.
template <bool BWD>
__global__ void __launch_bounds__(FDS_LDS_TPB)
fds_calibrate_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, const int32_t* __restrict__ bins,
                          long long npix, int C, int HW, int nb, int nbp, int CG,
                          const float* __restrict__ m1, const float* __restrict__ scale, const float* __restrict__ m2) {
    extern __shared__ __attribute__((aligned(16))) float fds_tab[];      // [T][CG][nbp], T = 3 (forward: m1, scale, m2) or 1 (scale)
    const int t = threadIdx.x;
    const int c0 = blockIdx.y * CG;
    {
        const float* src[3] = {BWD ? scale : m1, scale, m2};
        const int T = BWD ? 1 : 3;
        for (int i = t; i < T * nb * CG; i += FDS_LDS_TPB) {             // coalesced over the [nb][c0 .. c0 + CG) slices of the source
            const int which = i / (nb * CG), rem = i - which * nb * CG, b = rem / CG, c = rem - b * CG;
            fds_tab[((size_t)which * CG + c) * nbp + b] = src[which][(size_t)b * C + c0 + c];
        }
    }
    __syncthreads();
    const float* t1 = fds_tab;
    const float* ts = BWD ? fds_tab : fds_tab + (size_t)CG * nbp;
    const float* t2 = fds_tab + (size_t)2 * CG * nbp;
    const long long nquad = npix >> 2;
    for (long long q = (long long)blockIdx.x * FDS_LDS_TPB + t; q < nquad; q += (long long)gridDim.x * FDS_LDS_TPB) {
        const long long p = q << 2;
        const long long n = p / HW;
        const int hw = (int)(p - n * HW);
        const int4 b4 = *reinterpret_cast<const int4*>(bins + p);
        const int bn[4] = {b4.x, b4.y, b4.z, b4.w};
        const size_t base = ((size_t)n * C + c0) * HW + hw;
#pragma unroll 4
        for (int c = 0; c < CG; ++c) {
            const float4 v = fds_ldnt(x + base + (size_t)c * HW);
            float in[4] = {v.x, v.y, v.z, v.w}, out[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (bn[k] < 0) { out[k] = in[k]; continue; }
                const float sc = ts[c * nbp + bn[k]];
                if (BWD) out[k] = sc < 0.0f ? in[k] : in[k] * sc;
                else out[k] = calib1(in[k], t1[c * nbp + bn[k]], sc, t2[c * nbp + bn[k]]);
            }
            fds_stnt(y + base + (size_t)c * HW, make_float4(out[0], out[1], out[2], out[3]));
        }
    }
}

// Plane-chunk form (round 4; nb <= 128): short-lived 256-thread workgroups in ADDRESS order, each moving one contiguous 16 KB chunk of
// ONE channel plane (four 4 KB pieces in flight per workgroup) — the access pattern a bare stream prefers on this part (6.3 TB/s
// against 4.2-5.1 for grid-stride sweeps, profiles/r03_stream_patterns.txt). A plane needs only its channel's COLUMN of the tables:
// 3 x nb floats in LDS (1.1 KB, fetched from the L2-resident [nb][C] tables), the bins of the chunk's pixels come from the L2 too
// (re-read once per channel: 2.2 MB x 128 at NYUD2 shapes). Same calib1 arithmetic: bit-identical to the kernel above.
template <bool BWD>
__global__ void __launch_bounds__(DIR_TPB)
fds_calibrate_nchw_plane_kernel(const float* __restrict__ x, float* __restrict__ y, const int32_t* __restrict__ bins,
                                int C, int HW, int nb, int chunks,
                                const float* __restrict__ m1, const float* __restrict__ scale, const float* __restrict__ m2) {
    __shared__ float tab[3][128];
    const int t = threadIdx.x;
    const int chunk = blockIdx.x % chunks;
    const long long plane = blockIdx.x / chunks;                          // n * C + c
    const int c = (int)(plane % C);
    const long long n = plane / C;
    if (t < nb) {
        tab[1][t] = scale[(size_t)t * C + c];
        if (!BWD) { tab[0][t] = m1[(size_t)t * C + c]; tab[2][t] = m2[(size_t)t * C + c]; }
    }
    __syncthreads();
    const float* xp = x + (size_t)plane * HW;
    float* yp = y + (size_t)plane * HW;
    const int32_t* bp = bins + (size_t)n * HW;
    const int p0 = chunk * (4 * DIR_TPB * 4) + t * 4;                     // 4096 pixels per chunk: four pieces of 1024
    float4 v[4];
    int4 b4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = p0 + u * DIR_TPB * 4;
        if (p < HW) { v[u] = fds_ldnt(xp + p); b4[u] = *reinterpret_cast<const int4*>(bp + p); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = p0 + u * DIR_TPB * 4;
        if (p >= HW) continue;
        const float in[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const int bn[4] = {b4[u].x, b4[u].y, b4[u].z, b4[u].w};
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (bn[k] < 0) { out[k] = in[k]; continue; }
            const float sc = tab[1][bn[k]];
            if (BWD) out[k] = sc < 0.0f ? in[k] : in[k] * sc;
            else out[k] = calib1(in[k], tab[0][bn[k]], sc, tab[2][bn[k]]);
        }
        fds_stnt(yp + p, make_float4(out[0], out[1], out[2], out[3]));
    }
}

// x, y: [N, C, HW] float32 (NCHW maps, HW = H * W a multiple of 4, 16-byte aligned); bins: [N * HW] int32 (< 0: pixel copied
// unchanged). y may alias x. Returns DIR_EUNSUPPORTED when the tables do not fit a CU's LDS or the layout does not allow
// 16-byte accesses (the caller then takes the row form).
static int fds_nchw_launch(bool bwd, const float* x, float* y, const int32_t* bins, long long N, int C, int HW, int nb,
                           const float* m1, const float* scale, const float* m2, dir_stream_t stream) {
    DIR_RETURN_IF(N < 0 || C <= 0 || HW <= 0 || nb <= 0, DIR_EINVAL);
    if (N == 0) return DIR_OK;
    DIR_RETURN_IF(!x || !y || !bins || !scale || (!bwd && (!m1 || !m2)), DIR_EINVAL);
    DIR_RETURN_IF((HW & 3) || !dir_aligned16(x) || !dir_aligned16(y) || !dir_aligned16(bins), DIR_EUNSUPPORTED);
    if (nb <= 128 && N * (long long)C * dir_cdiv(HW, 4096) < (1ll << 31)) {
        const int chunks = dir_cdiv(HW, 4096);
        const unsigned grid = (unsigned)(N * C * chunks);
        if (bwd) hipLaunchKernelGGL(fds_calibrate_nchw_plane_kernel<true>, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), x, y, bins, C, HW, nb, chunks, m1, scale, m2);
        else hipLaunchKernelGGL(fds_calibrate_nchw_plane_kernel<false>, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), x, y, bins, C, HW, nb, chunks, m1, scale, m2);
        DIR_LAUNCH_CHECK();
        return DIR_OK;
    }
    const int nbp = (nb + 3) & ~3;
    const long long npix = N * (long long)HW;
    long long want = (npix / 4 + FDS_LDS_TPB - 1) / FDS_LDS_TPB;         // pixel blocks of 4096
    // channel groups: as many as it takes to put >= 2 workgroups on every CU (while a group keeps >= 16 channels and C divides), and
    // at least enough for the group's table slices to fit a CU's LDS
    int groups = 1;
    while ((size_t)(bwd ? 1 : 3) * (C / groups) * nbp * 4 > 160 * 1024 && (C / groups) % 2 == 0 && C / groups > 16) groups *= 2;
    while (want * groups < 512 && (C / groups) % 2 == 0 && C / groups > 16) groups *= 2;
    const int CG = C / groups;
    const size_t lds = (size_t)(bwd ? 1 : 3) * CG * nbp * 4;
    DIR_RETURN_IF(lds > 160 * 1024, DIR_EUNSUPPORTED);
    // one workgroup per pixel block while that stays under ~2048 workgroups (the dispatcher balances them over the 256-512 resident
    // slots: two 1024-thread workgroups per CU when the slices take <= 80 KB); beyond that a grid-stride sweep amortises the table staging
    long long cap = (2048 + groups - 1) / groups;
    int grid = (int)(want < cap ? (want < 1 ? 1 : want) : cap);
    DIR_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(fds_calibrate_nchw_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fds_calibrate_nchw_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (bwd) hipLaunchKernelGGL(fds_calibrate_nchw_kernel<true>, dim3(grid, groups), dim3(FDS_LDS_TPB), lds, dir_s(stream), x, y, bins, npix, C, HW, nb, nbp, CG, m1, scale, m2);
    else hipLaunchKernelGGL(fds_calibrate_nchw_kernel<false>, dim3(grid, groups), dim3(FDS_LDS_TPB), lds, dir_s(stream), x, y, bins, npix, C, HW, nb, nbp, CG, m1, scale, m2);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_fds_calibrate_fwd_nchw(const void* x, void* y, int dtype, const int32_t* bins, long long N, int C, int HW, int nb,
                                          const float* m1, const float* scale, const float* m2, dir_stream_t stream) {
    DIR_RETURN_IF(dtype != DIR_F32, DIR_EUNSUPPORTED);
    return fds_nchw_launch(false, static_cast<const float*>(x), static_cast<float*>(y), bins, N, C, HW, nb, m1, scale, m2, stream);
}

extern "C" int dir_fds_calibrate_bwd_nchw(const void* dy, void* dx, int dtype, const int32_t* bins, long long N, int C, int HW, int nb,
                                          const float* scale, dir_stream_t stream) {
    DIR_RETURN_IF(dtype != DIR_F32, DIR_EUNSUPPORTED);
    return fds_nchw_launch(true, static_cast<const float*>(dy), static_cast<float*>(dx), bins, N, C, HW, nb, nullptr, scale, nullptr, stream);
}

#define SMOOTH_FUSED_MAX_B 2048

extern "C" int dir_fds_smooth_fwd(void* x_inout, int dtype, const float* labels, int B, int C,
                                  int bucket_start, int bucket_num,
                                  const float* m1, const float* scale, const float* m2,
                                  int32_t* bins_out, dir_stream_t stream) {
    DIR_RETURN_IF(B < 0 || C <= 0, DIR_EINVAL);
    if (B == 0) return DIR_OK;
    DIR_RETURN_IF(!x_inout || !labels || !m1 || !scale || !m2 || !bins_out, DIR_EINVAL);
    DIR_RETURN_IF(check_buckets(bucket_start, bucket_num), DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32, DIR_EUNSUPPORTED);
    float* x = static_cast<float*>(x_inout);
    const float lo = (float)bucket_start, hi = (float)(bucket_num - 1);
    if (B > SMOOTH_FUSED_MAX_B) {
        // large batches: a per-workgroup rescan of the labels would be O(B^2) -> K1 then K5.
        // bins_out has B + 1 slots by contract; the extra one holds the presence-flag word.
        uint32_t* flags = reinterpret_cast<uint32_t*>(bins_out + B);
        int rc = dir_fds_bin_index(labels, B, bucket_start, bucket_num, bins_out, flags, stream);
        if (rc) return rc;
        if (B >= FDS_LDS_MIN_ROWS)      // tables staged in LDS, rows streamed in their natural order
            return dir_fds_calibrate_fwd_lds(x_inout, dtype, bins_out, B, C, bucket_num - bucket_start, m1, scale, m2, stream);
        return dir_fds_calibrate_fwd(x_inout, dtype, bins_out, B, C, m1, scale, m2, stream);
    }
    if (vec4_ok(C, x, m1, scale, m2)) {
        hipLaunchKernelGGL((fds_calibrate_fwd_kernel<4, true>), dim3(B, dir_cdiv(C, DIR_TPB * 4)), dim3(DIR_TPB), 0,
                           dir_s(stream), x, nullptr, labels, B, C, lo, hi, m1, scale, m2, bins_out);
    } else {
        hipLaunchKernelGGL((fds_calibrate_fwd_kernel<1, true>), dim3(B, dir_cdiv(C, DIR_TPB)), dim3(DIR_TPB), 0,
                           dir_s(stream), x, nullptr, labels, B, C, lo, hi, m1, scale, m2, bins_out);
    }
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// =============================================================================================
// STS-B variant (sts-b-dir/fds.py, sts-b-dir/util.py:63-73): histogram-edge bins, guarded calibration,
// empty-bucket fill
// =============================================================================================
// bin = bucket_num - 1 if label == last edge, else max((first i with edges[i] > label) - 1, bucket_start)
// (fds.py:51-57 `_get_bucket_idx`); -1 (row skipped) where the reference would raise (label beyond the last edge, NaN).
__global__ void __launch_bounds__(DIR_TPB)
fds_bin_edges_kernel(const float* __restrict__ labels, int n, const float* __restrict__ edges, int nedges,
                     int bucket_start, int bucket_num, int32_t* __restrict__ bins) {
    extern __shared__ float sh_edges[];
    for (int i = threadIdx.x; i < nedges; i += DIR_TPB) sh_edges[i] = edges[i];
    __syncthreads();
    for (int i = blockIdx.x * DIR_TPB + threadIdx.x; i < n; i += gridDim.x * DIR_TPB) {
        const float l = labels[i];
        int b = -1;
        if (l == sh_edges[nedges - 1]) b = bucket_num - 1;
        else {
            int first = -1;
            for (int k = 0; k < nedges; ++k) if (sh_edges[k] > l) { first = k; break; }
            if (first >= 0) { b = first - 1; if (b < bucket_start) b = bucket_start; }
        }
        bins[i] = (b < 0) ? -1 : b - bucket_start;
    }
}

extern "C" int dir_fds_bin_edges(const float* labels, int n, const float* edges, int nedges, int bucket_start,
                                 int bucket_num, int32_t* bins, dir_stream_t stream) {
    DIR_RETURN_IF(!labels || !edges || !bins || n < 0 || nedges < 2 || nedges > 4096, DIR_EINVAL);
    DIR_RETURN_IF(check_buckets(bucket_start, bucket_num), DIR_EINVAL);
    if (n == 0) return DIR_OK;
    int grid = dir_cdiv(n, DIR_TPB); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(fds_bin_edges_kernel, dim3(grid), dim3(DIR_TPB), sizeof(float) * nedges, dir_s(stream),
                       labels, n, edges, nedges, bucket_start, bucket_num, bins);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// "make up for zero training samples buckets" (sts-b-dir/fds.py:112-125): buckets without samples in this call take
// their neighbours' values, in increasing bucket order (so a run of empty buckets propagates from the left).
__global__ void __launch_bounds__(DIR_TPB)
fds_fill_empty_kernel(const double* __restrict__ count, int nb, int C, float* __restrict__ rm, float* __restrict__ rv) {
    const int c = blockIdx.x * DIR_TPB + threadIdx.x;
    if (c >= C) return;
    for (int b = 0; b < nb; ++b) {
        if (count[b] > 0.0) continue;
        if (b == 0) { if (nb > 1) { rm[c] = rm[C + c]; rv[c] = rv[C + c]; } }
        else if (b == nb - 1) { rm[(size_t)b * C + c] = rm[(size_t)(b - 1) * C + c]; rv[(size_t)b * C + c] = rv[(size_t)(b - 1) * C + c]; }
        else {
            rm[(size_t)b * C + c] = (rm[(size_t)(b - 1) * C + c] + rm[(size_t)(b + 1) * C + c]) / 2.0f;
            rv[(size_t)b * C + c] = (rv[(size_t)(b - 1) * C + c] + rv[(size_t)(b + 1) * C + c]) / 2.0f;
        }
    }
}

extern "C" int dir_fds_fill_empty_buckets(const double* count, int nb, int C, float* running_mean, float* running_var,
                                          dir_stream_t stream) {
    DIR_RETURN_IF(!count || !running_mean || !running_var || nb <= 0 || C <= 0, DIR_EINVAL);
    hipLaunchKernelGGL(fds_fill_empty_kernel, dim3(dir_cdiv(C, DIR_TPB)), dim3(DIR_TPB), 0, dir_s(stream), count, nb, C,
                       running_mean, running_var);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// Multiplier table with the STS-B / NYUD2 guards (sts-b-dir/util.py:63-73, nyud2-dir/util.py:151-162).
//   guard_mode 0: age variant — a column is left untouched when v1 == 0.
//   guard_mode 1: the reference AS IT EXECUTES on torch >= 1.2 — `((v1 > 0.) + (v2 >= 0.)) == 2` adds two bool
//                 tensors (logical or) and compares with 2, which is never true, so as soon as ANY column of the row
//                 has v1 <= 0 or v2 < 0 the WHOLE row is returned unchanged.
//   guard_mode 2: the evident intent (and the behaviour on the torch 0.4.1 the STS-B project pins): only the columns
//                 with v1 <= 0 or v2 < 0 are left untouched.
__global__ void __launch_bounds__(DIR_TPB)
fds_prepare_scale_ex_kernel(const float* __restrict__ v1, const float* __restrict__ v2, int C, float clip_min,
                            float clip_max, int guard_mode, float* __restrict__ scale) {
    __shared__ double wsum[DIR_TPB / DIR_WAVE];
    const int b = blockIdx.x;
    const float* r1 = v1 + (size_t)b * C;
    const float* r2 = v2 + (size_t)b * C;
    double s = 0.0;
    int guard = 0;
    for (int c = threadIdx.x; c < C; c += DIR_TPB) {
        const float a = r1[c], bb = r2[c];
        s += (double)a;
        guard |= (a <= 0.0f) || (bb < 0.0f);                 // exactly the reference's `.any()` tests (NaN compares false)
    }
    s = dir_wave_sum(s);
    if ((threadIdx.x & (DIR_WAVE - 1)) == 0) wsum[threadIdx.x / DIR_WAVE] = s;
    const int any_guard = __syncthreads_or(guard);
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < DIR_TPB / DIR_WAVE; ++i) tot += wsum[i];
    const bool row_identity = (tot < 1e-10) || (guard_mode == 1 && any_guard);
    for (int c = threadIdx.x; c < C; c += DIR_TPB) {
        const float a = r1[c], bb = r2[c];
        const bool untouched = row_identity || (guard_mode == 0 ? (a == 0.0f) : ((a <= 0.0f) || (bb < 0.0f)));
        float out = -1.0f;
        if (!untouched) {
            float q = bb / a;
            q = (q < clip_min) ? clip_min : ((q > clip_max) ? clip_max : q);
            out = sqrtf(q);
        }
        scale[(size_t)b * C + c] = out;
    }
}

extern "C" int dir_fds_prepare_scale_ex(const float* v1, const float* v2, int nb, int C, float clip_min, float clip_max,
                                        int guard_mode, float* scale, dir_stream_t stream) {
    DIR_RETURN_IF(!v1 || !v2 || !scale || nb <= 0 || C <= 0 || guard_mode < 0 || guard_mode > 2, DIR_EINVAL);
    hipLaunchKernelGGL(fds_prepare_scale_ex_kernel, dim3(nb), dim3(DIR_TPB), 0, dir_s(stream), v1, v2, C, clip_min, clip_max,
                       guard_mode, scale);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}


// =============================================================================================
// NYUD2-DIR dense variant (nyud2-dir/models/fds.py:51-53, :138-139): bucket = clamp(int(label * 10), start, num - 1)
// =============================================================================================
__global__ void __launch_bounds__(DIR_TPB)
fds_bin_scaled_kernel(const float* __restrict__ labels, long long n, float mult, int bucket_start, int bucket_num,
                      int32_t* __restrict__ bins) {
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < n; i += (long long)gridDim.x * DIR_TPB) {
        const float l = labels[i];
        int b = -1;
        if (l == l) {                                       // int(NaN) raises in the reference
            const float v = l * mult;                       // float32 product, then truncation (fds.py:52)
            int k = (v >= 2147483520.0f) ? 2147483647 : ((v <= -2147483520.0f) ? -2147483647 : (int)v);
            if (k > bucket_num - 1) k = bucket_num - 1;
            if (k < bucket_start) k = bucket_start;
            b = k - bucket_start;
        }
        bins[i] = b;
    }
}

extern "C" int dir_fds_bin_scaled(const float* labels, long long n, float mult, int bucket_start, int bucket_num,
                                  int32_t* bins, dir_stream_t stream) {
    DIR_RETURN_IF(!labels || !bins || n < 0, DIR_EINVAL);
    DIR_RETURN_IF(check_buckets(bucket_start, bucket_num), DIR_EINVAL);
    if (n == 0) return DIR_OK;
    int grid = dir_cdiv(n, DIR_TPB); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fds_bin_scaled_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), labels, n, mult, bucket_start,
                       bucket_num, bins);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
