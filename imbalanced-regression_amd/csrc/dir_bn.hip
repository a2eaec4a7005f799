// Fused BatchNorm (+ residual add) (+ ReLU) for NHWC activations on MI355X / gfx950 — forward and backward.
//
// Replaces, per BN layer of resnet.py:41-70 / :79-80 / :112-118, the chain the reference's eager modules launch:
//   forward : batch mean/var -> normalise (read+write) -> [+= residual (2 reads + write)] -> ReLU (read+write)
//   backward: ReLU-bwd (2 reads + write) -> dscale/dbias (2 reads) -> dx (2-3 reads + write)
// (measured round 1: 58 % of the MI355X step time) with
//   forward : 1 read (statistics) + 1 read [+1 residual read] + 1 write
//   backward: 3 reads (reduce) + 3 reads + 1 write [+1 write for the residual branch]
// All of it is HBM-bound streaming: activations are [M = N*H*W rows][C channels] with C contiguous (NHWC), so a
// thread owns 8 (bf16) / 4 (f32) consecutive channels = one 16-byte load per row and walks rows with a stride;
// per-channel constants live in registers; partial sums go through LDS once per workgroup and through a small
// [row-blocks][C] float buffer once per launch (no atomics -> bit-reproducible), combined in float64.
#include <cstdlib>
#include "dir_common.h"

namespace {

constexpr int BN_CAP_TOTAL = 768;   // workgroups per streaming launch = 3 per CU (256..4096 swept in round 1, 512..2048 again in round 3)

struct BnGeom {
    int ct;             // channel tile handled by one workgroup column (<= 256 for bf16, <= 128 for f32)
    int tpr;            // threads per row inside the tile = ct / VEC
    int rpi;            // rows per iteration = 256 / tpr
    int ctiles;         // C / ct
    int rblocks;        // row blocks (grid.x)
    int chunk;          // apply passes only: 0 = persistent sweep (grid.x = rblocks, stride rblocks * rpi); > 0 = rows per workgroup,
                        // consecutive (grid.x = ceil(M / chunk)): short-lived workgroups in address order (BN_APPLY_CHUNK)
};
constexpr int BN_APPLY_CHUNK = 0;    // compile-time measurement knob (round 3: 0.9998 at 16 KB chunks, worse at 8 / 32 KB): 0 = persistent sweep

struct BnWalk { int64_t row, stride, end; };
__device__ __forceinline__ BnWalk bn_walk(const BnGeom& g, int64_t M, int tr) {
    BnWalk w;
    if (g.chunk) {
        const int64_t r0 = (int64_t)blockIdx.x * g.chunk;
        w.row = r0 + tr; w.stride = g.rpi; w.end = (r0 + g.chunk < M) ? r0 + g.chunk : M;
    } else {
        w.row = (int64_t)blockIdx.x * g.rpi + tr; w.stride = (int64_t)gridDim.x * g.rpi; w.end = M;
    }
    return w;
}

template <int VEC>
BnGeom bn_geom(int64_t M, int C) {
    BnGeom g;
    const int max_ct = 32 * VEC;                        // 32 lanes x 16 B = 512 B contiguous per row segment
    g.ct = C < max_ct ? C : max_ct;
    g.tpr = g.ct / VEC;
    g.rpi = DIR_TPB / g.tpr;
    g.ctiles = C / g.ct;
    int64_t want = (M + (int64_t)g.rpi * 4 - 1) / ((int64_t)g.rpi * 4);   // >= 4 row iterations per workgroup
    const int cap_total = BN_CAP_TOTAL;
    int64_t cap = cap_total / g.ctiles; if (cap < 1) cap = 1;   // <= 768/ctiles partial rows per channel
    g.rblocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    g.chunk = 0;
    return g;
}
// geometry of an apply pass: the sweep of bn_geom, or (BN_APPLY_CHUNK = iters > 0) `iters` consecutive row groups per workgroup
inline BnGeom bn_apply_geom(BnGeom g, int64_t M) {
    if (BN_APPLY_CHUNK > 0) {
        g.chunk = BN_APPLY_CHUNK * g.rpi;
        g.rblocks = (int)((M + g.chunk - 1) / g.chunk);
    }
    return g;
}

// ---- 16-byte vector load/store of VEC elements as float ------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
    static __device__ __forceinline__ uint32_t store_bits(float* p, const float (&o)[4]) { store(p, o); return 0u; }
    // non-temporal load: the last read of a stream much larger than the L2 — its lines are not worth keeping there
    static __device__ __forceinline__ void load_nt(const float* p, float (&o)[4]) {
        typedef __attribute__((ext_vector_type(4))) float f32x4_t;
        const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p)); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
struct bf16_t { uint16_t v; };
__device__ __forceinline__ uint32_t f2bf(float f) {           // round to nearest even, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void load_nt(const bf16_t* p, float (&o)[8]) {
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                               // one v_cvt_pk_bf16_f32 (RNE, quiet NaN) per pair on gfx950
            typedef __attribute__((ext_vector_type(2))) float f32x2_t;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            const f32x2_t v = {o[2 * i], o[2 * i + 1]};
            w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
        }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // store + the ReLU mask byte of the STORED values: bit j = (halfword j, as a signed number, > 0) = "stored value > 0" for every
    // non-NaN value; read off the packed words, no second rounding of the eight elements
    static __device__ __forceinline__ uint32_t store_bits(bf16_t* p, const float (&o)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            typedef __attribute__((ext_vector_type(2))) float f32x2_t;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            const f32x2_t v = {o[2 * i], o[2 * i + 1]};
            w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
        }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        uint32_t b = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) b |= (((int32_t)(w[i] << 16) > 0 ? 1u : 0u) | ((int32_t)(w[i] & 0xffff0000u) > 0 ? 2u : 0u)) << (2 * i);
        return b;
    }
};

// Workgroup reduction of per-thread [VEC] partials over the rpi row-lanes that share a channel group, then one
// store per channel into partial[rblock][which][C].
template <int VEC, int NACC>
__device__ __forceinline__ void block_reduce_store(float (&acc)[NACC][VEC], int tpr, int rpi, int C, int c0,
                                                   float* __restrict__ partial, int nacc_stride_c) {
    __shared__ float sh[NACC][DIR_TPB * VEC];
    const int t = threadIdx.x;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int j = 0; j < VEC; ++j) sh[a][t * VEC + j] = acc[a][j];
    __syncthreads();
    if (t < tpr) {                                       // row-lane 0 of each channel group sums the others
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float s[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] = 0.0f;
            for (int r = 0; r < rpi; ++r)
#pragma unroll
                for (int j = 0; j < VEC; ++j) s[j] += sh[a][(r * tpr + t) * VEC + j];
            float* o = partial + ((size_t)blockIdx.x * NACC + a) * nacc_stride_c + c0 + t * VEC;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = s[j];
        }
    }
}

// ---- forward: statistics -------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(DIR_TPB)
bn_stats_partial_kernel(const T* __restrict__ x, int64_t M, int C, BnGeom g, float* __restrict__ partial) {
    constexpr int VEC = Vec<T>::N;
    const int t = threadIdx.x, tg = t % g.tpr, tr = t / g.tpr;
    const int c0 = blockIdx.y * g.ct;
    const T* base = x + c0 + tg * VEC;
    float acc[2][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc[0][j] = 0.0f; acc[1][j] = 0.0f; }
    const int64_t stride = (int64_t)gridDim.x * g.rpi;
    int64_t row = (int64_t)blockIdx.x * g.rpi + tr;
    for (; row + 3 * stride < M; row += 4 * stride) {     // 4 independent 16-B loads in flight per lane
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) Vec<T>::load(base + (row + u * stride) * C, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) { acc[0][j] += v[u][j]; acc[1][j] += v[u][j] * v[u][j]; }
    }
    for (; row < M; row += stride) {
        float v[VEC]; Vec<T>::load(base + row * C, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { acc[0][j] += v[j]; acc[1][j] += v[j] * v[j]; }
    }
    block_reduce_store<VEC, 2>(acc, g.tpr, g.rpi, C, c0, partial, C);
}


// Sum partial[b][which][c] over b for FC channels per workgroup: 256 threads = FC channels x (256/FC) slices of the
// row-block axis, float64, fixed-order LDS combine. FC = 8 for short partial lists; FC = 2 (128 slices) when the
// conv epilogue produced thousands of rows (one per 128-row tile) — the loop is a chain of L2 round trips, so its
// length, not the bytes, is what costs. Returns the two sums for channel blockIdx.x*FC + t in threads t < FC.
template <int FC, typename PT>
__device__ __forceinline__ bool column_sums(const PT* __restrict__ partial, int rblocks, int C, double& s0, double& s1) {
    constexpr int SL = DIR_TPB / FC;
    __shared__ double sh[2][DIR_TPB];
    const int t = threadIdx.x, ch = t % FC, sl = t / FC;
    const int c = blockIdx.x * FC + ch;
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 8
        for (int r = sl; r < rblocks; r += SL) {
            a += (double)partial[((size_t)r * 2 + 0) * C + c];
            b += (double)partial[((size_t)r * 2 + 1) * C + c];
        }
    }
    sh[0][t] = a; sh[1][t] = b;
    __syncthreads();
    if (t >= FC || c >= C) return false;
    a = 0.0; b = 0.0;
    for (int k = 0; k < SL; ++k) { a += sh[0][k * FC + t]; b += sh[1][k * FC + t]; }
    s0 = a; s1 = b;
    return true;
}

// coef layout in the workspace: [0][C] = a (scale), [1][C] = b (shift)
// Long partial lists (the convolution epilogue writes one row per 128 output pixels: 6272 rows for 256 x 56 x 56)
// are first folded to <= 32 rows of doubles by many workgroups; a lone C/8-workgroup pass over them is latency bound.
template <int FC>
__global__ void __launch_bounds__(DIR_TPB)
bn_fold_partials_kernel(const float* __restrict__ partial, int rows, int C, int rows_per_split, double* __restrict__ folded) {
    constexpr int SL = DIR_TPB / FC;
    __shared__ double sh[2][DIR_TPB];
    const int t = threadIdx.x, ch = t % FC, sl = t / FC;
    const int c = blockIdx.x * FC + ch;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int r = r0 + sl; r < r1; r += SL) {
            a += (double)partial[((size_t)r * 2 + 0) * C + c];
            b += (double)partial[((size_t)r * 2 + 1) * C + c];
        }
    }
    sh[0][t] = a; sh[1][t] = b;
    __syncthreads();
    if (t >= FC || c >= C) return;
    a = 0.0; b = 0.0;
    for (int k = 0; k < SL; ++k) { a += sh[0][k * FC + t]; b += sh[1][k * FC + t]; }
    folded[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
    folded[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
}

template <int FC, typename PT>
__global__ void __launch_bounds__(DIR_TPB)
bn_finalize_train_kernel(const PT* __restrict__ partial, int rblocks, int64_t M, int C,
                         const float* __restrict__ gamma, const float* __restrict__ beta,
                         float* __restrict__ running_mean, float* __restrict__ running_var,
                         double momentum, double eps, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                         float* __restrict__ coef) {
    double s, q;
    if (!column_sums<FC, PT>(partial, rblocks, C, s, q)) return;
    const int c = blockIdx.x * FC + threadIdx.x;
    const double n = (double)M;
    const double mean = s / n;
    double var = q / n - mean * mean;                     // biased (normalisation)
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    save_mean[c] = (float)mean;
    save_rstd[c] = (float)rstd;
    const double meanf = (double)(float)mean, rstdf = (double)(float)rstd;   // what the backward will see
    if (running_mean) {                                    // torch: running = (1-m)*running + m*batch, unbiased var
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
    coef[c] = (float)((double)gamma[c] * rstdf);
    coef[C + c] = (float)((double)beta[c] - meanf * (double)gamma[c] * rstdf);
}

__global__ void __launch_bounds__(DIR_TPB)
bn_finalize_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                        const float* __restrict__ running_mean, const float* __restrict__ running_var,
                        double eps, float* __restrict__ coef) {
    const int c = blockIdx.x * DIR_TPB + threadIdx.x;
    if (c >= C) return;
    const double rstd = 1.0 / sqrt((double)running_var[c] + eps);
    coef[c] = (float)((double)gamma[c] * rstd);
    coef[C + c] = (float)((double)beta[c] - (double)running_mean[c] * (double)gamma[c] * rstd);
}

// One byte per 8-channel group of a ReLU output row: bit j = (the STORED value of channel j) > 0 — the mask of that ReLU's
// backward at 1/16 of the bytes of the tensor itself (bf16 only; a positive float that rounds to bf16 zero counts as zero):
// Vec<bf16_t>::store_bits above.

// ---- forward: y = [relu]( x * a + b [+ residual] ) ---------------------------------------------------------
// RES: 0 = no residual, 1 = residual tensor added as is, 2 = residual is itself the INPUT of a BatchNorm whose
// normalisation (rcoef) is applied on the fly (projection shortcut: relu(bn3(x) + bn_d(r)), resnet.py:63-68, without
// materialising bn_d(r))
// The FINALIZE step inside the apply pass (round 3): the per-channel statistics -> coefficients arithmetic of
// bn_finalize_train_kernel used to be a launch of its own between the fold of the partial list and the apply pass — C/8
// workgroups chasing a chain of L2 round trips, ~5.5 us of an otherwise idle GPU per BatchNorm and pass (106 such launches
// per training step). Now every workgroup of the apply pass sums the <= BN_FOLD_ROWS folded rows (float64) of ITS channel tile
// itself — the same arithmetic, in the same order — and the workgroups of row block 0 also write save_mean / save_rstd and the
// running statistics. fin.folded == nullptr: coefficients come from `coef` as before (eval mode, the two-BatchNorm join).
constexpr int BN_FOLD_ROWS = 8;
struct BnFinF {
    const double* folded; int rows; double n, momentum, eps;
    const float* gamma; const float* beta; float* running_mean; float* running_var; float* save_mean; float* save_rstd;
};

template <typename T, int RES, bool RELU>
__global__ void __launch_bounds__(DIR_TPB)
bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, int64_t M, int C, BnGeom g,
                const float* __restrict__ coef, const float* __restrict__ rcoef, uint8_t* __restrict__ bits, BnFinF fin) {
    constexpr int VEC = Vec<T>::N;
    const int t = threadIdx.x, tg = t % g.tpr, tr = t / g.tpr;
    const int c = blockIdx.y * g.ct + tg * VEC;
    float a[VEC], b[VEC], a2[VEC], b2[VEC];
    if (fin.folded) {
        __shared__ float s_ab[2][DIR_TPB];
        if (t < g.ct) {
            const int cc = blockIdx.y * g.ct + t;
            double s = 0.0, q = 0.0;
            for (int r = 0; r < fin.rows; ++r) { s += fin.folded[((size_t)r * 2 + 0) * C + cc]; q += fin.folded[((size_t)r * 2 + 1) * C + cc]; }
            const double mean = s / fin.n;
            double var = q / fin.n - mean * mean;             // biased (normalisation)
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + fin.eps);
            const double meanf = (double)(float)mean, rstdf = (double)(float)rstd;   // what the backward will see
            s_ab[0][t] = (float)((double)fin.gamma[cc] * rstdf);
            s_ab[1][t] = (float)((double)fin.beta[cc] - meanf * (double)fin.gamma[cc] * rstdf);
            if (blockIdx.x == 0) {
                fin.save_mean[cc] = (float)mean;
                fin.save_rstd[cc] = (float)rstd;
                if (fin.running_mean) {                        // torch: running = (1-m)*running + m*batch, unbiased var
                    const double unbiased = fin.n > 1.0 ? var * fin.n / (fin.n - 1.0) : var;
                    fin.running_mean[cc] = (float)((1.0 - fin.momentum) * (double)fin.running_mean[cc] + fin.momentum * mean);
                    fin.running_var[cc] = (float)((1.0 - fin.momentum) * (double)fin.running_var[cc] + fin.momentum * unbiased);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < VEC; ++j) { a[j] = s_ab[0][tg * VEC + j]; b[j] = s_ab[1][tg * VEC + j]; }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { a[j] = coef[c + j]; b[j] = coef[C + c + j]; }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        a2[j] = RES == 2 ? rcoef[c + j] : 1.0f; b2[j] = RES == 2 ? rcoef[C + c + j] : 0.0f;
    }
    const BnWalk wk = bn_walk(g, M, tr);
    const int64_t stride = wk.stride, Mend = wk.end;
    // Mirrored row order (physical row = M-1-row): the statistics pass swept the tensor front to back, so its
    // tail is what L2 / the 256 MiB Infinity Cache still hold — re-read that first. x is read for the last time in the forward
    // pass: non-temporal (A/B -0.15 ms per train step, -0.20 ms per epoch-tail forward); the result is stored normally — the next
    // convolution reads it and finds part of it in cache (non-temporal stores here measured +0.12 ms).
    int64_t row = wk.row;
    for (; row + stride < Mend; row += 2 * stride) {
        float v[2][VEC], r[2][VEC];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Vec<T>::load_nt(x + (M - 1 - (row + u * stride)) * C + c, v[u]);
            if (RES) Vec<T>::load(res + (M - 1 - (row + u * stride)) * C + c, r[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float o = v[u][j] * a[j] + b[j];
                if (RES == 1) o += r[u][j];
                if (RES == 2) o += r[u][j] * a2[j] + b2[j];
                if (RELU) o = o > 0.0f ? o : 0.0f;
                v[u][j] = o;
            }
            const uint32_t mb = Vec<T>::store_bits(y + (M - 1 - (row + u * stride)) * C + c, v[u]);
            if (RELU && bits) bits[((M - 1 - (row + u * stride)) * C + c) / 8] = (uint8_t)mb;

        }
    }
    for (; row < Mend; row += stride) {
        float v[VEC], r[VEC];
        Vec<T>::load_nt(x + (M - 1 - row) * C + c, v);
        if (RES) Vec<T>::load(res + (M - 1 - row) * C + c, r);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float o = v[j] * a[j] + b[j];
            if (RES == 1) o += r[j];
            if (RES == 2) o += r[j] * a2[j] + b2[j];
            if (RELU) o = o > 0.0f ? o : 0.0f;
            v[j] = o;
        }
        const uint32_t mb = Vec<T>::store_bits(y + (M - 1 - row) * C + c, v);
        if (RELU && bits) bits[((M - 1 - row) * C + c) / 8] = (uint8_t)mb;

    }
}

// ---- backward: g = dout * [out > 0];  partial sums of g and g * x ------------------------------------------
// ReLU mask source: 0 = no ReLU, 1 = saved output (needed when a residual was added), 2 = recomputed from x with
// the forward's own coefficients (x*a+b > 0; bit-identical to the forward's decision, one tensor read less).
struct BnMaskCoef { const float* gamma; const float* beta; const float* mean; const float* rstd; };
template <int VEC>
__device__ __forceinline__ void bn_mask_coef(const BnMaskCoef& mc, int c, float (&af)[VEC], float (&bf)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const double gm = (double)mc.gamma[c + j], rs = (double)mc.rstd[c + j];
        af[j] = (float)(gm * rs);
        bf[j] = (float)((double)mc.beta[c + j] - (double)mc.mean[c + j] * gm * rs);
    }
}

template <typename T, int MASK>
__global__ void __launch_bounds__(DIR_TPB)
bn_bwd_partial_kernel(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ out,
                      int64_t M, int C, BnGeom g, float* __restrict__ partial, BnMaskCoef mc) {
    constexpr int VEC = Vec<T>::N;
    constexpr bool RELU = (MASK == 1);
    const int t = threadIdx.x, tg = t % g.tpr, tr = t / g.tpr;
    const int c0 = blockIdx.y * g.ct, c = c0 + tg * VEC;
    float af[VEC], bf[VEC];
    if (MASK == 2) bn_mask_coef<VEC>(mc, c, af, bf);
    float acc[2][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc[0][j] = 0.0f; acc[1][j] = 0.0f; }
    const int64_t stride = (int64_t)gridDim.x * g.rpi;
    int64_t row = (int64_t)blockIdx.x * g.rpi + tr;
    for (; row + stride < M; row += 2 * stride) {
        float d[2][VEC], v[2][VEC], o[2][VEC];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Vec<T>::load(dout + (row + u * stride) * C + c, d[u]);
            Vec<T>::load(x + (row + u * stride) * C + c, v[u]);
            if (RELU) Vec<T>::load(out + (row + u * stride) * C + c, o[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float gj = (RELU && !(o[u][j] > 0.0f)) ? 0.0f : d[u][j];
                if (MASK == 2 && !(v[u][j] * af[j] + bf[j] > 0.0f)) gj = 0.0f;
                acc[0][j] += gj; acc[1][j] += gj * v[u][j];
            }
    }
    for (; row < M; row += stride) {
        float d[VEC], v[VEC], o[VEC];
        Vec<T>::load(dout + row * C + c, d);
        Vec<T>::load(x + row * C + c, v);
        if (RELU) Vec<T>::load(out + row * C + c, o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float gj = (RELU && !(o[j] > 0.0f)) ? 0.0f : d[j];
            if (MASK == 2 && !(v[j] * af[j] + bf[j] > 0.0f)) gj = 0.0f;
            acc[0][j] += gj; acc[1][j] += gj * v[j];
        }
    }
    block_reduce_store<VEC, 2>(acc, g.tpr, g.rpi, C, c0, partial, C);
}

// dbeta = sum g;  dgamma = rstd * (sum g*x - mean * sum g);  dx = a*g + p*x + q with
// a = gamma*rstd, p = -a*rstd*dgamma/M, q = -a*dbeta/M - p*mean.   coef: [0]=a [1]=p [2]=q
template <int FC, typename PT>
__global__ void __launch_bounds__(DIR_TPB)
bn_bwd_finalize_kernel(const PT* __restrict__ partial, int rblocks, int64_t M, int C,
                       const float* __restrict__ gamma, const float* __restrict__ save_mean,
                       const float* __restrict__ save_rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                       float* __restrict__ coef) {
    double sg, sgx;
    if (!column_sums<FC, PT>(partial, rblocks, C, sg, sgx)) return;
    const int c = blockIdx.x * FC + threadIdx.x;
    const double mean = (double)save_mean[c], rstd = (double)save_rstd[c], n = (double)M;
    const double dg = rstd * (sgx - mean * sg);
    dbeta[c] = (float)sg;
    dgamma[c] = (float)dg;
    const double a = (double)gamma[c] * rstd;
    const double p = -a * rstd * dg / n;
    coef[c] = (float)a;
    coef[C + c] = (float)p;
    coef[2 * C + c] = (float)(-a * sg / n - p * mean);
}

// (finalize inside the apply pass, as in the forward: bn_bwd_finalize_kernel's arithmetic per channel tile; row block 0 writes
// dgamma / dbeta)
struct BnFinB {
    const double* folded; int rows; double n;
    const float* gamma; const float* save_mean; const float* save_rstd; float* dgamma; float* dbeta;
};

template <typename T, int MASK, bool DRES>
__global__ void __launch_bounds__(DIR_TPB)
bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ out,
                    T* __restrict__ dx, T* __restrict__ dres, int64_t M, int C, BnGeom g, const float* __restrict__ coef,
                    BnMaskCoef mc, BnFinB fin) {
    constexpr int VEC = Vec<T>::N;
    constexpr bool RELU = (MASK == 1);
    const int t = threadIdx.x, tg = t % g.tpr, tr = t / g.tpr;
    const int c = blockIdx.y * g.ct + tg * VEC;
    float af[VEC], bf[VEC];
    if (MASK == 2) bn_mask_coef<VEC>(mc, c, af, bf);
    float a[VEC], p[VEC], q[VEC];
    if (fin.folded) {
        __shared__ float s_apq[3][DIR_TPB];
        if (t < g.ct) {
            const int cc = blockIdx.y * g.ct + t;
            double sg = 0.0, sgx = 0.0;
            for (int r = 0; r < fin.rows; ++r) { sg += fin.folded[((size_t)r * 2 + 0) * C + cc]; sgx += fin.folded[((size_t)r * 2 + 1) * C + cc]; }
            const double mean = (double)fin.save_mean[cc], rstd = (double)fin.save_rstd[cc];
            const double dg = rstd * (sgx - mean * sg);
            const double ca = (double)fin.gamma[cc] * rstd;
            const double cp = -ca * rstd * dg / fin.n;
            s_apq[0][t] = (float)ca;
            s_apq[1][t] = (float)cp;
            s_apq[2][t] = (float)(-ca * sg / fin.n - cp * mean);
            if (blockIdx.x == 0) { fin.dbeta[cc] = (float)sg; fin.dgamma[cc] = (float)dg; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < VEC; ++j) { a[j] = s_apq[0][tg * VEC + j]; p[j] = s_apq[1][tg * VEC + j]; q[j] = s_apq[2][tg * VEC + j]; }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { a[j] = coef[c + j]; p[j] = coef[C + c + j]; q[j] = coef[2 * C + c + j]; }
    }
    const BnWalk wk = bn_walk(g, M, tr);
    const int64_t stride = wk.stride, Mend = wk.end;
    int64_t row = wk.row;
    for (; row + stride < Mend; row += 2 * stride) {          // mirrored rows (see bn_apply_kernel), 2 rows in flight
        float d[2][VEC], v[2][VEC], o[2][VEC];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t pr = M - 1 - (row + u * stride);
            Vec<T>::load_nt(dout + pr * C + c, d[u]);
            Vec<T>::load(x + pr * C + c, v[u]);
            if (RELU) Vec<T>::load(out + pr * C + c, o[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t pr = M - 1 - (row + u * stride);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float gj = (RELU && !(o[u][j] > 0.0f)) ? 0.0f : d[u][j];
                if (MASK == 2 && !(v[u][j] * af[j] + bf[j] > 0.0f)) gj = 0.0f;
                d[u][j] = gj;
                v[u][j] = a[j] * gj + (p[j] * v[u][j] + q[j]);
            }
            Vec<T>::store(dx + pr * C + c, v[u]);
            if (DRES) Vec<T>::store(dres + pr * C + c, d[u]);
        }
    }
    for (; row < Mend; row += stride) {
        const int64_t pr = M - 1 - row;
        float d[VEC], v[VEC], o[VEC];
        Vec<T>::load_nt(dout + pr * C + c, d);
        Vec<T>::load(x + pr * C + c, v);
        if (RELU) Vec<T>::load(out + pr * C + c, o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float gj = (RELU && !(o[j] > 0.0f)) ? 0.0f : d[j];
            if (MASK == 2 && !(v[j] * af[j] + bf[j] > 0.0f)) gj = 0.0f;
            d[j] = gj;
            v[j] = a[j] * gj + (p[j] * v[j] + q[j]);
        }
        Vec<T>::store(dx + pr * C + c, v);
        if (DRES) Vec<T>::store(dres + pr * C + c, d);
    }
}

// ---- backward of the projection-shortcut join relu(bn(x) + bn_r(r)) on its (already masked) gradient g: both BatchNorms
// share g, so ONE reduction pass forms (sum g, sum g x) and (sum g, sum g r) — two ordinary partial lists — and ONE apply pass
// writes dx and dr: g is read twice instead of four times (8 tensor passes instead of 10). Same row order and accumulation order as
// bn_bwd_partial_kernel<T, 0> / bn_bwd_apply_kernel<T, 0, false>: the results are bit-identical to two dir_bn_bwd calls.
template <typename T>
__global__ void __launch_bounds__(DIR_TPB)
bn_bwd_join_partial_kernel(const T* __restrict__ gout, const T* __restrict__ x, const T* __restrict__ r, int64_t M, int C, BnGeom g,
                           float* __restrict__ partial_x, float* __restrict__ partial_r) {
    constexpr int VEC = Vec<T>::N;
    const int t = threadIdx.x, tg = t % g.tpr, tr = t / g.tpr;
    const int c0 = blockIdx.y * g.ct, c = c0 + tg * VEC;
    float ax[2][VEC], ar[2][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { ax[0][j] = 0.0f; ax[1][j] = 0.0f; ar[0][j] = 0.0f; ar[1][j] = 0.0f; }
    const int64_t stride = (int64_t)gridDim.x * g.rpi;
    int64_t row = (int64_t)blockIdx.x * g.rpi + tr;
    for (; row + stride < M; row += 2 * stride) {
        float d[2][VEC], v[2][VEC], w[2][VEC];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Vec<T>::load(gout + (row + u * stride) * C + c, d[u]);
            Vec<T>::load(x + (row + u * stride) * C + c, v[u]);
            Vec<T>::load(r + (row + u * stride) * C + c, w[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) { ax[0][j] += d[u][j]; ax[1][j] += d[u][j] * v[u][j]; ar[1][j] += d[u][j] * w[u][j]; }
    }
    for (; row < M; row += stride) {
        float d[VEC], v[VEC], w[VEC];
        Vec<T>::load(gout + row * C + c, d);
        Vec<T>::load(x + row * C + c, v);
        Vec<T>::load(r + row * C + c, w);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { ax[0][j] += d[j]; ax[1][j] += d[j] * v[j]; ar[1][j] += d[j] * w[j]; }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) ar[0][j] = ax[0][j];
    block_reduce_store<VEC, 2>(ax, g.tpr, g.rpi, C, c0, partial_x, C);
    __syncthreads();                                     // (the helper's LDS buffer is reused)
    block_reduce_store<VEC, 2>(ar, g.tpr, g.rpi, C, c0, partial_r, C);
}

template <typename T>
__global__ void __launch_bounds__(DIR_TPB)
bn_bwd_join_apply_kernel(const T* __restrict__ gout, const T* __restrict__ x, const T* __restrict__ r, T* __restrict__ dx,
                         T* __restrict__ dr, int64_t M, int C, BnGeom g, const float* __restrict__ coef_x, const float* __restrict__ coef_r) {
    constexpr int VEC = Vec<T>::N;
    const int t = threadIdx.x, tg = t % g.tpr, tr = t / g.tpr;
    const int c = blockIdx.y * g.ct + tg * VEC;
    float a[VEC], p[VEC], q[VEC], a2[VEC], p2[VEC], q2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        a[j] = coef_x[c + j]; p[j] = coef_x[C + c + j]; q[j] = coef_x[2 * C + c + j];
        a2[j] = coef_r[c + j]; p2[j] = coef_r[C + c + j]; q2[j] = coef_r[2 * C + c + j];
    }
    const BnWalk wk = bn_walk(g, M, tr);
    const int64_t stride = wk.stride, Mend = wk.end;
    int64_t row = wk.row;
    for (; row + stride < Mend; row += 2 * stride) {          // mirrored rows (see bn_apply_kernel), 2 rows in flight
        float d[2][VEC], v[2][VEC], w[2][VEC];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t pr = M - 1 - (row + u * stride);
            Vec<T>::load_nt(gout + pr * C + c, d[u]);
            Vec<T>::load(x + pr * C + c, v[u]);
            Vec<T>::load(r + pr * C + c, w[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t pr = M - 1 - (row + u * stride);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                v[u][j] = a[j] * d[u][j] + (p[j] * v[u][j] + q[j]);
                w[u][j] = a2[j] * d[u][j] + (p2[j] * w[u][j] + q2[j]);
            }
            Vec<T>::store(dx + pr * C + c, v[u]);
            Vec<T>::store(dr + pr * C + c, w[u]);
        }
    }
    for (; row < Mend; row += stride) {
        const int64_t pr = M - 1 - row;
        float d[VEC], v[VEC], w[VEC];
        Vec<T>::load_nt(gout + pr * C + c, d);
        Vec<T>::load(x + pr * C + c, v);
        Vec<T>::load(r + pr * C + c, w);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            v[j] = a[j] * d[j] + (p[j] * v[j] + q[j]);
            w[j] = a2[j] * d[j] + (p2[j] * w[j] + q2[j]);
        }
        Vec<T>::store(dx + pr * C + c, v);
        Vec<T>::store(dr + pr * C + c, w);
    }
}

// Compile-time measurement knob (round 3: measured neutral, profiles/r03_ab_in_process.txt): true = the training-mode forward and the
// backward finalize inside their apply pass (fold + apply), false = fold (long lists) + finalize launch + apply.
constexpr bool BN_FUSED_FINALIZE = false;

struct BnWs { float* partial; float* coef; double* folded; size_t bytes; };
template <int VEC> BnWs bn_ws(void* base, int64_t M, int C) {
    BnGeom g = bn_geom<VEC>(M, C);
    BnWs w;
    const size_t pbytes = dir_align_up(sizeof(float) * (size_t)g.rblocks * 2 * C, 256);
    const size_t cbytes = dir_align_up(sizeof(float) * 3 * (size_t)C, 256);
    w.partial = reinterpret_cast<float*>(base);
    w.coef = reinterpret_cast<float*>(static_cast<char*>(base) + pbytes);
    w.folded = reinterpret_cast<double*>(static_cast<char*>(base) + pbytes + cbytes);     // [BN_FOLD_ROWS][2][C] float64
    w.bytes = pbytes + cbytes + dir_align_up(sizeof(double) * BN_FOLD_ROWS * 2 * (size_t)C, 256);
    return w;
}

// partial list [rows][2][C] (float) -> <= BN_FOLD_ROWS rows of float64 in `folded`; returns the folded row count
int bn_fold(const float* part, int rows, int C, double* folded, hipStream_t s) {
    int splits = rows / 48;                                          // >= 48 rows per fold workgroup
    if (splits > BN_FOLD_ROWS) splits = BN_FOLD_ROWS;
    if (splits < 1) splits = 1;
    const int rps = dir_cdiv(rows, splits);
    splits = dir_cdiv(rows, rps);
    hipLaunchKernelGGL(bn_fold_partials_kernel<8>, dim3(dir_cdiv(C, 8), splits), dim3(DIR_TPB), 0, s, part, rows, C, rps, folded);
    return splits;
}

bool bn_shape_ok(int dtype, int64_t M, int C) {
    const int vec = dtype == DIR_BF16 ? 8 : 4;
    if (M <= 0 || C <= 0 || C % vec) return false;
    const int max_ct = 32 * vec;
    return C <= max_ct ? (DIR_TPB % (C / vec) == 0) : (C % max_ct == 0);
}

// statistics (own pass, or the partial list of the producing convolution's epilogue) -> save_mean / save_rstd,
// running statistics and the apply coefficients coef[2][C]
template <typename T>
int prepare_impl(const void* x_, int64_t M, int C, const float* gamma, const float* beta, float* running_mean,
                 float* running_var, double momentum, double eps, float* save_mean, float* save_rstd, float* coef,
                 void* ws, size_t ws_bytes, hipStream_t s, const float* ext_partial, int ext_rows) {
    constexpr int VEC = Vec<T>::N;
    const T* x = static_cast<const T*>(x_);
    BnGeom g = bn_geom<VEC>(M, C);
    BnWs w = bn_ws<VEC>(ws, M, C);
    DIR_RETURN_IF(ws_bytes < w.bytes, DIR_EWORKSPACE);
    if (!coef) coef = w.coef;
    const float* part = w.partial;
    int prow = g.rblocks;
    if (ext_partial) { part = ext_partial; prow = ext_rows; }       // statistics came out of the conv epilogue
    else {
        hipLaunchKernelGGL(bn_stats_partial_kernel<T>, dim3(g.rblocks, g.ctiles), dim3(DIR_TPB), 0, s, x, M, C, g, w.partial);
        DIR_LAUNCH_CHECK();
    }
    int splits = prow / 64;                                        // >= 64 rows per fold workgroup
    if (splits > 32) splits = 32;
    if (splits > g.rblocks / 2) splits = g.rblocks / 2;            // folded doubles live in the (unused) partial area
    if (ext_partial && splits >= 16) {      // (lists under 1024 rows: one finalize launch reads them directly, 32 rows per trip)
        const int rps = dir_cdiv(prow, splits);
        splits = dir_cdiv(prow, rps);
        double* folded = reinterpret_cast<double*>(w.partial);
        hipLaunchKernelGGL(bn_fold_partials_kernel<8>, dim3(dir_cdiv(C, 8), splits), dim3(DIR_TPB), 0, s, part, prow, C, rps, folded);
        DIR_LAUNCH_CHECK();
        hipLaunchKernelGGL((bn_finalize_train_kernel<8, double>), dim3(dir_cdiv(C, 8)), dim3(DIR_TPB), 0, s, folded, splits, M, C,
                           gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, coef);
    } else {
        hipLaunchKernelGGL((bn_finalize_train_kernel<8, float>), dim3(dir_cdiv(C, 8)), dim3(DIR_TPB), 0, s, part, prow, M, C,
                           gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, coef);
    }
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

template <typename T>
int apply_impl(const void* x_, const void* res_, const float* rcoef, void* y_, int64_t M, int C, const float* coef, int relu,
               hipStream_t s, uint8_t* bits = nullptr, BnFinF fin = BnFinF{}) {
    constexpr int VEC = Vec<T>::N;
    const T* x = static_cast<const T*>(x_); const T* res = static_cast<const T*>(res_); T* y = static_cast<T*>(y_);
    const BnGeom g = bn_apply_geom(bn_geom<VEC>(M, C), M);
    const dim3 grid(g.rblocks, g.ctiles), blk(DIR_TPB);
    if (res && rcoef && relu) hipLaunchKernelGGL((bn_apply_kernel<T, 2, true>), grid, blk, 0, s, x, res, y, M, C, g, coef, rcoef, bits, fin);
    else if (res && rcoef) hipLaunchKernelGGL((bn_apply_kernel<T, 2, false>), grid, blk, 0, s, x, res, y, M, C, g, coef, rcoef, bits, fin);
    else if (res && relu) hipLaunchKernelGGL((bn_apply_kernel<T, 1, true>), grid, blk, 0, s, x, res, y, M, C, g, coef, rcoef, bits, fin);
    else if (res) hipLaunchKernelGGL((bn_apply_kernel<T, 1, false>), grid, blk, 0, s, x, res, y, M, C, g, coef, rcoef, bits, fin);
    else if (relu) hipLaunchKernelGGL((bn_apply_kernel<T, 0, true>), grid, blk, 0, s, x, res, y, M, C, g, coef, rcoef, bits, fin);
    else hipLaunchKernelGGL((bn_apply_kernel<T, 0, false>), grid, blk, 0, s, x, res, y, M, C, g, coef, rcoef, bits, fin);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

template <typename T>
int fwd_impl(const void* x_, const void* res_, void* y_, int64_t M, int C, const float* gamma, const float* beta,
             float* running_mean, float* running_var, double momentum, double eps, int relu, bool training,
             float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, hipStream_t s,
             const float* ext_partial = nullptr, int ext_rows = 0, uint8_t* bits = nullptr) {
    constexpr int VEC = Vec<T>::N;
    BnWs w = bn_ws<VEC>(ws, M, C);
    DIR_RETURN_IF(ws_bytes < w.bytes, DIR_EWORKSPACE);
    if (training && BN_FUSED_FINALIZE) {
        // statistics (own pass, or the producing convolution's partial list) -> fold to <= BN_FOLD_ROWS float64 rows -> the apply
        // pass finalizes per channel tile itself (no finalize launch)
        const float* part = ext_partial;
        int prow = ext_rows;
        if (!part) {
            BnGeom g = bn_geom<VEC>(M, C);
            hipLaunchKernelGGL(bn_stats_partial_kernel<T>, dim3(g.rblocks, g.ctiles), dim3(DIR_TPB), 0, s, static_cast<const T*>(x_), M, C, g, w.partial);
            DIR_LAUNCH_CHECK();
            part = w.partial; prow = g.rblocks;
        }
        const int frows = bn_fold(part, prow, C, w.folded, s);
        DIR_LAUNCH_CHECK();
        const BnFinF fin{w.folded, frows, (double)M, momentum, eps, gamma, beta, running_mean, running_var, save_mean, save_rstd};
        return apply_impl<T>(x_, res_, nullptr, y_, M, C, nullptr, relu, s, bits, fin);
    }
    if (training) {
        const int rc = prepare_impl<T>(x_, M, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd,
                                       w.coef, ws, ws_bytes, s, ext_partial, ext_rows);
        if (rc != DIR_OK) return rc;
    } else {
        hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3(dir_cdiv(C, DIR_TPB)), dim3(DIR_TPB), 0, s, C, gamma, beta, running_mean, running_var, eps, w.coef);
        DIR_LAUNCH_CHECK();
    }
    return apply_impl<T>(x_, res_, nullptr, y_, M, C, w.coef, relu, s, bits);
}

template <typename T>
int bwd_impl(const void* dout_, const void* x_, const void* out_, void* dx_, void* dres_, int64_t M, int C,
             const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, float* dgamma,
             float* dbeta, int relu, void* ws, size_t ws_bytes, hipStream_t s, const float* ext_partial = nullptr, int ext_rows = 0) {
    constexpr int VEC = Vec<T>::N;
    const T* dout = static_cast<const T*>(dout_); const T* x = static_cast<const T*>(x_); const T* out = static_cast<const T*>(out_);
    T* dx = static_cast<T*>(dx_); T* dres = static_cast<T*>(dres_);
    BnGeom g = bn_geom<VEC>(M, C);
    BnWs w = bn_ws<VEC>(ws, M, C);
    DIR_RETURN_IF(ws_bytes < w.bytes, DIR_EWORKSPACE);
    const dim3 grid(g.rblocks, g.ctiles), blk(DIR_TPB);
    const BnMaskCoef mc{gamma, beta, save_mean, save_rstd};
    // mask source: saved output if given, else recomputed from x (only valid when no residual was added)
    const int mask = !relu ? 0 : (out ? 1 : 2);
    BnFinB fin{};
    if (BN_FUSED_FINALIZE) {
        const float* part = ext_partial;
        int prow = ext_rows;
        if (part) { DIR_RETURN_IF(mask == 1, DIR_EINVAL); }
        else {
            if (mask == 1) hipLaunchKernelGGL((bn_bwd_partial_kernel<T, 1>), grid, blk, 0, s, dout, x, out, M, C, g, w.partial, mc);
            else if (mask == 2) hipLaunchKernelGGL((bn_bwd_partial_kernel<T, 2>), grid, blk, 0, s, dout, x, out, M, C, g, w.partial, mc);
            else hipLaunchKernelGGL((bn_bwd_partial_kernel<T, 0>), grid, blk, 0, s, dout, x, out, M, C, g, w.partial, mc);
            DIR_LAUNCH_CHECK();
            part = w.partial; prow = g.rblocks;
        }
        const int frows = bn_fold(part, prow, C, w.folded, s);
        DIR_LAUNCH_CHECK();
        fin = BnFinB{w.folded, frows, (double)M, gamma, save_mean, save_rstd, dgamma, dbeta};
    } else if (ext_partial) {
        // the first pass ran inside the data-gradient kernel that produced dout (dir_conv_dgrad_bnstats): one row of partials per
        // 128-pixel tile; long lists are folded to <= 32 rows of doubles first (as in the forward, prepare_impl)
        DIR_RETURN_IF(mask == 1, DIR_EINVAL);
        int splits = ext_rows / 64;
        if (splits > 32) splits = 32;
        if (splits > g.rblocks / 2) splits = g.rblocks / 2;
        if (splits >= 16) {
            const int rps = dir_cdiv(ext_rows, splits);
            splits = dir_cdiv(ext_rows, rps);
            double* folded = reinterpret_cast<double*>(w.partial);
            hipLaunchKernelGGL(bn_fold_partials_kernel<8>, dim3(dir_cdiv(C, 8), splits), blk, 0, s, ext_partial, ext_rows, C, rps, folded);
            DIR_LAUNCH_CHECK();
            hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, double>), dim3(dir_cdiv(C, 8)), blk, 0, s, folded, splits, M, C, gamma,
                               save_mean, save_rstd, dgamma, dbeta, w.coef);
        } else {
            hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, float>), dim3(dir_cdiv(C, 8)), blk, 0, s, ext_partial, ext_rows, M, C, gamma,
                               save_mean, save_rstd, dgamma, dbeta, w.coef);
        }
        DIR_LAUNCH_CHECK();
    } else {
        if (mask == 1) hipLaunchKernelGGL((bn_bwd_partial_kernel<T, 1>), grid, blk, 0, s, dout, x, out, M, C, g, w.partial, mc);
        else if (mask == 2) hipLaunchKernelGGL((bn_bwd_partial_kernel<T, 2>), grid, blk, 0, s, dout, x, out, M, C, g, w.partial, mc);
        else hipLaunchKernelGGL((bn_bwd_partial_kernel<T, 0>), grid, blk, 0, s, dout, x, out, M, C, g, w.partial, mc);
        DIR_LAUNCH_CHECK();
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, float>), dim3(dir_cdiv(C, 8)), blk, 0, s, w.partial, g.rblocks, M, C, gamma,
                           save_mean, save_rstd, dgamma, dbeta, w.coef);
        DIR_LAUNCH_CHECK();
    }
    const BnGeom ag = bn_apply_geom(g, M);
    const dim3 agrid(ag.rblocks, ag.ctiles);
    if (mask == 1 && dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1, true>), agrid, blk, 0, s, dout, x, out, dx, dres, M, C, ag, w.coef, mc, fin);
    else if (mask == 1) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1, false>), agrid, blk, 0, s, dout, x, out, dx, dres, M, C, ag, w.coef, mc, fin);
    else if (mask == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 2, false>), agrid, blk, 0, s, dout, x, out, dx, dres, M, C, ag, w.coef, mc, fin);
    else if (dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 0, true>), agrid, blk, 0, s, dout, x, out, dx, dres, M, C, ag, w.coef, mc, fin);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 0, false>), agrid, blk, 0, s, dout, x, out, dx, dres, M, C, ag, w.coef, mc, fin);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

template <typename T>
int bwd_join_impl(const void* g_, const void* x_, const void* r_, void* dx_, void* dr_, int64_t M, int C, const float* gamma,
                  const float* mean, const float* rstd, const float* gamma_r, const float* mean_r, const float* rstd_r, float* dgamma,
                  float* dbeta, float* dgamma_r, float* dbeta_r, void* ws, size_t ws_bytes, hipStream_t s) {
    constexpr int VEC = Vec<T>::N;
    BnGeom g = bn_geom<VEC>(M, C);
    const size_t one = bn_ws<VEC>(nullptr, M, C).bytes;
    DIR_RETURN_IF(ws_bytes < 2 * one, DIR_EWORKSPACE);
    BnWs wx = bn_ws<VEC>(ws, M, C), wr = bn_ws<VEC>(static_cast<char*>(ws) + one, M, C);
    const dim3 grid(g.rblocks, g.ctiles), blk(DIR_TPB);
    hipLaunchKernelGGL(bn_bwd_join_partial_kernel<T>, grid, blk, 0, s, static_cast<const T*>(g_), static_cast<const T*>(x_),
                       static_cast<const T*>(r_), M, C, g, wx.partial, wr.partial);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, float>), dim3(dir_cdiv(C, 8)), blk, 0, s, wx.partial, g.rblocks, M, C, gamma, mean, rstd,
                       dgamma, dbeta, wx.coef);
    DIR_LAUNCH_CHECK();
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, float>), dim3(dir_cdiv(C, 8)), blk, 0, s, wr.partial, g.rblocks, M, C, gamma_r, mean_r,
                       rstd_r, dgamma_r, dbeta_r, wr.coef);
    DIR_LAUNCH_CHECK();
    const BnGeom ag = bn_apply_geom(g, M);
    hipLaunchKernelGGL(bn_bwd_join_apply_kernel<T>, dim3(ag.rblocks, ag.ctiles), blk, 0, s, static_cast<const T*>(g_), static_cast<const T*>(x_),
                       static_cast<const T*>(r_), static_cast<T*>(dx_), static_cast<T*>(dr_), M, C, ag, wx.coef, wr.coef);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

}  // namespace

extern "C" int dir_bn_bwd_join(const void* g, const void* x, const void* r, void* dx, void* dr, int dtype, int64_t M, int C,
                               const float* gamma, const float* save_mean, const float* save_rstd, const float* gamma_r,
                               const float* save_mean_r, const float* save_rstd_r, float* dgamma, float* dbeta, float* dgamma_r,
                               float* dbeta_r, void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!g || !x || !r || !dx || !dr || !gamma || !save_mean || !save_rstd || !gamma_r || !save_mean_r || !save_rstd_r, DIR_EINVAL);
    DIR_RETURN_IF(!dgamma || !dbeta || !dgamma_r || !dbeta_r || !workspace, DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!bn_shape_ok(dtype, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(g) || !dir_aligned16(x) || !dir_aligned16(r) || !dir_aligned16(dx) || !dir_aligned16(dr), DIR_EINVAL);
    if (dtype == DIR_BF16)
        return bwd_join_impl<bf16_t>(g, x, r, dx, dr, M, C, gamma, save_mean, save_rstd, gamma_r, save_mean_r, save_rstd_r, dgamma, dbeta,
                                     dgamma_r, dbeta_r, workspace, workspace_bytes, dir_s(stream));
    return bwd_join_impl<float>(g, x, r, dx, dr, M, C, gamma, save_mean, save_rstd, gamma_r, save_mean_r, save_rstd_r, dgamma, dbeta,
                                dgamma_r, dbeta_r, workspace, workspace_bytes, dir_s(stream));
}


extern "C" size_t dir_bn_workspace(int dtype, int64_t M, int C) {
    if (!bn_shape_ok(dtype, M, C)) return 0;
    return dtype == DIR_BF16 ? bn_ws<8>(nullptr, M, C).bytes : bn_ws<4>(nullptr, M, C).bytes;
}

extern "C" int dir_bn_fwd_train(const void* x, const void* residual, void* y, int dtype, int64_t M, int C,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                double momentum, double eps, int relu, float* save_mean, float* save_rstd,
                                void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !gamma || !beta || !save_mean || !save_rstd || !workspace, DIR_EINVAL);
    DIR_RETURN_IF((running_mean == nullptr) != (running_var == nullptr), DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!bn_shape_ok(dtype, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(y) || (residual && !dir_aligned16(residual)), DIR_EINVAL);
    if (dtype == DIR_BF16)
        return fwd_impl<bf16_t>(x, residual, y, M, C, gamma, beta, running_mean, running_var, momentum, eps, relu, true,
                                save_mean, save_rstd, workspace, workspace_bytes, dir_s(stream));
    return fwd_impl<float>(x, residual, y, M, C, gamma, beta, running_mean, running_var, momentum, eps, relu, true,
                           save_mean, save_rstd, workspace, workspace_bytes, dir_s(stream));
}

extern "C" int dir_bn_fwd_train_partials(const void* x, const void* residual, void* y, int dtype, int64_t M, int C,
                                         const float* partial, int partial_rows,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                                         double momentum, double eps, int relu, float* save_mean, float* save_rstd,
                                         void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !gamma || !beta || !save_mean || !save_rstd || !workspace || !partial || partial_rows <= 0, DIR_EINVAL);
    DIR_RETURN_IF((running_mean == nullptr) != (running_var == nullptr), DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!bn_shape_ok(dtype, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(y) || (residual && !dir_aligned16(residual)), DIR_EINVAL);
    if (dtype == DIR_BF16)
        return fwd_impl<bf16_t>(x, residual, y, M, C, gamma, beta, running_mean, running_var, momentum, eps, relu, true,
                                save_mean, save_rstd, workspace, workspace_bytes, dir_s(stream), partial, partial_rows);
    return fwd_impl<float>(x, residual, y, M, C, gamma, beta, running_mean, running_var, momentum, eps, relu, true,
                           save_mean, save_rstd, workspace, workspace_bytes, dir_s(stream), partial, partial_rows);
}

extern "C" int dir_bn_fwd_eval(const void* x, const void* residual, void* y, int dtype, int64_t M, int C,
                               const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                               double eps, int relu, void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !gamma || !beta || !running_mean || !running_var || !workspace, DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!bn_shape_ok(dtype, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(y) || (residual && !dir_aligned16(residual)), DIR_EINVAL);
    float* rm = const_cast<float*>(running_mean); float* rv = const_cast<float*>(running_var);
    if (dtype == DIR_BF16)
        return fwd_impl<bf16_t>(x, residual, y, M, C, gamma, beta, rm, rv, 0.0, eps, relu, false, nullptr, nullptr,
                                workspace, workspace_bytes, dir_s(stream));
    return fwd_impl<float>(x, residual, y, M, C, gamma, beta, rm, rv, 0.0, eps, relu, false, nullptr, nullptr,
                           workspace, workspace_bytes, dir_s(stream));
}

extern "C" int dir_bn_bwd(const void* dout, const void* x, const void* out, void* dx, void* dres, int dtype,
                          int64_t M, int C, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_rstd, float* dgamma, float* dbeta, int relu, void* workspace,
                          size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!dout || !x || !dx || !gamma || !save_mean || !save_rstd || !dgamma || !dbeta || !workspace, DIR_EINVAL);
    DIR_RETURN_IF(relu && !out && (!beta || dres), DIR_EINVAL);     // mask recompute needs beta and no residual
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!bn_shape_ok(dtype, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(dout) || !dir_aligned16(x) || !dir_aligned16(dx) || (out && !dir_aligned16(out)) ||
                  (dres && !dir_aligned16(dres)), DIR_EINVAL);
    if (dtype == DIR_BF16)
        return bwd_impl<bf16_t>(dout, x, out, dx, dres, M, C, gamma, beta, save_mean, save_rstd, dgamma, dbeta, relu,
                                workspace, workspace_bytes, dir_s(stream));
    return bwd_impl<float>(dout, x, out, dx, dres, M, C, gamma, beta, save_mean, save_rstd, dgamma, dbeta, relu,
                           workspace, workspace_bytes, dir_s(stream));
}

// dir_bn_bwd whose first pass (the per-channel sums of g and g*x) already ran inside the data-gradient kernel that produced
// dout (dir_conv_dgrad_bnstats / dir_conv_dgrad_s2_bnstats): `partial` = its [partial_rows][2][C] output. relu != 0: the ReLU
// mask is recomputed from x in the apply pass (no residual); a mask from a saved output is not a case of this entry point.
extern "C" int dir_bn_bwd_partials(const void* dout, const void* x, void* dx, int dtype, int64_t M, int C, const float* gamma,
                                   const float* beta, const float* save_mean, const float* save_rstd, float* dgamma,
                                   float* dbeta, int relu, const float* partial, int partial_rows, void* workspace,
                                   size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!dout || !x || !dx || !gamma || !save_mean || !save_rstd || !dgamma || !dbeta || !workspace, DIR_EINVAL);
    DIR_RETURN_IF(!partial || partial_rows <= 0 || (relu && !beta), DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!bn_shape_ok(dtype, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(dout) || !dir_aligned16(x) || !dir_aligned16(dx), DIR_EINVAL);
    if (dtype == DIR_BF16)
        return bwd_impl<bf16_t>(dout, x, nullptr, dx, nullptr, M, C, gamma, beta, save_mean, save_rstd, dgamma, dbeta, relu,
                                workspace, workspace_bytes, dir_s(stream), partial, partial_rows);
    return bwd_impl<float>(dout, x, nullptr, dx, nullptr, M, C, gamma, beta, save_mean, save_rstd, dgamma, dbeta, relu,
                           workspace, workspace_bytes, dir_s(stream), partial, partial_rows);
}

extern "C" int dir_bn_prepare_train(const void* x, int dtype, int64_t M, int C, const float* partial, int partial_rows,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    double momentum, double eps, float* save_mean, float* save_rstd, float* coef,
                                    void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !gamma || !beta || !save_mean || !save_rstd || !coef || !workspace, DIR_EINVAL);
    DIR_RETURN_IF((running_mean == nullptr) != (running_var == nullptr) || (partial && partial_rows <= 0), DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!bn_shape_ok(dtype, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x), DIR_EINVAL);
    if (dtype == DIR_BF16)
        return prepare_impl<bf16_t>(x, M, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, coef,
                                    workspace, workspace_bytes, dir_s(stream), partial, partial_rows);
    return prepare_impl<float>(x, M, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, coef,
                               workspace, workspace_bytes, dir_s(stream), partial, partial_rows);
}

extern "C" int dir_bn_apply(const void* x, const void* residual, const float* residual_coef, void* y, int dtype, int64_t M,
                            int C, const float* coef, int relu, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !coef || (residual_coef && !residual), DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!bn_shape_ok(dtype, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(y) || (residual && !dir_aligned16(residual)), DIR_EINVAL);
    if (dtype == DIR_BF16) return apply_impl<bf16_t>(x, residual, residual_coef, y, M, C, coef, relu, dir_s(stream));
    return apply_impl<float>(x, residual, residual_coef, y, M, C, coef, relu, dir_s(stream));
}

// dir_bn_fwd_train[_partials] / dir_bn_apply that also emit the ReLU's backward mask as one bit per element (relu_bits
// [M][C/8] bytes, bit j of byte (m, c/8) = y[m][8 (c/8) + j] > 0): the consumer that applies this ReLU's backward inside its
// data-gradient store loop (dir_conv_dgrad_ex, relu_mask_bits) then reads M*C/8 bytes instead of the 2*M*C of y. bf16, relu != 0.
extern "C" int dir_bn_fwd_train_bits(const void* x, const void* residual, void* y, int64_t M, int C, const float* partial,
                                     int partial_rows, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, double momentum, double eps, float* save_mean, float* save_rstd,
                                     void* relu_bits_out, void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !gamma || !beta || !save_mean || !save_rstd || !workspace || !relu_bits_out, DIR_EINVAL);
    DIR_RETURN_IF((running_mean == nullptr) != (running_var == nullptr) || (partial && partial_rows <= 0), DIR_EINVAL);
    DIR_RETURN_IF(!bn_shape_ok(DIR_BF16, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(y) || (residual && !dir_aligned16(residual)), DIR_EINVAL);
    return fwd_impl<bf16_t>(x, residual, y, M, C, gamma, beta, running_mean, running_var, momentum, eps, 1, true, save_mean, save_rstd,
                            workspace, workspace_bytes, dir_s(stream), partial, partial_rows, static_cast<uint8_t*>(relu_bits_out));
}

extern "C" int dir_bn_apply_bits(const void* x, const void* residual, const float* residual_coef, void* y, int64_t M, int C,
                                 const float* coef, void* relu_bits_out, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !coef || (residual_coef && !residual) || !relu_bits_out, DIR_EINVAL);
    DIR_RETURN_IF(!bn_shape_ok(DIR_BF16, M, C), DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(y) || (residual && !dir_aligned16(residual)), DIR_EINVAL);
    return apply_impl<bf16_t>(x, residual, residual_coef, y, M, C, coef, 1, dir_s(stream), static_cast<uint8_t*>(relu_bits_out));
}

// (internal, used by dir_pool.hip's fused stem tail) partial [rows][2][C] of (sum g, sum g*x) -> dgamma, dbeta, coef[3][C]
extern "C" int dir_bn_bwd_finalize(const float* partial, int rows, int64_t M, int C, const float* gamma, const float* save_mean,
                                   const float* save_rstd, float* dgamma, float* dbeta, float* coef, dir_stream_t stream) {
    DIR_RETURN_IF(!partial || rows <= 0 || M <= 0 || C <= 0 || !gamma || !save_mean || !save_rstd || !dgamma || !dbeta || !coef, DIR_EINVAL);
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, float>), dim3(dir_cdiv(C, 8)), dim3(DIR_TPB), 0, dir_s(stream), partial, rows, M, C, gamma,
                       save_mean, save_rstd, dgamma, dbeta, coef);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

