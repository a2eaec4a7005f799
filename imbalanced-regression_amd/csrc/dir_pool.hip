// 3x3 / stride 2 / pad 1 max pooling for NHWC bf16 activations (the stem pool of imdb-wiki-dir/resnet.py:82,131),
// forward with an argmax byte per element and a gather-style (atomic-free) backward. HBM-bound streaming kernels:
// a thread owns 8 consecutive channels (16 B) of one pixel.
#include "dir_common.h"

namespace {

__device__ __forceinline__ float pl_bf2f(uint32_t h) { return __uint_as_float(h << 16); }

__global__ void __launch_bounds__(DIR_TPB)
maxpool_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, uint8_t* __restrict__ idx,
                   int N, int H, int W, int C, int Ho, int Wo) {
    const int cg = C / 8;
    const long long total = (long long)N * Ho * Wo * cg;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int g = (int)(i % cg); long long p = i / cg;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
        float best[8]; uint32_t bi[8], raw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; raw[j] = 0xff80u; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)n * H + hi) * W + wi) * C + g * 8);
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t lo = w4[q] & 0xffffu, hi16 = w4[q] >> 16;
                    const float f0 = pl_bf2f(lo), f1 = pl_bf2f(hi16);
                    if (f0 > best[2 * q] || f0 != f0) { best[2 * q] = f0; bi[2 * q] = r * 3 + s; raw[2 * q] = lo; }        // first max wins
                    if (f1 > best[2 * q + 1] || f1 != f1) { best[2 * q + 1] = f1; bi[2 * q + 1] = r * 3 + s; raw[2 * q + 1] = hi16; }
                }
            }
        }
        const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + g * 8;
        *reinterpret_cast<uint4*>(y + o) = make_uint4(raw[0] | (raw[1] << 16), raw[2] | (raw[3] << 16), raw[4] | (raw[5] << 16), raw[6] | (raw[7] << 16));
        *reinterpret_cast<uint2*>(idx + o) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                                        bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
    }
}

// dx[n,h,w,c] = sum over the (<= 2x2) windows containing (h,w) whose argmax is (h,w) of dy[window]
__global__ void __launch_bounds__(DIR_TPB)
maxpool_bwd_kernel(const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx, uint16_t* __restrict__ dx,
                   int N, int H, int W, int C, int Ho, int Wo) {
    const int cg = C / 8;
    const long long total = (long long)N * H * W * cg;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int g = (int)(i % cg); long long p = i / cg;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H); const int n = (int)(p / H);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        // windows with 2*ho - 1 <= h <= 2*ho + 1  ->  ho in {(h-1+1)/2 ...}: ho = (h + 1) / 2 and, if h odd, also (h - 1) / 2 ... enumerate both candidates
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ho = (h + 1) / 2 - a;                     // r = h - (2*ho - 1)
            const int r = h - (2 * ho - 1);
            if (ho < 0 || ho >= Ho || r < 0 || r > 2) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int wo = (w + 1) / 2 - b;
                const int s = w - (2 * wo - 1);
                if (wo < 0 || wo >= Wo || s < 0 || s > 2) continue;
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + g * 8;
                const uint2 iv = *reinterpret_cast<const uint2*>(idx + o);
                const uint4 gv = *reinterpret_cast<const uint4*>(dy + o);
                const uint32_t me = (uint32_t)(r * 3 + s);
                const uint32_t g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t k = ((j < 4 ? iv.x : iv.y) >> (8 * (j & 3))) & 0xffu;
                    const uint32_t hbits = (j & 1) ? (g4[j >> 1] >> 16) : (g4[j >> 1] & 0xffffu);
                    if (k == me) acc[j] += pl_bf2f(hbits);
                }
            }
        }
        uint32_t o16[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t u = __float_as_uint(acc[j]);
            o16[j] = ((u & 0x7fffffffu) > 0x7f800000u) ? ((u >> 16) | 0x40u) : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
        *reinterpret_cast<uint4*>(dx + (((size_t)n * H + h) * W + w) * C + g * 8) =
            make_uint4(o16[0] | (o16[1] << 16), o16[2] | (o16[3] << 16), o16[4] | (o16[5] << 16), o16[6] | (o16[7] << 16));
    }
}

}  // namespace

extern "C" int dir_maxpool3x3s2_fwd(const void* x, void* y, void* argmax, int N, int H, int W, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0, DIR_EINVAL);
    DIR_RETURN_IF(C % 8 != 0, DIR_EUNSUPPORTED);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 8);
    int grid = (int)((total + DIR_TPB - 1) / DIR_TPB); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const uint16_t*>(x),
                       static_cast<uint16_t*>(y), static_cast<uint8_t*>(argmax), N, H, W, C, Ho, Wo);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_maxpool3x3s2_bwd(const void* dy, const void* argmax, void* dx, int N, int H, int W, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !dx || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0, DIR_EINVAL);
    DIR_RETURN_IF(C % 8 != 0, DIR_EUNSUPPORTED);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * H * W * (C / 8);
    int grid = (int)((total + DIR_TPB - 1) / DIR_TPB); if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const uint16_t*>(dy),
                       static_cast<const uint8_t*>(argmax), static_cast<uint16_t*>(dx), N, H, W, C, Ho, Wo);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stem tail: relu(bn1(x)) -> MaxPool2d(3, 2, 1) (resnet.py:80-82,129-131) in one pass over the BatchNorm INPUT: the
// normalised 112x112 map (411 MB at batch 256) is never written or re-read. Forward: per window the maximum of
// x * a + b (float), clipped at 0, rounded to bf16 once; argmax byte 0..8 = window position, 9 = clipped by the ReLU (no
// gradient). Backward: the BatchNorm reductions run over the POOLED gradient (sum g = sum of unclipped dy,
// sum g x = sum dy * x[argmax]); the apply pass is the gather-style pool backward with dx = a g + p x + q folded in.
namespace {
constexpr int SP_BLOCKS = 1024;                                   // workgroups (= partial rows) of the reduction pass

__global__ void __launch_bounds__(DIR_TPB)
bn_relu_maxpool_fwd_kernel(const uint16_t* __restrict__ x, const float* __restrict__ coef, uint16_t* __restrict__ y,
                           uint8_t* __restrict__ idx, uint16_t* __restrict__ xmax, int N, int H, int W, int C, int Ho, int Wo) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const int cg = C / 8;
    const long long total = (long long)N * Ho * Wo * cg;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int g = (int)(i % cg); long long p = i / cg;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
        float a[8], b[8], best[8]; uint32_t bi[8], bx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = coef[g * 8 + j]; b[j] = coef[C + g * 8 + j]; best[j] = -INFINITY; bi[j] = 9; bx[j] = 0; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)n * H + hi) * W + wi) * C + g * 8);
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float f0 = pl_bf2f(w4[q] & 0xffffu) * a[2 * q] + b[2 * q];
                    const float f1 = pl_bf2f(w4[q] >> 16) * a[2 * q + 1] + b[2 * q + 1];
                    if (f0 > best[2 * q]) { best[2 * q] = f0; bi[2 * q] = r * 3 + s; bx[2 * q] = w4[q] & 0xffffu; }   // first max wins
                    if (f1 > best[2 * q + 1]) { best[2 * q + 1] = f1; bi[2 * q + 1] = r * 3 + s; bx[2 * q + 1] = w4[q] >> 16; }
                }
            }
        }
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v0 = best[2 * q], v1 = best[2 * q + 1];
            if (!(v0 > 0.0f)) { v0 = 0.0f; bi[2 * q] = 9; }                                               // ReLU
            if (!(v1 > 0.0f)) { v1 = 0.0f; bi[2 * q + 1] = 9; }
            const f32x2_t t = {v0, v1};
            o[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2_t));
        }
        const size_t oo = (((size_t)n * Ho + ho) * Wo + wo) * C + g * 8;
        *reinterpret_cast<uint4*>(y + oo) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint2*>(idx + oo) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                                         bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
        // the BatchNorm INPUT at the argmax (bf16 bits as read): the backward's sum g * x then streams this pooled-size tensor
        // instead of gathering 2-byte elements out of the 4x larger map
        if (xmax) *reinterpret_cast<uint4*>(xmax + oo) = make_uint4(bx[0] | (bx[1] << 16), bx[2] | (bx[3] << 16), bx[4] | (bx[5] << 16), bx[6] | (bx[7] << 16));
    }
}

// partial[block][2][C]: sums of g and g * x over the windows this workgroup visits (fixed order: deterministic)
__global__ void __launch_bounds__(DIR_TPB)
bn_relu_maxpool_bwd_partial_kernel(const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx, const uint16_t* __restrict__ x,
                                   const uint16_t* __restrict__ xmax, float* __restrict__ partial, int N, int H, int W, int C, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float sh[];    // [2][DIR_TPB][8]
    const int cg = C / 8;                                         // DIR_TPB % cg == 0: a thread keeps its channel group
    const long long total = (long long)N * Ho * Wo * cg;
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.0f; s1[j] = 0.0f; }
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int g = (int)(i % cg); long long p = i / cg;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
        const size_t oo = (((size_t)n * Ho + ho) * Wo + wo) * C + g * 8;
        const uint2 iv = *reinterpret_cast<const uint2*>(idx + oo);
        const uint4 gv = *reinterpret_cast<const uint4*>(dy + oo);
        const uint32_t g4[4] = {gv.x, gv.y, gv.z, gv.w};
        if (xmax) {                                               // x[argmax] was kept by the forward: three streams, no gather
            const uint4 mv = *reinterpret_cast<const uint4*>(xmax + oo);
            const uint32_t m4[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = ((j < 4 ? iv.x : iv.y) >> (8 * (j & 3))) & 0xffu;
                if (k < 9u) {
                    const float gj = pl_bf2f((j & 1) ? (g4[j >> 1] >> 16) : (g4[j >> 1] & 0xffffu));
                    const float xv = pl_bf2f((j & 1) ? (m4[j >> 1] >> 16) : (m4[j >> 1] & 0xffffu));
                    s0[j] += gj; s1[j] += gj * xv;
                }
            }
            continue;
        }
        const size_t xb = (((size_t)n * H + (2 * ho - 1)) * W + (2 * wo - 1)) * C + g * 8;   // window origin (may lie outside: never dereferenced there)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t k = ((j < 4 ? iv.x : iv.y) >> (8 * (j & 3))) & 0xffu;
            if (k < 9u) {
                const float gj = pl_bf2f((j & 1) ? (g4[j >> 1] >> 16) : (g4[j >> 1] & 0xffffu));
                const uint32_t r = k / 3u, s2 = k - 3u * r;
                const float xv = pl_bf2f(x[xb + ((size_t)r * W + s2) * C + j]);
                s0[j] += gj; s1[j] += gj * xv;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[(0 * DIR_TPB + threadIdx.x) * 8 + j] = s0[j]; sh[(1 * DIR_TPB + threadIdx.x) * 8 + j] = s1[j]; }
    __syncthreads();
    const int lanes = DIR_TPB / cg;                               // threads per channel group
    if ((int)threadIdx.x < 2 * C) {
        const int which = threadIdx.x / C, c = threadIdx.x - which * C, g = c / 8, j = c - g * 8;
        float sum = 0.0f;
        for (int l = 0; l < lanes; ++l) sum += sh[(which * DIR_TPB + l * cg + g) * 8 + j];
        partial[((size_t)blockIdx.x * 2 + which) * C + c] = sum;
    }
}

// coef[3][C] = (a, p, q) of the BatchNorm backward: dx = a g + p x + q with g gathered from the pooled gradient
__global__ void __launch_bounds__(DIR_TPB)
bn_relu_maxpool_bwd_apply_kernel(const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx, const uint16_t* __restrict__ x,
                                 const float* __restrict__ coef, uint16_t* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const int cg = C / 8;
    const long long total = (long long)N * H * W * cg;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int g = (int)(i % cg); long long p = i / cg;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H); const int n = (int)(p / H);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {                             // the (<= 2x2) windows containing (h, w)
            const int ho = (h + 1) / 2 - a;
            const int r = h - (2 * ho - 1);
            if (ho < 0 || ho >= Ho || r < 0 || r > 2) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int wo = (w + 1) / 2 - b;
                const int s = w - (2 * wo - 1);
                if (wo < 0 || wo >= Wo || s < 0 || s > 2) continue;
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + g * 8;
                const uint2 iv = *reinterpret_cast<const uint2*>(idx + o);
                const uint4 gv = *reinterpret_cast<const uint4*>(dy + o);
                const uint32_t me = (uint32_t)(r * 3 + s);
                const uint32_t g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t k = ((j < 4 ? iv.x : iv.y) >> (8 * (j & 3))) & 0xffu;
                    const uint32_t hbits = (j & 1) ? (g4[j >> 1] >> 16) : (g4[j >> 1] & 0xffffu);
                    if (k == me) acc[j] += pl_bf2f(hbits);
                }
            }
        }
        const size_t xo = (((size_t)n * H + h) * W + w) * C + g * 8;
        const uint4 xv = *reinterpret_cast<const uint4*>(x + xo);
        const uint32_t x4[4] = {xv.x, xv.y, xv.z, xv.w};
        uint32_t o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = g * 8 + 2 * q;
            const float v0 = coef[c0] * acc[2 * q] + coef[C + c0] * pl_bf2f(x4[q] & 0xffffu) + coef[2 * C + c0];
            const float v1 = coef[c0 + 1] * acc[2 * q + 1] + coef[C + c0 + 1] * pl_bf2f(x4[q] >> 16) + coef[2 * C + c0 + 1];
            const f32x2_t t = {v0, v1};
            o4[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2_t));
        }
        *reinterpret_cast<uint4*>(dx + xo) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
    }
}
// The same pass with a thread owning the 2 x 2 block of input pixels (2a..2a+1, 2b..2b+1) of its 8 channels: the four pooling
// windows (a..a+1, b..b+1) that cover the block are loaded once (the per-pixel kernel above loads 2.25 windows per pixel on
// average, 9 per block) and each pixel adds its windows in the order the per-pixel kernel does — bit-identical results.
__device__ __forceinline__ void pl_take(float (&acc)[8], const uint2& iv, const uint4& gv, uint32_t me) {
    const uint32_t g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t k = ((j < 4 ? iv.x : iv.y) >> (8 * (j & 3))) & 0xffu;
        const uint32_t hbits = (j & 1) ? (g4[j >> 1] >> 16) : (g4[j >> 1] & 0xffffu);
        if (k == me) acc[j] += pl_bf2f(hbits);
    }
}
__global__ void __launch_bounds__(DIR_TPB)
bn_relu_maxpool_bwd_apply2x2_kernel(const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx, const uint16_t* __restrict__ x,
                                    const float* __restrict__ coef, uint16_t* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const int cg = C / 8, Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    const long long total = (long long)N * Hb * Wb * cg;
    const long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x;
    if (i >= total) return;
    const int g = (int)(i % cg); long long p = i / cg;
    const int b = (int)(p % Wb); p /= Wb;
    const int a = (int)(p % Hb); const int n = (int)(p / Hb);
    float ca[8], cp[8], cq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ca[j] = coef[g * 8 + j]; cp[j] = coef[C + g * 8 + j]; cq[j] = coef[2 * C + g * 8 + j]; }
    // windows (a + u, b + v); a < Ho and b < Wo always (Ho = (H - 1) / 2 + 1 >= Hb)
    uint2 iv[2][2]; uint4 gv[2][2]; bool ok[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            ok[u][v] = (a + u < Ho) && (b + v < Wo);
            if (ok[u][v]) {
                const size_t o = (((size_t)n * Ho + a + u) * Wo + b + v) * C + g * 8;
                iv[u][v] = *reinterpret_cast<const uint2*>(idx + o);
                gv[u][v] = *reinterpret_cast<const uint4*>(dy + o);
            }
        }
    const bool h1 = 2 * a + 1 < H, w1 = 2 * b + 1 < W;
    uint4 xv[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
            if ((u == 0 || h1) && (v == 0 || w1)) xv[u][v] = *reinterpret_cast<const uint4*>(x + (((size_t)n * H + 2 * a + u) * W + 2 * b + v) * C + g * 8);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            if (!((u == 0 || h1) && (v == 0 || w1))) continue;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
            // pixel (2a + u, 2b + v): window rows {a + 1 (r = 0), a (r = 2)} for u = 1, {a (r = 1)} for u = 0; columns alike; larger index first
            if (u == 1 && v == 1) {
                if (ok[1][1]) pl_take(acc, iv[1][1], gv[1][1], 0u);
                if (ok[1][0]) pl_take(acc, iv[1][0], gv[1][0], 2u);
                if (ok[0][1]) pl_take(acc, iv[0][1], gv[0][1], 6u);
                pl_take(acc, iv[0][0], gv[0][0], 8u);
            } else if (u == 1) {
                if (ok[1][0]) pl_take(acc, iv[1][0], gv[1][0], 1u);
                pl_take(acc, iv[0][0], gv[0][0], 7u);
            } else if (v == 1) {
                if (ok[0][1]) pl_take(acc, iv[0][1], gv[0][1], 3u);
                pl_take(acc, iv[0][0], gv[0][0], 5u);
            } else {
                pl_take(acc, iv[0][0], gv[0][0], 4u);
            }
            const uint32_t x4[4] = {xv[u][v].x, xv[u][v].y, xv[u][v].z, xv[u][v].w};
            uint32_t o4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v0 = ca[2 * q] * acc[2 * q] + cp[2 * q] * pl_bf2f(x4[q] & 0xffffu) + cq[2 * q];
                const float v1 = ca[2 * q + 1] * acc[2 * q + 1] + cp[2 * q + 1] * pl_bf2f(x4[q] >> 16) + cq[2 * q + 1];
                const f32x2_t t = {v0, v1};
                o4[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2_t));
            }
            *reinterpret_cast<uint4*>(dx + (((size_t)n * H + 2 * a + u) * W + 2 * b + v) * C + g * 8) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        }
}
}  // namespace

extern "C" int dir_bn_relu_maxpool_fwd_xmax(const void* x, const float* coef, void* y, void* argmax, void* xmax, int N, int H, int W, int C,
                                            dir_stream_t stream) {
    DIR_RETURN_IF(!x || !coef || !y || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0, DIR_EINVAL);
    DIR_RETURN_IF(C % 8 != 0, DIR_EUNSUPPORTED);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 8);
    int grid = (int)((total + DIR_TPB - 1) / DIR_TPB); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const uint16_t*>(x), coef,
                       static_cast<uint16_t*>(y), static_cast<uint8_t*>(argmax), static_cast<uint16_t*>(xmax), N, H, W, C, Ho, Wo);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
extern "C" int dir_bn_relu_maxpool_fwd(const void* x, const float* coef, void* y, void* argmax, int N, int H, int W, int C,
                                       dir_stream_t stream) {
    return dir_bn_relu_maxpool_fwd_xmax(x, coef, y, argmax, nullptr, N, H, W, C, stream);
}

extern "C" size_t dir_bn_relu_maxpool_bwd_workspace(int C) {
    if (C <= 0 || C % 8 != 0) return 0;
    return dir_align_up(sizeof(float) * (size_t)SP_BLOCKS * 2 * C, 256) + dir_align_up(sizeof(float) * 3 * (size_t)C, 256);
}

// defined in dir_bn.hip: partial [rows][2][C] -> dgamma, dbeta, coef[3][C]
extern "C" int dir_bn_bwd_finalize(const float* partial, int rows, int64_t M, int C, const float* gamma, const float* save_mean,
                                   const float* save_rstd, float* dgamma, float* dbeta, float* coef, dir_stream_t stream);

extern "C" int dir_bn_relu_maxpool_bwd_xmax(const void* dy, const void* argmax, const void* x, const void* xmax, void* dx, int N, int H, int W, int C,
                                       const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma,
                                       float* dbeta, void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !argmax || !x || !dx || !gamma || !save_mean || !save_rstd || !dgamma || !dbeta || !workspace, DIR_EINVAL);
    DIR_RETURN_IF(N <= 0 || H <= 0 || W <= 0 || C <= 0, DIR_EINVAL);
    DIR_RETURN_IF(C % 8 != 0 || DIR_TPB % (C / 8) != 0 || 2 * C > DIR_TPB, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(workspace_bytes < dir_bn_relu_maxpool_bwd_workspace(C), DIR_EWORKSPACE);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    float* partial = static_cast<float*>(workspace);
    float* coef = reinterpret_cast<float*>(static_cast<char*>(workspace) + dir_align_up(sizeof(float) * (size_t)SP_BLOCKS * 2 * C, 256));
    hipStream_t s = dir_s(stream);
    hipLaunchKernelGGL(bn_relu_maxpool_bwd_partial_kernel, dim3(SP_BLOCKS), dim3(DIR_TPB), 2 * DIR_TPB * 8 * sizeof(float), s,
                       static_cast<const uint16_t*>(dy), static_cast<const uint8_t*>(argmax), static_cast<const uint16_t*>(x),
                       static_cast<const uint16_t*>(xmax), partial, N, H, W, C, Ho, Wo);
    DIR_LAUNCH_CHECK();
    const int rc = dir_bn_bwd_finalize(partial, SP_BLOCKS, (int64_t)N * H * W, C, gamma, save_mean, save_rstd, dgamma, dbeta, coef, stream);
    if (rc != DIR_OK) return rc;
    if (xmax) {
        const long long blocks = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
        hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply2x2_kernel, dim3((unsigned)((blocks + DIR_TPB - 1) / DIR_TPB)), dim3(DIR_TPB), 0, s,
                           static_cast<const uint16_t*>(dy), static_cast<const uint8_t*>(argmax), static_cast<const uint16_t*>(x), coef,
                           static_cast<uint16_t*>(dx), N, H, W, C, Ho, Wo);
    } else {
        const long long total = (long long)N * H * W * (C / 8);
        int grid = (int)((total + DIR_TPB - 1) / DIR_TPB); if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_kernel, dim3(grid), dim3(DIR_TPB), 0, s, static_cast<const uint16_t*>(dy),
                           static_cast<const uint8_t*>(argmax), static_cast<const uint16_t*>(x), coef, static_cast<uint16_t*>(dx), N, H, W, C, Ho, Wo);
    }
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_bn_relu_maxpool_bwd(const void* dy, const void* argmax, const void* x, void* dx, int N, int H, int W, int C,
                                       const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma,
                                       float* dbeta, void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    return dir_bn_relu_maxpool_bwd_xmax(dy, argmax, x, nullptr, dx, N, H, W, C, gamma, save_mean, save_rstd, dgamma, dbeta, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Global average pool of the final [N, HW, C] bf16 map -> [N, C] float32 (resnet.py:85,136: AvgPool2d(7) on the 7x7
// map) and its backward. The result goes straight into the float32 FDS / linear / loss tail, so the mean is formed
// and kept in float32 (the library pool rounds it to bf16 first and its backward runs at 0.4 TB/s).
namespace {
__global__ void __launch_bounds__(DIR_TPB)
avgpool_fwd_kernel(const uint16_t* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const int c8 = C / 8;
    const int i = blockIdx.x * DIR_TPB + threadIdx.x;                  // (n, 8-channel group)
    if (i >= N * c8) return;
    const int n = i / c8, g = i - n * c8;
    const uint16_t* p = x + ((size_t)n * HW) * C + g * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < HW; ++k) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + (size_t)k * C);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(w[q] << 16); acc[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u); }
    }
    const float inv = 1.0f / (float)HW;
    float* o = y + (size_t)n * C + g * 8;
    *reinterpret_cast<float4*>(o) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
}

__global__ void __launch_bounds__(DIR_TPB)
avgpool_bwd_kernel(const float* __restrict__ dy, uint16_t* __restrict__ dx, int N, int HW, int C) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const int c8 = C / 8;
    const size_t i = (size_t)blockIdx.x * DIR_TPB + threadIdx.x;       // (n, pixel, 8-channel group)
    if (i >= (size_t)N * HW * c8) return;
    const int g = (int)(i % c8);
    const int n = (int)(i / ((size_t)HW * c8));
    const float inv = 1.0f / (float)HW;
    const float* s = dy + (size_t)n * C + g * 8;
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
    const float v[8] = {a.x * inv, a.y * inv, a.z * inv, a.w * inv, b.x * inv, b.y * inv, b.z * inv, b.w * inv};
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x2_t t = {v[2 * q], v[2 * q + 1]}; w[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2_t)); }
    *reinterpret_cast<uint4*>(dx + i * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}
}  // namespace

extern "C" int dir_avgpool_fwd(const void* x, float* y, int N, int HW, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || N <= 0 || HW <= 0 || C <= 0, DIR_EINVAL);
    DIR_RETURN_IF(C % 8 != 0, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(y), DIR_EINVAL);
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(dir_cdiv((long long)N * (C / 8), DIR_TPB)), dim3(DIR_TPB), 0, dir_s(stream),
                       static_cast<const uint16_t*>(x), y, N, HW, C);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_avgpool_bwd(const float* dy, void* dx, int N, int HW, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !dx || N <= 0 || HW <= 0 || C <= 0, DIR_EINVAL);
    DIR_RETURN_IF(C % 8 != 0, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(dy) || !dir_aligned16(dx), DIR_EINVAL);
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(dir_cdiv((long long)N * HW * (C / 8), DIR_TPB)), dim3(DIR_TPB), 0, dir_s(stream),
                       dy, static_cast<uint16_t*>(dx), N, HW, C);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

