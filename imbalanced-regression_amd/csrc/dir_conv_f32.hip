// Exact-float32 convolutions for the PARITY MODE of the ResNet-50 stack (amp_dtype=None): forward, data gradient and weight
// gradient of any nn.Conv2d of imdb-wiki-dir/resnet.py:44-49,79,112-116 (7x7/2 stem, 1x1, 3x3, stride 1 or 2) on NHWC float32
// activations and [Cout][R][S][Cin] float32 weights, as implicit GEMMs on v_mfma_f32_32x32x2_f32.
//
// Why it exists: north_star's "training loss within 1e-5 relative" has to be shown on hand-written kernels, not on a library
// convolution. The f32-input MFMA is bit-for-bit a k-ordered fmaf chain (one rounding per product, fp32 accumulate,
// MI355X_MICROARCH.md "Matrix cores"), i.e. the same arithmetic class as the reference's fp32 cuDNN / CPU convolutions, at the
// f32 vector rate (157 TFLOP/s peak = 1/16 of bf16 MFMA). The product path stays the bf16 kernels of dir_conv.hip; this file
// is what `resnet50` runs on when it is asked for float32, so that the SAME autograd graph (fused BatchNorm nodes, joins,
// FDS / loss tail) can be compared with the reference at float32 accuracy.
//
// One tile shape for all three GEMMs: workgroup 256 threads = 4 wavefronts, output tile 64 x 64 (each wavefront one 32 x 32
// MFMA tile), K-step 16 (8 MFMAs per wavefront per step). Operands are gathered element-wise (4-byte loads, index decode per
// K-step) into k-major LDS tiles [16][68] so that fragment reads are 32 consecutive floats per half-wave (conflict free);
// the next K-step's 8 elements per thread are fetched into registers while the current one is multiplied.
//   forward : M = N*Ho*Wo, Ncol = Cout,     K = R*S*Cin    A gathers x (zero outside the image), B = w rows
//   dgrad   : M = N*H*W,   Ncol = Cin,      K = R*S*Cout   A gathers dy at (hi + pad - r) / stride when divisible, B = w^T
//   wgrad   : M = Cout,    Ncol = R*S*Cin,  K = N*Ho*Wo    A = dy^T, B gathers x; split-K over blockIdx.z into float32
//             partial tiles, summed in a fixed order by a second kernel (deterministic, no atomics)
#include "dir_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float cf_f32x16;
constexpr int CF_BM = 64, CF_BN = 64, CF_BK = 16, CF_LD = 68;

struct ConvF32P {
    const float* a;            // forward: x,  dgrad: dy,  wgrad: dy
    const float* b;            // forward: w,  dgrad: w,   wgrad: x
    float* out;                // forward: y,  dgrad: dx,  wgrad: partial tiles [splits][M][Ncol] (or dw when splits == 1)
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
    int M, Ncol, K;
    int klen;                  // K range per blockIdx.z (multiple of CF_BK)
    // fused store epilogue of the data gradient (same semantics as dir_conv_fwd_fused / dir_conv_dgrad_join of the bf16 path):
    const float* addend;       // [M][Ncol] added to the result (gradient accumulation of a fan-out)
    const float* addend2;      // COMPACT [N][H/2][W/2][Ncol], added at the even (h, w) pixels only (stride-2 1x1 sibling)
    const float* mask;         // [M][Ncol]: result zeroed where !(mask > 0) (ReLU backward of the tensor the gradient belongs to)
};

// 8 MFMA 32x32x2 on one staged K-step. A fragment: lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31].
__device__ __forceinline__ void cf_mfma_step(const float* __restrict__ As, const float* __restrict__ Bs, int wm, int wn, int lane,
                                             cf_f32x16& acc) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int kk = 0; kk < CF_BK / 2; ++kk) {
        const float a = As[(2 * kk + half) * CF_LD + wm * 32 + col];
        const float b = Bs[(2 * kk + half) * CF_LD + wn * 32 + col];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
__device__ __forceinline__ void cf_store_tile(const cf_f32x16& acc, float* __restrict__ out, int m0, int n0, int wm, int wn, int lane,
                                              int M, int Ncol) {
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= Ncol) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M) out[(size_t)row * Ncol + col] = acc[e];
    }
}

// Data-gradient store with the fused epilogue: rows are pixels (n, h, w) of dx [N][H][W][Ncol].
__device__ __forceinline__ void cf_store_tile_fused(const cf_f32x16& acc, const ConvF32P& p, int m0, int n0, int wm, int wn, int lane) {
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= p.Ncol) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row >= p.M) continue;
        const size_t o = (size_t)row * p.Ncol + col;
        float v = acc[e];
        if (p.addend) v += p.addend[o];
        if (p.addend2) {
            const int w = row % p.W, q = row / p.W, h = q % p.H, n = q / p.H;
            if (!((h | w) & 1)) v += p.addend2[(((size_t)n * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.Ncol + col];
        }
        if (p.mask && !(p.mask[o] > 0.0f)) v = 0.0f;
        p.out[o] = v;
    }
}

// MODE 0 = forward, 1 = data gradient. Loader: thread t fetches k = k0 + (t & 15) of rows / columns (t >> 4) + 16 i.
template <int MODE>
__global__ void __launch_bounds__(DIR_TPB) conv_f32_kfast_kernel(ConvF32P p) {
    __shared__ float As[CF_BK * CF_LD];
    __shared__ float Bs[CF_BK * CF_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * CF_BM, n0 = blockIdx.y * CF_BN;
    const int kq = t & 15, rq = t >> 4;
    // rows of the A operand owned by this thread: (image, y, x) of the output pixel (forward) / input pixel (dgrad)
    int r_n[4], r_y[4], r_x[4];
    bool r_ok[4];
    const int PH = MODE == 0 ? p.Ho : p.H, PW = MODE == 0 ? p.Wo : p.W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + rq + 16 * i;
        r_ok[i] = m < p.M;
        const int mm = r_ok[i] ? m : 0;
        r_x[i] = mm % PW; const int q = mm / PW; r_y[i] = q % PH; r_n[i] = q / PH;
    }
    const int CK = MODE == 0 ? p.Cin : p.Cout;                  // channel extent of the K axis (innermost of k)
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
        const int k = k0 + kq;
        const bool kok = k < p.K;
        const int kk = kok ? k : 0;
        const int c = kk % CK, tap = kk / CK, r = tap / p.S, s = tap - r * p.S;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = 0.0f;
            if (kok && r_ok[i]) {
                if (MODE == 0) {
                    const int hi = r_y[i] * p.stride - p.pad + r, wi = r_x[i] * p.stride - p.pad + s;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        v = p.a[(((size_t)r_n[i] * p.H + hi) * p.W + wi) * p.Cin + c];
                } else {
                    const int th = r_y[i] + p.pad - r, tw = r_x[i] + p.pad - s;
                    if (th >= 0 && tw >= 0) {
                        const int ho = th / p.stride, wo = tw / p.stride;
                        if (ho * p.stride == th && wo * p.stride == tw && ho < p.Ho && wo < p.Wo)
                            v = p.a[(((size_t)r_n[i] * p.Ho + ho) * p.Wo + wo) * p.Cout + c];
                    }
                }
            }
            ra[i] = v;
            const int n = n0 + rq + 16 * i;
            float w = 0.0f;
            if (kok && n < p.Ncol) {
                if (MODE == 0) w = p.b[(size_t)n * p.K + kk];                                        // w[co = n][r][s][ci]
                else w = p.b[(((size_t)c * p.R + r) * p.S + s) * p.Cin + n];                         // w[co = c][r][s][ci = n]
            }
            rb[i] = w;
        }
    };
    cf_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    fetch(0);
    for (int k0 = 0; k0 < p.K; k0 += CF_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[kq * CF_LD + rq + 16 * i] = ra[i]; Bs[kq * CF_LD + rq + 16 * i] = rb[i]; }
        __syncthreads();
        if (k0 + CF_BK < p.K) fetch(k0 + CF_BK);
        cf_mfma_step(As, Bs, wm, wn, lane, acc);
        __syncthreads();
    }
    if (MODE == 1 && (p.addend || p.addend2 || p.mask)) cf_store_tile_fused(acc, p, m0, n0, wm, wn, lane);
    else cf_store_tile(acc, p.out, m0, n0, wm, wn, lane, p.M, p.Ncol);
}

// Weight gradient. Loader: thread t fetches column c = t & 63 of the A tile (output channel m0 + c) and of the B tile
// (weight element n0 + c = (r, s, ci)) for the four pixels k = k0 + (t >> 6) + 4 i: both reads are contiguous along c.
__global__ void __launch_bounds__(DIR_TPB) conv_f32_wgrad_kernel(ConvF32P p) {
    __shared__ float As[CF_BK * CF_LD];
    __shared__ float Bs[CF_BK * CF_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * CF_BM, n0 = blockIdx.y * CF_BN;
    const int c = t & 63, kq = t >> 6;
    const int kbeg = blockIdx.z * p.klen;
    const int kend = (kbeg + p.klen < p.K) ? kbeg + p.klen : p.K;
    const int co = m0 + c;
    const bool co_ok = co < p.M;
    const int n = n0 + c;
    const bool n_ok = n < p.Ncol;
    const int nn = n_ok ? n : 0;
    const int ci = nn % p.Cin, tap = nn / p.Cin, r = tap / p.S, s = tap - r * p.S;
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kq + 4 * i;
            float a = 0.0f, b = 0.0f;
            if (k < kend) {
                if (co_ok) a = p.a[(size_t)k * p.Cout + co];
                if (n_ok) {
                    const int wo = k % p.Wo, q = k / p.Wo, ho = q % p.Ho, img = q / p.Ho;
                    const int hi = ho * p.stride - p.pad + r, wi = wo * p.stride - p.pad + s;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        b = p.b[(((size_t)img * p.H + hi) * p.W + wi) * p.Cin + ci];
                }
            }
            ra[i] = a; rb[i] = b;
        }
    };
    cf_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += CF_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[(kq + 4 * i) * CF_LD + c] = ra[i]; Bs[(kq + 4 * i) * CF_LD + c] = rb[i]; }
        __syncthreads();
        if (k0 + CF_BK < kend) fetch(k0 + CF_BK);
        cf_mfma_step(As, Bs, wm, wn, lane, acc);
        __syncthreads();
    }
    cf_store_tile(acc, p.out + (size_t)blockIdx.z * p.M * p.Ncol, m0, n0, wm, wn, lane, p.M, p.Ncol);
}

__global__ void __launch_bounds__(DIR_TPB) conv_f32_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                        size_t n, int splits) {
    for (size_t i = (size_t)blockIdx.x * DIR_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * DIR_TPB) {
        float s = part[i];
        for (int z = 1; z < splits; ++z) s += part[(size_t)z * n + i];          // fixed order: bit-reproducible
        dw[i] = s;
    }
}

int cf_check(const void* a, const void* b, const void* out, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
             int* Ho, int* Wo) {
    DIR_RETURN_IF(!a || !b || !out, DIR_EINVAL);
    DIR_RETURN_IF(N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0, DIR_EINVAL);
    *Ho = (H + 2 * pad - R) / stride + 1; *Wo = (W + 2 * pad - S) / stride + 1;
    DIR_RETURN_IF(*Ho <= 0 || *Wo <= 0, DIR_EINVAL);
    const long long lim = 1ll << 31;
    DIR_RETURN_IF((long long)N * H * W >= lim || (long long)N * *Ho * *Wo >= lim || (long long)R * S * Cin >= lim ||
                  (long long)R * S * Cout >= lim, DIR_EUNSUPPORTED);
    return DIR_OK;
}

int cf_wgrad_splits(int M, int Ncol, int K, int* klen) {
    const long long tiles = (long long)dir_cdiv(M, CF_BM) * dir_cdiv(Ncol, CF_BN);
    long long splits = (2048 + tiles - 1) / tiles;                              // ~8 workgroups per CU
    const long long max_splits = (K + 16 * CF_BK - 1) / (16 * CF_BK);          // >= 16 K-steps per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long len = (K + splits - 1) / splits;
    len = (len + CF_BK - 1) / CF_BK * CF_BK;
    *klen = (int)len;
    return (int)((K + len - 1) / len);
}

}  // namespace

extern "C" int dir_conv_f32_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                                int stride, int pad, dir_stream_t stream) {
    int Ho, Wo;
    const int rc = cf_check(x, w, y, N, H, W, Cin, Cout, R, S, stride, pad, &Ho, &Wo);
    if (rc != DIR_OK) return rc;
    ConvF32P p;
    p.a = x; p.b = w; p.out = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = N * Ho * Wo; p.Ncol = Cout; p.K = R * S * Cin; p.klen = p.K;
    p.addend = p.addend2 = p.mask = nullptr;
    DIR_RETURN_IF(dir_cdiv(p.Ncol, CF_BN) > 65535, DIR_EUNSUPPORTED);
    hipLaunchKernelGGL((conv_f32_kfast_kernel<0>), dim3(dir_cdiv(p.M, CF_BM), dir_cdiv(p.Ncol, CF_BN)), dim3(DIR_TPB), 0, dir_s(stream), p);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_conv_f32_dgrad_fused(const float* dy, const float* w, const float* addend, const float* addend_s2,
                                        const float* relu_mask, float* dx, int N, int H, int W, int Cin, int Cout, int R, int S,
                                        int stride, int pad, dir_stream_t stream) {
    int Ho, Wo;
    const int rc = cf_check(dy, w, dx, N, H, W, Cin, Cout, R, S, stride, pad, &Ho, &Wo);
    if (rc != DIR_OK) return rc;
    DIR_RETURN_IF(addend_s2 && ((H | W) & 1), DIR_EUNSUPPORTED);
    ConvF32P p;
    p.a = dy; p.b = w; p.out = dx;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = N * H * W; p.Ncol = Cin; p.K = R * S * Cout; p.klen = p.K;
    p.addend = addend; p.addend2 = addend_s2; p.mask = relu_mask;
    DIR_RETURN_IF(dir_cdiv(p.Ncol, CF_BN) > 65535, DIR_EUNSUPPORTED);
    hipLaunchKernelGGL((conv_f32_kfast_kernel<1>), dim3(dir_cdiv(p.M, CF_BM), dir_cdiv(p.Ncol, CF_BN)), dim3(DIR_TPB), 0, dir_s(stream), p);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_conv_f32_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int R, int S,
                                  int stride, int pad, dir_stream_t stream) {
    return dir_conv_f32_dgrad_fused(dy, w, nullptr, nullptr, nullptr, dx, N, H, W, Cin, Cout, R, S, stride, pad, stream);
}

extern "C" size_t dir_conv_f32_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return 0;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || (long long)N * Ho * Wo >= (1ll << 31)) return 0;
    int klen;
    const int splits = cf_wgrad_splits(Cout, R * S * Cin, N * Ho * Wo, &klen);
    return dir_align_up((size_t)splits * Cout * R * S * Cin * sizeof(float), 256);
}

extern "C" int dir_conv_f32_wgrad(const float* dy, const float* x, float* dw, int N, int H, int W, int Cin, int Cout, int R, int S,
                                  int stride, int pad, void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    int Ho, Wo;
    const int rc = cf_check(dy, x, dw, N, H, W, Cin, Cout, R, S, stride, pad, &Ho, &Wo);
    if (rc != DIR_OK) return rc;
    ConvF32P p;
    p.a = dy; p.b = x;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = Cout; p.Ncol = R * S * Cin; p.K = N * Ho * Wo;
    p.addend = p.addend2 = p.mask = nullptr;
    const int splits = cf_wgrad_splits(p.M, p.Ncol, p.K, &p.klen);
    const size_t need = dir_conv_f32_wgrad_workspace(N, H, W, Cin, Cout, R, S, stride, pad);
    DIR_RETURN_IF(splits > 1 && (!workspace || workspace_bytes < need), DIR_EWORKSPACE);
    DIR_RETURN_IF(dir_cdiv(p.Ncol, CF_BN) > 65535 || splits > 65535, DIR_EUNSUPPORTED);
    p.out = splits > 1 ? static_cast<float*>(workspace) : dw;
    hipLaunchKernelGGL(conv_f32_wgrad_kernel, dim3(dir_cdiv(p.M, CF_BM), dir_cdiv(p.Ncol, CF_BN), splits), dim3(DIR_TPB), 0, dir_s(stream), p);
    DIR_LAUNCH_CHECK();
    if (splits > 1) {
        const size_t n = (size_t)p.M * p.Ncol;
        int grid = dir_cdiv((long long)n, DIR_TPB); if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(conv_f32_wgrad_reduce_kernel, dim3(grid), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const float*>(workspace), dw, n, splits);
        DIR_LAUNCH_CHECK();
    }
    return DIR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// float32 pools of the parity mode (resnet.py:82,131 MaxPool2d(3, 2, 1); :85,136 AvgPool2d(7) on the 7x7 map), NHWC.
namespace {
__global__ void __launch_bounds__(DIR_TPB)
maxpool_f32_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int c = (int)(i % C); long long q = i / C;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho); const int n = (int)(q / Ho);
        float best = -INFINITY; int bi = 0;
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float v = x[(((size_t)n * H + hi) * W + wi) * C + c];
                if (v > best || v != v) { best = v; bi = r * 3 + s; }            // first maximum wins, NaN propagates (torch)
            }
        }
        y[i] = best; idx[i] = (uint8_t)bi;
    }
}

__global__ void __launch_bounds__(DIR_TPB)
maxpool_f32_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int c = (int)(i % C); long long q = i / C;
        const int w = (int)(q % W); q /= W;
        const int h = (int)(q % H); const int n = (int)(q / H);
        float acc = 0.0f;
        for (int a = 0; a < 2; ++a) {
            const int ho = (h + 1) / 2 - a, r = h - (2 * ho - 1);
            if (ho < 0 || ho >= Ho || r < 0 || r > 2) continue;
            for (int b = 0; b < 2; ++b) {
                const int wo = (w + 1) / 2 - b, s = w - (2 * wo - 1);
                if (wo < 0 || wo >= Wo || s < 0 || s > 2) continue;
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
                if (idx[o] == (uint8_t)(r * 3 + s)) acc += dy[o];
            }
        }
        dx[i] = acc;
    }
}

// y[n][c] = (sum over the HW pixels in order) / HW, like AvgPool2d's sequential window sum followed by one division
__global__ void __launch_bounds__(DIR_TPB)
avgpool_f32_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const long long total = (long long)N * C;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int c = (int)(i % C); const int n = (int)(i / C);
        float s = 0.0f;
        for (int q = 0; q < HW; ++q) s += x[((size_t)n * HW + q) * C + c];
        y[i] = s / (float)HW;
    }
}

__global__ void __launch_bounds__(DIR_TPB)
avgpool_f32_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int HW, int C) {
    const long long total = (long long)N * HW * C;
    for (long long i = (long long)blockIdx.x * DIR_TPB + threadIdx.x; i < total; i += (long long)gridDim.x * DIR_TPB) {
        const int c = (int)(i % C); const int n = (int)(i / ((long long)HW * C));
        dx[i] = dy[(size_t)n * C + c] / (float)HW;
    }
}
}  // namespace

static int cf_grid(long long total) { long long g = (total + DIR_TPB - 1) / DIR_TPB; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

extern "C" int dir_maxpool3x3s2_f32_fwd(const float* x, float* y, void* argmax, int N, int H, int W, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0, DIR_EINVAL);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_f32_fwd_kernel, dim3(cf_grid((long long)N * Ho * Wo * C)), dim3(DIR_TPB), 0, dir_s(stream), x, y,
                       static_cast<uint8_t*>(argmax), N, H, W, C, Ho, Wo);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_maxpool3x3s2_f32_bwd(const float* dy, const void* argmax, float* dx, int N, int H, int W, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !dx || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0, DIR_EINVAL);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_f32_bwd_kernel, dim3(cf_grid((long long)N * H * W * C)), dim3(DIR_TPB), 0, dir_s(stream), dy,
                       static_cast<const uint8_t*>(argmax), dx, N, H, W, C, Ho, Wo);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_avgpool_f32_fwd(const float* x, float* y, int N, int HW, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!x || !y || N <= 0 || HW <= 0 || C <= 0, DIR_EINVAL);
    hipLaunchKernelGGL(avgpool_f32_fwd_kernel, dim3(cf_grid((long long)N * C)), dim3(DIR_TPB), 0, dir_s(stream), x, y, N, HW, C);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_avgpool_f32_bwd(const float* dy, float* dx, int N, int HW, int C, dir_stream_t stream) {
    DIR_RETURN_IF(!dy || !dx || N <= 0 || HW <= 0 || C <= 0, DIR_EINVAL);
    hipLaunchKernelGGL(avgpool_f32_bwd_kernel, dim3(cf_grid((long long)N * HW * C)), dim3(DIR_TPB), 0, dir_s(stream), dy, dx, N, HW, C);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
