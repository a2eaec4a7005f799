// Fused network tail: global 7x7 average pool -> FDS calibration -> Linear(2048, 1), and its backward.
// Replaces, per training step, imdb-wiki-dir/resnet.py:136-148 (AvgPool2d(7) + view, FDS.smooth, Linear — rocBLAS gemv kernels
// in round 1: 72 + 15 + 15 us per step, ten times the rest of the FDS + loss tail) and their autograd:
//
//   forward  : ONE launch.  One workgroup per sample: 256 threads x 8 channels sweep the [HW, C] map (16-B loads, 4 KB per
//              pixel row per instruction), keep the float32 means in registers, look the sample's FDS bucket up (label scan
//              for the presence flags of SURVEY A.3 like dir_fds_smooth_fwd), calibrate (x - m1) * s + m2 in registers, store
//              the calibrated encoding [B, C] once (it is a model output, A.2, and the epoch tail's statistics input) and
//              reduce sum_c enc[c] * W[c] with wavefront shuffles + 4 LDS words -> pred[b] = dot + bias.
//   backward : dx[b, hw, c] = ((dpred[b] * W[c] (+ dencoding[b, c])) * s[bin_b, c]) / HW in ONE launch (the [B, C] float32
//              gradient of the encoding is never materialised), and dW / dbias by a fixed-order two-stage reduction over the
//              batch (deterministic, no atomics).
// The weighted loss stays its own single kernel (dir_weighted_loss: value + gradient in one launch): the reference's API
// computes it outside the model (train.py:255) and its mean over the batch would need a grid-wide reduction here.
// The float32 arithmetic is written out exactly as the unfused kernels do it (-ffp-contract=off), so encoding, dx and
// the data gradients are bit-identical to the pool -> smooth -> linear chain they replace.
#include "dir_common.h"

namespace {

__device__ __forceinline__ float tl_calib1(float x, float m1, float s, float m2) {
    return (s < 0.0f) ? x : (x - m1) * s + m2;               // utils.py:107; s < 0 = "leave untouched" (dir_fds_prepare_scale)
}

template <typename T> struct TailIO;
template <> struct TailIO<uint16_t> {                         // bf16 NHWC map
    static __device__ __forceinline__ void load8(const uint16_t* p, float (&v)[8]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store8(uint16_t* p, const float (&v)[8]) {
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x2_t t = {v[2 * q], v[2 * q + 1]}; w[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2_t)); }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ float mean(float sum, int hw) { return sum * (1.0f / (float)hw); }     // = dir_avgpool_fwd
    static __device__ __forceinline__ float spread(float g, int hw) { return g * (1.0f / (float)hw); }      // = dir_avgpool_bwd
};
template <> struct TailIO<float> {                            // float32 NHWC map (parity mode)
    static __device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    static __device__ __forceinline__ float mean(float sum, int hw) { return sum / (float)hw; }             // = dir_avgpool_f32_fwd
    static __device__ __forceinline__ float spread(float g, int hw) { return g / (float)hw; }               // = dir_avgpool_f32_bwd
};

template <typename T>
__global__ void __launch_bounds__(DIR_TPB)
tail_fwd_kernel(const T* __restrict__ x, const float* __restrict__ labels, const int32_t* __restrict__ bins_in, int B, int HW, int C,
                float lo, float hi, const float* __restrict__ m1, const float* __restrict__ scale, const float* __restrict__ m2,
                const float* __restrict__ weight, const float* __restrict__ bias, float* __restrict__ enc, float* __restrict__ pred,
                int32_t* __restrict__ bins_out) {
    __shared__ float red[4];
    const int row = blockIdx.x, t = threadIdx.x;
    int bin = -1;
    const bool calibrate = m1 != nullptr;
    if (calibrate) {
        if (bins_in) {
            bin = bins_in[row];
        } else {
            int has_lo = 0, has_hi = 0;                                        // presence of the boundary labels in THIS batch (A.3)
            for (int i = t; i < B; i += DIR_TPB) { const float l = labels[i]; has_lo |= (l == lo); has_hi |= (l == hi); }
            has_lo = __syncthreads_or(has_lo);
            has_hi = __syncthreads_or(has_hi);
            bin = dir_bin_of(labels[row], lo, hi, has_lo != 0, has_hi != 0);
        }
        if (t == 0 && bins_out) bins_out[row] = bin;
    }
    float dot = 0.0f;
    const int groups = C >> 3;
    for (int g = t; g < groups; g += DIR_TPB) {
        const T* p = x + (size_t)row * HW * C + g * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 7
        for (int k = 0; k < HW; ++k) {
            float v[8];
            TailIO<T>::load8(p + (size_t)k * C, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = TailIO<T>::mean(acc[j], HW);
        if (bin >= 0) {
            const size_t to = (size_t)bin * C + g * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = tl_calib1(e[j], m1[to + j], scale[to + j], m2[to + j]);
        }
        float* o = enc + (size_t)row * C + g * 8;
        *reinterpret_cast<float4*>(o) = make_float4(e[0], e[1], e[2], e[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(e[4], e[5], e[6], e[7]);
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += e[j] * weight[g * 8 + j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, DIR_WAVE);
    if ((t & 63) == 0) red[t >> 6] = dot;
    __syncthreads();
    if (t == 0) pred[row] = ((red[0] + red[1]) + (red[2] + red[3])) + bias[0];
}

template <typename T>
__global__ void __launch_bounds__(DIR_TPB)
tail_bwd_dx_kernel(const float* __restrict__ dpred, const float* __restrict__ denc, const int32_t* __restrict__ bins,
                   const float* __restrict__ scale, const float* __restrict__ weight, int HW, int C, T* __restrict__ dx) {
    const int row = blockIdx.x, t = threadIdx.x;
    const float dp = dpred[row];
    const int bin = bins ? bins[row] : -1;
    const int groups = C >> 3;
    for (int g = t; g < groups; g += DIR_TPB) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gr = dp * weight[g * 8 + j];                                 // d encoding_s = dpred * W (Linear backward)
            if (denc) gr += denc[(size_t)row * C + g * 8 + j];                  // + a gradient that reached the returned encoding
            if (bin >= 0) { const float s = scale[(size_t)bin * C + g * 8 + j]; gr = s < 0.0f ? gr : gr * s; }   // calibrate backward
            v[j] = TailIO<T>::spread(gr, HW);                                  // average-pool backward
        }
        T* p = dx + (size_t)row * HW * C + g * 8;
        for (int k = 0; k < HW; ++k) TailIO<T>::store8(p + (size_t)k * C, v);
    }
}

// partial[slice][c] = sum over the rows of the slice (in order) of dpred[b] * enc[b, c]
__global__ void __launch_bounds__(DIR_TPB)
tail_bwd_dw_partial_kernel(const float* __restrict__ dpred, const float* __restrict__ enc, int B, int C, int rows_per_slice,
                           float* __restrict__ partial) {
    const int c = blockIdx.x * DIR_TPB + threadIdx.x;
    if (c >= C) return;
    const int b0 = blockIdx.y * rows_per_slice;
    const int b1 = (b0 + rows_per_slice < B) ? b0 + rows_per_slice : B;
    float s = 0.0f;
    for (int b = b0; b < b1; ++b) s += dpred[b] * enc[(size_t)b * C + c];
    partial[(size_t)blockIdx.y * C + c] = s;
}

__global__ void __launch_bounds__(DIR_TPB)
tail_bwd_dw_final_kernel(const float* __restrict__ partial, int slices, const float* __restrict__ dpred, int B, int C,
                         float* __restrict__ dweight, float* __restrict__ dbias) {
    __shared__ float red[DIR_TPB];
    const int t = threadIdx.x;
    const int c = blockIdx.x * DIR_TPB + t;
    if (c < C) {
        float s = partial[c];
        for (int z = 1; z < slices; ++z) s += partial[(size_t)z * C + c];       // fixed order: bit-reproducible
        dweight[c] = s;
    }
    if (blockIdx.x == 0 && dbias) {
        float s = 0.0f;
        for (int b = t; b < B; b += DIR_TPB) s += dpred[b];
        red[t] = s;
        __syncthreads();
        for (int o = DIR_TPB / 2; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
        if (t == 0) dbias[0] = red[0];
    }
}

int tail_slices(int B, int* rows_per_slice) {
    int rps = (B + 31) / 32; if (rps < 8) rps = 8;
    *rows_per_slice = rps;
    return (B + rps - 1) / rps;
}

}  // namespace

extern "C" size_t dir_tail_bwd_workspace(int B, int C) {
    if (B <= 0 || C <= 0) return 0;
    int rps;
    return dir_align_up((size_t)tail_slices(B, &rps) * C * sizeof(float), 256);
}

extern "C" int dir_tail_fwd(const void* x, int dtype, const float* labels, const int32_t* bins_in, int B, int HW, int C,
                            int bucket_start, int bucket_num, const float* m1, const float* scale, const float* m2,
                            const float* weight, const float* bias, float* encoding, float* pred, int32_t* bins_out,
                            dir_stream_t stream) {
    DIR_RETURN_IF(B < 0 || HW <= 0 || C <= 0, DIR_EINVAL);
    if (B == 0) return DIR_OK;
    DIR_RETURN_IF(!x || !weight || !bias || !encoding || !pred, DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(C % 8 != 0, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(!dir_aligned16(x) || !dir_aligned16(encoding), DIR_EINVAL);
    const bool calibrate = m1 || scale || m2;
    DIR_RETURN_IF(calibrate && (!m1 || !scale || !m2 || (!labels && !bins_in) || !bins_out), DIR_EINVAL);
    DIR_RETURN_IF(calibrate && (bucket_start < 0 || bucket_num <= bucket_start), DIR_EINVAL);
    const float lo = (float)bucket_start, hi = (float)(bucket_num - 1);
    if (dtype == DIR_BF16)
        hipLaunchKernelGGL(tail_fwd_kernel<uint16_t>, dim3(B), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const uint16_t*>(x), labels,
                           bins_in, B, HW, C, lo, hi, m1, scale, m2, weight, bias, encoding, pred, bins_out);
    else
        hipLaunchKernelGGL(tail_fwd_kernel<float>, dim3(B), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const float*>(x), labels,
                           bins_in, B, HW, C, lo, hi, m1, scale, m2, weight, bias, encoding, pred, bins_out);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_tail_bwd(const float* dpred, const float* dencoding, const int32_t* bins, const float* scale, const float* weight,
                            const float* encoding, int B, int HW, int C, int dtype, void* dx, float* dweight, float* dbias,
                            void* workspace, size_t workspace_bytes, dir_stream_t stream) {
    DIR_RETURN_IF(B < 0 || HW <= 0 || C <= 0, DIR_EINVAL);
    if (B == 0) return DIR_OK;
    DIR_RETURN_IF(!dpred || !weight || (bins && !scale), DIR_EINVAL);
    DIR_RETURN_IF(dtype != DIR_F32 && dtype != DIR_BF16, DIR_EUNSUPPORTED);
    DIR_RETURN_IF(C % 8 != 0, DIR_EUNSUPPORTED);
    if (dx) {
        DIR_RETURN_IF(!dir_aligned16(dx), DIR_EINVAL);
        if (dtype == DIR_BF16)
            hipLaunchKernelGGL(tail_bwd_dx_kernel<uint16_t>, dim3(B), dim3(DIR_TPB), 0, dir_s(stream), dpred, dencoding, bins, scale, weight,
                               HW, C, static_cast<uint16_t*>(dx));
        else
            hipLaunchKernelGGL(tail_bwd_dx_kernel<float>, dim3(B), dim3(DIR_TPB), 0, dir_s(stream), dpred, dencoding, bins, scale, weight,
                               HW, C, static_cast<float*>(dx));
        DIR_LAUNCH_CHECK();
    }
    if (dweight) {
        DIR_RETURN_IF(!encoding || !workspace || workspace_bytes < dir_tail_bwd_workspace(B, C), DIR_EWORKSPACE);
        int rps;
        const int slices = tail_slices(B, &rps);
        float* partial = static_cast<float*>(workspace);
        hipLaunchKernelGGL(tail_bwd_dw_partial_kernel, dim3(dir_cdiv(C, DIR_TPB), slices), dim3(DIR_TPB), 0, dir_s(stream), dpred, encoding,
                           B, C, rps, partial);
        DIR_LAUNCH_CHECK();
        hipLaunchKernelGGL(tail_bwd_dw_final_kernel, dim3(dir_cdiv(C, DIR_TPB)), dim3(DIR_TPB), 0, dir_s(stream), partial, slices, dpred, B,
                           C, dweight, dbias);
        DIR_LAUNCH_CHECK();
    }
    return DIR_OK;
}
