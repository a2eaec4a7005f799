// Box calibration probes ("record a measured STREAM-style HBM number and a measured MFMA GEMM peak on the
// box, and report fractions against both nominal and measured"). Not on the hot path: bench.py times these with HIP events
// and prints the results next to the nominal peaks of MI355X_MICROARCH.md.
//   dir_probe_stream_copy : float4 copy, 16 B per lane, grid-stride from 2048 workgroups  -> achievable HBM bytes/s
//   dir_probe_stream_read : same loads, wave-reduced, one store per workgroup             -> achievable HBM read bytes/s
//   dir_probe_stream_write: stores only (16 KB blocks)                                    -> achievable HBM write bytes/s
//   dir_probe_mfma_bf16   : 8 independent v_mfma_f32_32x32x16_bf16 accumulator chains per wavefront, no memory traffic
//   dir_probe_mfma_f32    : the same with v_mfma_f32_32x32x2_f32 (the exact-f32 matrix path of the parity mode)
#include "dir_common.h"
#include "dir_hip_tools.h"

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 pb_bf16x8;
typedef __attribute__((ext_vector_type(16))) float pb_f32x16;

// A workgroup moves 16 KB contiguous blocks (4 x 4 KB wave-rows), blocks dealt round-robin to the workgroups: no
// power-of-two stride between the loads a lane has in flight (a 8 MB stride parks them all on one HBM channel).
// Streaming probes: non-temporal loads and stores (tools/stream/stream_variants.hip: 6.5 TB/s for the copy against 5.9 with plain
// accesses; issue order and depth per lane make no difference) — the best rate a kernel of this shape reaches, i.e. the yardstick.
typedef __attribute__((ext_vector_type(4))) float pb_f32x4;
__device__ __forceinline__ float4 pb_ldnt(const float4* p) { const pb_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pb_f32x4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void pb_stnt(float4* p, float4 v) { __builtin_nontemporal_store(pb_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<pb_f32x4*>(p)); }
__global__ void __launch_bounds__(DIR_TPB) probe_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t nblk = n4 / (4 * DIR_TPB);
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const size_t i = blk * (4 * DIR_TPB) + threadIdx.x;
        const float4 a = pb_ldnt(src + i), b = pb_ldnt(src + i + DIR_TPB), c = pb_ldnt(src + i + 2 * DIR_TPB), d = pb_ldnt(src + i + 3 * DIR_TPB);
        pb_stnt(dst + i, a); pb_stnt(dst + i + DIR_TPB, b); pb_stnt(dst + i + 2 * DIR_TPB, c); pb_stnt(dst + i + 3 * DIR_TPB, d);
    }
    for (size_t i = nblk * (4 * DIR_TPB) + (size_t)blockIdx.x * DIR_TPB + threadIdx.x; i < n4; i += (size_t)gridDim.x * DIR_TPB) dst[i] = src[i];
}

__global__ void __launch_bounds__(DIR_TPB) probe_read_kernel(const float4* __restrict__ src, float* __restrict__ out, size_t n4) {
    const size_t nblk = n4 / (4 * DIR_TPB);
    float acc = 0.0f;
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const size_t i = blk * (4 * DIR_TPB) + threadIdx.x;
        const float4 a = pb_ldnt(src + i), b = pb_ldnt(src + i + DIR_TPB), c = pb_ldnt(src + i + 2 * DIR_TPB), d = pb_ldnt(src + i + 3 * DIR_TPB);
        acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
    }
    for (size_t i = nblk * (4 * DIR_TPB) + (size_t)blockIdx.x * DIR_TPB + threadIdx.x; i < n4; i += (size_t)gridDim.x * DIR_TPB) {
        const float4 v = src[i]; acc += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, DIR_WAVE);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

__global__ void __launch_bounds__(DIR_TPB) probe_write_kernel(float4* __restrict__ dst, size_t n4, float v) {
    const size_t nblk = n4 / (4 * DIR_TPB);
    const float4 x = make_float4(v, v + 1.0f, v + 2.0f, v + 3.0f);
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const size_t i = blk * (4 * DIR_TPB) + threadIdx.x;
        pb_stnt(dst + i, x); pb_stnt(dst + i + DIR_TPB, x); pb_stnt(dst + i + 2 * DIR_TPB, x); pb_stnt(dst + i + 3 * DIR_TPB, x);
    }
}

// L2-resident re-read: every workgroup streams the same `region` bytes (a few MB: resident in each XCD's L2 after the first pass)
// `passes` times, `DEPTH` independent 16-B loads per lane in flight -> what the L2 -> CU path delivers when latency is covered.
template <int DEPTH>
__global__ void __launch_bounds__(DIR_TPB) probe_l2_read_kernel(const float4* __restrict__ src, float* __restrict__ out, size_t n4, int passes) {
    float acc = 0.0f;
    const size_t stride = (size_t)DIR_TPB * DEPTH;
    for (int ps = 0; ps < passes; ++ps) {
        // each workgroup starts at its own offset so that the workgroups of a CU do not walk in lockstep
        size_t base = ((size_t)blockIdx.x * 977 * stride + (size_t)ps * 131 * stride) % n4;
        for (size_t done = 0; done < n4; done += stride) {
            float4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                size_t i = base + (size_t)d * DIR_TPB + threadIdx.x;
                if (i >= n4) i -= n4;
                v[d] = src[i];
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += (v[d].x + v[d].y) + (v[d].z + v[d].w);
            base += stride; if (base >= n4) base -= n4;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, DIR_WAVE);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

template <int CHAINS>
__global__ void __launch_bounds__(DIR_TPB) probe_mfma_bf16_kernel(int iters, float* __restrict__ out) {
    pb_f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
    pb_bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {                      // full-range operands (zero-filled ones clock higher: guide §5.4 rule 25)
        a[e] = (__bf16)(((int)((threadIdx.x * 37u + e * 11u) & 63u) - 32) * 0.03125f);
        b[e] = (__bf16)(((int)((threadIdx.x * 53u + e * 7u + blockIdx.x) & 63u) - 32) * 0.03125f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[(size_t)blockIdx.x * DIR_TPB + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(DIR_TPB) probe_mfma_f32_kernel(int iters, float* __restrict__ out) {
    pb_f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
    const float a = ((int)((threadIdx.x * 37u) & 63u) - 32) * 0.03125f;
    const float b = ((int)((threadIdx.x * 53u + blockIdx.x) & 63u) - 32) * 0.03125f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[(size_t)blockIdx.x * DIR_TPB + threadIdx.x] = s;
}
}  // namespace

extern "C" int dir_probe_stream_copy(const void* src, void* dst, size_t bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!src || !dst || bytes < 16 || (bytes & 15) || !dir_aligned16(src) || !dir_aligned16(dst), DIR_EINVAL);
    hipLaunchKernelGGL(probe_copy_kernel, dim3(4096), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const float4*>(src),
                       static_cast<float4*>(dst), bytes / 16);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_probe_stream_read(const void* src, float* out /* >= 8192 floats */, size_t bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!src || !out || bytes < 16 || (bytes & 15) || !dir_aligned16(src), DIR_EINVAL);
    hipLaunchKernelGGL(probe_read_kernel, dim3(2048), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const float4*>(src), out, bytes / 16);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

extern "C" int dir_probe_stream_write(void* dst, size_t bytes, dir_stream_t stream) {
    DIR_RETURN_IF(!dst || bytes < 16384 || (bytes & 16383) || !dir_aligned16(dst), DIR_EINVAL);
    hipLaunchKernelGGL(probe_write_kernel, dim3(4096), dim3(DIR_TPB), 0, dir_s(stream), static_cast<float4*>(dst), bytes / 16, 1.0f);
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// region_bytes (multiple of 16 KB, e.g. 2-8 MB) re-read `passes` times by each of `workgroups` workgroups, `depth` (4 | 8 | 16)
// loads in flight per lane; bytes moved = workgroups * passes * region_bytes. out: >= workgroups * 4 floats.
extern "C" int dir_probe_l2_read(const void* src, float* out, size_t region_bytes, int workgroups, int passes, int depth, dir_stream_t stream) {
    DIR_RETURN_IF(!src || !out || region_bytes < 65536 || (region_bytes & 16383) || workgroups <= 0 || passes <= 0 || !dir_aligned16(src), DIR_EINVAL);
    const size_t n4 = region_bytes / 16;
    if (depth == 4) hipLaunchKernelGGL(probe_l2_read_kernel<4>, dim3(workgroups), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const float4*>(src), out, n4, passes);
    else if (depth == 8) hipLaunchKernelGGL(probe_l2_read_kernel<8>, dim3(workgroups), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const float4*>(src), out, n4, passes);
    else if (depth == 16) hipLaunchKernelGGL(probe_l2_read_kernel<16>, dim3(workgroups), dim3(DIR_TPB), 0, dir_s(stream), static_cast<const float4*>(src), out, n4, passes);
    else return DIR_EINVAL;
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}

// Returns the FLOPs the launch executes through *flops (host); out: >= workgroups * 256 floats.
extern "C" int dir_probe_mfma_bf16(int workgroups, int iters, float* out, double* flops, dir_stream_t stream) {
    DIR_RETURN_IF(workgroups <= 0 || iters <= 0 || !out, DIR_EINVAL);
    hipLaunchKernelGGL((probe_mfma_bf16_kernel<8>), dim3(workgroups), dim3(DIR_TPB), 0, dir_s(stream), iters, out);
    DIR_LAUNCH_CHECK();
    if (flops) *flops = (double)workgroups * 4.0 * iters * 8.0 * (2.0 * 32 * 32 * 16);
    return DIR_OK;
}

extern "C" int dir_probe_mfma_f32(int workgroups, int iters, float* out, double* flops, dir_stream_t stream) {
    DIR_RETURN_IF(workgroups <= 0 || iters <= 0 || !out, DIR_EINVAL);
    hipLaunchKernelGGL((probe_mfma_f32_kernel<4>), dim3(workgroups), dim3(DIR_TPB), 0, dir_s(stream), iters, out);
    DIR_LAUNCH_CHECK();
    if (flops) *flops = (double)workgroups * 4.0 * iters * 4.0 * (2.0 * 32 * 32 * 2);
    return DIR_OK;
}

// ---- probe of the hardware-transposing LDS read the 3x3 weight gradient is built on (tests/test_hip_conv_wgrad3.py pins the lane
// mapping with it): LDS holds the uint16 ramp 0, 1, 2, ...; lane l of ONE wavefront reads at byte address addr[l]
namespace {
typedef __attribute__((ext_vector_type(4))) __bf16 pb_bf16x4;
__global__ void __launch_bounds__(64) tr16_probe_kernel(const int* __restrict__ addr, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t ramp[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) ramp[i] = (uint16_t)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) pb_bf16x4* lds_v4_t;
    const pb_bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(reinterpret_cast<const unsigned char*>(ramp) + addr[threadIdx.x]));
    const uint64_t bits = __builtin_bit_cast(uint64_t, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(bits >> (16 * j));
}
}  // namespace

extern "C" int dir_probe_tr16(const int* addr_bytes, void* out, dir_stream_t stream) {
    DIR_RETURN_IF(!addr_bytes || !out, DIR_EINVAL);
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, dir_s(stream), addr_bytes, static_cast<uint16_t*>(out));
    DIR_LAUNCH_CHECK();
    return DIR_OK;
}
