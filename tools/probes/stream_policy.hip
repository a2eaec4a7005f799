// Streaming-policy probe (measurement tool, not product code): x = x * a + b in place (and out of place) over a buffer far larger
// than the caches, with plain / non-temporal loads and stores and two work distributions. Prints TB/s moved (read + write).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_policy tools/probes/stream_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f4;
template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// chunked: a workgroup owns TPB * U consecutive 16-byte pieces (torch's elementwise layout)
template <bool NTL, bool NTS, int U>
__global__ void __launch_bounds__(256) k_chunk(const f4* __restrict__ x, f4* __restrict__ y, size_t n4, float a, float b) {
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256; if (i < n4) v[u] = ld<NTL>(x + i); }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256; if (i < n4) st<NTS>(y + i, v[u] * a + b); }
}
// grid-stride: G workgroups sweep the buffer, U pieces in flight per lane
template <bool NTL, bool NTS, int U>
__global__ void __launch_bounds__(256) k_stride(const f4* __restrict__ x, f4* __restrict__ y, size_t n4, float a, float b) {
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += step * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t i = i0 + u * step; if (i < n4) v[u] = ld<NTL>(x + i); }
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t i = i0 + u * step; if (i < n4) st<NTS>(y + i, v[u] * a + b); }
    }
}
// read-only sum (no writes) and write-only fill, for the two directions alone
template <bool NTL>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ x, float* __restrict__ out, size_t n4) {
    const size_t step = (size_t)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += step * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t i = i0 + u * step; if (i < n4) acc += ld<NTL>(x + i); }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}
template <bool NTS>
__global__ void __launch_bounds__(256) k_fill(f4* __restrict__ y, size_t n4) {
    const size_t step = (size_t)gridDim.x * 256;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) st<NTS>(y + i, v);
}

static hipEvent_t e0, e1;
template <class F> static double timeit(F launch, int reps = 10) {
    launch(); launch();
    hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    const size_t bytes = (size_t)512 << 20, n4 = bytes / 16;          // 65 536 x 2048 float32
    f4 *x, *y; float* out;
    hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&out, 64);
    hipMemset(x, 0, bytes); hipMemset(y, 0, bytes);
    hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(name, moved, ...) { const double s = timeit([&] { __VA_ARGS__; }); printf("%-64s %8.1f us  %6.2f TB/s\n", name, s * 1e6, (moved) / s * 1e-12); }
#define CH(NTL, NTS, U, dst, tag) RUN("chunk  U=" #U " ld=" #NTL " st=" #NTS " " tag, 2.0 * bytes, \
        hipLaunchKernelGGL((k_chunk<NTL, NTS, U>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, x, dst, n4, 1.0001f, 0.5f))
#define ST(NTL, NTS, U, G, dst, tag) RUN("stride U=" #U " G=" #G " ld=" #NTL " st=" #NTS " " tag, 2.0 * bytes, \
        hipLaunchKernelGGL((k_stride<NTL, NTS, U>), dim3(G), dim3(256), 0, 0, x, dst, n4, 1.0001f, 0.5f))
    printf("in place (y == x), 512 MiB buffer, 1.07 GB moved; ld/st: 1 = non-temporal\n");
    CH(0, 0, 4, x, "in-place"); CH(1, 0, 4, x, "in-place"); CH(0, 1, 4, x, "in-place"); CH(1, 1, 4, x, "in-place");
    CH(0, 0, 8, x, "in-place"); CH(1, 1, 8, x, "in-place"); CH(0, 0, 2, x, "in-place"); CH(0, 0, 1, x, "in-place");
    ST(0, 0, 4, 2048, x, "in-place"); ST(1, 0, 4, 2048, x, "in-place"); ST(0, 1, 4, 2048, x, "in-place"); ST(1, 1, 4, 2048, x, "in-place");
    ST(0, 0, 4, 1024, x, "in-place"); ST(0, 0, 4, 4096, x, "in-place"); ST(0, 0, 8, 2048, x, "in-place"); ST(0, 0, 2, 4096, x, "in-place");
    ST(0, 0, 4, 8192, x, "in-place"); ST(1, 1, 4, 8192, x, "in-place");
    printf("out of place (y != x)\n");
    CH(0, 0, 4, y, "out-of-place"); CH(1, 0, 4, y, "out-of-place"); CH(0, 1, 4, y, "out-of-place"); CH(1, 1, 4, y, "out-of-place");
    ST(0, 0, 4, 2048, y, "out-of-place"); ST(1, 1, 4, 2048, y, "out-of-place"); ST(0, 1, 4, 2048, y, "out-of-place");
    printf("one direction alone\n");
    RUN("read-only plain G=2048", 1.0 * bytes, hipLaunchKernelGGL((k_read<false>), dim3(2048), dim3(256), 0, 0, x, out, n4));
    RUN("read-only nt    G=2048", 1.0 * bytes, hipLaunchKernelGGL((k_read<true>), dim3(2048), dim3(256), 0, 0, x, out, n4));
    RUN("fill plain G=2048", 1.0 * bytes, hipLaunchKernelGGL((k_fill<false>), dim3(2048), dim3(256), 0, 0, y, n4));
    RUN("fill nt    G=2048", 1.0 * bytes, hipLaunchKernelGGL((k_fill<true>), dim3(2048), dim3(256), 0, 0, y, n4));
    return 0;
}
