// Which shape of a streaming kernel gets the most out of MI355X's HBM?  Copy (1 load + 1 store per 16 B) and a "3 loads + 1 store"
// form (the BatchNorm backward apply), each as: plain grid-stride loop / non-temporal loads, stores or both / software-pipelined
// (the next iteration's loads issued before this iteration's stores: gfx950 counts loads and stores in ONE in-order counter, so a
// load issued after a store cannot be waited for without waiting for the store's acknowledgement) / more bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 -o stream_variants tools/stream/stream_variants.hip && ./stream_variants [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NTL> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NTL ? __builtin_nontemporal_load(p) : *p; }
template <bool NTS> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NTS) __builtin_nontemporal_store(v, p); else *p = v; }

// U independent 16-B accesses per lane and trip, grid-stride over blocks of U * 256 vectors
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) copy_plain(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    const size_t nblk = n / (U * 256);
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const size_t i = b * (U * 256) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NTL>(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(dst + i + u * 256, v[u]);
    }
}
// software-pipelined: loads of trip t + 1 are issued before the stores of trip t
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) copy_pipe(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    const size_t nblk = n / (U * 256);
    size_t b = blockIdx.x;
    if (b >= nblk) return;
    u32x4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NTL>(src + b * (U * 256) + threadIdx.x + u * 256);
    for (;;) {
        const size_t nb = b + gridDim.x;
        const bool more = nb < nblk;
        const size_t ni = (more ? nb : b) * (U * 256) + threadIdx.x;          // (the last trip re-reads its own block: no branch around the loads)
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = ld<NTL>(src + ni + u * 256);
        const size_t i = b * (U * 256) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(dst + i + u * 256, v[u]);
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = w[u];
        b = nb;
    }
}
// 3 loads + 1 store per 16 B (dx = f(dout, x, out))
template <int U, bool NTL, bool NTS, bool PIPE>
__global__ void __launch_bounds__(256) tri(const u32x4* __restrict__ a, const u32x4* __restrict__ bb, const u32x4* __restrict__ c, u32x4* __restrict__ dst, size_t n) {
    const size_t nblk = n / (U * 256);
    size_t b = blockIdx.x;
    if (b >= nblk) return;
    u32x4 va[U], vb[U], vc[U], wa[U], wb[U], wc[U];
    if (!PIPE) {
        for (; b < nblk; b += gridDim.x) {
            const size_t i = b * (U * 256) + threadIdx.x;
#pragma unroll
            for (int u = 0; u < U; ++u) { va[u] = ld<NTL>(a + i + u * 256); vb[u] = ld<NTL>(bb + i + u * 256); vc[u] = ld<NTL>(c + i + u * 256); }
#pragma unroll
            for (int u = 0; u < U; ++u) st<NTS>(dst + i + u * 256, va[u] + vb[u] * vc[u]);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = b * (U * 256) + threadIdx.x + u * 256; va[u] = ld<NTL>(a + i); vb[u] = ld<NTL>(bb + i); vc[u] = ld<NTL>(c + i); }
    for (;;) {
        const size_t nb = b + gridDim.x;
        const bool more = nb < nblk;
        const size_t ni = (more ? nb : b) * (U * 256) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) { wa[u] = ld<NTL>(a + ni + u * 256); wb[u] = ld<NTL>(bb + ni + u * 256); wc[u] = ld<NTL>(c + ni + u * 256); }
        const size_t i = b * (U * 256) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(dst + i + u * 256, va[u] + vb[u] * vc[u]);
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) { va[u] = wa[u]; vb[u] = wb[u]; vc[u] = wc[u]; }
        b = nb;
    }
}

struct Bufs { u32x4 *a, *b, *c, *d; size_t n; };
template <typename F> static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); launch(); CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms / reps < best) best = ms / reps;
    }
    return best;
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 512;
    Bufs B; B.n = mb * 1024 * 1024 / 16;
    CHECK(hipMalloc(&B.a, B.n * 16)); CHECK(hipMalloc(&B.b, B.n * 16)); CHECK(hipMalloc(&B.c, B.n * 16)); CHECK(hipMalloc(&B.d, B.n * 16));
    CHECK(hipMemset(B.a, 1, B.n * 16)); CHECK(hipMemset(B.b, 2, B.n * 16)); CHECK(hipMemset(B.c, 3, B.n * 16));
    const int grids[] = {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32};
    printf("%zu MB per stream; GB/s = bytes moved (loads + stores) / time; columns = workgroups (x256 threads)\n%-34s", mb, "kernel");
    for (int g : grids) printf(" %8d", g);
    printf("\n");
#define ROW(name, bytes_mult, ...)                                                           \
    { printf("%-34s", name);                                                                 \
      for (int g : grids) { const double ms = time_ms([&] { __VA_ARGS__; }, 8); printf(" %8.0f", (bytes_mult) * (double)B.n * 16 / ms / 1e6); } \
      printf("\n"); fflush(stdout); }
#define CP(K, U, L, S) ROW(#K " U=" #U " ntl=" #L " nts=" #S, 2.0, hipLaunchKernelGGL((K<U, L, S>), dim3(g), dim3(256), 0, 0, B.a, B.d, B.n))
    CP(copy_plain, 4, false, false) CP(copy_plain, 4, true, false) CP(copy_plain, 4, false, true) CP(copy_plain, 4, true, true)
    CP(copy_plain, 8, false, false) CP(copy_plain, 8, true, true) CP(copy_plain, 2, true, true)
    CP(copy_pipe, 2, false, false) CP(copy_pipe, 4, false, false) CP(copy_pipe, 4, true, true) CP(copy_pipe, 2, true, true) CP(copy_pipe, 8, true, true)
#define TR(U, L, S, P) ROW("tri U=" #U " ntl=" #L " nts=" #S " pipe=" #P, 4.0, hipLaunchKernelGGL((tri<U, L, S, P>), dim3(g), dim3(256), 0, 0, B.a, B.b, B.c, B.d, B.n))
    TR(2, false, false, false) TR(2, true, true, false) TR(2, false, false, true) TR(2, true, true, true) TR(4, true, true, false) TR(1, true, true, true) TR(4, true, true, true)
    return 0;
}
