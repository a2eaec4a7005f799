// Standalone residency census: how many workgroups of a given shape (wavefronts per workgroup, VGPRs per lane, dynamic LDS) does an MI355X CU
// really run at once? Each workgroup spins for a fixed time; grids of 256 x n workgroups take ceil(n / residency) spin times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/residency_probe.hip -o gpurun_alt/residency_probe && gpurun_alt/residency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WAVES, int VGPRS>
__global__ void __launch_bounds__(64 * WAVES) spin_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ unsigned char smem[];
    if (VGPRS > 128) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    else if (VGPRS > 96) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    else asm volatile("v_mov_b32 v90, 0" ::: "v90");
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t1 = t0;
    while (t1 - t0 < ticks) { __builtin_amdgcn_s_sleep(8); t1 = __builtin_readcyclecounter(); }
    if (sink && threadIdx.x == 0 && ticks == 1) { smem[0] = 1; sink[blockIdx.x] = smem[0]; }
}
template <int WAVES, int VGPRS> static void run(int lds) {
    auto k = spin_kernel<WAVES, VGPRS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k));
    int api = -1; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k, 64 * WAVES, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("waves %d, regs %3d, LDS %6d: API %d |", WAVES, a.numRegs, lds, api);
    const unsigned long long ticks = 40000;                    // shader clocks (~17-20 us)
    for (int n = 1; n <= 4; ++n) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(256 * n), dim3(64 * WAVES), lds, 0, ticks, (int*)nullptr);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("  %4d WGs: %6.1f us", 256 * n, best * 1e3f);
    }
    printf("\n");
}
int main() {
    run<6, 168>(71680); run<6, 168>(32768); run<5, 168>(71680); run<8, 168>(71680); run<4, 168>(71680);
    run<6, 128>(71680); run<6, 128>(49152); run<6, 91>(71680); run<4, 128>(39936); run<4, 128>(71680);
    return 0;
}
