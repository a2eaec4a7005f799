// Standalone: what the runtime says about the residency of the row-resident 1x1 kernel (workgroups per CU) for a range of dynamic-LDS sizes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I imbalanced-regression_amd/csrc tools/probes/rows_occupancy.hip -o gpurun_alt/rows_occupancy && gpurun_alt/rows_occupancy
#include "../../imbalanced-regression_amd/csrc/dir_conv_rows.hip"
#include <cstdio>
template <int KT, bool LEAN> static void one() {
    const void* f = reinterpret_cast<const void*>(conv1x1_rows_kernel<KT, LEAN>);
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, f);
    printf("KT=%d lean=%d: regs %d, static LDS %zu, max threads %d |", KT, (int)LEAN, a.numRegs, a.sharedSizeBytes, a.maxThreadsPerBlock);
    for (int lds : {32768, 49152, 65536, 71680, 81920}) {
        int n = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv1x1_rows_kernel<KT, LEAN>, WR_THREADS, lds);
        printf("  LDS %d -> %d%s", lds, n, e == hipSuccess ? "" : "(err)");
    }
    printf("\n");
}
int main() {
    one<1, true>(); one<2, true>(); one<4, true>(); one<4, false>();
    return 0;
}
