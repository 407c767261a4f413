// occ_probe: how many workgroups of a given shape (waves, VGPRs, LDS) does a gfx950 CU keep resident?  (diagnostic)
// Every workgroup spins for a fixed time; 2 x 256 of them finish in ~1 spin if two fit per CU, in ~2 spins if one does.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int THREADS, int VG, int LDS_BYTES>
__global__ __launch_bounds__(THREADS) void spin(unsigned long long ticks, int *sink) {
    __shared__ char lds[LDS_BYTES];
    lds[threadIdx.x] = (char)threadIdx.x;
    if constexpr (VG == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if constexpr (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (VG == 248) asm volatile("v_mov_b32 v247, 0" ::: "v247");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) % THREADS] == 77 && ticks == 1) *sink = 1;
}

template <int THREADS, int VG, int LDS_BYTES>
void run(const char *name, int wgs, int *sink) {
    const unsigned long long ticks = 100 * 100;          // 100 us at 100 MHz
    hipLaunchKernelGGL((spin<THREADS, VG, LDS_BYTES>), dim3(wgs), dim3(THREADS), 0, 0, ticks, sink);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((spin<THREADS, VG, LDS_BYTES>), dim3(wgs), dim3(THREADS), 0, 0, ticks, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %4d workgroups: %.3f ms  -> %.1f resident per CU\n", name, wgs, ms, wgs / 256.0 / (ms / 0.1));
}

int main() {
    int *sink; (void)hipMalloc(&sink, 4);
    run<256, 248, 57000>("4 waves, 248 VGPR, 57 KB (today)", 512, sink);
    run<256, 168, 53000>("4 waves, 168 VGPR, 53 KB", 768, sink);
    run<384, 168, 57000>("6 waves, 168 VGPR, 57 KB", 512, sink);
    run<384, 168, 75000>("6 waves, 168 VGPR, 75 KB", 512, sink);
    run<384, 248, 57000>("6 waves, 248 VGPR, 57 KB", 512, sink);
    run<512, 128, 75000>("8 waves, 128 VGPR, 75 KB", 512, sink);
    run<768, 168, 150000>("12 waves, 168 VGPR, 150 KB", 256, sink);
    run<320, 168, 57000>("5 waves, 168 VGPR, 57 KB", 512, sink);
    return 0;
}
