// i8_probe: operand and result layout of v_mfma_i32_16x16x64_i8 on gfx950, and its cost (diagnostic).
//   hipcc --offload-arch=gfx950 -O3 -o tools/i8_probe tools/i8_probe.hip && tools/i8_probe
// A[m][k], B[k][n] are filled with small known values under the ASSUMED layout (lane = row/col + 16 * kgroup, 16
// consecutive k per lane, byte i of the lane's 16 = k = 16 * kgroup + i); D is compared with the plain triple loop
// under the assumed result layout (lane = n + 16 * (m / 4), register m % 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k(const v4i *a, const v4i *b, v4i *d, unsigned long long *cyc) {
    const int lane = threadIdx.x;
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[lane], b[lane], acc, 0, 0, 0);
    d[lane] = acc;
    // cost: 64 back-to-back dependent-free MFMAs on 8 accumulators
    v4i c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = acc;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[lane], b[lane], c[i], 0, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    if (s == 0x7fffffff) d[0][0] = s;
    if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    for (int m = 0; m < 16; ++m) for (int kk = 0; kk < 64; ++kk) A[m * 64 + kk] = (int8_t)(((m * 7 + kk * 3) % 23) - 11);
    for (int kk = 0; kk < 64; ++kk) for (int n = 0; n < 16; ++n) B[kk * 16 + n] = (int8_t)(((kk * 5 + n * 11) % 29) - 14);
    std::vector<int8_t> ha(64 * 16), hb(64 * 16);
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
        ha[l * 16 + i] = A[(l & 15) * 64 + 16 * (l >> 4) + i];
        hb[l * 16 + i] = B[(16 * (l >> 4) + i) * 16 + (l & 15)];
    }
    v4i *da, *db, *dd; unsigned long long *dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024); hipMalloc(&dc, 8);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, dc);
    std::vector<int> hd(256); unsigned long long cyc;
    hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost); hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        int ref = 0;
        for (int kk = 0; kk < 64; ++kk) ref += (int)A[m * 64 + kk] * (int)B[kk * 16 + n];
        const int got = hd[(n + 16 * (m / 4)) * 4 + (m % 4)];
        if (got != ref && bad++ < 8) printf("D[%d][%d] = %d, expected %d\n", m, n, got, ref);
    }
    printf("layout %s (%d mismatches); 64 MFMAs: %llu s_memtime ticks = %.1f per MFMA\n", bad ? "DIFFERENT" : "as assumed", bad, cyc, cyc / 64.0);
    return bad != 0;
}
