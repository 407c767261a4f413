// Diagnostic: what one gfx950 SIMD sustains on the FFT codelets themselves (registers only, no LDS, no memory):
// clocks per codelet call with 1, 2, 3, 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o tools/codelet_probe tools/codelet_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../mfcc_amd/csrc/codelets_gen.hpp"
using mfcc_codelets::v2f;

template <int WHICH>
__global__ void probe(const float *in, float *out, unsigned long long *clk, int iters) {
    v2f a[16], b[16], c[16], z[16];
    float y = 0.f;
    const int l = threadIdx.x;
    for (int i = 0; i < 16; ++i) {
        a[i] = (v2f){in[l + i], in[l + 16 + i]};
        b[i] = (v2f){in[l + 32 + i], in[l + 48 + i]};
        c[i] = (v2f){in[l + 64 + i], in[l + 80 + i]};
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (WHICH == 0) mfcc_codelets::rfft32_tw(a, b, c, z, y);
        if constexpr (WHICH == 1) mfcc_codelets::cfft32_h0(a, b, z);
        if constexpr (WHICH == 2) mfcc_codelets::cfft32_h1(a, b, z);
        if constexpr (WHICH == 3) mfcc_codelets::cfft16(a, z);
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = z[i] * 0.5f;      // feed back (16 packed muls; keeps the values finite)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = y;
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) clk[blockIdx.x] = t1 - t0;
}

template <int WHICH>
void run(const char *name, int ops) {
    float *in, *out;
    unsigned long long *clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&clk, 4096 * 8);
    hipMemset(in, 0, 4096 * 4);
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
        const int threads = 64 * 4 * waves_per_simd;             // one workgroup per CU, waves_per_simd on every SIMD
        hipLaunchKernelGGL(probe<WHICH>, dim3(256), dim3(threads), 0, 0, in, out, clk, iters);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<WHICH>, dim3(256), dim3(threads), 0, 0, in, out, clk, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256];
        hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < 256; ++i) avg += double(h[i]);
        avg /= 256.0 * iters;
        const double tflops = 256.0 * 4 * waves_per_simd * iters * (ops + 16) * 256.0 / (ms * 1e-3) / 1e12;
        printf("%-10s %d wave(s)/SIMD: %7.1f ticks per call per wave  -> %6.1f per call per SIMD, %.2f ticks per packed op (%d + 16); wall %.3f ms = %.1f TFLOP/s, %.2f GHz if ticks were clocks\n",
               name, waves_per_simd, avg, avg / waves_per_simd, avg / waves_per_simd / (ops + 16), ops, ms, tflops, avg * iters / (ms * 1e-3) / 1e9);
    }
}

int main() {
    run<0>("rfft32_tw", 158);
    run<1>("cfft32_h0", 90);
    run<2>("cfft32_h1", 119);
    run<3>("cfft16", 74);
    return 0;
}
