// alu_probe: what the fp32 pipes of one gfx950 SIMD sustain, alone and together (diagnostic, not product).
//   hipcc --offload-arch=gfx950 -O3 -o tools/alu_probe tools/alu_probe.hip && tools/alu_probe
// Each block = 8 waves = two per SIMD (wave i and i + 4 share a SIMD).  Waves 0-3 run stream A, waves 4-7 stream B
// (or nothing); every stream is `iters` x 64 independent instructions.  Reported: cycles (s_memtime) per
// instruction of the slower side, so  A||B ~ max(A, B)  means separate pipes and  A||B ~ A + B  a shared one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum Op { NONE = 0, FMA32, PKFMA, PKADD, FMA64, CVT64, MFMA, LOG, DOT2, MFMABF, CVTBF, LDSR, MIX2, MIX4, MIX6, MIX8, MIXB4, MIXF4, MIXF8, MIXD4, MIXBF4, MIXBF8, MIXC4, SALU, SALUDEP, PKSALU, BRANCH };
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int OP>
__device__ __forceinline__ void stream(int iters, float seed, float *sink) {
    if constexpr (OP == NONE) return;
    float a[16]; v2f p[16]; double d[16]; f32x4 acc[8]; int q[16];
    bf16x8 hb = {1, 2, 3, 4, 5, 6, 7, 8};
    int sg[8] = {1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; p[i] = (v2f){seed + i, seed - i}; d[i] = seed + i; q[i] = (int)seed + i; }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    const float c = 1.0000001f; const v2f c2 = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
                if constexpr (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(c2));
                if constexpr (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if constexpr (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"((double)c));
                if constexpr (OP == CVT64) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
                if constexpr (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                if constexpr (OP == DOT2) asm volatile("v_dot2_i32_i16 %0, %1, %1, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
                if constexpr (OP == MFMABF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i & 7]) : "v"(hb));
                if constexpr (OP == CVTBF) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(q[i]) : "v"(a[i]), "v"(a[(i + 1) & 15]));
                if constexpr (OP == MIX2 || OP == MIX4 || OP == MIX6 || OP == MIX8 || OP == MIXB4) {
                    // one MFMA followed by k independent packed FMAs: "instr" = one such group
                    constexpr int k = OP == MIX2 ? 2 : OP == MIX4 ? 4 : OP == MIX6 ? 6 : OP == MIX8 ? 8 : 4;
                    if constexpr (OP == MIXB4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i & 7]) : "v"(hb));
                    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a[i]), "v"(c));
#pragma unroll
                    for (int j = 0; j < k; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[(i + j) & 15]) : "v"(c2));
                }
                if constexpr (OP == MIXF4 || OP == MIXF8 || OP == MIXD4 || OP == MIXBF4 || OP == MIXBF8 || OP == MIXC4) {
                    // one MFMA followed by k independent NON-packed VALU ops (fma32 / dot2 / cvt)
                    constexpr int k = (OP == MIXF8 || OP == MIXBF8) ? 8 : 4;
                    if constexpr (OP == MIXBF4 || OP == MIXBF8) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i & 7]) : "v"(hb));
                    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(p[i].x), "v"(c));
#pragma unroll
                    for (int j = 0; j < k; ++j) {
                        if constexpr (OP == MIXD4) asm volatile("v_dot2_i32_i16 %0, %1, %1, %0" : "+v"(q[(i + j) & 15]) : "v"(q[(i + j + 1) & 15]));
                        else if constexpr (OP == MIXC4) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[(i + j) & 15]) : "v"(q[(i + j) & 15]));
                        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(i + j) & 15]) : "v"(c));
                    }
                }
                if constexpr (OP == SALU) asm volatile("s_add_u32 s40, s41, 1\n\ts_add_u32 s42, s43, 1\n\ts_add_u32 s44, s45, 1\n\ts_add_u32 s46, s47, 1" ::: "s40", "s42", "s44", "s46", "scc");
                if constexpr (OP == SALUDEP) asm volatile("s_add_u32 s40, s40, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s40, s40, 1" ::: "s40", "scc");
                if constexpr (OP == PKSALU) {        // one packed op and two independent scalar ops: "instr" = the group
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(c2));
                    asm volatile("s_add_u32 s40, s41, 1\n\ts_add_u32 s42, s43, 1" ::: "s40", "s42", "scc");
                }
                if constexpr (OP == BRANCH) asm volatile("s_cmp_lg_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : : "s"(sg[0]) : "scc");
                if constexpr (OP == MFMA) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a[i]), "v"(c));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)sg[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i] + (float)q[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) *sink = s;
}

template <int A, int B>
__global__ __launch_bounds__(512) void probe(int iters, float seed, float *sink, unsigned long long *cyc) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) stream<A>(iters, seed, sink); else stream<B>(iters, seed, sink);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int A, int B>
void run(const char *name, int iters, float *sink, unsigned long long *cyc) {
    const int blocks = 256;
    hipLaunchKernelGGL((probe<A, B>), dim3(blocks), dim3(512), 0, 0, iters, 1.0f, sink, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<A, B>), dim3(blocks), dim3(512), 0, 0, iters, 1.0f, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[b * 8 + w];
    const double n = (double)iters * 64.0;
    printf("%-28s  A %7.2f cyc/instr   B %7.2f cyc/instr   kernel %.3f ms\n", name, sa / (blocks * 4) / n,
           B == NONE ? 0.0 : sb / (blocks * 4) / n, ms);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    float *sink; unsigned long long *cyc;
    hipMalloc(&sink, 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int it = 2000;
    run<FMA32, NONE>("fma32 alone", it, sink, cyc);
    run<FMA32, FMA32>("fma32 || fma32", it, sink, cyc);
    run<PKFMA, NONE>("pk_fma alone", it, sink, cyc);
    run<PKFMA, PKFMA>("pk_fma || pk_fma", it, sink, cyc);
    run<PKADD, NONE>("pk_add alone", it, sink, cyc);
    run<PKADD, PKADD>("pk_add || pk_add", it, sink, cyc);
    run<FMA64, NONE>("fma64 alone", it, sink, cyc);
    run<FMA64, FMA64>("fma64 || fma64", it, sink, cyc);
    run<CVT64, NONE>("cvt_f64_f32 alone", it, sink, cyc);
    run<CVT64, CVT64>("cvt_f64_f32 || same", it, sink, cyc);
    run<LOG, NONE>("log alone", it, sink, cyc);
    run<DOT2, NONE>("dot2_i32_i16 alone", it, sink, cyc);
    run<DOT2, DOT2>("dot2 || dot2", it, sink, cyc);
    run<MFMA, NONE>("mfma16x16x4f32 alone", it, sink, cyc);
    run<MFMA, MFMA>("mfma || mfma", it, sink, cyc);
    run<MFMA, FMA32>("mfma || fma32", it, sink, cyc);
    run<MFMA, PKFMA>("mfma || pk_fma", it, sink, cyc);
    run<MFMA, PKADD>("mfma || pk_add", it, sink, cyc);
    run<MFMA, FMA64>("mfma || fma64", it, sink, cyc);
    run<MIX2, NONE>("[mfma + 2 pk_fma] alone", it, sink, cyc);
    run<MIX4, NONE>("[mfma + 4 pk_fma] alone", it, sink, cyc);
    run<MIX6, NONE>("[mfma + 6 pk_fma] alone", it, sink, cyc);
    run<MIX8, NONE>("[mfma + 8 pk_fma] alone", it, sink, cyc);
    run<MIX4, MIX4>("[mfma + 4 pk] || same", it, sink, cyc);
    run<MIX4, PKFMA>("[mfma + 4 pk] || pk_fma", it, sink, cyc);
    run<MIX6, PKFMA>("[mfma + 6 pk] || pk_fma", it, sink, cyc);
    run<MIXB4, NONE>("[mfma_bf16 + 4 pk] alone", it, sink, cyc);
    run<MIXF4, NONE>("[mfma + 4 fma32] alone", it, sink, cyc);
    run<MIXF8, NONE>("[mfma + 8 fma32] alone", it, sink, cyc);
    run<MIXD4, NONE>("[mfma + 4 dot2] alone", it, sink, cyc);
    run<MIXC4, NONE>("[mfma + 4 cvt_f32_i32] alone", it, sink, cyc);
    run<MIXBF4, NONE>("[mfma_bf16 + 4 fma32] alone", it, sink, cyc);
    run<MIXBF8, NONE>("[mfma_bf16 + 8 fma32] alone", it, sink, cyc);
    run<MIXF4, MIXF4>("[mfma + 4 fma32] || same", it, sink, cyc);
    run<MIXF4, PKFMA>("[mfma + 4 fma32] || pk_fma", it, sink, cyc);
    run<SALU, NONE>("4 x s_add (independent)", it, sink, cyc);
    run<SALUDEP, NONE>("4 x s_add (one chain)", it, sink, cyc);
    run<PKSALU, NONE>("[pk_fma + 2 s_add] alone", it, sink, cyc);
    run<BRANCH, NONE>("[s_cmp + taken branch] alone", it, sink, cyc);
    run<SALU, PKFMA>("s_add || pk_fma", it, sink, cyc);
    run<CVTBF, NONE>("cvt_pk_bf16_f32 alone", it, sink, cyc);
    run<CVTBF, CVTBF>("cvt_pk_bf16 || same", it, sink, cyc);
    run<MFMABF, NONE>("mfma16x16x32bf16 alone", it, sink, cyc);
    run<MFMABF, MFMABF>("mfma_bf16 || mfma_bf16", it, sink, cyc);
    run<MFMABF, FMA32>("mfma_bf16 || fma32", it, sink, cyc);
    run<MFMABF, PKFMA>("mfma_bf16 || pk_fma", it, sink, cyc);
    run<MFMABF, MFMA>("mfma_bf16 || mfma_f32", it, sink, cyc);
    return 0;
}
