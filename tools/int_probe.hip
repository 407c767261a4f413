// int_probe: issue and pipe cost of the integer instructions the fixed-point kernels are made of (diagnostic, not
// product).   hipcc --offload-arch=gfx950 -O3 -o tools/int_probe tools/int_probe.hip && tools/int_probe
// Same harness as alu_probe: a block = 8 waves = two per SIMD; waves 0-3 run stream A, waves 4-7 stream B (or
// nothing); a stream is iters x 64 independent instructions; reported: cycles (s_memtime) per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Op { NONE = 0, ADD, SDWA_ADD, SDWA_SHR, ASHR, DOT2S, MADU64, MUL24, CNDMASK, PERM, PKADD16, PKASHR16, MADI16, ALIGNBIT, BFE, MULLO, SWIZZLE, BFLY, CND64, BFI, CNDDEP, CMPCND, I8MFMA, I8MFMA32 };

template <int OP>
__device__ __forceinline__ void stream(int iters, int seed, int *sink) {
    if constexpr (OP == NONE) return;
    int q[16]; unsigned long long w[8];
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v16i __attribute__((ext_vector_type(16)));
    v4i acc4[8]; v16i acc16[4]; v4i opa = {seed, 2, 3, 4}, opb = {5, 6, 7, seed};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc4[i] = (v4i){i, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc16[i] = (v16i){i};
#pragma unroll
    for (int i = 0; i < 16; ++i) q[i] = seed * 77 + i * 12345;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = seed + i;
    const int bias = 8191, one = 1;
    const unsigned long long msk = 0x5555555555555555ull ^ (unsigned long long)seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = (i + 1) & 15;
                if constexpr (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q[i]) : "v"(q[j]));
                if constexpr (OP == SDWA_ADD) asm volatile("v_add_u32_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(q[i]) : "v"(q[j]));
                if constexpr (OP == SDWA_SHR) asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(q[i]) : "s"(one), "v"(q[j]));
                if constexpr (OP == ASHR) asm volatile("v_ashrrev_i32 %0, 14, %0" : "+v"(q[i]));
                if constexpr (OP == DOT2S) asm volatile("v_dot2_i32_i16 %0, %0, %1, %2" : "+v"(q[i]) : "v"(q[j]), "s"(bias));
                if constexpr (OP == MADU64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i & 7]) : "v"(q[i]), "v"(q[j]) : "vcc");
                if constexpr (OP == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(q[i]) : "v"(q[j]));
                if constexpr (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(q[i]) : "v"(q[j]));
                if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(q[i]) : "v"(q[j]));
                if constexpr (OP == CND64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(q[i]) : "v"(q[j]), "s"(msk));
                if constexpr (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(q[i]) : "v"(q[15]), "v"(q[j]));
                if constexpr (OP == CMPCND) asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(q[i]) : "v"(q[j]), "v"(q[(i + 2) & 15]) : "vcc");
                if constexpr (OP == I8MFMA) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc4[i & 7]) : "v"(opa), "v"(opb));
                if constexpr (OP == I8MFMA32) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc16[i & 3]) : "v"(opa), "v"(opb));
                if constexpr (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(q[j]), "s"(0x07060302));
                if constexpr (OP == PKADD16) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(q[i]) : "v"(q[j]));
                if constexpr (OP == PKASHR16) asm volatile("v_pk_ashrrev_i16 %0, 1, %0" : "+v"(q[i]));
                if constexpr (OP == MADI16) asm volatile("v_mad_i32_i16 %0, %0, %1, %2" : "+v"(q[i]) : "v"(q[j]), "s"(bias));
                if constexpr (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 15" : "+v"(q[i]) : "v"(q[j]));
                if constexpr (OP == BFE) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(q[i]));
                if constexpr (OP == SWIZZLE) asm volatile("ds_swizzle_b32 %0, %0 offset:0x41f\n\ts_waitcnt lgkmcnt(8)" : "+v"(q[i]));
                if constexpr (OP == BFLY) {      // the packed butterfly of kernels_generic.hpp: "instr" = one butterfly (12 ops)
                    int a1, a2, t0, t1, t2, t3;
                    asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(a1) : "v"(q[j]), "v"(q[(i + 2) & 15]), "s"(bias));
                    asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(a2) : "v"(q[j]), "v"(q[(i + 3) & 15]), "s"(bias));
                    asm volatile("v_ashrrev_i32 %0, 14, %0" : "+v"(a1));
                    asm volatile("v_ashrrev_i32 %0, 14, %0" : "+v"(a2));
                    asm volatile("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(t0) : "v"(q[i]), "v"(a1));
                    asm volatile("v_sub_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(t1) : "v"(q[i]), "v"(a1));
                    asm volatile("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(t2) : "v"(q[i]), "v"(a2));
                    asm volatile("v_sub_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(t3) : "v"(q[i]), "v"(a2));
                    asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(t0));
                    asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(t1));
                    asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(t0) : "s"(one), "v"(t2));
                    asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(t1) : "s"(one), "v"(t3));
                    q[i] = t0; q[j] = t1;
                }
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += q[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (int)w[i] + acc4[i][0] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][15];
    if (s == 123456789) *sink = s;
}

template <int A, int B>
__global__ __launch_bounds__(512) void probe(int iters, int seed, int *sink, unsigned long long *cyc) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) stream<A>(iters, seed, sink); else stream<B>(iters, seed, sink);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int A, int B>
void run(const char *name, int iters, int *sink, unsigned long long *cyc) {
    const int blocks = 256;
    hipLaunchKernelGGL((probe<A, B>), dim3(blocks), dim3(512), 0, 0, iters, 1, sink, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<A, B>), dim3(blocks), dim3(512), 0, 0, iters, 1, sink, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[b * 8 + w];
    const double n = (double)iters * 64.0;
    printf("%-26s  alone/A %7.2f   B %7.2f  (s_memtime ticks per instr; x24 = shader clocks at 2.4 GHz / 100 MHz)\n", name,
           sa / (blocks * 4) / n, B == NONE ? 0.0 : sb / (blocks * 4) / n);
}

#define BOTH(OP, name) run<OP, NONE>(name " alone", it, sink, cyc); run<OP, OP>(name " || same", it, sink, cyc)
int main() {
    int *sink; unsigned long long *cyc;
    hipMalloc(&sink, 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int it = 1000;
    BOTH(ADD, "v_add_u32");
    BOTH(SDWA_ADD, "v_add_u32_sdwa sext");
    BOTH(SDWA_SHR, "v_lshrrev_sdwa W1");
    BOTH(ASHR, "v_ashrrev_i32");
    BOTH(DOT2S, "v_dot2_i32_i16 +s");
    BOTH(MADU64, "v_mad_u64_u32");
    BOTH(MUL24, "v_mul_u32_u24");
    BOTH(MULLO, "v_mul_lo_u32");
    BOTH(CNDMASK, "v_cndmask_b32");
    BOTH(CND64, "v_cndmask_e64 sgpr");
    BOTH(BFI, "v_bfi_b32");
    BOTH(CMPCND, "v_cmp + v_cndmask");
    BOTH(PERM, "v_perm_b32");
    BOTH(PKADD16, "v_pk_add_i16");
    BOTH(PKASHR16, "v_pk_ashrrev_i16");
    BOTH(MADI16, "v_mad_i32_i16");
    BOTH(ALIGNBIT, "v_alignbit_b32");
    BOTH(BFE, "v_bfe_i32");
    BOTH(SWIZZLE, "ds_swizzle_b32");
    BOTH(BFLY, "packed butterfly (12)");
    BOTH(I8MFMA, "v_mfma_i32_16x16x64_i8");
    BOTH(I8MFMA32, "v_mfma_i32_32x32x32_i8");
    run<I8MFMA, ADD>("mfma_i8 || v_add_u32", it, sink, cyc);
    run<I8MFMA, DOT2S>("mfma_i8 || dot2", it, sink, cyc);
    return 0;
}
