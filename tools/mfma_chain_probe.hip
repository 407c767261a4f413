// mfma_chain_probe: is a back-to-back DEPENDENT chain of v_mfma_f32_16x16x32_bf16 (each taking the one before it as SrcC)
// safe as hipcc emits it on gfx950?  (diagnostic; DESIGN.md 4.1 item 4: the twelve-wave kernel's tail got wrong sums from one)
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain_probe tools/mfma_chain_probe.hip && tools/mfma_chain_probe
// Every wave computes D = A1 B1 + A2 B2 + A3 B3 twice -- as ONE accumulator chain and as three independent products summed
// afterwards -- on operands that are small integers (every product and sum exact in fp32, so both must agree bit for bit),
// many times, with other waves of the workgroup hammering the matrix and vector pipes in between (ALONE = 0) or not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

__device__ __forceinline__ uint32_t bf16_pair(int x, int y) {           // two small integers as a bf16 pair (exact)
    return (__float_as_uint((float)x) >> 16) | (__float_as_uint((float)y) & 0xffff0000u);
}

template <bool ALONE, bool OVERLAP, int NOPS = -1>
__global__ __launch_bounds__(768) void probe(int iters, int ncep, unsigned *mismatches, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    unsigned bad = 0;
    float junk = 0.f;
    if (wave == 11 && !ALONE) __builtin_amdgcn_s_setprio(3);     // the tail wave runs at the highest priority
    for (int it = 0; it < iters; ++it) {
        if (wave == 11 || ALONE) {
            u32x4 a[3], b[3];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int s = lane * 7 + it * 13 + t * 5 + d * 3;
                    a[t][d] = bf16_pair((s % 15) - 7, ((s >> 2) % 13) - 6);
                    b[t][d] = bf16_pair(((s >> 1) % 11) - 5, ((s >> 3) % 9) - 4);
                }
            // the chain, exactly as the tail had it: nothing scheduled in between -- and (OVERLAP) with the register
            // assignment the compiler had chosen there: the THIRD instruction's destination is the SECOND's B operand
            f32x4 c;
            if (NOPS >= 0) {
                // the tail's sequence by hand: producer, two vector moves, a TAKEN branch over the other tile's product, NOPS wait
                // states behind the label, the dependent instruction -- twice
                f32x4 acc = zero, e = zero;
                float m0 = 0.f, m1 = 0.f;
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, 0\n\t"
                             "v_mov_b32 %2, 0\n\tv_mov_b32 %3, 0\n\t"
                             "s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\t"
                             "v_mfma_f32_16x16x32_bf16 %1, %6, %5, 0\n"
                             "1:\n\t"
                             ".rept %10\n\ts_nop 0\n\t.endr\n\t"
                             "v_mfma_f32_16x16x32_bf16 %0, %6, %7, %0\n\t"
                             "s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 2f\n\t"
                             "v_mfma_f32_16x16x32_bf16 %1, %8, %7, %1\n"
                             "2:\n\t"
                             ".rept %10\n\ts_nop 0\n\t.endr\n\t"
                             "v_mfma_f32_16x16x32_bf16 %0, %8, %9, %0\n\t"
                             "s_nop 7\n\ts_nop 7\n\ts_nop 7"
                             : "+v"(acc), "+v"(e), "+v"(m0), "+v"(m1)
                             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "n"(NOPS < 0 ? 0 : NOPS)
                             : "scc");
                c = acc;
                junk += e[0] + m0 + m1;
            } else if (OVERLAP) {
                f32x4 acc = zero;
                u32x4 b1 = b[1];
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, 0\n\ts_nop 0\n\t"
                             "v_mfma_f32_16x16x32_bf16 %0, %4, %1, %0\n\ts_nop 1\n\t"
                             "v_mfma_f32_16x16x32_bf16 %1, %5, %6, %0\n\ts_nop 7\n\ts_nop 7"
                             : "+v"(acc), "+v"(b1)
                             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(a[2]), "v"(b[2]));
                c = __builtin_bit_cast(f32x4, b1);
            } else {
                // ... with the uniform branches the tail had between them (a second M tile that n_cep <= 16 skips)
                f32x4 e = zero;
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));   // operands are in registers
                __builtin_amdgcn_sched_barrier(0);
                c = MFMA_BF(a[0], b[0], zero);
                __builtin_amdgcn_sched_barrier(0);
                if (ncep > 16) e = MFMA_BF(a[1], b[0], zero);
                __builtin_amdgcn_sched_barrier(0);
                c = MFMA_BF(a[1], b[1], c);
                __builtin_amdgcn_sched_barrier(0);
                if (ncep > 16) e = MFMA_BF(a[2], b[1], e);
                __builtin_amdgcn_sched_barrier(0);
                c = MFMA_BF(a[2], b[2], c);
                __builtin_amdgcn_sched_barrier(0);
                if (ncep > 16) e = MFMA_BF(a[0], b[2], e);
                junk += e[0] + e[3];
            }
            // independent products
            const f32x4 p0 = MFMA_BF(a[0], b[0], zero), p1 = MFMA_BF(a[1], b[1], zero), p2 = MFMA_BF(a[2], b[2], zero);
            const f32x4 s = (p0 + p1) + p2;
#pragma unroll
            for (int r = 0; r < 4; ++r) bad += (__float_as_uint(c[r]) != __float_as_uint(s[r]));
        } else {
            // the other waves: what the workers do -- packed fp32 and bf16 matrix instructions on independent accumulators
            f32x4 acc[3] = {zero, zero, zero};
            u32x4 x = {(uint32_t)lane, (uint32_t)it, 0x3f803f80u, 0x40004000u};
            float v = (float)(lane + it);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                acc[k % 3] = MFMA_BF(x, x, acc[k % 3]);
                v = __builtin_fmaf(v, 1.0001f, 0.5f);
                v = __builtin_fmaf(v, 0.9999f, -0.5f);
            }
            junk += acc[0][0] + acc[1][1] + acc[2][2] + v;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
    if (junk == 12345.678f) sink[0] = junk;
}

int main() {
    unsigned *d_bad;
    float *d_sink;
    hipMalloc(&d_bad, 8);
    hipMalloc(&d_sink, 4);
    for (int mode = 0; mode < 4; ++mode) {
        const int alone = mode & 1, overlap = mode >> 1;
        hipMemset(d_bad, 0, 8);
        const int iters = 20000, grid = 256;
        if (alone && overlap) hipLaunchKernelGGL((probe<true, true>), dim3(grid), dim3(768), 0, 0, iters, 13, d_bad, d_sink);
        else if (alone) hipLaunchKernelGGL((probe<true, false>), dim3(grid), dim3(768), 0, 0, iters, 13, d_bad, d_sink);
        else if (overlap) hipLaunchKernelGGL((probe<false, true>), dim3(grid), dim3(768), 0, 0, iters, 13, d_bad, d_sink);
        else hipLaunchKernelGGL((probe<false, false>), dim3(grid), dim3(768), 0, 0, iters, 13, d_bad, d_sink);
        hipDeviceSynchronize();
        unsigned bad = 0;
        hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        const double n = (double)grid * (alone ? 12 : 1) * iters * 256.0;
        printf("%s, %s: %u of %.3g result values of the chain differ from the independent sum\n",
               alone ? "every wave runs the chain" : "one wave runs the chain beside eleven busy waves",
               overlap ? "third destination = second B operand" : "uniform branches between the three as in the tail", bad, n);
    }
    // the hand-written sequence: how many wait states behind the label does the dependent instruction need?
    auto run = [&](auto kern, int nops) {
        hipMemset(d_bad, 0, 8);
        const int iters = 20000, grid = 256;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(768), 0, 0, iters, 13, d_bad, d_sink);
        hipDeviceSynchronize();
        unsigned bad = 0;
        hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        printf("taken branch between the links, %d wait state(s) behind the label, one wave at priority 3 beside eleven busy waves: "
               "%u of %.3g values differ\n", nops, bad, 256.0 * 20000 * 256.0);
    };
    run(probe<false, false, 0>, 0);
    run(probe<false, false, 1>, 1);
    run(probe<false, false, 2>, 2);
    run(probe<false, false, 4>, 4);
    run(probe<false, false, 8>, 8);
    return 0;
}
