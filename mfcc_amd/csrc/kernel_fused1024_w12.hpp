// Fused 1024/341/40 float kernel, TWELVE-wave form: the arithmetic, codelets, MFMA lists and operand tables of
// kernel_fused1024_f32.hpp (read its header first), staged like kernel_fused512_w12.hpp -- every SIMD always has three
// waves in three DIFFERENT phases: a worker in pass 1, a worker in pass 2, a helper.
//
// Why.  The eight-wave kernel runs all its waves in lockstep (pass 1 | barrier | pass 2 + MFMAs | barrier): the two
// waves of a SIMD wait for LDS, for the barrier and for the matrix pipe at the same moments, and a 16-frame tile takes
// ~7 200 clocks of which the vector + matrix pipes are busy ~3 900 (profiles/r03_fused1024_pmc.json).  Three re-stagings
// of round 3 lost to it because each paid more instructions per frame.  This one does not:
//
//   waves 0..3   group A: the four workers of a 16-frame tile
//   waves 4..7   group B: the four workers of ANOTHER tile, half a period behind A
//   waves 8, 9   fetch the sample windows from HBM two half-steps ahead and park them (pre-emphasised fp32)
//   wave  10     column 16 of the group in pass 2: 32-point DFT matrix (16 MFMAs) + its mel contribution
//   wave  11     the tail of the tile that finished pass 2 in the previous half-step: log2, DCT-II, store
//
//   * pass 1: a worker runs the real FFT-32 codelet TWICE per tile (frames 4 wi + 2 b + (lane >> 5), b = 0, 1) -- 512
//     lanes of work on four waves; window and twiddle constants are read from an LDS copy once per tile for both;
//   * pass 2: lane (frame lo, column k1 = 4 wi + q) reads its column ONCE (16 ds_read_b128) and runs BOTH halves of the
//     decimation-in-frequency split (cfft32_h0, cfft32_h1) on it: half the T reads of the eight-wave kernel, and both
//     halves' 17 + 18 mel MFMAs accumulate in the same registers;
//   * the MFMA A operands of a half (19 dwords per lane) are fetched from the L2-resident table while that half's FFT
//     runs -- resident they cost 38 registers, and three waves per SIMD have 168;
//   * ONE transpose tile T (64 KB) for both groups -- two do not fit beside the windows.  The group in pass 1 stores
//     its columns while the group in pass 2 is long past reading its own: a pass-2 wave bumps an LDS counter behind
//     its 16 reads (the LDS executes a wave's instructions in order), a pass-1 wave polls it before its first store
//     (one read that practically always succeeds: the stores come ~1 500 clocks after the half-step's barrier, the
//     reads take ~500).
//
// Time runs in half-steps h separated by ONE LDS-only workgroup barrier each:
//   group A: pass 1 of its tile k at h = 2 k,     pass 2 + mel at h = 2 k + 1
//   group B: pass 1 of its tile k at h = 2 k + 1, pass 2 + mel at h = 2 k + 2
//   parkers: S_A(k) is rewritten while A runs pass 2 (odd h: tile (h + 1) / 2), S_B(k) at even h >= 2 (tile h / 2)
//   column 16 at h: the group in pass 2 (its V was written in h - 1); partial sums to Q slot 4 of the group
//   tail at h: the group that was in pass 2 at h - 1
// LDS 159 KB: T, per group V + Q (5 slots x 3 blocks) + S, the window / twiddle constants, two counters.
//
// Two kernels in this file: mfcc_fused1024_w12_kernel<R> (this namespace: the fp32 lists of kernel_fused1024_f32.hpp, five
// sample rates, an A/B form: MFCC_HIP_FUSED1024=w12) and, at the end, mfcc_fused1024_w12bf_kernel<VAR> (the bf16-split set
// lists of kernel_fused1024.hpp, every sample rate): the one a handle runs, 1.5 % ahead of the other at 16 kHz.
#pragma once

#include "kernel_fused1024.hpp"
#include "kernel_fused1024_f32.hpp"

namespace mfcc_fused1024_w12 {

using namespace mfcc_fused1024_f32;

constexpr int kW12Waves = 12;
constexpr int kQSlots = 5;                                   // 4 workers + column 16
constexpr int kQGroupWords = kQSlots * kBlocks * 256;
constexpr int kGroupWords = kTile * kVStride + kQGroupWords + kSUsed;
constexpr int kCRow = 36;                                    // words per n2 row of the constant tables (9 x 16 B: conflict free)
constexpr int kConstWords = 2 * 32 * kCRow;
constexpr int kW12LdsWords = kTile * kTFrame + 2 * kGroupWords + kConstWords + 4;
static_assert(kW12LdsWords * 4 <= 160 * 1024, "LDS");
constexpr int kParkers = 128, kParkPieces = (kPieces + kParkers - 1) / kParkers;      // 7: the 7th is piece 768 alone

struct FetchN {
    i32x4 v[kParkPieces];
    int p[kParkPieces];          // dword in front of v[k]: its high half is the piece's predecessor sample
};

__device__ __forceinline__ void fetch_window_n(const mfcc_k::StreamDesc &s, const Window &w, int u, FetchN &f) {
    if (w.inside) {
        const i32x4 *g = reinterpret_cast<const i32x4 *>(w.ptr - w.shift);
        const int *g32 = reinterpret_cast<const int *>(g);
#pragma unroll
        for (int k = 0; k < kParkPieces; ++k)
            if (k * kParkers + u < kPieces) {
                f.v[k] = g[k * kParkers + u];
                f.p[k] = g32[4 * (k * kParkers + u) - 1];
            }
    } else {
        const long long first = (long long)w.t_in * kTileHop;      // channel-relative
        const int16_t *base = w.ptr - first;
#pragma unroll
        for (int k = 0; k < kParkPieces; ++k)
            if (k * kParkers + u < kPieces) {
                int h[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = mfcc_k::sample_at_i(s, base, first + 8 * (k * kParkers + u) + j) & 0xFFFF;
                f.v[k] = (i32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
                f.p[k] = mfcc_k::sample_at_i(s, base, first + 8 * (k * kParkers + u) - 1) << 16;
            }
    }
}

__device__ __forceinline__ void park_window_n(float *Sf, int u, const FetchN &f) {
#ifdef MFCC_1K12_EXP_NOPARK
    if (u == 0) Sf[0] = (float)(f.v[0][0] + f.p[0] + f.v[5][1]);
    return;
#endif
#pragma unroll
    for (int k = 0; k < kParkPieces; ++k)
        if (k * kParkers + u < kPieces) preemph8(f.p[k], f.v[k], Sf + 8 * (k * kParkers + u));
}

__device__ __forceinline__ Cursor cursor_of(const mfcc_k::StreamDesc &s, const LaunchGeom &g, unsigned v) {
    Cursor c;
    c.ch = (int)(v / (unsigned)g.tiles_per_ch);
    c.t_in = (int)(v - (unsigned)c.ch * (unsigned)g.tiles_per_ch);
    c.ptr = s.pcm + (long long)c.ch * s.ch_stride + (long long)c.t_in * kTileHop;
    return c;
}

// The A operands of one half's mel MFMAs, lane-private dwords of a [n][64] table: issued as ONE burst of loads off a
// scalar base (no address registers) and waited for by hand right before the MFMAs.  Left to the compiler the loads
// sank to one load + s_waitcnt vmcnt(0) in front of each MFMA -- an L2 round trip per matrix instruction (first run of
// this kernel: 1.66 ms against the eight-wave kernel's 1.44).
template <int N>
__device__ __forceinline__ void load_a_burst(float (&am)[kAmel], const float *base, int voff) {
    const float *base2 = base + 15 * 64;               // offsets are 13-bit signed: a second base from element 15 on
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < 15) asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(am[i]) : "v"(voff), "s"(base), "n"(i * 256));
        else asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(am[i]) : "v"(voff), "s"(base2), "n"((i - 15) * 256));
    }
}
__device__ __forceinline__ void wait_a_burst(float (&am)[kAmel]) {
    static_assert(kAmel == 19, "operand list");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(am[0]), "+v"(am[1]), "+v"(am[2]), "+v"(am[3]), "+v"(am[4]), "+v"(am[5]), "+v"(am[6]), "+v"(am[7]),
                   "+v"(am[8]), "+v"(am[9]), "+v"(am[10]), "+v"(am[11]), "+v"(am[12]), "+v"(am[13]), "+v"(am[14]),
                   "+v"(am[15]), "+v"(am[16]), "+v"(am[17]), "+v"(am[18]));
}

#ifndef MFCC_1K12_PRIO_P1
#define MFCC_1K12_PRIO_P1 1
#endif
#ifndef MFCC_1K12_PRIO_P2
#define MFCC_1K12_PRIO_P2 0
#endif

// Diagnostic build only (-DMFCC_1K12_STAMPS, tools/stamps1k.py): per wave, s_memtime ticks spent in up to eight
// segments of a half-step, summed over workgroups; written to a buffer nothing else reads.
#ifdef MFCC_1K12_STAMPS
__device__ unsigned long long g_stamps1k[kW12Waves * 12];
#define ST1K_BEGIN unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define ST1K(k) do { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_acc[k] += st_now - st_last; st_last = st_now; } while (0)
#define ST1K_LDS(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ST1K(k); } while (0)
#define ST1K_END do { if (lane == 0) { for (int st_k = 0; st_k < 12; ++st_k) atomicAdd(&g_stamps1k[wave * 12 + st_k], st_acc[st_k]); } } while (0)
#else
#define ST1K_BEGIN
#define ST1K(k)
#define ST1K_LDS(k)
#define ST1K_END
#endif

template <int R>
__global__ __launch_bounds__(64 * kW12Waves) __attribute__((amdgpu_waves_per_eu(3, 3)))
void mfcc_fused1024_w12_kernel(mfcc_k::StreamDesc s, Tables t, LaunchGeom g, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[kW12LdsWords];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;         // 0: A, 1: B, 2: helpers
    const int wi = wave & 3;
    const int lo = lane & 15;
    const int q = lane >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    float *const T = lds;
    auto Vt = [&](int gi) { return lds + kTile * kTFrame + gi * kGroupWords; };
    auto Qt = [&](int gi) { return lds + kTile * kTFrame + gi * kGroupWords + kTile * kVStride; };
    auto Sf = [&](int gi) { return lds + kTile * kTFrame + gi * kGroupWords + kTile * kVStride + kQGroupWords; };
    float *const WinC = lds + kTile * kTFrame + 2 * kGroupWords;          // [32 n2][kCRow]: 32 n1
    float *const TwC = WinC + 32 * kCRow;                                  // [32 n2][kCRow]: 16 k1 x (cos, sin)
    int *const Flag = reinterpret_cast<int *>(TwC + 32 * kCRow);          // [group]: pass-2 waves that have read their column

    // XCD-aware tile order (kernel_fused512_w12.hpp): consecutive tile pairs on workgroups of the same XCD
    const unsigned nwg = gridDim.x;
    const unsigned bid = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    const unsigned va = 2u * bid, vb = va + 1u;
    const int n_tiles = g.tiles_per_ch * g.n_ch;                          // < 2^30 (host check)
    const int gv = 2 * (int)gridDim.x;
    const int nA = (int)va < n_tiles ? (n_tiles - (int)va + gv - 1) / gv : 0;
    const int nB = (int)vb < n_tiles ? (n_tiles - (int)vb + gv - 1) / gv : 0;
    const int last_h = 2 * nA + 1;                                        // B's last tail (nB <= nA) is at 2 nB + 1

    // the constants of pass 1 into LDS, once per workgroup
    for (int i = tid; i < 32 * 32; i += 64 * kW12Waves) {
        WinC[(i >> 5) * kCRow + (i & 31)] = t.win[i];
        TwC[(i >> 5) * kCRow + (i & 31)] = t.tw[i];
    }
    if (tid < 2) Flag[tid] = 0;

    if (grp < 2) {
        // =========================================================================== workers
        const int gi = grp;
        const int n2 = lane & 31;
        float *const V = Vt(gi), *const Q = Qt(gi), *const S = Sf(gi);
        const int n_mine = gi ? nB : nA;
        const int n_other = gi ? nA : nB;
        Cursor cur = cursor_of(s, g, gi ? vb : va);
        const float *const amp = t.a_mel + (size_t)(2 * wi) * kAmel * 64;      // uniform; lane offset in bytes below
        const int lane4 = lane * 4;

        ST1K_BEGIN
        auto pass1 = [&](int i) {
            // ---------------- pass 1: windowed real FFT-32 over n1, two batches of two frames
            __builtin_amdgcn_s_setprio(MFCC_1K12_PRIO_P1);
            const int shift = window_of(cur, g).shift;
            advance(cur, g);
            const int fr0 = 4 * wi + (lane >> 5);
            v2f ep[16];
            {
                const float *sp = S + fr0 * kHop + n2 + shift;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[32 * n1];
            }
            v2f wp[16], tw[16];
            {
                const f32x4 *w4 = reinterpret_cast<const f32x4 *>(WinC + n2 * kCRow);
                const f32x4 *t4 = reinterpret_cast<const f32x4 *>(TwC + n2 * kCRow);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 a = w4[j], b = t4[j];
                    wp[2 * j] = (v2f){a[0], a[1]};
                    wp[2 * j + 1] = (v2f){a[2], a[3]};
                    tw[2 * j] = (v2f){b[0], b[1]};
                    tw[2 * j + 1] = (v2f){b[2], b[3]};
                }
            }
            v2f ty[16];
            float y16;
            ST1K_LDS(0);
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
            ST1K(1);
            {
                // the second batch's operands fly while the first batch's columns are stored
                const float *sp = S + (fr0 + 2) * kHop + n2 + shift;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[32 * n1];
            }
            // T still holds the partner group's tile until its four pass-2 waves have read their columns
            const int need = 4 * (gi ? i + 1 : (i < n_other ? i : n_other));
#ifndef MFCC_1K12_EXP_NOPOLL
            while (__hip_atomic_load(Flag + (gi ^ 1), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - need < 0)
                __builtin_amdgcn_s_sleep(1);
#else
            (void)need;
#endif
            ST1K(2);
            {
                v2f *tcol0 = reinterpret_cast<v2f *>(T + fr0 * kTFrame) + n2;      // a store's lanes: consecutive n2
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
                V[fr0 * kVStride + n2] = y16;
            }
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
            {
                v2f *tcol0 = reinterpret_cast<v2f *>(T + (fr0 + 2) * kTFrame) + n2;
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
                V[(fr0 + 2) * kVStride + n2] = y16;
            }
            ST1K_LDS(3);
        };
        auto pass2 = [&]() {
            // ---------------- pass 2: the complex FFT-32 over n2 of column k1 = 4 wi + q, frame lo; mel MFMAs
            __builtin_amdgcn_s_setprio(MFCC_1K12_PRIO_P2);
            v2f xl[16], xh[16], pp[8];
            const f32x4 *trow = reinterpret_cast<const f32x4 *>(T + lo * kTFrame + (4 * wi + q) * kTRow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 a = trow[i], b = trow[8 + i];
                xl[2 * i] = (v2f){a[0], a[1]};
                xl[2 * i + 1] = (v2f){a[2], a[3]};
                xh[2 * i] = (v2f){b[0], b[1]};
                xh[2 * i + 1] = (v2f){b[2], b[3]};
            }
            ST1K_LDS(5);
            if (lane == 0) __hip_atomic_fetch_add(Flag + gi, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            float am[kAmel], pw[16];
#pragma unroll
            for (int i = 0; i < kAmel; ++i) am[i] = 0.0f;
            f32x4 acc[kBlocks] = {zero, zero, zero};
            load_a_burst<Sched<R>::N0>(am, amp, lane4);
            __builtin_amdgcn_sched_barrier(0);
            mfcc_codelets::cfft32_h0_pow(xl, xh, pp);
#pragma unroll
            for (int m = 0; m < 8; ++m) pw[m] = pp[m].x, pw[m + 8] = pp[m].y;
            __builtin_amdgcn_sched_barrier(0);         // (the wait is hoisted over the codelet otherwise)
            ST1K(8);
            wait_a_burst(am);
            ST1K(9);
#ifndef MFCC_1K12_EXP_NOMFMA
            mel_mfmas<R, 0>(pw, am, acc);
#else
            acc[0][0] += pw[0] + pw[5] + am[0] + am[7] + pw[9] + pw[15];
#endif
            __builtin_amdgcn_sched_barrier(0);
            ST1K(10);
            load_a_burst<Sched<R>::N1>(am, amp + kAmel * 64, lane4);
            __builtin_amdgcn_sched_barrier(0);
            mfcc_codelets::cfft32_h1_pow(xl, xh, pp);
#pragma unroll
            for (int m = 0; m < 8; ++m) pw[m] = pp[m].x, pw[m + 8] = pp[m].y;
            __builtin_amdgcn_sched_barrier(0);
            ST1K(11);
            wait_a_burst(am);
#ifndef MFCC_1K12_EXP_NOMFMA
            mel_mfmas<R, 1>(pw, am, acc);
#else
            acc[1][0] += pw[0] + pw[5] + am[0] + am[7] + pw[9] + pw[15];
#endif
#pragma unroll
            for (int b = 0; b < kBlocks; ++b)
                *reinterpret_cast<f32x4 *>(Q + ((wi * kBlocks + b) * 64 + lane) * 4) = acc[b];
            ST1K_LDS(6);
        };
        lds_barrier();                                 // prologue: S_A(0), S_B(0), the constants and the counters are in LDS
        ST1K(7);
        int bars = last_h + 1;                         // every wave of the workgroup passes this many barriers
        if (gi) {                                      // h = 0: group B idles
            lds_barrier();
            --bars;
        }
        for (int i = 0; i < n_mine; ++i) {
            ST1K(7);
            pass1(i);
            lds_barrier();
            ST1K(4);
            pass2();
            lds_barrier();
            bars -= 2;
        }
        ST1K_END;
        for (; bars > 0; --bars) lds_barrier();
    } else if (wi < 2) {
        // =========================================================================== parkers (waves 8, 9)
        const int u = wi * 64 + lane;                  // 0..127
        __builtin_amdgcn_s_setprio(3);
        Cursor pa = cursor_of(s, g, va), pb = cursor_of(s, g, vb);
        int ka = 0, kb = 0;                            // next tile of each stream to fetch
        FetchN fa, fb;
        bool have_a = false, have_b = false;
        if (nA > 0) {                                  // prologue: S_A(0) and S_B(0) directly
            fetch_window_n(s, window_of(pa, g), u, fa);
            park_window_n(Sf(0), u, fa);
            advance(pa, g);
            ++ka;
        }
        if (nB > 0) {
            fetch_window_n(s, window_of(pb, g), u, fb);
            park_window_n(Sf(1), u, fb);
            advance(pb, g);
            ++kb;
        }
        if (ka < nA) {                                 // S_A(1): parked at h = 1
            fetch_window_n(s, window_of(pa, g), u, fa);
            advance(pa, g);
            ++ka;
            have_a = true;
        }
        if (kb < nB) {                                 // S_B(1): parked at h = 2
            fetch_window_n(s, window_of(pb, g), u, fb);
            advance(pb, g);
            ++kb;
            have_b = true;
        }
        lds_barrier();
        ST1K_BEGIN
        for (int h = 0; h <= last_h; ++h) {
            ST1K(4);
            // a group's window is rewritten while that group runs pass 2; it reads it in its next pass 1.  The next
            // window of the same stream is fetched right behind: two half-steps of lead over its park
            if (h & 1) {
                if (have_a) park_window_n(Sf(0), u, fa);
                have_a = false;
                if (ka < nA) {
                    fetch_window_n(s, window_of(pa, g), u, fa);
                    advance(pa, g);
                    ++ka;
                    have_a = true;
                }
            } else if (h >= 2) {
                if (have_b) park_window_n(Sf(1), u, fb);
                have_b = false;
                if (kb < nB) {
                    fetch_window_n(s, window_of(pb, g), u, fb);
                    advance(pb, g);
                    ++kb;
                    have_b = true;
                }
            }
            ST1K_LDS(0);
            lds_barrier();
        }
        ST1K_END;
    } else if (wi == 2) {
        // =========================================================================== column 16 (wave 10)
        __builtin_amdgcn_s_setprio(3);
        float a1[kAextra], a2[kAextra];
#pragma unroll
        for (int i = 0; i < kAextra; ++i) {
            a1[i] = t.a_extra[(1 * kAextra + i) * 64 + lane];
            a2[i] = t.a_extra[(2 * kAextra + i) * 64 + lane];
        }
        lds_barrier();
        ST1K_BEGIN
        for (int h = 0; h <= last_h; ++h) {
            ST1K(4);
            // the group in pass 2 at h: A (tile (h - 1) / 2) for odd h, B (tile h / 2 - 1) for even h >= 2
            const int gi = (h & 1) ? 0 : 1;
            const int k = (h & 1) ? (h - 1) / 2 : h / 2 - 1;
            if (k >= 0 && k < (gi ? nB : nA)) {
                const float *V = Vt(gi) + lo * kVStride + q;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = V[4 * j];
                // bins 16 + 32 j: role 1 of the eight-wave kernel (j = 0..7) and role 2 (j = 8..15) share the B operands
                f32x4 s1 = zero, s1b = zero, s2 = zero, s2b = zero;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    s1 = MFCC1K_MFMA(a1[j], v[j], s1);
                    s2 = MFCC1K_MFMA(a2[j], v[j], s2);
                    s1b = MFCC1K_MFMA(a1[j + 1], v[j + 1], s1b);
                    s2b = MFCC1K_MFMA(a2[j + 1], v[j + 1], s2b);
                }
                s1 += s1b;
                s2 += s2b;
                const float p10 = fmaf(s1[0], s1[0], s1[1] * s1[1]), p11 = fmaf(s1[2], s1[2], s1[3] * s1[3]);
                const float p20 = fmaf(s2[0], s2[0], s2[1] * s2[1]), p21 = fmaf(s2[2], s2[2], s2[3] * s2[3]);
                f32x4 acc[kBlocks] = {zero, zero, zero};
#pragma unroll
                for (int i = 0; i < Sched<R>::NS1; ++i)
                    acc[Sched<R>::S1blk[i]] = MFCC1K_MFMA(a1[8 + i], Sched<R>::S1step[i] ? p11 : p10, acc[Sched<R>::S1blk[i]]);
#pragma unroll
                for (int i = 0; i < Sched<R>::NS2; ++i)
                    acc[Sched<R>::S2blk[i]] = MFCC1K_MFMA(a2[8 + i], Sched<R>::S2step[i] ? p21 : p20, acc[Sched<R>::S2blk[i]]);
                float *Q = Qt(gi);
#pragma unroll
                for (int b = 0; b < kBlocks; ++b)
                    *reinterpret_cast<f32x4 *>(Q + ((4 * kBlocks + b) * 64 + lane) * 4) = acc[b];
            }
            ST1K_LDS(0);
            lds_barrier();
        }
        ST1K_END;
    } else {
        // =========================================================================== tail (wave 11)
        __builtin_amdgcn_s_setprio(3);
        float ax[kAextra];
#pragma unroll
        for (int i = 0; i < kAextra; ++i) ax[i] = t.a_extra[(0 * kAextra + i) * 64 + lane];
        const int lane_off = lo * t.n_cep + 4 * q;
        Cursor ta = cursor_of(s, g, va), tb = cursor_of(s, g, vb);
        lds_barrier();
        ST1K_BEGIN
        for (int h = 0; h <= last_h; ++h) {
            ST1K(4);
            // the group that was in pass 2 at h - 1: A (tile h / 2 - 1) for even h, B (tile (h - 3) / 2) for odd h
            const int gi = (h & 1) ? 1 : 0;
            const int k = (h & 1) ? (h - 3) / 2 : h / 2 - 1;
            if (h >= 2 && k >= 0 && k < (gi ? nB : nA)) {
                const f32x4 *Q4 = reinterpret_cast<const f32x4 *>(Qt(gi)) + lane;
                f32x4 lm[kBlocks];
#pragma unroll
                for (int b = 0; b < kBlocks; ++b) {
                    const f32x4 m = ((Q4[(0 * kBlocks + b) * 64] + Q4[(1 * kBlocks + b) * 64]) +
                                     (Q4[(2 * kBlocks + b) * 64] + Q4[(3 * kBlocks + b) * 64])) + Q4[(4 * kBlocks + b) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) lm[b][r] = __builtin_amdgcn_logf(m[r]);
                }
                if (q >= 2) lm[2] = zero;              // filters 40..47 do not exist: no -inf * 0 in the DCT
                f32x4 d[kBlocks] = {zero, zero, zero};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int b = 0; b < kBlocks; ++b) d[b] = MFCC1K_MFMA(ax[4 * b + r], lm[b][r], d[b]);
                if (gi) {
                    dct_store(s, t, d, lm, tb, lo, q, lane, lane_off, out);
                    advance(tb, g);
                } else {
                    dct_store(s, t, d, lm, ta, lo, q, lane, lane_off, out);
                    advance(ta, g);
                }
            }
            ST1K_LDS(0);
            lds_barrier();
        }
        ST1K_END;
    }
}

inline const char *kernel_name() { return "mfcc_fused1024_w12_kernel"; }

// returns false when the problem does not fit (then the eight-wave kernel runs)
inline bool launch(const mfcc_k::StreamDesc &s, const Tables &t, float *out, int n_cu, hipStream_t stream) {
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    if (n_tiles >= (1ll << 30) || tiles_per_ch >= (1ll << 26) || n_ch >= (1ll << 30)) return false;
    long long wgs = (n_tiles + 1) / 2;
    if (wgs > n_cu) wgs = n_cu;
    if (wgs < 1) wgs = 1;
    const long long grid = 2 * wgs;                      // virtual workgroups: the cursor stride
    LaunchGeom g;
    g.tiles_per_ch = (int)tiles_per_ch;
    g.n_ch = (int)n_ch;
    g.grid_div = (int)(grid / tiles_per_ch);
    g.grid_mod = (int)(grid % tiles_per_ch);
    g.step_ptr = (long long)g.grid_div * s.ch_stride + (long long)g.grid_mod * kTileHop;
    g.wrap_ptr = s.ch_stride - tiles_per_ch * (long long)kTileHop;
    g.t_lo = (int)((9 - (long long)s.halo + kTileHop - 1) / kTileHop);
    if (g.t_lo < 0) g.t_lo = 0;
    const long long hi = (s.n_samples - kSUsed) / kTileHop;
    g.t_hi = s.n_samples < kSUsed ? -1 : (int)(hi < tiles_per_ch ? hi : tiles_per_ch);
    const dim3 grid3((unsigned)wgs), block3(64 * kW12Waves);
    switch (t.sched) {
    case 1: hipLaunchKernelGGL(mfcc_fused1024_w12_kernel<1>, grid3, block3, 0, stream, s, t, g, out); break;
    case 2: hipLaunchKernelGGL(mfcc_fused1024_w12_kernel<2>, grid3, block3, 0, stream, s, t, g, out); break;
    case 3: hipLaunchKernelGGL(mfcc_fused1024_w12_kernel<3>, grid3, block3, 0, stream, s, t, g, out); break;
    case 4: hipLaunchKernelGGL(mfcc_fused1024_w12_kernel<4>, grid3, block3, 0, stream, s, t, g, out); break;
    default: hipLaunchKernelGGL(mfcc_fused1024_w12_kernel<0>, grid3, block3, 0, stream, s, t, g, out); break;
    }
    return true;
}

}  // namespace mfcc_fused1024_w12

// ---------------------------------------------------------------------------------------------------------------------
// The same staging with the mel contraction of kernel_fused1024.hpp: both operands split in two bf16 terms, one
// v_mfma_f32_16x16x32_bf16 triple per (K group, filter block) set -- 12-15 matrix instructions of 16 clocks per half
// instead of 17-18 of 32 (during which the SIMD issues nothing else), for 48 vector instructions of splitting; every
// sample rate.  The A operands of a half (sets x (hi, lo) x 16 bytes per lane) are streamed like the fp32 form's.
namespace mfcc_fused1024_w12bf {

using namespace mfcc_fused1024;
using mfcc_fused1024_w12::FetchN;
using mfcc_fused1024_w12::fetch_window_n;
using mfcc_fused1024_w12::park_window_n;
using mfcc_fused1024_w12::cursor_of;
using mfcc_fused1024_w12::kW12Waves;
using mfcc_fused1024_w12::kQSlots;
using mfcc_fused1024_w12::kQGroupWords;
using mfcc_fused1024_w12::kGroupWords;
using mfcc_fused1024_w12::kCRow;
using mfcc_fused1024_w12::kW12LdsWords;

static_assert(kPieces == mfcc_fused1024_f32::kPieces && kTFrame == mfcc_fused1024_f32::kTFrame &&
              kVStride == mfcc_fused1024_f32::kVStride && kSUsed == mfcc_fused1024_f32::kSUsed, "shared staging");

// operands of set st of table row wv: hi at byte 0, lo at byte 1024 of a 2-KB record, lane-private 16 bytes
template <int NS>
__device__ __forceinline__ void load_bf_burst(u32x4 (&ah)[NS], u32x4 (&al)[NS], const uint32_t *base, int voff) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const uint32_t *b = base + st * 512;           // scalar
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ah[st]) : "v"(voff), "s"(b));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(al[st]) : "v"(voff), "s"(b));
    }
}
template <int NS>
__device__ __forceinline__ void wait_bf_burst(u32x4 (&ah)[NS], u32x4 (&al)[NS]) {
    static_assert(NS == 4 || NS == 5, "set lists");
    if constexpr (NS == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]));
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]), "+v"(al[4]));
}

template <int VAR>
__global__ __launch_bounds__(64 * kW12Waves) __attribute__((amdgpu_waves_per_eu(3, 3)))
void mfcc_fused1024_w12bf_kernel(mfcc_k::StreamDesc s, Tables t, LaunchGeom g, float *__restrict__ out) {
    using S = Sets<VAR>;
    constexpr int NS = S::N;
    __shared__ __attribute__((aligned(16))) float lds[kW12LdsWords];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;         // 0: A, 1: B, 2: helpers
    const int wi = wave & 3;
    const int lo = lane & 15;
    const int q = lane >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    float *const T = lds;
    auto Vt = [&](int gi) { return lds + kTile * kTFrame + gi * kGroupWords; };
    auto Qt = [&](int gi) { return lds + kTile * kTFrame + gi * kGroupWords + kTile * kVStride; };
    auto Sf = [&](int gi) { return lds + kTile * kTFrame + gi * kGroupWords + kTile * kVStride + kQGroupWords; };
    float *const WinC = lds + kTile * kTFrame + 2 * kGroupWords;
    float *const TwC = WinC + 32 * kCRow;
    int *const Flag = reinterpret_cast<int *>(TwC + 32 * kCRow);

    const unsigned nwg = gridDim.x;
    const unsigned bid = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    const unsigned va = 2u * bid, vb = va + 1u;
    const int n_tiles = g.tiles_per_ch * g.n_ch;
    const int gv = 2 * (int)gridDim.x;
    const int nA = (int)va < n_tiles ? (n_tiles - (int)va + gv - 1) / gv : 0;
    const int nB = (int)vb < n_tiles ? (n_tiles - (int)vb + gv - 1) / gv : 0;
    const int last_h = 2 * nA + 1;

    for (int i = tid; i < 32 * 32; i += 64 * kW12Waves) {
        WinC[(i >> 5) * kCRow + (i & 31)] = t.win[i];
        TwC[(i >> 5) * kCRow + (i & 31)] = t.tw[i];
    }
    if (tid < 2) Flag[tid] = 0;

    if (grp < 2) {
        // =========================================================================== workers
        const int gi = grp;
        const int n2 = lane & 31;
        float *const V = Vt(gi), *const Q = Qt(gi), *const S1 = Sf(gi);
        const int n_mine = gi ? nB : nA;
        const int n_other = gi ? nA : nB;
        Cursor cur = cursor_of(s, g, gi ? vb : va);
        const uint32_t *const abase = t.a_bf4 + (size_t)(2 * wi) * NS * 512;      // uniform; rows 2 wi (h = 0), 2 wi + 1
        const int lane16 = lane * 16;

        auto pass1 = [&](int i) {
            __builtin_amdgcn_s_setprio(MFCC_1K12_PRIO_P1);
            const int shift = window_of(cur, g).shift;
            advance(cur, g);
            const int fr0 = 4 * wi + (lane >> 5);
            v2f ep[16];
            {
                const float *sp = S1 + fr0 * kHop + n2 + shift;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[32 * n1];
            }
            v2f wp[16], tw[16];
            {
                const f32x4 *w4 = reinterpret_cast<const f32x4 *>(WinC + n2 * kCRow);
                const f32x4 *t4 = reinterpret_cast<const f32x4 *>(TwC + n2 * kCRow);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 a = w4[j], b = t4[j];
                    wp[2 * j] = (v2f){a[0], a[1]};
                    wp[2 * j + 1] = (v2f){a[2], a[3]};
                    tw[2 * j] = (v2f){b[0], b[1]};
                    tw[2 * j + 1] = (v2f){b[2], b[3]};
                }
            }
            v2f ty[16];
            float y16;
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
            {
                const float *sp = S1 + (fr0 + 2) * kHop + n2 + shift;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[32 * n1];
            }
            const int need = 4 * (gi ? i + 1 : (i < n_other ? i : n_other));
            while (__hip_atomic_load(Flag + (gi ^ 1), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - need < 0)
                __builtin_amdgcn_s_sleep(1);
            {
                v2f *tcol0 = reinterpret_cast<v2f *>(T + fr0 * kTFrame) + n2;
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
                V[fr0 * kVStride + n2] = y16;
            }
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
            {
                v2f *tcol0 = reinterpret_cast<v2f *>(T + (fr0 + 2) * kTFrame) + n2;
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
                V[(fr0 + 2) * kVStride + n2] = y16;
            }
        };
        auto half = [&](const v2f (&pp)[8], u32x4 (&ah)[NS], u32x4 (&al)[NS], f32x4 (&acc)[NS]) {
            float pw[16];
#pragma unroll
            for (int m = 0; m < 8; ++m) pw[m] = pp[m].x, pw[m + 8] = pp[m].y;
            u32x4 ph[2], pl[2];
#pragma unroll
            for (int gk = 0; gk < 2; ++gk)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t hi, lw;
                    split_bf16_pair(pw[kGrpM[gk][2 * d]], pw[kGrpM[gk][2 * d + 1]], hi, lw);
                    ph[gk][d] = hi;
                    pl[gk][d] = lw;
                }
            __builtin_amdgcn_sched_barrier(0);
            wait_bf_burst<NS>(ah, al);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    const u32x4 &a = term == 2 ? al[st] : ah[st];
                    const u32x4 &b = term == 1 ? pl[S::grp[st]] : ph[S::grp[st]];
                    acc[st] = MFCC1K_MFMA_BF(a, b, acc[st]);
                }
        };
        auto pass2 = [&]() {
            __builtin_amdgcn_s_setprio(MFCC_1K12_PRIO_P2);
            v2f xl[16], xh[16], pp[8];
            const f32x4 *trow = reinterpret_cast<const f32x4 *>(T + lo * kTFrame + (4 * wi + q) * kTRow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 a = trow[i], b = trow[8 + i];
                xl[2 * i] = (v2f){a[0], a[1]};
                xl[2 * i + 1] = (v2f){a[2], a[3]};
                xh[2 * i] = (v2f){b[0], b[1]};
                xh[2 * i + 1] = (v2f){b[2], b[3]};
            }
            if (lane == 0) __hip_atomic_fetch_add(Flag + gi, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            u32x4 ah[NS], al[NS];
            f32x4 acc[NS];
#pragma unroll
            for (int st = 0; st < NS; ++st) acc[st] = zero;
            load_bf_burst<NS>(ah, al, abase, lane16);
            __builtin_amdgcn_sched_barrier(0);
            mfcc_codelets::cfft32_h0_pow(xl, xh, pp);
            half(pp, ah, al, acc);
            __builtin_amdgcn_sched_barrier(0);
            load_bf_burst<NS>(ah, al, abase + NS * 512, lane16);
            __builtin_amdgcn_sched_barrier(0);
            mfcc_codelets::cfft32_h1_pow(xl, xh, pp);
            half(pp, ah, al, acc);
            f32x4 fin[kBlocks] = {zero, zero, zero};
#pragma unroll
            for (int st = 0; st < NS; ++st) fin[S::blk[st]] += acc[st];
#pragma unroll
            for (int b = 0; b < kBlocks; ++b)
                *reinterpret_cast<f32x4 *>(Q + ((wi * kBlocks + b) * 64 + lane) * 4) = fin[b];
        };
        lds_barrier();
        int bars = last_h + 1;
        if (gi) {
            lds_barrier();
            --bars;
        }
        for (int i = 0; i < n_mine; ++i) {
            pass1(i);
            lds_barrier();
            pass2();
            lds_barrier();
            bars -= 2;
        }
        for (; bars > 0; --bars) lds_barrier();
    } else if (wi < 2) {
        // =========================================================================== parkers (waves 8, 9)
        const int u = wi * 64 + lane;
        __builtin_amdgcn_s_setprio(3);
        Cursor pa = cursor_of(s, g, va), pb = cursor_of(s, g, vb);
        int ka = 0, kb = 0;
        FetchN fa, fb;
        bool have_a = false, have_b = false;
        if (nA > 0) {
            fetch_window_n(s, window_of(pa, g), u, fa);
            park_window_n(Sf(0), u, fa);
            advance(pa, g);
            ++ka;
        }
        if (nB > 0) {
            fetch_window_n(s, window_of(pb, g), u, fb);
            park_window_n(Sf(1), u, fb);
            advance(pb, g);
            ++kb;
        }
        if (ka < nA) {
            fetch_window_n(s, window_of(pa, g), u, fa);
            advance(pa, g);
            ++ka;
            have_a = true;
        }
        if (kb < nB) {
            fetch_window_n(s, window_of(pb, g), u, fb);
            advance(pb, g);
            ++kb;
            have_b = true;
        }
        lds_barrier();
        for (int h = 0; h <= last_h; ++h) {
            if (h & 1) {
                if (have_a) park_window_n(Sf(0), u, fa);
                have_a = false;
                if (ka < nA) {
                    fetch_window_n(s, window_of(pa, g), u, fa);
                    advance(pa, g);
                    ++ka;
                    have_a = true;
                }
            } else if (h >= 2) {
                if (have_b) park_window_n(Sf(1), u, fb);
                have_b = false;
                if (kb < nB) {
                    fetch_window_n(s, window_of(pb, g), u, fb);
                    advance(pb, g);
                    ++kb;
                    have_b = true;
                }
            }
            lds_barrier();
        }
    } else if (wi == 2) {
        // =========================================================================== column 16 (wave 10)
        __builtin_amdgcn_s_setprio(3);
        float a1[kAextra], a2[kAextra];
#pragma unroll
        for (int i = 0; i < kAextra; ++i) {
            a1[i] = t.a_extra[(1 * kAextra + i) * 64 + lane];
            a2[i] = t.a_extra[(2 * kAextra + i) * 64 + lane];
        }
        lds_barrier();
        for (int h = 0; h <= last_h; ++h) {
            const int gi = (h & 1) ? 0 : 1;
            const int k = (h & 1) ? (h - 1) / 2 : h / 2 - 1;
            if (k >= 0 && k < (gi ? nB : nA)) {
                const float *V = Vt(gi) + lo * kVStride + q;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = V[4 * j];
                f32x4 s1 = zero, s1b = zero, s2 = zero, s2b = zero;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    s1 = MFCC1K_MFMA(a1[j], v[j], s1);
                    s2 = MFCC1K_MFMA(a2[j], v[j], s2);
                    s1b = MFCC1K_MFMA(a1[j + 1], v[j + 1], s1b);
                    s2b = MFCC1K_MFMA(a2[j + 1], v[j + 1], s2b);
                }
                s1 += s1b;
                s2 += s2b;
                const float p10 = fmaf(s1[0], s1[0], s1[1] * s1[1]), p11 = fmaf(s1[2], s1[2], s1[3] * s1[3]);
                const float p20 = fmaf(s2[0], s2[0], s2[1] * s2[1]), p21 = fmaf(s2[2], s2[2], s2[3] * s2[3]);
                f32x4 fin[kBlocks] = {zero, zero, zero};
#pragma unroll
                for (int b = 0; b < kBlocks; ++b) {
                    if (S::c16[0][b]) {
                        fin[b] = MFCC1K_MFMA(a1[8 + 2 * b], p10, fin[b]);
                        fin[b] = MFCC1K_MFMA(a1[9 + 2 * b], p11, fin[b]);
                    }
                    if (S::c16[1][b]) {
                        fin[b] = MFCC1K_MFMA(a2[8 + 2 * b], p20, fin[b]);
                        fin[b] = MFCC1K_MFMA(a2[9 + 2 * b], p21, fin[b]);
                    }
                }
                float *Q = Qt(gi);
#pragma unroll
                for (int b = 0; b < kBlocks; ++b)
                    *reinterpret_cast<f32x4 *>(Q + ((4 * kBlocks + b) * 64 + lane) * 4) = fin[b];
            }
            lds_barrier();
        }
    } else {
        // =========================================================================== tail (wave 11)
        __builtin_amdgcn_s_setprio(3);
        float ax[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) ax[i] = t.a_extra[(0 * kAextra + i) * 64 + lane];
        const int lane_off = lo * t.n_cep + 4 * q;
        Cursor ta = cursor_of(s, g, va), tb = cursor_of(s, g, vb);
        lds_barrier();
        for (int h = 0; h <= last_h; ++h) {
            const int gi = (h & 1) ? 1 : 0;
            const int k = (h & 1) ? (h - 3) / 2 : h / 2 - 1;
            if (h >= 2 && k >= 0 && k < (gi ? nB : nA)) {
                const f32x4 *Q4 = reinterpret_cast<const f32x4 *>(Qt(gi)) + lane;
                f32x4 lm[kBlocks];
#pragma unroll
                for (int b = 0; b < kBlocks; ++b) {
                    const f32x4 m = ((Q4[(0 * kBlocks + b) * 64] + Q4[(1 * kBlocks + b) * 64]) +
                                     (Q4[(2 * kBlocks + b) * 64] + Q4[(3 * kBlocks + b) * 64])) + Q4[(4 * kBlocks + b) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) lm[b][r] = __builtin_amdgcn_logf(m[r]);
                }
                if (q >= 2) lm[2] = zero;
                f32x4 d[kBlocks] = {zero, zero, zero};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int b = 0; b < kBlocks; ++b) d[b] = MFCC1K_MFMA(ax[4 * b + r], lm[b][r], d[b]);
                if (gi) {
                    dct_store(s, t, d, lm, tb, lo, q, lane, lane_off, out);
                    advance(tb, g);
                } else {
                    dct_store(s, t, d, lm, ta, lo, q, lane, lane_off, out);
                    advance(ta, g);
                }
            }
            lds_barrier();
        }
    }
}

inline const char *kernel_name() { return "mfcc_fused1024_w12bf_kernel"; }

inline bool launch(const mfcc_k::StreamDesc &s, const Tables &t, float *out, int n_cu, hipStream_t stream) {
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    if (n_tiles >= (1ll << 30) || tiles_per_ch >= (1ll << 26) || n_ch >= (1ll << 30)) return false;
    long long wgs = (n_tiles + 1) / 2;
    if (wgs > n_cu) wgs = n_cu;
    if (wgs < 1) wgs = 1;
    const long long grid = 2 * wgs;
    LaunchGeom g;
    g.tiles_per_ch = (int)tiles_per_ch;
    g.n_ch = (int)n_ch;
    g.grid_div = (int)(grid / tiles_per_ch);
    g.grid_mod = (int)(grid % tiles_per_ch);
    g.step_ptr = (long long)g.grid_div * s.ch_stride + (long long)g.grid_mod * kTileHop;
    g.wrap_ptr = s.ch_stride - tiles_per_ch * (long long)kTileHop;
    g.t_lo = (int)((9 - (long long)s.halo + kTileHop - 1) / kTileHop);
    if (g.t_lo < 0) g.t_lo = 0;
    const long long hi = (s.n_samples - kSUsed) / kTileHop;
    g.t_hi = s.n_samples < kSUsed ? -1 : (int)(hi < tiles_per_ch ? hi : tiles_per_ch);
    const dim3 grid3((unsigned)wgs), block3(64 * kW12Waves);
    switch (t.variant) {
    case 1: hipLaunchKernelGGL(mfcc_fused1024_w12bf_kernel<1>, grid3, block3, 0, stream, s, t, g, out); break;
    case 2: hipLaunchKernelGGL(mfcc_fused1024_w12bf_kernel<2>, grid3, block3, 0, stream, s, t, g, out); break;
    default: hipLaunchKernelGGL(mfcc_fused1024_w12bf_kernel<0>, grid3, block3, 0, stream, s, t, g, out); break;
    }
    return true;
}

}  // namespace mfcc_fused1024_w12bf
