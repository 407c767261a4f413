// NOT COMPILED INTO THE LIBRARY: the 32-frame-tile staging of the fused 512 kernel, built and measured in round 3 (DESIGN.md
// 4.1: correct against the parity tests, 0.987 ms against the 16-frame kernel's 0.980) and kept here as the record of it.
// To run it again: copy next to kernel_fused512_w12.hpp, include it in mfcc_hip.hip and try mfcc_fused12x2::launch first.
// Fused 512/170/32 float kernel, twelve waves on 32-FRAME tiles: the arithmetic, codelets and tables of
// kernel_fused512.hpp and the staging of kernel_fused512_w12.hpp (read both headers first) with every phase twice as long.
//
// Why.  In the twelve-wave kernel a half-step is ~2 500 clocks, and what it loses is per half-step: every wave starts a
// phase on LDS reads at the same moment (the workers on their T columns, the helpers on V and Q), the waves of a SIMD
// finish apart and wait for the slowest at the barrier, and removing 12 % of the vector instructions (the bf16 split,
// timing-only build) bought 3.8 %.  Running every phase twice per half-step on the same data (timing only) took 1.776 ms
// for twice the work of 0.977: 9 % less per frame.  This kernel does that for real:
//
//   * a tile is 32 consecutive frames, two BATCHES of 16 (the N dimension of the matrix instructions); a worker runs
//     pass 1 twice (its four frames of batch 0, then of batch 1) and pass 2 twice per tile;
//   * ONE transpose tile T for both groups, two halves of 16 frames (two tiles of 32 do not fit beside the windows):
//     the group in pass 1 stores batch b behind the partner group's read of half b -- a pass-2 wave bumps an LDS counter
//     behind its 8 reads of a half (the LDS executes a wave's instructions in order), a pass-1 wave polls it before its
//     first store to that half (one read that practically always succeeds);
//   * ONE partial-sum area Q for both groups: the tail reads the sums of the previous half-step at the start of a
//     half-step and bumps a counter, the workers and column 16 poll it before they write theirs;
//   * the first batch's pass-1 operands are read one half-step early, in the middle of the group's pass 2 (as in the
//     16-frame kernel: the half-step starts on registers), the second batch's while the first batch's columns are stored;
//     a group's window is re-parked while the group runs pass 1 -- its first 2 048 slots at once (only the first batch
//     reads below slot 2 720, and it holds its operands already), the rest behind a counter the workers bump after their
//     second batch's reads.
//
// Roles and the half-step schedule are those of kernel_fused1024_w12.hpp:
//   group A: pass 1 of its tile k at h = 2 k,     pass 2 + mel at h = 2 k + 1
//   group B: pass 1 of its tile k at h = 2 k + 1, pass 2 + mel at h = 2 k + 2
//   parkers (waves 8, 9): S_A(h / 2 + 1) at even h, S_B((h + 1) / 2) at odd h, fetched two half-steps ahead
//   column 16 (wave 10) at h: the group in pass 2;  tail (wave 11) at h: the group that was in pass 2 at h - 1
// Not here: the integer DC chain (a filter with weight on bin 0: 44.1 / 48 kHz) and ragged corpora -- those run on
// kernel_fused512_w12.hpp.  LDS 146 KB.
#pragma once

#include "kernel_fused512.hpp"

namespace mfcc_fused12x2 {

using namespace mfcc_fused;

constexpr int kW12Waves = 12;
constexpr int kTile2 = 32;
constexpr int kTileHop2 = kTile2 * kHop;            // 5440 samples between consecutive tiles
constexpr int kParkers = 128, kParkPieces = 6;      // waves 8, 9: 128 lanes x 6 pieces of 8 samples
constexpr int kSUsed2 = 8 * kParkers * kParkPieces; // 6144 fp32 slots (7 + 31 * 170 + 512 = 5789 are read)
static_assert(7 + (kTile2 - 1) * kHop + kNfft <= kSUsed2, "window");
constexpr int kQSlots = 5;                          // 4 workers + column 16
constexpr int kQBatch = kQSlots * 2 * 256;          // [slot][block][lane * 4]
constexpr int kQWords2 = 2 * kQBatch;               // both batches, ONE area for both groups
constexpr int kGroupWords2 = kTile2 * kVStride + kSUsed2;
constexpr int kLdsWords2 = kTile2 * kTFrame + kQWords2 + 2 * kGroupWords2 + 8;
static_assert(kLdsWords2 * 4 <= 160 * 1024, "LDS");

struct Fetch6 {
    i32x4 v[kParkPieces];
    int p[kParkPieces];          // dword in front of v[k]: its high half is the piece's predecessor sample
};

__device__ __forceinline__ void fetch_window6(const mfcc_k::StreamDesc &s, const Window &w, int u, Fetch6 &f) {
    if (w.inside) {
        const i32x4 *g = reinterpret_cast<const i32x4 *>(w.ptr - w.shift);
        const int *g32 = reinterpret_cast<const int *>(g);
#pragma unroll
        for (int k = 0; k < kParkPieces; ++k) {
            f.v[k] = g[k * kParkers + u];
            f.p[k] = g32[4 * (k * kParkers + u) - 1];
        }
    } else {
        const long long first = (long long)w.t_in * kTileHop2;     // channel-relative
        const int16_t *base = w.ptr - first;
#pragma unroll
        for (int k = 0; k < kParkPieces; ++k) {
            int h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = mfcc_k::sample_at_i(s, base, first + 8 * (k * kParkers + u) + j) & 0xFFFF;
            f.v[k] = (i32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            f.p[k] = mfcc_k::sample_at_i(s, base, first + 8 * (k * kParkers + u) - 1) << 16;
        }
    }
}

template <int K0, int K1>
__device__ __forceinline__ void park_window6(float *Sf, int u, const Fetch6 &f) {
#pragma unroll
    for (int k = K0; k < K1; ++k) preemph8(f.p[k], f.v[k], Sf + 8 * (k * kParkers + u));
}
constexpr int kParkEarly = 2;                       // pieces 0 .. 255: slots below 2 048 (the second batch reads from 2 720 on)
static_assert(8 * kParkEarly * kParkers <= 16 * kHop, "early part of the window");

__device__ __forceinline__ Cursor cursor_of(const mfcc_k::StreamDesc &s, const LaunchGeom &g, unsigned v) {
    Cursor c;
    c.ch = (int)(v / (unsigned)g.tiles_per_ch);
    c.t_in = (int)(v - (unsigned)c.ch * (unsigned)g.tiles_per_ch);
    c.ptr = s.pcm + (long long)c.ch * s.ch_stride + (long long)c.t_in * kTileHop2;
    return c;
}

// the counters: waits are one acquire load that practically always succeeds
__device__ __forceinline__ void wait_count(const int *cnt, int need) {
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - need < 0) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void bump_count(int *cnt, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// rows of batch b of a finished tile (coefficients 0..15 summed by the caller's MFMAs, 16..31 here)
__device__ __forceinline__ void dct_store32(const mfcc_k::StreamDesc &s, const FusedTables &t, const f32x4 &l0,
                                            const f32x4 &l1, const f32x4 &d0, const f32x4 &d1, const float (&ax)[kAextra],
                                            const Cursor &c, int b, int lo, int q, int lane_off, float *__restrict__ out) {
    const long long fr0 = (long long)c.t_in * kTile2 + 16 * b;
    const long long rows_left = s.frames_per_ch - fr0;
    float *o = out + ((long long)c.ch * s.frames_per_ch + fr0) * t.n_cep + lane_off;
    if (lo < rows_left) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < t.n_cep) o[r] = d0[r] + d1[r];
    }
    if (t.n_cep > 16) {                        // coefficients 16..31: a second M tile (uniform branch)
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e0 = MFCC_MFMA(ax[8 + r], l0[r], e0);
            e1 = MFCC_MFMA(ax[12 + r], l1[r], e1);
        }
        if (lo < rows_left) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 + 4 * q + r < t.n_cep) o[16 + r] = e0[r] + e1[r];
        }
    }
}

#ifndef MFCC_W12X2_PRIO_P1
#define MFCC_W12X2_PRIO_P1 1
#endif
#ifndef MFCC_W12X2_PRIO_P2
#define MFCC_W12X2_PRIO_P2 0
#endif

template <bool DENSE>
__global__ __launch_bounds__(64 * kW12Waves) __attribute__((amdgpu_waves_per_eu(3, 3)))
void mfcc_fused512_w12x2_kernel(mfcc_k::StreamDesc s, FusedTables t, LaunchGeom g, float *__restrict__ out) {
    constexpr int kSets = SetsBf<DENSE>::N;
    __shared__ __attribute__((aligned(16))) float lds[kLdsWords2];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;         // 0: A, 1: B, 2: helpers
    const int wi = wave & 3;
    const int lo = lane & 15;
    const int q = lane >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    float *const T = lds;                                                  // [32 frames][kTFrame]
    float *const Q = lds + kTile2 * kTFrame;                               // [batch][slot][block][256]
    auto Vt = [&](int gi) { return lds + kTile2 * kTFrame + kQWords2 + gi * kGroupWords2; };
    auto Sf = [&](int gi) { return lds + kTile2 * kTFrame + kQWords2 + gi * kGroupWords2 + kTile2 * kVStride; };
    int *const Cnt = reinterpret_cast<int *>(lds + kTile2 * kTFrame + kQWords2 + 2 * kGroupWords2);
    // Cnt[2 gi + b]: pass-2 waves of group gi that have read half b of T (4 per tile); Cnt[4]: tiles the tail has read out of
    // Q; Cnt[5 + gi]: pass-1 waves of group gi that have read their second batch's operands out of S (4 per tile)

    const unsigned nwg = gridDim.x;
    const unsigned bid = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    const unsigned va = 2u * bid, vb = va + 1u;
    const int n_tiles = g.tiles_per_ch * g.n_ch;                          // < 2^30 (host check)
    const int gv = 2 * (int)gridDim.x;
    const int nA = (int)va < n_tiles ? (n_tiles - (int)va + gv - 1) / gv : 0;
    const int nB = (int)vb < n_tiles ? (n_tiles - (int)vb + gv - 1) / gv : 0;
    const int last_h = 2 * nA + 1;                                        // B's last tail (nB <= nA) is at 2 nB + 1
    if (tid < 8) Cnt[tid] = 0;

    if (grp < 2) {
        // =========================================================================== workers
        const int gi = grp;
        const int fr_id = wi + 8 * (q & 1) + 4 * (q >> 1);
        using mfcc_codelets::v2f;
        v2f wp[16], tw[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) wp[i] = reinterpret_cast<const v2f *>(t.win)[lo * 16 + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) tw[i] = reinterpret_cast<const v2f *>(t.tw)[lo * 16 + i];
        u32x4 ah[kSets], al[kSets];
#pragma unroll
        for (int st = 0; st < kSets; ++st)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                ah[st][d] = t.a_mel_bf[((wi * kSets + st) * 2 + 0) * 256 + d * 64 + lane];
                al[st][d] = t.a_mel_bf[((wi * kSets + st) * 2 + 1) * 256 + d * 64 + lane];
            }
        float *const V = Vt(gi), *const S = Sf(gi);
        const int n_mine = gi ? nB : nA;
        const int n_other = gi ? nA : nB;
        Cursor cur = cursor_of(s, g, gi ? vb : va);

        // batch 0's operands of a tile are read out of its window ONE half-step early, in the middle of the group's pass 2
        // of the previous tile: pass 1 then starts on registers while the partner group reads its T columns
        v2f ep[16];
        int shift = 0;
        auto load_ep = [&]() {
            shift = window_of(cur, g).shift;
            const float *sp = S + fr_id * kHop + lo + shift;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[16 * n1];
        };
        auto pass1 = [&](int i) {
            // ---------------- pass 1: windowed real FFT-32 over n1, batch 0 then batch 1 (four frames each)
            __builtin_amdgcn_s_setprio(MFCC_W12X2_PRIO_P1);
            advance(cur, g);
            v2f ty[16];
            float y16;
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
            {
                // the second batch's operands fly while the first batch's columns are stored; behind them the parkers may
                // overwrite the rest of the window (the LDS executes this wave's reads before its counter bump)
                const float *sp = S + (16 + fr_id) * kHop + lo + shift;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[16 * n1];
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(Cnt + 5 + gi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
            }
            // T still holds the partner group's tile until its four pass-2 waves have read the half
            const int need = 4 * (gi ? i + 1 : (i < n_other ? i : n_other));
            wait_count(Cnt + 2 * (gi ^ 1) + 0, need);
            {
                v2f *tcol0 = reinterpret_cast<v2f *>(T + fr_id * kTFrame) + lo;        // a store's lanes: consecutive n2
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
                V[fr_id * kVStride + lo] = y16;
            }
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
            wait_count(Cnt + 2 * (gi ^ 1) + 1, need);
            {
                v2f *tcol0 = reinterpret_cast<v2f *>(T + (16 + fr_id) * kTFrame) + lo;
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
                V[(16 + fr_id) * kVStride + lo] = y16;
            }
        };
        auto pass2 = [&](int i) {
            // ---------------- pass 2: complex FFT-16 over n2 for frame 16 b + lo, column k1 = 4 wi + q; mel MFMAs
            __builtin_amdgcn_s_setprio(MFCC_W12X2_PRIO_P2);
            // tiles the tail must have read out of Q before this tile's sums may go there
            const int q_need = gi ? 2 * i + 1 : i + (i < n_other ? i : n_other);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float pw[16];
                {
                    v2f x[16], pp[8];
                    const f32x4 *trow = reinterpret_cast<const f32x4 *>(T + (16 * b + lo) * kTFrame + (4 * wi + q) * kTRow);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f32x4 a = trow[j];
                        x[2 * j] = (v2f){a[0], a[1]};
                        x[2 * j + 1] = (v2f){a[2], a[3]};
                    }
                    bump_count(Cnt + 2 * gi + b, lane);
                    mfcc_codelets::cfft16_pow(x, pp);
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) pw[k2] = pp[k2].x, pw[k2 + 8] = pp[k2].y;
                }
                PowerBf pb;
                split_power(pw, pb);
                if (b == 1) load_ep();                 // the next tile's first operands fly during the MFMAs
                f32x4 acc[kSets];
#pragma unroll
                for (int st = 0; st < kSets; ++st) acc[st] = zero;
                mel_bf_all<DENSE, 0>(ah, al, pb, acc, [](auto) {});
                f32x4 b0, b1;
                mel_bf_blocks<DENSE>(acc, b0, b1);
                if (b == 0) wait_count(Cnt + 4, q_need);
                *reinterpret_cast<f32x4 *>(Q + b * kQBatch + (2 * wi + 0) * 256 + lane * 4) = b0;
                *reinterpret_cast<f32x4 *>(Q + b * kQBatch + (2 * wi + 1) * 256 + lane * 4) = b1;
            }
        };
        lds_barrier();                                 // prologue: S_A(0), S_B(0) and the counters are in LDS
        if (n_mine > 0) load_ep();                     // tile 0 (group B idles through h = 0 with its operands loaded)
        lds_barrier();                                 // second prologue barrier: the parkers may now re-park S_A (h = 0)
        int bars = last_h + 1;                         // every wave of the workgroup passes this many barriers
        if (gi) {                                      // h = 0: group B idles
            lds_barrier();
            --bars;
        }
        for (int i = 0; i < n_mine; ++i) {
            pass1(i);
            lds_barrier();
            pass2(i);
            lds_barrier();
            bars -= 2;
        }
        for (; bars > 0; --bars) lds_barrier();
    } else if (wi < 2) {
        // =========================================================================== parkers (waves 8, 9)
        const int u = wi * 64 + lane;                  // 0..127
        __builtin_amdgcn_s_setprio(3);
        Cursor pa = cursor_of(s, g, va), pb = cursor_of(s, g, vb);
        int ka = 0, kb = 0;                            // next tile of each stream to fetch
        Fetch6 fa, fb;
        bool have_a = false, have_b = false;
        if (nA > 0) {                                  // prologue: S_A(0) and S_B(0) directly
            fetch_window6(s, window_of(pa, g), u, fa);
            park_window6<0, kParkPieces>(Sf(0), u, fa);
            advance(pa, g);
            ++ka;
        }
        if (nB > 0) {
            fetch_window6(s, window_of(pb, g), u, fb);
            park_window6<0, kParkPieces>(Sf(1), u, fb);
            advance(pb, g);
            ++kb;
        }
        if (ka < nA) {                                 // S_A(1): parked at h = 0
            fetch_window6(s, window_of(pa, g), u, fa);
            advance(pa, g);
            ++ka;
            have_a = true;
        }
        if (kb < nB) {                                 // S_B(1): parked at h = 1
            fetch_window6(s, window_of(pb, g), u, fb);
            advance(pb, g);
            ++kb;
            have_b = true;
        }
        lds_barrier();                                 // S_A(0), S_B(0) are parked: the workers fetch tile 0's first operands
        lds_barrier();                                 // ... and hold them in registers: h = 0 may start
        for (int h = 0; h <= last_h; ++h) {
            // A window is re-parked in the half-step in which its group runs pass 1: S_A(h / 2 + 1) at even h (A is in
            // tile h / 2), S_B((h + 1) / 2) at odd h (B is in tile (h - 1) / 2).  The early part at once, the rest when the
            // group's four waves have read their second batch's operands of the tile they are in.
            if (!(h & 1)) {
                if (have_a) {
                    park_window6<0, kParkEarly>(Sf(0), u, fa);
                    wait_count(Cnt + 5 + 0, 4 * (h / 2 + 1));
                    park_window6<kParkEarly, kParkPieces>(Sf(0), u, fa);
                }
                have_a = false;
                if (ka < nA) {
                    fetch_window6(s, window_of(pa, g), u, fa);
                    advance(pa, g);
                    ++ka;
                    have_a = true;
                }
            } else {
                if (have_b) {
                    park_window6<0, kParkEarly>(Sf(1), u, fb);
                    wait_count(Cnt + 5 + 1, 4 * ((h - 1) / 2 + 1));
                    park_window6<kParkEarly, kParkPieces>(Sf(1), u, fb);
                }
                have_b = false;
                if (kb < nB) {
                    fetch_window6(s, window_of(pb, g), u, fb);
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
        float ax[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ax[i] = t.a_extra[(1 * kAextra + i) * 64 + lane];
        lds_barrier();
        lds_barrier();                                 // the parkers' two prologue barriers
        for (int h = 0; h <= last_h; ++h) {
            // the group in pass 2 at h: A (tile (h - 1) / 2) for odd h, B (tile h / 2 - 1) for even h >= 2
            const int gi = (h & 1) ? 0 : 1;
            const int k = (h & 1) ? (h - 1) / 2 : h / 2 - 1;
            if (k >= 0 && k < (gi ? nB : nA)) {
                const int n_other = gi ? nA : nB;
                const int q_need = gi ? 2 * k + 1 : k + (k < n_other ? k : n_other);
                const float *V = Vt(gi);
                f32x4 r[2][2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float *vp = V + (16 * b + lo) * kVStride + q;
                    const float v0 = vp[0], v1 = vp[4], v2 = vp[8], v3 = vp[12];
                    f32x4 sp = MFCC_MFMA(ax[0], v0, zero);
                    f32x4 sp2 = MFCC_MFMA(ax[1], v1, zero);
                    sp = MFCC_MFMA(ax[2], v2, sp);
                    sp2 = MFCC_MFMA(ax[3], v3, sp2);
                    sp += sp2;
                    const float s0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);      // bin 16 + 64 q
                    const float s1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);      // bin 48 + 64 q
                    const f32x4 x0 = MFCC_MFMA(ax[4], s0, zero), y0 = MFCC_MFMA(ax[5], s1, zero);
                    const f32x4 x1 = MFCC_MFMA(ax[6], s0, zero), y1 = MFCC_MFMA(ax[7], s1, zero);
                    r[b][0] = x0 + y0;
                    r[b][1] = x1 + y1;
                }
                wait_count(Cnt + 4, q_need);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    *reinterpret_cast<f32x4 *>(Q + b * kQBatch + (2 * 4 + 0) * 256 + lane * 4) = r[b][0];
                    *reinterpret_cast<f32x4 *>(Q + b * kQBatch + (2 * 4 + 1) * 256 + lane * 4) = r[b][1];
                }
            }
            lds_barrier();
        }
    } else {
        // =========================================================================== tail (wave 11)
        __builtin_amdgcn_s_setprio(3);
        float ax[kAextra];
#pragma unroll
        for (int i = 0; i < kAextra; ++i) ax[i] = t.a_extra[(0 * kAextra + i) * 64 + lane];
        const int lane_off = lo * t.n_cep + 4 * q;
        Cursor ta = cursor_of(s, g, va), tb = cursor_of(s, g, vb);
        lds_barrier();
        lds_barrier();
        for (int h = 0; h <= last_h; ++h) {
            // the group that was in pass 2 at h - 1: A (tile h / 2 - 1) for even h, B (tile (h - 3) / 2) for odd h
            const int gi = (h & 1) ? 1 : 0;
            const int k = (h & 1) ? (h - 3) / 2 : h / 2 - 1;
            if (h >= 2 && k >= 0 && k < (gi ? nB : nA)) {
                const f32x4 *Q4 = reinterpret_cast<const f32x4 *>(Q) + lane;
                f32x4 m[2][2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        const f32x4 *qb = Q4 + b * (kQBatch / 4) + blk * 64;
                        m[b][blk] = ((qb[0 * 128] + qb[1 * 128]) + (qb[2 * 128] + qb[3 * 128])) + qb[4 * 128];
                    }
                bump_count(Cnt + 4, lane);             // Q is read: the group in pass 2 now may write its own sums
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 l0, l1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        l0[r] = __builtin_amdgcn_logf(m[b][0][r]);
                        l1[r] = __builtin_amdgcn_logf(m[b][1][r]);
                    }
                    if (t.n_mel <= 16) l1 = zero;      // no filters 16..31 (uniform)
                    f32x4 d0 = zero, d1 = zero;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d0 = MFCC_MFMA(ax[r], l0[r], d0);
                        d1 = MFCC_MFMA(ax[4 + r], l1[r], d1);
                    }
                    if (gi) dct_store32(s, t, l0, l1, d0, d1, ax, tb, b, lo, q, lane_off, out);
                    else dct_store32(s, t, l0, l1, d0, d1, ax, ta, b, lo, q, lane_off, out);
                }
                if (gi) advance(tb, g);
                else advance(ta, g);
            }
            lds_barrier();
        }
    }
}

inline const char *kernel_name() { return "mfcc_fused512_w12x2_kernel"; }

// returns false when the problem does not fit or needs what only kernel_fused512_w12.hpp has (the integer DC chain)
inline bool launch(const mfcc_k::StreamDesc &s, const FusedTables &t, bool dense, float *out, int n_cu,
                   hipStream_t stream) {
    if (t.win_dc != nullptr) return false;
    const long long tiles_per_ch = (s.frames_per_ch + kTile2 - 1) / kTile2;
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
    g.step_ptr = (long long)g.grid_div * s.ch_stride + (long long)g.grid_mod * kTileHop2;
    g.wrap_ptr = s.ch_stride - tiles_per_ch * (long long)kTileHop2;
    g.t_lo = (int)((9 - (long long)s.halo + kTileHop2 - 1) / kTileHop2);
    if (g.t_lo < 0) g.t_lo = 0;
    const long long hi = (s.n_samples - kSUsed2) / kTileHop2;
    g.t_hi = s.n_samples < kSUsed2 ? -1 : (int)(hi < tiles_per_ch ? hi : tiles_per_ch);
    if (dense)
        hipLaunchKernelGGL((mfcc_fused512_w12x2_kernel<true>), dim3((unsigned)wgs), dim3(64 * kW12Waves), 0, stream, s, t, g, out);
    else
        hipLaunchKernelGGL((mfcc_fused512_w12x2_kernel<false>), dim3((unsigned)wgs), dim3(64 * kW12Waves), 0, stream, s, t, g, out);
    return true;
}

}  // namespace mfcc_fused12x2
