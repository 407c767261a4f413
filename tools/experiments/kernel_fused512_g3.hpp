// NOT COMPILED INTO THE LIBRARY: the three-worker-group staging of the fused 512 kernel, built and measured in round 3
// (correct against the parity tests and bit-identical to the twelve-wave kernel, 1.61 ms against its 1.00).
// It needs two lane-major operand tables next to FusedTables' (a_mel_bf4, a_extra4: see load_mel_burst / load_role_burst).
// Fused 512/170/32 float kernel, THREE worker groups: the arithmetic, codelets and tables of kernel_fused512.hpp (read its
// header first) on twelve waves that are ALL workers -- every SIMD always has three busy waves.
//
// Why.  In the twelve-wave kernel (kernel_fused512_w12.hpp) a SIMD has two busy waves and a helper that mostly waits.  A wave
// issues one instruction per 5-6 clocks and its phases are serial chains, so two of them keep the vector pipe 60 % busy;
// removing 12 % of the vector instructions bought 3.8 %.  A timing-only build with a THIRD worker group on the helper
// waves' slots did 1.5 x the FFT work in 1.33 x the time.  This kernel is that, with the helper duties
// folded into the workers the way the four-wave kernel has them:
//
//   waves 0..3   group A     waves 4..7   group B     waves 8..11  group C      (waves i, i + 4, i + 8 share SIMD i)
//   A and C run pass 1 at even steps and pass 2 at odd steps, B the other way round; ONE LDS-only barrier per step
//   wave 0 of a group: the tail of the group's previous tile (log2 at the start of pass 1, DCT-II MFMAs riding on its mel
//     chain in pass 2, store); wave 1: column 16 (DFT + mel MFMAs, a fifth summand); all four park the group's
//     next window during pass 2 (waves 2, 3 two pieces per lane, waves 0, 1 one) and fetch the one after it right behind
//
// What makes it fit (three waves per SIMD have 168 registers, and three T tiles + three windows do not fit the LDS):
//   * the MFMA operands are STREAMED: the mel weights of a wave (sets x (hi, lo) x 16 bytes per lane) and the role
//     operands of waves 0 / 1 are fetched from the L2-resident tables while pass 2's FFT runs -- one burst of
//     global_load_dwordx4 off a scalar base in inline asm, waited for by hand right before the MFMAs (left to the compiler
//     such loads sink to one load + s_waitcnt in front of each MFMA: kernel_fused1024_w12.hpp; round 2's attempt at three
//     four-wave workgroups per CU died of 33 dword loads per wave and tile);
//   * the twiddles are read from an LDS copy at the start of pass 1 (8 ds_read_b128), the window constants stay resident;
//   * A and B share ONE transpose tile: the group in pass 1 stores its columns behind the partner's reads of its own --
//     a pass-2 wave bumps an LDS counter behind its 8 reads (the LDS executes a wave's instructions in order), a pass-1
//     wave polls it before its first store (kernel_fused1024_w12.hpp).  C, in phase with A, has its own.
// LDS 145 KB: T_AB, T_C, per group V + Q (5 slots x 2 blocks) + S, the twiddles, two counters.
// Not here: the integer DC chain (a filter with weight on bin 0: 44.1 / 48 kHz) and ragged corpora -- those run on
// kernel_fused512_w12.hpp.
#pragma once

#include "kernel_fused512.hpp"

namespace mfcc_fused_g3 {

using namespace mfcc_fused;

constexpr int kG3Waves = 12;
constexpr int kQGroup = (kWaves + 1) * 2 * 256;      // [wave, then column 16][block][lane * 4]: the twelve-wave kernel's five
                                                     // summands in its order, so that both kernels give the same bits
constexpr int kCRow = 36;                            // words per n2 row of the twiddle table (9 x 16 B: conflict free)
constexpr int kGroupW = kTile * kVStride + kQGroup + kSUsed;
constexpr int kG3LdsWords = 2 * kTile * kTFrame + 3 * kGroupW + 16 * kCRow + 4;
static_assert(kG3LdsWords * 4 <= 160 * 1024, "LDS");
constexpr int kPiecesG3 = kSUsed / 8;                // 384 pieces of 8 samples per window
static_assert(kPiecesG3 == 64 * (1 + 1 + 2 + 2), "waves 0, 1 park one piece per lane, waves 2, 3 two");

struct Fetch2 {
    i32x4 v[2];
    int p[2];                    // dword in front of v[k]: its high half is the piece's predecessor sample
};

// this lane's pieces of a window: piece0 and, for waves 2 / 3, piece0 + 64
template <int N>
__device__ __forceinline__ void fetch_pieces(const mfcc_k::StreamDesc &s, const Window &w, int piece0, Fetch2 &f) {
    if (w.inside) {
        const i32x4 *g = reinterpret_cast<const i32x4 *>(w.ptr - w.shift);
        const int *g32 = reinterpret_cast<const int *>(g);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            f.v[k] = g[piece0 + 64 * k];
            f.p[k] = g32[4 * (piece0 + 64 * k) - 1];
        }
    } else {
        const long long first = (long long)w.t_in * kTileHop;      // channel-relative
        const int16_t *base = w.ptr - first;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            int h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = mfcc_k::sample_at_i(s, base, first + 8 * (piece0 + 64 * k) + j) & 0xFFFF;
            f.v[k] = (i32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            f.p[k] = mfcc_k::sample_at_i(s, base, first + 8 * (piece0 + 64 * k) - 1) << 16;
        }
    }
}
template <int N>
__device__ __forceinline__ void park_pieces(float *Sf, int piece0, const Fetch2 &f) {
#pragma unroll
    for (int k = 0; k < N; ++k) preemph8(f.p[k], f.v[k], Sf + 8 * (piece0 + 64 * k));
}

__device__ __forceinline__ Cursor cursor_of(const mfcc_k::StreamDesc &s, const LaunchGeom &g, unsigned v) {
    Cursor c;
    c.ch = (int)(v / (unsigned)g.tiles_per_ch);
    c.t_in = (int)(v - (unsigned)c.ch * (unsigned)g.tiles_per_ch);
    c.ptr = s.pcm + (long long)c.ch * s.ch_stride + (long long)c.t_in * kTileHop;
    return c;
}

// the streamed operands: one burst of 16-byte loads off scalar bases, waited for by hand
template <int NS>
__device__ __forceinline__ void load_mel_burst(u32x4 (&ah)[NS], u32x4 (&al)[NS], const uint32_t *base, int voff) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const uint32_t *b = base + st * 512;           // scalar: a 2-KB record per set, hi at byte 0, lo at byte 1024
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ah[st]) : "v"(voff), "s"(b));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(al[st]) : "v"(voff), "s"(b));
    }
}
__device__ __forceinline__ void load_role_burst(f32x4 (&ax4)[4], const float *base, int voff, bool second_half) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ax4[0]) : "v"(voff), "s"(base));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ax4[1]) : "v"(voff), "s"(base));
    if (second_half) {                                 // DCT rows of coefficients 16..31 (uniform)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(ax4[2]) : "v"(voff), "s"(base));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(ax4[3]) : "v"(voff), "s"(base));
    }
}
template <int NS>
__device__ __forceinline__ void wait_bursts(u32x4 (&ah)[NS], u32x4 (&al)[NS], f32x4 (&ax4)[4]) {
    static_assert(NS == 3 || NS == 4, "set lists");
    if constexpr (NS == 3)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(ax4[0]), "+v"(ax4[1]),
                       "+v"(ax4[2]), "+v"(ax4[3]));
    else
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]),
                       "+v"(ax4[0]), "+v"(ax4[1]), "+v"(ax4[2]), "+v"(ax4[3]));
}

template <bool DENSE>
__global__ __launch_bounds__(64 * kG3Waves) __attribute__((amdgpu_waves_per_eu(3, 3)))
void mfcc_fused512_g3_kernel(mfcc_k::StreamDesc s, FusedTables t, LaunchGeom g, float *__restrict__ out) {
    constexpr int kSets = SetsBf<DENSE>::N;
    __shared__ __attribute__((aligned(16))) float lds[kG3LdsWords];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave >> 2;          // 0: A, 1: B, 2: C
    const int wi = wave & 3;
    const int lo = lane & 15;
    const int q = lane >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    using mfcc_codelets::v2f;

    float *const T = lds + (gi == 2 ? kTile * kTFrame : 0);                // A and B share the first tile
    float *const grp = lds + 2 * kTile * kTFrame + gi * kGroupW;
    float *const V = grp, *const Q = grp + kTile * kVStride, *const S = grp + kTile * kVStride + kQGroup;
    float *const TwC = lds + 2 * kTile * kTFrame + 3 * kGroupW;            // [16 n2][kCRow]: 16 k1 x (cos, sin)
    int *const Cnt = reinterpret_cast<int *>(TwC + 16 * kCRow);            // [A, B]: pass-2 waves that have read their column

    // XCD-aware tile order (kernel_fused512_w12.hpp): consecutive tile triples on workgroups of the same XCD
    const unsigned nwg = gridDim.x;
    const unsigned bid = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    const int n_tiles = g.tiles_per_ch * g.n_ch;                          // < 2^30 (host check)
    const int gv = 3 * (int)gridDim.x;
    auto count = [&](unsigned v) { return (int)v < n_tiles ? (n_tiles - (int)v + gv - 1) / gv : 0; };
    const unsigned v0 = 3u * bid + (unsigned)gi;
    const int nA = count(3u * bid), nB = count(3u * bid + 1u);
    const int n_mine = count(v0);
    const int n_other = gi == 0 ? nB : nA;             // the partner on the shared tile (unused by C)
    const int total = 2 * (nA + 1) + 1;                // barriers after the prologue, the same for every wave

    for (int i = tid; i < 16 * 32; i += 64 * kG3Waves)
        TwC[(i >> 5) * kCRow + (i & 31)] = reinterpret_cast<const float *>(t.tw)[i];
    if (tid < 2) Cnt[tid] = 0;

    const int fr_id = wi + 8 * (q & 1) + 4 * (q >> 1);
    v2f wp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wp[i] = reinterpret_cast<const v2f *>(t.win)[lo * 16 + i];
    // The compiler's wait for these loads must not end up inside the tile loop: there a vmcnt wait also waits for the
    // window fetch this wave issued just before the barrier -- an HBM round trip per step (first run of this kernel:
    // 1.60 ms against the twelve-wave kernel's 1.00).  Reading the registers here puts the wait here.
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(wp[i]));
    const uint32_t *const abase = t.a_mel_bf4 + (size_t)wi * kSets * 512;  // uniform
    const float *const xbase = t.a_extra4 + (size_t)wi * 64 * kAextra;    // uniform; roles 0 (tail) and 1 (column 16)
    const int lane16 = lane * 16, lane64 = lane * (kAextra * 4);
    const bool more_cep = t.n_cep > 16;
    const int lane_off = lo * t.n_cep + 4 * q;
    // this lane's pieces of a window
    const int piece0 = wi < 2 ? wi * 64 + lane : 128 + (wi - 2) * 128 + lane;

    Cursor cur = cursor_of(s, g, v0);                  // the tile in the FFT
    Cursor pf = cur;                                   // the tile whose window is fetched next
    Cursor done = cur;                                 // the tile the tail finishes next (wave 0)
    Fetch2 fx;
    bool have_fx = false;
    int kf = 0;                                        // windows fetched so far
    auto fetch_next = [&]() {
        have_fx = false;
        if (kf < n_mine) {
            const Window w = window_of(pf, g);
            if (wi < 2) fetch_pieces<1>(s, w, piece0, fx);
            else fetch_pieces<2>(s, w, piece0, fx);
            advance(pf, g);
            ++kf;
            have_fx = true;
        }
    };
    auto park = [&]() {
        if (have_fx) {
            if (wi < 2) park_pieces<1>(S, piece0, fx);
            else park_pieces<2>(S, piece0, fx);
        }
    };
    fetch_next();                                      // prologue: S(0) directly, then the fetch of S(1)
    park();
    fetch_next();

    f32x4 lm0 = zero, lm1 = zero;
    lds_barrier();                                     // S(0) of every group, the twiddles and the counters are in LDS
    int bars = total;
    if (gi == 1) {                                     // step 0: group B idles
        lds_barrier();
        --bars;
    }
    for (int i = 0; i <= n_mine; ++i) {
        const bool fft = i < n_mine, fin = i >= 1;     // a tile to transform / a previous tile to finish (uniform)
        // ================================================================ pass-1 step
        if (wi == 0 && fin) {
            // the summed mel energies of the previous tile (the twelve-wave kernel's order), then log2
            const f32x4 *Q4 = reinterpret_cast<const f32x4 *>(Q) + lane;
            const f32x4 m0 = ((Q4[0 * 64] + Q4[2 * 64]) + (Q4[4 * 64] + Q4[6 * 64])) + Q4[8 * 64];
            const f32x4 m1 = ((Q4[1 * 64] + Q4[3 * 64]) + (Q4[5 * 64] + Q4[7 * 64])) + Q4[9 * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lm0[r] = __builtin_amdgcn_logf(m0[r]);
                lm1[r] = __builtin_amdgcn_logf(m1[r]);
            }
            if (t.n_mel <= 16) lm1 = zero;             // no filters 16..31 (uniform)
        }
        if (fft) {
            const int shift = window_of(cur, g).shift;
            v2f ep[16];
            {
                const float *sp = S + fr_id * kHop + lo + shift;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[16 * n1];
            }
            v2f tw[16];
            {
                const f32x4 *t4 = reinterpret_cast<const f32x4 *>(TwC + lo * kCRow);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 b = t4[j];
                    tw[2 * j] = (v2f){b[0], b[1]};
                    tw[2 * j + 1] = (v2f){b[2], b[3]};
                }
            }
            v2f ty[16];
            float y16;
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
            if (gi != 2) {
                // the shared tile still holds the partner group's columns until its four pass-2 waves have read them
                const int need = 4 * (gi ? i + 1 : (i < n_other ? i : n_other));
                while (__hip_atomic_load(Cnt + (gi ^ 1), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - need < 0)
                    __builtin_amdgcn_s_sleep(1);
            }
            v2f *tcol0 = reinterpret_cast<v2f *>(T + fr_id * kTFrame) + lo;            // a store's lanes: consecutive n2
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
            V[fr_id * kVStride + lo] = y16;
        }
        lds_barrier();
        // ================================================================ pass-2 step
        f32x4 ax4[4] = {zero, zero, zero, zero};
        f32x4 b0 = zero, b1 = zero, d0 = zero, d1 = zero;
        if (fft) {
            float pw[16];
            u32x4 ah[kSets], al[kSets];
            {
                v2f x[16], pp[8];
                const f32x4 *trow = reinterpret_cast<const f32x4 *>(T + lo * kTFrame + (4 * wi + q) * kTRow);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 a = trow[j];
                    x[2 * j] = (v2f){a[0], a[1]};
                    x[2 * j + 1] = (v2f){a[2], a[3]};
                }
                if (gi != 2 && lane == 0)
                    __hip_atomic_fetch_add(Cnt + gi, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                load_mel_burst<kSets>(ah, al, abase, lane16);
                if (wi < 2) load_role_burst(ax4, xbase, lane64, wi == 0 && more_cep);
                __builtin_amdgcn_sched_barrier(0);
                mfcc_codelets::cfft16_pow(x, pp);
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) pw[k2] = pp[k2].x, pw[k2 + 8] = pp[k2].y;
            }
            PowerBf pb;
            split_power(pw, pb);
            __builtin_amdgcn_sched_barrier(0);
            wait_bursts<kSets>(ah, al, ax4);
            f32x4 acc[kSets];
#pragma unroll
            for (int st = 0; st < kSets; ++st) acc[st] = zero;
            if (wi == 0 && fin) {
                // this tile's mel MFMAs with the previous tile's DCT MFMAs (coefficients 0..15, fp32) in between
                mel_bf_all<DENSE, 0>(ah, al, pb, acc, [&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    if constexpr (I < 8) {
                        constexpr int r = I >> 1;
                        if constexpr (I & 1) d1 = MFCC_MFMA(ax4[1][r], lm1[r], d1);
                        else d0 = MFCC_MFMA(ax4[0][r], lm0[r], d0);
                    }
                });
            } else {
                mel_bf_all<DENSE, 0>(ah, al, pb, acc, [](auto) {});
            }
            mel_bf_blocks<DENSE>(acc, b0, b1);
            if (wi == 1) {
                // column 16 -> bins 16 + 32 j (a 16 x 16 real DFT matrix on fp32 MFMAs), fed to both filter blocks
                const float *vp = V + lo * kVStride + q;
                const float v0 = vp[0], v1 = vp[4], v2 = vp[8], v3 = vp[12];
                f32x4 sp = MFCC_MFMA(ax4[0][0], v0, zero);
                f32x4 sp2 = MFCC_MFMA(ax4[0][1], v1, zero);
                sp = MFCC_MFMA(ax4[0][2], v2, sp);
                sp2 = MFCC_MFMA(ax4[0][3], v3, sp2);
                sp += sp2;
                const float s0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);          // bin 16 + 64 q
                const float s1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);          // bin 48 + 64 q
                const f32x4 x0 = MFCC_MFMA(ax4[1][0], s0, zero), y0 = MFCC_MFMA(ax4[1][1], s1, zero);
                const f32x4 x1 = MFCC_MFMA(ax4[1][2], s0, zero), y1 = MFCC_MFMA(ax4[1][3], s1, zero);
                *reinterpret_cast<f32x4 *>(Q + (2 * 4 + 0) * 256 + lane * 4) = x0 + y0;
                *reinterpret_cast<f32x4 *>(Q + (2 * 4 + 1) * 256 + lane * 4) = x1 + y1;
            }
        } else if (wi == 0 && fin) {
            // the group's last tile: its DCT alone
            load_role_burst(ax4, xbase, lane64, more_cep);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ax4[0]), "+v"(ax4[1]), "+v"(ax4[2]), "+v"(ax4[3]));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d0 = MFCC_MFMA(ax4[0][r], lm0[r], d0);
                d1 = MFCC_MFMA(ax4[1][r], lm1[r], d1);
            }
        }
        if (wi == 0 && fin) {
            float ax[kAextra];
#pragma unroll
            for (int k = 0; k < kAextra; ++k) ax[k] = ax4[k >> 2][k & 3];
            dct_store(s, t, lm0, lm1, d0, d1, ax, done, lo, q, lane_off, out);
            advance(done, g);
        }
        if (fft) {
            *reinterpret_cast<f32x4 *>(Q + (2 * wi + 0) * 256 + lane * 4) = b0;
            *reinterpret_cast<f32x4 *>(Q + (2 * wi + 1) * 256 + lane * 4) = b1;
            advance(cur, g);
            park();                                    // S(i + 1): the group reads it in its next pass 1
            fetch_next();                              // S(i + 2): two steps of lead
        }
        lds_barrier();
        bars -= 2;
    }
    for (; bars > 0; --bars) lds_barrier();
}

inline const char *kernel_name() { return "mfcc_fused512_g3_kernel"; }

// returns false when the problem does not fit or needs what only kernel_fused512_w12.hpp has (the integer DC chain)
inline bool launch(const mfcc_k::StreamDesc &s, const FusedTables &t, bool dense, float *out, int n_cu,
                   hipStream_t stream) {
    if (t.win_dc != nullptr) return false;
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    if (n_tiles >= (1ll << 30) || tiles_per_ch >= (1ll << 26) || n_ch >= (1ll << 30)) return false;
    long long wgs = (n_tiles + 2) / 3;
    if (wgs > n_cu) wgs = n_cu;
    if (wgs < 1) wgs = 1;
    const long long grid = 3 * wgs;                      // virtual workgroups: the cursor stride
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
    if (dense)
        hipLaunchKernelGGL((mfcc_fused512_g3_kernel<true>), dim3((unsigned)wgs), dim3(64 * kG3Waves), 0, stream, s, t, g, out);
    else
        hipLaunchKernelGGL((mfcc_fused512_g3_kernel<false>), dim3((unsigned)wgs), dim3(64 * kG3Waves), 0, stream, s, t, g, out);
    return true;
}

}  // namespace mfcc_fused_g3
