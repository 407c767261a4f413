// Fused 512/170/32 float kernel for gfx950 (MI355X): the whole chain of mfcc/core --
// pre-emphasis -> 512-sample frames (hop 170) -> Hamming -> FFT -> |.|^2 -> 32 mel -> log2 ->
// DCT-II -> first n_cep (<= 16) -- in one launch.
//
// Work unit: a workgroup of 4 waves owns a tile of 16 consecutive frames (the N dimension of
// v_mfma_f32_16x16x4_f32); wave w transforms frames 4w..4w+3, then the waves split the mel
// contraction.  Per tile:
//
//  input   the tile's contiguous sample span (16 frames = 3062 samples, ~6 KB) is fetched by the
//          whole workgroup with two 16-byte loads per thread -- each sample crosses HBM/L2 once per
//          tile -- one tile ahead, and parked in the LDS sample window S just before the second
//          barrier of the previous tile.
//  pass 1  lane = (q = lane>>4: frame 4w+q, n2 = lane&15) picks the 32 pairs (x[i-1], x[i]),
//          i = 16 n1 + n2, out of S; pre-emphasis 32 x[i] - 31 x[i-1] is one v_dot2c_i32_i16.  Then a
//          register-resident REAL 32-point FFT over n1 with the Hamming window folded into its first
//          butterfly layer (codelets_gen.hpp): Y[k1, n2], k1 = 0..16.  Columns 0..15 are multiplied
//          by W512^(n2 k1) and written to the wave's LDS transpose buffer T; column 16 (real) goes to
//          the LDS tile V.
//  pass 2  lane = (q, k1 = lane&15) reads its column from T and runs a complex 16-point FFT over
//          n2: X[k1 + 32 k2], k2 = 0..15.  The input being real, each of these is a distinct needed
//          bin (k or 512-k): no real-FFT split step.  |X|^2 goes to the LDS power tile P[frame][bin].
//  ---- workgroup barrier B1 ----
//  MFMA    mel energies = W (32 x 256, block-banded) . P as 68 MFMAs 16x16x4 (A = weights, kept in
//          registers for the whole kernel; B = P straight from the tile, one ds_read_b64 per two
//          MFMAs), split over waves 1..3 by bin range; partial sums go to LDS.  Wave 0 turns column 16
//          into bins 16+32j with a 16x16 real DFT matrix (4 MFMAs), feeds them from registers
//          (4 MFMAs), and -- in the same window -- finishes the PREVIOUS tile: log2, DCT-II as 8 MFMAs
//          whose B operand IS the mel accumulator registers (the K index is permuted so no lane
//          movement is needed), store n_cep floats per frame.  Nobody waits for that tail.
//  ---- workgroup barrier B2 ----
//
// HBM traffic per frame: 170 new int16 samples + 13 floats out = 392 B (plus the 342-sample overlap
// between consecutive tiles, 11 %).  The kernel is fp32-VALU bound (about 9 k lane-ops per frame),
// not HBM bound; DESIGN.md has the accounting.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "codelets_gen.hpp"
#include "kernels_generic.hpp"
#include "tables.hpp"

namespace mfcc_fused {

constexpr int kNfft = 512, kHop = 170, kMel = 32, kMaxCep = 16;
constexpr int kTile = 16;                 // frames per workgroup tile (MFMA N dimension)
constexpr int kWaves = 4;
constexpr int kTileHop = kTile * kHop;    // 2720 samples between consecutive tiles
constexpr int kPStride = 260;             // words per frame in the power tile (== 4 mod 64)
constexpr int kTRow = 34;                 // words per n2 row of a transpose buffer
constexpr int kTQ = 16 * kTRow;           // 544 words per frame (== 32 mod 64)
constexpr int kTWave = 4 * kTQ;           // one wave's transpose buffer
constexpr int kVStride = 18;              // words per frame in the column-16 tile
constexpr int kAregs = 24;                // MFMA A operands resident per wave
constexpr int kSHalf = 2048;              // slots of the sample window covered by every thread's first piece
constexpr int kSLead = 8;                 // the window starts 8 samples before the tile's first
constexpr int kSUsed = 3136;              // fp32 slots of the window (392 pieces of 8), see fetch_window
constexpr int kSSecond = kSUsed / 8 - 256;   // threads that fetch a second piece (136)
constexpr int kLdsWords = kTile * kPStride + kWaves * kTWave + kTile * kVStride + 2 * 4 * 256 + kSUsed;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct FusedTables {
    const float *win;     // [16 n2][32 n1]   hamming[16 n1 + n2] / 64 (pre-emphasis x32, real-FFT split x2)
    const float2 *tw;     // [16 n2][16 k1]   W512^(n2 k1)
    const float *a_all;   // [4 waves][24][64] MFMA A operands in the order each wave consumes them
    int n_cep;
};

// ---- the split of the mel contraction.  Chunk pair C covers bins 8C..8C+7 (two MFMAs: even bins,
// odd bins).  Filters 0..15 ("block 0") only touch bins < 64, filters 16..31 only bins >= 48
// (checked by build_tables), so block 0 needs C = 0..7 and block 1 needs C = 6..31.
//   wave 0: special DFT (4) + special-bin mel (4) + DCT (8)                            = 16 A operands
//   wave 1: block 0 C 0..7 (16) + block 1 C 6..8 (6)                                   = 22
//   wave 2: block 1 C 9..19                                                            = 22
//   wave 3: block 1 C 20..31                                                           = 24
constexpr int kW1_B0_LO = 0, kW1_B0_HI = 8, kW1_B1_LO = 6, kW1_B1_HI = 9;
constexpr int kW2_B1_LO = 9, kW2_B1_HI = 20;
constexpr int kW3_B1_LO = 20, kW3_B1_HI = 32;

inline bool supported(int nfft, int hop, int n_mel, int n_cep) {
    return nfft == kNfft && hop == kHop && n_mel == kMel && n_cep >= 1 && n_cep <= kMaxCep;
}

// ---- host: constant tables in the exact order the kernel consumes them
inline bool build_tables(int sample_rate, double power_scale, double lifter, int n_cep,
                         std::vector<char> &blob) {
    using namespace mfcc_tables;
    std::vector<float> win(16 * 32), tw(16 * 16 * 2), aall(size_t(kWaves) * kAregs * 64, 0.0f);
    std::vector<double> w = hamming_periodic(kNfft);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int n1 = 0; n1 < 32; ++n1) win[n2 * 32 + n1] = float(w[16 * n1 + n2] / 64.0);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int k1 = 0; k1 < 16; ++k1) {
            double a = -2.0 * kPi * double(n2 * k1) / 512.0;
            tw[(n2 * 16 + k1) * 2 + 0] = float(std::cos(a));
            tw[(n2 * 16 + k1) * 2 + 1] = float(std::sin(a));
        }
    std::vector<double> md = mel_dense(kNfft, kMel, double(sample_rate));     // [32][257]
    const double inv = 1.0 / (power_scale * power_scale);
    std::vector<char> covered(size_t(kMel) * 257, 0);
    auto A = [&](int wave, int idx, int lane) -> float & { return aall[(size_t(wave) * kAregs + idx) * 64 + lane]; };
    // chunk operands: lane l holds W[blk*16 + (l&15)][8C + 2(l>>4) + step]; bins == 16 (mod 32) are
    // fed separately from registers, so their weight is 0 here
    auto chunk = [&](int wave, int &idx, int C, int blk) {
        for (int step = 0; step < 2; ++step, ++idx)
            for (int l = 0; l < 64; ++l) {
                int filt = blk * 16 + (l & 15), bin = 8 * C + 2 * (l >> 4) + step;
                if ((bin & 31) == 16) continue;
                A(wave, idx, l) = float(md[size_t(filt) * 257 + bin] * inv);
                covered[size_t(filt) * 257 + bin] = 1;
            }
    };
    int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    // wave 0 -- column 16: X[16 + 32 k2] = sum_n2 v[n2] W512^(n2 (16 + 32 k2)); MFMA row i = 4g + r:
    // r=0: Re k2=2g, r=1: Im k2=2g, r=2: Re k2=2g+1, r=3: Im k2=2g+1
    for (int t = 0; t < 4; ++t, ++i0)
        for (int l = 0; l < 64; ++l) {
            int i = l & 15, n2 = 4 * t + (l >> 4);
            int g = i >> 2, r = i & 3, k2 = 2 * g + (r >> 1);
            double th = 2.0 * kPi * double(n2 * (16 + 32 * k2)) / 512.0;
            A(0, i0, l) = float((r & 1) ? -std::sin(th) : std::cos(th));
        }
    // wave 0 -- special bins as a K step: lane g supplies bin 16 + 64 g (step 0) / 48 + 64 g (step 1)
    for (int blk = 0; blk < 2; ++blk)
        for (int step = 0; step < 2; ++step, ++i0)
            for (int l = 0; l < 64; ++l) {
                int filt = blk * 16 + (l & 15), bin = 16 + 64 * (l >> 4) + 32 * step;
                A(0, i0, l) = float(md[size_t(filt) * 257 + bin] * inv);
                covered[size_t(filt) * 257 + bin] = 1;
            }
    std::vector<double> dd = dct_rows(n_cep, kMel, lifter);                    // [n_cep][32]
    for (int blk = 0; blk < 2; ++blk)
        for (int r = 0; r < 4; ++r, ++i0)
            for (int l = 0; l < 64; ++l) {
                int coeff = l & 15, filt = 16 * blk + 4 * (l >> 4) + r;
                A(0, i0, l) = coeff < n_cep ? float(dd[size_t(coeff) * kMel + filt]) : 0.0f;
            }
    for (int C = kW1_B0_LO; C < kW1_B0_HI; ++C) chunk(1, i1, C, 0);
    for (int C = kW1_B1_LO; C < kW1_B1_HI; ++C) chunk(1, i1, C, 1);
    for (int C = kW2_B1_LO; C < kW2_B1_HI; ++C) chunk(2, i2, C, 1);
    for (int C = kW3_B1_LO; C < kW3_B1_HI; ++C) chunk(3, i3, C, 1);
    if (i0 != 16 || i1 != 22 || i2 != 22 || i3 != 24) return false;
    for (int f = 0; f < kMel; ++f)
        for (int k = 0; k < 257; ++k)
            if (md[size_t(f) * 257 + k] != 0.0 && !covered[size_t(f) * 257 + k]) return false;
    auto put = [&](const std::vector<float> &v) {
        size_t off = blob.size();
        blob.resize(off + v.size() * 4);
        std::memcpy(blob.data() + off, v.data(), v.size() * 4);
    };
    blob.clear();
    put(win); put(tw); put(aall);
    return true;
}

inline void bind_tables(const char *b, int n_cep, FusedTables &t) {
    // device pointer arithmetic only; layout = build_tables' put() order
    t.n_cep = n_cep;
    const float *f = reinterpret_cast<const float *>(b);
    t.win = f;                  f += 16 * 32;
    t.tw = reinterpret_cast<const float2 *>(f); f += 16 * 16 * 2;
    t.a_all = f;
}

// ---- device

__device__ __forceinline__ void wave_lds_fence() {
    // orders this wave's LDS writes before its later LDS reads (the data crosses lanes, not waves)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define MFCC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Diagnostic build only (-DMFCC_FUSED_STAMPS): per-wave cycle sums of the phases of a tile, written
// to a buffer of their own that nothing else reads.  No stamp executes in the product build.
#ifdef MFCC_FUSED_STAMPS
__device__ unsigned long long g_stamps[8 * 8];     // [wave role 0..3 (+4: count)][phase]
#define MFCC_STAMP(i)                                                                         \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long now__;                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now__)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        st_sum[i] += now__ - st_prev;                                                         \
        st_prev = now__;                                                                      \
    } while (0)
#else
#define MFCC_STAMP(i) do {} while (0)
#endif

// chunk pairs [LO, HI) of one filter block: accumulate into (mx, my); A operands a[base ...]
template <int LO, int HI, int BASE>
__device__ __forceinline__ void mel_chunks(const mfcc_codelets::v2f *pp, const float (&a)[kAregs],
                                           f32x4 &mx, f32x4 &my) {
#pragma unroll
    for (int C = LO; C < HI; ++C) {
        const mfcc_codelets::v2f p = pp[4 * C];          // volatile: single ds_read_b64, see pass 2
        mx = MFCC_MFMA(a[BASE + 2 * (C - LO) + 0], p.x, mx);
        my = MFCC_MFMA(a[BASE + 2 * (C - LO) + 1], p.y, my);
    }
}

// Uniform (SGPR) cursor over the workgroup's tiles: tile = ch * tiles_per_ch + t_in.  Advancing by
// the grid size is an add with carry -- no division in the loop.
struct Cursor {
    int ch, t_in;
};

struct LaunchGeom {
    int tiles_per_ch, n_ch, grid_div, grid_mod;      // grid = grid_div * tiles_per_ch + grid_mod
};

__device__ __forceinline__ void advance(Cursor &c, const LaunchGeom &g) {
    c.t_in += g.grid_mod;
    c.ch += g.grid_div;
    if (c.t_in >= g.tiles_per_ch) {
        c.t_in -= g.tiles_per_ch;
        ++c.ch;
    }
}

// The tile's sample window: slot j stands for sample i = tile_first - kSLead - shift + j of the
// channel, j = 0..3135, where shift = 0..7 makes the first 16-byte global load aligned.  Thread tid
// fetches slots [8 tid, 8 tid + 8) and, for tid < 136, [2048 + 8 tid, ...), plus the dword holding the
// sample in front of each piece.  What is parked in LDS is the pre-emphasised sample
// e[i] = 32 x[i] - 31 x[i-1] as fp32 (exact: |e| < 2^21) -- computed once per sample here instead of
// once per (frame, sample) in pass 1, where three overlapping frames would each redo it.  Windows that
// stick out of the channel (stream start without history, zero-padded tail) are filled sample by
// sample with the stream's edge rules.
struct Fetch {
    i32x4 v0, v1;
    int p0, p1;          // dword in front of v0 / v1: its high half is the piece's predecessor sample
    int shift;
};

__device__ __forceinline__ void fetch_window(const mfcc_k::StreamDesc &s, const Cursor &c, int tid, Fetch &f) {
    const long long first = (long long)c.t_in * kTileHop - kSLead;                 // channel-relative
    const int16_t *base = s.pcm + (long long)c.ch * s.ch_stride;
    const int mis = (int)((reinterpret_cast<uintptr_t>(base + first) & 15) >> 1);  // samples past alignment
    // only slots 0 .. 3077 are ever read (kSLead + 7 + 15 * 170 + 511 + 1): 392 16-byte pieces, so the
    // second load is needed from threads 0..135 only -- 3136 samples fetched per 2720-sample tile
    // step instead of 4096
    const bool inside = first - mis >= -(long long)s.halo && first - mis + kSUsed <= s.n_samples;
    if (inside) {
        const i32x4 *g = reinterpret_cast<const i32x4 *>(base + first - mis);
        const int *g32 = reinterpret_cast<const int *>(g);
        f.shift = mis;
        f.v0 = g[tid];
        f.p0 = tid ? g32[4 * tid - 1] : 0;         // slot 0's e is never read (kSLead >= 1)
        f.v1 = (i32x4){0, 0, 0, 0};
        f.p1 = 0;
        if (tid < kSSecond) {
            f.v1 = g[256 + tid];
            f.p1 = g32[1024 + 4 * tid - 1];
        }
    } else {
        f.shift = 0;
        int h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long long i = first + (k < 8 ? 0 : kSHalf) + 8 * tid + (k & 7);
            h[k] = mfcc_k::sample_at_i(s, base, i) & 0xFFFF;
        }
        f.v0 = (i32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        f.v1 = (i32x4){h[8] | (h[9] << 16), h[10] | (h[11] << 16), h[12] | (h[13] << 16), h[14] | (h[15] << 16)};
        f.p0 = mfcc_k::sample_at_i(s, base, first + 8 * tid - 1) << 16;
        f.p1 = mfcc_k::sample_at_i(s, base, first + kSHalf + 8 * tid - 1) << 16;
    }
}

// e[k] = 32 x[k] - 31 x[k-1] for the 8 samples packed in v, x[-1] = high half of prev.  One
// v_dot2_i32_i16 per sample (the three-operand form: for the builtin hipcc picks v_dot2c, which costs
// an extra v_mov 0 per sample); the 1/32 is in the window table.
__device__ __forceinline__ void preemph8(int prev, const i32x4 &v, float *__restrict__ dst) {
    const int c3132 = 0x0020ffe1;                  // (int16 -31, int16 32)
    float e[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int before = m ? v[m - 1] : prev;
        const int pe = (int)__builtin_amdgcn_alignbit((unsigned)v[m], (unsigned)before, 16u);   // (x[2m-1], x[2m])
        int e0, e1;
        asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(e0) : "v"(pe), "s"(c3132));
        asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(e1) : "v"(v[m]), "s"(c3132));
        e[2 * m] = (float)e0;
        e[2 * m + 1] = (float)e1;
    }
    reinterpret_cast<f32x4 *>(dst)[0] = (f32x4){e[0], e[1], e[2], e[3]};
    reinterpret_cast<f32x4 *>(dst)[1] = (f32x4){e[4], e[5], e[6], e[7]};
}

__device__ __forceinline__ void park_window(float *Sf, int tid, const Fetch &f) {
    preemph8(f.p0, f.v0, Sf + 8 * tid);
    if (tid < kSSecond) preemph8(f.p1, f.v1, Sf + kSHalf + 8 * tid);
}

// log2 (MFCC.ipynb cell 36), DCT-II (cells 38-39) and store for one finished tile.  Accumulator
// register r of block b is filter 16 b + 4 q + r of frame lo == B[k = q][j = lo] of the DCT
// product, so the mel accumulators feed the DCT MFMAs without any lane movement.
__device__ __forceinline__ void finish_tile(const mfcc_k::StreamDesc &s, const FusedTables &t, const float *Qb,
                                            f32x4 m0, f32x4 m1, const float (&a)[kAregs], const Cursor &c,
                                            int lane, int lo, int q, float *__restrict__ out) {
    m0 += *reinterpret_cast<const f32x4 *>(Qb + 0 * 256 + lane * 4);
    m1 += *reinterpret_cast<const f32x4 *>(Qb + 1 * 256 + lane * 4);
    m1 += *reinterpret_cast<const f32x4 *>(Qb + 2 * 256 + lane * 4);
    m1 += *reinterpret_cast<const f32x4 *>(Qb + 3 * 256 + lane * 4);
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // v_log_f32 (1 ulp; a denormal mel energy -- far below anything int16 PCM produces --
        // counts as 0, like an exact zero: -inf)
        d0 = MFCC_MFMA(a[8 + r], __builtin_amdgcn_logf(m0[r]), d0);
        d1 = MFCC_MFMA(a[12 + r], __builtin_amdgcn_logf(m1[r]), d1);
    }
    const long long fr = (long long)c.t_in * kTile + lo;
    if (fr < s.frames_per_ch) {
        float *o = out + ((long long)c.ch * s.frames_per_ch + fr) * t.n_cep;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < t.n_cep) o[4 * q + r] = d0[r] + d1[r];
    }
}

__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_fused512_kernel(mfcc_k::StreamDesc s, FusedTables t, LaunchGeom g, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsWords];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15;          // n2 in pass 1, k1 in pass 2, frame column in the MFMA phase
    const int q = lane >> 4;           // quarter of the wave; K index g in the MFMA phase
    // frame of the tile this quarter transforms.  The two quarters of a 32-lane half are 8 frames =
    // 1360 samples = 16 (mod 32) LDS banks apart, so their ds_read_b32 of the window never collide.
    const int fr_id = wave + 8 * (q & 1) + 4 * (q >> 1);

    float *const Pt = lds;                                         // [16 frames][260]
    float *const Tt = lds + kTile * kPStride + wave * kTWave;      // this wave's [4 q][16 n2][34]
    float *const Vt = lds + kTile * kPStride + kWaves * kTWave;    // [16 frames][18]
    float *const Qt = Vt + kTile * kVStride;                       // [2][4 partial-sum blocks][256]
    float *const Sf = Qt + 2 * 4 * 256;                            // pre-emphasised sample window, fp32

    // per-lane constants, resident for the whole kernel
    using mfcc_codelets::v2f;
    v2f wp[16];                                    // window pairs (w[2m], w[2m+1]) of this lane's samples
#pragma unroll
    for (int i = 0; i < 16; ++i) wp[i] = reinterpret_cast<const v2f *>(t.win)[lo * 16 + i];
    v2f tw[16];                                    // W512^(n2 k1) as (cos, sin)
#pragma unroll
    for (int i = 0; i < 16; ++i) tw[i] = reinterpret_cast<const v2f *>(t.tw)[lo * 16 + i];
    float a[kAregs];
#pragma unroll
    for (int i = 0; i < kAregs; ++i) a[i] = t.a_all[(wave * kAregs + i) * 64 + lane];

    // the slots of bins 16 (mod 32) are never written (those bins are fed from registers with the
    // chunk weights zeroed) -- make them finite once
    if (tid < 128) Pt[(tid >> 3) * kPStride + 16 + 32 * (tid & 7)] = 0.0f;

    // slot of this lane's sample n1 = 0 in the window, before the per-tile alignment shift
    const int lane_slot = kSLead + fr_id * kHop + lo;

    Cursor cur;
    cur.ch = (int)(blockIdx.x / (unsigned)g.tiles_per_ch);
    cur.t_in = (int)(blockIdx.x - (unsigned)cur.ch * (unsigned)g.tiles_per_ch);

    // first tile: fetch and park the sample window
    Fetch fx;
    fx.shift = 0;
    if (cur.ch < g.n_ch) {
        fetch_window(s, cur, tid, fx);
        park_window(Sf, tid, fx);
    }
    int shift = fx.shift;
    __syncthreads();

    // wave 0 finishes tile t (log2, DCT, store) inside the MFMA phase of tile t + 1, so that the
    // other waves never wait for it: its partial sums and the tile's coordinates are carried here
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 keep0 = zero, keep1 = zero;
    Cursor prev = cur;
    bool have_prev = false;
    int par = 0;                                   // which half of Qt this tile's partial sums use

#ifdef MFCC_FUSED_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    while (cur.ch < g.n_ch) {
        // ---------------- pass 1: windowed real FFT-32 over n1 of the pre-emphasised samples
        mfcc_codelets::v2f ep[16];                 // (e[2m], e[2m+1]) of this lane's samples i = 16 n1 + n2
        {
            const float *sp = Sf + lane_slot + shift;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[16 * n1];
        }
        // next tile's samples fly while this tile is processed
        const Cursor me = cur;
        advance(cur, g);
        const bool more = cur.ch < g.n_ch;
        if (more) fetch_window(s, cur, tid, fx);
        MFCC_STAMP(6);

        // windowed real FFT-32 over n1, twiddled by W512^(n2 k1): columns 0..15 as (re, im) pairs, column 16
        mfcc_codelets::v2f ty[16];
        float y16;
        mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
        MFCC_STAMP(7);

        // transpose through LDS: T[q][n2][k1]
        mfcc_codelets::v2f *trow = reinterpret_cast<mfcc_codelets::v2f *>(Tt + q * kTQ + lo * kTRow);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) trow[k1] = ty[k1];
        Vt[fr_id * kVStride + lo] = y16;
        MFCC_STAMP(0);
        wave_lds_fence();

        // ---------------- pass 2: complex FFT-16 over n2 for column k1 = lo
        {
            mfcc_codelets::v2f x[16], z[16];
            // volatile: keeps these 16 ds_read_b64 from being paired into ds_read2_b64, which moves
            // half the bytes per LDS cycle (MI355X_MICROARCH.md, LDS table)
            const mfcc_codelets::v2f *tcol =
                reinterpret_cast<const mfcc_codelets::v2f *>(Tt + q * kTQ + 2 * lo);
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) x[n2] = tcol[n2 * (kTRow / 2)];
            MFCC_STAMP(1);
            mfcc_codelets::cfft16(x, z);
            float *prow_lo = Pt + fr_id * kPStride + lo;            // bin k1 + 32 k2
            float *prow_hi = Pt + fr_id * kPStride + 32 - lo;       // bin 32 - k1 + 32 (15 - k2)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) prow_lo[32 * k2] = fmaf(z[k2].x, z[k2].x, z[k2].y * z[k2].y);
#pragma unroll
            for (int k2 = 8; k2 < 16; ++k2) prow_hi[32 * (15 - k2)] = fmaf(z[k2].x, z[k2].x, z[k2].y * z[k2].y);
        }
        MFCC_STAMP(2);
        __syncthreads();                         // B1: P and V of all 16 frames are in LDS; S is consumed
        MFCC_STAMP(3);

        // ---------------- MFMA phase (frame column = lo, K index = q), split by wave
        const mfcc_codelets::v2f *pp =
            reinterpret_cast<const mfcc_codelets::v2f *>(Pt + lo * kPStride + 2 * q);
        float *const Qw = Qt + par * 1024;       // partial sums of this tile
        if (wave == 0) {
            // column 16 -> bins 16 + 32 j of this tile, fed to both filter blocks from registers
            f32x4 sp = zero;
#pragma unroll
            for (int k = 0; k < 4; ++k) sp = MFCC_MFMA(a[k], Vt[lo * kVStride + 4 * k + q], sp);
            // meanwhile: the previous tile's partial sums (other half of Qt) are complete since its B2
            if (have_prev) finish_tile(s, t, Qt + (par ^ 1) * 1024, keep0, keep1, a, prev, lane, lo, q, out);
            const float s0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);      // bin 16 + 64 q
            const float s1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);      // bin 48 + 64 q
            f32x4 m0 = MFCC_MFMA(a[4], s0, zero);
            f32x4 m1 = MFCC_MFMA(a[6], s0, zero);
            keep0 = MFCC_MFMA(a[5], s1, m0);
            keep1 = MFCC_MFMA(a[7], s1, m1);
        } else if (wave == 1) {
            f32x4 x0 = zero, y0 = zero, x1 = zero, y1 = zero;
            mel_chunks<kW1_B0_LO, kW1_B0_HI, 0>(pp, a, x0, y0);
            mel_chunks<kW1_B1_LO, kW1_B1_HI, 16>(pp, a, x1, y1);
            *reinterpret_cast<f32x4 *>(Qw + 0 * 256 + lane * 4) = x0 + y0;
            *reinterpret_cast<f32x4 *>(Qw + 1 * 256 + lane * 4) = x1 + y1;
        } else if (wave == 2) {
            f32x4 x1 = zero, y1 = zero;
            mel_chunks<kW2_B1_LO, kW2_B1_HI, 0>(pp, a, x1, y1);
            *reinterpret_cast<f32x4 *>(Qw + 2 * 256 + lane * 4) = x1 + y1;
        } else {
            f32x4 x1 = zero, y1 = zero;
            mel_chunks<kW3_B1_LO, kW3_B1_HI, 0>(pp, a, x1, y1);
            *reinterpret_cast<f32x4 *>(Qw + 3 * 256 + lane * 4) = x1 + y1;
        }
        prev = me;
        have_prev = true;
        par ^= 1;
        // park the next tile's sample window (every read of the current one happened before B1)
        if (more) {
            park_window(Sf, tid, fx);
            shift = fx.shift;
        }
        MFCC_STAMP(4);
        __syncthreads();                         // B2: partial sums and S are in LDS, P/V may be overwritten
        MFCC_STAMP(5);
    }
    // the last tile of this workgroup
    if (wave == 0 && have_prev) finish_tile(s, t, Qt + (par ^ 1) * 1024, keep0, keep1, a, prev, lane, lo, q, out);
#ifdef MFCC_FUSED_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[wave * 8 + i], st_sum[i]);
        atomicAdd(&g_stamps[32 + wave], 1ull);
    }
#endif
}

inline const char *kernel_name() { return "mfcc_fused512_kernel"; }

// returns false when the problem does not fit the kernel's 32-bit tile arithmetic
inline bool launch(const mfcc_k::StreamDesc &s, const FusedTables &t, float *out, int n_cu,
                   hipStream_t stream) {
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    if (n_tiles >= (1ll << 31) || tiles_per_ch >= (1ll << 26) || n_ch >= (1ll << 31)) return false;
    long long grid = n_tiles < (long long)n_cu * 2 ? n_tiles : (long long)n_cu * 2;
    if (grid < 1) grid = 1;
    LaunchGeom g;
    g.tiles_per_ch = (int)tiles_per_ch;
    g.n_ch = (int)n_ch;
    g.grid_div = (int)(grid / tiles_per_ch);
    g.grid_mod = (int)(grid % tiles_per_ch);
    hipLaunchKernelGGL(mfcc_fused512_kernel, dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t, g, out);
    return true;
}

}  // namespace mfcc_fused
