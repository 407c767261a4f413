// Fused 512/170/32 float kernel for gfx950 (MI355X): the whole chain of mfcc/core --
// pre-emphasis -> 512-sample frames (hop 170) -> Hamming -> FFT -> |.|^2 -> 32 mel -> log2 ->
// DCT-II -> first n_cep (<= 16) -- in one launch.
//
// Work unit: a workgroup of 4 waves owns a tile of 16 consecutive frames (the N dimension of
// v_mfma_f32_16x16x4_f32); wave w transforms frames 4w..4w+3, then the four waves split the
// mel contraction.  Per tile:
//
//  pass 1  lane = (q = lane>>4: frame 4w+q, n2 = lane&15).  The lane holds the 32 samples
//          x[16 n1 + n2] (n1 = 0..31) of its frame -- fetched one tile ahead as one 2-byte-aligned
//          dword per sample holding (x[i-1], x[i]), so pre-emphasis 32 x[i] - 31 x[i-1] is one
//          v_dot2c_i32_i16 -- and runs a register-resident REAL 32-point FFT over n1 with the
//          Hamming window folded into its first butterfly layer (codelets_gen.hpp): Y[k1, n2],
//          k1 = 0..16.  Columns 0..15 are multiplied by W512^(n2 k1) and written to the wave's LDS
//          transpose buffer T; column 16 (real) goes to the LDS tile V.
//  pass 2  lane = (q, k1 = lane&15) reads its column from T and runs a complex 16-point FFT over
//          n2: X[k1 + 32 k2], k2 = 0..15.  The input being real, each of these is a distinct needed
//          bin (k or 512-k): no real-FFT split step.  |X|^2 goes to the LDS power tile P[frame][bin].
//  ---- workgroup barrier ----
//  MFMA    mel energies = W (32 x 256, block-banded) . P as 68 + 4 MFMAs 16x16x4 (A = weights, kept
//          in registers for the whole kernel; B = P straight from the tile, one ds_read_b64 per two
//          MFMAs), split four ways by bin range.  Wave 0 also turns column 16 into bins 16+32j with a
//          16x16 real DFT matrix (4 MFMAs) and feeds them from registers.  Partial sums meet in LDS.
//  ---- workgroup barrier ----
//  tail    wave 0: log2, DCT-II as 8 MFMAs whose B operand IS the mel accumulator registers (the K
//          index is permuted so no lane movement is needed), store n_cep floats per frame.  Waves
//          1..3 already work on the next tile.
//
// HBM traffic per frame: 170 new int16 samples (the 3x overlap between frames is served by
// L1/L2) + 13 floats out = 392 B.  The kernel is fp32-VALU bound (about 9 k lane-ops per frame),
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
constexpr int kPStride = 260;             // words per frame in the power tile (== 4 mod 64)
constexpr int kTRow = 34;                 // words per n2 row of a transpose buffer
constexpr int kTQ = 16 * kTRow;           // 544 words per frame (== 32 mod 64)
constexpr int kTWave = 4 * kTQ;           // one wave's transpose buffer
constexpr int kVStride = 18;              // words per frame in the column-16 tile
constexpr int kAregs = 22;                // MFMA A operands resident per wave
constexpr int kLdsWords = kTile * kPStride + kWaves * kTWave + kTile * kVStride + 4 * 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int int_a2 __attribute__((aligned(2)));

struct FusedTables {
    const float *win;     // [16 n2][32 n1]   hamming[16 n1 + n2] / 32
    const float2 *tw;     // [16 n2][16 k1]   W512^(n2 k1)
    const float *a_all;   // [4 waves][22][64] MFMA A operands in the order each wave consumes them
    int n_cep;
};

// ---- the split of the mel contraction.  Chunk pair C covers bins 8C..8C+7 (two MFMAs: even bins,
// odd bins).  Filters 0..15 ("block 0") only touch bins < 64, filters 16..31 only bins >= 48
// (checked by build_tables), so block 0 needs C = 0..7 and block 1 needs C = 6..31.
//   wave 0: special DFT (4) + special-bin mel (4) + block 0 C 0..2 (6) + DCT (8)      = 22 A operands
//   wave 1: block 0 C 3..7 (10) + block 1 C 6..10 (10)                                 = 20
//   wave 2: block 1 C 11..20                                                           = 20
//   wave 3: block 1 C 21..31                                                           = 22
constexpr int kW0_B0_LO = 0, kW0_B0_HI = 3;
constexpr int kW1_B0_LO = 3, kW1_B0_HI = 8, kW1_B1_LO = 6, kW1_B1_HI = 11;
constexpr int kW2_B1_LO = 11, kW2_B1_HI = 21;
constexpr int kW3_B1_LO = 21, kW3_B1_HI = 32;

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
        for (int n1 = 0; n1 < 32; ++n1) win[n2 * 32 + n1] = float(w[16 * n1 + n2] / 32.0);
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
    for (int C = kW0_B0_LO; C < kW0_B0_HI; ++C) chunk(0, i0, C, 0);
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
    if (i0 != 22 || i1 != 20 || i2 != 20 || i3 != 22) return false;
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

__device__ __forceinline__ float preemph_x32(int packed) {
    // packed = (x[i-1], x[i]) as two int16: 32 x[i] - 31 x[i-1] (exact; 1/32 is in the window table)
    const s16x2 c = {(short)-31, (short)32};
    return (float)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, packed), c, 0, false);
}

__device__ __forceinline__ void wave_lds_fence() {
    // orders this wave's LDS writes before its later LDS reads (the data crosses lanes, not waves)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define MFCC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// chunk pairs [LO, HI) of one filter block: accumulate into (mx, my); A operands a[base ...]
template <int LO, int HI, int BASE>
__device__ __forceinline__ void mel_chunks(const float2 *pp, const float (&a)[kAregs], f32x4 &mx, f32x4 &my) {
#pragma unroll
    for (int C = LO; C < HI; ++C) {
        const float2 p = pp[4 * C];
        mx = MFCC_MFMA(a[BASE + 2 * (C - LO) + 0], p.x, mx);
        my = MFCC_MFMA(a[BASE + 2 * (C - LO) + 1], p.y, my);
    }
}

struct TileRef {
    long long ch, f0;
    bool inside;
};

__device__ __forceinline__ TileRef tile_ref(const mfcc_k::StreamDesc &s, long long tile, long long tiles_per_ch) {
    TileRef r;
    r.ch = tile / tiles_per_ch;
    r.f0 = (tile - r.ch * tiles_per_ch) * kTile;
    // Interior tiles (every sample index in [first - 1, last] exists) load one unaligned dword per
    // sample; edge tiles (stream start without history, zero-padded tail) are bounds-checked.
    r.inside = (r.f0 > 0 || s.halo) && (r.f0 + kTile - 1) * (long long)kHop + kNfft - 1 < s.n_samples;
    return r;
}

__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_fused512_kernel(mfcc_k::StreamDesc s, FusedTables t, long long tiles_per_ch, long long n_tiles,
                          float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsWords];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lo = lane & 15;          // n2 in pass 1, k1 in pass 2, frame column in the MFMA phase
    const int q = lane >> 4;           // frame 4*wave + q in the passes; K index g in the MFMA phase

    float *const Pt = lds;                                         // [16 frames][260]
    float *const Tt = lds + kTile * kPStride + wave * kTWave;      // this wave's [4 q][16 n2][34]
    float *const Vt = lds + kTile * kPStride + kWaves * kTWave;    // [16 frames][18]
    float *const Qt = Vt + kTile * kVStride;                       // 4 partial-sum blocks of 256 words

    // per-lane constants, resident for the whole kernel
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = t.win[lo * 32 + i];
    float2 tw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tw[i] = t.tw[lo * 16 + i];
    float a[kAregs];
#pragma unroll
    for (int i = 0; i < kAregs; ++i) a[i] = t.a_all[(wave * kAregs + i) * 64 + lane];

    // the slots of bins 16 (mod 32) are never written (those bins are fed from registers with the
    // chunk weights zeroed) -- make them finite once
    if (threadIdx.x < 128) Pt[(threadIdx.x >> 3) * kPStride + 16 + 32 * (threadIdx.x & 7)] = 0.0f;

    // prefetch the first tile's samples
    int raw[32];
    long long tile = blockIdx.x;
    TileRef cur = tile_ref(s, tile < n_tiles ? tile : 0, tiles_per_ch);
    if (tile < n_tiles && cur.inside) {
        const int16_t *p = s.pcm + cur.ch * s.ch_stride + (cur.f0 + 4 * wave + q) * (long long)kHop + lo - 1;
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) raw[n1] = *reinterpret_cast<const int_a2 *>(p + 16 * n1);
    }

    for (; tile < n_tiles; tile += gridDim.x) {
        // ---------------- pass 1: pre-emphasis + windowed real FFT-32 over n1
        float e[32];
        if (cur.inside) {
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) e[n1] = preemph_x32(raw[n1]);
        } else {
            const int16_t *base = s.pcm + cur.ch * s.ch_stride;
            const long long i0 = (cur.f0 + 4 * wave + q) * (long long)kHop + lo;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const long long i = i0 + 16 * n1;
                const int x0 = mfcc_k::sample_at_i(s, base, i);
                const int x1 = mfcc_k::sample_at_i(s, base, i - 1);
                e[n1] = (float)(32 * x0 - 31 * x1);
            }
        }
        // next tile's samples fly while this tile is processed
        const TileRef me = cur;
        const long long ntile = tile + gridDim.x;
        if (ntile < n_tiles) {
            cur = tile_ref(s, ntile, tiles_per_ch);
            if (cur.inside) {
                const int16_t *p = s.pcm + cur.ch * s.ch_stride + (cur.f0 + 4 * wave + q) * (long long)kHop + lo - 1;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) raw[n1] = *reinterpret_cast<const int_a2 *>(p + 16 * n1);
            }
        }

        float yr[17], yi[17];
        mfcc_codelets::rfft32_win(e, w, yr, yi);

        // twiddle W512^(n2 k1) and transpose through LDS: T[q][n2][k1]
        float2 *trow = reinterpret_cast<float2 *>(Tt + q * kTQ + lo * kTRow);
        trow[0] = make_float2(yr[0], 0.0f);
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
            const float re = fmaf(-yi[k1], tw[k1].y, yr[k1] * tw[k1].x);
            const float im = fmaf(yi[k1], tw[k1].x, yr[k1] * tw[k1].y);
            trow[k1] = make_float2(re, im);
        }
        Vt[(4 * wave + q) * kVStride + lo] = yr[16];
        wave_lds_fence();

        // ---------------- pass 2: complex FFT-16 over n2 for column k1 = lo
        {
            float xr[16], xi[16], zr[16], zi[16];
            const float2 *tcol = reinterpret_cast<const float2 *>(Tt + q * kTQ + 2 * lo);
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) {
                const float2 v = tcol[n2 * (kTRow / 2)];
                xr[n2] = v.x;
                xi[n2] = v.y;
            }
            mfcc_codelets::cfft16(xr, xi, zr, zi);
            float *prow_lo = Pt + (4 * wave + q) * kPStride + lo;            // bin k1 + 32 k2
            float *prow_hi = Pt + (4 * wave + q) * kPStride + 32 - lo;       // bin 32 - k1 + 32 (15 - k2)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) prow_lo[32 * k2] = fmaf(zr[k2], zr[k2], zi[k2] * zi[k2]);
#pragma unroll
            for (int k2 = 8; k2 < 16; ++k2) prow_hi[32 * (15 - k2)] = fmaf(zr[k2], zr[k2], zi[k2] * zi[k2]);
        }
        __syncthreads();                         // B1: P and V of all 16 frames are in LDS

        // ---------------- MFMA phase (frame column = lo, K index = q), split by wave
        const float2 *pp = reinterpret_cast<const float2 *>(Pt + lo * kPStride + 2 * q);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 m0 = zero, m1 = zero;              // wave 0 keeps its partial sums in registers
        if (wave == 0) {
            f32x4 sp = zero;
#pragma unroll
            for (int k = 0; k < 4; ++k) sp = MFCC_MFMA(a[k], Vt[lo * kVStride + 4 * k + q], sp);
            f32x4 m0y = zero, m1y = zero;
            mel_chunks<kW0_B0_LO, kW0_B0_HI, 8>(pp, a, m0, m0y);
            const float s0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);      // bin 16 + 64 q
            const float s1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);      // bin 48 + 64 q
            m0 = MFCC_MFMA(a[4], s0, m0);
            m0y = MFCC_MFMA(a[5], s1, m0y);
            m1 = MFCC_MFMA(a[6], s0, m1);
            m1y = MFCC_MFMA(a[7], s1, m1y);
            m0 += m0y;
            m1 += m1y;
        } else if (wave == 1) {
            f32x4 x0 = zero, y0 = zero, x1 = zero, y1 = zero;
            mel_chunks<kW1_B0_LO, kW1_B0_HI, 0>(pp, a, x0, y0);
            mel_chunks<kW1_B1_LO, kW1_B1_HI, 10>(pp, a, x1, y1);
            *reinterpret_cast<f32x4 *>(Qt + 0 * 256 + lane * 4) = x0 + y0;
            *reinterpret_cast<f32x4 *>(Qt + 1 * 256 + lane * 4) = x1 + y1;
        } else if (wave == 2) {
            f32x4 x1 = zero, y1 = zero;
            mel_chunks<kW2_B1_LO, kW2_B1_HI, 0>(pp, a, x1, y1);
            *reinterpret_cast<f32x4 *>(Qt + 2 * 256 + lane * 4) = x1 + y1;
        } else {
            f32x4 x1 = zero, y1 = zero;
            mel_chunks<kW3_B1_LO, kW3_B1_HI, 0>(pp, a, x1, y1);
            *reinterpret_cast<f32x4 *>(Qt + 3 * 256 + lane * 4) = x1 + y1;
        }
        __syncthreads();                         // B2: partial sums are in LDS, P/V may be overwritten

        // ---------------- tail (wave 0): log2 (MFCC.ipynb cell 36), DCT-II (cells 38-39), store.
        // Accumulator register r of block b is filter 16 b + 4 q + r of frame lo == B[k = q][j = lo]
        // of the DCT product, so the mel accumulators feed the DCT MFMAs without any lane movement.
        if (wave == 0) {
            m0 += *reinterpret_cast<const f32x4 *>(Qt + 0 * 256 + lane * 4);
            m1 += *reinterpret_cast<const f32x4 *>(Qt + 1 * 256 + lane * 4);
            m1 += *reinterpret_cast<const f32x4 *>(Qt + 2 * 256 + lane * 4);
            m1 += *reinterpret_cast<const f32x4 *>(Qt + 3 * 256 + lane * 4);
            f32x4 d0 = zero, d1 = zero;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d0 = MFCC_MFMA(a[14 + r], log2f(m0[r]), d0);
                d1 = MFCC_MFMA(a[18 + r], log2f(m1[r]), d1);
            }
            const long long fr = me.f0 + lo;
            if (fr < s.frames_per_ch) {
                float *o = out + (me.ch * s.frames_per_ch + fr) * t.n_cep;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * q + r < t.n_cep) o[4 * q + r] = d0[r] + d1[r];
            }
        }
    }
}

inline const char *kernel_name() { return "mfcc_fused512_kernel"; }

inline void launch(const mfcc_k::StreamDesc &s, const FusedTables &t, float *out, int n_cu,
                   hipStream_t stream) {
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    long long grid = n_tiles < (long long)n_cu * 2 ? n_tiles : (long long)n_cu * 2;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(mfcc_fused512_kernel, dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t,
                       tiles_per_ch, n_tiles, out);
}

}  // namespace mfcc_fused
