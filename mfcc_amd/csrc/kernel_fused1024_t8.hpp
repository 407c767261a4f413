// Fused 1024/341/40 float kernel for gfx950 (MI355X), EIGHT-frame tiles -- BASELINE.json configs[3]: nfft 1024, hop
// 1024 // 3 = 341 (mfcc/core/mfcc.py:43), 40 mel bands, n_cep <= 40, the mel contraction on the matrix cores.
//
// Why eight frames.  Round 2's kernel (kernel_fused1024.hpp) gave a 16-frame tile to ONE workgroup of eight waves per CU
// (a 16-frame tile of 1024-point frames fills the LDS once).  All eight waves then walk through the same phases in
// lockstep -- window reads, FFT, T writes | T reads, FFT, MFMAs -- so the LDS bursts (2 300 of the tile's 6 700 clocks),
// the fp32 MFMAs (1 150, during which the SIMD issues no vector instruction) and the FFT arithmetic never overlap:
// 52 % of the vector pipe.  An 8-frame tile of 1024-point frames is the data volume of the 512 kernel's 16-frame tile,
// so this kernel is that kernel's four-wave form (kernel_fused512.hpp): FOUR waves per workgroup, 67 KB of LDS, TWO
// workgroups per CU that drift apart and fill each other's LDS and barrier waits.
//
//  pass 1   n = 32 n1 + n2.  Wave w, lane (f = lane >> 5, n2 = lane & 31) owns frame 2 w + f: the register-resident REAL
//           32-point FFT over n1 (codelet rfft32_tw, Hamming folded in), columns 0..15 twiddled by W1024^(n2 k1) -- the
//           twiddles are read from LDS right before the codelet, they are dead weight in every other phase -- written to
//           T[frame][k1][n2] (lanes of a store are consecutive n2: conflict free); column 16 (real) goes to V.
//  pass 2   wave w, lane (j = lane & 7, sl = lane >> 3 & 1, q = lane >> 4) takes frame j, column k1 = 8 (w >> 1) + sl + 2 q and
//           the outputs k2 = 2 m + h, h = w & 1, of the complex 32-point FFT over n2 (codelets cfft32_h0 / _h1: one
//           decimation-in-frequency step, then a 16-point FFT).  Its 32 inputs are CONTIGUOUS in T: 16 ds_read_b128 at
//           256 B/clk instead of 16 ds_read2_b64 at 128 (row stride 68 words, frame stride 1104: conflict free).
//  mel      |X|^2 is in registers in the B-operand layout of v_mfma_f32_16x16x32_bf16 -- but the MFMA has 16 columns and the
//           tile 8 frames: column n = (j, sl), and the two halves of the columns hold DIFFERENT bins (k1 even / odd).  So
//           every wave runs the contraction twice, once with the weights of its even columns k1 and once with those of the
//           odd ones, into separate accumulators; half of each result's columns are the other half's bins against the wrong
//           weights and are simply never read (12 v_cndmask pick the right half at the end).  Both operands split in two
//           bf16 terms (W = Wh + Wl, P = Ph + Pl; Wh Ph + Wh Pl + Wl Ph, fp32 accumulation, 2^-17 relative): 24 MFMAs of
//           16 clocks per wave and tile that run beside vector work, where round 2's kernel spent 17-18 fp32 MFMAs of 32
//           clocks per wave that block the SIMD.  The 16 outputs of a lane are two K groups of eight: X = the eight lowest
//           in frequency (m = 0..3, 12..15), Y = the rest.  Sets (K group, filter block): (X, 0), (X, 1), (Y, 2) and, by
//           sample rate, (Y, 1) [<= 22.05 kHz], (X, 2) [44.1 / 48 kHz] or both [32 kHz]: three instantiations, every
//           common rate fused (round 2: 44.1 / 48 kHz ran on the generic kernel, six times slower).
//  column 16  -> bins 16 + 32 j', j' = 0..15, by a 32-point DFT matrix on the fp32 matrix cores, j' < 8 on wave 2, the rest
//           on wave 1.  The K dimension (n2) is split over the two column halves -- sl = 1 takes n2 = 16..31, whose matrix
//           is the first half's times (-i)(-1)^j' per row -- so 4 MFMAs instead of 8, one DPP add per register.
//  tail     wave 3 finishes the PREVIOUS tile: sums of the partial mel energies (both column halves), log2, DCT-II as 12
//           fp32 MFMAs per 16 coefficients, store.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "codelets_gen.hpp"
#include "fused_common.hpp"
#include "kernels_generic.hpp"
#include "tables.hpp"

// experiment switches (A/B builds, tools/ab1k.sh); the defaults are what ships
#ifndef F1K_PRESUM
#define F1K_PRESUM 1        // each wave adds its two column halves (DPP) before the Q write
#endif
#ifndef F1K_SPLIT_TAIL
#define F1K_SPLIT_TAIL 1    // wave 0: sums + log2 before B1, wave 3: DCT behind B1 (0: all of it on wave 3)
#endif
#ifndef F1K_TAIL_PRIO
#define F1K_TAIL_PRIO 0     // s_setprio of the waves with a tail / column-16 job while they do it
#endif

namespace mfcc_f1k {

constexpr int kNfft = 1024, kHop = 341, kMel = 40, kMaxCep = 40;
constexpr int kTile = 8, kWaves = 4;
constexpr int kTileHop = kTile * kHop;            // 2728 samples between consecutive tiles
constexpr int kTRow = 68;                         // words per k1 row of the transpose tile: 32 complex + 4
constexpr int kTFrame = 16 * kTRow + 16;          // 1104 words per frame: (row, frame) strides of (17, 276) 16-byte units
                                                  // make every ds_read_b128 of pass 2 conflict free (brute-forced)
constexpr int kVStride = 34;                      // words per frame in the column-16 tile
constexpr int kBlocks = 3;                        // 16-filter blocks of the 40 filters
constexpr int kQWords = kWaves * kBlocks * 256;   // partial mel sums: [wave][block][lane * 4]
constexpr int kFetchers = 64 * kWaves;            // every lane fetches and parks a piece of the sample window
constexpr int kPieces = (7 + (kTile - 1) * kHop + kNfft + 7) / 8;       // 428 pieces of 8 samples
constexpr int kSecond = kPieces - kFetchers;      // lanes that take a second piece (172)
constexpr int kSUsed = 8 * kPieces;               // 3424 fp32 slots
constexpr int kTwRow = 36;                        // words per n2 row of the twiddle table in LDS (9 16-byte units: the
                                                  // 16 lanes of a ds_read_b128 group hit 16 different units mod 16)
constexpr int kLWords = kBlocks * 256;             // log-mel values of the finished tile, wave 0 -> wave 3: [block][lane * 4]
constexpr int kLdsWords = kTile * kTFrame + kTile * kVStride + kQWords + kSUsed + 2 * 32 * kTwRow + kLWords;   // + window rows, same shape
constexpr int kArole = 12;                        // per-wave role operands (fp32 MFMA A operands)

// K slots of the two bf16 MFMAs of a lane: K index 8 q + i  <->  output m = kGrpM[grp][i] of lane group q
constexpr int kGrpM[2][8] = {{0, 1, 2, 3, 12, 13, 14, 15}, {4, 5, 6, 7, 8, 9, 10, 11}};

// (K group, filter block) sets of a variant.  Sets 0..2 are the same everywhere; what a sample rate adds is (Y, 1) (the
// second block reaches past bin 256), (X, 2) (the third block starts below it) or both.
template <int VAR> struct Sets;
// c16[mb][blk]: the filter blocks that column 16's bins 16 + 32 j' reach, j' < 8 (mb = 0, bins 16..240) / j' >= 8 (mb = 1)
template <> struct Sets<0> {                      // <= 22.05 kHz
    static constexpr int N = 4;
    static constexpr int grp[N] = {0, 0, 1, 1}, blk[N] = {0, 1, 2, 1};
    static constexpr int c16[2][3] = {{1, 1, 0}, {0, 1, 1}};
};
template <> struct Sets<1> {                      // 44.1, 48 kHz
    static constexpr int N = 4;
    static constexpr int grp[N] = {0, 0, 1, 0}, blk[N] = {0, 1, 2, 2};
    static constexpr int c16[2][3] = {{1, 1, 1}, {0, 0, 1}};
};
template <> struct Sets<2> {                      // 32 kHz
    static constexpr int N = 5;
    static constexpr int grp[N] = {0, 0, 1, 1, 0}, blk[N] = {0, 1, 2, 1, 2};
    static constexpr int c16[2][3] = {{1, 1, 0}, {0, 0, 1}};
};
constexpr int kVariants = 3, kMaxSets = 5;
struct SetsView {
    int n;
    const int *grp, *blk;
    const int (*c16)[3];
};
inline SetsView sets_view(int v) {
    if (v == 1) return {Sets<1>::N, Sets<1>::grp, Sets<1>::blk, Sets<1>::c16};
    if (v == 2) return {Sets<2>::N, Sets<2>::grp, Sets<2>::blk, Sets<2>::c16};
    return {Sets<0>::N, Sets<0>::grp, Sets<0>::blk, Sets<0>::c16};
}

using mfcc_fc::f32x4;
using mfcc_fc::i32x4;
using mfcc_fc::Cursor;
using mfcc_fc::LaunchGeom;
using mfcc_fc::Window;
using mfcc_fc::advance;
using mfcc_fc::window_of;
using mfcc_fc::preemph8;
using mfcc_fc::lds_barrier;
using mfcc_codelets::v2f;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Tables {
    int variant;           // which Sets<VAR> the operand tables were laid out for
    const float *win;      // [32 n2][32 n1]  hamming[32 n1 + n2] / 64
    const float *tw;       // [32 n2][16 k1][2] W1024^(n2 k1)
    const uint32_t *a_bf;  // [4 waves][2 column halves][sets][hi, lo][4 dwords][64 lanes] mel weights as bf16 pairs
    const float *a_role;   // [4 waves][kArole][64]  wave 3: DCT rows 0..15; waves 2 / 1: column-16 DFT (4) + its mel weights (6)
    const float *a_dct_hi; // [2][kArole][64]  DCT rows of coefficients 16..31 and 32..47 (fetched per tile, only when n_cep > 16)
    int n_cep;
};

inline bool supported(int nfft, int hop, int n_mel, int n_cep) {
    return nfft == kNfft && hop == kHop && n_mel == kMel && n_cep >= 1 && n_cep <= kMaxCep;
}

// bin of output m of the (k1, h) lane; -1: a duplicate that another lane supplies
inline int bin_of(int k1, int h, int m) {
    const int k2 = 2 * m + h;
    if (k2 < 16) return k1 + 32 * k2;
    if (k1 == 0 && k2 > 16) return -1;
    return 32 * (32 - k2) - k1;
}

inline bool build_tables_for(int variant, int sample_rate, double power_scale, double lifter, int n_cep,
                             std::vector<char> &blob) {
    using namespace mfcc_tables;
    const SetsView sv = sets_view(variant);
    std::vector<float> win(32 * 32), tw(32 * 16 * 2), arole(size_t(kWaves) * kArole * 64, 0.0f),
        adct(size_t(2) * kArole * 64, 0.0f);
    std::vector<double> w = hamming_periodic(kNfft);
    for (int n2 = 0; n2 < 32; ++n2)
        for (int n1 = 0; n1 < 32; ++n1) win[n2 * 32 + n1] = float(w[32 * n1 + n2] / 64.0);
    for (int n2 = 0; n2 < 32; ++n2)
        for (int k1 = 0; k1 < 16; ++k1) {
            double a = -2.0 * kPi * double(n2 * k1) / 1024.0;
            tw[(n2 * 16 + k1) * 2 + 0] = float(std::cos(a));
            tw[(n2 * 16 + k1) * 2 + 1] = float(std::sin(a));
        }
    const int nb = kNfft / 2 + 1;                                             // 513
    std::vector<double> md = mel_dense(kNfft, kMel, double(sample_rate));     // [40][513]
    for (int f = 0; f < kMel; ++f)
        if (md[size_t(f) * nb] != 0.0) return false;      // weight on the real-valued DC bin: not summed in fp32
    const double inv = 1.0 / (power_scale * power_scale);
    std::vector<char> covered(size_t(kMel) * nb, 0);
    auto Wt = [&](int filt, int bin) -> double { return filt < kMel ? md[size_t(filt) * nb + bin] * inv : 0.0; };
    auto bf16_round = [](float v) -> uint32_t {                              // round to nearest even, like v_cvt_pk_bf16_f32
        uint32_t u;
        std::memcpy(&u, &v, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    };
    auto bf16_val = [](uint32_t h) -> float {
        uint32_t u = h << 16;
        float v;
        std::memcpy(&v, &u, 4);
        return v;
    };
    // mel operands: lane l of wave wv, column half s, set st holds rows l & 15 of filter block blk[st] at K slots i = 0..7
    // <-> bin(k1 = 8 (wv >> 1) + s + 2 (l >> 4), h = wv & 1, m = kGrpM[grp[st]][i]); dword d = slots (2 d, 2 d + 1)
    std::vector<uint32_t> abf(size_t(kWaves) * 2 * sv.n * 2 * 4 * 64, 0u);
    for (int wv = 0; wv < kWaves; ++wv)
        for (int s = 0; s < 2; ++s)
            for (int st = 0; st < sv.n; ++st)
                for (int l = 0; l < 64; ++l) {
                    uint32_t hi[8], lo[8];
                    for (int i = 0; i < 8; ++i) {
                        const int filt = sv.blk[st] * 16 + (l & 15), k1 = 8 * (wv >> 1) + s + 2 * (l >> 4);
                        const int bin = bin_of(k1, wv & 1, kGrpM[sv.grp[st]][i]);
                        float wgt = 0.0f;
                        if (bin >= 0 && filt < kMel) {
                            wgt = float(Wt(filt, bin));
                            covered[size_t(filt) * nb + bin] = 1;
                        }
                        hi[i] = bf16_round(wgt);
                        lo[i] = bf16_round(wgt - bf16_val(hi[i]));
                    }
                    const size_t base = ((size_t(wv) * 2 + s) * sv.n + st) * 2 * 256;
                    for (int d = 0; d < 4; ++d) {
                        abf[base + 0 * 256 + d * 64 + l] = hi[2 * d] | (hi[2 * d + 1] << 16);
                        abf[base + 1 * 256 + d * 64 + l] = lo[2 * d] | (lo[2 * d + 1] << 16);
                    }
                }
    auto R = [&](int wave, int idx, int lane) -> float & { return arole[(size_t(wave) * kArole + idx) * 64 + lane]; };
    // wave 3 -- DCT rows: lane (coeff = l & 15, g = l >> 4) holds D[16 tile + coeff][16 blk + 4 g + r]
    std::vector<double> dd = dct_rows(n_cep, kMel, lifter);                   // [n_cep][40]
    for (int tile = 0; tile < 3; ++tile)
        for (int blk = 0; blk < kBlocks; ++blk)
            for (int r = 0; r < 4; ++r)
                for (int l = 0; l < 64; ++l) {
                    const int coeff = 16 * tile + (l & 15), filt = 16 * blk + 4 * (l >> 4) + r;
                    const float v = (coeff < n_cep && filt < kMel) ? float(dd[size_t(coeff) * kMel + filt]) : 0.0f;
                    if (tile == 0) R(3, 4 * blk + r, l) = v;
                    else adct[(size_t(tile - 1) * kArole + 4 * blk + r) * 64 + l] = v;
                }
    // waves 2 (mb = 0) and 1 (mb = 1) -- column 16: X[16 + 32 j'] = sum_n2 v[n2] W1024^(n2 (16 + 32 j')), j' = 8 mb + 0..7.
    // MFMA row i = 4 g + r holds r = 0: Re j' = 8 mb + 2 g, r = 1: Im, r = 2: Re j' + 1, r = 3: Im; K step t covers
    // n2 = 4 t + (l >> 4) for the columns sl = 0 and n2 + 16 for sl = 1 (the kernel rotates those by (-i)(-1)^j')
    for (int mb = 0; mb < 2; ++mb) {
        const int wave = mb ? 1 : 2;
        for (int t = 0; t < 4; ++t)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, n2 = 4 * t + (l >> 4);
                const int g = i >> 2, r = i & 3, jp = 8 * mb + 2 * g + (r >> 1);
                const double th = 2.0 * kPi * double(n2 * (16 + 32 * jp)) / 1024.0;
                R(wave, t, l) = float((r & 1) ? -std::sin(th) : std::cos(th));
            }
        // its bins as K steps of fp32 MFMAs: lane group q supplies bin 16 + 32 (8 mb + 2 q + step) to block blk
        for (int blk = 0; blk < kBlocks; ++blk)
            for (int step = 0; step < 2; ++step)
                for (int l = 0; l < 64; ++l) {
                    const int filt = blk * 16 + (l & 15), bin = 16 + 32 * (8 * mb + 2 * (l >> 4) + step);
                    if (filt >= kMel || !sv.c16[mb][blk]) continue;
                    R(wave, 4 + 2 * blk + step, l) = float(Wt(filt, bin));
                    covered[size_t(filt) * nb + bin] = 1;
                }
    }
    for (int f = 0; f < kMel; ++f)
        for (int k = 0; k < nb; ++k)
            if (md[size_t(f) * nb + k] != 0.0 && !covered[size_t(f) * nb + k]) return false;
    auto put = [&](const void *p, size_t bytes) {
        size_t off = blob.size();
        blob.resize(off + bytes);
        std::memcpy(blob.data() + off, p, bytes);
    };
    blob.clear();
    put(win.data(), win.size() * 4);
    put(tw.data(), tw.size() * 4);
    put(arole.data(), arole.size() * 4);
    put(adct.data(), adct.size() * 4);
    put(abf.data(), abf.size() * 4);
    return true;
}

// the first variant whose sets cover the rate's filterbank (build_tables_for checks every non-zero weight)
inline bool build_tables(int sample_rate, double power_scale, double lifter, int n_cep, std::vector<char> &blob, int &variant) {
    for (variant = 0; variant < kVariants; ++variant)
        if (build_tables_for(variant, sample_rate, power_scale, lifter, n_cep, blob)) return true;
    return false;
}

inline void bind_tables(const char *b, int n_cep, int variant, Tables &t) {
    t.variant = variant;
    t.n_cep = n_cep;
    const float *f = reinterpret_cast<const float *>(b);
    t.win = f;        f += 32 * 32;
    t.tw = f;         f += 32 * 16 * 2;
    t.a_role = f;     f += kWaves * kArole * 64;
    t.a_dct_hi = f;   f += 2 * kArole * 64;
    t.a_bf = reinterpret_cast<const uint32_t *>(f);
}

// ---- device

#define MFCC1K8_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFCC1K8_MFMA_BF(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

struct Fetch {
    i32x4 v0, v1;
    int p0, p1;
};

// lane u (0..255) takes piece u and, u < 172, piece 256 + u of the window, plus the dword in front of each
__device__ __forceinline__ void fetch_window(const mfcc_k::StreamDesc &s, const Window &w, int u, Fetch &f) {
    if (w.inside) {
        const i32x4 *g = reinterpret_cast<const i32x4 *>(w.ptr - w.shift);
        const int *g32 = reinterpret_cast<const int *>(g);
        f.v0 = g[u];
        f.p0 = g32[4 * u - 1];
        f.v1 = (i32x4){0, 0, 0, 0};
        f.p1 = 0;
        if (u < kSecond) {
            f.v1 = g[kFetchers + u];
            f.p1 = g32[4 * (kFetchers + u) - 1];
        }
    } else {
        const long long first = (long long)w.t_in * kTileHop;
        const int16_t *base = w.ptr - first;
        int h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long long i = first + (k < 8 ? 0 : 8 * kFetchers) + 8 * u + (k & 7);
            h[k] = mfcc_k::sample_at_i(s, base, i) & 0xFFFF;
        }
        f.v0 = (i32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        f.v1 = (i32x4){h[8] | (h[9] << 16), h[10] | (h[11] << 16), h[12] | (h[13] << 16), h[14] | (h[15] << 16)};
        f.p0 = mfcc_k::sample_at_i(s, base, first + 8 * u - 1) << 16;
        f.p1 = mfcc_k::sample_at_i(s, base, first + 8 * (kFetchers + u) - 1) << 16;
    }
}

__device__ __forceinline__ void park_window(float *Sf, int u, const Fetch &f) {
    preemph8(f.p0, f.v0, Sf + 8 * u);
    if (u < kSecond) preemph8(f.p1, f.v1, Sf + 8 * (kFetchers + u));
}

// the value of lane ^ 8 (the other column half of the same frame): a rotation by 8 inside each row of 16 lanes
__device__ __forceinline__ float other_half(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}

// (a, b) -> their bf16 roundings packed in one dword (a low) and the bf16 roundings of what the first rounding lost
__device__ __forceinline__ void split_bf16_pair(float a, float b, uint32_t &hi, uint32_t &lo) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
}

// summed mel energies of a finished tile and their log2: register r of block b is filter 16 b + 4 q + r of frame j (the
// four waves' partial sums; each wave has already added its two column halves).  Filters 40..47 do not exist: their
// (zero) sums must not reach the DCT as -inf * 0
__device__ __forceinline__ void mel_log2(const float *Qt, int lane, int q, f32x4 (&lm)[kBlocks]) {
    const f32x4 *Qa = reinterpret_cast<const f32x4 *>(Qt) + lane;
#if !F1K_PRESUM
    const f32x4 *Qb = reinterpret_cast<const f32x4 *>(Qt) + (lane ^ 8);
#endif
#pragma unroll
    for (int b = 0; b < kBlocks; ++b) {
        f32x4 m = (Qa[(0 * kBlocks + b) * 64] + Qa[(1 * kBlocks + b) * 64]) +
                  (Qa[(2 * kBlocks + b) * 64] + Qa[(3 * kBlocks + b) * 64]);
#if !F1K_PRESUM
        m += (Qb[(0 * kBlocks + b) * 64] + Qb[(1 * kBlocks + b) * 64]) + (Qb[(2 * kBlocks + b) * 64] + Qb[(3 * kBlocks + b) * 64]);
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) lm[b][r] = __builtin_amdgcn_logf(m[r]);
    }
    if (q >= 2) lm[2] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// d[0] + d[1] + d[2] = coefficients 0..15 of the previous tile (their 12 MFMAs are issued by the caller); 16..31 and
// 32..39 are further M tiles whose A operands are fetched here (uniform branches; 12 coalesced dwords per lane out of L2)
__device__ __forceinline__ void dct_store(const mfcc_k::StreamDesc &s, const Tables &t, const f32x4 (&d)[kBlocks],
                                          const f32x4 (&lm)[kBlocks], const Cursor &c, int j, int sl, int q, int lane,
                                          float *__restrict__ out) {
    const long long fr0 = (long long)c.t_in * kTile;
    const long long rows_left = s.frames_per_ch - fr0;
    float *o = out + ((long long)c.ch * s.frames_per_ch + fr0) * t.n_cep + j * t.n_cep + 4 * q;
    const bool mine = sl == 0 && j < rows_left;
    if (mine) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < t.n_cep) o[r] = (d[0][r] + d[1][r]) + d[2][r];
    }
    for (int tile = 1; 16 * tile < t.n_cep; ++tile) {
        const float *hi = t.a_dct_hi + (size_t)(tile - 1) * kArole * 64 + lane;
        asm volatile("" : "+v"(hi));                   // not hoisted out of the tile loop: its 12 values are not worth 12 registers
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 e[kBlocks] = {zero, zero, zero};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) e[b] = MFCC1K8_MFMA(hi[(4 * b + r) * 64], lm[b][r], e[b]);
        if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * tile + 4 * q + r < t.n_cep) o[16 * tile + r] = (e[0][r] + e[1][r]) + e[2][r];
        }
    }
}

// Diagnostic build only (-DMFCC_F1K_STAMPS): per-wave cycle sums of the phases of a tile, written to a buffer of their
// own that nothing else reads.  No stamp executes in the product build.
#ifdef MFCC_F1K_STAMPS
__device__ unsigned long long g_stamps1k[4 * 12 + 4];   // [wave][12 phases], then [48 + wave]: tiles
#define F1K_STAMP(i)                                                                          \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long now__;                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now__)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        st_sum[i] += now__ - st_prev;                                                         \
        st_prev = now__;                                                                      \
    } while (0)
#else
#define F1K_STAMP(i) do {} while (0)
#endif

template <int VAR>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_fused1024_kernel(mfcc_k::StreamDesc s, Tables t, LaunchGeom g, float *__restrict__ out) {
    using S = Sets<VAR>;
    constexpr int NS = S::N;
    __shared__ __attribute__((aligned(16))) float lds[kLdsWords];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave & 1;
    const int j = lane & 7;            // frame in pass 2 and the MFMA window
    const int sl = (lane >> 3) & 1;    // column half: k1 even / odd
    const int q = lane >> 4;           // K index in the MFMA window
    const int n2 = lane & 31;          // pass 1
    const int fr_id = 2 * wave + (lane >> 5);

    float *const Tt = lds;                                         // [8 frames][1104]: [16 k1][68] each
    float *const Vt = Tt + kTile * kTFrame;                        // [8 frames][34]
    float *const Qt = Vt + kTile * kVStride;                       // [4 waves][3 blocks][256]
    float *const Sf = Qt + kQWords;                                // pre-emphasised sample window, fp32
    float *const Tw = Sf + kSUsed;                                 // [32 n2][36]: W1024^(n2 k1), k1 = 0..15

    float *const Wn = Tw + 32 * kTwRow;                            // [32 n2][36]: hamming[32 n1 + n2] / 64, n1 = 0..31
    float *const Lt = Wn + 32 * kTwRow;                            // [3 blocks][256]: log-mel of the finished tile
    // window and twiddles are needed in pass 1 only: 64 registers of constants that every other phase would carry
    // around (with them resident the kernel spilled 23 VGPRs); 16 ds_read_b128 per lane and tile instead
    for (int i = tid; i < 32 * 32; i += 64 * kWaves) {
        Tw[(i >> 5) * kTwRow + (i & 31)] = t.tw[i];
        Wn[(i >> 5) * kTwRow + (i & 31)] = t.win[i];
    }
    u32x4 ah[2][NS], al[2][NS];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh)
#pragma unroll
        for (int st = 0; st < NS; ++st)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const size_t base = (((size_t)wave * 2 + sh) * NS + st) * 2 * 256;
                ah[sh][st][d] = t.a_bf[base + 0 * 256 + d * 64 + lane];
                al[sh][st][d] = t.a_bf[base + 1 * 256 + d * 64 + lane];
            }
    float ax[kArole];
#pragma unroll
    for (int i = 0; i < kArole; ++i) ax[i] = t.a_role[(wave * kArole + i) * 64 + lane];

    const int lane_slot = fr_id * kHop + n2;
    const int fetcher = tid;

    Cursor cur;
    // XCD-aware tile order (kernel_fused512_w12.hpp): consecutive tiles, whose windows overlap, on one XCD's L2
    const unsigned nwg = gridDim.x;
    const unsigned bid = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    cur.ch = (int)(bid / (unsigned)g.tiles_per_ch);
    cur.t_in = (int)(bid - (unsigned)cur.ch * (unsigned)g.tiles_per_ch);
    cur.ptr = s.pcm + (long long)cur.ch * s.ch_stride + (long long)cur.t_in * kTileHop;

    Fetch fx;
    int shift = 0;
    if (cur.ch < g.n_ch) {
        const Window w0 = window_of(cur, g);
        shift = w0.shift;
        fetch_window(s, w0, fetcher, fx);
        park_window(Sf, fetcher, fx);
    }
    __syncthreads();

    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 lm[kBlocks] = {zero, zero, zero};
    Cursor prev = cur;
    bool have_prev = false;

#ifdef MFCC_F1K_STAMPS
    unsigned long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
    unsigned long long st_tiles = 0;
#endif
    while (cur.ch < g.n_ch) {
        // ---------------- pass 1: windowed real FFT-32 over n1 of the pre-emphasised samples
        v2f ep[16];
        {
            const float *sp = Sf + lane_slot + shift;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[32 * n1];
        }
        const Cursor me = cur;
        advance(cur, g);
        const bool more = cur.ch < g.n_ch;
        int next_shift = 0;
        if (more) {
            const Window wn = window_of(cur, g);
            next_shift = wn.shift;
            fetch_window(s, wn, fetcher, fx);
        }
        // The tail of the previous tile is split over the two waves without a column-16 job: wave 0 sums the partial mel
        // energies and takes their log2 here and leaves them in LDS; wave 3 picks them up behind B1 for the DCT.  (With
        // the whole tail on one wave the other three waited a third of every tile for it.)
#if F1K_SPLIT_TAIL
        if (wave == 0 && have_prev) {
            if (F1K_TAIL_PRIO) __builtin_amdgcn_s_setprio(F1K_TAIL_PRIO);
            mel_log2(Qt, lane, q, lm);
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) *reinterpret_cast<f32x4 *>(Lt + (b * 64 + lane) * 4) = lm[b];
            if (F1K_TAIL_PRIO) __builtin_amdgcn_s_setprio(0);
        }
#else
        if (wave == 3 && have_prev) mel_log2(Qt, lane, q, lm);
#endif
        F1K_STAMP(0);

        v2f ty[16];
        float y16;
        {
            v2f tw[16], wp[16];
            const f32x4 *twr = reinterpret_cast<const f32x4 *>(Tw + n2 * kTwRow);
            const f32x4 *wnr = reinterpret_cast<const f32x4 *>(Wn + n2 * kTwRow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 v = twr[i], u = wnr[i];
                tw[2 * i] = (v2f){v[0], v[1]};
                tw[2 * i + 1] = (v2f){v[2], v[3]};
                wp[2 * i] = (v2f){u[0], u[1]};
                wp[2 * i + 1] = (v2f){u[2], u[3]};
            }
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
        }
        F1K_STAMP(1);
        v2f *tcol0 = reinterpret_cast<v2f *>(Tt + fr_id * kTFrame) + n2;
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
        Vt[fr_id * kVStride + n2] = y16;
        F1K_STAMP(2);
        lds_barrier();                         // B1: T and V of all 8 frames are in LDS; S and Q are consumed
        F1K_STAMP(3);

        // ---------------- pass 2: outputs k2 = 2 m + h of the complex FFT-32 over n2, frame j, column k1
        float pw[16];
        {
            v2f xl[16], xh[16], z[16];
            const f32x4 *trow = reinterpret_cast<const f32x4 *>(Tt + j * kTFrame + (8 * (wave >> 1) + sl + 2 * q) * kTRow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 a = trow[i], b = trow[8 + i];
                xl[2 * i] = (v2f){a[0], a[1]};
                xl[2 * i + 1] = (v2f){a[2], a[3]};
                xh[2 * i] = (v2f){b[0], b[1]};
                xh[2 * i + 1] = (v2f){b[2], b[3]};
            }
            F1K_STAMP(4);
            if (h) mfcc_codelets::cfft32_h1(xl, xh, z);
            else mfcc_codelets::cfft32_h0(xl, xh, z);
#pragma unroll
            for (int m = 0; m < 16; ++m) pw[m] = fmaf(z[m].x, z[m].x, z[m].y * z[m].y);
        }
        F1K_STAMP(5);

        // ---------------- the mel contraction (frame column = (j, sl), K index = q): both column halves' weights
        u32x4 ph[2], pl[2];
#pragma unroll
        for (int gk = 0; gk < 2; ++gk)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t hi, lo;
                split_bf16_pair(pw[kGrpM[gk][2 * d]], pw[kGrpM[gk][2 * d + 1]], hi, lo);
                ph[gk][d] = hi;
                pl[gk][d] = lo;
            }
        F1K_STAMP(6);
        f32x4 acc[2][NS];
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int st = 0; st < NS; ++st) acc[sh][st] = zero;
        // term-major: consecutive MFMAs never share an accumulator
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh)
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    const u32x4 &a = term == 2 ? al[sh][st] : ah[sh][st];
                    const u32x4 &b = term == 1 ? pl[S::grp[st]] : ph[S::grp[st]];
                    acc[sh][st] = MFCC1K8_MFMA_BF(a, b, acc[sh][st]);
                }
        // a lane's columns belong to ONE column half: keep that half's sums, per filter block
        f32x4 fin[kBlocks] = {zero, zero, zero};
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = sl ? acc[1][st][r] : acc[0][st][r];
            fin[S::blk[st]] += v;
        }
        F1K_STAMP(7);

        if (wave == 1 || wave == 2) {
            // column 16 -> bins 16 + 32 j', j' = 8 mb + 2 q + {0, 1}: the column halves split n2 (sl = 1: n2 + 16)
            const float *vp = Vt + j * kVStride + 16 * sl + q;
            f32x4 sp = MFCC1K8_MFMA(ax[0], vp[0], zero);
            f32x4 sp2 = MFCC1K8_MFMA(ax[1], vp[4], zero);
            sp = MFCC1K8_MFMA(ax[2], vp[8], sp);
            sp2 = MFCC1K8_MFMA(ax[3], vp[12], sp2);
            sp += sp2;
            // W1024^(16 (16 + 32 j')) = (-i) (-1)^j': rows (Re, Im) of the even j' become (Im, -Re), of the odd one (-Im, Re)
            const float t0 = sl ? sp[1] : sp[0], t1 = sl ? -sp[0] : sp[1];
            const float t2 = sl ? -sp[3] : sp[2], t3 = sl ? sp[2] : sp[3];
            const float x0 = t0 + other_half(t0), x1 = t1 + other_half(t1);
            const float x2 = t2 + other_half(t2), x3 = t3 + other_half(t3);
            // both halves now hold the frame's bins: only one of them may feed the filters
            const float c0 = sl ? 0.0f : fmaf(x0, x0, x1 * x1);
            const float c1 = sl ? 0.0f : fmaf(x2, x2, x3 * x3);
            if (wave == 2) {
#pragma unroll
                for (int b = 0; b < kBlocks; ++b)
                    if (S::c16[0][b]) {
                        fin[b] = MFCC1K8_MFMA(ax[4 + 2 * b], c0, fin[b]);
                        fin[b] = MFCC1K8_MFMA(ax[5 + 2 * b], c1, fin[b]);
                    }
            } else {
#pragma unroll
                for (int b = 0; b < kBlocks; ++b)
                    if (S::c16[1][b]) {
                        fin[b] = MFCC1K8_MFMA(ax[4 + 2 * b], c0, fin[b]);
                        fin[b] = MFCC1K8_MFMA(ax[5 + 2 * b], c1, fin[b]);
                    }
            }
        }
        if (wave == 3 && have_prev) {
            if (F1K_TAIL_PRIO) __builtin_amdgcn_s_setprio(F1K_TAIL_PRIO);
#if F1K_SPLIT_TAIL
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) lm[b] = *reinterpret_cast<const f32x4 *>(Lt + (b * 64 + lane) * 4);
#endif
            f32x4 d[kBlocks] = {zero, zero, zero};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < kBlocks; ++b) d[b] = MFCC1K8_MFMA(ax[4 * b + r], lm[b][r], d[b]);
            dct_store(s, t, d, lm, prev, j, sl, q, lane, out);
            if (F1K_TAIL_PRIO) __builtin_amdgcn_s_setprio(0);
        }
        F1K_STAMP(8);
        // a frame's sums lie in both column halves: add them here (one DPP add per register), so that the tail reads
        // one lane's worth per wave
#pragma unroll
        for (int b = 0; b < kBlocks; ++b) {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = F1K_PRESUM ? fin[b][r] + other_half(fin[b][r]) : fin[b][r];
            *reinterpret_cast<f32x4 *>(Qt + ((wave * kBlocks + b) * 64 + lane) * 4) = o;
        }
        prev = me;
        have_prev = true;
        if (more) park_window(Sf, fetcher, fx);
        shift = next_shift;
        F1K_STAMP(9);
        lds_barrier();                         // B2: partial sums and S are in LDS, T/V may be overwritten
        F1K_STAMP(10);
#ifdef MFCC_F1K_STAMPS
        ++st_tiles;
#endif
    }
#ifdef MFCC_F1K_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 12; ++i) atomicAdd(&g_stamps1k[wave * 12 + i], st_sum[i]);
        atomicAdd(&g_stamps1k[48 + wave], st_tiles);
    }
#endif
    if (wave == 3 && have_prev) {                  // the last tile's tail, whole
        mel_log2(Qt, lane, q, lm);
        f32x4 d[kBlocks] = {zero, zero, zero};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) d[b] = MFCC1K8_MFMA(ax[4 * b + r], lm[b][r], d[b]);
        dct_store(s, t, d, lm, prev, j, sl, q, lane, out);
    }
}

inline const char *kernel_name() { return "mfcc_fused1024_kernel"; }

inline bool launch(const mfcc_k::StreamDesc &s, const Tables &t, float *out, int n_cu, hipStream_t stream) {
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    if (n_tiles >= (1ll << 31) || tiles_per_ch >= (1ll << 26) || n_ch >= (1ll << 31)) return false;
    long long per_cu = 2;                                              // two workgroups per CU (70 KB of LDS each)
    if (const char *e = std::getenv("MFCC_HIP_F1K_GRID")) per_cu = std::atoi(e) > 0 ? std::atoi(e) : 2;   // diagnostic
    long long grid = n_tiles < per_cu * n_cu ? n_tiles : per_cu * n_cu;
    if (grid < 1) grid = 1;
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
    switch (t.variant) {
    case 1: hipLaunchKernelGGL(mfcc_fused1024_kernel<1>, dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t, g, out); break;
    case 2: hipLaunchKernelGGL(mfcc_fused1024_kernel<2>, dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t, g, out); break;
    default: hipLaunchKernelGGL(mfcc_fused1024_kernel<0>, dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t, g, out); break;
    }
    return true;
}

}  // namespace mfcc_f1k
