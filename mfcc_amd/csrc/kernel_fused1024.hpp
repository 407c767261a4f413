// Fused 1024/341/40 float kernel for gfx950 (MI355X) -- BASELINE.json configs[3]: nfft 1024, hop 1024 // 3 = 341
// (mfcc/core/mfcc.py:43), 40 mel bands, n_cep <= 40 (the reference tops keep nceptrums = nfilters), the mel contraction on
// the matrix cores.  Same scheme as kernel_fused512.hpp (read its header first); what differs:
//
//  * a workgroup of EIGHT waves owns a tile of 16 consecutive frames, one workgroup per CU (117 KB of LDS: a 16-frame tile
//    of 1024-point frames is twice the data);
//  * pass 1: n = 32 n1 + n2.  Wave w, lane (f = lane >> 5, n2 = lane & 31) owns frame 2 w + f and runs the same
//    register-resident REAL 32-point FFT over n1 (codelet rfft32_tw, Hamming folded in), twiddles columns 0..15 by
//    W1024^(n2 k1) and writes T[frame][k1][n2] -- the lanes of a store are consecutive n2; column 16 (real) goes to V;
//  * pass 2: the complex 32-point FFT over n2 of a column is split by ONE decimation-in-frequency step into its even and
//    odd outputs, two lanes per column: wave w, lane (j = lane & 15, q = lane >> 4) takes column k1 = 4 (w >> 1) + q of
//    frame j and the outputs k2 = 2 m + h, h = w & 1 (codelets cfft32_h0 / cfft32_h1).  The column's 32 values are
//    contiguous in T: 16 ds_read_b128 (256 B/clk; round 2 read T[frame][n2][k1] with 16 ds_read2_b64 at 128).
//    X[k1 + 32 k2] is bin k1 + 32 k2 or, by the symmetry of a real signal, bin 1024 - (k1 + 32 k2): 512 lanes x 16 outputs =
//    every bin once;
//  * |X|^2 is again in the MFMA B-operand layout.  The mel contraction runs on v_mfma_f32_16x16x32_bf16 with both
//    operands split in two bf16 terms (W = Wh + Wl, P = Ph + Pl; Wh Ph + Wh Pl + Wl Ph, fp32 accumulation, 2^-17
//    relative): a lane's 16 outputs are two K groups of eight -- X = the eight lowest in frequency (m = 0..3, 12..15),
//    Y = the rest -- and a wave issues one MFMA triple per (K group, 16-filter block) SET with non-zero weights: (X, 0),
//    (X, 1), (Y, 2) at every sample rate, plus (Y, 1) [<= 22.05 kHz: the second block reaches past bin 256], (X, 2)
//    [44.1 / 48 kHz: the third block starts below it] or both [32 kHz] -- THREE instantiations cover every common rate
//    (round 2: five per-rate lists of 17-19 fp32 MFMAs of 32 clocks, during which the SIMD issues no vector instruction;
//    44.1 / 48 kHz ran on the generic kernel, six times slower);
//  * column 16 -> bins 16 + 32 j' by a 32-point DFT matrix on the fp32 matrix cores, split over two waves (role 1:
//    j' = 0..7, role 2: j' = 8..15; 8 MFMAs + their mel MFMAs each); role 0 finishes the previous tile (log2, DCT-II as 12
//    fp32 MFMAs per 16 coefficients, store) and neither fetches nor parks samples.
//
// This form serves every sample rate and coefficient count.  Since the second half of round 3 this file is (a) the tables, set
// lists and helpers of the kernel that runs -- the TWELVE-wave staging of this contraction, kernel_fused1024_w12.hpp
// (mfcc_fused1024_w12bf_kernel: 1.00 ms against 1.36 for the kernel below on config 4's shape) -- and (b) the eight-wave
// LOCKSTEP staging below, an A/B form (MFCC_HIP_FUSED1024=bf16).  What else was built and measured in round 3 and dropped
// (DESIGN.md 4.2): 8-frame tiles in two 4-wave workgroups per CU (every per-tile cost twice, MFMAs half empty: 260 vector
// instructions per frame against 195, 9-13 % slower), and producer / consumer waves on the 16-frame tile (pass 1 of tile k + 1
// in registers beside pass 2 of tile k: the stores of T behind the barrier are an LDS-bound interval nothing overlaps, 12 %
// slower).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "codelets_gen.hpp"
#include "fused_common.hpp"
#include "kernels_generic.hpp"
#include "tables.hpp"

namespace mfcc_fused1024 {

constexpr int kNfft = 1024, kHop = 341, kMel = 40, kMaxCep = 40;
constexpr int kTile = 16, kWaves = 8;
constexpr int kTileHop = kTile * kHop;            // 5456 samples between consecutive tiles
constexpr int kTRow = 64;                         // words per k1 row of the transpose tile T[frame][k1][n2]: 32 complex
constexpr int kTFrame = 16 * kTRow + 4;           // 1028 words per frame: (row, frame) strides of (16, 257) 16-byte units make
                                                  // every ds_read_b128 of pass 2 conflict free (brute-forced over the lane groups)
constexpr int kVStride = 34;                      // words per frame in the column-16 tile
constexpr int kBlocks = 3;                        // 16-filter blocks of the 40 filters
constexpr int kQWords = kWaves * kBlocks * 256;   // partial mel sums: [wave][block][lane * 4]
constexpr int kFetchers = 64 * (kWaves - 1);      // roles 1..7 fetch and park the sample window
constexpr int kPieces = (7 + (kTile - 1) * kHop + kNfft + 7) / 8;       // 769 pieces of 8 samples are read
constexpr int kSecond = kPieces - kFetchers;      // fetchers that take a second piece (321)
constexpr int kSUsed = 8 * kPieces;               // 6152 fp32 slots
constexpr int kTwRow = 36;                        // words per n2 row of the twiddle table in LDS (9 16-byte units: the
                                                  // 16 lanes of a ds_read_b128 group hit 16 different units mod 16)
constexpr int kLdsWords = kTile * kTFrame + kTile * kVStride + kQWords + kSUsed + 32 * kTwRow;
constexpr int kAextra = 14;                       // role operands: DCT rows (12) / column-16 DFT (8) + its mel weights (6)

// K slots of the two bf16 MFMAs of a lane: K index 8 q + i  <->  output m = kGrpM[grp][i] of lane group q
constexpr int kGrpM[2][8] = {{0, 1, 2, 3, 12, 13, 14, 15}, {4, 5, 6, 7, 8, 9, 10, 11}};

// (K group, filter block) sets of a variant; c16[mb][blk]: the filter blocks that column 16's bins 16 + 32 j' reach,
// j' < 8 (mb = 0, bins 16..240) / j' >= 8 (mb = 1, bins 272..496).  build_tables picks the first variant that covers the
// rate's matrix (and verifies it against the matrix itself).
template <int VAR> struct Sets;
template <> struct Sets<0> {                      // <= 22.05 kHz (16 kHz: the reference's rate and BASELINE's config 4)
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
constexpr int kVariants = 3;
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
    const uint32_t *a_bf;  // [8 waves][sets][hi, lo][4 dwords][64 lanes] mel weights as bf16 pairs
    const float *a_extra;  // [3 roles][kAextra][64]  role 0: DCT rows 0..15; role 1 / 2: column-16 DFT (8) + its mel weights (6)
    const float *a_dct_hi; // [2][12][64]  DCT rows of coefficients 16..31 and 32..47 (fetched per tile, only when n_cep > 16:
                           // the kernel has no registers to keep them)
    const uint32_t *a_bf4; // the same mel weights as [8 waves][sets][hi, lo][64 lanes][4 dwords]: one 16-byte load per
                           // lane and operand for the twelve-wave form, which streams them (kernel_fused1024_w12.hpp)
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
    std::vector<float> win(32 * 32), tw(32 * 16 * 2), aext(size_t(3) * kAextra * 64, 0.0f), adct(size_t(2) * 12 * 64, 0.0f);
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
        if (md[size_t(f) * nb] != 0.0) return false;      // weight on the real-valued DC bin: not summed in fp32;
                                                          // no common rate has it at 1024 points / 40 filters
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
    // mel operands: lane l of wave wv, set st holds rows l & 15 of filter block blk[st] at K slots i = 0..7
    // <-> bin(k1 = 4 (wv >> 1) + (l >> 4), h = wv & 1, m = kGrpM[grp[st]][i]); dword d = slots (2 d, 2 d + 1)
    std::vector<uint32_t> abf(size_t(kWaves) * sv.n * 2 * 4 * 64, 0u);
    for (int wv = 0; wv < kWaves; ++wv)
        for (int st = 0; st < sv.n; ++st)
            for (int l = 0; l < 64; ++l) {
                uint32_t hi[8], lo[8];
                for (int i = 0; i < 8; ++i) {
                    const int filt = sv.blk[st] * 16 + (l & 15), k1 = 4 * (wv >> 1) + (l >> 4);
                    const int bin = bin_of(k1, wv & 1, kGrpM[sv.grp[st]][i]);
                    float wgt = 0.0f;
                    if (bin >= 0 && filt < kMel) {
                        wgt = float(Wt(filt, bin));
                        covered[size_t(filt) * nb + bin] = 1;
                    }
                    hi[i] = bf16_round(wgt);
                    lo[i] = bf16_round(wgt - bf16_val(hi[i]));
                }
                const size_t base = (size_t(wv) * sv.n + st) * 2 * 256;
                for (int d = 0; d < 4; ++d) {
                    abf[base + 0 * 256 + d * 64 + l] = hi[2 * d] | (hi[2 * d + 1] << 16);
                    abf[base + 1 * 256 + d * 64 + l] = lo[2 * d] | (lo[2 * d + 1] << 16);
                }
            }
    auto E = [&](int role, int idx, int lane) -> float & { return aext[(size_t(role) * kAextra + idx) * 64 + lane]; };
    // role 0 -- DCT rows: lane (coeff = l & 15, g = l >> 4) holds D[16 tile + coeff][16 blk + 4 g + r]
    std::vector<double> dd = dct_rows(n_cep, kMel, lifter);                   // [n_cep][40]
    for (int tile = 0; tile < 3; ++tile)
        for (int blk = 0; blk < kBlocks; ++blk)
            for (int r = 0; r < 4; ++r)
                for (int l = 0; l < 64; ++l) {
                    const int coeff = 16 * tile + (l & 15), filt = 16 * blk + 4 * (l >> 4) + r;
                    const float v = (coeff < n_cep && filt < kMel) ? float(dd[size_t(coeff) * kMel + filt]) : 0.0f;
                    if (tile == 0) E(0, 4 * blk + r, l) = v;
                    else adct[(size_t(tile - 1) * 12 + 4 * blk + r) * 64 + l] = v;
                }
    // roles 1, 2 -- column 16: X[16 + 32 j'] = sum_n2 v[n2] W1024^(n2 (16 + 32 j')), j' = 8 (role - 1) + 0..7; MFMA row
    // i = 4 g + r holds r = 0: Re j' = jb + 2 g, r = 1: Im (same j'), r = 2: Re j' + 1, r = 3: Im; K step t covers n2 = 4 t + (l >> 4)
    for (int role = 1; role <= 2; ++role) {
        const int mb = role - 1;
        for (int t = 0; t < 8; ++t)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, n2 = 4 * t + (l >> 4);
                const int g = i >> 2, r = i & 3, jp = 8 * mb + 2 * g + (r >> 1);
                const double th = 2.0 * kPi * double(n2 * (16 + 32 * jp)) / 1024.0;
                E(role, t, l) = float((r & 1) ? -std::sin(th) : std::cos(th));
            }
        // its bins as K steps of fp32 MFMAs: lane group q supplies bin 16 + 32 (8 mb + 2 q + step) to block blk
        for (int blk = 0; blk < kBlocks; ++blk)
            for (int step = 0; step < 2; ++step)
                for (int l = 0; l < 64; ++l) {
                    const int filt = blk * 16 + (l & 15), bin = 16 + 32 * (8 * mb + 2 * (l >> 4) + step);
                    if (filt >= kMel || !sv.c16[mb][blk]) continue;
                    E(role, 8 + 2 * blk + step, l) = float(Wt(filt, bin));
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
    put(aext.data(), aext.size() * 4);
    put(adct.data(), adct.size() * 4);
    put(abf.data(), abf.size() * 4);
    std::vector<uint32_t> abf4(abf.size());
    for (size_t op = 0; op < abf.size() / 256; ++op)
        for (int d = 0; d < 4; ++d)
            for (int l = 0; l < 64; ++l) abf4[op * 256 + l * 4 + d] = abf[op * 256 + d * 64 + l];
    put(abf4.data(), abf4.size() * 4);
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
    t.a_extra = f;    f += 3 * kAextra * 64;
    t.a_dct_hi = f;   f += 2 * 12 * 64;
    t.a_bf = reinterpret_cast<const uint32_t *>(f);
    t.a_bf4 = t.a_bf + (size_t)kWaves * sets_view(variant).n * 2 * 256;
}

// ---- device (helpers shared in spirit with kernel_fused512.hpp; kept local so the two kernels stay independent)

#define MFCC1K_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFCC1K_MFMA_BF(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

struct Fetch {
    i32x4 v0, v1;
    int p0, p1;
};

// fetcher u (0..447) takes pieces u and 448 + u (< 769) of the window, plus the dword in front of each
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

// (a, b) -> their bf16 roundings packed in one dword (a low) and the bf16 roundings of what the first rounding lost
__device__ __forceinline__ void split_bf16_pair(float a, float b, uint32_t &hi, uint32_t &lo) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
}

// summed mel energies of a finished tile and their log2; register r of block b is filter 16 b + 4 q + r of
// frame lo.  Filters 40..47 do not exist: their (zero) sums must not reach the DCT as -inf * 0
__device__ __forceinline__ void mel_log2(const float *Qt, int lane, int q, f32x4 (&lm)[kBlocks]) {
    const f32x4 *Q4 = reinterpret_cast<const f32x4 *>(Qt) + lane;
#pragma unroll
    for (int b = 0; b < kBlocks; ++b) {
        f32x4 m = Q4[(0 * kBlocks + b) * 64];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) m += Q4[(w * kBlocks + b) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) lm[b][r] = __builtin_amdgcn_logf(m[r]);
    }
    if (q >= 2) lm[2] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// d[0] + d[1] + d[2] = coefficients 0..15 of the previous tile (their 12 MFMAs are issued by the caller); 16..31 and
// 32..39 are further M tiles whose A operands are fetched here (uniform branches; 12 coalesced dwords per lane out of L2)
__device__ __forceinline__ void dct_store(const mfcc_k::StreamDesc &s, const Tables &t, const f32x4 (&d)[kBlocks],
                                          const f32x4 (&lm)[kBlocks], const Cursor &c, int lo, int q, int lane,
                                          int lane_off, float *__restrict__ out) {
    const long long fr0 = (long long)c.t_in * kTile;
    const long long rows_left = s.frames_per_ch - fr0;
    float *o = out + ((long long)c.ch * s.frames_per_ch + fr0) * t.n_cep + lane_off;
    const bool mine = lo < rows_left;
    if (mine) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < t.n_cep) o[r] = (d[0][r] + d[1][r]) + d[2][r];
    }
    for (int tile = 1; 16 * tile < t.n_cep; ++tile) {
        const float *hi = t.a_dct_hi + (size_t)(tile - 1) * 12 * 64 + lane;
        asm volatile("" : "+v"(hi));                   // not hoisted out of the tile loop: no registers to hold it
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 e[kBlocks] = {zero, zero, zero};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) e[b] = MFCC1K_MFMA(hi[(4 * b + r) * 64], lm[b][r], e[b]);
        if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * tile + 4 * q + r < t.n_cep) o[16 * tile + r] = (e[0][r] + e[1][r]) + e[2][r];
        }
    }
}

template <int VAR>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_fused1024_kernel(mfcc_k::StreamDesc s, Tables t, LaunchGeom g, float *__restrict__ out) {
    using S = Sets<VAR>;
    constexpr int NS = S::N;
    __shared__ __attribute__((aligned(16))) float lds[kLdsWords];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave;
    const int h = wave & 1;
    const int lo = lane & 15;          // frame column in pass 2 and the MFMA window
    const int q = lane >> 4;           // column offset in pass 2; K index in the MFMA window
    const int n2 = lane & 31;          // pass 1
    const int fr_id = 2 * wave + (lane >> 5);

    float *const Tt = lds;                                         // [16 frames][1028]: [16 k1][64] each
    float *const Vt = Tt + kTile * kTFrame;                        // [16 frames][34]
    float *const Qt = Vt + kTile * kVStride;                       // [8 waves][3 blocks][256]
    float *const Sf = Qt + kQWords;                                // pre-emphasised sample window, fp32

    v2f wp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wp[i] = reinterpret_cast<const v2f *>(t.win)[n2 * 16 + i];
    // the twiddles are needed at the end of pass 1 only: 32 registers of constants that every other phase would carry
    // around (with them resident the kernel spilled 16 VGPRs); 8 ds_read_b128 per lane and tile instead
    float *const Tw = Sf + kSUsed;                                 // [32 n2][36]: W1024^(n2 k1), k1 = 0..15
    for (int i = tid; i < 32 * 32; i += 64 * kWaves) Tw[(i >> 5) * kTwRow + (i & 31)] = t.tw[i];
    u32x4 ah[NS], al[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const size_t base = ((size_t)wave * NS + st) * 2 * 256;
            ah[st][d] = t.a_bf[base + 0 * 256 + d * 64 + lane];
            al[st][d] = t.a_bf[base + 1 * 256 + d * 64 + lane];
        }
    float ax[kAextra];
#pragma unroll
    for (int i = 0; i < kAextra; ++i) ax[i] = role < 3 ? t.a_extra[(role * kAextra + i) * 64 + lane] : 0.0f;

    const int lane_slot = fr_id * kHop + n2;
    const int fetcher = (role - 1) * 64 + lane;     // 0..447 in roles 1..7
    const int lane_off = lo * t.n_cep + 4 * q;

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
        if (role != 0) {
            fetch_window(s, w0, fetcher, fx);
            park_window(Sf, fetcher, fx);
        }
    }
    __syncthreads();

    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 lm[kBlocks] = {zero, zero, zero};
    Cursor prev = cur;
    bool have_prev = false;

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
            if (role != 0) fetch_window(s, wn, fetcher, fx);
        }
        if (role == 0 && have_prev) mel_log2(Qt, lane, q, lm);

        v2f ty[16];
        float y16;
        {
            v2f tw[16];
            const f32x4 *twr = reinterpret_cast<const f32x4 *>(Tw + n2 * kTwRow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 v = twr[i];
                tw[2 * i] = (v2f){v[0], v[1]};
                tw[2 * i + 1] = (v2f){v[2], v[3]};
            }
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
        }

        v2f *tcol0 = reinterpret_cast<v2f *>(Tt + fr_id * kTFrame) + n2;      // a store's lanes are consecutive n2
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
        Vt[fr_id * kVStride + n2] = y16;
        lds_barrier();                         // B1: T and V of all 16 frames are in LDS; S and Q are consumed

        // ---------------- pass 2: outputs k2 = 2 m + h of the complex FFT-32 over n2, frame lo, column k1
        float pw[16];
        {
            v2f xl[16], xh[16];
            const f32x4 *trow = reinterpret_cast<const f32x4 *>(Tt + lo * kTFrame + (4 * (wave >> 1) + q) * kTRow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 a = trow[i], b = trow[8 + i];
                xl[2 * i] = (v2f){a[0], a[1]};
                xl[2 * i + 1] = (v2f){a[2], a[3]};
                xh[2 * i] = (v2f){b[0], b[1]};
                xh[2 * i + 1] = (v2f){b[2], b[3]};
            }
            v2f pp[8];                               // (|z[m]|^2, |z[m + 8]|^2): the codelet's last layer is transposed
            if (h) mfcc_codelets::cfft32_h1_pow(xl, xh, pp);
            else mfcc_codelets::cfft32_h0_pow(xl, xh, pp);
#pragma unroll
            for (int m = 0; m < 8; ++m) pw[m] = pp[m].x, pw[m + 8] = pp[m].y;
        }

        // ---------------- MFMA window (frame column = lo, K index = q): the mel contraction on bf16 pairs
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
        f32x4 acc[NS];
#pragma unroll
        for (int st = 0; st < NS; ++st) acc[st] = zero;
        f32x4 d[kBlocks] = {zero, zero, zero};
        // term-major: consecutive MFMAs never share an accumulator; role 0 rides the previous tile's 12 DCT MFMAs (fp32)
        // on the chain, one behind each of the first twelve
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const u32x4 &a = term == 2 ? al[st] : ah[st];
                const u32x4 &b = term == 1 ? pl[S::grp[st]] : ph[S::grp[st]];
                acc[st] = MFCC1K_MFMA_BF(a, b, acc[st]);
                const int i = term * NS + st;
                if (role == 0 && i < 12) d[i % 3] = MFCC1K_MFMA(ax[4 * (i % 3) + i / 3], lm[i % 3][i / 3], d[i % 3]);
            }
        f32x4 fin[kBlocks] = {zero, zero, zero};
#pragma unroll
        for (int st = 0; st < NS; ++st) fin[S::blk[st]] += acc[st];

        if (role == 1 || role == 2) {
            // column 16 -> bins 16 + 32 j', j' = 8 (role - 1) + 2 q + {0, 1}, fed to the filter blocks from registers
            const float *vp = Vt + lo * kVStride + q;
            f32x4 sp = zero, sp2 = zero;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                sp = MFCC1K_MFMA(ax[k], vp[4 * k], sp);
                sp2 = MFCC1K_MFMA(ax[k + 1], vp[4 * (k + 1)], sp2);
            }
            sp += sp2;
            const float c0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);
            const float c1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);
            if (role == 1) {
#pragma unroll
                for (int b = 0; b < kBlocks; ++b)
                    if (S::c16[0][b]) {
                        fin[b] = MFCC1K_MFMA(ax[8 + 2 * b], c0, fin[b]);
                        fin[b] = MFCC1K_MFMA(ax[9 + 2 * b], c1, fin[b]);
                    }
            } else {
#pragma unroll
                for (int b = 0; b < kBlocks; ++b)
                    if (S::c16[1][b]) {
                        fin[b] = MFCC1K_MFMA(ax[8 + 2 * b], c0, fin[b]);
                        fin[b] = MFCC1K_MFMA(ax[9 + 2 * b], c1, fin[b]);
                    }
            }
        }
        if (role == 0 && have_prev) dct_store(s, t, d, lm, prev, lo, q, lane, lane_off, out);
#pragma unroll
        for (int b = 0; b < kBlocks; ++b)
            *reinterpret_cast<f32x4 *>(Qt + ((wave * kBlocks + b) * 64 + lane) * 4) = fin[b];
        prev = me;
        have_prev = true;
        if (more && role != 0) park_window(Sf, fetcher, fx);
        shift = next_shift;
        lds_barrier();                         // B2: partial sums and S are in LDS, T/V may be overwritten
    }
    if (role == 0 && have_prev) {
        mel_log2(Qt, lane, q, lm);
        f32x4 d[kBlocks] = {zero, zero, zero};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) d[b] = MFCC1K_MFMA(ax[4 * b + r], lm[b][r], d[b]);
        dct_store(s, t, d, lm, prev, lo, q, lane, lane_off, out);
    }
}

inline const char *kernel_name() { return "mfcc_fused1024_kernel"; }

inline bool launch(const mfcc_k::StreamDesc &s, const Tables &t, float *out, int n_cu, hipStream_t stream) {
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    if (n_tiles >= (1ll << 31) || tiles_per_ch >= (1ll << 26) || n_ch >= (1ll << 31)) return false;
    long long grid = n_tiles < (long long)n_cu ? n_tiles : (long long)n_cu;
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

}  // namespace mfcc_fused1024
