// Fused 512/170/32 float kernel for gfx950 (MI355X): the whole chain of mfcc/core --
// pre-emphasis -> 512-sample frames (hop 170) -> Hamming -> FFT -> |.|^2 -> 32 mel -> log2 ->
// DCT-II -> first n_cep (<= 32: the driver and the notebook keep all 32) -- in one launch.
//
// Work unit: a workgroup of 4 waves owns a tile of 16 consecutive frames (the N dimension of
// v_mfma_f32_16x16x4_f32).  Per tile:
//
//  input   the tile's contiguous sample span (16 frames = 3062 samples, ~6 KB) is fetched by the
//          whole workgroup with 16-byte loads -- each sample crosses HBM/L2 once per tile -- one tile
//          ahead; pre-emphasis 32 x[i] - 31 x[i-1] (one v_dot2_i32_i16, exact) is applied once per
//          sample when the window is parked in LDS as fp32, just before the second barrier of the
//          previous tile.
//  pass 1  wave w, lane (q = lane>>4, n2 = lane&15) owns frame w + 8 (q&1) + 4 (q>>1) and reads its
//          32 samples i = 16 n1 + n2 from the window (ds_read_b32, conflict free).  A register-resident
//          REAL 32-point FFT over n1 with the Hamming window folded into its first butterfly layer
//          (packed fp32, codelets_gen.hpp) gives Y[k1, n2], k1 = 0..16.  Columns 0..15 are multiplied
//          by W512^(n2 k1) and written to the LDS transpose tile T[frame][n2][k1]; column 16 (real)
//          goes to the LDS tile V.
//  ---- workgroup barrier B1 ----
//  pass 2  wave w, lane (j = lane&15, g = lane>>4) reads column k1 = 4 w + g of frame j from T and
//          runs a complex 16-point FFT over n2: X[k1 + 32 k2], k2 = 0..15.  The input being real,
//          each of these is a distinct needed bin (k or 512-k): no real-FFT split step.
//  MFMA    |X|^2 is now in registers in exactly the B-operand layout of the matrix instructions (column = frame j,
//          K index = g): the mel contraction W (32 x 257, block-banded) . P runs straight from registers (A = this
//          wave's weights, resident for the whole kernel) -- the power spectrum never touches LDS.  It runs on
//          v_mfma_f32_16x16x32_bf16 with both operands split in two bf16 terms (W = Wh + Wl, P = Ph + Pl; Wh Ph +
//          Wh Pl + Wl Ph, fp32 accumulation: 2^-17 relative, far inside the fp32 noise of the FFT): 9 MFMAs of 16
//          clocks per wave and tile instead of 17 fp32 ones of 32 -- on gfx950 an fp32 MFMA holds the SIMD's vector
//          pipe for its whole duration (tools/alu_probe.hip), a bf16 one does not.  Each wave ends with partial sums
//          over its 64 bins; they meet in the LDS tile Q.  One wave ("role 1") also turns column 16 into bins
//          16+32j with a 16x16 real DFT matrix (4 fp32 MFMAs) and feeds them (4).  Another ("role 0") finishes the
//          PREVIOUS tile in this window: log2 of the summed mel energies, DCT-II as 8 fp32 MFMAs (16 for n_cep > 16)
//          whose B operand IS the mel accumulator layout (K index permuted, no lane movement), interleaved with its
//          mel MFMAs; store n_cep floats per frame.  Nobody waits for that tail.
//  ---- workgroup barrier B2 ----
//
// Instantiations: the BANDED set list (struct SetsBf) is the mel matrix at 16 kHz; DENSE issues every (filter block,
// K group) pair and serves every other sample rate and the 16-filter bank; DCX adds the double-precision DC bin.
//
// HBM traffic per frame: 170 new int16 samples + 13 floats out = 392 B (plus the 342-sample overlap
// between consecutive tiles, 11 %).  The kernel is bound by the CU's VALU and LDS pipes (86 VALU
// wave-instructions and 8 KB of LDS traffic per frame), not by HBM; DESIGN.md has the accounting.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "codelets_gen.hpp"
#include "fused_common.hpp"
#include "kernels_generic.hpp"
#include "tables.hpp"

namespace mfcc_fused {

constexpr int kNfft = 512, kHop = 170, kMel = 32, kMaxCep = 32;   // 32 = all of them (main.c:13, notebook cell 39)
constexpr int kTile = 16;                 // frames per workgroup tile (MFMA N dimension)
constexpr int kWaves = 4;
constexpr int kTileHop = kTile * kHop;    // 2720 samples between consecutive tiles
constexpr int kTRow = 36;                 // words per k1 row of the transpose tile T[frame][k1][n2]: 16 complex + 4
constexpr int kTFrame = 16 * kTRow + 8;   // 584 words per frame.  Pass 1 stores T[frame][k1][n2] with the lanes of a store on
                                          // consecutive n2; pass 2 reads a column's 16 values -- contiguous -- with 8
                                          // ds_read_b128 (256 B/clk; rounds 1-2: T[frame][n2][k1], 8 ds_read2_b64 at 128).
                                          // (row, frame) strides of (9, 146) 16-byte units: conflict free (brute-forced
                                          // over the ds_read_b128 lane groups)
constexpr int kVStride = 18;              // words per frame in the column-16 tile
constexpr int kAmelBanded = 17;           // mel A operands per wave at 16 kHz: block 0 k2 = 0,1,14,15; block 1 k2 = 2..14
constexpr int kAmelDense = 32;            // any other band structure: every (k2, block) pair
// MFCC_MEL_BF16: the mel contraction on v_mfma_f32_16x16x32_bf16 with both operands split in two bf16 terms
// (W = Wh + Wl, P = Ph + Pl, products Wh Ph + Wh Pl + Wl Ph: 2^-17 relative, fp32 accumulation).  On gfx950 the fp32
// MFMA is vector-pipe time (tools/alu_probe.hip: an MFMA and k VALU ops of ONE wave take 32 + 4.5 k clocks, and a wave
// issuing them back to back holds its SIMD partner to one VALU op per MFMA), so 17 of them per wave and tile were 40 %
// of the kernel's pipe clocks; a K = 32 bf16 MFMA takes 16 clocks on the matrix pipe and covers eight times the bins.
#ifndef MFCC_MEL_BF16
#define MFCC_MEL_BF16 1
#endif
// K slots of the two bf16 MFMAs of a lane: K index 8 g + j  <->  bin(w, g, k2 = kGrp[grp][j])
constexpr int kGrpK2[2][8] = {{0, 1, 14, 15, 2, 3, 12, 13}, {4, 5, 6, 7, 8, 9, 10, 11}};
// (filter block, K group) sets of a wave.  BANDED (16 kHz): block 0 only touches k2 in {0, 1, 14, 15} -- group 0
template <bool DENSE>
struct SetsBf;
template <>
struct SetsBf<false> {
    static constexpr int N = 3;
    static constexpr int blk[N] = {0, 1, 1};
    static constexpr int grp[N] = {0, 0, 1};
};
template <>
struct SetsBf<true> {
    static constexpr int N = 4;
    static constexpr int blk[N] = {0, 0, 1, 1};
    static constexpr int grp[N] = {0, 1, 0, 1};
};
constexpr int kAextra = 16;               // role operands: role 0 DCT (8 + 8 for coefficients 16..31); role 1
                                          // column-16 DFT (4) + its mel (4)
constexpr int kFetchers = 192;            // threads that fetch and park the sample window: roles 1..3
constexpr int kSUsed = 2 * 8 * kFetchers; // 3072 fp32 slots of the window (7 + 15 * 170 + 512 = 3069 are read)
constexpr int kQWords = kWaves * 2 * 256;  // partial mel sums: [wave][block][lane*4]
constexpr int kLdsWords = kTile * kTFrame + kTile * kVStride + kQWords + kSUsed;
// DCX instantiation only (a mel filter with weight on bin 0, see FusedTables::win_dc): the double-precision
// window rows [16 n2][32 n1] at a row stride of 34 doubles (lane n2 reads 16 bytes at 272 n2 + 16 i: the 16
// lanes of a ds_read_b128 group cover all 64 banks), and the per-lane partial sums [16 frames][16 n2] at a
// row stride of 17 doubles (lane j reads row j pair by pair: 34 j mod 64 are 16 distinct bank pairs)
constexpr int kWdRow = 34;
constexpr int kDcRow = 17;
constexpr int kDcxWords = 2 * (16 * kWdRow + kTile * kDcRow);

using mfcc_fc::f32x4;
using mfcc_fc::i32x4;
using mfcc_fc::Cursor;
using mfcc_fc::LaunchGeom;
using mfcc_fc::Window;
using mfcc_fc::advance;
using mfcc_fc::window_of;
using mfcc_fc::preemph8;
using mfcc_fc::lds_barrier;
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int kDcDigits = 7;

struct FusedTables {
    const float *win;     // [16 n2][32 n1]   hamming[16 n1 + n2] / 64 (pre-emphasis x32, real-FFT split x2)
    const float2 *tw;     // [16 n2][16 k1]   W512^(n2 k1)
    const float *a_mel;   // [4 waves][17][64] mel weights of the bins wave w transforms, in consumption order
    const uint32_t *a_mel_bf; // [4 waves][sets][hi, lo][4 dwords][64] the same weights as bf16 pairs (MFCC_MEL_BF16)
    const float *a_extra; // [4 roles][8][64]  role 0: DCT rows; role 1: column-16 DFT + its mel weights
    const double *win_dc; // [16 n2][32 n1]   hamming[16 n1 + n2] / 32 in double -- or nullptr.  Set when a mel filter has
                          // weight on bin 0 (any sample rate whose first two filter points are both 0: 44.1 kHz, 48 kHz ...).
                          // X[0] = sum w e is REAL: it comes arbitrarily close to 0 by cancellation, the log turns its
                          // relative error into an absolute one, and an fp32 sum is then off by 1e-6 rms / |X[0]|
                          // (measured: 2.2 in log2 units at |X[0]| = 5e-6 rms, profiles/r02_dc_band.json) -- so bin 0
                          // alone is accumulated in double (every other bin is complex and does not cancel that way)
    // twelve-wave form of the DC path: the workers' mel weights with bin 0 taken out, the window in double in sample
    // order (hamming[n] / 32, n = 0 .. 256: the window is symmetric about 256), bin 0's weight per filter
    const uint32_t *a_mel_bf_nodc;
    const double *win_dc_lin;
    const float *w_dc;    // [32]
    // ... and its integer form (twelve-wave kernel): X[0] = sum_m c[m] x[m] over the RAW samples x[-1 .. 511]
    // (c[m] = w[m] - 31/32 w[m+1], the pre-emphasis folded into the window), c as 49-bit integers C = round(c 2^B) in
    // seven balanced base-128 digits, the raw bytes as they lie in memory: one v_mfma_i32_16x16x64_i8 chain per tile,
    // exact in int32.  [17 k-blocks][64 lanes][4 dwords] A operands; dc_consts = {2^-B, lo, hi of 128 sum(C) = hi 2^28 + lo}
    const uint32_t *a_dc_i8;
    const double *dc_consts;
    // the DCT rows as bf16 pairs for the twelve-wave kernel's tail: [2 M tiles (coefficients 0..15, 16..31)][hi, lo][4 dwords]
    // [64 lanes]; lane (row = l & 15, q = l >> 4) holds D[16 tile + row][f(q, j)], j = 0..7, K slot 8 q + j <-> filter
    // f(q, j) = 4 q + j (j < 4) or 16 + 4 q + j - 4: the order in which a lane of the tail holds its eight log-mel values
    const uint32_t *a_dct_bf;
    int n_cep;
    int n_mel;            // 32, or 16: block 1 does not exist (its zero sums must not reach the DCT as -inf * 0)
};

// ---- the mel contraction.  After pass 2, lane (j, g) of wave w holds |X|^2 of frame j at the 16
// bins  bin(w, g, k2) = k1 + 32 k2 (k2 < 8)  or  32 (16 - k2) - k1 (k2 >= 8),  k1 = 4 w + g.  MFMA
// number k2 contracts the 4 bins {bin(w, g, k2) : g = 0..3} against the wave's weights for them.
// Filters 0..15 ("block 0") only touch bins < 64 -> k2 in {0, 1, 14, 15}; filters 16..31 only bins
// >= 48 -> k2 in 2..14 (both checked by build_tables): 17 MFMAs per wave, the same code in every wave.
// Bins == 16 (mod 32) come from column 16 and are fed by role 1.
// The list of (k2, block) MFMAs of a wave.  BANDED is the 16 kHz structure above; DENSE issues all 32 pairs and
// so serves any sample rate (other mel band edges) at the price of 15 more MFMAs per wave.
template <bool DENSE>
struct Sched;
template <>
struct Sched<false> {
    static constexpr int N = kAmelBanded;
    static constexpr int k2[N] = {0, 1, 14, 15, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14};
    static constexpr int blk[N] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
};
template <>
struct Sched<true> {
    static constexpr int N = kAmelDense;
    static constexpr int k2[N] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                  0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    static constexpr int blk[N] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
};

// 32 filters (every reference target) or 16 (the core's constructor default, mfcc.py:20): a 16-filter bank is
// block 0 only -- it runs on the dense schedule with block 1's weights zero and its log-mel lanes masked
inline bool supported(int nfft, int hop, int n_mel, int n_cep) {
    return nfft == kNfft && hop == kHop && (n_mel == kMel || n_mel == 16) && n_cep >= 1 && n_cep <= n_mel;
}

// ---- host: constant tables in the exact order the kernel consumes them
// true when some filter of the bank has weight on the (real-valued) DC bin
inline bool needs_dc_exact(int sample_rate, int n_mel) {
    std::vector<double> m0 = mfcc_tables::mel_dense(kNfft, n_mel, double(sample_rate));
    for (int f = 0; f < n_mel; ++f)
        if (m0[size_t(f) * 257] != 0.0) return true;
    return false;
}

template <bool DENSE>
inline bool build_tables(int sample_rate, double power_scale, double lifter, int n_cep, int n_mel,
                         std::vector<char> &blob) {
    constexpr int kAmel = Sched<DENSE>::N;
    if (n_mel != kMel && !DENSE) return false;
    using namespace mfcc_tables;
    std::vector<float> win(16 * 32), tw(16 * 16 * 2), amel(size_t(kWaves) * kAmel * 64, 0.0f),
        aext(size_t(kWaves) * kAextra * 64, 0.0f);
    std::vector<double> w = hamming_periodic(kNfft);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int n1 = 0; n1 < 32; ++n1) win[n2 * 32 + n1] = float(w[16 * n1 + n2] / 64.0);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int k1 = 0; k1 < 16; ++k1) {
            double a = -2.0 * kPi * double(n2 * k1) / 512.0;
            tw[(n2 * 16 + k1) * 2 + 0] = float(std::cos(a));
            tw[(n2 * 16 + k1) * 2 + 1] = float(std::sin(a));
        }
    std::vector<double> md(size_t(kMel) * 257, 0.0);                          // [32][257], rows >= n_mel stay 0
    {
        std::vector<double> m0 = mel_dense(kNfft, n_mel, double(sample_rate));
        std::copy(m0.begin(), m0.end(), md.begin());
    }
    const double inv = 1.0 / (power_scale * power_scale);
    std::vector<char> covered(size_t(kMel) * 257, 0);
    auto M = [&](int wave, int idx, int lane) -> float & { return amel[(size_t(wave) * kAmel + idx) * 64 + lane]; };
    auto E = [&](int role, int idx, int lane) -> float & { return aext[(size_t(role) * kAextra + idx) * 64 + lane]; };
    // mel operands: lane l of wave wv holds W[blk*16 + (l&15)][bin(wv, l>>4, k2)]
    auto mel_op = [&](int wv, int idx, int blk, int k2) {
        for (int l = 0; l < 64; ++l) {
            int filt = blk * 16 + (l & 15), k1 = 4 * wv + (l >> 4);
            if (k1 == 0 && k2 > 8) continue;           // bins 32 (16 - k2): already supplied by k2' = 16 - k2
            int bin = k2 < 8 ? k1 + 32 * k2 : 32 * (16 - k2) - k1;
            M(wv, idx, l) = float(md[size_t(filt) * 257 + bin] * inv);
            covered[size_t(filt) * 257 + bin] = 1;
        }
    };
    for (int wv = 0; wv < kWaves; ++wv) {
        for (int idx = 0; idx < kAmel; ++idx) mel_op(wv, idx, Sched<DENSE>::blk[idx], Sched<DENSE>::k2[idx]);
    }
    // role 0 -- DCT rows: lane (coeff = l&15, g = l>>4) holds D[16 half + coeff][16 blk + 4 g + r]
    std::vector<double> dd = dct_rows(n_cep, n_mel, lifter);                   // [n_cep][n_mel]
    for (int half = 0; half < 2; ++half)
        for (int blk = 0; blk < 2; ++blk)
            for (int r = 0; r < 4; ++r)
                for (int l = 0; l < 64; ++l) {
                    int coeff = 16 * half + (l & 15), filt = 16 * blk + 4 * (l >> 4) + r;
                    E(0, 8 * half + 4 * blk + r, l) =
                        (coeff < n_cep && filt < n_mel) ? float(dd[size_t(coeff) * n_mel + filt]) : 0.0f;
                }
    // role 1 -- column 16: X[16 + 32 k2] = sum_n2 v[n2] W512^(n2 (16 + 32 k2)); MFMA row i = 4g + r:
    // r=0: Re k2=2g, r=1: Im k2=2g, r=2: Re k2=2g+1, r=3: Im k2=2g+1
    for (int t = 0; t < 4; ++t)
        for (int l = 0; l < 64; ++l) {
            int i = l & 15, n2 = 4 * t + (l >> 4);
            int g = i >> 2, r = i & 3, k2 = 2 * g + (r >> 1);
            double th = 2.0 * kPi * double(n2 * (16 + 32 * k2)) / 512.0;
            E(1, t, l) = float((r & 1) ? -std::sin(th) : std::cos(th));
        }
    // role 1 -- those bins as a K step: lane g supplies bin 16 + 64 g (step 0) / 48 + 64 g (step 1)
    for (int blk = 0; blk < 2; ++blk)
        for (int step = 0; step < 2; ++step)
            for (int l = 0; l < 64; ++l) {
                int filt = blk * 16 + (l & 15), bin = 16 + 64 * (l >> 4) + 32 * step;
                E(1, 4 + 2 * blk + step, l) = float(md[size_t(filt) * 257 + bin] * inv);
                covered[size_t(filt) * 257 + bin] = 1;
            }
    for (int f = 0; f < kMel; ++f)
        for (int k = 0; k < 257; ++k)
            if (md[size_t(f) * 257 + k] != 0.0 && !covered[size_t(f) * 257 + k]) return false;
    auto put = [&](const std::vector<float> &v) {
        size_t off = blob.size();
        blob.resize(off + v.size() * 4);
        std::memcpy(blob.data() + off, v.data(), v.size() * 4);
    };
    // bf16 split of the same weights for the K = 32 MFMAs: lane l of wave wv, set s, holds rows m = l & 15 of filter
    // block blk[s] at K slots j = 0..7 <-> bin(wv, l >> 4, kGrpK2[grp[s]][j]); dword d = slots (2 d, 2 d + 1)
    constexpr int kSets = SetsBf<DENSE>::N;
    std::vector<float> abf(size_t(kWaves) * kSets * 2 * 4 * 64, 0.0f);      // uint32 payload, moved as floats
    std::vector<float> abf_nodc(abf.size(), 0.0f);                           // the same with bin 0's weights taken out
    bool covered_ok = true;
    auto make_abf = [&](std::vector<float> &dst, bool zero_dc) {
        auto bf16_round = [](float v) -> uint32_t {                          // round to nearest even, like v_cvt_pk_bf16_f32
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
        std::vector<char> cov2(size_t(kMel) * 257, 0);
        for (int wv = 0; wv < kWaves; ++wv)
            for (int st = 0; st < kSets; ++st)
                for (int l = 0; l < 64; ++l) {
                    uint32_t hi[8], lo[8];
                    for (int j = 0; j < 8; ++j) {
                        const int k2 = kGrpK2[SetsBf<DENSE>::grp[st]][j];
                        const int filt = SetsBf<DENSE>::blk[st] * 16 + (l & 15), k1 = 4 * wv + (l >> 4);
                        float wgt = 0.0f;
                        if (!(k1 == 0 && k2 > 8)) {
                            const int bin = k2 < 8 ? k1 + 32 * k2 : 32 * (16 - k2) - k1;
                            wgt = (zero_dc && bin == 0) ? 0.0f : float(md[size_t(filt) * 257 + bin] * inv);
                            cov2[size_t(filt) * 257 + bin] = 1;
                        }
                        hi[j] = bf16_round(wgt);
                        lo[j] = bf16_round(wgt - bf16_val(hi[j]));
                    }
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t vh = hi[2 * d] | (hi[2 * d + 1] << 16), vl = lo[2 * d] | (lo[2 * d + 1] << 16);
                        std::memcpy(&dst[((size_t(wv) * kSets + st) * 2 + 0) * 256 + d * 64 + l], &vh, 4);
                        std::memcpy(&dst[((size_t(wv) * kSets + st) * 2 + 1) * 256 + d * 64 + l], &vl, 4);
                    }
                }
        // the sets must cover what the fp32 schedule covers (bins 16 mod 32 come from column 16 either way)
        for (int f = 0; f < kMel; ++f)
            for (int k = 0; k < 257; ++k)
                if (md[size_t(f) * 257 + k] != 0.0 && !cov2[size_t(f) * 257 + k] && (k & 31) != 16) covered_ok = false;
    };
    make_abf(abf, false);
    make_abf(abf_nodc, true);
    if (!covered_ok) return false;
    blob.clear();
    put(win); put(tw); put(amel); put(aext); put(abf);
    // double-precision window rows for the DC bin (8-byte aligned: everything before is a multiple of 8 bytes)
    std::vector<double> wd(16 * 32);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int n1 = 0; n1 < 32; ++n1) wd[n2 * 32 + n1] = w[16 * n1 + n2] / 32.0;
    size_t off = blob.size();
    blob.resize(off + wd.size() * 8);
    std::memcpy(blob.data() + off, wd.data(), wd.size() * 8);
    // ... and for the twelve-wave form of the DC path: the window in sample order (n = 0 .. 256, padded to 264), the
    // weights without bin 0, bin 0's weight per filter
    std::vector<double> wl(264, 0.0);
    for (int n = 0; n <= 256; ++n) wl[n] = w[n] / 32.0;
    off = blob.size();
    blob.resize(off + wl.size() * 8);
    std::memcpy(blob.data() + off, wl.data(), wl.size() * 8);
    put(abf_nodc);
    std::vector<float> wdc(32, 0.0f);
    for (int f = 0; f < kMel; ++f) wdc[f] = float(md[size_t(f) * 257] * inv);
    put(wdc);
    // bin 0 on the raw samples: e[n] = 32 x[n] - 31 x[n-1], X[0] = sum_n (w[n] / 32) e[n] = sum_{m=-1}^{511} c[m] x[m]
    {
        double c[513], maxc = 0.0;
        for (int m = -1; m <= 511; ++m) {
            c[m + 1] = (m >= 0 ? w[m] : 0.0) - (m <= 510 ? 31.0 / 32.0 * w[m + 1] : 0.0);
            if (std::fabs(c[m + 1]) > maxc) maxc = std::fabs(c[m + 1]);
        }
        // seven digits in [-64, 63]: |C| <= 64 (128^7 - 1) / 127 = 2.8e14 -- 48 bits, what the double sum resolved
        int B = 0;
        while (std::ldexp(maxc, B + 1) < 2.7e14) ++B;
        long long sumC = 0;
        std::vector<uint32_t> adc(size_t(17) * 64 * 4, 0u);
        for (int tau = 0; tau < 513; ++tau) {
            long long C = std::llround(std::ldexp(c[tau], B));
            sumC += C;
            for (int j = 0; j < kDcDigits; ++j) {
                long long r = ((C + 64) % 128 + 128) % 128 - 64;
                C = (C - r) / 128;
                // the lo byte of sample tau is byte k = 2 tau of the frame's span, its hi byte k = 2 tau + 1;
                // byte k sits in k-block k / 64, lane group (k % 64) / 16, byte (k % 16) of the lane's four dwords;
                // MFMA row j takes the lo bytes with digit j, row kDcDigits + j the hi bytes
                for (int hi = 0; hi < 2; ++hi) {
                    const int k = 2 * tau + hi, row = hi ? kDcDigits + j : j;
                    const int lane = row + 16 * ((k % 64) / 16);
                    const size_t dw = (size_t(k / 64) * 64 + lane) * 4 + (k % 16) / 4;
                    adc[dw] |= uint32_t(uint8_t(int8_t(r))) << (8 * (k % 4));
                }
            }
            if (C != 0) return false;
        }
        {
            size_t o1 = blob.size();
            blob.resize(o1 + adc.size() * 4);
            std::memcpy(blob.data() + o1, adc.data(), adc.size() * 4);
        }
        // 128 sum(C) (the lo bytes are taken as lo - 128), split like the kernel's two partial sums: hi 2^28 + lo
        const long long bias = 128 * sumC, bias_lo = ((bias % (1ll << 28)) + (1ll << 28)) % (1ll << 28);
        const double consts[3] = {std::ldexp(1.0, -B), double(bias_lo), double((bias - bias_lo) >> 28)};
        size_t o2 = blob.size();
        blob.resize(o2 + sizeof(consts));
        std::memcpy(blob.data() + o2, consts, sizeof(consts));
    }
    {
        auto bf16_round = [](float v) -> uint32_t {                          // round to nearest even, like v_cvt_pk_bf16_f32
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
        std::vector<float> adct(size_t(2) * 2 * 256, 0.0f);                  // uint32 payload, moved as floats
        for (int tile = 0; tile < 2; ++tile)
            for (int l = 0; l < 64; ++l) {
                uint32_t hi[8], lo[8];
                for (int j = 0; j < 8; ++j) {
                    const int coeff = 16 * tile + (l & 15), q = l >> 4, filt = j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4);
                    const float w = (coeff < n_cep && filt < n_mel) ? float(dd[size_t(coeff) * n_mel + filt]) : 0.0f;
                    hi[j] = bf16_round(w);
                    lo[j] = bf16_round(w - bf16_val(hi[j]));
                }
                for (int d = 0; d < 4; ++d) {
                    const uint32_t vh = hi[2 * d] | (hi[2 * d + 1] << 16), vl = lo[2 * d] | (lo[2 * d + 1] << 16);
                    std::memcpy(&adct[(size_t(tile) * 2 + 0) * 256 + d * 64 + l], &vh, 4);
                    std::memcpy(&adct[(size_t(tile) * 2 + 1) * 256 + d * 64 + l], &vl, 4);
                }
            }
        put(adct);
    }
    return true;
}

inline void bind_tables(const char *b, int n_cep, int n_mel, bool dense, bool dc_exact, FusedTables &t) {
    t.n_mel = n_mel;
    const int kAmel = dense ? kAmelDense : kAmelBanded;
    // device pointer arithmetic only; layout = build_tables' put() order
    t.n_cep = n_cep;
    const float *f = reinterpret_cast<const float *>(b);
    t.win = f;                  f += 16 * 32;
    t.tw = reinterpret_cast<const float2 *>(f); f += 16 * 16 * 2;
    t.a_mel = f;                f += kWaves * kAmel * 64;
    t.a_extra = f;              f += kWaves * kAextra * 64;
    t.a_mel_bf = reinterpret_cast<const uint32_t *>(f);
    const int n_abf = kWaves * (dense ? SetsBf<true>::N : SetsBf<false>::N) * 2 * 4 * 64;
    f += n_abf;
    t.win_dc = dc_exact ? reinterpret_cast<const double *>(f) : nullptr;
    f += 2 * 16 * 32;
    t.win_dc_lin = reinterpret_cast<const double *>(f);
    f += 2 * 264;
    t.a_mel_bf_nodc = reinterpret_cast<const uint32_t *>(f);
    f += n_abf;
    t.w_dc = f;
    f += 32;
    t.a_dc_i8 = reinterpret_cast<const uint32_t *>(f);
    f += 17 * 64 * 4;
    t.dc_consts = reinterpret_cast<const double *>(f);
    f += 6;
    t.a_dct_bf = reinterpret_cast<const uint32_t *>(f);
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
__device__ unsigned long long g_stamps[4 * 12 + 16];   // [wave][12 phases], then [48 + wave]: tile loops
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

// The tile's sample window: slot j stands for sample i = tile_first - shift + j of the channel,
// j = 0..3071, where shift = 0..7 makes the 16-byte global loads aligned.  Fetcher u (0..191: the
// lanes of roles 1..3; role 0 spends that time on the previous tile's tail) fetches the pieces
// [8 u, 8 u + 8) and [1536 + 8 u, ...), plus the dword holding the sample in front of each piece.
// What is parked in LDS is the pre-emphasised sample e[i] = 32 x[i] - 31 x[i-1] as fp32 (exact:
// |e| < 2^21) -- computed once per sample here instead of once per (frame, sample) in pass 1, where
// three overlapping frames would each redo it.  Windows that stick out of the channel (stream start
// without history, zero-padded tail) are filled sample by sample with the stream's edge rules.
struct Fetch {
    i32x4 v0, v1;
    int p0, p1;          // dword in front of v0 / v1: its high half is the piece's predecessor sample
};

__device__ __forceinline__ void fetch_window(const mfcc_k::StreamDesc &s, const Window &w, int u, Fetch &f) {
    if (w.inside) {
        const i32x4 *g = reinterpret_cast<const i32x4 *>(w.ptr - w.shift);
        const int *g32 = reinterpret_cast<const int *>(g);
        f.v0 = g[u];
        f.p0 = g32[4 * u - 1];
        f.v1 = g[kFetchers + u];
        f.p1 = g32[4 * (kFetchers + u) - 1];
    } else {
        const long long first = (long long)w.t_in * kTileHop;      // channel-relative
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
    preemph8(f.p1, f.v1, Sf + 8 * (kFetchers + u));
}

// The summed mel energies of a finished tile, then log2 (MFCC.ipynb cell 36): register r of block b
// is filter 16 b + 4 q + r of frame lo.
__device__ __forceinline__ void mel_log2(const float *Qt, int lane, int n_mel, f32x4 &l0, f32x4 &l1) {
    const f32x4 *Q4 = reinterpret_cast<const f32x4 *>(Qt) + lane;
    const f32x4 m0 = (Q4[0 * 64] + Q4[2 * 64]) + (Q4[4 * 64] + Q4[6 * 64]);
    const f32x4 m1 = (Q4[1 * 64] + Q4[3 * 64]) + (Q4[5 * 64] + Q4[7 * 64]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // v_log_f32 (1 ulp; a denormal mel energy -- far below anything int16 PCM produces --
        // counts as 0, like an exact zero: -inf)
        l0[r] = __builtin_amdgcn_logf(m0[r]);
        l1[r] = __builtin_amdgcn_logf(m1[r]);
    }
    if (n_mel <= 16) l1 = (f32x4){0.f, 0.f, 0.f, 0.f};          // no filters 16..31 (uniform)
}

// DCT-II (cells 38-39) and store: log-mel register r of block b == B[k = q][j = lo] of the DCT
// product, so the mel accumulator layout feeds the DCT MFMAs without any lane movement.
__device__ __forceinline__ void dct_store(const mfcc_k::StreamDesc &s, const FusedTables &t, const f32x4 &l0,
                                          const f32x4 &l1, const f32x4 &d0, const f32x4 &d1,
                                          const float (&ax)[kAextra], const Cursor &c, int lo,
                                          int q, int lane_off, float *__restrict__ out) {
    // d0 + d1 = coefficients 0..15 (their MFMAs are issued by the caller, interleaved with the mel MFMAs)
    // uniform part of the address on the scalar unit; lane_off = lo * n_cep + 4 q
    const long long fr0 = (long long)c.t_in * kTile;
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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFCC_MFMA_BF(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// (a, b) -> their bf16 roundings packed in one dword (a low) and the bf16 roundings of what the first rounding lost
__device__ __forceinline__ void split_bf16_pair(float a, float b, uint32_t &hi, uint32_t &lo) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
}

// |X|^2 of a lane's 16 bins -> the B operands of the two K groups, high and low terms
struct PowerBf {
    u32x4 hi[2], lo[2];
};
__device__ __forceinline__ void split_power(const float (&pw)[16], PowerBf &p) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t h, l;
            split_bf16_pair(pw[kGrpK2[g][2 * d]], pw[kGrpK2[g][2 * d + 1]], h, l);
            p.hi[g][d] = h;
            p.lo[g][d] = l;
        }
}

// term T (0: Wh Ph, 1: Wh Pl, 2: Wl Ph) of set S; one accumulator per set (chains of three, interleaved by the caller)
template <bool DENSE, int S, int T>
__device__ __forceinline__ void mel_bf_term(const u32x4 (&ah)[SetsBf<DENSE>::N], const u32x4 (&al)[SetsBf<DENSE>::N],
                                            const PowerBf &p, f32x4 (&acc)[SetsBf<DENSE>::N]) {
    constexpr int g = SetsBf<DENSE>::grp[S];
    if constexpr (T == 0) acc[S] = MFCC_MFMA_BF(ah[S], p.hi[g], acc[S]);
    if constexpr (T == 1) acc[S] = MFCC_MFMA_BF(ah[S], p.lo[g], acc[S]);
    if constexpr (T == 2) acc[S] = MFCC_MFMA_BF(al[S], p.hi[g], acc[S]);
}

// all terms of all sets, term-major (consecutive MFMAs never share an accumulator); AFTER(i) runs after MFMA number i
template <bool DENSE, int I, typename After>
__device__ __forceinline__ void mel_bf_all(const u32x4 (&ah)[SetsBf<DENSE>::N], const u32x4 (&al)[SetsBf<DENSE>::N],
                                           const PowerBf &p, f32x4 (&acc)[SetsBf<DENSE>::N], After &&after) {
    constexpr int N = SetsBf<DENSE>::N;
    if constexpr (I < 3 * N) {
        mel_bf_term<DENSE, I % N, I / N>(ah, al, p, acc);
        after(std::integral_constant<int, I>{});
        mel_bf_all<DENSE, I + 1>(ah, al, p, acc, after);
    }
}

// sums of the set accumulators per filter block
template <bool DENSE>
__device__ __forceinline__ void mel_bf_blocks(const f32x4 (&acc)[SetsBf<DENSE>::N], f32x4 &b0, f32x4 &b1) {
    if constexpr (DENSE) {
        b0 = acc[0] + acc[1];
        b1 = acc[2] + acc[3];
    } else {
        b0 = acc[0];
        b1 = acc[1] + acc[2];
    }
}

// one mel MFMA of the schedule: block 0 accumulates in (x0, y0), block 1 in (x1, y1), alternating
template <bool DENSE, int I>
__device__ __forceinline__ void mel_step(const float (&am)[Sched<DENSE>::N], const float (&pw)[16], f32x4 &x0, f32x4 &y0,
                                         f32x4 &x1, f32x4 &y1) {
    constexpr int k2 = Sched<DENSE>::k2[I], blk = Sched<DENSE>::blk[I];
    f32x4 &acc = blk ? ((I & 1) ? y1 : x1) : ((I & 1) ? y0 : x0);
    acc = MFCC_MFMA(am[I], pw[k2], acc);
}

template <bool DENSE, int LO, int HI>
__device__ __forceinline__ void mel_steps(const float (&am)[Sched<DENSE>::N], const float (&pw)[16], f32x4 &x0, f32x4 &y0,
                                          f32x4 &x1, f32x4 &y1) {
    if constexpr (LO < HI) {
        mel_step<DENSE, LO>(am, pw, x0, y0, x1, y1);
        mel_steps<DENSE, LO + 1, HI>(am, pw, x0, y0, x1, y1);
    }
}

// role 0: mel MFMAs with one DCT MFMA (coefficients 0..15 of the previous tile) after every second one
template <bool DENSE, int I>
__device__ __forceinline__ void mel_dct_steps(const float (&am)[Sched<DENSE>::N], const float (&pw)[16],
                                              const float (&ax)[kAextra], const f32x4 &lm0, const f32x4 &lm1, f32x4 &x0,
                                              f32x4 &y0, f32x4 &x1, f32x4 &y1, f32x4 &d0, f32x4 &d1) {
    if constexpr (I < Sched<DENSE>::N) {
        mel_step<DENSE, I>(am, pw, x0, y0, x1, y1);
        if constexpr ((I & 1) && I / 2 < 8) {
            constexpr int j = I / 2, r = j >> 1;
            if constexpr (j & 1) d1 = MFCC_MFMA(ax[4 + r], lm1[r], d1);
            else d0 = MFCC_MFMA(ax[r], lm0[r], d0);
        }
        mel_dct_steps<DENSE, I + 1>(am, pw, ax, lm0, lm1, x0, y0, x1, y1, d0, d1);
    }
}

template <bool DENSE, bool DCX>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_fused512_kernel(mfcc_k::StreamDesc s, FusedTables t, LaunchGeom g, float *__restrict__ out) {
    [[maybe_unused]] constexpr int kAmel = Sched<DENSE>::N;
    __shared__ __attribute__((aligned(16))) float lds[kLdsWords + (DCX ? kDcxWords : 0)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave;             // which extra job the wave has in the MFMA window (see header)
    const int lo = lane & 15;          // n2 in pass 1, k1 in pass 2, frame column in the MFMA phase
    const int q = lane >> 4;           // quarter of the wave; K index g in the MFMA phase
    // frame of the tile this quarter transforms.  The two quarters of a 32-lane half are 8 frames =
    // 1360 samples = 16 (mod 32) LDS banks apart, so their ds_read_b32 of the window never collide.
    const int fr_id = wave + 8 * (q & 1) + 4 * (q >> 1);

    float *const Tt = lds;                                         // [16 frames][548]: [16 n2][34] each
    float *const Vt = Tt + kTile * kTFrame;                        // [16 frames][18]
    float *const Qt = Vt + kTile * kVStride;                       // [4 waves][2 blocks][256]
    float *const Sf = Qt + kQWords;                                // pre-emphasised sample window, fp32
    double *const Wd = reinterpret_cast<double *>(Sf + kSUsed);    // DCX: double window rows, [16][kWdRow]
    double *const Dc = Wd + 16 * kWdRow;                           // DCX: partial DC sums, [16 frames][kDcRow]
    if constexpr (DCX) {
        for (int i = tid; i < 16 * 32; i += 64 * kWaves) Wd[(i >> 5) * kWdRow + (i & 31)] = t.win_dc[i];
    }

    // per-lane constants, resident for the whole kernel
    using mfcc_codelets::v2f;
    v2f wp[16];                                    // window pairs (w[2m], w[2m+1]) of this lane's samples
#pragma unroll
    for (int i = 0; i < 16; ++i) wp[i] = reinterpret_cast<const v2f *>(t.win)[lo * 16 + i];
    v2f tw[16];                                    // W512^(n2 k1) as (cos, sin)
#pragma unroll
    for (int i = 0; i < 16; ++i) tw[i] = reinterpret_cast<const v2f *>(t.tw)[lo * 16 + i];
    float ax[kAextra];
#if MFCC_MEL_BF16
    constexpr int kSets = SetsBf<DENSE>::N;
    u32x4 ah[kSets], al[kSets];
#pragma unroll
    for (int st = 0; st < kSets; ++st)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            ah[st][d] = t.a_mel_bf[((wave * kSets + st) * 2 + 0) * 256 + d * 64 + lane];
            al[st][d] = t.a_mel_bf[((wave * kSets + st) * 2 + 1) * 256 + d * 64 + lane];
        }
#else
    float am[kAmel];
#pragma unroll
    for (int i = 0; i < kAmel; ++i) am[i] = t.a_mel[(wave * kAmel + i) * 64 + lane];
#endif
#pragma unroll
    for (int i = 0; i < kAextra; ++i) ax[i] = t.a_extra[(role * kAextra + i) * 64 + lane];

    // slot of this lane's sample n1 = 0 in the window, before the per-tile alignment shift
    const int lane_slot = fr_id * kHop + lo;
    const int fetcher = (role - 1) * 64 + lane;     // 0..191 in roles 1..3
    const bool fetches = role != 0;
    const int lane_off = lo * t.n_cep + 4 * q;

    Cursor cur;
    cur.ch = (int)(blockIdx.x / (unsigned)g.tiles_per_ch);
    cur.t_in = (int)(blockIdx.x - (unsigned)cur.ch * (unsigned)g.tiles_per_ch);
    cur.ptr = s.pcm + (long long)cur.ch * s.ch_stride + (long long)cur.t_in * kTileHop;

    // first tile: fetch and park the sample window
    Fetch fx;
    int shift = 0;
    if (cur.ch < g.n_ch) {
        const Window w0 = window_of(cur, g);
        shift = w0.shift;
        if (fetches) {
            fetch_window(s, w0, fetcher, fx);
            park_window(Sf, fetcher, fx);
        }
    }
    __syncthreads();

    // the role-0 wave finishes tile t (log2, DCT, store) during tile t + 1, so that the other waves
    // never wait for it: it picks the summed mel energies out of Q right after B2 (Q is rewritten only
    // after the next B1) and carries their log2 and the tile's coordinates to its MFMA window
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 lm0 = zero, lm1 = zero;
    Cursor prev = cur;
    bool have_prev = false;

#ifdef MFCC_FUSED_STAMPS
    unsigned long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
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
        if constexpr (DCX) {
            // bin 0 of this lane's 32 samples in double: sum_n1 w[16 n1 + n2] e[16 n1 + n2] (the products of a
            // 24-bit window value and a 22-bit integer are exact in double; so is their sum to 2^-53)
            const double *wr = Wd + lo * kWdRow;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int m = 0; m < 16; m += 2) {
                a0 = __builtin_fma(wr[2 * m + 0], (double)ep[m][0], a0);
                a1 = __builtin_fma(wr[2 * m + 1], (double)ep[m][1], a1);
                a2 = __builtin_fma(wr[2 * m + 2], (double)ep[m + 1][0], a2);
                a3 = __builtin_fma(wr[2 * m + 3], (double)ep[m + 1][1], a3);
            }
            Dc[fr_id * kDcRow + lo] = (a0 + a1) + (a2 + a3);
        }
        // next tile's samples fly while this tile is processed; role 0 picks up the previous tile's
        // mel sums instead (behind the sample reads in the LDS queue)
        const Cursor me = cur;
        advance(cur, g);
        const bool more = cur.ch < g.n_ch;
        int next_shift = 0;
        if (more) {
            const Window wn = window_of(cur, g);
            next_shift = wn.shift;
            if (fetches) fetch_window(s, wn, fetcher, fx);
        }
        if (role == 0 && have_prev) mel_log2(Qt, lane, t.n_mel, lm0, lm1);
        MFCC_STAMP(6);

        // windowed real FFT-32 over n1, twiddled by W512^(n2 k1): columns 0..15 as (re, im) pairs, column 16
        mfcc_codelets::v2f ty[16];
        float y16;
        mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
        MFCC_STAMP(7);

        // transpose through LDS: T[frame][n2][k1]
        mfcc_codelets::v2f *tcol0 = reinterpret_cast<mfcc_codelets::v2f *>(Tt + fr_id * kTFrame) + lo;
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
        Vt[fr_id * kVStride + lo] = y16;
        MFCC_STAMP(0);
        lds_barrier();                         // B1: T and V of all 16 frames are in LDS; S and Q are consumed
        MFCC_STAMP(3);

        // ---------------- pass 2: complex FFT-16 over n2 for frame lo, column k1 = 4 wave + q
        float pw[16];                            // |X|^2 at bin(wave, q, k2)
        {
            mfcc_codelets::v2f x[16];
            const f32x4 *trow = reinterpret_cast<const f32x4 *>(Tt + lo * kTFrame + (4 * wave + q) * kTRow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 a = trow[i];
                x[2 * i] = (mfcc_codelets::v2f){a[0], a[1]};
                x[2 * i + 1] = (mfcc_codelets::v2f){a[2], a[3]};
            }
            MFCC_STAMP(1);
            mfcc_codelets::v2f pp[8];                // (|z[k2]|^2, |z[k2 + 8]|^2): the codelet's last layer is transposed
            mfcc_codelets::cfft16_pow(x, pp);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) pw[k2] = pp[k2].x, pw[k2 + 8] = pp[k2].y;
        }
        if constexpr (DCX) {
            if (wave == 0) {                     // lanes (frame lo, k1 = 0) hold bin 0 in pw[0]
                const double *dr = Dc + lo * kDcRow;
                double x0 = 0.0, x1 = 0.0;
#pragma unroll
                for (int n2 = 0; n2 < 16; n2 += 2) {
                    x0 += dr[n2];
                    x1 += dr[n2 + 1];
                }
                x0 += x1;
                if (q == 0) pw[0] = (float)(x0 * x0);
            }
        }
        MFCC_STAMP(2);

        // ---------------- MFMA window (frame column = lo, K index = q)
        f32x4 x0 = zero, y0 = zero, x1 = zero, y1 = zero;
#if MFCC_MEL_BF16
        PowerBf pb;
        split_power(pw, pb);
        f32x4 acc[kSets];
#pragma unroll
        for (int st = 0; st < kSets; ++st) acc[st] = zero;
        if (role == 0) {
            // this tile's mel MFMAs with the previous tile's DCT MFMAs (coefficients 0..15, fp32) in between
            f32x4 d0 = zero, d1 = zero;
            mel_bf_all<DENSE, 0>(ah, al, pb, acc, [&](auto i) {
                constexpr int I = decltype(i)::value;
                if constexpr (I < 8) {
                    constexpr int r = I >> 1;
                    if constexpr (I & 1) d1 = MFCC_MFMA(ax[4 + r], lm1[r], d1);
                    else d0 = MFCC_MFMA(ax[r], lm0[r], d0);
                }
            });
            mel_bf_blocks<DENSE>(acc, x0, x1);
            MFCC_STAMP(8);
            if (have_prev) dct_store(s, t, lm0, lm1, d0, d1, ax, prev, lo, q, lane_off, out);
        } else if (role == 1) {
            // column 16 -> bins 16 + 32 j of this tile (a 16 x 16 real DFT matrix on fp32 MFMAs), fed to both filter
            // blocks from registers at the end
            const float v0 = Vt[lo * kVStride + 0 + q], v1 = Vt[lo * kVStride + 4 + q];
            const float v2 = Vt[lo * kVStride + 8 + q], v3 = Vt[lo * kVStride + 12 + q];
            f32x4 sp = zero, sp2 = zero;
            mel_bf_all<DENSE, 0>(ah, al, pb, acc, [&](auto i) {
                constexpr int I = decltype(i)::value;
                if constexpr (I == 0) sp = MFCC_MFMA(ax[0], v0, sp);
                if constexpr (I == 1) sp2 = MFCC_MFMA(ax[1], v1, sp2);
                if constexpr (I == 2) sp = MFCC_MFMA(ax[2], v2, sp);
                if constexpr (I == 3) sp2 = MFCC_MFMA(ax[3], v3, sp2);
            });
            mel_bf_blocks<DENSE>(acc, x0, x1);
            sp += sp2;
            const float s0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);      // bin 16 + 64 q
            const float s1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);      // bin 48 + 64 q
            x0 = MFCC_MFMA(ax[4], s0, x0);
            y0 = MFCC_MFMA(ax[5], s1, y0);
            x1 = MFCC_MFMA(ax[6], s0, x1);
            y1 = MFCC_MFMA(ax[7], s1, y1);
            MFCC_STAMP(8);
        } else {
            mel_bf_all<DENSE, 0>(ah, al, pb, acc, [](auto) {});
            mel_bf_blocks<DENSE>(acc, x0, x1);
            MFCC_STAMP(8);
        }
#else
        if (role == 0) {
            // this tile's mel MFMAs and the previous tile's DCT MFMAs (coefficients 0..15) in ONE basic block,
            // interleaved: six independent accumulator chains instead of two long tails (lm = 0 before the
            // first tile; only the store depends on have_prev)
            f32x4 d0 = zero, d1 = zero;
            mel_dct_steps<DENSE, 0>(am, pw, ax, lm0, lm1, x0, y0, x1, y1, d0, d1);
            MFCC_STAMP(8);
            if (have_prev) dct_store(s, t, lm0, lm1, d0, d1, ax, prev, lo, q, lane_off, out);
        } else if (role == 1) {
            // column 16 -> bins 16 + 32 j of this tile (a 16 x 16 real DFT matrix, two chains of two MFMAs),
            // fed to both filter blocks from registers at the end; its chain hides among the mel MFMAs
            const float v0 = Vt[lo * kVStride + 0 + q], v1 = Vt[lo * kVStride + 4 + q];
            const float v2 = Vt[lo * kVStride + 8 + q], v3 = Vt[lo * kVStride + 12 + q];
            f32x4 sp = MFCC_MFMA(ax[0], v0, zero);
            f32x4 sp2 = MFCC_MFMA(ax[1], v1, zero);
            mel_steps<DENSE, 0, 2>(am, pw, x0, y0, x1, y1);
            sp = MFCC_MFMA(ax[2], v2, sp);
            sp2 = MFCC_MFMA(ax[3], v3, sp2);
            mel_steps<DENSE, 2, kAmel>(am, pw, x0, y0, x1, y1);
            sp += sp2;
            const float s0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);      // bin 16 + 64 q
            const float s1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);      // bin 48 + 64 q
            x0 = MFCC_MFMA(ax[4], s0, x0);
            y0 = MFCC_MFMA(ax[5], s1, y0);
            x1 = MFCC_MFMA(ax[6], s0, x1);
            y1 = MFCC_MFMA(ax[7], s1, y1);
            MFCC_STAMP(8);
        } else {
            mel_steps<DENSE, 0, kAmel>(am, pw, x0, y0, x1, y1);
            MFCC_STAMP(8);
        }
#endif
        MFCC_STAMP(9);
        *reinterpret_cast<f32x4 *>(Qt + (2 * wave + 0) * 256 + lane * 4) = x0 + y0;
        *reinterpret_cast<f32x4 *>(Qt + (2 * wave + 1) * 256 + lane * 4) = x1 + y1;
        prev = me;
        have_prev = true;
        MFCC_STAMP(10);
        // park the next tile's sample window (every read of the current one happened before B1)
        if (more && fetches) park_window(Sf, fetcher, fx);
        shift = next_shift;
        MFCC_STAMP(4);
        lds_barrier();                         // B2: partial sums and S are in LDS, T/V may be overwritten
        MFCC_STAMP(5);
    }
    // the last tile of this workgroup
    if (role == 0 && have_prev) {
        mel_log2(Qt, lane, t.n_mel, lm0, lm1);
        f32x4 d0 = zero, d1 = zero;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d0 = MFCC_MFMA(ax[r], lm0[r], d0);
            d1 = MFCC_MFMA(ax[4 + r], lm1[r], d1);
        }
        dct_store(s, t, lm0, lm1, d0, d1, ax, prev, lo, q, lane_off, out);
    }
#ifdef MFCC_FUSED_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 12; ++i) atomicAdd(&g_stamps[wave * 12 + i], st_sum[i]);
        atomicAdd(&g_stamps[48 + wave], 1ull);
    }
#endif
}

inline const char *kernel_name() { return "mfcc_fused512_kernel"; }

// returns false when the problem does not fit the kernel's 32-bit tile arithmetic
inline bool launch(const mfcc_k::StreamDesc &s, const FusedTables &t, bool dense, float *out, int n_cu,
                   hipStream_t stream) {
    const bool dcx = t.win_dc != nullptr;        // only ever set together with the dense schedule
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
    g.step_ptr = (long long)g.grid_div * s.ch_stride + (long long)g.grid_mod * kTileHop;
    g.wrap_ptr = s.ch_stride - tiles_per_ch * (long long)kTileHop;
    // window of tile t_in: samples [t_in * kTileHop - mis - 2, t_in * kTileHop - mis + kSUsed), mis <= 7
    g.t_lo = (int)((9 - (long long)s.halo + kTileHop - 1) / kTileHop);
    if (g.t_lo < 0) g.t_lo = 0;
    const long long hi = (s.n_samples - kSUsed) / kTileHop;
    g.t_hi = s.n_samples < kSUsed ? -1 : (int)(hi < tiles_per_ch ? hi : tiles_per_ch);
    if (dense && dcx)
        hipLaunchKernelGGL((mfcc_fused512_kernel<true, true>), dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t, g, out);
    else if (dense)
        hipLaunchKernelGGL((mfcc_fused512_kernel<true, false>), dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t, g, out);
    else
        hipLaunchKernelGGL((mfcc_fused512_kernel<false, false>), dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t, g, out);
    return true;
}

}  // namespace mfcc_fused
