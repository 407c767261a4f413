// Fused 1024/341/40 float kernel for gfx950 (MI355X) -- BASELINE.json configs[3]: nfft 1024, hop
// 1024//3 = 341 (mfcc/core/mfcc.py:43), 40 mel bands, n_cep <= 32, the mel contraction on the matrix
// cores.  Same scheme as kernel_fused512.hpp (read its header first); what differs:
//
//  * a workgroup of EIGHT waves owns a tile of 16 consecutive frames, one workgroup per CU (118 KB of
//    LDS: a 16-frame tile of 1024-point frames is twice the data);
//  * pass 1: n = 32 n1 + n2.  Wave w, lane (f = lane >> 5, n2 = lane & 31) owns frame 2 w + f and runs
//    the same register-resident REAL 32-point FFT over n1 (codelet rfft32_tw, Hamming folded in), twiddles
//    columns 0..15 by W1024^(n2 k1) and writes T[frame][n2][k1]; column 16 (real) goes to V;
//  * pass 2: the complex 32-point FFT over n2 of a column is split by ONE decimation-in-frequency step
//    into its even and odd outputs, two lanes per column: wave w, lane (j = lane & 15, g = lane >> 4)
//    takes column k1 = 4 (w >> 1) + g of frame j and the outputs k2 = 2 m + h, h = w & 1 (codelets
//    cfft32_h0 / cfft32_h1: the DIF step, then a 16-point FFT).  X[k1 + 32 k2] is bin k1 + 32 k2 or, by
//    the symmetry of a real signal, bin 1024 - (k1 + 32 k2): 512 lanes x 16 outputs = every bin once;
//  * |X|^2 is again in the MFMA B-operand layout: the block-banded 40 x 513 mel matrix needs 17
//    (h = 0) / 18 (h = 1) MFMAs per wave over three 16-filter blocks; partial sums meet in Q;
//  * column 16 -> bins 16 + 32 j by a 32-point DFT matrix on the matrix cores, split over two waves
//    (role 1: j = 0..7, role 2: j = 8..15; 8 MFMAs + their mel MFMAs each); role 0 finishes the
//    previous tile (log2, DCT-II as 12 MFMAs, store) and neither fetches nor parks samples.
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

constexpr int kNfft = 1024, kHop = 341, kMel = 40, kMaxCep = 32;   // 32: what the reference tops keep (main.c:13)
constexpr int kTile = 16, kWaves = 8;
constexpr int kTileHop = kTile * kHop;            // 5456 samples between consecutive tiles
constexpr int kTRow = 34;                         // words per n2 row of the transpose tile (16 complex + 2)
constexpr int kTFrame = 32 * kTRow + 2;           // 1090 words per frame (== 2 mod 32, see kernel_fused512.hpp)
constexpr int kVStride = 34;                      // words per frame in the column-16 tile
constexpr int kBlocks = 3;                        // 16-filter blocks of the 40 filters
constexpr int kQWords = kWaves * kBlocks * 256;   // partial mel sums: [wave][block][lane * 4]
constexpr int kFetchers = 64 * (kWaves - 1);      // roles 1..7 fetch and park the sample window
constexpr int kPieces = (7 + (kTile - 1) * kHop + kNfft + 7) / 8;       // 769 pieces of 8 samples are read
constexpr int kSecond = kPieces - kFetchers;      // fetchers that take a second piece (321)
constexpr int kSUsed = 8 * kPieces;               // 6152 fp32 slots
constexpr int kLdsWords = kTile * kTFrame + kTile * kVStride + kQWords + kSUsed;
constexpr int kAmel = 18, kAextra = 12;

// (output index m of the 16-point FFT, filter block) pairs with non-zero weights, per h = k2 & 1
// (verified by build_tables against the actual matrix)
constexpr int kN0 = 17, kN1 = 18;
constexpr int kM0[kN0] = {0, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
constexpr int kB0[kN0] = {0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 0};
constexpr int kM1[kN1] = {0, 1, 2, 3, 4, 4, 5, 6, 7, 8, 9, 10, 11, 11, 12, 13, 14, 15};
constexpr int kB1[kN1] = {0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 1, 2, 1, 1, 1, 0};
// special column, role 1 (bins 16 + 32 j, j = 2 g + step, g = 0..3): (step, block); role 2: j = 8 + 2 g + step
constexpr int kNS1 = 4, kNS2 = 3;
constexpr int kS1step[kNS1] = {0, 0, 1, 1}, kS1blk[kNS1] = {0, 1, 0, 1};
constexpr int kS2step[kNS2] = {0, 0, 1}, kS2blk[kNS2] = {1, 2, 2};

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

struct Tables {
    const float *win;      // [32 n2][32 n1]  hamming[32 n1 + n2] / 64
    const float *tw;       // [32 n2][16 k1][2] W1024^(n2 k1)
    const float *a_mel;    // [8 waves][18][64]
    const float *a_extra;  // [3 roles][12][64]  role 0: DCT rows; role 1 / 2: column-16 DFT (8) + its mel weights;
                           // then [12][64]: DCT rows of coefficients 16..31 (fetched per tile, only when n_cep > 16: the
                           // kernel has no registers left to keep them)
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

inline bool build_tables(int sample_rate, double power_scale, double lifter, int n_cep, std::vector<char> &blob) {
    using namespace mfcc_tables;
    std::vector<float> win(32 * 32), tw(32 * 16 * 2), amel(size_t(kWaves) * kAmel * 64, 0.0f),
        aext(size_t(4) * kAextra * 64, 0.0f);
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
        if (md[size_t(f) * nb] != 0.0) return false;      // weight on the DC bin: the generic kernel sums it in double
    const double inv = 1.0 / (power_scale * power_scale);
    std::vector<char> covered(size_t(kMel) * nb, 0);
    auto Wt = [&](int filt, int bin) -> double { return filt < kMel ? md[size_t(filt) * nb + bin] * inv : 0.0; };
    for (int wv = 0; wv < kWaves; ++wv) {
        const int h = wv & 1, n = h ? kN1 : kN0;
        for (int idx = 0; idx < n; ++idx) {
            const int m = h ? kM1[idx] : kM0[idx], blk = h ? kB1[idx] : kB0[idx];
            for (int l = 0; l < 64; ++l) {
                const int filt = blk * 16 + (l & 15), k1 = 4 * (wv >> 1) + (l >> 4);
                const int bin = bin_of(k1, h, m);
                if (bin < 0 || filt >= kMel) continue;
                amel[(size_t(wv) * kAmel + idx) * 64 + l] = float(Wt(filt, bin));
                covered[size_t(filt) * nb + bin] = 1;
            }
        }
    }
    auto E = [&](int role, int idx, int lane) -> float & { return aext[(size_t(role) * kAextra + idx) * 64 + lane]; };
    // role 0 -- DCT rows: lane (coeff = l & 15, g = l >> 4) holds D[coeff][16 blk + 4 g + r]
    std::vector<double> dd = dct_rows(n_cep, kMel, lifter);                   // [n_cep][40]
    for (int blk = 0; blk < kBlocks; ++blk)
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                const int coeff = l & 15, filt = 16 * blk + 4 * (l >> 4) + r;
                E(0, 4 * blk + r, l) = (coeff < n_cep && filt < kMel) ? float(dd[size_t(coeff) * kMel + filt]) : 0.0f;
                E(3, 4 * blk + r, l) = (16 + coeff < n_cep && filt < kMel) ? float(dd[size_t(16 + coeff) * kMel + filt]) : 0.0f;
            }
    // roles 1, 2 -- column 16: X[16 + 32 j] = sum_n2 v[n2] W1024^(n2 (16 + 32 j)); MFMA row i = 4 g + r holds
    // r = 0: Re j = jb + 2 g, r = 1: Im (same j), r = 2: Re j + 1, r = 3: Im; K step t covers n2 = 4 t + (l >> 4)
    for (int role = 1; role <= 2; ++role) {
        const int jb = role == 1 ? 0 : 8;
        for (int t = 0; t < 8; ++t)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, n2 = 4 * t + (l >> 4);
                const int g = i >> 2, r = i & 3, j = jb + 2 * g + (r >> 1);
                const double th = 2.0 * kPi * double(n2 * (16 + 32 * j)) / 1024.0;
                E(role, t, l) = float((r & 1) ? -std::sin(th) : std::cos(th));
            }
        // its bins as K steps: lane g supplies bin 16 + 32 (jb + 2 g + step)
        const int ns = role == 1 ? kNS1 : kNS2;
        for (int idx = 0; idx < ns; ++idx) {
            const int step = role == 1 ? kS1step[idx] : kS2step[idx], blk = role == 1 ? kS1blk[idx] : kS2blk[idx];
            for (int l = 0; l < 64; ++l) {
                const int filt = blk * 16 + (l & 15), bin = 16 + 32 * (jb + 2 * (l >> 4) + step);
                if (filt >= kMel) continue;
                E(role, 8 + idx, l) = float(Wt(filt, bin));
                covered[size_t(filt) * nb + bin] = 1;
            }
        }
    }
    for (int f = 0; f < kMel; ++f)
        for (int k = 0; k < nb; ++k)
            if (md[size_t(f) * nb + k] != 0.0 && !covered[size_t(f) * nb + k]) return false;
    auto put = [&](const std::vector<float> &v) {
        size_t off = blob.size();
        blob.resize(off + v.size() * 4);
        std::memcpy(blob.data() + off, v.data(), v.size() * 4);
    };
    blob.clear();
    put(win); put(tw); put(amel); put(aext);
    return true;
}

inline void bind_tables(const char *b, int n_cep, Tables &t) {
    t.n_cep = n_cep;
    const float *f = reinterpret_cast<const float *>(b);
    t.win = f;      f += 32 * 32;
    t.tw = f;       f += 32 * 16 * 2;
    t.a_mel = f;    f += kWaves * kAmel * 64;
    t.a_extra = f;
}

// ---- device (helpers shared in spirit with kernel_fused512.hpp; kept local so the two kernels stay independent)

#define MFCC1K_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

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

// d[0] + d[1] + d[2] = the DCT of the previous tile (its 12 MFMAs are issued by the caller); coefficients 16..31 are a
// second M tile whose A operands are fetched here (uniform branch; 12 coalesced dwords per lane out of L1 / L2)
__device__ __forceinline__ void dct_store(const mfcc_k::StreamDesc &s, const Tables &t, const f32x4 (&d)[kBlocks],
                                          const f32x4 (&lm)[kBlocks], const Cursor &c, int lo, int q, int lane,
                                          int lane_off, float *__restrict__ out) {
    const f32x4 d0 = d[0], d1 = d[1], d2 = d[2];
    const long long fr0 = (long long)c.t_in * kTile;
    const long long rows_left = s.frames_per_ch - fr0;
    float *o = out + ((long long)c.ch * s.frames_per_ch + fr0) * t.n_cep + lane_off;
    if (lo < rows_left) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < t.n_cep) o[r] = (d0[r] + d1[r]) + d2[r];
    }
    if (t.n_cep > 16) {
        const float *hi = t.a_extra + (size_t)3 * kAextra * 64 + lane;
        asm volatile("" : "+v"(hi));                   // not hoisted out of the tile loop: no registers to hold it
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 e[kBlocks] = {zero, zero, zero};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) e[b] = MFCC1K_MFMA(hi[(4 * b + r) * 64], lm[b][r], e[b]);
        if (lo < rows_left) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 + 4 * q + r < t.n_cep) o[16 + r] = (e[0][r] + e[1][r]) + e[2][r];
        }
    }
}

// role 0 (wave 0, h = 0): this tile's 17 mel MFMAs and the previous tile's 12 DCT MFMAs in one basic block,
// interleaved -- six independent accumulator chains instead of two long tails
__device__ __forceinline__ void mel_dct_mfmas(const float (&pw)[16], const float (&am)[kAmel], const float (&ax)[kAextra],
                                              const f32x4 (&lm)[kBlocks], f32x4 (&acc)[kBlocks], f32x4 (&d)[kBlocks]) {
#pragma unroll
    for (int i = 0; i < kN0; ++i) {
        acc[kB0[i]] = MFCC1K_MFMA(am[i], pw[kM0[i]], acc[kB0[i]]);
        if (i < 12) d[i % 3] = MFCC1K_MFMA(ax[4 * (i % 3) + i / 3], lm[i % 3][i / 3], d[i % 3]);
    }
}

// mel MFMAs of one wave: template on h so that each wave's list is compile-time
template <int H>
__device__ __forceinline__ void mel_mfmas(const float (&pw)[16], const float (&am)[kAmel], f32x4 (&acc)[kBlocks]) {
    constexpr int n = H ? kN1 : kN0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const int m = H ? kM1[i] : kM0[i], b = H ? kB1[i] : kB0[i];
        acc[b] = MFCC1K_MFMA(am[i], pw[m], acc[b]);
    }
}

__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_fused1024_kernel(mfcc_k::StreamDesc s, Tables t, LaunchGeom g, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsWords];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave;
    const int h = wave & 1;
    const int lo = lane & 15;          // frame column in pass 2 and the MFMA window
    const int q = lane >> 4;           // column offset g in pass 2; K index in the MFMA window
    const int n2 = lane & 31;          // pass 1
    const int fr_id = 2 * wave + (lane >> 5);

    float *const Tt = lds;                                         // [16 frames][1090]: [32 n2][34] each
    float *const Vt = Tt + kTile * kTFrame;                        // [16 frames][34]
    float *const Qt = Vt + kTile * kVStride;                       // [8 waves][3 blocks][256]
    float *const Sf = Qt + kQWords;                                // pre-emphasised sample window, fp32

    v2f wp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wp[i] = reinterpret_cast<const v2f *>(t.win)[n2 * 16 + i];
    v2f tw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tw[i] = reinterpret_cast<const v2f *>(t.tw)[n2 * 16 + i];
    float am[kAmel], ax[kAextra];
#pragma unroll
    for (int i = 0; i < kAmel; ++i) am[i] = t.a_mel[(wave * kAmel + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < kAextra; ++i) ax[i] = role < 3 ? t.a_extra[(role * kAextra + i) * 64 + lane] : 0.0f;

    const int lane_slot = fr_id * kHop + n2;
    const int fetcher = (role - 1) * 64 + lane;     // 0..447 in roles 1..7
    const int lane_off = lo * t.n_cep + 4 * q;

    Cursor cur;
    cur.ch = (int)(blockIdx.x / (unsigned)g.tiles_per_ch);
    cur.t_in = (int)(blockIdx.x - (unsigned)cur.ch * (unsigned)g.tiles_per_ch);
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
        mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);

        v2f *trow = reinterpret_cast<v2f *>(Tt + fr_id * kTFrame + n2 * kTRow);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) trow[k1] = ty[k1];
        Vt[fr_id * kVStride + n2] = y16;
        lds_barrier();                         // B1: T and V of all 16 frames are in LDS; S and Q are consumed

        // ---------------- pass 2: outputs k2 = 2 m + h of the complex FFT-32 over n2, frame lo, column k1
        float pw[16];
        {
            v2f xl[16], xh[16], z[16];
            const v2f *tcol = reinterpret_cast<const v2f *>(Tt + lo * kTFrame + 2 * (4 * (wave >> 1) + q));
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                xl[n] = tcol[n * (kTRow / 2)];
                xh[n] = tcol[(n + 16) * (kTRow / 2)];
            }
            if (h) mfcc_codelets::cfft32_h1(xl, xh, z);
            else mfcc_codelets::cfft32_h0(xl, xh, z);
#pragma unroll
            for (int m = 0; m < 16; ++m) pw[m] = fmaf(z[m].x, z[m].x, z[m].y * z[m].y);
        }

        // ---------------- MFMA window (frame column = lo, K index = q)
        f32x4 acc[kBlocks] = {zero, zero, zero};
        if (role == 1 || role == 2) {
            // column 16 -> bins 16 + 32 j, j = jb + 2 q + {0, 1}, fed to the filter blocks from registers
            f32x4 sp = zero, sp2 = zero;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                sp = MFCC1K_MFMA(ax[k], Vt[lo * kVStride + 4 * k + q], sp);
                sp2 = MFCC1K_MFMA(ax[k + 1], Vt[lo * kVStride + 4 * (k + 1) + q], sp2);
            }
            sp += sp2;
            const float s0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);
            const float s1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);
            if (role == 1) {
#pragma unroll
                for (int i = 0; i < kNS1; ++i) acc[kS1blk[i]] = MFCC1K_MFMA(ax[8 + i], kS1step[i] ? s1 : s0, acc[kS1blk[i]]);
            } else {
#pragma unroll
                for (int i = 0; i < kNS2; ++i) acc[kS2blk[i]] = MFCC1K_MFMA(ax[8 + i], kS2step[i] ? s1 : s0, acc[kS2blk[i]]);
            }
        }
        if (role == 0) {
            f32x4 d[kBlocks] = {zero, zero, zero};
            mel_dct_mfmas(pw, am, ax, lm, acc, d);           // lm = 0 before the first tile
            if (have_prev) dct_store(s, t, d, lm, prev, lo, q, lane, lane_off, out);
        } else if (h) {
            mel_mfmas<1>(pw, am, acc);
        } else {
            mel_mfmas<0>(pw, am, acc);
        }
#pragma unroll
        for (int b = 0; b < kBlocks; ++b)
            *reinterpret_cast<f32x4 *>(Qt + ((wave * kBlocks + b) * 64 + lane) * 4) = acc[b];
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
    hipLaunchKernelGGL(mfcc_fused1024_kernel, dim3((unsigned)grid), dim3(64 * kWaves), 0, stream, s, t, g, out);
    return true;
}

}  // namespace mfcc_fused1024
