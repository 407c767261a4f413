// Fused 512/170/32 float kernel for gfx950 (MI355X): the whole chain of mfcc/core --
// pre-emphasis -> 512-sample frames (hop 170) -> Hamming -> FFT -> |.|^2 -> 32 mel -> log2 ->
// DCT-II -> first n_cep (<= 16) -- in one launch, one wave per tile of 16 consecutive frames,
// no inter-wave synchronisation.
//
// Data flow of one wave (64 lanes), tile = 16 frames, four sub-iterations of 4 frames:
//
//  pass 1  lane = (q = lane>>4: frame of the sub-iteration, n2 = lane&15).  The lane loads the 32
//          samples x[16 n1 + n2] (n1 = 0..31) of its frame -- one 2-byte-aligned dword per sample
//          holding (x[i-1], x[i]), so pre-emphasis 32 x[i] - 31 x[i-1] is one v_dot2c_i32_i16 --
//          and runs a register-resident REAL 32-point FFT over n1 with the Hamming window folded
//          into its first butterfly layer (codelets_gen.hpp).  Outputs Y[k1, n2], k1 = 0..16.
//          Columns k1 = 0..15 are multiplied by W512^(n2 k1) and go to the LDS transpose buffer T;
//          column 16 (real) goes to the LDS tile V.
//  pass 2  lane = (q, k1 = lane&15) reads its column from T and runs a complex 16-point FFT over
//          n2: X[k1 + 32 k2], k2 = 0..15.  Because the input is real, every one of these 256 values
//          is a distinct needed bin (k or 512-k), so there is no real-FFT split/pairing step.
//          |X|^2 goes to the LDS power tile P[frame][bin].
//  MFMA    after four sub-iterations the wave owns P for 16 frames = the N dimension of
//          v_mfma_f32_16x16x4_f32:  (a) column 16 -> bins 16+32j by a 16x16 real matrix (4 MFMAs);
//          (b) mel energies = W (32x256, block-banded) . P, 68 MFMAs, B operand straight from the
//          P tile (one ds_read_b64 per two MFMAs), A operand = constant table; (c) log2 on the
//          accumulators; (d) DCT-II as 8 MFMAs whose B operand IS the mel accumulator registers
//          (the K index is permuted so no lane movement is needed); (e) store 13 floats per frame.
//
// HBM traffic per frame: 170 new int16 samples (the 3x overlap between frames is served by
// L1/L2) + 13 floats out = 392 B.  The kernel is fp32-VALU bound (about 8.5 k lane-ops per
// frame), not HBM bound; DESIGN.md has the accounting.
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
constexpr int kTile = 16;                 // frames per wave tile (MFMA N dimension)
constexpr int kPStride = 260;             // words per frame in the power tile (== 4 mod 64)
constexpr int kTRow = 34;                 // words per n2 row of the transpose buffer
constexpr int kTQ = 16 * kTRow;           // 544 words per frame (== 32 mod 64)
constexpr int kVStride = 18;              // words per frame in the column-16 tile
constexpr int kMelMfma = 68;              // see mel_schedule()
constexpr int kLdsWords = kTile * kPStride + 4 * kTQ + kTile * kVStride;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int int_a2 __attribute__((aligned(2)));

struct FusedTables {
    const float *win;     // [16 n2][32 n1]   hamming[16 n1 + n2] / 32
    const float2 *tw;     // [16 n2][16 k1]   W512^(n2 k1)
    const float *a_sp;    // [4][64]          column-16 DFT matrix, MFMA A layout
    const float *a_mel;   // [68][64]         mel weights / power_scale^2, MFMA A layout
    const float *a_dct;   // [8][64]          DCT-II rows (x lifter), MFMA A layout
    int n_cep;
};

// chunk pair C covers bins 8C .. 8C+7.  Filters 0..15 only touch bins < 64, filters 16..31 only
// bins >= 48 (checked by build_tables), so C <= 7 feeds filter block 0 and C >= 6 feeds block 1.
__host__ __device__ constexpr bool mel_uses(int C, int blk) { return blk == 0 ? C <= 7 : C >= 6; }

inline bool supported(int nfft, int hop, int n_mel, int n_cep) {
    return nfft == kNfft && hop == kHop && n_mel == kMel && n_cep >= 1 && n_cep <= kMaxCep;
}

// ---- host: constant tables in the exact order the kernel consumes them
inline bool build_tables(int sample_rate, double power_scale, double lifter, int n_cep,
                         std::vector<char> &blob) {
    using namespace mfcc_tables;
    std::vector<float> win(16 * 32), tw(16 * 16 * 2), asp(4 * 64), amel(size_t(kMelMfma) * 64),
        adct(8 * 64);
    std::vector<double> w = hamming_periodic(kNfft);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int n1 = 0; n1 < 32; ++n1) win[n2 * 32 + n1] = float(w[16 * n1 + n2] / 32.0);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int k1 = 0; k1 < 16; ++k1) {
            double a = -2.0 * kPi * double(n2 * k1) / 512.0;
            tw[(n2 * 16 + k1) * 2 + 0] = float(std::cos(a));
            tw[(n2 * 16 + k1) * 2 + 1] = float(std::sin(a));
        }
    // column 16: X[16 + 32 k2] = sum_n2 v[n2] W512^(n2 (16 + 32 k2)); MFMA row i = 4g + r holds
    // r=0: Re k2=2g, r=1: Im k2=2g, r=2: Re k2=2g+1, r=3: Im k2=2g+1
    for (int t = 0; t < 4; ++t)
        for (int l = 0; l < 64; ++l) {
            int i = l & 15, n2 = 4 * t + (l >> 4);
            int g = i >> 2, r = i & 3, k2 = 2 * g + (r >> 1);
            double th = 2.0 * kPi * double(n2 * (16 + 32 * k2)) / 512.0;
            asp[t * 64 + l] = float((r & 1) ? -std::sin(th) : std::cos(th));
        }
    std::vector<double> md = mel_dense(kNfft, kMel, double(sample_rate));     // [32][257]
    const double inv = 1.0 / (power_scale * power_scale);
    std::vector<char> covered(size_t(kMel) * 257, 0);
    int idx = 0;
    for (int C = 0; C < 32; ++C)
        for (int blk = 0; blk < 2; ++blk) {
            if (!mel_uses(C, blk)) continue;
            for (int step = 0; step < 2; ++step, ++idx)
                for (int l = 0; l < 64; ++l) {
                    int filt = blk * 16 + (l & 15), bin = 8 * C + 2 * (l >> 4) + step;
                    amel[size_t(idx) * 64 + l] = float(md[size_t(filt) * 257 + bin] * inv);
                    covered[size_t(filt) * 257 + bin] = 1;
                }
        }
    if (idx != kMelMfma) return false;
    for (int f = 0; f < kMel; ++f)
        for (int k = 0; k < 257; ++k)
            if (md[size_t(f) * 257 + k] != 0.0 && !covered[size_t(f) * 257 + k]) return false;
    std::vector<double> dd = dct_rows(n_cep, kMel, lifter);                    // [n_cep][32]
    for (int blk = 0; blk < 2; ++blk)
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                int coeff = l & 15, filt = 16 * blk + 4 * (l >> 4) + r;
                adct[(blk * 4 + r) * 64 + l] = coeff < n_cep ? float(dd[size_t(coeff) * kMel + filt]) : 0.0f;
            }
    auto put = [&](const std::vector<float> &v) {
        size_t off = blob.size();
        blob.resize(off + v.size() * 4);
        std::memcpy(blob.data() + off, v.data(), v.size() * 4);
    };
    blob.clear();
    int32_t hdr[4] = {n_cep, 0, 0, 0};
    blob.resize(16);
    std::memcpy(blob.data(), hdr, 16);
    put(win); put(tw); put(asp); put(amel); put(adct);
    return true;
}

inline void bind_tables(const char *b, int n_cep, FusedTables &t) {
    // device pointer arithmetic only; layout = build_tables' put() order after the 16-byte header
    t.n_cep = n_cep;
    const float *f = reinterpret_cast<const float *>(b + 16);
    t.win = f;                  f += 16 * 32;
    t.tw = reinterpret_cast<const float2 *>(f); f += 16 * 16 * 2;
    t.a_sp = f;                 f += 4 * 64;
    t.a_mel = f;                f += kMelMfma * 64;
    t.a_dct = f;
}

// ---- device

__device__ __forceinline__ float preemph_x32(int packed) {
    // packed = (x[i-1], x[i]) as two int16: 32 x[i] - 31 x[i-1] (exact; 1/32 is in the window table)
    const s16x2 c = {(short)-31, (short)32};
    return (float)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, packed), c, 0, false);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mfcc_fused512_kernel(mfcc_k::StreamDesc s, FusedTables t,
                                                          long long tiles_per_ch, long long n_tiles,
                                                          float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsWords];
    float *const Pt = lds;                                  // [16][260]
    float *const Tt = lds + kTile * kPStride;               // [4][16 n2][34]  (float2 at 2*k1)
    float *const Vt = Tt + 4 * kTQ;                         // [16][18]

    const int lane = threadIdx.x;
    const int lo = lane & 15;          // n2 in pass 1, k1 in pass 2, frame / row index in the MFMA phase
    const int q = lane >> 4;           // frame of the sub-iteration; K index g in the MFMA phase

    // per-lane constants, resident for the whole kernel
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = t.win[lo * 32 + i];
    float2 tw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tw[i] = t.tw[lo * 16 + i];

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long ch = tile / tiles_per_ch;
        const long long f0 = (tile - ch * tiles_per_ch) * kTile;
        const int16_t *base = s.pcm + ch * s.ch_stride;
        // Interior tiles (every sample index in [first - 1, last] exists) load one unaligned dword
        // per sample; edge tiles (stream start without history, zero-padded tail) are bounds-checked.
        // Wave-uniform, so this is a scalar branch.
        const bool inside = (f0 > 0 || s.halo) &&
                            (f0 + kTile - 1) * (long long)kHop + kNfft - 1 < s.n_samples;

        for (int sub = 0; sub < 4; ++sub) {
            // ---------------- pass 1: load + pre-emphasis + windowed real FFT-32 over n1
            const long long i0 = (f0 + sub * 4 + q) * (long long)kHop + lo;   // sample index of n1 = 0
            float e[32];
            if (inside) {
                const int16_t *p = base + i0 - 1;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1)
                    e[n1] = preemph_x32(*reinterpret_cast<const int_a2 *>(p + 16 * n1));
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    const long long i = i0 + 16 * n1;
                    const int x0 = mfcc_k::sample_at_i(s, base, i);
                    const int x1 = mfcc_k::sample_at_i(s, base, i - 1);
                    e[n1] = (float)(32 * x0 - 31 * x1);
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
            Vt[(sub * 4 + q) * kVStride + lo] = yr[16];
            __syncthreads();                 // single-wave workgroup: orders the LDS traffic only

            // ---------------- pass 2: complex FFT-16 over n2 for column k1 = lo
            float xr[16], xi[16], zr[16], zi[16];
            const float2 *tcol = reinterpret_cast<const float2 *>(Tt + q * kTQ + 2 * lo);
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) {
                const float2 v = tcol[n2 * (kTRow / 2)];
                xr[n2] = v.x;
                xi[n2] = v.y;
            }
            mfcc_codelets::cfft16(xr, xi, zr, zi);
            float *prow_lo = Pt + (sub * 4 + q) * kPStride + lo;            // bin k1 + 32 k2
            float *prow_hi = Pt + (sub * 4 + q) * kPStride + 32 - lo;       // bin 32 - k1 + 32 (15 - k2)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) prow_lo[32 * k2] = fmaf(zr[k2], zr[k2], zi[k2] * zi[k2]);
#pragma unroll
            for (int k2 = 8; k2 < 16; ++k2) prow_hi[32 * (15 - k2)] = fmaf(zr[k2], zr[k2], zi[k2] * zi[k2]);
            __syncthreads();
        }

        // ---------------- MFMA phase over the 16 frames of the tile (frame = lo, K index = q)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.a_sp[k * 64 + lane], Vt[lo * kVStride + 4 * k + q],
                                                       acc, 0, 0, 0);
        Pt[lo * kPStride + 16 + 64 * q] = fmaf(acc[0], acc[0], acc[1] * acc[1]);
        Pt[lo * kPStride + 48 + 64 * q] = fmaf(acc[2], acc[2], acc[3] * acc[3]);
        __syncthreads();

        f32x4 m0x = {0.f, 0.f, 0.f, 0.f}, m0y = m0x, m1x = m0x, m1y = m0x;
        const float2 *pp = reinterpret_cast<const float2 *>(Pt + lo * kPStride + 2 * q);
        const float *am = t.a_mel + lane;
        int idx = 0;
#pragma unroll
        for (int C = 0; C < 32; ++C) {
            const float2 p = pp[4 * C];
            if (mel_uses(C, 0)) {
                m0x = __builtin_amdgcn_mfma_f32_16x16x4f32(am[(idx + 0) * 64], p.x, m0x, 0, 0, 0);
                m0y = __builtin_amdgcn_mfma_f32_16x16x4f32(am[(idx + 1) * 64], p.y, m0y, 0, 0, 0);
                idx += 2;
            }
            if (mel_uses(C, 1)) {
                m1x = __builtin_amdgcn_mfma_f32_16x16x4f32(am[(idx + 0) * 64], p.x, m1x, 0, 0, 0);
                m1y = __builtin_amdgcn_mfma_f32_16x16x4f32(am[(idx + 1) * 64], p.y, m1y, 0, 0, 0);
                idx += 2;
            }
        }
        // log2 of the mel energies (MFCC.ipynb cell 36), then DCT-II rows as MFMA: the
        // accumulator register r of block b is filter 16 b + 4 q + r of frame lo == B[k = q][j = lo]
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float l0 = log2f(m0x[r] + m0y[r]);
            const float l1 = log2f(m1x[r] + m1y[r]);
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t.a_dct[r * 64 + lane], l0, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t.a_dct[(4 + r) * 64 + lane], l1, d1, 0, 0, 0);
        }
        const long long fr = f0 + lo;
        if (fr < s.frames_per_ch) {
            float *o = out + (ch * s.frames_per_ch + fr) * t.n_cep;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * q + r < t.n_cep) o[4 * q + r] = d0[r] + d1[r];
        }
        __syncthreads();
    }
}

inline const char *kernel_name() { return "mfcc_fused512_kernel"; }

inline void launch(const mfcc_k::StreamDesc &s, const FusedTables &t, float *out, int n_cu,
                   hipStream_t stream) {
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    long long grid = n_tiles < (long long)n_cu * 6 ? n_tiles : (long long)n_cu * 6;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(mfcc_fused512_kernel, dim3((unsigned)grid), dim3(64), 0, stream, s, t, tiles_per_ch,
                       n_tiles, out);
}

}  // namespace mfcc_fused
