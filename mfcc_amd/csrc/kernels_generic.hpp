// Generic (correctness-first) gfx950 kernels: one frame per 64-lane wave, four waves per
// workgroup, the frame staged in LDS.
//
//   mfcc_float_generic_kernel<NFFT> : float contract (notebook/MFCC.ipynb), any n_mel <= 64
//   mfcc_fixed_kernel               : fixed contract (the RTL arithmetic), bit-exact int16
//
// The specialised 512/170/32 kernel lives in kernel_fused512.hpp; these are the reference
// implementations on the device and the fallback for other parameter sets.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfcc_k {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;
constexpr int kMaxMel = 64;

struct StreamDesc {
    const int16_t *pcm;        // first frame sample of channel 0 (after the halo sample, if any)
    long long ch_stride;       // samples between channels
    long long n_samples;       // valid samples per channel (x[i] = 0 for i >= n_samples)
    int halo;                  // 1: pcm[-1] is real history; 0: history is 0
    long long frames_per_ch;
    long long total_frames;    // frames_per_ch * n_channels
    int hop;
};

__device__ __forceinline__ float sample_at(const StreamDesc &s, const int16_t *base, long long i) {
    // x[i] of one channel: 0 beyond the end (stream padding), history before the start
    if (i >= s.n_samples) return 0.0f;
    if (i < 0) return s.halo ? float(base[-1]) : 0.0f;
    return float(base[i]);
}

__device__ __forceinline__ int sample_at_i(const StreamDesc &s, const int16_t *base, long long i) {
    if (i >= s.n_samples) return 0;
    if (i < 0) return s.halo ? int(base[-1]) : 0;
    return int(base[i]);
}

// ------------------------------------------------------------------------------ float

struct FloatTables {
    const float  *window;     // [NFFT]
    const float2 *tw_fft;     // [NFFT/2]     W_M^m, M = NFFT/2 (complex FFT length)
    const float2 *tw_split;   // [NFFT/2 + 1] W_NFFT^k
    const int    *mel_start;  // [n_mel]
    const int    *mel_count;  // [n_mel]
    const int    *mel_off;    // [n_mel] offset into mel_w
    const float  *mel_w;      // packed weights, 1/power_scale^2 folded in
    int mel_w_total;          // entries of mel_w: at most 2 per bin (staged in LDS by the kernel)
    const float  *dct;        // [n_cep][n_mel] (lifter folded in)
    const double *window_d;   // [NFFT] the window in double, or nullptr: set when a mel filter has weight on bin 0 --
                              // that bin is then accumulated in double (kernel_fused512.hpp, FusedTables::win_dc)
    int n_mel, n_cep;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// One workgroup = 4 waves, each wave owns one frame per iteration.
// LDS per wave: two ping-pong buffers of M complex (Stockham), later reused for power/mel.
// every LDS buffer below is private to a wave: a wavefront-scope fence orders its lanes' accesses
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NFFT>
__global__ __launch_bounds__(kBlock) void mfcc_float_generic_kernel(StreamDesc s, FloatTables t,
                                                                  float *__restrict__ out) {
    constexpr int M = NFFT / 2;                 // complex FFT length (real-FFT packing)
    __shared__ float2 bufA[kWavesPerBlock][M];
    __shared__ float2 bufB[kWavesPerBlock][M + 1];
    __shared__ float  melv[kWavesPerBlock][kMaxMel];
    __shared__ float  melw[NFFT + 8];           // the filterbank weights, once per workgroup: a filter's tap loop would
    for (int i = threadIdx.x; i < t.mel_w_total; i += kBlock) melw[i] = t.mel_w[i];   // otherwise wait for a global load per tap
    // ... and so would every butterfly for its twiddles: up to 512 points the window, both twiddle tables and (when small
    // enough) the DCT rows are staged too -- what round 2 found on the fixed-point twin of this kernel (SQ_WAIT_ANY 59 %
    // on flat loads): 512 points 10.2 -> 7.2 ms.  At 1024 points the tables would take the LDS that keeps a second and
    // third workgroup on the CU (2.28 -> 3.38 ms, measured): there they stay in global memory.
    constexpr bool kStage = NFFT <= 512;
    constexpr int kDctLds = kStage ? 1024 : 1;
    __shared__ float  winl[kStage ? NFFT : 1];
    __shared__ float2 twl[kStage ? M : 1];
    __shared__ float2 twsl[kStage ? M + 1 : 1];
    __shared__ float  dctl[kDctLds];
    const bool dct_in_lds = kStage && t.n_cep * t.n_mel <= kDctLds;
    if constexpr (kStage) {
        for (int i = threadIdx.x; i < NFFT; i += kBlock) winl[i] = t.window[i];
        for (int i = threadIdx.x; i < M; i += kBlock) twl[i] = t.tw_fft[i];
        for (int i = threadIdx.x; i <= M; i += kBlock) twsl[i] = t.tw_split[i];
        if (dct_in_lds)
            for (int i = threadIdx.x; i < t.n_cep * t.n_mel; i += kBlock) dctl[i] = t.dct[i];
    }
    const float *const win_p = kStage ? winl : t.window;
    const float2 *const tw_p = kStage ? twl : t.tw_fft;
    const float2 *const tws_p = kStage ? twsl : t.tw_split;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // frame cursor of this wave: (ch, f), advanced by the number of waves in the grid without a division
    const long long waves_total = (long long)gridDim.x * kWavesPerBlock;
    const long long step_ch = waves_total / s.frames_per_ch, step_f = waves_total % s.frames_per_ch;
    long long fid = (long long)blockIdx.x * kWavesPerBlock + wave;
    long long ch = fid / s.frames_per_ch, f = fid % s.frames_per_ch;

    for (; fid < s.total_frames; fid += waves_total, ch += step_ch, f += step_f) {
        if (f >= s.frames_per_ch) {
            f -= s.frames_per_ch;
            ++ch;
        }
        const bool valid = true;
        const int16_t *base = s.pcm + ch * s.ch_stride;
        const long long n0 = f * (long long)s.hop;

        // pre-emphasis (MFCC.ipynb cell 7) + Hamming (cell 18); z[m] = y[2m] + i y[2m+1]
        float *za = reinterpret_cast<float *>(bufA[wave]);
        for (int i = lane; i < NFFT; i += 64) {
            float x0 = valid ? sample_at(s, base, n0 + i) : 0.0f;
            float x1 = valid ? sample_at(s, base, n0 + i - 1) : 0.0f;
            // y[0] = x[0] for the very first sample of a stream: history is 0 there
            float y = x0 - 0.96875f * x1;
            za[i] = y * win_p[i];
        }
        double dc = 0.0;
        if (t.window_d) {                       // uniform: X[0] = sum w y in double (y is exact)
            for (int i = lane; i < NFFT; i += 64) {
                const double yd = (double)sample_at_i(s, base, n0 + i) - 0.96875 * (double)sample_at_i(s, base, n0 + i - 1);
                dc = __builtin_fma(t.window_d[i], yd, dc);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dc += __shfl_xor(dc, o, 64);
        }
        wave_sync();

        // Stockham autosort FFT of length M: radix-4 passes, one radix-2 pass if needed
        float2 *src = bufA[wave];
        float2 *dst = bufB[wave];
        int Ns = 1;
        for (; Ns * 4 <= M; Ns *= 4) {
            const int tstride = M / (Ns * 4);
            for (int j = lane; j < M / 4; j += 64) {
                const int k = j & (Ns - 1);
                float2 v0 = src[j];
                float2 v1 = cmul(src[j + M / 4], tw_p[k * tstride]);
                float2 v2 = cmul(src[j + 2 * (M / 4)], tw_p[2 * k * tstride]);
                float2 v3 = cmul(src[j + 3 * (M / 4)], tw_p[3 * k * tstride]);
                float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
                float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
                float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
                float2 a3 = make_float2(v1.y - v3.y, v3.x - v1.x);      // -i * (v1 - v3)
                const int j0 = ((j - k) << 2) + k;
                dst[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
                dst[j0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
                dst[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
                dst[j0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
            }
            wave_sync();
            float2 *tmp = src; src = dst; dst = tmp;
        }
        if (Ns < M) {                       // remaining radix-2 pass (M = 2 * 4^n)
            const int tstride = M / (Ns * 2);
            for (int j = lane; j < M / 2; j += 64) {
                const int k = j & (Ns - 1);
                float2 v0 = src[j];
                float2 v1 = cmul(src[j + M / 2], tw_p[k * tstride]);
                const int j0 = ((j - k) << 1) + k;
                dst[j0] = make_float2(v0.x + v1.x, v0.y + v1.y);
                dst[j0 + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
            }
            wave_sync();
            float2 *tmp = src; src = dst; dst = tmp;
        }

        // real-FFT split + power spectrum (cells 20, 22): P[k], k = 0..M, into dst (as floats)
        float *P = reinterpret_cast<float *>(dst);
        for (int k = lane; k <= M; k += 64) {
            float2 a = src[k & (M - 1)];
            float2 b = src[(M - k) & (M - 1)];
            float er = 0.5f * (a.x + b.x), ei = 0.5f * (a.y - b.y);      // E = (a + conj b)/2
            float dr = a.x - b.x, di = a.y + b.y;                         // a - conj b
            float orr = 0.5f * di, oi = -0.5f * dr;                       // O = -i/2 (a - conj b)
            float2 w = tws_p[k];
            float xr = er + (w.x * orr - w.y * oi);
            float xi = ei + (w.x * oi + w.y * orr);
            P[k] = (k == 0 && t.window_d) ? (float)(dc * dc) : xr * xr + xi * xi;
        }
        wave_sync();

        // mel filterbank (cells 30, 36) + log2
        if (lane < t.n_mel) {
            const int st = t.mel_start[lane], cnt = t.mel_count[lane];
            const float *w = melw + t.mel_off[lane];
            float acc = 0.0f;
#pragma unroll 4
            for (int j = 0; j < cnt; ++j) acc = fmaf(P[st + j], w[j], acc);
            melv[wave][lane] = log2f(acc);
        }
        wave_sync();

        // DCT-II, first n_cep rows (cells 38-39)
        if (valid && lane < t.n_cep) {
            float acc = 0.0f;
            if (dct_in_lds) {
                const float *d = dctl + lane * t.n_mel;
                for (int n = 0; n < t.n_mel; ++n) acc = fmaf(d[n], melv[wave][n], acc);
            } else {
                const float *d = t.dct + lane * t.n_mel;
                for (int n = 0; n < t.n_mel; ++n) acc = fmaf(d[n], melv[wave][n], acc);
            }
            out[fid * t.n_cep + lane] = acc;
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------ fixed

struct FixedTables {
    const int   *curve;       // [nfft]     window curve (mfcc/core/window.py)
    const uint2 *tw_fft;      // [nfft/2]   Q14 twiddles of the nfft-point FFT as dot2 operand pairs (see fx_bfly)
    const uint2 *tw_dct;      // [2*n_mel]  the same for the (4*n_mel)-point FFT
    const int   *mel_start;   // [n_mel]
    const int   *mel_count;   // [n_mel]
    const int   *mel_off;     // [n_mel]
    const uint32_t *mel_w;    // packed weights (x 2^-30)
    int mel_w_total;          // entries of mel_w (staged in LDS by the kernel)
    int mel_shift;
    int nfft, log2_nfft, n_mel, log2_mel, log2_dct, n_cep;
};

__device__ __forceinline__ int wrap16(int v) { return (int)(short)(v & 0xFFFF); }

// ---- packed fixed-point butterfly, shared with kernel_fixed512.hpp.  A complex value is one dword, (re, im)
// as two int16 (every stage wraps to 16 bits anyway); a twiddle is a pair of dot2 operands:
// A = (twr, -twi) gives s1 = x1r*twr - x1i*twi, B = (twi, twr) gives s2 = x1r*twi + x1i*twr -- the same integers as
// the RTL's three-multiplier form (mfcc/misc/fft.py:140-192; nothing overflows 33 bits).

// (x0 +- a) >> 1 with 16-bit wrap on packed x0; a1 / a2 are the rotated x1 (re, im).  8 VALU ops: four
// SDWA adds that sign-extend x0's halves on the fly, and per output a shift plus an SDWA shift that
// lands in the high word (the 16-bit wrap is the truncation to a half).
__device__ __forceinline__ void fx_combine(uint32_t p0, int a1, int a2, uint32_t &o0, uint32_t &o1) {
    int t0, t1, t2, t3;
    asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
        : "=v"(t0) : "v"(p0), "v"(a1));
    asm("v_sub_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
        : "=v"(t1) : "v"(p0), "v"(a1));
    asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
        : "=v"(t2) : "v"(p0), "v"(a2));
    asm("v_sub_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
        : "=v"(t3) : "v"(p0), "v"(a2));
    const int one = 1;
    uint32_t r0 = (uint32_t)t0 >> 1, r1 = (uint32_t)t1 >> 1;
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(r0) : "s"(one), "v"(t2));
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(r1) : "s"(one), "v"(t3));
    o0 = r0;
    o1 = r1;
}

// (dot2(p1, tw) + 8191) >> 14: the three-operand form of the dot product (the builtin becomes v_dot2c, which
// needs a v_mov of the bias first)
__device__ __forceinline__ int fx_rot14(uint32_t p1, uint32_t tw) {
    const int bias = 8191;
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(p1), "v"(tw), "s"(bias));
    return r >> 14;
}

// The whole butterfly in NINE vector instructions (round 2: twelve -- fx_rot14 x 2 + fx_combine).  With
// s = dot2(x1, tw) + 8191 and b = s >> 14 (the rotated x1, floor), the RTL's outputs are y0 = wrap16((x0 + b) >> 1) and
// y1 = wrap16((x0 - b) >> 1) per component (mfcc/misc/fft.py:140-192).  Two identities over the integers:
//   (x0 + (s >> 14)) >> 1  ==  (x0 * 2^14 + s) >> 15          floor of a floor; |x0 2^14 + s| < 2^31, SURVEY.md A.4
//   (x0 + b) >> 1  -  (x0 - b) >> 1  ==  b                     x0 + b and x0 - b have the same parity
// so y0 is ONE v_mad_i32_i16 per component on top of the dot product (the multiply-add takes the 16-bit half of the
// packed x0 it needs, sign-extended, by op_sel) plus the pack, and y1 = y0 - b is one PACKED 16-bit subtraction once b's
// low halves are packed -- the wrap to 16 bits is the packed arithmetic itself.
__device__ __forceinline__ void fx_bfly(uint32_t &p0, uint32_t &p1, uint32_t twa, uint32_t twb) {
    const int bias = 8191, two14 = 16384, fifteen = 15, fourteen = 14;
    int s1, s2, tr, ti;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(s1) : "v"(p1), "v"(twa), "s"(bias));
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(s2) : "v"(p1), "v"(twb), "s"(bias));
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(tr) : "v"(p0), "s"(two14), "v"(s1));                    // x0r 2^14 + s1
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ti) : "v"(p0), "s"(two14), "v"(s2));   // x0i 2^14 + s2
    uint32_t y0 = (uint32_t)tr >> 15, b = (uint32_t)s1 >> 14;            // the low halves; the high ones are written next
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(y0) : "s"(fifteen), "v"(ti));
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(b) : "s"(fourteen), "v"(s2));
    uint32_t y1;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(y1) : "v"(y0), "v"(b));
    p0 = y0;
    p1 = y1;
}

// the same when only y0 is read out (the last stage of the frame FFT keeps bins 0..255): six instructions
__device__ __forceinline__ void fx_bfly_y0(uint32_t &p0, uint32_t p1, uint32_t twa, uint32_t twb) {
    const int bias = 8191, two14 = 16384, fifteen = 15;
    int s1, s2, tr, ti;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(s1) : "v"(p1), "v"(twa), "s"(bias));
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(s2) : "v"(p1), "v"(twb), "s"(bias));
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(tr) : "v"(p0), "s"(two14), "v"(s1));
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ti) : "v"(p0), "s"(two14), "v"(s2));
    uint32_t y0 = (uint32_t)tr >> 15;
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(y0) : "s"(fifteen), "v"(ti));
    p0 = y0;
}

// twiddle T[0] = (16384, 0): the rotated x1 is x1 itself ((x 16384 + 8191) >> 14 == x), so b is the packed x1 as it is:
// y0 = (x0 + x1) >> 1 per component in 17 bits (two SDWA adds, a shift, an SDWA shift), y1 = y0 - x1 packed.  Five.
__device__ __forceinline__ void fx_bfly_one(uint32_t &p0, uint32_t &p1) {
    const int one = 1;
    int tr, ti;
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0"
        : "=v"(tr) : "v"(p0), "v"(p1));
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1"
        : "=v"(ti) : "v"(p0), "v"(p1));
    uint32_t y0 = (uint32_t)tr >> 1;
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(y0) : "s"(one), "v"(ti));
    uint32_t y1;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(y1) : "v"(y0), "v"(p1));
    p0 = y0;
    p1 = y1;
}

// LDS index of point i of the in-place FFT buffer: one pad word per 32 points.  The butterfly groups of a stage
// sit 2^st points apart, so without the pad the lanes of a wave hit 4..8 distinct banks (measured on the plain
// layout: SQ_LDS_BANK_CONFLICT = 69 % of SQ_LDS_IDX_ACTIVE)
__device__ __forceinline__ int fxi(int i) { return i + (i >> 5); }

// S consecutive radix-2 DIT stages st .. st + S - 1 on 2^S register-resident points per lane group, ONE LDS round
// trip: group gidx holds the points base + a 2^st, a = 0 .. 2^S - 1, with j = gidx mod 2^st and
// base = (gidx >> st) << (st + S) | j.  Stage st + u pairs a with a | 2^u; the element's index inside that stage's
// butterfly group is j + (a mod 2^u) 2^st.  Every butterfly is the RTL's own (fx_bfly) on the same operands as in the
// stage-by-stage schedule of mfcc/misc/fft.py:216-344, so the result is the same bit for bit.
template <int S>
__device__ __forceinline__ void fx_fft_pass(uint32_t *x, int size, int L, int st, const uint2 *tw, int lane) {
    const int half = size >> 1;
    for (int gidx = lane; gidx < (size >> S); gidx += 64) {
        const int j = gidx & ((1 << st) - 1);
        const int base = ((gidx >> st) << (st + S)) | j;
        uint32_t v[1 << S];
#pragma unroll
        for (int a = 0; a < (1 << S); ++a) v[a] = x[fxi(base + (a << st))];
#pragma unroll
        for (int u = 0; u < S; ++u) {
#pragma unroll
            for (int a = 0; a < (1 << S); ++a) {
                if (a & (1 << u)) continue;
                const int jj = j + ((a & ((1 << u) - 1)) << st);
                const uint2 w = tw[(jj << (L - 1 - (st + u))) & (half - 1)];
                fx_bfly(v[a], v[a | (1 << u)], w.x, w.y);
            }
        }
#pragma unroll
        for (int a = 0; a < (1 << S); ++a) x[fxi(base + (a << st))] = v[a];
    }
    wave_sync();
}

// the same with every size known at compile time (the frame FFT): index arithmetic folds to a few shifts, loops unroll
template <int S, int L, int ST>
__device__ __forceinline__ void fx_fft_pass_ct(uint32_t *x, const uint2 *tw, int lane) {
    constexpr int size = 1 << L, half = size >> 1, groups = size >> S;
#pragma unroll 1
    for (int g0 = 0; g0 < groups; g0 += 64) {
        const int gidx = g0 + lane;
        if (groups >= 64 || gidx < groups) {
            const int j = gidx & ((1 << ST) - 1);
            const int base = ((gidx >> ST) << (ST + S)) | j;
            uint32_t v[1 << S];
#pragma unroll
            for (int a = 0; a < (1 << S); ++a) v[a] = x[fxi(base + (a << ST))];
#pragma unroll
            for (int u = 0; u < S; ++u) {
#pragma unroll
                for (int a = 0; a < (1 << S); ++a) {
                    if (a & (1 << u)) continue;
                    const int jj = j + ((a & ((1 << u) - 1)) << ST);
                    const uint2 w = tw[(jj << (L - 1 - (ST + u))) & (half - 1)];
                    fx_bfly(v[a], v[a | (1 << u)], w.x, w.y);
                }
            }
#pragma unroll
            for (int a = 0; a < (1 << S); ++a) x[fxi(base + (a << ST))] = v[a];
        }
    }
    wave_sync();
}

// stages ST .. L - 1 in passes of S stages where that keeps all 64 lanes busy (2^L >> S >= 64), fewer otherwise
template <int L, int ST>
__device__ __forceinline__ void fx_fft_ct(uint32_t *x, const uint2 *tw, int lane) {
    if constexpr (ST < L) {
        constexpr int left = L - ST;
        constexpr int want = (L >= 9) ? 3 : (L >= 8 ? 2 : 1);
        constexpr int S = left < want ? left : want;
        fx_fft_pass_ct<S, L, ST>(x, tw, lane);
        fx_fft_ct<L, ST + S>(x, tw, lane);
    }
}

// in-place radix-2 DIT over `size` packed points held in LDS (bit-reversed input order), three stages per LDS round
// trip (then what is left); schedule of mfcc/misc/fft.py:216-344 (see oracle/mfcc_fixed.py: fft_fixed); tw: packed
// operand pairs [size/2]
__device__ __forceinline__ void fx_fft_inplace(uint32_t *x, int size, int L, const uint2 *tw, int lane) {
    int st = 0;
    for (; st + 3 <= L; st += 3) fx_fft_pass<3>(x, size, L, st, tw, lane);
    if (L - st == 2) fx_fft_pass<2>(x, size, L, st, tw, lane);
    else if (L - st == 1) fx_fft_pass<1>(x, size, L, st, tw, lane);
}

constexpr int kFxMaxNfft = 1024;

template <int NFFT>
__global__ __launch_bounds__(kBlock) void mfcc_fixed_kernel(StreamDesc s, FixedTables t,
                                                          int16_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // per wave (nothing is shared between waves): uint32 x[nfft] packed (re, im); uint32 P[nfft/2]; int mel[64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int nfft = NFFT;
    constexpr int xw = nfft + nfft / 32;                       // padded FFT buffer (fxi)
    uint32_t *x = reinterpret_cast<uint32_t *>(smem) + (size_t)wave * xw;
    uint32_t *P = reinterpret_cast<uint32_t *>(smem + (size_t)kWavesPerBlock * xw * sizeof(uint32_t)) +
                  (size_t)wave * (nfft / 2);
    int *melv = reinterpret_cast<int *>(smem + (size_t)kWavesPerBlock * xw * sizeof(uint32_t) +
                                        (size_t)kWavesPerBlock * (nfft / 2) * sizeof(uint32_t)) +
                wave * kMaxMel;
    // the filterbank weights, once per workgroup (the loop over a filter's taps would otherwise wait for a global
    // load per tap: at 16 filters and nfft 1024 that was most of the kernel's time)
    uint32_t *melw = reinterpret_cast<uint32_t *>(smem + (size_t)kWavesPerBlock * (xw + nfft / 2 + kMaxMel) * sizeof(uint32_t));
    for (int i = threadIdx.x; i < t.mel_w_total; i += kBlock) melw[i] = t.mel_w[i];
    // ... and both twiddle ROMs: a butterfly's twiddle out of global memory is a load the whole stage waits for
    // (28 dependent round trips per frame at nfft 256: SQ_WAIT_ANY was 59 % of the wave cycles)
    uint2 *tw1 = reinterpret_cast<uint2 *>(melw + ((t.mel_w_total + 1) & ~1));
    uint2 *tw2 = tw1 + nfft / 2;
    for (int i = threadIdx.x; i < nfft / 2; i += kBlock) tw1[i] = t.tw_fft[i];
    for (int i = threadIdx.x; i < 2 * t.n_mel; i += kBlock) tw2[i] = t.tw_dct[i];
    __syncthreads();

    // frame cursor of this wave: (ch, f), advanced by the number of waves in the grid without a division (a 64-bit
    // division per frame cost as many instructions as a 256-point FFT)
    const long long waves_total = (long long)gridDim.x * kWavesPerBlock;
    const long long step_ch = waves_total / s.frames_per_ch, step_f = waves_total % s.frames_per_ch;
    constexpr int L = __builtin_ctz(NFFT);
    long long fid = (long long)blockIdx.x * kWavesPerBlock + wave;
    long long ch = fid / s.frames_per_ch, f = fid % s.frames_per_ch;

    for (; fid < s.total_frames; fid += waves_total, ch += step_ch, f += step_f) {
        if (f >= s.frames_per_ch) {
            f -= s.frames_per_ch;
            ++ch;
        }
        const bool valid = true;
        const int16_t *base = s.pcm + ch * s.ch_stride;
        const long long n0 = f * (long long)s.hop;

        // preemph.py:24  y = wrap16(x + (o >> 5) - o);  window.py:84  (y * curve) >> 9;
        // fft.py:413-424 bit-reversed load, imag = 0
        // frames that lie inside the stream (all but the first and the padded tail) take plain loads
        const bool inside = valid && n0 - 1 >= -(long long)s.halo && n0 + nfft <= s.n_samples;
#pragma unroll 4
        for (int i0 = 0; i0 < nfft; i0 += 64) {
            const int i = i0 + lane;
            int x0, o;
            if (inside) {
                x0 = base[n0 + i];
                o = base[n0 + i - 1];
            } else {
                x0 = valid ? sample_at_i(s, base, n0 + i) : 0;
                o = valid ? sample_at_i(s, base, n0 + i - 1) : 0;
            }
            int y = wrap16(x0 + (o >> 5) - o);
            int w = __mul24(y, t.curve[i]) >> 9;
            int r = (int)(__brev((unsigned)i) >> (32 - L));
            x[fxi(r)] = (uint32_t)w & 0xffffu;
        }
        wave_sync();
        fx_fft_ct<L, 0>(x, tw1, lane);

        // pow2.py:32,64  (re^2 + im^2) >> 2, 30 bits
#pragma unroll 4
        for (int k0 = 0; k0 < nfft / 2; k0 += 64) {
            const int k = k0 + lane;
            if (nfft / 2 < 64 && k >= nfft / 2) break;           // nfft 64: half a wave
            const uint32_t xk = x[fxi(k)];
            const int re = (int)(short)(xk & 0xffffu), im = (int)xk >> 16;
            uint32_t r = (uint32_t)(re * re) + (uint32_t)(im * im);
            P[k] = r >> 2;
        }
        wave_sync();

        // filterbank.py:88-142 in closed form (tables.hpp: fx_mel), then log.py Log2Fix(16, 15).  64 / n_mel lanes
        // share a filter (lane = part * n_mel + filter), each sums every (64 / n_mel)-th of its taps out of the LDS copy
        // of the weights; 64-bit integer partial sums are exact, so their order does not matter
        {
            const int f = lane & (t.n_mel - 1), part = lane >> t.log2_mel, parts = 64 >> t.log2_mel;
            const int st = t.mel_start[f], cnt = t.mel_count[f];
            const uint32_t *w = melw + t.mel_off[f];
            unsigned long long acc = 0;
#pragma unroll 4
            for (int j = part; j < cnt; j += parts) acc += (unsigned long long)P[st + j] * (unsigned long long)w[j];
            for (int o = t.n_mel; o < 64; o <<= 1) {
                const unsigned lo = __shfl_xor((unsigned)acc, o, 64), hi = __shfl_xor((unsigned)(acc >> 32), o, 64);
                acc += ((unsigned long long)hi << 32) | lo;
            }
            if (lane < t.n_mel) {
                unsigned v = (unsigned)(acc >> t.mel_shift) & 0xFFFFu;
                // Turner log2, Q4.11: precision 11, 10 squarings (log.py:33-102); the normalisation loop as one clz
                unsigned xx = (v ? v : 1u) << 11;
                const int sh = (32 - 12) - __clz(xx) > 0 ? (32 - 12) - __clz(xx) : 0;       // xx >= 2^12  <=>  sh > 0
                unsigned o = (unsigned)sh << 11;
                xx >>= sh;
                unsigned z = xx, b = 1u << 10;
#pragma unroll
                for (int c = 0; c < 10; ++c) {
                    unsigned cc = z * z;
                    if (cc & (1u << 23)) { z = cc >> 12; o += b; } else { z = cc >> 11; }
                    b >>= 1;
                }
                melv[lane] = (int)(o & 0x7FFFu);
            }
        }
        wave_sync();

        // dct_stream.py:23-33: y[2n+1] = y[size-1-2n] = x[n], zeros elsewhere; FFT(4*n_mel)
        const int dsz = 4 * t.n_mel;
        for (int i = lane; i < dsz; i += 64) {
            int v = 0;
            if (i & 1) {
                int n = (i < 2 * t.n_mel) ? (i >> 1) : ((dsz - 1 - i) >> 1);
                v = melv[n];
            }
            int r = (int)(__brev((unsigned)i) >> (32 - t.log2_dct));
            x[fxi(r)] = (uint32_t)v & 0xffffu;
        }
        wave_sync();
        switch (t.log2_dct) {                         // 4 n_mel points: compile-time sizes for the usual filter counts
            case 6: fx_fft_ct<6, 0>(x, tw2, lane); break;
            case 7: fx_fft_ct<7, 0>(x, tw2, lane); break;
            case 8: fx_fft_ct<8, 0>(x, tw2, lane); break;
            default: fx_fft_inplace(x, dsz, t.log2_dct, tw2, lane); break;
        }
        if (valid && lane < t.n_cep) out[fid * t.n_cep + lane] = (int16_t)(x[fxi(lane)] & 0xffffu);
        wave_sync();
    }
}

}  // namespace mfcc_k
