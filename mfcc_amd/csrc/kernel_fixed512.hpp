// Fused fixed-point kernel for gfx950 (MI355X): the RTL arithmetic of mfcc/core for its own
// configuration -- MFCC(width=16, nfft=512, nfilters=32), mfcc/core/mfcc.py:20-88 -- bit for bit:
// pre-emphasis (preemph.py:24) -> window curve (window.py:84) -> 512-point radix-2 DIT FFT with Q14
// twiddles, +8191 >>14 and a >>1 per stage with 16-bit wrap (misc/fft.py:93-96,140-192) ->
// (re^2+im^2)>>2 (pow2.py:32,64) -> filterbank in closed form (filterbank.py:88-142, tables.hpp:
// fx_mel) -> Turner log2 Q4.11 (log.py:33-102) -> 128-point fixed FFT as DCT (dct_stream.py:23-44)
// -> first n_cep int16 (misc/discard.py).  Other parameter sets take mfcc_fixed_kernel.
//
// One frame per wave, four waves per workgroup, nothing shared between waves but read-only tables.
// Every butterfly is the RTL's own (same products, same bias, same shifts, same wraps); what changes
// is where the data lives:
//
//  * a complex value is one dword, (re, im) as two int16 -- every stage wraps to 16 bits anyway;
//  * s1 = x1r*twr - x1i*twi + 8191 and s2 = x1r*twi + x1i*twr + 8191 are one v_dot2_i32_i16 each on
//    the packed value (the RTL's three-multiplier form (x1r+x1i)*twr - x1i*(twr+twi) is the same
//    integer, it never overflows 33 bits: SURVEY.md A.4);
//  * the 9 stages run as three rounds of three stages on 8 register-resident values per lane (index
//    bits 0-2, 3-5, 6-8), with two transposes through LDS instead of nine stage round trips; the
//    twiddles of a round depend on the lane only and live in registers for the whole kernel;
//  * the last stage computes only the outputs that are read out (bins 0..255);
//  * in the DCT's 128-point FFT every even input is 0, so the lower half of the bit-reversed array
//    stays 0 until the last stage: stages 0-5 are a 64-point FFT of the upper half, one value per
//    lane, partners exchanged with lane shuffles, and the last stage is one rotation per output.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <vector>

#include "kernels_generic.hpp"
#include "tables.hpp"

namespace mfcc_fixed512 {

constexpr int kNfft = 512, kMel = 32, kWaves = 4;
constexpr int kXWords = 512 + 64;        // transpose buffer: index i lives at i + 8 (i >> 6)
constexpr int kMelWMax = 1024;           // packed filterbank weights held in LDS (sum of row lengths, + 8 slack)
constexpr int kRawWords = 65 * 4;        // raw-sample staging: 65 aligned 16-byte pieces cover a frame + history

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int16_t __attribute__((may_alias)) i16_alias;     // LDS staged as 16-byte pieces, read back as samples

struct Tables {
    const int *curve8;        // [64 lanes][8]   window curve of sample lane + 64 m
    const uint32_t *tw_r2;    // [8 lo3][7][2]   round-2 twiddles (stages 3, 4, 5) as dot2 operand pairs
    const uint32_t *tw_r3;    // [64 lanes][7][2] round-3 twiddles (stages 6, 7, 8)
    const uint32_t *tw_dct;   // [64 lanes][7][2] DCT FFT: stages 0-5 of the upper half + last-stage rotation
    uint32_t tw64a, tw64b, tw192a, tw192b;   // stage-2 twiddles T[64], T[192]
    const int *mel_start, *mel_count, *mel_off;
    const uint32_t *mel_w;
    int mel_shift, mel_w_total, n_cep;
};

inline bool supported(int nfft, int n_mel, int n_cep) { return nfft == kNfft && n_mel == kMel && n_cep >= 1 && n_cep <= kMel; }

// dot2 operand pair of a twiddle: A = (twr, -twi) gives s1, B = (twi, twr) gives s2
inline void tw_pair(int re, int im, uint32_t &a, uint32_t &b) {
    a = (uint32_t(re) & 0xffffu) | (uint32_t(-im) << 16);
    b = (uint32_t(im) & 0xffffu) | (uint32_t(re) << 16);
}

// host: tables in the order the kernel consumes them.  blob layout: curve8 | tw_r2 | tw_r3 | tw_dct
inline bool build_tables(std::vector<char> &blob, uint32_t (&tw_s2)[4]) {
    using namespace mfcc_tables;
    std::vector<int> cv = fx_window_curve(kNfft);
    std::vector<int> re, im;
    fx_twiddles(kNfft, re, im);                       // 256 entries
    // every twiddle must fit the int16 halves of a dot2 operand, negated too
    for (size_t k = 0; k < re.size(); ++k)
        if (re[k] > 16384 || re[k] < -16384 || im[k] > 16384 || im[k] < -16384) return false;
    if (re[0] != 16384 || im[0] != 0 || re[128] != 0 || im[128] != -16384) return false;   // stages 0-1 are mult-free
    std::vector<int> c8(64 * 8);
    for (int l = 0; l < 64; ++l)
        for (int m = 0; m < 8; ++m) c8[l * 8 + m] = cv[l + 64 * m];
    std::vector<uint32_t> r2(8 * 7 * 2), r3(64 * 7 * 2), rd(64 * 7 * 2);
    auto put_tw = [&](std::vector<uint32_t> &v, size_t at, const std::vector<int> &R, const std::vector<int> &I, int ta) {
        tw_pair(R[ta], I[ta], v[at], v[at + 1]);
    };
    // stage s, butterfly with low index bits j: ta = (j << (8 - s)) & 255   (fft.py:310-331)
    for (int lo3 = 0; lo3 < 8; ++lo3) {
        size_t at = size_t(lo3) * 14;
        put_tw(r2, at, re, im, (lo3 << 5) & 255);                                            // stage 3: j = lo3
        for (int b = 0; b < 2; ++b) put_tw(r2, at + 2 + 2 * b, re, im, (((b << 3) | lo3) << 4) & 255);   // stage 4
        for (int b = 0; b < 4; ++b) put_tw(r2, at + 6 + 2 * b, re, im, (((b << 3) | lo3) << 3) & 255);   // stage 5
    }
    for (int l = 0; l < 64; ++l) {
        size_t at = size_t(l) * 14;
        put_tw(r3, at, re, im, (l << 2) & 255);                                              // stage 6: j = lane
        for (int b = 0; b < 2; ++b) put_tw(r3, at + 2 + 2 * b, re, im, (((b << 6) | l) << 1) & 255);     // stage 7
        for (int b = 0; b < 4; ++b) put_tw(r3, at + 6 + 2 * b, re, im, ((b << 6) | l) & 255);            // stage 8
    }
    tw_pair(re[64], im[64], tw_s2[0], tw_s2[1]);
    tw_pair(re[192], im[192], tw_s2[2], tw_s2[3]);
    // DCT: 128-point FFT, stage s of the element at 64 + e: j = e & (2^s - 1), ta = (j << (6 - s)) & 63
    std::vector<int> dr, di;
    fx_twiddles(4 * kMel, dr, di);                    // 64 entries
    for (int e = 0; e < 64; ++e) {
        size_t at = size_t(e) * 14;
        for (int s = 0; s < 6; ++s) put_tw(rd, at + 2 * s, dr, di, ((e & ((1 << s) - 1)) << (6 - s)) & 63);
        put_tw(rd, at + 12, dr, di, e & 63);          // last stage: i0 = e, ta = e
    }
    auto put = [&](const void *p, size_t n) {
        size_t off = blob.size();
        blob.resize(off + n);
        std::memcpy(blob.data() + off, p, n);
    };
    blob.clear();
    put(c8.data(), c8.size() * 4);
    put(r2.data(), r2.size() * 4);
    put(r3.data(), r3.size() * 4);
    put(rd.data(), rd.size() * 4);
    return true;
}

inline void bind_tables(const char *b, Tables &t) {
    t.curve8 = reinterpret_cast<const int *>(b);                 b += 64 * 8 * 4;
    t.tw_r2 = reinterpret_cast<const uint32_t *>(b);             b += 8 * 14 * 4;
    t.tw_r3 = reinterpret_cast<const uint32_t *>(b);             b += 64 * 14 * 4;
    t.tw_dct = reinterpret_cast<const uint32_t *>(b);
}

// ---- device

__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the packed butterfly (12 VALU ops) lives in kernels_generic.hpp: fx_combine / fx_rot14 / fx_bfly
__device__ __forceinline__ void combine(uint32_t p0, int a1, int a2, uint32_t &o0, uint32_t &o1) {
    mfcc_k::fx_combine(p0, a1, a2, o0, o1);
}
__device__ __forceinline__ int rot14(uint32_t p1, uint32_t tw) { return mfcc_k::fx_rot14(p1, tw); }
__device__ __forceinline__ void bfly(uint32_t &p0, uint32_t &p1, uint32_t twa, uint32_t twb) {
    mfcc_k::fx_bfly(p0, p1, twa, twb);
}

// twiddle T[0] = (16384, 0): (x * 16384 + 8191) >> 14 == x
__device__ __forceinline__ void bfly_one(uint32_t &p0, uint32_t &p1) {
    combine(p0, (int)(short)(p1 & 0xffffu), (int)p1 >> 16, p0, p1);
}

// twiddle T[size/4] = (0, -16384): a1 = x1i, a2 = (-16384 x1r + 8191) >> 14 == -x1r
__device__ __forceinline__ void bfly_mi(uint32_t &p0, uint32_t &p1) {
    combine(p0, (int)p1 >> 16, -(int)(short)(p1 & 0xffffu), p0, p1);
}

// three radix-2 stages on the 8 values of a lane (register index = the three index bits of the round);
// tw: 7 operand pairs -- 1 for the first stage, 2 for the second (by register bit 0), 4 for the third
template <bool LAST>
__device__ __forceinline__ void round3(uint32_t (&x)[8], const uint32_t (&tw)[14]) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) bfly(x[r], x[r + 1], tw[0], tw[1]);
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (!(r & 2)) bfly(x[r], x[r + 2], tw[2 + 2 * (r & 1)], tw[3 + 2 * (r & 1)]);
#pragma unroll
    for (int r = 0; r < 4; ++r) bfly(x[r], x[r + 4], tw[6 + 2 * r], tw[7 + 2 * r]);   // LAST: x[r+4] is not read out
}

struct FrameCursor {
    int ch;
    long long f;
};

// Where a frame's samples are: p = the history sample n0 - 1, mis = samples between the 16-byte
// boundary below p and p.  fast: the 65 aligned pieces from that boundary lie inside the channel, so the
// frame can be fetched with one 16-byte load per lane (+1) -- otherwise (stream start without history,
// zero-padded tail) it is read sample by sample with the stream's edge rules.
struct FrameGeom {
    const int16_t *base;
    long long n0;
    int mis;
    bool fast;
};

__device__ __forceinline__ FrameGeom geom_of(const mfcc_k::StreamDesc &s, const FrameCursor &c) {
    FrameGeom q;
    q.base = s.pcm + (long long)c.ch * s.ch_stride;
    q.n0 = c.f * (long long)s.hop;
    q.mis = (int)((reinterpret_cast<uintptr_t>(q.base + q.n0 - 1) & 15) >> 1);
    const long long lo = q.n0 - 1 - q.mis;
    q.fast = lo >= -(long long)s.halo && lo + 65 * 8 <= s.n_samples;
    return q;
}

struct Geom {
    long long frames_per_ch;
    int n_ch;
    long long step_f;     // frames per stride of all waves, modulo frames_per_ch
    int step_ch;
};

__global__ __launch_bounds__(64 * kWaves)
void mfcc_fixed512_kernel(mfcc_k::StreamDesc s, Tables t, Geom g, int16_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t xbuf[kWaves][kXWords];     // gather / transposes / power
    __shared__ __attribute__((aligned(16))) uint32_t rawbuf[kWaves][kRawWords];
    __shared__ uint32_t melw[kMelWMax];
    __shared__ int melv[kWaves][kMel];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *X = xbuf[wave];
    uint32_t *Rb = rawbuf[wave];

    // read-only tables: filterbank weights in LDS, the rest in registers for the whole kernel
    for (int i = tid; i < t.mel_w_total; i += 64 * kWaves) melw[i] = t.mel_w[i];
    int curve[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) curve[m] = t.curve8[lane * 8 + m];
    uint32_t tw2[14], tw3[14], twd[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        tw2[i] = t.tw_r2[(lane & 7) * 14 + i];
        tw3[i] = t.tw_r3[lane * 14 + i];
        twd[i] = t.tw_dct[lane * 14 + i];
    }
    // filter (lane >> 1), half (lane & 1) of its bins
    const int filt = lane >> 1;
    const int m_start = t.mel_start[filt], m_count = t.mel_count[filt], m_off = t.mel_off[filt];
    const int br6 = (int)(__brev((unsigned)lane) >> 26);
    __syncthreads();

    // first frame of this wave; then strides of (all waves of the grid)
    const long long wid = (long long)blockIdx.x * kWaves + wave;
    FrameCursor c;
    c.ch = (int)(wid / g.frames_per_ch);
    c.f = wid - (long long)c.ch * g.frames_per_ch;

    // the next frame's samples are fetched one frame ahead: 16 bytes per lane, lane 63 also takes piece 64
    // piece 64 is fetched by lane 63 through an index the compiler cannot see through: a provably uniform
    // address would be turned into s_load_dwordx4, whose base address must be dword aligned -- ours
    // is only 2-byte aligned plus an odd offset, and SMEM drops the low bits of the base
    int lane_op = lane;
    asm volatile("" : "+v"(lane_op));
    // the next frame's samples are fetched one frame ahead: 16 bytes per lane, lane 63 also takes piece 64
    uint4 raw = make_uint4(0, 0, 0, 0), raw64 = raw;
    FrameGeom q = geom_of(s, c);
    if (c.ch < g.n_ch && q.fast) {
        const uint4 *src = reinterpret_cast<const uint4 *>(q.base + q.n0 - 1 - q.mis);
        raw = src[lane];
        if (lane == 63) raw64 = src[lane_op + 1];
    }
    i16_alias *const R16 = reinterpret_cast<i16_alias *>(Rb);

    while (c.ch < g.n_ch) {
        FrameCursor nc = c;
        nc.f += g.step_f;
        nc.ch += g.step_ch;
        if (nc.f >= g.frames_per_ch) {
            nc.f -= g.frames_per_ch;
            ++nc.ch;
        }
        // ---- this frame's samples n0 - 1 .. n0 + 511 into the LDS staging buffer, sample n0 - 1 at slot `first`
        int first;
        if (q.fast) {
            reinterpret_cast<uint4 *>(Rb)[lane] = raw;
            if (lane == 63) reinterpret_cast<uint4 *>(Rb)[64] = raw64;
            first = q.mis;
        } else {
            // stream start without history / zero-padded tail: sample by sample with the stream's edge rules
            for (int k = lane; k < kNfft + 1; k += 64) R16[k] = (int16_t)mfcc_k::sample_at_i(s, q.base, q.n0 - 1 + k);
            first = 0;
        }
        wave_fence();
        // ---- the next frame's samples start their way from HBM
        const bool more = nc.ch < g.n_ch;
        FrameGeom qn = geom_of(s, nc);
        if (more && qn.fast) {
            const uint4 *src = reinterpret_cast<const uint4 *>(qn.base + qn.n0 - 1 - qn.mis);
            raw = src[lane];
            if (lane == 63) raw64 = src[lane_op + 1];
        }

        // ---- pre-emphasis and window of sample a = lane + 64 m, parked as int16 for the bit-reversed gather
        {
            uint16_t *W = reinterpret_cast<uint16_t *>(X);
            const i16_alias *r = R16 + first + lane;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int o = r[64 * m], x0 = r[64 * m + 1];
                const int y = (int)(short)((x0 + (o >> 5) - o) & 0xffff);       // preemph.py:24
                W[lane + 64 * m] = (uint16_t)(__mul24(y, curve[m]) >> 9);       // window.py:84
            }
        }
        wave_fence();

        // ---- FFT 512.  Bit-reversed load (fft.py:413-424): element i = 8 lane + r holds sample bitrev9(i)
        uint32_t x[8];
        {
            const uint16_t *W = reinterpret_cast<const uint16_t *>(X);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int br3 = ((r & 1) << 2) | (r & 2) | (r >> 2);
                x[r] = W[(br3 << 6) + br6];                                     // imag = 0
            }
        }
        wave_fence();
        // round 1: stages 0, 1, 2 on index bits 0, 1, 2; twiddle index (j << (8 - s)): T[0], T[128], T[64], T[192]
#pragma unroll
        for (int r = 0; r < 8; r += 2) bfly_one(x[r], x[r + 1]);
        bfly_one(x[0], x[2]); bfly_mi(x[1], x[3]); bfly_one(x[4], x[6]); bfly_mi(x[5], x[7]);
        bfly_one(x[0], x[4]); bfly(x[1], x[5], t.tw64a, t.tw64b); bfly_mi(x[2], x[6]); bfly(x[3], x[7], t.tw192a, t.tw192b);
        // transpose 1: index i at i + 8 (i >> 6); lane (hi3, lo3) takes i = 64 hi3 + 8 r + lo3
        {
            uint32_t *w = X + lane * 8 + (lane >> 3) * 8;
            *reinterpret_cast<uint4 *>(w) = make_uint4(x[0], x[1], x[2], x[3]);
            *reinterpret_cast<uint4 *>(w + 4) = make_uint4(x[4], x[5], x[6], x[7]);
            wave_fence();
            const uint32_t *rd = X + (lane >> 3) * 72 + (lane & 7);
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = rd[8 * r];
            wave_fence();
        }
        round3<false>(x, tw2);
        // transpose 2: lane takes i = 64 r + lane
        {
            uint32_t *w = X + (lane >> 3) * 72 + (lane & 7);
#pragma unroll
            for (int r = 0; r < 8; ++r) w[8 * r] = x[r];
            wave_fence();
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = X[72 * r + lane];
            wave_fence();
        }
        round3<true>(x, tw3);

        // ---- power (pow2.py:32,64) of bins lane + 64 c, c = 0..3, to LDS for the filterbank
#pragma unroll
        for (int cbin = 0; cbin < 4; ++cbin) {
            const uint32_t p = (uint32_t)__builtin_amdgcn_sdot2(as_s16x2(x[cbin]), as_s16x2(x[cbin]), 0, false);
            X[lane + 64 * cbin] = p >> 2;
        }
        wave_fence();

        // ---- filterbank, closed form (tables.hpp: fx_mel): two lanes per filter, then log2 (log.py:33-102)
        {
            unsigned long long acc = 0;
            // four bins per trip, so that their LDS reads are in flight together
            for (int j = (lane & 1); j < m_count; j += 8) {
                unsigned long long part = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + 2 * u;
                    const uint32_t w = jj < m_count ? melw[m_off + jj] : 0u;
                    part += (unsigned long long)X[m_start + jj] * (unsigned long long)w;
                }
                acc += part;
            }
            const unsigned lo = (unsigned)acc, hi = (unsigned)(acc >> 32);
            const unsigned lo2 = (unsigned)__shfl_xor((int)lo, 1), hi2 = (unsigned)__shfl_xor((int)hi, 1);
            acc += ((unsigned long long)hi2 << 32) | lo2;
            const unsigned v = (unsigned)(acc >> t.mel_shift) & 0xFFFFu;
            const unsigned v1 = v ? v : 1u;
            const int msb = 31 - __clz((int)v1);
            unsigned z = (v1 << 11) >> msb;                 // the shift-right loop of log.py:57-62 in one step
            unsigned o = (unsigned)msb << 11;
#pragma unroll
            for (int cc = 0; cc < 10; ++cc) {
                const unsigned q = __umul24(z, z);
                const unsigned bit = (q >> 23) & 1u;
                z = q >> (11 + bit);
                o += bit << (10 - cc);
            }
            if (!(lane & 1)) melv[wave][filt] = (int)(o & 0x7FFFu);
        }
        wave_fence();

        // ---- DCT (dct_stream.py:23-33): 128-point FFT of y[2n+1] = y[127-2n] = x[n].  Natural index 2m+1 sits
        // at bit-reversed 64 + bitrev6(m): the upper half; element e = lane holds u[bitrev6(lane)],
        // u[m] = x[m] (m < 32), x[63 - m] (m >= 32)
        {
            const int m = br6;
            uint32_t v = (uint32_t)melv[wave][m < 32 ? m : 63 - m] & 0xffffu;
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                const uint32_t other = (uint32_t)__shfl_xor((int)v, 1 << st);
                const bool up = (lane >> st) & 1;             // this lane holds x1 of its butterfly
                uint32_t p0 = up ? other : v, p1 = up ? v : other;
                if (st == 0) bfly_one(p0, p1);
                else bfly(p0, p1, twd[2 * st], twd[2 * st + 1]);
                v = up ? p1 : p0;
            }
            // last stage: x0 = 0 (lower half), x1 = v, twiddle index lane; Re of y0 only
            const int a1 = rot14(v, twd[12]);
            if (lane < t.n_cep)
                out[((long long)c.ch * g.frames_per_ch + c.f) * t.n_cep + lane] = (int16_t)(a1 >> 1);
        }
        wave_fence();

        c = nc;
        q = qn;
    }
}

inline const char *kernel_name() { return "mfcc_fixed512_kernel"; }

inline void launch(const mfcc_k::StreamDesc &s, const Tables &t, int16_t *out, int n_cu, hipStream_t stream) {
    const long long total = s.total_frames;
    long long blocks = (total + kWaves - 1) / kWaves;
    const long long cap = (long long)n_cu * 4;          // 16 waves per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    Geom g;
    g.frames_per_ch = s.frames_per_ch;
    g.n_ch = (int)(total / s.frames_per_ch);
    const long long stride = blocks * kWaves;
    g.step_ch = (int)(stride / s.frames_per_ch);
    g.step_f = stride % s.frames_per_ch;
    hipLaunchKernelGGL(mfcc_fixed512_kernel, dim3((unsigned)blocks), dim3(64 * kWaves), 0, stream, s, t, g, out);
}

}  // namespace mfcc_fixed512
