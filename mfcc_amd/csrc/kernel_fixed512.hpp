// Fused fixed-point kernel for gfx950 (MI355X): the RTL arithmetic of mfcc/core for its own
// configuration -- MFCC(width=16, nfft=512, nfilters=32), mfcc/core/mfcc.py:20-88, and for the constructor's default
// of 16 filters (mfcc.py:20-21) -- bit for bit:
// pre-emphasis (preemph.py:24) -> window curve (window.py:84) -> 512-point radix-2 DIT FFT with Q14
// twiddles, +8191 >>14 and a >>1 per stage with 16-bit wrap (misc/fft.py:93-96,140-192) ->
// (re^2+im^2)>>2 (pow2.py:32,64) -> filterbank in closed form (filterbank.py:88-142, tables.hpp:
// fx_mel) -> Turner log2 Q4.11 (log.py:33-102) -> 128-point fixed FFT as DCT (dct_stream.py:23-44)
// -> first n_cep int16 (misc/discard.py).  Other parameter sets take mfcc_fixed_kernel.
//
// One frame per wave, four waves per workgroup, nothing shared between waves but read-only tables.  The tables a
// lane needs once per frame or less (round-3 twiddles, filterbank weights, DCT twiddles) are staged in LDS per
// workgroup instead of held in registers: 79 VGPRs, six waves per SIMD.  The grid is many times what is resident
// (96 workgroups per CU): see launch().
// Every butterfly is the RTL's own (same products, same bias, same shifts, same wraps); what changes
// is where the data lives:
//
//  * a complex value is one dword, (re, im) as two int16 -- every stage wraps to 16 bits anyway;
//  * s1 = x1r*twr - x1i*twi + 8191 and s2 = x1r*twi + x1i*twr + 8191 are one v_dot2_i32_i16 each on
//    the packed value (the RTL's three-multiplier form (x1r+x1i)*twr - x1i*(twr+twi) is the same
//    integer, it never overflows 33 bits: SURVEY.md A.4);
//  * the 9 stages run as three rounds of three stages on 8 register-resident values per lane (index
//    bits 0-2, 3-5, 6-8), with two transposes through LDS instead of nine stage round trips; the
//    twiddles of a round depend on the lane only (round 2's stay in registers, round 3's come from LDS);
//  * the bit-reversed load is free: lane l computes pre-emphasis and window of the samples
//    bitrev6(l) + 64 bitrev3(r) themselves, straight from the staged raw samples;
//  * the last stage computes only the outputs that are read out (bins 0..255);
//  * the filterbank's products (two per bin, 490 at 16 kHz) are dealt out evenly: a filter's row is cut
//    into pieces of at most `chunk` bins, one piece per lane (63 lanes at chunk 10), and the pieces of a
//    filter are added up with a segmented shuffle reduction -- 64-bit integer sums, any order is exact;
//  * in the DCT's 128-point FFT every even input is 0, so the lower half of the bit-reversed array
//    stays 0 until the last stage: stages 0-5 are a 64-point FFT of the upper half and the last stage
//    is one rotation per output;
//  * log2 and that 64-point FFT run once per PAIR of frames (16 filters: a 32-point FFT, once per FOUR frames): a
//    frame's filterbank sums wait in LDS for the next frame's, then each half (quarter) of the wave takes one frame -- one log2 per lane, and the FFT
//    with two values per lane: stage 0 pairs them in the lane, each later stage swaps one value with the
//    lane 2^(st-1) away (ds_swizzle) so that the butterfly is again inside the lane.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "kernels_generic.hpp"
#include "tables.hpp"

namespace mfcc_fixed512 {

constexpr int kNfft = 512, kMel = 32, kWaves = 4;   // kMel: the larger of the two filter counts (32, 16)
constexpr int kXWords = 512 + 64;        // transpose buffer: index i lives at i + 8 (i >> 6)
constexpr int kRawWords = 65 * 4;        // raw-sample staging: 65 aligned 16-byte pieces cover a frame + history

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int16_t __attribute__((may_alias)) i16_alias;     // LDS staged as 16-byte pieces, read back as samples

struct Tables {
    const int *curve8;        // [64 lanes][8]   window curve of sample bitrev6(lane) + 64 bitrev3(r)
    const uint32_t *tw_r2;    // [8 lo3][7][2]   round-2 twiddles (stages 3, 4, 5) as dot2 operand pairs
    const uint32_t *tw_r3;    // [64 lanes][7][2] round-3 twiddles (stages 6, 7, 8)
    const uint32_t *tw_dct;   // [n_mel lanes][16] DCT FFT, two values per lane: stages 1.. (pairs), last-stage rotations, indices
    uint32_t tw64a, tw64b, tw192a, tw192b;   // stage-2 twiddles T[64], T[192]
    const int4 *mel_lane;     // [64 lanes] piece of a filter row: first bin, -, (filter | head << 8), last lane of the filter
    const uint32_t *mel_wl;   // [chunk][64 lanes] the piece's weights (x 2^-30), 0 past its end
    int mel_chunk, mel_span;  // bins per piece (kMelChunk or less); lanes of the widest filter - 1
    int mel_shift, n_cep, n_mel;
};

inline bool supported(int nfft, int n_mel, int n_cep) {
    return nfft == kNfft && (n_mel == 32 || n_mel == 16) && n_cep >= 1 && n_cep <= n_mel;
}

// dot2 operand pair of a twiddle: A = (twr, -twi) gives s1, B = (twi, twr) gives s2
inline void tw_pair(int re, int im, uint32_t &a, uint32_t &b) {
    a = (uint32_t(re) & 0xffffu) | (uint32_t(-im) << 16);
    b = (uint32_t(im) & 0xffffu) | (uint32_t(re) << 16);
}

// host: tables in the order the kernel consumes them.  blob layout: curve8 | tw_r2 | tw_r3 | tw_dct
inline bool build_tables(int n_mel, std::vector<char> &blob, uint32_t (&tw_s2)[4]) {
    using namespace mfcc_tables;
    std::vector<int> cv = fx_window_curve(kNfft);
    std::vector<int> re, im;
    fx_twiddles(kNfft, re, im);                       // 256 entries
    // every twiddle must fit the int16 halves of a dot2 operand, negated too
    for (size_t k = 0; k < re.size(); ++k)
        if (re[k] > 16384 || re[k] < -16384 || im[k] > 16384 || im[k] < -16384) return false;
    if (re[0] != 16384 || im[0] != 0 || re[128] != 0 || im[128] != -16384) return false;   // stages 0-1 are mult-free
    std::vector<int> c8(64 * 8);
    for (int l = 0; l < 64; ++l) {
        int b6 = 0;
        for (int k = 0; k < 6; ++k) b6 |= ((l >> k) & 1) << (5 - k);
        for (int r = 0; r < 8; ++r) c8[l * 8 + r] = cv[b6 + 64 * (((r & 1) << 2) | (r & 2) | (r >> 2))];
    }
    std::vector<uint32_t> r2(8 * 7 * 2), r3(64 * 7 * 2), rd(32 * 16, 0u);
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
    // DCT: the (4 n_mel)-point FFT (fft.py:310-331 again), upper half only: H = 2 n_mel elements, LH = log2 H stages
    // inside the half.  Element e of that half (array index H + e) starts as u[bitrev_LH(e)], u[m] = x[m] (m < n_mel),
    // x[H - 1 - m] (m >= n_mel).  Lane l (of n_mel) starts with elements 2l, 2l + 1; before stage st >= 1 the lanes l
    // and l ^ 2^(st-1) swap one value (the lower lane its second, the upper its first), so that every lane again holds a
    // butterfly's x0 and x1.  Stage s of element e: j = e mod 2^s, ta = (j << (LH - s)) & (H - 1).
    // Per lane: [2(st-1)], [2(st-1)+1] operand pair of stage st; [10], [11] last-stage rotation (ta = e) of the two
    // final elements; [12] = load index 0 | load index 1 << 8 | final element 0 << 16 | final element 1 << 24.
    std::vector<int> dr, di;
    fx_twiddles(4 * n_mel, dr, di);                   // H entries
    {
        const int H = 2 * n_mel, LH = n_mel == 32 ? 6 : 5;
        int el[32][2];
        auto brv = [&](int e) { int b = 0; for (int k = 0; k < LH; ++k) b |= ((e >> k) & 1) << (LH - 1 - k); return b; };
        for (int l = 0; l < n_mel; ++l) {
            el[l][0] = 2 * l; el[l][1] = 2 * l + 1;
            const int m0 = brv(2 * l), m1 = brv(2 * l + 1);
            rd[l * 16 + 12] = uint32_t(m0 < n_mel ? m0 : H - 1 - m0) | uint32_t(m1 < n_mel ? m1 : H - 1 - m1) << 8;
        }
        if (dr[0] != 16384 || di[0] != 0) return false;       // stage 0 is mult-free
        for (int st = 1; st < LH; ++st) {
            const int mask = 1 << (st - 1);
            int nx[32][2];
            for (int l = 0; l < n_mel; ++l) {
                const int pl = l ^ mask;
                if (!(l & mask)) { nx[l][0] = el[l][0]; nx[l][1] = el[pl][0]; }
                else             { nx[l][0] = el[pl][1]; nx[l][1] = el[l][1]; }
                if ((nx[l][0] & (1 << st)) || nx[l][1] != (nx[l][0] | (1 << st))) return false;
                put_tw(rd, size_t(l) * 16 + 2 * (st - 1), dr, di, ((nx[l][0] & ((1 << st) - 1)) << (LH - st)) & (H - 1));
            }
            for (int l = 0; l < n_mel; ++l) { el[l][0] = nx[l][0]; el[l][1] = nx[l][1]; }
        }
        for (int l = 0; l < n_mel; ++l) {
            uint32_t a, b;
            tw_pair(dr[el[l][0] & (H - 1)], di[el[l][0] & (H - 1)], a, b); rd[l * 16 + 10] = a;
            tw_pair(dr[el[l][1] & (H - 1)], di[el[l][1] & (H - 1)], a, b); rd[l * 16 + 11] = a;
            rd[l * 16 + 12] |= uint32_t(el[l][0]) << 16 | uint32_t(el[l][1]) << 24;
        }
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

// host: the filterbank rows (tables.hpp: pack_rows) cut into pieces of `chunk` bins, one per lane -- the smallest
// chunk that fits 64 lanes.  lanes: 4 ints per lane, see Tables::mel_lane; wl: the weights, [kMelChunk][64].
constexpr int kMelChunk = 12;            // bins per lane the kernel is unrolled for (10 are needed at 16 kHz)
constexpr int kMelSpanMax = 15;          // four doubling steps of the segmented reduction (16 filters: up to 11 lanes per filter)
inline bool build_mel_lanes(const std::vector<int> &start, const std::vector<int> &count, const std::vector<int> &off,
                            const std::vector<uint32_t> &w, std::vector<int> &lanes, std::vector<uint32_t> &wl,
                            int &chunk, int &span) {
    const int nf = (int)start.size();
    for (chunk = 1; chunk <= kMelChunk; ++chunk) {
        int need = 0;
        for (int f = 0; f < nf; ++f) need += count[f] > 0 ? (count[f] + chunk - 1) / chunk : 1;
        if (need <= 64) break;
    }
    if (chunk > kMelChunk) return false;
    span = 0;
    for (int f = 0; f < nf; ++f) {
        const int k = count[f] > 0 ? (count[f] + chunk - 1) / chunk : 1;
        if (k - 1 > span) span = k - 1;
    }
    if (span > kMelSpanMax) return false;
    // Which lane takes which piece, and where its reads begin.  Lane l reads the power values X[first_l + u], u = 0 ..
    // taps - 1 (taps = chunk rounded up to even: the kernel's loop runs in pairs), one ds_read_b32 per u over all 64 lanes,
    // i.e. two groups of 32 lanes with 32 banks each: the cost of every one of those reads is the largest number of
    // DIFFERENT addresses on one bank, which only depends on the residues first_l mod 32 inside each half of the wave.
    // With the pieces in filter order that was 3 + 3 LDS cycles per read where 1 + 1 is free of conflicts (round 2's
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 19 % came from here).  Two freedoms cost nothing: the ORDER of the
    // filters over the lanes (a filter's pieces must stay in consecutive lanes for the segmented reduction, nothing else),
    // and a piece shorter than `taps` may BEGIN up to taps - len bins early against zero weights.  A few thousand random
    // orders with greedy shifts find 1 + 2 or better for every rate tried; the search is deterministic (fixed seed).
    const int taps = (chunk + 1) & ~1;
    struct Piece { int first, len, filt, head, last_lane_rel; };
    std::vector<std::vector<Piece>> P(nf);
    for (int f = 0; f < nf; ++f) {
        const int k = count[f] > 0 ? (count[f] + chunk - 1) / chunk : 1;
        for (int i = 0; i < k; ++i) {
            int n = count[f] - i * chunk;
            n = n < 0 ? 0 : (n > chunk ? chunk : n);
            P[f].push_back({start[f] + i * chunk, n, f, i == 0, k - 1 - i});
        }
    }
    std::vector<int> order(nf), best_order, best_first;
    for (int f = 0; f < nf; ++f) order[f] = f;
    int best_cost = 1 << 30;
    uint32_t rng = 12345u;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    for (int iter = 0; iter < 4000 && best_cost > 2; ++iter) {
        if (iter) for (int i = nf - 1; i > 0; --i) std::swap(order[i], order[next() % (uint32_t)(i + 1)]);
        std::vector<int> first;
        std::vector<int> addr[2][32];                 // distinct addresses per (half, bank)
        int l = 0;
        for (int oi = 0; oi < nf; ++oi)
            for (const Piece &p : P[order[oi]]) {
                const int half = l < 32 ? 0 : 1;
                int pick = p.first, pick_mult = 1 << 30;
                for (int d = 0; d <= taps - p.len && p.first - d >= 0; ++d) {
                    const int a = p.first - d;
                    int mult = 0;
                    bool same = false;
                    for (int o : addr[half][a & 31]) { if (o == a) same = true; else ++mult; }
                    (void)same;                       // an address another lane already reads is a broadcast: free
                    if (mult < pick_mult) { pick_mult = mult; pick = a; }
                    if (mult == 0) break;
                }
                bool have = false;
                for (int o : addr[half][pick & 31]) have = have || o == pick;
                if (!have) addr[half][pick & 31].push_back(pick);
                first.push_back(pick);
                ++l;
            }
        int cost = 0;
        for (int half = 0; half < 2; ++half) {
            size_t m = 1;
            for (int b = 0; b < 32; ++b) m = std::max(m, addr[half][b].size());
            cost += (int)m;
        }
        if (cost < best_cost) { best_cost = cost; best_order = order; best_first = first; }
    }
    lanes.assign(64 * 4, 0);
    wl.assign(size_t(kMelChunk) * 64, 0u);
    int l = 0;
    for (int oi = 0; oi < nf; ++oi)
        for (const Piece &p : P[best_order[oi]]) {
            const int d = p.first - best_first[l];         // bins the lane reads before its piece: weight 0
            lanes[l * 4 + 0] = best_first[l];
            lanes[l * 4 + 2] = p.filt | (p.head << 8);
            lanes[l * 4 + 3] = l + p.last_lane_rel;
            for (int u = 0; u < p.len; ++u) wl[size_t(d + u) * 64 + l] = w[off[p.filt] + (p.first - start[p.filt]) + u];
            ++l;
        }
    for (; l < 64; ++l) lanes[l * 4 + 3] = l;         // spare lanes: no bins, no filter, a segment of their own
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

// the packed butterfly (9 VALU ops) lives in kernels_generic.hpp: fx_bfly (+ fx_combine / fx_rot14 for the odd forms)
__device__ __forceinline__ void combine(uint32_t p0, int a1, int a2, uint32_t &o0, uint32_t &o1) {
    mfcc_k::fx_combine(p0, a1, a2, o0, o1);
}
__device__ __forceinline__ int rot14(uint32_t p1, uint32_t tw) { return mfcc_k::fx_rot14(p1, tw); }
__device__ __forceinline__ void bfly(uint32_t &p0, uint32_t &p1, uint32_t twa, uint32_t twb) {
    mfcc_k::fx_bfly(p0, p1, twa, twb);
}

// twiddle T[0] = (16384, 0): (x * 16384 + 8191) >> 14 == x
__device__ __forceinline__ void bfly_one(uint32_t &p0, uint32_t &p1) { mfcc_k::fx_bfly_one(p0, p1); }

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
    for (int r = 0; r < 4; ++r) {
        if constexpr (LAST) mfcc_k::fx_bfly_y0(x[r], x[r + 4], tw[6 + 2 * r], tw[7 + 2 * r]);   // x[r+4] is not read out
        else bfly(x[r], x[r + 4], tw[6 + 2 * r], tw[7 + 2 * r]);
    }
}

// stage ST >= 1 of the DCT's 64-point FFT on two values per lane: swap one value with the lane 2^(ST-1) away
// (ds_swizzle in bit-mask mode: xor inside each half of the wave), then the butterfly is inside the lane
template <int ST>
__device__ __forceinline__ void dct_stage(uint32_t &s0, uint32_t &s1, const uint32_t (&twd)[13], int lane) {
    const bool lower = !(lane & (1 << (ST - 1)));
    const uint32_t send = lower ? s1 : s0;
    const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_swizzle((int)send, ((1 << (ST - 1)) << 10) | 0x1f);
    s0 = lower ? s0 : recv;
    s1 = lower ? recv : s1;
    bfly(s0, s1, twd[2 * (ST - 1)], twd[2 * (ST - 1) + 1]);
}

// One utterance of a ragged corpus (mfcc_hip_process_ragged_*): an independent stream that starts from reset and is
// zero-padded at its end, exactly like a channel of the plain call.  Only utterances with frames > 0 get a record.
struct RaggedRec {
    long long pcm_off;       // first sample, relative to StreamDesc::pcm
    long long out_row;       // first output row (frame) of the utterance == frames of all utterances before it
    int n_samples, frames;
    int pad0, pad1;
};

struct Geom {
    long long frames_per_ch;          // plain
    int n_ch;                         // plain: channels; ragged: utterances with frames
    long long step_f;                 // plain: frames per stride of all waves, modulo frames_per_ch
    int step_ch;
    int chunk;                        // ragged: consecutive frames per wave
    const RaggedRec *recs;            // nullptr: the plain multi-channel call
};

// the plain call's cursor.  The two walks keep their own few lines of cursor code: versions with ONE cursor type for
// both (copied as a whole per frame, or captured by a [&] lambda) had the compiler keep the cursors in scratch memory --
// 3.17 and 5.78 ms against 3.10 on config 3, with "VGPRs Spill: 0" in the remarks; ScratchSize is the line to read
struct StridedCursor {
    int ch;
    long long f;
};
struct FrameGeom;
__device__ __forceinline__ FrameGeom geom_of_strided(const mfcc_k::StreamDesc &s, const StridedCursor &c);

typedef const long long __attribute__((address_space(4))) ConstLL;
__device__ __forceinline__ const ConstLL *rec_ptr(const RaggedRec *recs, int u) {
    return reinterpret_cast<const ConstLL *>(reinterpret_cast<uintptr_t>(recs + u));
}

// A wave's position: the stream (channel / utterance) it is in and the frame there.  Kept small: the kernel holds two of
// them (this frame, the next one) in scalar registers; the stream's constants travel separately (StreamConst).
struct FrameCursor {
    int u, f;                // f < 2^31: a channel that fills the device's memory has < 2^30 frames
    const int16_t *base;     // the stream's first sample
    long long out_row;       // output row of the stream's frame 0
};
// what is the same for every frame of a stream (for every frame of the plain call)
struct StreamConst {
    long long n_samples;
    int frames, halo;
};

template <bool RUNS>
__device__ __forceinline__ void enter_stream(const mfcc_k::StreamDesc &s, const Geom &g, FrameCursor &c, StreamConst &k) {
    if (c.u >= g.n_ch) return;
    if constexpr (RUNS) {
        // read through the constant address space: the records are written before the launch and never by the kernel, and
        // only so does the compiler fetch them with scalar loads (as plain global memory it issued vector loads plus
        // v_readfirstlane and spilled for them)
        const ConstLL *r = rec_ptr(g.recs, c.u);
        c.base = s.pcm + r[0];
        c.out_row = r[1];
        const long long nf = r[2];                  // n_samples | frames << 32
        k.n_samples = (int)(nf & 0xffffffffll);
        k.frames = (int)(nf >> 32);
        k.halo = 0;
    }
}
// Where a frame's samples are: p = the history sample n0 - 1, mis = samples between the 16-byte
// boundary below p and p.  fast: the 65 aligned pieces from that boundary lie inside the channel, so the
// frame can be fetched with one 16-byte load per lane (+1) -- otherwise (stream start without history,
// zero-padded tail) it is read sample by sample with the stream's edge rules.
struct FrameGeom {
    const int16_t *base;     // the stream's first sample
    long long n0;
    int mis;
    bool fast;
};

__device__ __forceinline__ FrameGeom geom_of_strided(const mfcc_k::StreamDesc &s, const StridedCursor &c) {
    FrameGeom q;
    q.base = s.pcm + (long long)c.ch * s.ch_stride;
    q.n0 = c.f * (long long)s.hop;
    q.mis = (int)((reinterpret_cast<uintptr_t>(q.base + q.n0 - 1) & 15) >> 1);
    const long long lo = q.n0 - 1 - q.mis;
    q.fast = lo >= -(long long)s.halo && lo + 65 * 8 <= s.n_samples;
    return q;
}

__device__ __forceinline__ FrameGeom geom_of(const int16_t *base, const FrameCursor &c, const StreamConst &k, int hop) {
    FrameGeom q;
    q.base = base;
    q.n0 = c.f * (long long)hop;
    q.mis = (int)((reinterpret_cast<uintptr_t>(base + q.n0 - 1) & 15) >> 1);
    const long long lo = q.n0 - 1 - q.mis;
    q.fast = lo >= -(long long)k.halo && lo + 65 * 8 <= k.n_samples;
    return q;
}

// RUNS = false, the plain call: wave w takes the frames w, w + W, w + 2 W ... (W = all waves of the grid) -- at any
// moment the resident waves work on neighbouring frames.  RUNS = true, a ragged corpus: wave w takes `chunk` CONSECUTIVE
// frames, so that moving on is ++f with a rare step into the next utterance (a strided walk would have to search the
// utterance of every frame).  Same speed: 10 000 equal utterances as channels (strides) 7.99 ms, as a ragged corpus of
// five lengths (runs) 7.94 ms.
template <int MEL, bool RUNS>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(6, 6)))
void mfcc_fixed512_kernel(mfcc_k::StreamDesc s, Tables t, Geom g, int16_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t xbuf[kWaves][kXWords];     // gather / transposes / power
    __shared__ __attribute__((aligned(16))) uint32_t rawbuf[kWaves][kRawWords];
    constexpr int kPass = 64 / MEL;                             // frames per log2 + DCT pass: 2 (32 filters) or 4 (16)
    __shared__ int melv[kWaves][64];                            // [frame of the pass][filter]: filterbank sums, then their logs
    __shared__ long long rowsh[kWaves][4];                      // output rows of the frames of the pass
    // per-lane tables, [entry][lane]: conflict-free to read, fetched where they are used
    __shared__ uint32_t mwl[kMelChunk * 64];                    // filterbank weights of the lane's piece
    __shared__ uint32_t twdl[13 * 64];                          // DCT twiddles and indices (lanes 32..63 = 0..31)
    __shared__ uint32_t tw3l[14 * 64];                          // round-3 twiddles
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *X = xbuf[wave];
    uint32_t *Rb = rawbuf[wave];

    // read-only tables: the window curve and round 2's twiddles in registers, the rest staged in LDS here
    int curve[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) curve[m] = t.curve8[lane * 8 + m];
    for (int i = tid; i < kMelChunk * 64; i += 64 * kWaves) mwl[i] = t.mel_wl[i];
    for (int i = tid; i < 13 * 64; i += 64 * kWaves) twdl[i] = t.tw_dct[(i & (MEL - 1)) * 16 + (i >> 6)];
    for (int i = tid; i < 14 * 64; i += 64 * kWaves) tw3l[i] = t.tw_r3[(i & 63) * 14 + (i >> 6)];
    uint32_t tw2[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) tw2[i] = t.tw_r2[(lane & 7) * 14 + i];
    int slot = 0;                                   // frames waiting for the shared log2 + DCT pass
    const int4 ml = t.mel_lane[lane];               // this lane's piece of a filterbank row
    __syncthreads();
    const int br6 = (int)(__brev((unsigned)lane) >> 26);
    
    // XCD-aware order (kernel_fused512_w12.hpp): neighbouring workgroups, whose frames overlap, share an XCD's L2
    const unsigned nwg = gridDim.x;
    const unsigned bid = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    const long long wid = (long long)bid * kWaves + wave;
    // RUNS: this wave's run of consecutive frames; plain: its first frame, then strides of all waves of the grid
    FrameCursor c = {};
    StreamConst kc = {0, 0, 0};
    StridedCursor sc = {0, 0};
    int left = 0;
    if constexpr (RUNS) {
        const long long first_frame = wid * g.chunk;
        left = first_frame >= s.total_frames ? 0 : (int)(s.total_frames - first_frame < g.chunk ? s.total_frames - first_frame : g.chunk);
        if (left > 0) {                                 // the utterance that holds the frame: out_row <= first_frame
            int lo_u = 0, hi_u = g.n_ch - 1;
            while (lo_u < hi_u) {
                const int mid = (lo_u + hi_u + 1) >> 1;
                if (rec_ptr(g.recs, mid)[1] <= first_frame) lo_u = mid;
                else hi_u = mid - 1;
            }
            c.u = lo_u;
            enter_stream<RUNS>(s, g, c, kc);
            c.f = (int)(first_frame - c.out_row);
        } else {
            c.u = g.n_ch;
        }
    } else {
        sc.ch = (int)(wid / g.frames_per_ch);
        sc.f = wid - (long long)sc.ch * g.frames_per_ch;
    }

    // the next frame's samples are fetched one frame ahead: 16 bytes per lane, lane 63 also takes piece 64
    // piece 64 is fetched by lane 63 through an index the compiler cannot see through: a provably uniform
    // address would be turned into s_load_dwordx4, whose base address must be dword aligned -- ours
    // is only 2-byte aligned plus an odd offset, and SMEM drops the low bits of the base
    int lane_op = lane;
    asm volatile("" : "+v"(lane_op));
    // the next frame's samples are fetched one frame ahead: 16 bytes per lane, lane 63 also takes piece 64
    uint4 raw = make_uint4(0, 0, 0, 0), raw64 = raw;
    FrameGeom q;
    if constexpr (RUNS) q = geom_of(c.base, c, kc, s.hop);
    else q = geom_of_strided(s, sc);
    if ((RUNS ? c.u < g.n_ch : sc.ch < g.n_ch) && q.fast) {
        const uint4 *src = reinterpret_cast<const uint4 *>(q.base + q.n0 - 1 - q.mis);
        raw = src[lane];
        if (lane == 63) raw64 = src[lane_op + 1];
    }
    i16_alias *const R16 = reinterpret_cast<i16_alias *>(Rb);

    while (RUNS ? c.u < g.n_ch : sc.ch < g.n_ch) {
        FrameCursor nc;
        StreamConst kn;                                 // RUNS: the next frame's utterance (another one now and then)
        StridedCursor snc;
        bool more;
        if constexpr (RUNS) {
            nc = c;
            kn = kc;
            if (--left > 0) {
                if (++nc.f == kc.frames) {
                    nc.f = 0;
                    ++nc.u;
                    enter_stream<RUNS>(s, g, nc, kn);
                }
            } else {
                nc.u = g.n_ch;
            }
            more = nc.u < g.n_ch;
        } else {
            snc = sc;
            snc.f += g.step_f;
            snc.ch += g.step_ch;
            if (snc.f >= g.frames_per_ch) {
                snc.f -= g.frames_per_ch;
                ++snc.ch;
            }
            more = snc.ch < g.n_ch;
        }
        // ---- this frame's samples n0 - 1 .. n0 + 511 into the LDS staging buffer, sample n0 - 1 at slot `first`
        int first;
        if (q.fast) {
            reinterpret_cast<uint4 *>(Rb)[lane] = raw;
            if (lane == 63) reinterpret_cast<uint4 *>(Rb)[64] = raw64;
            first = q.mis;
        } else {
            // stream start without history / zero-padded tail: sample by sample with the stream's edge rules
            if constexpr (RUNS) {
                mfcc_k::StreamDesc sl = s;              // the edge rules of sample_at_i on the utterance's numbers
                sl.n_samples = kc.n_samples;
                sl.halo = kc.halo;
                for (int k = lane; k < kNfft + 1; k += 64) R16[k] = (int16_t)mfcc_k::sample_at_i(sl, q.base, q.n0 - 1 + k);
            } else {
                for (int k = lane; k < kNfft + 1; k += 64) R16[k] = (int16_t)mfcc_k::sample_at_i(s, q.base, q.n0 - 1 + k);
            }
            first = 0;
        }
        wave_fence();
        // ---- the next frame's samples start their way from HBM
        FrameGeom qn;
        if constexpr (RUNS) qn = geom_of(nc.base, nc, kn, s.hop);
        else qn = geom_of_strided(s, snc);
        if (more && qn.fast) {
            const uint4 *src = reinterpret_cast<const uint4 *>(qn.base + qn.n0 - 1 - qn.mis);
            raw = src[lane];
            if (lane == 63) raw64 = src[lane_op + 1];
        }

        // ---- pre-emphasis and window, already in the FFT's bit-reversed order (fft.py:413-424): element
        // i = 8 lane + r is sample bitrev9(i) = bitrev6(lane) + 64 bitrev3(r); imag = 0
        uint32_t x[8];
        {
            int w[8];
            const i16_alias *r16 = R16 + first + br6;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int m = ((r & 1) << 2) | (r & 2) | (r >> 2);
                const int o = r16[64 * m], x0 = r16[64 * m + 1];
                const int y = (int)(short)((x0 + (o >> 5) - o) & 0xffff);       // preemph.py:24
                w[r] = (int)(short)((__mul24(y, curve[r]) >> 9) & 0xffff);      // window.py:84
            }
            // round 1: stages 0, 1, 2 on index bits 0, 1, 2; twiddle index (j << (8 - s)): T[0], T[128], T[64], T[192].
            // The input is real and T[0] = (16384, 0), T[128] = (0, -16384) rotate exactly ((x 16384 + 8191) >> 14
            // == x, (-16384 x + 8191) >> 14 == -x), so the RTL's butterfly (fft.py:140-192) comes down to
            //   T[0]   on reals a, b:   ((a + b) >> 1, 0), ((a - b) >> 1, 0)       -- 16-bit a, b: no wrap can occur
            //   T[128] on reals a, b:   (a >> 1, (-b) >> 1), (a >> 1, b >> 1)
            // stage 0: four T[0] butterflies on reals
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                const int a = w[r], b = w[r + 1];
                w[r] = (a + b) >> 1;
                w[r + 1] = (a - b) >> 1;
            }
            // stage 1: (0,2), (4,6) T[0] on reals; (1,3), (5,7) T[128] on reals, complex from here on
            uint32_t c[8];
#pragma unroll
            for (int k = 0; k < 8; k += 4) {
                const int a = w[k], b = w[k + 2];
                w[k] = (a + b) >> 1;
                w[k + 2] = (a - b) >> 1;
                const int h1 = w[k + 1] >> 1, h3 = w[k + 3] >> 1, n3 = (-w[k + 3]) >> 1;
                c[k + 1] = __builtin_amdgcn_perm((uint32_t)n3, (uint32_t)h1, 0x05040100u);     // (re, im) = (h1, n3)
                c[k + 3] = __builtin_amdgcn_perm((uint32_t)h3, (uint32_t)h1, 0x05040100u);
            }
            // stage 2: (0,4) T[0] on reals, (2,6) T[128] on reals, (1,5) T[64] and (3,7) T[192] in full
            x[0] = (uint32_t)((w[0] + w[4]) >> 1) & 0xffffu;
            x[4] = (uint32_t)((w[0] - w[4]) >> 1) & 0xffffu;
            {
                const int h2 = w[2] >> 1, h6 = w[6] >> 1, n6 = (-w[6]) >> 1;
                x[2] = __builtin_amdgcn_perm((uint32_t)n6, (uint32_t)h2, 0x05040100u);
                x[6] = __builtin_amdgcn_perm((uint32_t)h6, (uint32_t)h2, 0x05040100u);
            }
            x[1] = c[1]; x[5] = c[5]; x[3] = c[3]; x[7] = c[7];
            bfly(x[1], x[5], t.tw64a, t.tw64b);
            bfly(x[3], x[7], t.tw192a, t.tw192b);
        }
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
        {
            uint32_t tw3[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) tw3[i] = tw3l[i * 64 + lane];
            round3<true>(x, tw3);
        }

        // ---- power (pow2.py:32,64) of bins lane + 64 c, c = 0..3, to LDS for the filterbank
#pragma unroll
        for (int cbin = 0; cbin < 4; ++cbin) {
            const uint32_t p = (uint32_t)__builtin_amdgcn_sdot2(as_s16x2(x[cbin]), as_s16x2(x[cbin]), 0, false);
            X[lane + 64 * cbin] = p >> 2;
        }
        wave_fence();

        // ---- filterbank, closed form (tables.hpp: fx_mel): one piece of a row per lane, pieces of a filter added
        // up by a segmented reduction, then log2 (log.py:33-102) on the filter's first lane
        {
            unsigned long long acc = 0;
            const uint32_t *xp = X + ml.x;
#pragma unroll
            for (int u = 0; u < kMelChunk; u += 2) {        // bins past the piece: weight 0
                if (u < t.mel_chunk)
                    acc += (unsigned long long)xp[u] * mwl[u * 64 + lane] + (unsigned long long)xp[u + 1] * mwl[(u + 1) * 64 + lane];
            }
#pragma unroll
            for (int d = 1; d <= kMelSpanMax; d <<= 1) {
                if (d <= t.mel_span) {
                    const unsigned lo2 = (unsigned)__shfl_down((int)(unsigned)acc, d), hi2 = (unsigned)__shfl_down((int)(unsigned)(acc >> 32), d);
                    if (lane + d <= ml.w) acc += ((unsigned long long)hi2 << 32) | lo2;
                }
            }
            if (ml.z >> 8) melv[wave][slot * MEL + (ml.z & 0xff)] = (int)((unsigned)(acc >> t.mel_shift) & 0xFFFFu);
        }
        if (lane == 0) rowsh[wave][slot] = RUNS ? c.out_row + c.f : (long long)sc.ch * g.frames_per_ch + sc.f;
        ++slot;
        wave_fence();

        if (slot == kPass || !more) {
            uint32_t twd[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) twd[i] = twdl[i * 64 + lane];
            // ---- log2 (log.py:33-102) of all the pass's sums: part h of the wave takes frame h
            const int h = lane / MEL, l5 = lane & (MEL - 1);
            int *mv = melv[wave] + h * MEL;
            {
                const unsigned v = (unsigned)mv[l5];
                const unsigned v1 = v ? v : 1u;
                const int msb = 31 - __clz((int)v1);
                unsigned z = (v1 << 11) >> msb;             // the shift-right loop of log.py:57-62 in one step
                unsigned o = (unsigned)msb << 11;
#pragma unroll
                for (int cc = 0; cc < 10; ++cc) {
                    const unsigned q2 = __umul24(z, z);
                    const unsigned bit = (q2 >> 23) & 1u;
                    z = q2 >> (11 + bit);
                    o += bit << (10 - cc);
                }
                mv[l5] = (int)(o & 0x7FFFu);
            }
            wave_fence();
            // ---- DCT (dct_stream.py:23-33): (4 MEL)-point FFT of y[2n+1] = y[4 MEL - 1 - 2n] = x[n]; see build_tables
            {
                uint32_t s0 = (uint32_t)mv[twd[12] & 0xff], s1 = (uint32_t)mv[(twd[12] >> 8) & 0xff];
                bfly_one(s0, s1);
                dct_stage<1>(s0, s1, twd, lane); dct_stage<2>(s0, s1, twd, lane); dct_stage<3>(s0, s1, twd, lane);
                dct_stage<4>(s0, s1, twd, lane);
                if constexpr (MEL == 32) dct_stage<5>(s0, s1, twd, lane);
                // last stage: x0 = 0 (lower half), x1 = the element, twiddle index = its number; Re of y0 only
                const int a0 = rot14(s0, twd[10]), a1 = rot14(s1, twd[11]);
                const int e0 = (twd[12] >> 16) & 0xff, e1 = twd[12] >> 24;
                if (h < slot) {
                    int16_t *o = out + rowsh[wave][h] * t.n_cep;
                    if (e0 < t.n_cep) o[e0] = (int16_t)(a0 >> 1);
                    if (e1 < t.n_cep) o[e1] = (int16_t)(a1 >> 1);
                }
            }
            slot = 0;
            wave_fence();
        }

        if constexpr (RUNS) {
            c = nc;
            kc = kn;
        } else {
            sc = snc;
        }
        q = qn;
    }
}

inline const char *kernel_name() { return "mfcc_fixed512_kernel"; }

inline void launch_impl(const mfcc_k::StreamDesc &s, const Tables &t, int16_t *out, int n_cu, hipStream_t stream,
                        const RaggedRec *recs, int n_recs, long long total) {
    long long blocks = (total + 4 * kWaves - 1) / (4 * kWaves);
    // Many more workgroups than are resident (6 per CU by registers and LDS).  Measured on config 3, kernel ms: the
    // exactly resident grid 3.57, 2 x 3.32, 4 x 3.19, 8 x 3.11, 16 x 3.08, 32 x 3.10, 64 x 3.17 -- and with round 2's
    // register-resident tables (4 waves per SIMD) 3.78 at 1 x, 3.20 at 16 x.  A grid that is merely full leaves slots
    // empty: workgroups are not placed evenly over the CUs, and one that waits for a slot runs after the others; with
    // a deep queue every slot is refilled the moment it frees up.  (The float kernels place ONE workgroup per CU, 256
    // of them: nothing to gain there, measured.)
    const long long cap = (long long)n_cu * 96;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    Geom g;
    g.frames_per_ch = s.frames_per_ch;
    g.n_ch = recs ? n_recs : (int)(total / s.frames_per_ch);
    g.recs = recs;
    if (recs) {
        // consecutive frames per wave, a multiple of the frames per log2 + DCT pass
        long long chunk = (total + blocks * kWaves - 1) / (blocks * kWaves);
        chunk = (chunk + 3) & ~3ll;
        g.chunk = (int)chunk;
        g.step_f = 0;
        g.step_ch = 0;
        blocks = (total + chunk * kWaves - 1) / (chunk * kWaves);
        if (t.n_mel == 16)
            hipLaunchKernelGGL((mfcc_fixed512_kernel<16, true>), dim3((unsigned)blocks), dim3(64 * kWaves), 0, stream, s, t, g, out);
        else
            hipLaunchKernelGGL((mfcc_fixed512_kernel<32, true>), dim3((unsigned)blocks), dim3(64 * kWaves), 0, stream, s, t, g, out);
    } else {
        blocks = (total + kWaves - 1) / kWaves < cap ? (total + kWaves - 1) / kWaves : cap;
        const long long stride = blocks * kWaves;
        g.chunk = 0;
        g.step_ch = (int)(stride / s.frames_per_ch);
        g.step_f = stride % s.frames_per_ch;
        if (t.n_mel == 16)
            hipLaunchKernelGGL((mfcc_fixed512_kernel<16, false>), dim3((unsigned)blocks), dim3(64 * kWaves), 0, stream, s, t, g, out);
        else
            hipLaunchKernelGGL((mfcc_fixed512_kernel<32, false>), dim3((unsigned)blocks), dim3(64 * kWaves), 0, stream, s, t, g, out);
    }
}

// the plain call: channels of one length
inline void launch(const mfcc_k::StreamDesc &s, const Tables &t, int16_t *out, int n_cu, hipStream_t stream) {
    launch_impl(s, t, out, n_cu, stream, nullptr, 0, s.total_frames);
}

// a ragged corpus straight out of the caller's buffer: d_recs (device) = one record per utterance WITH frames, in order
inline void launch_ragged(const int16_t *d_pcm, const RaggedRec *d_recs, int n_recs, long long total_frames, int hop,
                          const Tables &t, int16_t *out, int n_cu, hipStream_t stream) {
    mfcc_k::StreamDesc s;
    s.pcm = d_pcm;
    s.ch_stride = 0;
    s.n_samples = 0;
    s.halo = 0;
    s.frames_per_ch = 1;
    s.total_frames = total_frames;
    s.hop = hop;
    launch_impl(s, t, out, n_cu, stream, d_recs, n_recs, total_frames);
}

}  // namespace mfcc_fixed512
