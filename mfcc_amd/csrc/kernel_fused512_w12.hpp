// Fused 512/170/32 float kernel, TWELVE-wave form: the same arithmetic, codelets and tables as
// kernel_fused512.hpp (read its header first), re-staged so that every SIMD always has three waves in
// three DIFFERENT phases of the pipeline.
//
// Why.  Measured on the 4-wave form (profiles/r02_*, tools/alu_probe.hip): a wave issues one instruction per
// ~5 clocks whatever its type, so a tile costs a wave its ~460 instructions x 5 clocks plus MFMA / LDS / barrier
// waits = 5 300 clocks, and with 203 VGPRs only two such waves fit on a SIMD: its fp32 pipe idles ~60 % of the
// time.  Three waves per SIMD need <= 168 VGPRs; six-wave workgroups do not co-reside (tools/occ_probe.hip: the
// dispatcher places waves on SIMDs cyclically), so it is ONE workgroup of twelve waves per CU:
//
//   waves 0..3   group A: the four workers of a 16-frame tile (pass 1 / pass 2 / mel, nothing else)
//   waves 4..7   group B: the four workers of ANOTHER tile, half a period behind A
//   waves 8, 9   fetch the sample windows from HBM one half-step ahead and park them (pre-emphasised fp32)
//   wave  10     column 16 of the group that is in pass 2: 16 x 16 DFT + its mel contribution (fp32 MFMAs)
//   wave  11     the tail of the tile that finished pass 2 in the previous half-step: log2, DCT-II (on bf16-split matrix
//                instructions since round 3: three independent v_mfma_f32_16x16x32_bf16 per 16 coefficients instead of
//                eight fp32 ones that block the SIMD; frames with a -inf band take the fp32 chain), store
//
// Waves i, i + 4 and i + 8 share SIMD i's slot (cyclic placement): one worker in pass 1, one in pass 2, one helper.
// Registers: a worker without the helpers' state needs 156 VGPRs (164 with the dense mel sets).
//
// Time runs in HALF-STEPS h = 0, 1, 2, ... separated by ONE workgroup barrier each:
//   group A: pass 1 of its tile k at h = 2 k,     pass 2 + mel at h = 2 k + 1
//   group B: pass 1 of its tile k at h = 2 k + 1, pass 2 + mel at h = 2 k + 2
//   Pass 1 runs on operands the wave read out of its group's sample window one half-step EARLIER, in the middle of
//   its own pass 2 of the previous tile (between the power split and the MFMAs): the half-step then starts on
//   registers instead of on 32 LDS reads, while the partner group reads its T columns.
//   parkers: at h they park the window fetched during h - 2: S_A(h / 2 + 1) for even h, S_B((h + 1) / 2) for odd h
//            (a group's window is rewritten while that group runs pass 1 out of registers; it reads it in its next
//            pass 2); HBM latency under load is longer than a half-step, hence the two half-steps of lead
//   column 16 at h: the group in pass 2 (its V was written in h - 1); its partial sums go to Q slot 4 of the group
//   tail at h: the group that was in pass 2 at h - 1 (Q complete at the barrier; rewritten only at h + 1)
// DCX (a filter with weight on bin 0, kernel_fused512.hpp: that bin must not be summed in fp32): bin 0 is an EXACT
// integer sum over the raw samples.  With the pre-emphasis folded into the window, X[0] = sum_m c[m] x[m] over the
// 513 raw int16 samples x[-1 .. 511] of the frame; c is quantised to 49-bit integers held as seven balanced base-128
// digits, and the raw samples' two bytes are int8 operands as they lie in memory (lo byte ^ 0x80 = lo - 128, signed;
// the constant 128 sum(C) is added back): the parkers keep a raw copy R of the window next to S (shifted by one sample
// when the frames would start in the high half of a dword), wave 10 runs one chain of 17 v_mfma_i32_16x16x64_i8 per
// tile -- rows = digits x {lo, hi} byte, columns = the tile's 16 frames, K = the frame's 1026 bytes, accumulated
// exactly in int32 -- and the tail combines a frame's fourteen sums in double (exactly: two partial sums below 2^53) and adds the bin's weight x |X[0]|^2.
// (Round 2's first DC path, 32 cvt + 32 DFMA per worker lane and tile, cost 0.33 ms of 1.28; this one rides in
// wave 10's slack.)
// LDS (114.5 KB; 131 KB with DCX): T, V, Q (5 slots), S per group.  Virtual grid = 2 x workgroups: group A is virtual workgroup
// 2 w, group B 2 w + 1 of the 4-wave form's tile order.
#pragma once

#include "kernel_fused512.hpp"

namespace mfcc_fused12 {

using namespace mfcc_fused;

constexpr int kW12Waves = 12;
constexpr int kQSlots = 5;                          // 4 workers + column 16
constexpr int kQGroupWords = kQSlots * 2 * 256;
constexpr int kGroupWords = kTile * kTFrame + kTile * kVStride + kQGroupWords + kSUsed;
constexpr int kW12LdsWords = 2 * kGroupWords;
// DCX: the raw copy of a group's window (packed int16 pairs; the chain's last k-block reads up to 32 dwords past it,
// against zero weights) and the integer sums [group][tile parity][16 rows][16 frames]
constexpr int kRawWords = kSUsed / 2 + 32;
constexpr int kDcBlocks = 17;                       // k-blocks of 64 bytes: 513 samples = 1026 bytes
constexpr int kDcLdsWords = 2 * kRawWords + 2 * 2 * 256;
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kParkers = 128, kParkPieces = 3;      // waves 8, 9: 128 lanes x 3 pieces of 8 samples = the 3072-slot window
static_assert(kParkers * kParkPieces * 8 == kSUsed, "window pieces");

struct Fetch3 {
    i32x4 v[kParkPieces];
    int p[kParkPieces];          // dword in front of v[k]: its high half is the piece's predecessor sample
    int odd;                     // DCX: the window's shift is odd (the frames' sample -1 lies in a low half as fetched)
    int shift;                   // RAGGED: published for the workers
};

__device__ __forceinline__ void fetch_window3(const mfcc_k::StreamDesc &s, const Window &w, int u, Fetch3 &f) {
    f.odd = w.shift & 1;
    f.shift = w.shift;
    if (w.inside) {
        const i32x4 *g = reinterpret_cast<const i32x4 *>(w.ptr - w.shift);
        const int *g32 = reinterpret_cast<const int *>(g);
#pragma unroll
        for (int k = 0; k < kParkPieces; ++k) {
            f.v[k] = g[k * kParkers + u];
            f.p[k] = g32[4 * (k * kParkers + u) - 1];
        }
    } else {
        const long long first = (long long)w.t_in * kTileHop;      // channel-relative
        const int16_t *base = w.ptr - first;
#pragma unroll
        for (int k = 0; k < kParkPieces; ++k) {
            int h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = mfcc_k::sample_at_i(s, base, first + 8 * (k * kParkers + u) + j) & 0xFFFF;
            f.v[k] = (i32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            f.p[k] = mfcc_k::sample_at_i(s, base, first + 8 * (k * kParkers + u) - 1) << 16;
        }
    }
}

// RAW: also the raw copy for the integer DC sum.  Dword d of R holds the samples (2 d, 2 d + 1) of the window as
// fetched when the shift is odd, (2 d - 1, 2 d) when it is even: either way sample -1 of frame f is the LOW half of
// dword (shift >> 1) + 85 f.  Stored with the lo bytes' top bit flipped (lo - 128 as a signed byte).
template <bool RAW>
__device__ __forceinline__ void park_window3(float *Sf, int *R, int u, const Fetch3 &f, int *shift_slot = nullptr) {
    if (shift_slot && u == 0) *shift_slot = f.shift;
#pragma unroll
    for (int k = 0; k < kParkPieces; ++k) {
        preemph8(f.p[k], f.v[k], Sf + 8 * (k * kParkers + u));
        if constexpr (RAW) {
            i32x4 r = f.v[k];
            if (!f.odd) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    r[m] = (int)__builtin_amdgcn_alignbit((unsigned)f.v[k][m], (unsigned)(m ? f.v[k][m - 1] : f.p[k]), 16u);
            }
            r ^= 0x00800080;                           // lo bytes: unsigned -> signed - 128 (the int8 operands' form)
            *reinterpret_cast<i32x4 *>(R + 4 * (k * kParkers + u)) = r;
        }
    }
}

// cursor of virtual workgroup v
__device__ __forceinline__ Cursor cursor_of(const mfcc_k::StreamDesc &s, const LaunchGeom &g, unsigned v) {
    Cursor c;
    c.ch = (int)(v / (unsigned)g.tiles_per_ch);
    c.t_in = (int)(v - (unsigned)c.ch * (unsigned)g.tiles_per_ch);
    c.ptr = s.pcm + (long long)c.ch * s.ch_stride + (long long)c.t_in * kTileHop;
    return c;
}

// ---- ragged corpora without a packing copy: the tiles of utterances of different lengths, straight out of the
// caller's buffer.  The host writes one record per utterance, a small kernel expands them into one 32-byte record per
// tile, and a wave fetches its next tile's record with one scalar load, a tile ahead of its use.  Every utterance is an independent stream that starts from reset
// (history 0) and is zero-padded at its end, exactly like a channel of the plain call -- so the frames are the same
// bits as one call per utterance.
struct RaggedChan {
    long long pcm_off;       // first sample of the utterance, relative to StreamDesc::pcm
    long long out_row;       // first output row (frame) of the utterance
    int n_samples, frames;   // of the utterance
    int t_hi;                // tiles 1 .. t_hi have their whole window inside the utterance
    int tile0;               // index of its first tile in the tile map
};
// what a wave needs to know about a tile, in one 32-byte record (one s_load_dwordx8): expanded from the utterance
// records by ragged_tile_map_kernel
struct RaggedTile {
    long long pcm_off;       // first sample of the tile's utterance, relative to StreamDesc::pcm
    long long out_row;       // first output row (frame) of that utterance
    int n_samples, frames;   // of the utterance
    int t_hi;                // tiles 1 .. t_hi of the utterance have their whole window inside it
    int t_in;                // the tile's index inside its utterance
};
struct RaggedTables {
    const RaggedTile *tiles; // [n_tiles]
    int n_tiles;
};
// A record is read through the CONSTANT address space (written by ragged_tile_map_kernel before the launch, never by
// this kernel): only then does the compiler fetch it with a scalar load.  As plain global memory it became vector loads
// plus v_readfirstlane -- and a vmcnt wait, which in the parker waves means waiting for the window fetch in flight.
typedef const long long __attribute__((address_space(4))) ConstLL;
__device__ __forceinline__ RaggedTile load_tile_record(const RaggedTile *tiles, unsigned v) {
    const ConstLL *r = reinterpret_cast<const ConstLL *>(reinterpret_cast<uintptr_t>(tiles + v));
    const long long a = r[0], b = r[1], c = r[2], d = r[3];
    RaggedTile t;
    t.pcm_off = a;
    t.out_row = b;
    t.n_samples = (int)(c & 0xffffffffll);
    t.frames = (int)(c >> 32);
    t.t_hi = (int)(d & 0xffffffffll);
    t.t_in = (int)(d >> 32);
    return t;
}

// One role's walk over one group's tiles (virtual workgroup v0, stride gv): where the tile's samples are, how its
// window lies in its stream, where its rows go.  RAGGED = false is the arithmetic cursor of the plain call.
// RAGGED: the record of the NEXT tile of the walk is fetched while the current one is worked on (a scalar load that
// is waited for only a tile later; fetched on demand it cost every wave an L2 round trip per tile: 3.24 -> see DESIGN 6b).
template <bool RAGGED>
struct TileStream {
    Cursor c;
    mfcc_k::StreamDesc sl;   // the stream this tile belongs to (ragged: n_samples / frames of its utterance, halo 0)
    LaunchGeom gl;
    float *outp;
    unsigned v;
    int gv;
    RaggedTile nx;           // record of tile v + gv
    __device__ __forceinline__ void adopt(const mfcc_k::StreamDesc &s, const RaggedTile &r, int n_cep, float *out) {
        c.ch = 0;
        c.t_in = r.t_in;
        c.ptr = s.pcm + r.pcm_off + (long long)r.t_in * kTileHop;
        sl.n_samples = r.n_samples;
        sl.frames_per_ch = r.frames;
        gl.t_hi = r.t_hi;
        outp = out + r.out_row * n_cep;
    }
    // called right before a barrier: the wait that the next LDS access brings (lgkmcnt counts scalar loads too) then
    // falls into the barrier wait instead of into the work
    __device__ __forceinline__ void prefetch(const RaggedTables &r) {
        if constexpr (RAGGED) {
            const unsigned vn = v + (unsigned)gv;
            if ((int)vn < r.n_tiles) nx = load_tile_record(r.tiles, vn);
        }
    }
    __device__ __forceinline__ void start(const mfcc_k::StreamDesc &s, const LaunchGeom &g, const RaggedTables &r,
                                          unsigned v0, int stride, int n_cep, float *out) {
        sl = s;
        gl = g;
        outp = out;
        v = v0;
        gv = stride;
        if constexpr (RAGGED) {
            sl.halo = 0;
            gl.t_lo = 1;
            c.ch = 0;
            c.t_in = 0;
            c.ptr = s.pcm;
            nx = RaggedTile{0, 0, 0, 0, -1, 0};
            if ((int)v < r.n_tiles) adopt(s, load_tile_record(r.tiles, v), n_cep, out);
            prefetch(r);
        } else {
            c = cursor_of(s, g, v0);
        }
    }
    __device__ __forceinline__ void next(const mfcc_k::StreamDesc &s, const RaggedTables &r, int n_cep, float *out) {
        if constexpr (RAGGED) {
            v += (unsigned)gv;
            if ((int)v < r.n_tiles) adopt(s, nx, n_cep, out);      // the caller prefetches the one after it (see prefetch)
        } else {
            advance(c, gl);
        }
    }
    __device__ __forceinline__ Window window() const { return window_of(c, gl); }
};

#ifndef MFCC_W12_DCT_BF16
#define MFCC_W12_DCT_BF16 1
#endif
#ifndef MFCC_W12_PRIO_P1
#define MFCC_W12_PRIO_P1 1
#endif
#ifndef MFCC_W12_PRIO_P2
#define MFCC_W12_PRIO_P2 0
#endif

// Diagnostic build only (-DMFCC_W12_STAMPS): per wave, the clocks spent working (loop top -> barrier) and the clocks of
// the whole loop, summed over workgroups; written to a buffer nothing else reads.
#ifdef MFCC_W12_STAMPS
__device__ unsigned long long g_stamps12[kW12Waves * 4];      // [wave]: work even h, work odd h, total, half-steps
#define W12_LOOP_BEGIN unsigned long long w12_work[2] = {0, 0}, w12_t0, w12_begin = __builtin_amdgcn_s_memtime(); int w12_n = 0;
#define W12_T0 w12_t0 = __builtin_amdgcn_s_memtime();
#define W12_T1(h) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); w12_work[(h) & 1] += __builtin_amdgcn_s_memtime() - w12_t0; ++w12_n; } while (0)
#define W12_LOOP_END do { if (lane == 0) { atomicAdd(&g_stamps12[wave * 4 + 0], w12_work[0]); atomicAdd(&g_stamps12[wave * 4 + 1], w12_work[1]); \
    atomicAdd(&g_stamps12[wave * 4 + 2], __builtin_amdgcn_s_memtime() - w12_begin); atomicAdd(&g_stamps12[wave * 4 + 3], (unsigned long long)w12_n); } } while (0)
#else
#define W12_LOOP_BEGIN
#define W12_T0
#define W12_T1(h)
#define W12_LOOP_END
#endif

template <bool DENSE, bool RAGGED, bool DCX>
__global__ __launch_bounds__(64 * kW12Waves) __attribute__((amdgpu_waves_per_eu(3, 3)))
void mfcc_fused512_w12_kernel(mfcc_k::StreamDesc s, FusedTables t, LaunchGeom g, RaggedTables rag, float *__restrict__ out) {
    constexpr int kSets = SetsBf<DENSE>::N;
    __shared__ __attribute__((aligned(16))) float lds[kW12LdsWords + (DCX ? kDcLdsWords : 0) + (RAGGED ? 4 : 0)];
    // RAGGED: the shift of the window that lies in S_A / S_B, published by the parkers next to the window itself -- the
    // eight worker waves then need no per-tile record at all (they used to fetch one each: an L2 round trip per tile)
    int *const Shw = reinterpret_cast<int *>(lds + kW12LdsWords + (DCX ? kDcLdsWords : 0));
    int *const Rw = reinterpret_cast<int *>(lds + kW12LdsWords);           // DCX: raw windows [group][kRawWords]
    int *const DcI = Rw + 2 * kRawWords;                                    // DCX: [group][tile parity][16 rows][16 frames]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;         // 0: A, 1: B, 2: helpers
    const int wi = wave & 3;
    const int lo = lane & 15;
    const int q = lane >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    auto Tt = [&](int gi) { return lds + gi * kGroupWords; };
    auto Vt = [&](int gi) { return lds + gi * kGroupWords + kTile * kTFrame; };
    auto Qt = [&](int gi) { return lds + gi * kGroupWords + kTile * kTFrame + kTile * kVStride; };
    auto Sf = [&](int gi) { return lds + gi * kGroupWords + kTile * kTFrame + kTile * kVStride + kQGroupWords; };

    // tiles of the two groups: virtual workgroups 2 w and 2 w + 1 of a grid of 2 x gridDim.x
    // XCD-aware tile order: workgroup b runs on XCD b mod 8 (round-robin dispatch), and consecutive tiles share 342
    // samples of their windows -- so consecutive tile pairs go to workgroups of the SAME XCD (one L2 sees both windows):
    // the workgroups of an XCD take a contiguous eighth of the logical workgroup ids
    const unsigned nwg = gridDim.x;
    const unsigned bid = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    const unsigned va = 2u * bid, vb = va + 1u;
    const int n_tiles = RAGGED ? rag.n_tiles : g.tiles_per_ch * g.n_ch;   // < 2^30 (host check)
    const int gv = 2 * (int)gridDim.x;
    const int nA = (int)va < n_tiles ? (n_tiles - (int)va + gv - 1) / gv : 0;
    const int nB = (int)vb < n_tiles ? (n_tiles - (int)vb + gv - 1) / gv : 0;
    const int last_h = 2 * nA + 1;                                     // B's last tail (nB <= nA) is at 2 nB + 1

    if (grp < 2) {
        // =========================================================================== workers
        const int gi = grp;
        const int fr_id = wi + 8 * (q & 1) + 4 * (q >> 1);
        using mfcc_codelets::v2f;
        v2f wp[16], tw[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) wp[i] = reinterpret_cast<const v2f *>(t.win)[lo * 16 + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) tw[i] = reinterpret_cast<const v2f *>(t.tw)[lo * 16 + i];
        u32x4 ah[kSets], al[kSets];
#pragma unroll
        for (int st = 0; st < kSets; ++st)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // DCX: bin 0 comes from wave 10's integer sum, so the workers' weights leave it out
                const uint32_t *abf = DCX ? t.a_mel_bf_nodc : t.a_mel_bf;
                ah[st][d] = abf[((wi * kSets + st) * 2 + 0) * 256 + d * 64 + lane];
                al[st][d] = abf[((wi * kSets + st) * 2 + 1) * 256 + d * 64 + lane];
            }
        float *const T = Tt(gi), *const V = Vt(gi), *const Q = Qt(gi), *const S = Sf(gi);
        const int lane_slot = fr_id * kHop + lo;
        const int n_mine = gi ? nB : nA;
        TileStream<false> cur;                         // the plain call's arithmetic cursor (RAGGED: unused, see Shw)
        if constexpr (!RAGGED) cur.start(s, g, rag, gi ? vb : va, gv, t.n_cep, out);

        lds_barrier();                                 // the parkers' prologue: S_A(0) and S_B(0) are in LDS
        // The pass-1 operands of a tile are read out of its window ONE half-step early, in the middle of the same
        // group's pass 2 of the previous tile (the window was re-parked the half-step before that): pass 1 then starts
        // on registers instead of waiting for 32 LDS reads while the partner group does the same for its T columns.
        v2f ep[16];
        auto load_ep = [&]() {
            const float *sp = S + lane_slot + (RAGGED ? Shw[gi] : cur.window().shift);
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) ep[n1 >> 1][n1 & 1] = sp[16 * n1];
        };
        if (n_mine > 0) load_ep();                     // tile 0 (group B idles through h = 0 with its operands loaded)
        lds_barrier();                                 // second prologue barrier: the parkers may now re-park S_A (h = 0)
        auto pass1 = [&]() {
            // ---------------- pass 1: windowed real FFT-32 over n1 of the pre-emphasised samples
            // (the longer of the two phases: it gets the SIMD's issue priority over the partner group's pass 2,
            // whichever of the two waves is older -- measured: 2000 clocks per half-step instead of 2400)
            __builtin_amdgcn_s_setprio(MFCC_W12_PRIO_P1);
            v2f ty[16];
            float y16;
            mfcc_codelets::rfft32_tw(ep, wp, tw, ty, y16);
            v2f *tcol0 = reinterpret_cast<v2f *>(T + fr_id * kTFrame) + lo;        // a store's lanes: consecutive n2
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) tcol0[k1 * (kTRow / 2)] = ty[k1];
            V[fr_id * kVStride + lo] = y16;
        };
        auto pass2 = [&]() {
            // ---------------- pass 2: complex FFT-16 over n2 for frame lo, column k1 = 4 wi + q; mel MFMAs
            __builtin_amdgcn_s_setprio(MFCC_W12_PRIO_P2);
            float pw[16];
            {
                v2f x[16];
                const f32x4 *trow = reinterpret_cast<const f32x4 *>(T + lo * kTFrame + (4 * wi + q) * kTRow);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 a = trow[i];
                    x[2 * i] = (v2f){a[0], a[1]};
                    x[2 * i + 1] = (v2f){a[2], a[3]};
                }
                v2f pp[8];                           // (|z[k2]|^2, |z[k2 + 8]|^2): the codelet's last layer is transposed
                mfcc_codelets::cfft16_pow(x, pp);
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) pw[k2] = pp[k2].x, pw[k2 + 8] = pp[k2].y;
            }
            PowerBf pb;
            split_power(pw, pb);
            if constexpr (!RAGGED) cur.next(s, rag, t.n_cep, out);
            load_ep();           // the next tile's operands fly during the MFMAs (after the last tile they are never used)
            f32x4 acc[kSets];
#pragma unroll
            for (int st = 0; st < kSets; ++st) acc[st] = zero;
            mel_bf_all<DENSE, 0>(ah, al, pb, acc, [](auto) {});
            f32x4 b0, b1;
            mel_bf_blocks<DENSE>(acc, b0, b1);
            *reinterpret_cast<f32x4 *>(Q + (2 * wi + 0) * 256 + lane * 4) = b0;
            *reinterpret_cast<f32x4 *>(Q + (2 * wi + 1) * 256 + lane * 4) = b1;
        };
        // pass 1 and pass 2 of a tile are consecutive half-steps for either group (B one half-step behind A): an
        // unconditional loop body, so that `ep` -- defined at the end of pass 2, consumed by the next pass 1 -- is not
        // live through pass 2's register peak; the half-steps in which the group has nothing to do are bare barriers
        W12_LOOP_BEGIN
        int bars = last_h + 1;                         // every wave of the workgroup passes this many barriers
        if (gi) {                                      // h = 0: group B idles
            lds_barrier();
            --bars;
        }
        for (int i = 0; i < n_mine; ++i) {
            W12_T0
            pass1();
            W12_T1(0);
            lds_barrier();
            W12_T0
            pass2();
            W12_T1(1);
            lds_barrier();
            bars -= 2;
        }
        for (; bars > 0; --bars) lds_barrier();
        W12_LOOP_END;
    } else if (wi < 2) {
        // =========================================================================== parkers (waves 8, 9)
        const int u = wi * 64 + lane;                  // 0..127
        // the youngest waves of the workgroup lose the issue arbitration against the eight workers (priority, then age)
        // although they have the least to do and everybody waits for them at the barrier
        __builtin_amdgcn_s_setprio(3);
        TileStream<RAGGED> pa, pb;
        pa.start(s, g, rag, va, gv, t.n_cep, out);
        pb.start(s, g, rag, vb, gv, t.n_cep, out);
        int ka = 0, kb = 0;                            // next tile of each stream to fetch
        // one register set per stream: a window is fetched TWO half-steps before it is parked (HBM latency under
        // load is longer than a half-step), i.e. right after the same stream's previous window has been parked
        Fetch3 fa, fb;
        bool have_a = false, have_b = false;
        if (nA > 0) {                                  // prologue: S_A(0) and S_B(0) directly
            fetch_window3(pa.sl, pa.window(), u, fa);
            park_window3<DCX>(Sf(0), Rw, u, fa, RAGGED ? Shw + 0 : nullptr);
            pa.next(s, rag, t.n_cep, out);
            pa.prefetch(rag);
            ++ka;
        }
        if (nB > 0) {
            fetch_window3(pb.sl, pb.window(), u, fb);
            park_window3<DCX>(Sf(1), Rw + kRawWords, u, fb, RAGGED ? Shw + 1 : nullptr);
            pb.next(s, rag, t.n_cep, out);
            pb.prefetch(rag);
            ++kb;
        }
        if (ka < nA) {                                 // S_A(1): parked at h = 0
            fetch_window3(pa.sl, pa.window(), u, fa);
            pa.next(s, rag, t.n_cep, out);
            pa.prefetch(rag);
            ++ka;
            have_a = true;
        }
        if (kb < nB) {                                 // S_B(1): parked at h = 1
            fetch_window3(pb.sl, pb.window(), u, fb);
            pb.next(s, rag, t.n_cep, out);
            pb.prefetch(rag);
            ++kb;
            have_b = true;
        }
        pa.prefetch(rag);
        pb.prefetch(rag);
        lds_barrier();                                 // S_A(0), S_B(0) are parked: the workers fetch tile 0's operands
        lds_barrier();                                 // ... and hold them in registers: h = 0 may start
        W12_LOOP_BEGIN
        for (int h = 0; h <= last_h; ++h) {
            W12_T0
            // A window is re-parked in the half-step in which its group runs pass 1 on operands it already holds in
            // registers: S_A(h / 2 + 1) at even h, S_B((h + 1) / 2) at odd h; the group reads it in its next pass 2
            if (!(h & 1)) {
                if (have_a) park_window3<DCX>(Sf(0), Rw, u, fa, RAGGED ? Shw + 0 : nullptr);
                have_a = false;
                if (ka < nA) {
                    fetch_window3(pa.sl, pa.window(), u, fa);
                    pa.next(s, rag, t.n_cep, out);
                    ++ka;
                    have_a = true;
                }
            } else {
                if (have_b) park_window3<DCX>(Sf(1), Rw + kRawWords, u, fb, RAGGED ? Shw + 1 : nullptr);
                have_b = false;
                if (kb < nB) {
                    fetch_window3(pb.sl, pb.window(), u, fb);
                    pb.next(s, rag, t.n_cep, out);
                    ++kb;
                    have_b = true;
                }
            }
            pa.prefetch(rag);
            pb.prefetch(rag);
            W12_T1(h);
            lds_barrier();
        }
        W12_LOOP_END;
    } else if (wi == 2) {
        // =========================================================================== column 16 (wave 10)
        __builtin_amdgcn_s_setprio(3);
        float ax[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ax[i] = t.a_extra[(1 * kAextra + i) * 64 + lane];
        // DCX: bin 0 of a tile's 16 frames as exact integer sums over the raw window (see the header).  A window is
        // valid for one half-step: R_A(k) during h = 2 k - 1 (tile 0: between the prologue barriers), R_B(k) during 2 k.
        // (This wave has no global load in flight.  In the parker waves, which have the registers too, every vmcnt the
        // compiler puts into the chain waits for the window fetch just issued: +1 800 clocks per half-step, measured.)
        v4i adc[DCX ? kDcBlocks : 1];
        TileStream<RAGGED> da, db;
        if constexpr (DCX) {
#pragma unroll
            for (int b = 0; b < kDcBlocks; ++b) adc[b] = reinterpret_cast<const v4i *>(t.a_dc_i8)[b * 64 + lane];
            da.start(s, g, rag, va, gv, t.n_cep, out);
            db.start(s, g, rag, vb, gv, t.n_cep, out);
        }
        auto dc_tile = [&](int gi, int k, TileStream<RAGGED> &dts) {
            // lane (frame lo, k-group q): 16 bytes = 4 dwords per k-block, 16 dwords apart
            const int *rp = Rw + gi * kRawWords + (dts.window().shift >> 1) + 85 * lo + 4 * q;
            v4i acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};       // two chains: an MFMA does not wait for the one before it
            // the operands of nine (then eight) k-blocks are read in one go: under the workers' LDS traffic a read
            // takes several hundred clocks, and two in flight at a time made the chain 1 400 clocks long
            constexpr int kHalf = (kDcBlocks + 1) / 2;
            v4i x[kHalf];
#pragma unroll
            for (int b = 0; b < kHalf; ++b) x[b] = (v4i){rp[16 * b], rp[16 * b + 1], rp[16 * b + 2], rp[16 * b + 3]};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < kHalf; ++b) {
                if (b & 1) acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(adc[b], x[b], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(adc[b], x[b], acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = kHalf; b < kDcBlocks; ++b)
                x[b - kHalf] = (v4i){rp[16 * b], rp[16 * b + 1], rp[16 * b + 2], rp[16 * b + 3]};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = kHalf; b < kDcBlocks; ++b) {
                if (b & 1) acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(adc[b], x[b - kHalf], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(adc[b], x[b - kHalf], acc, 0, 0, 0);
            }
            acc += acc2;
            // rows 4 q + r (digit j = row of the lo bytes, kDcDigits + j of the hi bytes), column = frame lo
            int *d = DcI + (gi * 2 + (k & 1)) * 256 + lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) d[(4 * q + r) * 16] = acc[r];
            dts.next(s, rag, t.n_cep, out);
        };
        lds_barrier();                                 // the parkers' two prologue barriers
        if constexpr (DCX) {
            if (nA > 0) dc_tile(0, 0, da);
            da.prefetch(rag);
        }
        lds_barrier();
        W12_LOOP_BEGIN
        for (int h = 0; h <= last_h; ++h) {
            W12_T0
            // the group in pass 2 at h: A (tile (h - 1) / 2) for odd h, B (tile h / 2 - 1) for even h >= 2
            const int gi = (h & 1) ? 0 : 1;
            const int k = (h & 1) ? (h - 1) / 2 : h / 2 - 1;
            if (k >= 0 && k < (gi ? nB : nA)) {
                const float *V = Vt(gi);
                float *Q = Qt(gi);
                const float v0 = V[lo * kVStride + 0 + q], v1 = V[lo * kVStride + 4 + q];
                const float v2 = V[lo * kVStride + 8 + q], v3 = V[lo * kVStride + 12 + q];
                f32x4 sp = MFCC_MFMA(ax[0], v0, zero);
                f32x4 sp2 = MFCC_MFMA(ax[1], v1, zero);
                sp = MFCC_MFMA(ax[2], v2, sp);
                sp2 = MFCC_MFMA(ax[3], v3, sp2);
                sp += sp2;
                const float s0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);      // bin 16 + 64 q
                const float s1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);      // bin 48 + 64 q
                const f32x4 x0 = MFCC_MFMA(ax[4], s0, zero), y0 = MFCC_MFMA(ax[5], s1, zero);
                const f32x4 x1 = MFCC_MFMA(ax[6], s0, zero), y1 = MFCC_MFMA(ax[7], s1, zero);
                *reinterpret_cast<f32x4 *>(Q + (2 * 4 + 0) * 256 + lane * 4) = x0 + y0;
                *reinterpret_cast<f32x4 *>(Q + (2 * 4 + 1) * 256 + lane * 4) = x1 + y1;
            }
            if constexpr (DCX) {
                if (h & 1) {
                    if ((h + 1) / 2 < nA) dc_tile(0, (h + 1) / 2, da);
                } else {
                    if (h / 2 < nB) dc_tile(1, h / 2, db);
                }
                da.prefetch(rag);
                db.prefetch(rag);
            }
            W12_T1(h);
            lds_barrier();
        }
        W12_LOOP_END;
    } else {
        // =========================================================================== tail (wave 11)
        __builtin_amdgcn_s_setprio(3);
        float ax[kAextra];
#pragma unroll
        for (int i = 0; i < kAextra; ++i) ax[i] = t.a_extra[(0 * kAextra + i) * 64 + lane];
#if MFCC_W12_DCT_BF16
        u32x4 dct_h[2], dct_l[2];
#pragma unroll
        for (int tile = 0; tile < 2; ++tile)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                dct_h[tile][d] = t.a_dct_bf[((size_t)tile * 2 + 0) * 256 + d * 64 + lane];
                dct_l[tile][d] = t.a_dct_bf[((size_t)tile * 2 + 1) * 256 + d * 64 + lane];
            }
#endif
        const int lane_off = lo * t.n_cep + 4 * q;
        TileStream<RAGGED> ta, tb;
        ta.start(s, g, rag, va, gv, t.n_cep, out);
        tb.start(s, g, rag, vb, gv, t.n_cep, out);
        // bin 0's weight in this lane's eight filters (4 q + r of block 0, 16 + 4 q + r of block 1)
        f32x4 wdc0 = zero, wdc1 = zero;
        double dc_scale = 0.0, dc_bias_lo = 0.0, dc_bias_hi = 0.0;
        if constexpr (DCX) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wdc0[r] = t.w_dc[4 * q + r];
                wdc1[r] = t.w_dc[16 + 4 * q + r];
            }
            dc_scale = t.dc_consts[0];
            dc_bias_lo = t.dc_consts[1];
            dc_bias_hi = t.dc_consts[2];
        }
        lds_barrier();
        lds_barrier();
        W12_LOOP_BEGIN
        for (int h = 0; h <= last_h; ++h) {
            W12_T0
            // the group that was in pass 2 at h - 1: A (tile h / 2 - 1) for even h, B (tile (h - 3) / 2) for odd h
            const int gi = (h & 1) ? 1 : 0;
            const int k = (h & 1) ? (h - 3) / 2 : h / 2 - 1;
            if (h >= 2 && k >= 0 && k < (gi ? nB : nA)) {
                const f32x4 *Q4 = reinterpret_cast<const f32x4 *>(Qt(gi)) + lane;
                f32x4 m0 = ((Q4[0 * 64] + Q4[2 * 64]) + (Q4[4 * 64] + Q4[6 * 64])) + Q4[8 * 64];
                f32x4 m1 = ((Q4[1 * 64] + Q4[3 * 64]) + (Q4[5 * 64] + Q4[7 * 64])) + Q4[9 * 64];
                if constexpr (DCX) {                   // bin 0 of frame lo: wave 10's fourteen integer sums, combined in double
                    // sum_j 128^j (lo_j + 256 hi_j) + 128 sum(C), exactly: digits 0..3 and the low 28 bits of the
                    // constant stay below 2^52, digits 4..6 (x 2^-28) and its high part below 2^45 -- every FMA is
                    // exact, and the one rounding is that of the final sum
                    const int *dr = DcI + (gi * 2 + (k & 1)) * 256 + lo;
                    double L = dc_bias_lo, H = dc_bias_hi, sc = 1.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        L = __builtin_fma((double)dr[j * 16], sc, L);
                        L = __builtin_fma((double)dr[(kDcDigits + j) * 16], sc * 256.0, L);
                        sc *= 128.0;
                    }
                    sc = 1.0;
#pragma unroll
                    for (int j = 4; j < kDcDigits; ++j) {
                        H = __builtin_fma((double)dr[j * 16], sc, H);
                        H = __builtin_fma((double)dr[(kDcDigits + j) * 16], sc * 256.0, H);
                        sc *= 128.0;
                    }
                    const double x0 = __builtin_fma(H, 268435456.0, L) * dc_scale;
                    const float p0 = (float)(x0 * x0);
                    m0 += wdc0 * p0;
                    m1 += wdc1 * p0;
                }
                f32x4 l0, l1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    l0[r] = __builtin_amdgcn_logf(m0[r]);
                    l1[r] = __builtin_amdgcn_logf(m1[r]);
                }
                if (t.n_mel <= 16) l1 = zero;          // no filters 16..31 (uniform)
#if MFCC_W12_DCT_BF16
                // The DCT on bf16-split matrix instructions: K = 32 is the 32 log-mel values of a frame, so an M tile of 16
                // coefficients is ONE v_mfma_f32_16x16x32_bf16 per term (3 of 16 clocks, which the workers of this SIMD can
                // issue beside) instead of 8 fp32 ones of 32 clocks each, during which they issue nothing.  A -inf (a
                // silent band) would split into (-inf, NaN) and turn c0 = -inf into NaN -- in its own frame only, the
                // columns of a matrix product do not mix: a tile that holds one runs the fp32 chain as well and the frames
                // with a -inf take its results (a frame's bits do not depend on what else is in its tile).
                const float lmin = fminf(fminf(__builtin_fminf(l0[0], l0[1]), __builtin_fminf(l0[2], l0[3])),
                                         fminf(__builtin_fminf(l1[0], l1[1]), __builtin_fminf(l1[2], l1[3])));
                const unsigned long long special = __builtin_amdgcn_ballot_w64(!(lmin > -3.0e38f));
                u32x4 bh, bl;
                {
                    uint32_t hi, lw;
                    split_bf16_pair(l0[0], l0[1], hi, lw); bh[0] = hi; bl[0] = lw;
                    split_bf16_pair(l0[2], l0[3], hi, lw); bh[1] = hi; bl[1] = lw;
                    split_bf16_pair(l1[0], l1[1], hi, lw); bh[2] = hi; bl[2] = lw;
                    split_bf16_pair(l1[2], l1[3], hi, lw); bh[3] = hi; bl[3] = lw;
                }
                // THREE INDEPENDENT products per M tile, summed afterwards.  Written as `d = mfma(.., 0); if (n_cep > 16) e =
                // mfma(.., 0); d = mfma(.., d); if (n_cep > 16) e = mfma(.., e); ...` -- an accumulator chain with the second M
                // tile's products and their UNIFORM BRANCHES between its links -- this tail gave wrong sums in 150-270 of 3 530
                // tiles per run, not repeatably, mostly the tile's last column (asm or plain-C conversions, any number of wait
                // states around it).  The same chain without the branches between its links is right, and so is this form
                // (tools/dbg_tail.py on builds of each).  The failing instruction sequence in isolation is right too
                // (tools/mfma_chain_probe.hip: compiled or written by hand with NO wait state behind the branch target, alone,
                // beside eleven busy waves, at priority 3: 0 of 1.6e10 values), so the dependent matrix instruction is not the
                // culprit and what is was not isolated.  Every other contraction of these kernels is term-major: consecutive
                // matrix instructions never share an accumulator.
                const f32x4 dA = MFCC_MFMA_BF(dct_h[0], bh, zero), dB = MFCC_MFMA_BF(dct_h[0], bl, zero);
                const f32x4 dC = MFCC_MFMA_BF(dct_l[0], bh, zero);
                f32x4 d = (dA + dB) + dC, e = zero;
                if (t.n_cep > 16) {
                    const f32x4 eA = MFCC_MFMA_BF(dct_h[1], bh, zero), eB = MFCC_MFMA_BF(dct_h[1], bl, zero);
                    const f32x4 eC = MFCC_MFMA_BF(dct_l[1], bh, zero);
                    e = (eA + eB) + eC;
                }
                if (special != 0) {                    // uniform, rare
                    f32x4 d0 = zero, d1 = zero, e0 = zero, e1 = zero;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d0 = MFCC_MFMA(ax[r], l0[r], d0);
                        d1 = MFCC_MFMA(ax[4 + r], l1[r], d1);
                        e0 = MFCC_MFMA(ax[8 + r], l0[r], e0);
                        e1 = MFCC_MFMA(ax[12 + r], l1[r], e1);
                    }
                    // this lane's frame is column lo: its 32 values sit in lanes lo, lo + 16, lo + 32, lo + 48
                    const bool mine = ((special >> lo) & 0x0001000100010001ull) != 0;
                    if (mine) {
                        d = d0 + d1;
                        e = e0 + e1;
                    }
                }
                // (not `TileStream &c = gi ? tb : ta`: a reference picked at run time puts both walks into scratch)
                auto finish = [&](TileStream<RAGGED> &c) {
                    const long long fr0 = (long long)c.c.t_in * kTile;
                    const long long rows_left = c.sl.frames_per_ch - fr0;
                    float *o = c.outp + ((long long)c.c.ch * c.sl.frames_per_ch + fr0) * t.n_cep + lane_off;
                    if (lo < rows_left) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (4 * q + r < t.n_cep) o[r] = d[r];
                            if (16 + 4 * q + r < t.n_cep) o[16 + r] = e[r];
                        }
                    }
                    c.next(s, rag, t.n_cep, out);
                };
                if (gi) finish(tb);
                else finish(ta);
#else
                f32x4 d0 = zero, d1 = zero;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d0 = MFCC_MFMA(ax[r], l0[r], d0);
                    d1 = MFCC_MFMA(ax[4 + r], l1[r], d1);
                }
                // (not `TileStream &c = gi ? tb : ta`: a reference picked at run time puts both walks into scratch)
                auto finish = [&](TileStream<RAGGED> &c) {
                    dct_store(c.sl, t, l0, l1, d0, d1, ax, c.c, lo, q, lane_off, c.outp);
                    c.next(s, rag, t.n_cep, out);
                };
                if (gi) finish(tb);
                else finish(ta);
#endif
            }
            ta.prefetch(rag);
            tb.prefetch(rag);
            W12_T1(h);
            lds_barrier();
        }
        W12_LOOP_END;
    }
}

inline const char *kernel_name() { return "mfcc_fused512_w12_kernel"; }

// returns false when the problem does not fit (then the 4-wave kernel runs)
inline bool launch(const mfcc_k::StreamDesc &s, const FusedTables &t, bool dense, float *out, int n_cu,
                   hipStream_t stream) {
    const bool dcx = t.win_dc != nullptr;                // a filter has weight on bin 0: that bin in double (dense sets)
    const long long tiles_per_ch = (s.frames_per_ch + kTile - 1) / kTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    if (n_tiles >= (1ll << 30) || tiles_per_ch >= (1ll << 26) || n_ch >= (1ll << 30)) return false;
    long long wgs = (n_tiles + 1) / 2;
    if (wgs > n_cu) wgs = n_cu;
    if (wgs < 1) wgs = 1;
    const long long grid = 2 * wgs;                      // virtual workgroups: the cursor stride
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
    const RaggedTables none = {nullptr, 0};
    if (dcx)
        hipLaunchKernelGGL((mfcc_fused512_w12_kernel<true, false, true>), dim3((unsigned)wgs), dim3(64 * kW12Waves), 0, stream, s, t, g, none, out);
    else if (dense)
        hipLaunchKernelGGL((mfcc_fused512_w12_kernel<true, false, false>), dim3((unsigned)wgs), dim3(64 * kW12Waves), 0, stream, s, t, g, none, out);
    else
        hipLaunchKernelGGL((mfcc_fused512_w12_kernel<false, false, false>), dim3((unsigned)wgs), dim3(64 * kW12Waves), 0, stream, s, t, g, none, out);
    return true;
}

// ---- ragged corpus, host side: per-utterance records (built by the caller in pinned memory), the tile map (built on
// the device from them), the launch
inline int ragged_t_hi(long long n_samples, long long tiles) {
    if (n_samples < kSUsed) return -1;
    const long long hi = (n_samples - kSUsed) / kTileHop;
    return (int)(hi < tiles ? hi : tiles);
}

__global__ void ragged_tile_map_kernel(const RaggedChan *__restrict__ chans, int n_chan, RaggedTile *__restrict__ tiles) {
    for (int u = blockIdx.x; u < n_chan; u += gridDim.x) {
        const RaggedChan ch = chans[u];
        const int n = (ch.frames + kTile - 1) / kTile;
        for (int k = threadIdx.x; k < n; k += blockDim.x)
            tiles[ch.tile0 + k] = RaggedTile{ch.pcm_off, ch.out_row, ch.n_samples, ch.frames, ch.t_hi, k};
    }
}

inline bool launch_ragged(const int16_t *d_pcm, const RaggedChan *d_chans, int n_chan, RaggedTile *d_map, int n_tiles,
                          const FusedTables &t, bool dense, float *out, int n_cu, hipStream_t stream) {
    if (n_tiles <= 0) return false;
    const bool dcx = t.win_dc != nullptr;
    unsigned blocks = (unsigned)(n_chan < n_cu * 8 ? n_chan : n_cu * 8);
    hipLaunchKernelGGL(ragged_tile_map_kernel, dim3(blocks), dim3(64), 0, stream, d_chans, n_chan, d_map);
    long long wgs = (n_tiles + 1) / 2;
    if (wgs > n_cu) wgs = n_cu;
    mfcc_k::StreamDesc s;
    s.pcm = d_pcm;
    s.ch_stride = 0;
    s.n_samples = 0;
    s.halo = 0;
    s.frames_per_ch = 1;
    s.total_frames = 0;
    s.hop = kHop;
    LaunchGeom g = {};
    g.tiles_per_ch = 1;
    g.n_ch = 0;
    g.t_lo = 1;
    g.t_hi = -1;
    const RaggedTables r = {d_map, n_tiles};
    if (dcx)
        hipLaunchKernelGGL((mfcc_fused512_w12_kernel<true, true, true>), dim3((unsigned)wgs), dim3(64 * kW12Waves), 0, stream, s, t, g, r, out);
    else if (dense)
        hipLaunchKernelGGL((mfcc_fused512_w12_kernel<true, true, false>), dim3((unsigned)wgs), dim3(64 * kW12Waves), 0, stream, s, t, g, r, out);
    else
        hipLaunchKernelGGL((mfcc_fused512_w12_kernel<false, true, false>), dim3((unsigned)wgs), dim3(64 * kW12Waves), 0, stream, s, t, g, r, out);
    return true;
}

}  // namespace mfcc_fused12
