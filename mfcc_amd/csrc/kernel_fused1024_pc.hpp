// Fused 1024/341/40 float kernel for gfx950 (MI355X), PRODUCER / CONSUMER form -- BASELINE.json configs[3]: nfft 1024,
// hop 1024 // 3 = 341 (mfcc/core/mfcc.py:43), 40 mel bands, n_cep <= 40, the mel contraction on the matrix cores.
// Same arithmetic, codelets and operand sets as kernel_fused1024_t8.hpp (read its header first); what differs is who does
// what, when.
//
// Round 2's kernel ran a 16-frame tile through eight IDENTICAL waves: all of them in pass 1, a barrier, all of them in
// pass 2 and the matrix instructions, a barrier.  The two waves of a SIMD were always in the same phase, so the LDS bursts
// (window reads, T writes, T reads), the fp32 MFMAs and the FFT arithmetic of a tile ran one after the other: 6 700 clocks
// per tile for ~1 750 clocks of instruction issue per wave.  Eight-frame tiles in two workgroups per CU (the _t8 kernel)
// overlap the phases but pay every per-tile cost twice and run every MFMA half empty: 260 vector instructions per frame
// against 195, 13 % slower (profiles/r03_notes.md).  Here the tile keeps its 16 frames and the WAVES specialise:
//
//   waves 0..3   producers: pass 1 of tile k + 1 -- 2 batches of (2 frames x 32 n2) per wave: window reads, the REAL
//                32-point FFT over n1 (rfft32_tw) -- while the consumers work on tile k.  T holds tile k until the consumers
//                are done with it, so a producer keeps its 2 x 33 outputs in REGISTERS across the barrier and stores them to
//                T right behind it (the window and twiddle pairs are its only resident constants: 64 registers);
//   waves 4..7   consumers: pass 2 of tile k -- 2 batches of (16 frames x 4 columns k1) per wave: 16 ds_read_b128 of a
//                column's 32 contiguous values, the complex FFT-32 over n2 split in its even / odd outputs (cfft32_h0 /
//                _h1, h = wave & 1), |X|^2, bf16 split, 12 (15) v_mfma_f32_16x16x32_bf16 per batch with all 16 columns
//                in use (the mel weights are their only resident constants: 64 registers) -- then, behind the barrier,
//                column 16 (waves 4, 5: a 32-point DFT matrix on fp32 MFMAs), the partial sums to Q, and on wave 6 the tail
//                of tile k - 1: log2, DCT-II (12 fp32 MFMAs per 16 coefficients), store.
//
// Waves i and i + 4 share a SIMD: one producer beside one consumer, FFT arithmetic beside LDS reads and matrix
// instructions.  Two barriers per tile: [A: produce k + 1 | consume k] [B: T, V <- registers, park S(k + 2) | column 16, Q,
// tail].  The sample windows are fetched from HBM by all 512 lanes a whole interval A before they are parked; S, V and
// Q are double buffers (144 KB of LDS in all, one workgroup per CU).
#pragma once

#include "kernel_fused1024_t8.hpp"

namespace mfcc_f1kpc {

using namespace mfcc_f1k;      // Sets<VAR>, bin_of, kGrpM, the MFMA macros, split_bf16_pair, f32x4 ...

constexpr int kPcTile = 16, kPcWaves = 8;
constexpr int kPcTileHop = kPcTile * kHop;        // 5456 samples between consecutive tiles
constexpr int kPcTRow = 64;                       // words per k1 row of T[frame][k1][n2]: 32 complex
constexpr int kPcTFrame = 16 * kPcTRow + 4;       // 1028: (row, frame) strides of (16, 257) 16-byte units: conflict-free b128 reads
constexpr int kPcPieces = (7 + (kPcTile - 1) * kHop + kNfft + 7) / 8;   // 769 pieces of 8 samples
constexpr int kPcFetchers = 64 * kPcWaves;        // 512
constexpr int kPcSecond = kPcPieces - kPcFetchers;   // lanes that take a second piece (257)
constexpr int kPcSUsed = 8 * kPcPieces;           // 6152 fp32 slots
constexpr int kPcQWords = 4 * kBlocks * 256;      // partial mel sums of the four consumers: [wave][block][lane * 4]
constexpr int kPcVWords = kPcTile * kVStride;
constexpr int kPcLdsWords = kPcTile * kPcTFrame + 2 * kPcSUsed + 2 * kPcQWords + 2 * kPcVWords;
constexpr int kPcArole = 14;                      // role operands: column-16 DFT (8) + its mel weights (6) / DCT rows (12)

struct Tables {
    int variant;
    const float *win;      // [32 n2][32 n1]  hamming[32 n1 + n2] / 64
    const float *tw;       // [32 n2][16 k1][2] W1024^(n2 k1)
    const uint32_t *a_bf;  // [4 consumers][2 batches][sets][hi, lo][4 dwords][64 lanes] mel weights as bf16 pairs
    const float *a_role;   // [3 roles][kPcArole][64]  0 / 1: column 16, j' < 8 / >= 8; 2: DCT rows 0..15
    const float *a_dct_hi; // [2][12][64]  DCT rows of coefficients 16..31 and 32..47
    int n_cep;
};

inline bool build_tables_for(int variant, int sample_rate, double power_scale, double lifter, int n_cep,
                             std::vector<char> &blob) {
    using namespace mfcc_tables;
    const SetsView sv = sets_view(variant);
    std::vector<float> win(32 * 32), tw(32 * 16 * 2), arole(size_t(3) * kPcArole * 64, 0.0f), adct(size_t(2) * 12 * 64, 0.0f);
    std::vector<double> w = hamming_periodic(kNfft);
    for (int n2 = 0; n2 < 32; ++n2)
        for (int n1 = 0; n1 < 32; ++n1) win[n2 * 32 + n1] = float(w[32 * n1 + n2] / 64.0);
    for (int n2 = 0; n2 < 32; ++n2)
        for (int k1 = 0; k1 < 16; ++k1) {
            double a = -2.0 * kPi * double(n2 * k1) / 1024.0;
            tw[(n2 * 16 + k1) * 2 + 0] = float(std::cos(a));
            tw[(n2 * 16 + k1) * 2 + 1] = float(std::sin(a));
        }
    const int nb = kNfft / 2 + 1;
    std::vector<double> md = mel_dense(kNfft, kMel, double(sample_rate));     // [40][513]
    for (int f = 0; f < kMel; ++f)
        if (md[size_t(f) * nb] != 0.0) return false;      // weight on the real-valued DC bin: not summed in fp32
    const double inv = 1.0 / (power_scale * power_scale);
    std::vector<char> covered(size_t(kMel) * nb, 0);
    auto Wt = [&](int filt, int bin) -> double { return filt < kMel ? md[size_t(filt) * nb + bin] * inv : 0.0; };
    auto bf16_round = [](float v) -> uint32_t {
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
    // mel operands: lane l of consumer c, batch b, set st holds rows l & 15 of filter block blk[st] at K slots i = 0..7
    // <-> bin(k1 = 8 (c >> 1) + 4 b + (l >> 4), h = c & 1, m = kGrpM[grp[st]][i]); dword d = slots (2 d, 2 d + 1)
    std::vector<uint32_t> abf(size_t(4) * 2 * sv.n * 2 * 4 * 64, 0u);
    for (int c = 0; c < 4; ++c)
        for (int b = 0; b < 2; ++b)
            for (int st = 0; st < sv.n; ++st)
                for (int l = 0; l < 64; ++l) {
                    uint32_t hi[8], lo[8];
                    for (int i = 0; i < 8; ++i) {
                        const int filt = sv.blk[st] * 16 + (l & 15), k1 = 8 * (c >> 1) + 4 * b + (l >> 4);
                        const int bin = bin_of(k1, c & 1, kGrpM[sv.grp[st]][i]);
                        float wgt = 0.0f;
                        if (bin >= 0 && filt < kMel) {
                            wgt = float(Wt(filt, bin));
                            covered[size_t(filt) * nb + bin] = 1;
                        }
                        hi[i] = bf16_round(wgt);
                        lo[i] = bf16_round(wgt - bf16_val(hi[i]));
                    }
                    const size_t base = ((size_t(c) * 2 + b) * sv.n + st) * 2 * 256;
                    for (int d = 0; d < 4; ++d) {
                        abf[base + 0 * 256 + d * 64 + l] = hi[2 * d] | (hi[2 * d + 1] << 16);
                        abf[base + 1 * 256 + d * 64 + l] = lo[2 * d] | (lo[2 * d + 1] << 16);
                    }
                }
    auto R = [&](int role, int idx, int lane) -> float & { return arole[(size_t(role) * kPcArole + idx) * 64 + lane]; };
    // role 2 -- DCT rows: lane (coeff = l & 15, g = l >> 4) holds D[16 tile + coeff][16 blk + 4 g + r]
    std::vector<double> dd = dct_rows(n_cep, kMel, lifter);                   // [n_cep][40]
    for (int tile = 0; tile < 3; ++tile)
        for (int blk = 0; blk < kBlocks; ++blk)
            for (int r = 0; r < 4; ++r)
                for (int l = 0; l < 64; ++l) {
                    const int coeff = 16 * tile + (l & 15), filt = 16 * blk + 4 * (l >> 4) + r;
                    const float v = (coeff < n_cep && filt < kMel) ? float(dd[size_t(coeff) * kMel + filt]) : 0.0f;
                    if (tile == 0) R(2, 4 * blk + r, l) = v;
                    else adct[(size_t(tile - 1) * 12 + 4 * blk + r) * 64 + l] = v;
                }
    // roles 0, 1 -- column 16: X[16 + 32 j'] = sum_n2 v[n2] W1024^(n2 (16 + 32 j')), j' = 8 mb + 0..7.  MFMA row i = 4 g + r
    // holds r = 0: Re j' = 8 mb + 2 g, r = 1: Im, r = 2: Re j' + 1, r = 3: Im; K step t covers n2 = 4 t + (l >> 4)
    for (int mb = 0; mb < 2; ++mb) {
        for (int t = 0; t < 8; ++t)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, n2 = 4 * t + (l >> 4);
                const int g = i >> 2, r = i & 3, jp = 8 * mb + 2 * g + (r >> 1);
                const double th = 2.0 * kPi * double(n2 * (16 + 32 * jp)) / 1024.0;
                R(mb, t, l) = float((r & 1) ? -std::sin(th) : std::cos(th));
            }
        for (int blk = 0; blk < kBlocks; ++blk)
            for (int step = 0; step < 2; ++step)
                for (int l = 0; l < 64; ++l) {
                    const int filt = blk * 16 + (l & 15), bin = 16 + 32 * (8 * mb + 2 * (l >> 4) + step);
                    if (filt >= kMel || !sv.c16[mb][blk]) continue;
                    R(mb, 8 + 2 * blk + step, l) = float(Wt(filt, bin));
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
    t.a_role = f;     f += 3 * kPcArole * 64;
    t.a_dct_hi = f;   f += 2 * 12 * 64;
    t.a_bf = reinterpret_cast<const uint32_t *>(f);
}

// ---- device

// lane u (0..511) takes piece u and, u < 257, piece 512 + u of the window, plus the dword in front of each
__device__ __forceinline__ void fetch_window16(const mfcc_k::StreamDesc &s, const Window &w, int u, Fetch &f) {
    if (w.inside) {
        const i32x4 *g = reinterpret_cast<const i32x4 *>(w.ptr - w.shift);
        const int *g32 = reinterpret_cast<const int *>(g);
        f.v0 = g[u];
        f.p0 = g32[4 * u - 1];
        f.v1 = (i32x4){0, 0, 0, 0};
        f.p1 = 0;
        if (u < kPcSecond) {
            f.v1 = g[kPcFetchers + u];
            f.p1 = g32[4 * (kPcFetchers + u) - 1];
        }
    } else {
        const long long first = (long long)w.t_in * kPcTileHop;
        const int16_t *base = w.ptr - first;
        int h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long long i = first + (k < 8 ? 0 : 8 * kPcFetchers) + 8 * u + (k & 7);
            h[k] = mfcc_k::sample_at_i(s, base, i) & 0xFFFF;
        }
        f.v0 = (i32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        f.v1 = (i32x4){h[8] | (h[9] << 16), h[10] | (h[11] << 16), h[12] | (h[13] << 16), h[14] | (h[15] << 16)};
        f.p0 = mfcc_k::sample_at_i(s, base, first + 8 * u - 1) << 16;
        f.p1 = mfcc_k::sample_at_i(s, base, first + 8 * (kPcFetchers + u) - 1) << 16;
    }
}

__device__ __forceinline__ void park_window16(float *Sf, int u, const Fetch &f) {
    preemph8(f.p0, f.v0, Sf + 8 * u);
    if (u < kPcSecond) preemph8(f.p1, f.v1, Sf + 8 * (kPcFetchers + u));
}

// summed mel energies of a finished tile (the four consumers' partial sums) and their log2; register r of block b is
// filter 16 b + 4 q + r of frame lo.  Filters 40..47 do not exist: their (zero) sums must not reach the DCT as -inf * 0
__device__ __forceinline__ void mel_log2(const float *Q, int lane, int q, f32x4 (&lm)[kBlocks]) {
    const f32x4 *Qa = reinterpret_cast<const f32x4 *>(Q) + lane;
#pragma unroll
    for (int b = 0; b < kBlocks; ++b) {
        const f32x4 m = (Qa[(0 * kBlocks + b) * 64] + Qa[(1 * kBlocks + b) * 64]) +
                        (Qa[(2 * kBlocks + b) * 64] + Qa[(3 * kBlocks + b) * 64]);
#pragma unroll
        for (int r = 0; r < 4; ++r) lm[b][r] = __builtin_amdgcn_logf(m[r]);
    }
    if (q >= 2) lm[2] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// the tail of a tile: log2, DCT-II, store (16 frames = the 16 MFMA columns); coefficients 16..31 and 32..39 are further M
// tiles whose A operands are fetched here (uniform branches; 12 coalesced dwords per lane out of L2)
__device__ __forceinline__ void tail(const mfcc_k::StreamDesc &s, const Tables &t, const float *Q, const float (&ax)[kPcArole],
                                     const Cursor &c, int lane, float *__restrict__ out) {
    const int lo = lane & 15, q = lane >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 lm[kBlocks];
    mel_log2(Q, lane, q, lm);
    f32x4 d[kBlocks] = {zero, zero, zero};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < kBlocks; ++b) d[b] = MFCC1K8_MFMA(ax[4 * b + r], lm[b][r], d[b]);
    const long long fr0 = (long long)c.t_in * kPcTile;
    const long long rows_left = s.frames_per_ch - fr0;
    float *o = out + ((long long)c.ch * s.frames_per_ch + fr0) * t.n_cep + lo * t.n_cep + 4 * q;
    const bool mine = lo < rows_left;
    if (mine) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < t.n_cep) o[r] = (d[0][r] + d[1][r]) + d[2][r];
    }
    for (int tile = 1; 16 * tile < t.n_cep; ++tile) {
        const float *hi = t.a_dct_hi + (size_t)(tile - 1) * 12 * 64 + lane;
        asm volatile("" : "+v"(hi));
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

// Diagnostic build only (-DMFCC_F1KPC_STAMPS): per wave, clocks of interval A's work, the wait at barrier 1, interval B's
// work, the wait at barrier 2 -- summed over workgroups into a buffer nothing else reads.
#ifdef MFCC_F1KPC_STAMPS
__device__ unsigned long long g_stampspc[kPcWaves * 5];
#define PC_ST_BEGIN unsigned long long pst[4] = {0, 0, 0, 0}, pst_prev = __builtin_amdgcn_s_memtime(); unsigned long long pst_n = 0;
#define PC_ST(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now__ = __builtin_amdgcn_s_memtime(); pst[i] += now__ - pst_prev; pst_prev = now__; } while (0)
#define PC_ST_END do { if (lane == 0) { for (int i = 0; i < 4; ++i) atomicAdd(&g_stampspc[wave * 5 + i], pst[i]); atomicAdd(&g_stampspc[wave * 5 + 4], pst_n); } } while (0)
#else
#define PC_ST_BEGIN
#define PC_ST(i)
#define PC_ST_END
#endif

#ifndef F1KPC_PRIO_P
#define F1KPC_PRIO_P 0          // s_setprio of the producers
#endif

template <int VAR>
__global__ __launch_bounds__(64 * kPcWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_fused1024_kernel(mfcc_k::StreamDesc s, Tables t, LaunchGeom g, float *__restrict__ out) {
    using S = Sets<VAR>;
    constexpr int NS = S::N;
    __shared__ __attribute__((aligned(16))) float lds[kPcLdsWords];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    float *const Tt = lds;                                         // [16 frames][1028]: [16 k1][64] each
    auto Sb = [&](int b) { return lds + kPcTile * kPcTFrame + b * kPcSUsed; };                               // sample windows
    auto Qb = [&](int b) { return lds + kPcTile * kPcTFrame + 2 * kPcSUsed + b * kPcQWords; };               // partial mel sums
    auto Vb = [&](int b) { return lds + kPcTile * kPcTFrame + 2 * kPcSUsed + 2 * kPcQWords + b * kPcVWords; };   // column 16

    // this workgroup's tiles: virtual workgroup bid, stride = the grid.  XCD-aware order (kernel_fused512_w12.hpp):
    // consecutive tiles, whose windows overlap, on one XCD's L2
    const unsigned nwg = gridDim.x;
    const unsigned bid = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    const int n_tiles = g.tiles_per_ch * g.n_ch;                    // < 2^31 (host check)
    const int n_mine = (int)bid < n_tiles ? (n_tiles - (int)bid + (int)nwg - 1) / (int)nwg : 0;

    auto cursor_of = [&](unsigned v) {
        Cursor c;
        c.ch = (int)(v / (unsigned)g.tiles_per_ch);
        c.t_in = (int)(v - (unsigned)c.ch * (unsigned)g.tiles_per_ch);
        c.ptr = s.pcm + (long long)c.ch * s.ch_stride + (long long)c.t_in * kPcTileHop;
        return c;
    };
    Cursor cf = cursor_of(bid);        // the next tile to fetch
    Fetch fx;
    int sh0 = 0, sh1 = 0;              // alignment shifts of the windows parked in S[0], S[1]
    int sh_fetched = 0;

    // ---- prologue: S[0] <- window 0; fetch window 1
    if (n_mine > 0) {
        const Window w0 = window_of(cf, g);
        fetch_window16(s, w0, tid, fx);
        park_window16(Sb(0), tid, fx);
        sh0 = w0.shift;
        advance(cf, g);
    }
    if (n_mine > 1) {
        const Window w1 = window_of(cf, g);
        fetch_window16(s, w1, tid, fx);
        sh_fetched = w1.shift;
        advance(cf, g);
    }
    __syncthreads();                                               // S[0] is parked

    if (wave < 4) {
        // =========================================================================== producers
        if (F1KPC_PRIO_P) __builtin_amdgcn_s_setprio(F1KPC_PRIO_P);
        const int n2 = lane & 31;
        v2f wp[16], tw[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) wp[i] = reinterpret_cast<const v2f *>(t.win)[n2 * 16 + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) tw[i] = reinterpret_cast<const v2f *>(t.tw)[n2 * 16 + i];
        const int fr0 = 4 * wave + (lane >> 5), fr1 = fr0 + 2;      // this lane's frames of the two batches
        v2f ty0[16], ty1[16];
        float y0 = 0.f, y1 = 0.f;
        auto produce = [&](const float *Sf, int shift) {
            v2f ep0[16], ep1[16];
            const float *sp0 = Sf + fr0 * kHop + n2 + shift, *sp1 = Sf + fr1 * kHop + n2 + shift;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) ep0[n1 >> 1][n1 & 1] = sp0[32 * n1];
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) ep1[n1 >> 1][n1 & 1] = sp1[32 * n1];      // in flight during the first FFT
            mfcc_codelets::rfft32_tw(ep0, wp, tw, ty0, y0);
            mfcc_codelets::rfft32_tw(ep1, wp, tw, ty1, y1);
        };
        auto store_t = [&](float *V) {
            v2f *c0 = reinterpret_cast<v2f *>(Tt + fr0 * kPcTFrame) + n2;              // a store's lanes are consecutive n2
            v2f *c1 = reinterpret_cast<v2f *>(Tt + fr1 * kPcTFrame) + n2;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                c0[k1 * (kPcTRow / 2)] = ty0[k1];
                c1[k1 * (kPcTRow / 2)] = ty1[k1];
            }
            V[fr0 * kVStride + n2] = y0;
            V[fr1 * kVStride + n2] = y1;
        };
        if (n_mine > 0) produce(Sb(0), sh0);
        if (n_mine > 1) {
            park_window16(Sb(1), tid, fx);
            sh1 = sh_fetched;
        }
        if (n_mine > 0) store_t(Vb(0));
        lds_barrier();                                             // T(0), V(0) and S[1] are in LDS
        PC_ST_BEGIN
        for (int k = 0; k < n_mine; ++k) {
            // ---------------- interval A: tile k + 1 through pass 1, into registers
            const bool fetch2 = k + 2 < n_mine;
            if (fetch2) {
                const Window wn = window_of(cf, g);
                fetch_window16(s, wn, tid, fx);
                sh_fetched = wn.shift;
                advance(cf, g);
            }
            const bool have_next = k + 1 < n_mine;
            if (have_next) produce(Sb((k + 1) & 1), ((k + 1) & 1) ? sh1 : sh0);
            PC_ST(0);
            lds_barrier();                                         // 1: the consumers are done with T(k)
            PC_ST(1);
            // ---------------- interval B: registers -> T, V; park the window fetched during A
            if (have_next) store_t(Vb((k + 1) & 1));
            if (fetch2) {
                park_window16(Sb(k & 1), tid, fx);
                if (k & 1) sh1 = sh_fetched;
                else sh0 = sh_fetched;
            }
            PC_ST(2);
            lds_barrier();                                         // 2
            PC_ST(3);
#ifdef MFCC_F1KPC_STAMPS
            ++pst_n;
#endif
        }
        PC_ST_END;
    } else {
        // =========================================================================== consumers
        const int c = wave - 4;
        const int h = c & 1;
        const int lo = lane & 15, q = lane >> 4;
        u32x4 ah[2][NS], al[2][NS];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int st = 0; st < NS; ++st)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const size_t base = (((size_t)c * 2 + b) * NS + st) * 2 * 256;
                    ah[b][st][d] = t.a_bf[base + 0 * 256 + d * 64 + lane];
                    al[b][st][d] = t.a_bf[base + 1 * 256 + d * 64 + lane];
                }
        float ax[kPcArole];
#pragma unroll
        for (int i = 0; i < kPcArole; ++i) ax[i] = c < 3 ? t.a_role[(c * kPcArole + i) * 64 + lane] : 0.0f;
        Cursor co = cursor_of(bid);    // the tile whose rows the tail stores next (wave 6)

        if (n_mine > 1) {
            park_window16(Sb(1), tid, fx);
            sh1 = sh_fetched;
        }
        lds_barrier();                                             // T(0), V(0) and S[1] are in LDS
        PC_ST_BEGIN
        for (int k = 0; k < n_mine; ++k) {
            // ---------------- interval A: tile k through pass 2 and the mel contraction
            const bool fetch2 = k + 2 < n_mine;
            if (fetch2) {
                const Window wn = window_of(cf, g);
                fetch_window16(s, wn, tid, fx);
                sh_fetched = wn.shift;
                advance(cf, g);
            }
            f32x4 acc[NS];
#pragma unroll
            for (int st = 0; st < NS; ++st) acc[st] = zero;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float pw[16];
                {
                    v2f xl[16], xh[16], z[16];
                    const f32x4 *trow = reinterpret_cast<const f32x4 *>(Tt + lo * kPcTFrame + (8 * (c >> 1) + 4 * b + q) * kPcTRow);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const f32x4 a = trow[i], bb = trow[8 + i];
                        xl[2 * i] = (v2f){a[0], a[1]};
                        xl[2 * i + 1] = (v2f){a[2], a[3]};
                        xh[2 * i] = (v2f){bb[0], bb[1]};
                        xh[2 * i + 1] = (v2f){bb[2], bb[3]};
                    }
                    if (h) mfcc_codelets::cfft32_h1(xl, xh, z);
                    else mfcc_codelets::cfft32_h0(xl, xh, z);
#pragma unroll
                    for (int m = 0; m < 16; ++m) pw[m] = fmaf(z[m].x, z[m].x, z[m].y * z[m].y);
                }
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
                // term-major: consecutive MFMAs never share an accumulator
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int st = 0; st < NS; ++st) {
                        const u32x4 &a = term == 2 ? al[b][st] : ah[b][st];
                        const u32x4 &bo = term == 1 ? pl[S::grp[st]] : ph[S::grp[st]];
                        acc[st] = MFCC1K8_MFMA_BF(a, bo, acc[st]);
                    }
            }
            f32x4 fin[kBlocks] = {zero, zero, zero};
#pragma unroll
            for (int st = 0; st < NS; ++st) fin[S::blk[st]] += acc[st];
            PC_ST(0);
            lds_barrier();                                         // 1: T(k) is consumed
            PC_ST(1);
            // ---------------- interval B: column 16 of tile k, partial sums to Q, tail of tile k - 1, park
            if (c < 2) {
                // column 16 -> bins 16 + 32 j', j' = 8 c + 2 q + {0, 1}, fed to the filter blocks from registers
                const float *vp = Vb(k & 1) + lo * kVStride + q;
                f32x4 sp = zero, sp2 = zero;
#pragma unroll
                for (int tt = 0; tt < 8; tt += 2) {
                    sp = MFCC1K8_MFMA(ax[tt], vp[4 * tt], sp);
                    sp2 = MFCC1K8_MFMA(ax[tt + 1], vp[4 * (tt + 1)], sp2);
                }
                sp += sp2;
                const float c0 = fmaf(sp[0], sp[0], sp[1] * sp[1]);
                const float c1 = fmaf(sp[2], sp[2], sp[3] * sp[3]);
                if (c == 0) {
#pragma unroll
                    for (int b = 0; b < kBlocks; ++b)
                        if (S::c16[0][b]) {
                            fin[b] = MFCC1K8_MFMA(ax[8 + 2 * b], c0, fin[b]);
                            fin[b] = MFCC1K8_MFMA(ax[9 + 2 * b], c1, fin[b]);
                        }
                } else {
#pragma unroll
                    for (int b = 0; b < kBlocks; ++b)
                        if (S::c16[1][b]) {
                            fin[b] = MFCC1K8_MFMA(ax[8 + 2 * b], c0, fin[b]);
                            fin[b] = MFCC1K8_MFMA(ax[9 + 2 * b], c1, fin[b]);
                        }
                }
            }
            float *Q = Qb(k & 1);
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) *reinterpret_cast<f32x4 *>(Q + ((c * kBlocks + b) * 64 + lane) * 4) = fin[b];
            if (c == 2 && k > 0) {
                tail(s, t, Qb((k - 1) & 1), ax, co, lane, out);
                advance(co, g);
            }
            if (fetch2) {
                park_window16(Sb(k & 1), tid, fx);
                if (k & 1) sh1 = sh_fetched;
                else sh0 = sh_fetched;
            }
            PC_ST(2);
            lds_barrier();                                         // 2: Q(k), T(k + 1), V(k + 1), S(k + 2) are in LDS
            PC_ST(3);
#ifdef MFCC_F1KPC_STAMPS
            ++pst_n;
#endif
        }
        PC_ST_END;
        if (c == 2 && n_mine > 0) tail(s, t, Qb((n_mine - 1) & 1), ax, co, lane, out);
    }
}

inline const char *kernel_name() { return "mfcc_fused1024_kernel"; }

inline bool launch(const mfcc_k::StreamDesc &s, const Tables &t, float *out, int n_cu, hipStream_t stream) {
    const long long tiles_per_ch = (s.frames_per_ch + kPcTile - 1) / kPcTile;
    const long long n_ch = s.total_frames / s.frames_per_ch;
    const long long n_tiles = tiles_per_ch * n_ch;
    if (n_tiles >= (1ll << 30) || tiles_per_ch >= (1ll << 26) || n_ch >= (1ll << 30)) return false;
    long long grid = n_tiles < (long long)n_cu ? n_tiles : (long long)n_cu;      // one workgroup per CU (144 KB of LDS)
    if (grid < 1) grid = 1;
    LaunchGeom g;
    g.tiles_per_ch = (int)tiles_per_ch;
    g.n_ch = (int)n_ch;
    g.grid_div = (int)(grid / tiles_per_ch);
    g.grid_mod = (int)(grid % tiles_per_ch);
    g.step_ptr = (long long)g.grid_div * s.ch_stride + (long long)g.grid_mod * kPcTileHop;
    g.wrap_ptr = s.ch_stride - tiles_per_ch * (long long)kPcTileHop;
    g.t_lo = (int)((9 - (long long)s.halo + kPcTileHop - 1) / kPcTileHop);
    if (g.t_lo < 0) g.t_lo = 0;
    const long long hi = (s.n_samples - kPcSUsed) / kPcTileHop;
    g.t_hi = s.n_samples < kPcSUsed ? -1 : (int)(hi < tiles_per_ch ? hi : tiles_per_ch);
    switch (t.variant) {
    case 1: hipLaunchKernelGGL(mfcc_fused1024_kernel<1>, dim3((unsigned)grid), dim3(64 * kPcWaves), 0, stream, s, t, g, out); break;
    case 2: hipLaunchKernelGGL(mfcc_fused1024_kernel<2>, dim3((unsigned)grid), dim3(64 * kPcWaves), 0, stream, s, t, g, out); break;
    default: hipLaunchKernelGGL(mfcc_fused1024_kernel<0>, dim3((unsigned)grid), dim3(64 * kPcWaves), 0, stream, s, t, g, out); break;
    }
    return true;
}

}  // namespace mfcc_f1kpc
