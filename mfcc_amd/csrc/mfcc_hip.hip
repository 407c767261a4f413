// libmfcc_hip.so -- C ABI (include/mfcc_hip.h) over the gfx950 MFCC kernels.
//
// Host side: parameter validation, table building (tables.hpp), device buffers, launches.
// There is NO CPU compute path in this library: without a HIP device mfcc_hip_create fails
// with MFCC_HIP_ERROR_NOT_FOUND (the host-only helpers -- frame counts, table dumps, error
// strings -- keep working so the CPU test-suite can check the host logic).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mfcc_hip.h"
#include "kernels_generic.hpp"
#include "kernel_fixed512.hpp"
#include "kernel_fused1024.hpp"
#include "kernel_fused1024_f32.hpp"
#include "kernel_fused1024_w12.hpp"
#include "kernel_fused512.hpp"
#include "kernel_fused512_w12.hpp"
#include "tables.hpp"

namespace {

using namespace mfcc_tables;

struct Resolved {
    int nfft, hop, n_mel, n_cep, sample_rate, pad_mode;
    double power_scale, lifter;
    int device, float_impl;
};

int resolve(const mfcc_hip_params *p, Resolved &r) {
    if (!p) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (p->struct_size != sizeof(mfcc_hip_params)) return MFCC_HIP_ERROR_INVALID_PARAM;
    for (int v : p->reserved)
        if (v != 0) return MFCC_HIP_ERROR_INVALID_PARAM;
    r.nfft = p->nfft;
    r.hop = p->hop == 0 ? p->nfft / 3 : p->hop;
    r.n_mel = p->n_mel;
    r.n_cep = p->n_cep;
    r.sample_rate = p->sample_rate;
    r.pad_mode = p->pad_mode;
    r.power_scale = p->power_scale == 0.0f ? double(p->nfft) : double(p->power_scale);
    r.lifter = double(p->lifter);
    r.device = p->device;
    r.float_impl = p->float_impl;
    if (!is_pow2(r.nfft) || r.nfft < 64 || r.nfft > 1024) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (r.hop < 1 || r.hop > r.nfft) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (r.n_mel < 1 || r.n_mel > mfcc_k::kMaxMel) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (r.n_cep < 1 || r.n_cep > r.n_mel) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (r.sample_rate < 1) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (r.pad_mode != MFCC_HIP_PAD_NOTEBOOK && r.pad_mode != MFCC_HIP_PAD_STREAM)
        return MFCC_HIP_ERROR_INVALID_PARAM;
    if (!(r.power_scale > 0.0) || r.lifter < 0.0) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (r.float_impl < MFCC_HIP_IMPL_AUTO || r.float_impl > MFCC_HIP_IMPL_FUSED512)
        return MFCC_HIP_ERROR_INVALID_PARAM;
    return MFCC_HIP_SUCCESS;
}

size_t count_frames(const Resolved &r, size_t n) {
    if (r.pad_mode == MFCC_HIP_PAD_NOTEBOOK) {
        if (n < size_t(r.nfft)) return 0;
        return (n - size_t(r.nfft)) / size_t(r.hop) + 1;
    }
    if (n < size_t(r.nfft)) return 1;
    return (n - size_t(r.nfft)) / size_t(r.hop) + 2;
}

bool fixed_supported(const Resolved &r) {
    // RTL constraints: FFT sizes are powers of two (mfcc/misc/fft.py:351-353) for both the
    // nfft-point FFT and the (4 * nfilters)-point DCT FFT; hop = nfft // 3 (mfcc/core/mfcc.py:43)
    if (!(is_pow2(4 * r.n_mel) && 4 * r.n_mel <= r.nfft && r.n_mel >= 4 && r.hop == r.nfft / 3 && r.nfft >= 64))
        return false;
    // filter points too dense for the streaming filterbank's ramp logic (filterbank.py:22-34, 88-142): the RTL then
    // emits fewer than n_mel values per frame and the frame structure falls apart -- not a configuration to reproduce
    return fx_mel(r.nfft, r.n_mel, double(r.sample_rate)).n_out == r.n_mel;
}

struct SparseRows {
    std::vector<int> start, count, off;
};

template <typename T>
SparseRows pack_rows(const std::vector<T> &dense, int rows, int cols, std::vector<T> &packed) {
    SparseRows s;
    packed.clear();
    for (int r = 0; r < rows; ++r) {
        int lo = cols, hi = -1;
        for (int k = 0; k < cols; ++k)
            if (dense[size_t(r) * cols + k] != T(0)) {
                if (k < lo) lo = k;
                hi = k;
            }
        int cnt = hi >= lo ? hi - lo + 1 : 0;
        s.start.push_back(cnt ? lo : 0);
        s.count.push_back(cnt);
        s.off.push_back(int(packed.size()));
        for (int k = 0; k < cnt; ++k) packed.push_back(dense[size_t(r) * cols + lo + k]);
    }
    return s;
}

}  // namespace

struct mfcc_hip_handle {
    Resolved r;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int last_hip = 0;
    int n_cu = 0;
    bool fixed_ok = false;
    bool fused_ok = false;
    bool fused_dense = false;     // the fused kernel's banded MFMA list does not fit this sample rate: all pairs
    bool fused_w12 = false;       // the twelve-wave form of the fused 512 kernel runs (kernel_fused512_w12.hpp)
    bool fused1k_ok = false;      // the fused 1024/341/40 float kernel covers this handle's parameters
    bool fixed512_ok = false;     // the fused fixed-point kernel covers this handle's parameters
    // device tables (one arena)
    void *arena = nullptr;
    mfcc_k::FloatTables ft{};
    mfcc_k::FixedTables xt{};
    mfcc_fused::FusedTables fu{};
    mfcc_fixed512::Tables x5{};
    mfcc_fused1024::Tables f1k{};          // bf16-split contraction, set lists for every rate (kernel_fused1024.hpp)
    mfcc_fused1024_f32::Tables f1k_f32{};  // fp32 contraction, one MFMA list per rate (kernel_fused1024_f32.hpp)
    bool f1k_is_f32 = false;      // which of the two forms this handle's rate runs on
    bool f1k_w12 = false;         // the fp32 form's tables, staged as twelve waves (kernel_fused1024_w12.hpp)
    // descriptor tables of the ragged calls live in pinned host memory, two buffers used in turn: the H2D copy of
    // an asynchronous call reads buffer i while the next call fills buffer 1 - i; the call after that waits for the
    // event recorded behind buffer i's copy before it overwrites it
    struct PinnedDesc {
        long long *p = nullptr;
        size_t cap = 0;
        hipEvent_t copied = nullptr;
        bool in_flight = false;
    } desc[2];
    int desc_next = 0;
    // scratch for the host-buffer and ragged entry points.  It is reused by calls that may run on different
    // streams (mfcc_hip_set_stream): scratch_done is recorded behind every use and the next use on ANOTHER stream
    // waits for it on the device (an event outlives the stream it was recorded on)
    void *d_in = nullptr;
    size_t d_in_bytes = 0;
    void *d_out = nullptr;
    size_t d_out_bytes = 0;
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    // host-buffer pipeline (process_host): copy streams of their own, three chunks in flight
    static constexpr int kPipe = 3;
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[kPipe] = {}, ev_k[kPipe] = {}, ev_out[kPipe] = {};
    void *p_in[kPipe] = {}, *p_out[kPipe] = {};
    size_t p_in_bytes[kPipe] = {}, p_out_bytes[kPipe] = {};
    // streaming sessions opened on this handle that are still alive.  mfcc_hip_destroy with live sessions only marks
    // the handle; the last mfcc_hip_stream_destroy then tears it down (include/mfcc_hip.h: lifetime)
    int n_sessions = 0;
    bool destroy_pending = false;
};

namespace {

#define HIP_TRY(h, expr)                         \
    do {                                         \
        hipError_t e__ = (expr);                 \
        if (e__ != hipSuccess) {                 \
            (h)->last_hip = int(e__);            \
            return e__ == hipErrorOutOfMemory ? MFCC_HIP_ERROR_NO_MEM : MFCC_HIP_ERROR_OTHER; \
        }                                        \
    } while (0)

// Makes h's device current for the scope and puts the caller's device back afterwards (a process that drives
// several GPUs -- torch included -- must not find its current device changed by a library call).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// order a new use of the handle's scratch buffers behind the previous one when the stream has changed
int scratch_acquire(mfcc_hip_handle *h) {
    if (h->scratch_used && h->scratch_stream != h->stream)
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->scratch_done, 0));
    return MFCC_HIP_SUCCESS;
}

int scratch_release(mfcc_hip_handle *h) {
    HIP_TRY(h, hipEventRecord(h->scratch_done, h->stream));
    h->scratch_stream = h->stream;
    h->scratch_used = true;
    return MFCC_HIP_SUCCESS;
}

// a pinned descriptor buffer of at least n entries that no copy in flight still reads
int desc_acquire(mfcc_hip_handle *h, size_t n, mfcc_hip_handle::PinnedDesc **out) {
    mfcc_hip_handle::PinnedDesc &d = h->desc[h->desc_next];
    h->desc_next ^= 1;
    if (d.in_flight) {
        HIP_TRY(h, hipEventSynchronize(d.copied));
        d.in_flight = false;
    }
    if (d.cap < n) {
        if (d.p) HIP_TRY(h, hipHostFree(d.p));
        d.p = nullptr;
        d.cap = 0;
        const size_t want = n + n / 2 + 64;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&d.p), want * sizeof(long long), hipHostMallocDefault));
        d.cap = want;
    }
    if (!d.copied) HIP_TRY(h, hipEventCreateWithFlags(&d.copied, hipEventDisableTiming));
    *out = &d;
    return MFCC_HIP_SUCCESS;
}

struct Arena {
    std::vector<char> host;
    template <typename T>
    size_t put(const std::vector<T> &v) {
        size_t off = (host.size() + 255) & ~size_t(255);
        host.resize(off + v.size() * sizeof(T));
        if (!v.empty()) std::memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
        return off;
    }
};

int build_tables(mfcc_hip_handle *h) {
    const Resolved &r = h->r;
    Arena a;
    // ---- float
    std::vector<double> wd = hamming_periodic(r.nfft);
    std::vector<float> win(wd.begin(), wd.end());
    const int M = r.nfft / 2;
    std::vector<float2> twf(M), tws(M + 1);
    for (int m = 0; m < M; ++m) {
        double ang = -2.0 * kPi * double(m) / double(M);
        twf[m] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
    }
    for (int k = 0; k <= M; ++k) {
        double ang = -2.0 * kPi * double(k) / double(r.nfft);
        tws[k] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
    }
    std::vector<double> md = mel_dense(r.nfft, r.n_mel, double(r.sample_rate));
    const double inv_s2 = 1.0 / (r.power_scale * r.power_scale);
    std::vector<float> mdf(md.size());
    for (size_t i = 0; i < md.size(); ++i) mdf[i] = float(md[i] * inv_s2);
    std::vector<float> melw;
    SparseRows ms = pack_rows(mdf, r.n_mel, M + 1, melw);
    if (melw.empty()) melw.push_back(0.0f);
    std::vector<double> dd = dct_rows(r.n_cep, r.n_mel, r.lifter);
    std::vector<float> dct(dd.begin(), dd.end());

    bool dc_exact = false;                     // some filter has weight on the real-valued DC bin: that bin in double
    for (int f = 0; f < r.n_mel; ++f) dc_exact = dc_exact || md[size_t(f) * (M + 1)] != 0.0;
    size_t o_wd = dc_exact ? a.put(wd) : 0;
    size_t o_win = a.put(win), o_twf = a.put(twf), o_tws = a.put(tws);
    size_t o_ms = a.put(ms.start), o_mc = a.put(ms.count), o_mo = a.put(ms.off);
    size_t o_mw = a.put(melw), o_dct = a.put(dct);

    // ---- fixed
    size_t o_cv = 0, o_xt = 0, o_xd = 0, o_xs = 0, o_xc = 0, o_xo = 0, o_xw = 0;
    int x5_w_total = 0, x5_chunk = 0, x5_span = 0;
    bool x5_lanes_ok = false;
    std::vector<int> x5_lanes;
    std::vector<uint32_t> x5_wl;
    FxMel fm;
    h->fixed_ok = fixed_supported(r);
    if (h->fixed_ok) {
        std::vector<int> cv = fx_window_curve(r.nfft);
        std::vector<int> re, im;
        fx_twiddles(r.nfft, re, im);
        // twiddles as dot2 operand pairs: A = (twr, -twi), B = (twi, twr) (kernels_generic.hpp: fx_bfly)
        auto pack = [](int tre, int tim) {
            return make_uint2((uint32_t(tre) & 0xffffu) | (uint32_t(-tim) << 16), (uint32_t(tim) & 0xffffu) | (uint32_t(tre) << 16));
        };
        std::vector<uint2> t1(re.size());
        for (size_t i = 0; i < re.size(); ++i) t1[i] = pack(re[i], im[i]);
        fx_twiddles(4 * r.n_mel, re, im);
        std::vector<uint2> t2(re.size());
        for (size_t i = 0; i < re.size(); ++i) t2[i] = pack(re[i], im[i]);
        fm = fx_mel(r.nfft, r.n_mel, double(r.sample_rate));
        std::vector<uint32_t> xw;
        SparseRows xs = pack_rows(fm.dense, r.n_mel, r.nfft / 2, xw);
        if (xw.empty()) xw.push_back(0u);
        o_cv = a.put(cv); o_xt = a.put(t1); o_xd = a.put(t2);
        o_xs = a.put(xs.start); o_xc = a.put(xs.count); o_xo = a.put(xs.off); o_xw = a.put(xw);
        x5_w_total = (int)xw.size();
        x5_lanes_ok = mfcc_fixed512::build_mel_lanes(xs.start, xs.count, xs.off, xw, x5_lanes, x5_wl, x5_chunk, x5_span);
    }
    // ---- fused fixed-point kernel (the RTL's own configuration)
    std::vector<char> x5_blob;
    uint32_t x5_tw[4] = {0, 0, 0, 0};
    h->fixed512_ok = h->fixed_ok && mfcc_fixed512::supported(r.nfft, r.n_mel, r.n_cep) && x5_lanes_ok &&
                     mfcc_fixed512::build_tables(r.n_mel, x5_blob, x5_tw);
    size_t o_x5 = 0, o_x5l = 0, o_x5w = 0;
    if (h->fixed512_ok) {
        o_x5 = a.put(x5_blob);
        o_x5l = a.put(x5_lanes);
        o_x5w = a.put(x5_wl);
    }

    // ---- fused 512/170/32 kernel tables
    std::vector<char> fused_blob;
    h->fused_ok = false;
    h->fused_dense = false;
    bool fused_dcx = false;
    if (mfcc_fused::supported(r.nfft, r.hop, r.n_mel, r.n_cep)) {
        fused_dcx = mfcc_fused::needs_dc_exact(r.sample_rate, r.n_mel);    // only the dense instantiation has the DC path
        h->fused_ok = !fused_dcx &&
                      mfcc_fused::build_tables<false>(r.sample_rate, r.power_scale, r.lifter, r.n_cep, r.n_mel, fused_blob);
        if (!h->fused_ok) {
            h->fused_dense = true;
            h->fused_ok = mfcc_fused::build_tables<true>(r.sample_rate, r.power_scale, r.lifter, r.n_cep, r.n_mel, fused_blob);
        }
    }
    size_t o_fu = 0;
    if (h->fused_ok) o_fu = a.put(fused_blob);
    std::vector<char> f1k_blob;
    int f1k_var = 0;
    h->fused1k_ok = mfcc_fused1024::supported(r.nfft, r.hop, r.n_mel, r.n_cep);
    if (h->fused1k_ok) {
        // Default: the twelve-wave staging with the bf16-split contraction (kernel_fused1024_w12.hpp), every rate.
        // MFCC_HIP_FUSED1024 is a diagnostic override for A/B runs -- f32 / bf16: the eight-wave lockstep staging of
        // either contraction; w12 / w12bf: the twelve-wave staging of either (fp32: the five rates it has lists for)
        const char *e = std::getenv("MFCC_HIP_FUSED1024");
        auto is = [&](const char *v) { return e && !std::strcmp(e, v); };
        const bool no_f32 = !(is("f32") || is("w12")), no_bf16 = is("f32") || is("w12");
        h->f1k_w12 = !(is("f32") || is("bf16"));
        h->f1k_is_f32 = !no_f32 && mfcc_fused1024_f32::build_tables(r.sample_rate, r.power_scale, r.lifter, r.n_cep, f1k_blob, f1k_var);
        if (!h->f1k_is_f32)
            h->fused1k_ok = !no_bf16 && mfcc_fused1024::build_tables(r.sample_rate, r.power_scale, r.lifter, r.n_cep, f1k_blob, f1k_var);
    }
    size_t o_f1k = 0;
    if (h->fused1k_ok) o_f1k = a.put(f1k_blob);

    HIP_TRY(h, hipMalloc(&h->arena, a.host.size() + 256));
    HIP_TRY(h, hipMemcpy(h->arena, a.host.data(), a.host.size(), hipMemcpyHostToDevice));
    char *b = static_cast<char *>(h->arena);
    h->ft.window = reinterpret_cast<const float *>(b + o_win);
    h->ft.tw_fft = reinterpret_cast<const float2 *>(b + o_twf);
    h->ft.tw_split = reinterpret_cast<const float2 *>(b + o_tws);
    h->ft.mel_start = reinterpret_cast<const int *>(b + o_ms);
    h->ft.mel_count = reinterpret_cast<const int *>(b + o_mc);
    h->ft.mel_off = reinterpret_cast<const int *>(b + o_mo);
    h->ft.mel_w = reinterpret_cast<const float *>(b + o_mw);
    h->ft.mel_w_total = (int)melw.size();
    h->ft.dct = reinterpret_cast<const float *>(b + o_dct);
    h->ft.window_d = dc_exact ? reinterpret_cast<const double *>(b + o_wd) : nullptr;
    h->ft.n_mel = r.n_mel;
    h->ft.n_cep = r.n_cep;
    if (h->fixed_ok) {
        h->xt.curve = reinterpret_cast<const int *>(b + o_cv);
        h->xt.tw_fft = reinterpret_cast<const uint2 *>(b + o_xt);
        h->xt.tw_dct = reinterpret_cast<const uint2 *>(b + o_xd);
        h->xt.mel_start = reinterpret_cast<const int *>(b + o_xs);
        h->xt.mel_count = reinterpret_cast<const int *>(b + o_xc);
        h->xt.mel_off = reinterpret_cast<const int *>(b + o_xo);
        h->xt.mel_w = reinterpret_cast<const uint32_t *>(b + o_xw);
        h->xt.mel_w_total = x5_w_total;
        h->xt.mel_shift = fm.shift;
        h->xt.log2_mel = ilog2(r.n_mel);
        h->xt.nfft = r.nfft;
        h->xt.log2_nfft = ilog2(r.nfft);
        h->xt.n_mel = r.n_mel;
        h->xt.log2_dct = ilog2(4 * r.n_mel);
        h->xt.n_cep = r.n_cep;
    }
    if (h->fused_ok) mfcc_fused::bind_tables(b + o_fu, r.n_cep, r.n_mel, h->fused_dense, fused_dcx, h->fu);
    {
        // diagnostic override for A/B runs: MFCC_HIP_FUSED512=w4 keeps the four-wave form
        const char *e = std::getenv("MFCC_HIP_FUSED512");
        h->fused_w12 = h->fused_ok && !(e && std::strcmp(e, "w4") == 0);
    }
    if (h->fused1k_ok) {
        if (h->f1k_is_f32) mfcc_fused1024_f32::bind_tables(b + o_f1k, r.n_cep, f1k_var, h->f1k_f32);
        else mfcc_fused1024::bind_tables(b + o_f1k, r.n_cep, f1k_var, h->f1k);
    }
    if (h->fixed512_ok) {
        mfcc_fixed512::bind_tables(b + o_x5, h->x5);
        h->x5.tw64a = x5_tw[0]; h->x5.tw64b = x5_tw[1]; h->x5.tw192a = x5_tw[2]; h->x5.tw192b = x5_tw[3];
        h->x5.mel_lane = reinterpret_cast<const int4 *>(b + o_x5l);
        h->x5.mel_chunk = x5_chunk;
        h->x5.mel_span = x5_span;
        h->x5.mel_wl = reinterpret_cast<const uint32_t *>(b + o_x5w);
        h->x5.mel_shift = fm.shift;
        h->x5.n_cep = r.n_cep;
        h->x5.n_mel = r.n_mel;
    }
    return MFCC_HIP_SUCCESS;
}

bool use_fused(const mfcc_hip_handle *h) {
    if (h->r.float_impl == MFCC_HIP_IMPL_GENERIC) return false;
    return h->fused_ok;
}

int launch(mfcc_hip_handle *h, bool fixed, const void *d_pcm, size_t n, size_t stride, size_t nch,
           int halo, void *d_out, size_t *n_frames, size_t force_frames = 0) {
    if (!h || (!d_pcm && n * nch) || halo < 0 || halo > 1) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (fixed && !h->fixed_ok) return MFCC_HIP_ERROR_UNSUPPORTED;
    if (!fixed && h->r.float_impl == MFCC_HIP_IMPL_FUSED512 && !h->fused_ok)
        return MFCC_HIP_ERROR_UNSUPPORTED;
    // force_frames: the packed stream of the ragged entry points -- every hop position is a frame
    const size_t nf = force_frames ? force_frames : count_frames(h->r, n);
    if (n_frames) *n_frames = nf;
    if (nf == 0 || nch == 0) return MFCC_HIP_SUCCESS;
    if (!d_out) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (nch > 1 && stride < n + size_t(halo)) return MFCC_HIP_ERROR_INVALID_PARAM;

    mfcc_k::StreamDesc s;
    s.pcm = static_cast<const int16_t *>(d_pcm) + halo;
    s.ch_stride = (long long)stride;
    s.n_samples = (long long)n;
    s.halo = halo;
    s.frames_per_ch = (long long)nf;
    s.total_frames = (long long)(nf * nch);
    s.hop = h->r.hop;

    DeviceGuard guard(h->device);
    const long long total = s.total_frames;
    if (fixed && h->fixed512_ok) {
        mfcc_fixed512::launch(s, h->x5, static_cast<int16_t *>(d_out), h->n_cu, h->stream);
    } else if (fixed) {
        long long blocks = (total + mfcc_k::kWavesPerBlock - 1) / mfcc_k::kWavesPerBlock;
        long long cap = (long long)h->n_cu * 32;     // a deep queue of workgroups, not a merely full grid (kernel_fixed512.hpp: launch)
        if (blocks > cap) blocks = cap;
        size_t lds = size_t(mfcc_k::kWavesPerBlock) *
                         (size_t(h->r.nfft + h->r.nfft / 32) * sizeof(uint32_t) + size_t(h->r.nfft / 2) * 4 + mfcc_k::kMaxMel * 4) +
                     size_t((h->xt.mel_w_total + 1) & ~1) * sizeof(uint32_t) +            // filterbank weights,
                     (size_t(h->r.nfft / 2) + size_t(2 * h->r.n_mel)) * sizeof(uint2);     // both twiddle ROMs
        int16_t *o = static_cast<int16_t *>(d_out);
        switch (h->r.nfft) {
#define MFCC_FX_CASE(N)                                                                                            \
    case N:                                                                                                        \
        hipLaunchKernelGGL(mfcc_k::mfcc_fixed_kernel<N>, dim3((unsigned)blocks), dim3(mfcc_k::kBlock), lds, h->stream, \
                           s, h->xt, o);                                                                           \
        break;
            MFCC_FX_CASE(64)
            MFCC_FX_CASE(128)
            MFCC_FX_CASE(256)
            MFCC_FX_CASE(512)
            MFCC_FX_CASE(1024)
#undef MFCC_FX_CASE
            default:
                return MFCC_HIP_ERROR_UNSUPPORTED;
        }
    } else if (use_fused(h)) {
        const bool done = h->fused_w12 &&
                          mfcc_fused12::launch(s, h->fu, h->fused_dense, static_cast<float *>(d_out), h->n_cu, h->stream);
        if (!done && !mfcc_fused::launch(s, h->fu, h->fused_dense, static_cast<float *>(d_out), h->n_cu, h->stream))
            return MFCC_HIP_ERROR_UNSUPPORTED;
    } else if (h->fused1k_ok && h->r.float_impl == MFCC_HIP_IMPL_AUTO &&
               (h->f1k_is_f32 ? ((h->f1k_w12 && mfcc_fused1024_w12::launch(s, h->f1k_f32, static_cast<float *>(d_out), h->n_cu, h->stream)) ||
                                 mfcc_fused1024_f32::launch(s, h->f1k_f32, static_cast<float *>(d_out), h->n_cu, h->stream))
                              : ((h->f1k_w12 && mfcc_fused1024_w12bf::launch(s, h->f1k, static_cast<float *>(d_out), h->n_cu, h->stream)) ||
                                 mfcc_fused1024::launch(s, h->f1k, static_cast<float *>(d_out), h->n_cu, h->stream)))) {
        // fused 1024/341/40 kernel launched
    } else {
        long long blocks = (total + mfcc_k::kWavesPerBlock - 1) / mfcc_k::kWavesPerBlock;
        long long cap = (long long)h->n_cu * 128;    // measured: 8 per CU 3.55 ms, 32 3.24, 128 3.06 (nfft 256)
        if (blocks > cap) blocks = cap;
        float *o = static_cast<float *>(d_out);
        switch (h->r.nfft) {
            case 64:
                hipLaunchKernelGGL(mfcc_k::mfcc_float_generic_kernel<64>, dim3((unsigned)blocks),
                                   dim3(mfcc_k::kBlock), 0, h->stream, s, h->ft, o);
                break;
            case 128:
                hipLaunchKernelGGL(mfcc_k::mfcc_float_generic_kernel<128>, dim3((unsigned)blocks),
                                   dim3(mfcc_k::kBlock), 0, h->stream, s, h->ft, o);
                break;
            case 256:
                hipLaunchKernelGGL(mfcc_k::mfcc_float_generic_kernel<256>, dim3((unsigned)blocks),
                                   dim3(mfcc_k::kBlock), 0, h->stream, s, h->ft, o);
                break;
            case 512:
                hipLaunchKernelGGL(mfcc_k::mfcc_float_generic_kernel<512>, dim3((unsigned)blocks),
                                   dim3(mfcc_k::kBlock), 0, h->stream, s, h->ft, o);
                break;
            case 1024:
                hipLaunchKernelGGL(mfcc_k::mfcc_float_generic_kernel<1024>, dim3((unsigned)blocks),
                                   dim3(mfcc_k::kBlock), 0, h->stream, s, h->ft, o);
                break;
            default:
                return MFCC_HIP_ERROR_UNSUPPORTED;
        }
    }
    HIP_TRY(h, hipGetLastError());
    return MFCC_HIP_SUCCESS;
}

int ensure(mfcc_hip_handle *h, void **p, size_t *have, size_t want) {
    if (*have >= want && *p) return MFCC_HIP_SUCCESS;
    if (*p) {
        HIP_TRY(h, hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    size_t sz = want + want / 4 + 4096;
    HIP_TRY(h, hipMalloc(p, sz));
    *have = sz;
    return MFCC_HIP_SUCCESS;
}

// ---- host buffers in, host buffers out: the shape of the reference's own caller (software/main.c:100-177: file in,
// file out).  The input crosses PCIe at 340 B per frame, which is 25 times what the kernel needs per frame in time, so this
// path is a COPY pipeline: the batch is cut into chunks of ~64 MB (whole channels, or frame ranges of a long channel with
// a one-sample history halo), three chunks in flight -- H2D of chunk k + 1 on one copy stream, the kernel of chunk k on
// the handle's stream, D2H of chunk k - 1 on another.  An asynchronous copy from pageable memory blocks the calling thread
// until it is done (measured: hipMemcpyAsync of 154 MB returned after 8.6 ms); pinned in place with hipHostRegister (13 us
// per MB, done for chunk k + 1 while chunk k is on the wire) it returns in 20 us.  Round 2 ran copy, kernel, copy in series.
// Measured on 8 channels x 10 min (154 MB in, 23 MB out; tools/hostio_sweep.sh, median of 15): one chunk unpinned 3.42 ms,
// 64 MB chunks pinned 3.20 ms (0.141 G frames/s = 48 GB/s of input against 53 GB/s for the bare H2D copy), 32 MB 3.68,
// 16 MB 3.94 -- every chunk costs ~60 us of calls and a kernel launch that cannot fill the chip.  A batch that fits one
// chunk takes the plain blocking copies (pinning it first only adds its 13 us per MB in front).
struct HostChunk {
    const int16_t *in;      // first sample handed to the kernel (the halo sample when halo = 1)
    size_t in_samples;      // samples copied (all channels of the chunk; incl. the halo sample)
    size_t n, stride, nch;  // launch geometry: samples per channel after the halo, channel stride, channels
    int halo;
    size_t frames;          // frames per channel of this chunk (forced: a frame range of a longer stream)
    size_t out_elems;       // coefficients written
    size_t out_off;         // offset into `out`, in elements
};

// A caller's buffer pinned block by block for the duration of a call: page-aligned blocks of 16 MB, registered when a
// copy first touches them and released when every chunk that could touch them is done.  A copy must stay inside ONE
// registration (the runtime resolves a host pointer to the allocation it lies in and rejects a size that runs past it),
// so copies are split at block boundaries.
struct PinnedSpan {
    static constexpr size_t kBlock = size_t(16) << 20;
    char *base = nullptr;
    size_t nblocks = 0, total = 0;
    std::vector<char> state;      // 0: untouched, 1: pinned by us, 2: tried, not ours (already pinned or not pinnable)
    void init(const void *p, size_t bytes) {
        const uintptr_t page = 4096, a = reinterpret_cast<uintptr_t>(p) & ~(page - 1);
        const uintptr_t b = (reinterpret_cast<uintptr_t>(p) + bytes + page - 1) & ~(page - 1);
        base = reinterpret_cast<char *>(a);
        total = size_t(b - a);
        nblocks = (total + kBlock - 1) / kBlock;
        state.assign(nblocks, 0);
    }
    size_t block_bytes(size_t k) const { return std::min(kBlock, total - k * kBlock); }
    void ensure(const void *p, size_t bytes) {
        if (!bytes) return;
        const size_t k0 = size_t(static_cast<const char *>(p) - base) / kBlock;
        const size_t k1 = size_t(static_cast<const char *>(p) + bytes - 1 - base) / kBlock;
        for (size_t k = k0; k <= k1 && k < nblocks; ++k)
            if (!state[k]) {
                state[k] = hipHostRegister(base + k * kBlock, block_bytes(k), hipHostRegisterDefault) == hipSuccess ? 1 : 2;
                if (state[k] == 2) (void)hipGetLastError();
            }
    }
    // blocks that end at or below p are no longer needed
    void release_below(const void *p) {
        for (size_t k = 0; k < nblocks && base + k * kBlock + block_bytes(k) <= static_cast<const char *>(p); ++k)
            if (state[k] == 1) {
                (void)hipHostUnregister(base + k * kBlock);
                state[k] = 2;
            }
    }
    void release_all() {
        for (size_t k = 0; k < nblocks; ++k)
            if (state[k] == 1) {
                (void)hipHostUnregister(base + k * kBlock);
                state[k] = 2;
            }
    }
    // hipMemcpyAsync in pieces that do not cross a block boundary
    hipError_t copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st) const {
        const char *host = static_cast<const char *>(kind == hipMemcpyHostToDevice ? src : dst);
        size_t done = 0;
        while (done < bytes) {
            const size_t k = size_t(host + done - base) / kBlock;
            const size_t room = size_t(base + k * kBlock + block_bytes(k) - (host + done));
            const size_t len = std::min(bytes - done, room);
            const hipError_t e = hipMemcpyAsync(static_cast<char *>(dst) + done, static_cast<const char *>(src) + done, len, kind, st);
            if (e != hipSuccess) return e;
            done += len;
        }
        return hipSuccess;
    }
};

// one chunk of a host-buffer call as the pipeline sees it: bytes in, bytes out
struct PipeChunk {
    const void *in;
    size_t in_bytes;
    void *out;
    size_t out_bytes;
};

// the pipeline itself: H2D of chunk i + 1 | enqueue(i, d_in, d_out) on the handle's stream | D2H of chunk i - 1.  The
// chunks must ascend in both host buffers ([in_lo, in_lo + in_total) and [out_lo, out_lo + out_total) are what gets pinned).
template <typename Enqueue>
int run_host_pipeline(mfcc_hip_handle *h, const std::vector<PipeChunk> &chunks, const void *in_lo, size_t in_total,
                      void *out_lo, size_t out_total, Enqueue &&enqueue) {
    constexpr int kPipe = mfcc_hip_handle::kPipe;
    const size_t nchunks = chunks.size();
    if (!nchunks) return MFCC_HIP_SUCCESS;
    bool no_pin = false;
    if (const char *e = std::getenv("MFCC_HIP_HOST_NOPIN")) no_pin = e[0] == '1';                  // diagnostic
    if (!h->s_in) HIP_TRY(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    if (!h->s_out) HIP_TRY(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    size_t max_in = 0, max_out = 0;
    for (const PipeChunk &c : chunks) {
        max_in = std::max(max_in, c.in_bytes);
        max_out = std::max(max_out, c.out_bytes);
    }
    const int nbuf = int(std::min<size_t>(kPipe, nchunks));
    for (int i = 0; i < nbuf; ++i) {
        if (!h->ev_in[i]) {
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_k[i], hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_out[i], hipEventDisableTiming));
        }
        int rc = ensure(h, &h->p_in[i], &h->p_in_bytes[i], max_in + 64);
        if (rc) return rc;
        if ((rc = ensure(h, &h->p_out[i], &h->p_out_bytes[i], max_out + 64))) return rc;
    }
    // the copy streams start behind whatever is queued on the handle's stream
    HIP_TRY(h, hipEventRecord(h->ev_k[0], h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->s_in, h->ev_k[0], 0));

    // a batch that fits one chunk takes the plain blocking copies: pinning it first only adds its cost in front
    const bool pin_it = !no_pin && nchunks > 1;
    PinnedSpan pin_in, pin_out;
    pin_in.init(in_lo, in_total);
    pin_out.init(out_lo, out_total);
    if (!pin_it) {
        pin_in.state.assign(pin_in.nblocks, 2);
        pin_out.state.assign(pin_out.nblocks, 2);
    }
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(h->s_in);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->s_out);
        pin_in.release_all();
        pin_out.release_all();
        return code;
    };
    // what lies below the next unfinished chunk is released as the pipeline moves on, what lies ahead is pinned one chunk
    // early (on the CPU, while the wire is busy)
    auto pin = [&](size_t i) {
        pin_in.ensure(chunks[i].in, chunks[i].in_bytes);
        pin_out.ensure(chunks[i].out, chunks[i].out_bytes);
    };
    pin(0);
#define PIPE_TRY(expr)                                   \
    do {                                                 \
        const hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) {                         \
            h->last_hip = int(e__);                      \
            return fail(MFCC_HIP_ERROR_OTHER);           \
        }                                                \
    } while (0)
    for (size_t i = 0; i < nchunks; ++i) {
        const PipeChunk &c = chunks[i];
        const int slot = int(i % kPipe);
        if (i + 1 < nchunks) pin(i + 1);
        if (i >= size_t(kPipe)) {
            // slot reuse: chunk i - kPipe's kernel has read p_in[slot] / its rows have left p_out[slot]
            PIPE_TRY(hipStreamWaitEvent(h->s_in, h->ev_k[slot], 0));
            PIPE_TRY(hipStreamWaitEvent(h->stream, h->ev_out[slot], 0));
            PIPE_TRY(hipEventSynchronize(h->ev_out[slot]));
            // chunks 0 .. i - kPipe are done (copies, kernel, rows): nothing touches what lies below the next one
            pin_in.release_below(chunks[i - kPipe + 1].in);
            pin_out.release_below(chunks[i - kPipe + 1].out);
        }
        if (c.in_bytes) PIPE_TRY(pin_in.copy(h->p_in[slot], c.in, c.in_bytes, hipMemcpyHostToDevice, h->s_in));
        PIPE_TRY(hipEventRecord(h->ev_in[slot], h->s_in));
        PIPE_TRY(hipStreamWaitEvent(h->stream, h->ev_in[slot], 0));
        const int rc = enqueue(i, h->p_in[slot], h->p_out[slot]);
        if (rc) return fail(rc);
        PIPE_TRY(hipEventRecord(h->ev_k[slot], h->stream));
        PIPE_TRY(hipStreamWaitEvent(h->s_out, h->ev_k[slot], 0));
        if (c.out_bytes) PIPE_TRY(pin_out.copy(c.out, h->p_out[slot], c.out_bytes, hipMemcpyDeviceToHost, h->s_out));
        PIPE_TRY(hipEventRecord(h->ev_out[slot], h->s_out));
    }
    PIPE_TRY(hipStreamSynchronize(h->s_out));
#undef PIPE_TRY
    // the handle's stream continues behind the last rows
    (void)hipStreamWaitEvent(h->stream, h->ev_out[int((nchunks - 1) % kPipe)], 0);
    pin_in.release_all();
    pin_out.release_all();
    return MFCC_HIP_SUCCESS;
}

inline size_t host_chunk_bytes() {
    size_t b = size_t(64) << 20;
    if (const char *e = std::getenv("MFCC_HIP_HOST_CHUNK_MB")) b = size_t(std::atoi(e) > 0 ? std::atoi(e) : 64) << 20;   // diagnostic
    return b;
}

template <typename OutT>
int process_host(mfcc_hip_handle *h, bool fixed, const int16_t *pcm, size_t n, size_t nch, OutT *out,
                 size_t cap, size_t *n_frames) {
    if (!h || (!pcm && n * nch)) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (fixed && !h->fixed_ok) return MFCC_HIP_ERROR_UNSUPPORTED;
    const size_t nf = count_frames(h->r, n);
    if (n_frames) *n_frames = nf;
    if (nf == 0 || nch == 0) return MFCC_HIP_SUCCESS;
    const size_t ncep = size_t(h->r.n_cep), hop = size_t(h->r.hop), nfft = size_t(h->r.nfft);
    const size_t n_out = nf * nch * ncep;
    if (!out || cap < n_out) return MFCC_HIP_ERROR_BUFFER_SMALL;
    DeviceGuard guard(h->device);
    const size_t kChunkBytes = host_chunk_bytes();

    // ---- the chunks
    std::vector<HostChunk> chunks;
    const size_t ch_bytes = n * sizeof(int16_t);
    if (ch_bytes <= kChunkBytes) {
        size_t per = kChunkBytes / (ch_bytes ? ch_bytes : 1);
        if (per < 1) per = 1;
        for (size_t c0 = 0; c0 < nch; c0 += per) {
            const size_t k = std::min(per, nch - c0);
            chunks.push_back({pcm + c0 * n, k * n, n, n, k, 0, nf, k * nf * ncep, c0 * nf * ncep});
        }
    } else {
        // long channels: frame ranges [f0, f1) of one channel -- samples f0 * hop - 1 (history) .. (f1 - 1) * hop + nfft
        size_t per = (kChunkBytes / sizeof(int16_t)) / hop;
        if (per < 1) per = 1;
        for (size_t c = 0; c < nch; ++c)
            for (size_t f0 = 0; f0 < nf; f0 += per) {
                const size_t f1 = std::min(nf, f0 + per), halo = f0 ? 1 : 0;
                const size_t s0 = f0 * hop - halo;
                size_t s1 = (f1 - 1) * hop + nfft;
                if (s1 > n) s1 = n;                                  // the zero-padded tail of STREAM framing
                chunks.push_back({pcm + c * n + s0, s1 - s0, s1 - s0 - halo, s1 - s0, 1, int(halo), f1 - f0,
                                  (f1 - f0) * ncep, (c * nf + f0) * ncep});
            }
    }
    std::vector<PipeChunk> pc;
    pc.reserve(chunks.size());
    for (const HostChunk &c : chunks)
        pc.push_back({c.in, c.in_samples * sizeof(int16_t), out + c.out_off, c.out_elems * sizeof(OutT)});
    return run_host_pipeline(h, pc, pcm, nch * ch_bytes, out, n_out * sizeof(OutT), [&](size_t i, void *d_in, void *d_out) {
        const HostChunk &c = chunks[i];
        return launch(h, fixed, d_in, c.n, c.stride, c.nch, c.halo, d_out, nullptr, c.halo || c.frames != nf ? c.frames : 0);
    });
}

// ---- ragged batch: many utterances of different lengths, one launch.
// The utterances are packed into ONE stream at offsets that are multiples of the hop, separated by
// zeros: at least one zero in front (pre-emphasis history 0, like after the driver's soft reset,
// main.c:21-34) and zeros behind up to the end of the last frame (the zero-padded tail of main.c:134-144).
// Frame k of utterance u is then frame start_u / hop + k of the packed stream -- the same samples, the
// same arithmetic, bit for bit -- and the frames that fall into the gaps are simply not copied out.
template <typename OutT>
__global__ void gather_rows_kernel(const OutT *__restrict__ src, OutT *__restrict__ dst,
                                   const long long *__restrict__ desc, long long n_utt, int row) {
    for (long long u = blockIdx.x; u < n_utt; u += gridDim.x) {
        const long long s0 = desc[3 * u] * row, d0 = desc[3 * u + 1] * row, n = desc[3 * u + 2] * row;
        for (long long i = threadIdx.x; i < n; i += blockDim.x) dst[d0 + i] = src[s0 + i];
    }
}

// device-resident input: copy every utterance to its place in the packed stream and zero what lies between it and
// the next one (no memset of the whole stream: every sample of it is written exactly once).  16 bytes per lane:
// the destination is brought to 16-byte alignment first, the source is read as it lies (2-byte aligned; global
// memory takes unaligned vector loads)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) Unaligned16 {
    u32x4_t v;
};

__global__ void pack_utterances_kernel(const int16_t *__restrict__ src, int16_t *__restrict__ dst,
                                       const long long *__restrict__ desc, long long n_utt) {
    // desc: (source offset, destination offset, samples, end of this utterance's part of the stream) per utterance
    for (long long u = blockIdx.x; u < n_utt; u += gridDim.x) {
        const long long n = desc[4 * u + 2], span = desc[4 * u + 3] - desc[4 * u + 1];
        if (span <= 0) continue;
        const int16_t *s0 = src + desc[4 * u];
        int16_t *d0 = dst + desc[4 * u + 1];
        long long head = (long long)(((16 - (reinterpret_cast<uintptr_t>(d0) & 15)) & 15) >> 1);
        if (head > n) head = n;
        for (long long i = threadIdx.x; i < head; i += blockDim.x) d0[i] = s0[i];
        const long long nvec = (n - head) >> 3;
        u32x4_t *dv = reinterpret_cast<u32x4_t *>(d0 + head);
        const Unaligned16 *sv = reinterpret_cast<const Unaligned16 *>(s0 + head);
        for (long long v = threadIdx.x; v < nvec; v += blockDim.x) dv[v] = sv[v].v;
        for (long long i = head + 8 * nvec + threadIdx.x; i < span; i += blockDim.x) d0[i] = i < n ? s0[i] : int16_t(0);
    }
}

template <typename OutT>
int process_ragged_dev(mfcc_hip_handle *h, bool fixed, const int16_t *d_pcm, const size_t *offsets, size_t n_utt,
                       OutT *d_out, size_t cap, size_t *frame_offsets) {
    if (!h || !offsets || !frame_offsets) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (fixed && !h->fixed_ok) return MFCC_HIP_ERROR_UNSUPPORTED;
    const size_t hop = size_t(h->r.hop), nfft = size_t(h->r.nfft), ncep = size_t(h->r.n_cep);
    DeviceGuard guard(h->device);
    // A corpus of equal-length utterances lying back to back (BASELINE config 5: 10 000 x 10 s) IS a multi-channel
    // stream, channel stride = utterance length: no packing copy, no row gather, no descriptors, the same bits (every
    // channel of the plain call starts from reset, its frames are the utterance's frames and the rows come out dense).
    // Checked first: this is the per-step host work of a sharded corpus (bench.py `enqueue_us_per_step`).
    if (n_utt >= 1 && offsets[1] >= offsets[0]) {
        const size_t n0 = offsets[1] - offsets[0];
        const size_t nf0 = count_frames(h->r, n0);
        bool uniform = nf0 > 0;
        for (size_t u = 1; u < n_utt && uniform; ++u) uniform = offsets[u + 1] - offsets[u] == n0;
        if (uniform) {
            if (!d_pcm) return MFCC_HIP_ERROR_INVALID_PARAM;
            for (size_t u = 0; u <= n_utt; ++u) frame_offsets[u] = u * nf0;
            if (!d_out || cap < n_utt * nf0 * ncep) return MFCC_HIP_ERROR_BUFFER_SMALL;
            return launch(h, fixed, d_pcm + offsets[0], n0, n0, n_utt, 0, d_out, nullptr);
        }
    }
    // [0, 4n): pack descriptors, [4n, 7n): row-gather descriptors, [7n, 11n): the fused kernels' per-utterance records
    // (their own region: a corpus the records cannot describe falls through to pack-and-gather with [0, 7n) intact);
    // in pinned memory: the H2D copies below are asynchronous
    mfcc_hip_handle::PinnedDesc *pd = nullptr;
    int rc = desc_acquire(h, 11 * n_utt, &pd);
    if (rc) return rc;
    long long *desc = pd->p;
    long long *rec_area = desc + 7 * n_utt;
    for (size_t i = 0; i < 7 * n_utt; ++i) desc[i] = 0;
    size_t last_with_frames = n_utt;
    size_t pos = 0, total = 0;
    bool rec_fits = n_utt < (size_t(1) << 31);       // the records' int fields (else: pack and gather, below)
    frame_offsets[0] = 0;
    for (size_t u = 0; u < n_utt; ++u) {
        if (offsets[u + 1] < offsets[u]) return MFCC_HIP_ERROR_INVALID_PARAM;
        const size_t n = offsets[u + 1] - offsets[u];
        if (n && !d_pcm) return MFCC_HIP_ERROR_INVALID_PARAM;
        if (n >= (size_t(1) << 31)) rec_fits = false;
        const size_t nf = count_frames(h->r, n);
        desc[4 * u] = (long long)offsets[u];
        desc[4 * u + 1] = (long long)pos;
        desc[4 * u + 2] = nf ? (long long)n : 0;
        desc[4 * n_utt + 3 * u] = (long long)(pos / hop);
        desc[4 * n_utt + 3 * u + 1] = (long long)total;
        desc[4 * n_utt + 3 * u + 2] = (long long)nf;
        total += nf;
        frame_offsets[u + 1] = total;
        if (nf) {
            const size_t extent = std::max(n, hop * (nf - 1) + nfft);
            pos = (pos + extent + 1 + hop - 1) / hop * hop;
            last_with_frames = u;
        }
        desc[4 * u + 3] = (long long)pos;          // this utterance's part of the stream ends where the next begins
    }
    if (total == 0) return MFCC_HIP_SUCCESS;
    if (!d_out || cap < total * ncep) return MFCC_HIP_ERROR_BUFFER_SMALL;
    // Float contract on the twelve-wave kernel: no packed copy at all -- per-utterance records, expanded on the device
    // into one record per tile; the kernel reads every utterance where it lies and writes its rows where they belong
    // (kernel_fused512_w12.hpp)
    if (!fixed && use_fused(h) && h->fused_w12 && std::is_same<OutT, float>::value && rec_fits) {
        static_assert(sizeof(mfcc_fused12::RaggedChan) == 4 * sizeof(long long), "record layout");
        mfcc_fused12::RaggedChan *rc_host = reinterpret_cast<mfcc_fused12::RaggedChan *>(rec_area);   // 4 long longs each
        long long n_tiles = 0;
        for (size_t u = 0; u < n_utt; ++u) {
            const size_t n = offsets[u + 1] - offsets[u];
            const size_t nf = frame_offsets[u + 1] - frame_offsets[u];
            const long long tiles = (long long)((nf + mfcc_fused::kTile - 1) / mfcc_fused::kTile);
            mfcc_fused12::RaggedChan c;
            c.pcm_off = (long long)offsets[u];
            c.out_row = (long long)frame_offsets[u];
            c.n_samples = (int)n;
            c.frames = (int)nf;
            c.t_hi = mfcc_fused12::ragged_t_hi((long long)n, tiles);
            c.tile0 = (int)n_tiles;
            rc_host[u] = c;
            n_tiles += tiles;
        }
        if (n_tiles < (1ll << 30)) {
            const size_t chan_bytes = n_utt * sizeof(mfcc_fused12::RaggedChan);
            const size_t map_off = (chan_bytes + 255) & ~size_t(255);
            rc = ensure(h, &h->d_in, &h->d_in_bytes, map_off + size_t(n_tiles) * sizeof(mfcc_fused12::RaggedTile) + 64);
            if (rc) return rc;
            if ((rc = scratch_acquire(h))) return rc;
            HIP_TRY(h, hipMemcpyAsync(h->d_in, rc_host, chan_bytes, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipEventRecord(pd->copied, h->stream));
            pd->in_flight = true;
            auto *d_chans = static_cast<const mfcc_fused12::RaggedChan *>(h->d_in);
            auto *d_map = reinterpret_cast<mfcc_fused12::RaggedTile *>(static_cast<char *>(h->d_in) + map_off);
            if (mfcc_fused12::launch_ragged(d_pcm, d_chans, (int)n_utt, d_map, (int)n_tiles, h->fu, h->fused_dense,
                                            reinterpret_cast<float *>(d_out), h->n_cu, h->stream)) {
                HIP_TRY(h, hipGetLastError());
                return scratch_release(h);
            }
            return MFCC_HIP_ERROR_OTHER;               // not launched: cannot happen once fused_w12 is set
        }
        // more tiles than the tile map's int index: pack and gather below (the pack descriptors are untouched)
    }
    // Fixed contract on the fused kernel: the same -- one record per utterance with frames, the kernel's waves walk
    // consecutive frames and step from one utterance into the next (kernel_fixed512.hpp)
    if (fixed && h->fixed512_ok && std::is_same<OutT, int16_t>::value && rec_fits) {
        static_assert(sizeof(mfcc_fixed512::RaggedRec) == 4 * sizeof(long long), "record layout");
        mfcc_fixed512::RaggedRec *rr = reinterpret_cast<mfcc_fixed512::RaggedRec *>(rec_area);
        size_t n_recs = 0;
        for (size_t u = 0; u < n_utt; ++u) {
            const size_t n = offsets[u + 1] - offsets[u];
            const size_t nf = frame_offsets[u + 1] - frame_offsets[u];
            if (!nf) continue;
            mfcc_fixed512::RaggedRec c;
            c.pcm_off = (long long)offsets[u];
            c.out_row = (long long)frame_offsets[u];
            c.n_samples = (int)n;
            c.frames = (int)nf;
            c.pad0 = c.pad1 = 0;
            rr[n_recs++] = c;
        }
        {
            const size_t rec_bytes = n_recs * sizeof(mfcc_fixed512::RaggedRec);
            rc = ensure(h, &h->d_in, &h->d_in_bytes, rec_bytes + 64);
            if (rc) return rc;
            if ((rc = scratch_acquire(h))) return rc;
            HIP_TRY(h, hipMemcpyAsync(h->d_in, rr, rec_bytes, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipEventRecord(pd->copied, h->stream));
            pd->in_flight = true;
            mfcc_fixed512::launch_ragged(d_pcm, static_cast<const mfcc_fixed512::RaggedRec *>(h->d_in), (int)n_recs,
                                         (long long)total, h->r.hop, h->x5, reinterpret_cast<int16_t *>(d_out), h->n_cu,
                                         h->stream);
            HIP_TRY(h, hipGetLastError());
            return scratch_release(h);
        }
    }
    const size_t F = pos / hop, len = pos + nfft + hop;
    desc[4 * last_with_frames + 3] = (long long)len;     // the last one also zeroes the tail of the stream
    const size_t desc_bytes = 7 * n_utt * sizeof(long long);
    rc = ensure(h, &h->d_in, &h->d_in_bytes, len * sizeof(int16_t) + 64);
    if (rc) return rc;
    rc = ensure(h, &h->d_out, &h->d_out_bytes, F * ncep * sizeof(OutT) + desc_bytes + 64);
    if (rc) return rc;
    if ((rc = scratch_acquire(h))) return rc;
    OutT *d_all = static_cast<OutT *>(h->d_out);
    const size_t desc_off = (F * ncep * sizeof(OutT) + 7) & ~size_t(7);
    long long *d_desc = reinterpret_cast<long long *>(static_cast<char *>(h->d_out) + desc_off);
    HIP_TRY(h, hipMemcpyAsync(d_desc, desc, desc_bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipEventRecord(pd->copied, h->stream));
    pd->in_flight = true;
    const unsigned blocks = (unsigned)std::min<size_t>(n_utt, size_t(h->n_cu) * 8);
    hipLaunchKernelGGL(pack_utterances_kernel, dim3(blocks), dim3(256), 0, h->stream, d_pcm,
                       static_cast<int16_t *>(h->d_in), d_desc, (long long)n_utt);
    rc = launch(h, fixed, h->d_in, len, len, 1, 0, d_all, nullptr, F);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_rows_kernel<OutT>, dim3(blocks), dim3(256), 0, h->stream, d_all, d_out,
                       d_desc + 4 * n_utt, (long long)n_utt, (int)ncep);
    HIP_TRY(h, hipGetLastError());
    return scratch_release(h);
}

// Host buffers: the corpus goes to the device in ONE copy (the span offsets[0] .. offsets[n_utt] as it lies), runs through
// the device-resident path above -- no copy at all for equal lengths, per-tile / per-utterance records for the fused
// kernels, pack and gather for the rest -- and its rows come back in one copy.  (Round 1 copied utterance by
// utterance into a zeroed packed stream: n_utt memcpy calls, 50 ms of call overhead for 10 000 utterances.)
template <typename OutT>
int process_ragged(mfcc_hip_handle *h, bool fixed, const int16_t *pcm, const size_t *offsets, size_t n_utt,
                   OutT *out, size_t cap, size_t *frame_offsets) {
    if (!h || !offsets || !frame_offsets) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (fixed && !h->fixed_ok) return MFCC_HIP_ERROR_UNSUPPORTED;
    const size_t ncep = size_t(h->r.n_cep);
    size_t total = 0;
    std::vector<size_t> rel(n_utt + 1, 0);
    for (size_t u = 0; u < n_utt; ++u) {
        if (offsets[u + 1] < offsets[u]) return MFCC_HIP_ERROR_INVALID_PARAM;
        const size_t n = offsets[u + 1] - offsets[u];
        if (n && !pcm) return MFCC_HIP_ERROR_INVALID_PARAM;
        rel[u + 1] = offsets[u + 1] - offsets[0];
        total += count_frames(h->r, n);
    }
    if (total == 0) {
        for (size_t u = 0; u <= n_utt; ++u) frame_offsets[u] = 0;
        return MFCC_HIP_SUCCESS;
    }
    if (!out || cap < total * ncep) {
        frame_offsets[0] = 0;                            // the counts, as before, so that the caller can size `out`
        for (size_t u = 0; u < n_utt; ++u) frame_offsets[u + 1] = frame_offsets[u] + count_frames(h->r, rel[u + 1] - rel[u]);
        return MFCC_HIP_ERROR_BUFFER_SMALL;
    }
    DeviceGuard guard(h->device);
    // the same copy pipeline as process_host: chunks of whole utterances, ~64 MB each; every chunk is one call of the
    // device-resident ragged path on its own offsets, its rows land behind those of the chunk before it
    const size_t kChunkBytes = host_chunk_bytes();
    struct Range { size_t u0, u1, rows0; };
    std::vector<Range> ranges;
    std::vector<PipeChunk> pc;
    {
        size_t u0 = 0, rows = 0, rows0 = 0;
        for (size_t u = 0; u < n_utt; ++u) {
            rows += count_frames(h->r, rel[u + 1] - rel[u]);
            const bool last = u + 1 == n_utt;
            if (last || (rel[u + 1] - rel[u0]) * sizeof(int16_t) >= kChunkBytes) {
                ranges.push_back({u0, u + 1, rows0});
                pc.push_back({pcm + offsets[0] + rel[u0], (rel[u + 1] - rel[u0]) * sizeof(int16_t), out + rows0 * ncep,
                              (rows - rows0) * ncep * sizeof(OutT)});
                u0 = u + 1;
                rows0 = rows;
            }
        }
    }
    frame_offsets[0] = 0;
    std::vector<size_t> loc, fo;
    return run_host_pipeline(h, pc, pcm + offsets[0], rel[n_utt] * sizeof(int16_t), out, total * ncep * sizeof(OutT),
                             [&](size_t i, void *d_in, void *d_out) {
        const Range &r = ranges[i];
        const size_t k = r.u1 - r.u0;
        loc.assign(k + 1, 0);
        fo.assign(k + 1, 0);
        for (size_t j = 0; j <= k; ++j) loc[j] = rel[r.u0 + j] - rel[r.u0];
        const int rc = process_ragged_dev<OutT>(h, fixed, static_cast<const int16_t *>(d_in), loc.data(), k,
                                                static_cast<OutT *>(d_out), pc[i].out_bytes / sizeof(OutT), fo.data());
        for (size_t j = 1; j <= k; ++j) frame_offsets[r.u0 + j] = r.rows0 + fo[j];
        return rc;
    });
}

// ---- minimal RIFF/WAVE reader (the reference uses the un-vendored libwav, main.c:58-98)
int read_wav_i16(const char *path, int want_rate, std::vector<int16_t> &pcm) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return MFCC_HIP_ERROR_IO;
    std::vector<unsigned char> d;
    unsigned char buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + got);
    std::fclose(f);
    auto u32 = [&](size_t o) { return uint32_t(d[o]) | uint32_t(d[o + 1]) << 8 | uint32_t(d[o + 2]) << 16 | uint32_t(d[o + 3]) << 24; };
    auto u16 = [&](size_t o) { return uint16_t(d[o] | d[o + 1] << 8); };
    if (d.size() < 12 || std::memcmp(d.data(), "RIFF", 4) || std::memcmp(d.data() + 8, "WAVE", 4))
        return MFCC_HIP_ERROR_INVALID_PARAM;
    size_t o = 12;
    int fmt = 0, ch = 0, bits = 0;
    uint32_t rate = 0;
    bool have_fmt = false;
    while (o + 8 <= d.size()) {
        uint32_t sz = u32(o + 4);
        size_t body = o + 8;
        if (!std::memcmp(d.data() + o, "fmt ", 4) && body + 16 <= d.size()) {
            fmt = u16(body); ch = u16(body + 2); rate = u32(body + 4); bits = u16(body + 14);
            have_fmt = true;
        } else if (!std::memcmp(d.data() + o, "data", 4)) {
            // same checks as mfcc_wav_open (software/main.c:72-92): PCM, 16 bit, expected rate
            if (!have_fmt || fmt != 1 || bits != 16 || ch != 1 || int(rate) != want_rate)
                return MFCC_HIP_ERROR_UNSUPPORTED;
            size_t avail = d.size() - body;
            size_t n = (sz < avail ? sz : avail) / 2;
            pcm.resize(n);
            for (size_t i = 0; i < n; ++i) pcm[i] = int16_t(u16(body + 2 * i));
            return MFCC_HIP_SUCCESS;
        }
        o = body + sz + (sz & 1);
    }
    return MFCC_HIP_ERROR_INVALID_PARAM;
}

}  // namespace

// ================================================================================ C ABI

extern "C" {

int mfcc_hip_abi_version(void) { return MFCC_HIP_ABI_VERSION; }

int mfcc_hip_default_params(mfcc_hip_params *p) {
    if (!p) return MFCC_HIP_ERROR_INVALID_PARAM;
    std::memset(p, 0, sizeof *p);
    p->struct_size = sizeof *p;
    p->nfft = 512;             // NFFT        software/main.c:11
    p->hop = 170;              // STEPSIZE    software/main.c:12
    p->n_mel = 32;             // nfilters    mfcc/targets/wav2mfcc.py:19
    p->n_cep = 13;             // BASELINE.json metric (reference tops keep 16/32)
    p->sample_rate = 16000;    // SAMPLERATE  software/main.c:14
    p->pad_mode = MFCC_HIP_PAD_NOTEBOOK;
    p->power_scale = 512.0f;   // MFCC.ipynb cell 22
    p->lifter = 0.0f;
    p->device = -1;
    p->float_impl = MFCC_HIP_IMPL_AUTO;
    return MFCC_HIP_SUCCESS;
}

int mfcc_hip_num_frames(const mfcc_hip_params *p, size_t n_samples, size_t *n_frames) {
    Resolved r;
    int rc = resolve(p, r);
    if (rc) return rc;
    if (!n_frames) return MFCC_HIP_ERROR_INVALID_PARAM;
    *n_frames = count_frames(r, n_samples);
    return MFCC_HIP_SUCCESS;
}

const char *mfcc_hip_strerror(int err) {
    switch (err) {
        case MFCC_HIP_SUCCESS: return "success";
        case MFCC_HIP_ERROR_INVALID_PARAM: return "invalid parameter";
        case MFCC_HIP_ERROR_NOT_FOUND: return "no usable HIP device (this library has no CPU path)";
        case MFCC_HIP_ERROR_NO_MEM: return "out of memory";
        case MFCC_HIP_ERROR_BUSY: return "busy";
        case MFCC_HIP_ERROR_UNSUPPORTED: return "parameter combination not supported by any kernel";
        case MFCC_HIP_ERROR_BUFFER_SMALL: return "output buffer too small";
        case MFCC_HIP_ERROR_IO: return "file i/o error";
        case MFCC_HIP_ERROR_OTHER: return "HIP runtime error (see mfcc_hip_last_hip_error)";
        default: return "unknown error";
    }
}

static thread_local int g_create_hip_error = 0;
int mfcc_hip_last_hip_error(const mfcc_hip_handle *h) { return h ? h->last_hip : g_create_hip_error; }

int mfcc_hip_get_table(const mfcc_hip_params *p, int which, void *buf, size_t cap, size_t *n_bytes) {
    Resolved r;
    int rc = resolve(p, r);
    if (rc) return rc;
    std::vector<char> blob;
    auto put = [&](const void *d, size_t n) { blob.assign((const char *)d, (const char *)d + n); };
    switch (which) {
        case MFCC_HIP_TABLE_WINDOW_F32: {
            std::vector<double> w = hamming_periodic(r.nfft);
            std::vector<float> f(w.begin(), w.end());
            put(f.data(), f.size() * 4);
            break;
        }
        case MFCC_HIP_TABLE_MEL_POINTS_I32: {
            std::vector<int> v = mel_points(r.nfft, r.n_mel, double(r.sample_rate));
            put(v.data(), v.size() * 4);
            break;
        }
        case MFCC_HIP_TABLE_MEL_DENSE_F32: {
            std::vector<double> w = mel_dense(r.nfft, r.n_mel, double(r.sample_rate));
            std::vector<float> f(w.begin(), w.end());
            put(f.data(), f.size() * 4);
            break;
        }
        case MFCC_HIP_TABLE_DCT_F32: {
            std::vector<double> w = dct_rows(r.n_cep, r.n_mel, r.lifter);
            std::vector<float> f(w.begin(), w.end());
            put(f.data(), f.size() * 4);
            break;
        }
        case MFCC_HIP_TABLE_FX_CURVE_I32: {
            std::vector<int> v = fx_window_curve(r.nfft);
            put(v.data(), v.size() * 4);
            break;
        }
        case MFCC_HIP_TABLE_FX_TWIDDLE_I32: {
            std::vector<int> re, im, v;
            fx_twiddles(r.nfft, re, im);
            for (size_t i = 0; i < re.size(); ++i) { v.push_back(re[i]); v.push_back(im[i]); }
            put(v.data(), v.size() * 4);
            break;
        }
        case MFCC_HIP_TABLE_FX_MEL_DENSE_U32: {
            if (!fixed_supported(r)) return MFCC_HIP_ERROR_UNSUPPORTED;
            FxMel m = fx_mel(r.nfft, r.n_mel, double(r.sample_rate));
            // first word: the shift; then the dense table
            std::vector<uint32_t> v;
            v.push_back(uint32_t(m.shift));
            v.insert(v.end(), m.dense.begin(), m.dense.end());
            put(v.data(), v.size() * 4);
            break;
        }
        default:
            return MFCC_HIP_ERROR_INVALID_PARAM;
    }
    if (n_bytes) *n_bytes = blob.size();
    if (buf) {
        if (cap < blob.size()) return MFCC_HIP_ERROR_BUFFER_SMALL;
        std::memcpy(buf, blob.data(), blob.size());
    }
    return MFCC_HIP_SUCCESS;
}

int mfcc_hip_create(const mfcc_hip_params *p, mfcc_hip_handle **out) {
    if (!out) return MFCC_HIP_ERROR_INVALID_PARAM;
    *out = nullptr;
    Resolved r;
    int rc = resolve(p, r);
    if (rc) return rc;
    int ndev = 0;
    g_create_hip_error = int(hipGetDeviceCount(&ndev));
    if (g_create_hip_error != int(hipSuccess) || ndev <= 0) return MFCC_HIP_ERROR_NOT_FOUND;
    int dev = r.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return MFCC_HIP_ERROR_NOT_FOUND;
    }
    if (dev >= ndev) return MFCC_HIP_ERROR_NOT_FOUND;
    mfcc_hip_handle *h = new (std::nothrow) mfcc_hip_handle();
    if (!h) return MFCC_HIP_ERROR_NO_MEM;
    h->r = r;
    h->device = dev;
    auto fail = [&](int code) {
        mfcc_hip_destroy(h);
        return code;
    };
    DeviceGuard guard(dev);
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != dev) return fail(MFCC_HIP_ERROR_NOT_FOUND);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(MFCC_HIP_ERROR_OTHER);
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
        return fail(MFCC_HIP_ERROR_OTHER);
    h->stream = h->own_stream;
    if (hipEventCreateWithFlags(&h->scratch_done, hipEventDisableTiming) != hipSuccess)
        return fail(MFCC_HIP_ERROR_OTHER);
    rc = build_tables(h);
    if (rc) return fail(rc);
    if (r.float_impl == MFCC_HIP_IMPL_FUSED512 && !h->fused_ok) return fail(MFCC_HIP_ERROR_UNSUPPORTED);
    *out = h;
    return MFCC_HIP_SUCCESS;
}

void mfcc_hip_destroy(mfcc_hip_handle *h) {
    if (!h) return;
    if (h->n_sessions > 0) {           // sessions still use the handle's stream and tables: the last one frees it
        h->destroy_pending = true;
        return;
    }
    DeviceGuard guard(h->device);
    if (h->scratch_used) (void)hipEventSynchronize(h->scratch_done);     // scratch may be in use on a caller's stream
    if (h->own_stream) {
        (void)hipStreamSynchronize(h->own_stream);
        (void)hipStreamDestroy(h->own_stream);
    }
    if (h->scratch_done) (void)hipEventDestroy(h->scratch_done);
    for (auto &d : h->desc) {
        if (d.copied) {
            if (d.in_flight) (void)hipEventSynchronize(d.copied);
            (void)hipEventDestroy(d.copied);
        }
        if (d.p) (void)hipHostFree(d.p);
    }
    for (int i = 0; i < mfcc_hip_handle::kPipe; ++i) {
        if (h->ev_in[i]) (void)hipEventDestroy(h->ev_in[i]);
        if (h->ev_k[i]) (void)hipEventDestroy(h->ev_k[i]);
        if (h->ev_out[i]) (void)hipEventDestroy(h->ev_out[i]);
        if (h->p_in[i]) (void)hipFree(h->p_in[i]);
        if (h->p_out[i]) (void)hipFree(h->p_out[i]);
    }
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    if (h->arena) (void)hipFree(h->arena);
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->d_out) (void)hipFree(h->d_out);
    delete h;
}

int mfcc_hip_set_stream(mfcc_hip_handle *h, void *hip_stream) {
    if (!h) return MFCC_HIP_ERROR_INVALID_PARAM;
    h->stream = static_cast<hipStream_t>(hip_stream);      // NULL = the HIP null stream
    return MFCC_HIP_SUCCESS;
}

int mfcc_hip_use_own_stream(mfcc_hip_handle *h) {
    if (!h) return MFCC_HIP_ERROR_INVALID_PARAM;
    h->stream = h->own_stream;
    return MFCC_HIP_SUCCESS;
}

int mfcc_hip_synchronize(mfcc_hip_handle *h) {
    if (!h) return MFCC_HIP_ERROR_INVALID_PARAM;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MFCC_HIP_SUCCESS;
}

int mfcc_hip_process_i16(mfcc_hip_handle *h, const int16_t *pcm, size_t n, size_t nch, float *out,
                         size_t cap, size_t *n_frames) {
    return process_host<float>(h, false, pcm, n, nch, out, cap, n_frames);
}

int mfcc_hip_process_fixed_i16(mfcc_hip_handle *h, const int16_t *pcm, size_t n, size_t nch,
                               int16_t *out, size_t cap, size_t *n_frames) {
    return process_host<int16_t>(h, true, pcm, n, nch, out, cap, n_frames);
}

int mfcc_hip_process_ragged_i16(mfcc_hip_handle *h, const int16_t *pcm, const size_t *offsets, size_t n_utt,
                                float *out, size_t cap, size_t *frame_offsets) {
    return process_ragged<float>(h, false, pcm, offsets, n_utt, out, cap, frame_offsets);
}

int mfcc_hip_process_ragged_fixed_i16(mfcc_hip_handle *h, const int16_t *pcm, const size_t *offsets, size_t n_utt,
                                      int16_t *out, size_t cap, size_t *frame_offsets) {
    return process_ragged<int16_t>(h, true, pcm, offsets, n_utt, out, cap, frame_offsets);
}

int mfcc_hip_process_ragged_i16_dev(mfcc_hip_handle *h, const void *d_pcm, const size_t *offsets, size_t n_utt,
                                    void *d_out, size_t cap, size_t *frame_offsets) {
    return process_ragged_dev<float>(h, false, static_cast<const int16_t *>(d_pcm), offsets, n_utt,
                                     static_cast<float *>(d_out), cap, frame_offsets);
}

int mfcc_hip_process_ragged_fixed_i16_dev(mfcc_hip_handle *h, const void *d_pcm, const size_t *offsets, size_t n_utt,
                                          void *d_out, size_t cap, size_t *frame_offsets) {
    return process_ragged_dev<int16_t>(h, true, static_cast<const int16_t *>(d_pcm), offsets, n_utt,
                                       static_cast<int16_t *>(d_out), cap, frame_offsets);
}

int mfcc_hip_process_i16_dev(mfcc_hip_handle *h, const void *d_pcm, size_t n, size_t stride, size_t nch,
                             int halo, void *d_out, size_t *n_frames) {
    return launch(h, false, d_pcm, n, stride, nch, halo, d_out, n_frames);
}

int mfcc_hip_process_fixed_i16_dev(mfcc_hip_handle *h, const void *d_pcm, size_t n, size_t stride,
                                   size_t nch, int halo, void *d_out, size_t *n_frames) {
    return launch(h, true, d_pcm, n, stride, nch, halo, d_out, n_frames);
}

int mfcc_hip_time_dev(mfcc_hip_handle *h, int fixed, const void *d_pcm, size_t n, size_t stride,
                      size_t nch, void *d_out, int warmup, int iters, float *avg_ms) {
    if (!h || iters < 1 || warmup < 0 || !avg_ms) return MFCC_HIP_ERROR_INVALID_PARAM;
    DeviceGuard guard(h->device);
    for (int i = 0; i < warmup; ++i) {
        int rc = launch(h, fixed != 0, d_pcm, n, stride, nch, 0, d_out, nullptr);
        if (rc) return rc;
    }
    struct Events {                                // destroyed on every return path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
    HIP_TRY(h, hipEventCreate(&ev.e0));
    HIP_TRY(h, hipEventCreate(&ev.e1));
    HIP_TRY(h, hipEventRecord(ev.e0, h->stream));
    for (int i = 0; i < iters; ++i) {
        int rc = launch(h, fixed != 0, d_pcm, n, stride, nch, 0, d_out, nullptr);
        if (rc) return rc;
    }
    HIP_TRY(h, hipEventRecord(ev.e1, h->stream));
    HIP_TRY(h, hipEventSynchronize(ev.e1));
    float ms = 0.0f;
    HIP_TRY(h, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *avg_ms = ms / float(iters);
    return MFCC_HIP_SUCCESS;
}

const char *mfcc_hip_kernel_name(const mfcc_hip_handle *h, int fixed) {
    if (!h) return "";
    if (fixed) return h->fixed512_ok ? mfcc_fixed512::kernel_name() : "mfcc_fixed_kernel";
    if (use_fused(h)) return h->fused_w12 ? mfcc_fused12::kernel_name() : mfcc_fused::kernel_name();
    if (h->fused1k_ok && h->r.float_impl == MFCC_HIP_IMPL_AUTO)
        return !h->f1k_w12 ? mfcc_fused1024::kernel_name()
                           : h->f1k_is_f32 ? mfcc_fused1024_w12::kernel_name() : mfcc_fused1024_w12bf::kernel_name();
    return "mfcc_float_generic_kernel";
}

int mfcc_hip_convert_wav(mfcc_hip_handle *h, const char *wav_in, const char *mfcc_out, int fixed,
                         size_t *n_frames_out) {
    if (!h || !wav_in || !mfcc_out) return MFCC_HIP_ERROR_INVALID_PARAM;
    std::vector<int16_t> pcm;
    int rc = read_wav_i16(wav_in, h->r.sample_rate, pcm);
    if (rc) return rc;
    const size_t nf = count_frames(h->r, pcm.size());
    std::vector<int16_t> cep(nf * size_t(h->r.n_cep));
    if (fixed) {
        rc = mfcc_hip_process_fixed_i16(h, pcm.data(), pcm.size(), 1, cep.data(), cep.size(), nullptr);
        if (rc) return rc;
    } else {
        std::vector<float> f(cep.size());
        rc = mfcc_hip_process_i16(h, pcm.data(), pcm.size(), 1, f.data(), f.size(), nullptr);
        if (rc) return rc;
        for (size_t i = 0; i < f.size(); ++i) {
            float v = f[i];                                   // astype(np.int16): truncate
            if (!(v == v)) v = 0.0f;
            if (v > 32767.0f) v = 32767.0f;
            if (v < -32768.0f) v = -32768.0f;
            cep[i] = int16_t(v);
        }
    }
    FILE *o = std::fopen(mfcc_out, "wb");
    if (!o) return MFCC_HIP_ERROR_IO;
    size_t wr = cep.empty() ? 0 : std::fwrite(cep.data(), sizeof(int16_t), cep.size(), o);
    std::fclose(o);
    if (wr != cep.size()) return MFCC_HIP_ERROR_IO;
    if (n_frames_out) *n_frames_out = nf;
    return MFCC_HIP_SUCCESS;
}

// float coefficients -> the int16 a `.mfcc` file holds: astype(np.int16) of software/lift.py:39 (truncate)
static inline int16_t to_mfcc_i16(float v) {
    if (!(v == v)) v = 0.0f;
    if (v > 32767.0f) v = 32767.0f;
    if (v < -32768.0f) v = -32768.0f;
    return int16_t(v);
}

int mfcc_hip_convert_wavs(mfcc_hip_handle *h, const char *const *wav_in, const char *const *mfcc_out, size_t n_files,
                          int fixed, size_t *n_frames_each) {
    if (!h || (n_files && (!wav_in || !mfcc_out))) return MFCC_HIP_ERROR_INVALID_PARAM;
    std::vector<int16_t> pcm;
    std::vector<size_t> off(n_files + 1, 0), fo(n_files + 1, 0);
    for (size_t i = 0; i < n_files; ++i) {
        if (!wav_in[i] || !mfcc_out[i]) return MFCC_HIP_ERROR_INVALID_PARAM;
        std::vector<int16_t> one;
        int rc = read_wav_i16(wav_in[i], h->r.sample_rate, one);
        if (rc) return rc;
        pcm.insert(pcm.end(), one.begin(), one.end());
        off[i + 1] = pcm.size();
    }
    size_t total = 0;
    for (size_t i = 0; i < n_files; ++i) total += count_frames(h->r, off[i + 1] - off[i]);
    const size_t ncep = size_t(h->r.n_cep);
    std::vector<int16_t> cep(total * ncep);
    int rc;
    if (fixed) {
        rc = mfcc_hip_process_ragged_fixed_i16(h, pcm.data(), off.data(), n_files, cep.data(), cep.size(), fo.data());
        if (rc) return rc;
    } else {
        std::vector<float> f(cep.size());
        rc = mfcc_hip_process_ragged_i16(h, pcm.data(), off.data(), n_files, f.data(), f.size(), fo.data());
        if (rc) return rc;
        for (size_t i = 0; i < f.size(); ++i) cep[i] = to_mfcc_i16(f[i]);
    }
    for (size_t i = 0; i < n_files; ++i) {
        const size_t nf = fo[i + 1] - fo[i];
        FILE *o = std::fopen(mfcc_out[i], "wb");
        if (!o) return MFCC_HIP_ERROR_IO;
        const size_t want = nf * ncep;
        const size_t wr = want ? std::fwrite(cep.data() + fo[i] * ncep, sizeof(int16_t), want, o) : 0;
        std::fclose(o);
        if (wr != want) return MFCC_HIP_ERROR_IO;
        if (n_frames_each) n_frames_each[i] = nf;
    }
    return MFCC_HIP_SUCCESS;
}

// ---- online / streaming session (include/mfcc_hip.h: mfcc_hip_stream_*) -----------------------------------
// Device buffer layout, both ping-pong buffers: [0] = the history sample x[first - 1] (0 after reset),
// [1 .. 1 + pending) = the samples of the frame in progress, then the new samples of this push.

}  // extern "C"

struct mfcc_hip_stream {
    mfcc_hip_handle *h = nullptr;
    bool fixed = false;
    int16_t *buf[2] = {nullptr, nullptr};
    size_t cap = 0;                 // samples each buffer holds behind the history slot
    int cur = 0;                    // buffer that holds history + pending
    size_t pending = 0;
    void *d_out = nullptr;
    size_t d_out_bytes = 0;
};

namespace {

int stream_reserve(mfcc_hip_stream *s, size_t samples, size_t out_bytes) {
    mfcc_hip_handle *h = s->h;
    if (samples > s->cap) {
        const size_t want = samples + samples / 2 + 4096;
        int16_t *nb[2] = {nullptr, nullptr};
        for (int i = 0; i < 2; ++i) HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&nb[i]), (want + 1 + 64) * sizeof(int16_t)));
        // carry history + pending over (the other buffer holds nothing that is still needed)
        HIP_TRY(h, hipMemsetAsync(nb[0], 0, sizeof(int16_t), h->stream));
        if (s->buf[s->cur])
            HIP_TRY(h, hipMemcpyAsync(nb[0], s->buf[s->cur], (1 + s->pending) * sizeof(int16_t), hipMemcpyDeviceToDevice,
                                      h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < 2; ++i) {
            if (s->buf[i]) (void)hipFree(s->buf[i]);
            s->buf[i] = nb[i];
        }
        s->cur = 0;
        s->cap = want;
    }
    if (out_bytes > s->d_out_bytes) {
        if (s->d_out) HIP_TRY(h, hipFree(s->d_out));
        s->d_out = nullptr;
        s->d_out_bytes = 0;
        const size_t want = out_bytes + out_bytes / 2 + 4096;
        HIP_TRY(h, hipMalloc(&s->d_out, want));
        s->d_out_bytes = want;
    }
    return MFCC_HIP_SUCCESS;
}

// run `nf` frames over history + the first `total` samples of the current buffer, copy them to `out`
int stream_emit(mfcc_hip_stream *s, size_t total, size_t nf, void *out) {
    mfcc_hip_handle *h = s->h;
    const size_t esz = s->fixed ? sizeof(int16_t) : sizeof(float);
    const size_t bytes = nf * size_t(h->r.n_cep) * esz;
    int rc = launch(h, s->fixed, s->buf[s->cur], total, total + 1, 1, /*halo=*/1, s->d_out, nullptr, nf);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(out, s->d_out, bytes, hipMemcpyDeviceToHost, h->stream));
    return MFCC_HIP_SUCCESS;
}

}  // namespace

extern "C" {

int mfcc_hip_stream_create(mfcc_hip_handle *h, int fixed, mfcc_hip_stream **out) {
    if (!h || !out) return MFCC_HIP_ERROR_INVALID_PARAM;
    *out = nullptr;
    if (h->destroy_pending) return MFCC_HIP_ERROR_INVALID_PARAM;        // the handle was already given back
    if (fixed && !h->fixed_ok) return MFCC_HIP_ERROR_UNSUPPORTED;
    mfcc_hip_stream *s = new (std::nothrow) mfcc_hip_stream();
    if (!s) return MFCC_HIP_ERROR_NO_MEM;
    s->h = h;
    s->fixed = fixed != 0;
    ++h->n_sessions;
    DeviceGuard guard(h->device);
    int rc = stream_reserve(s, size_t(h->r.nfft) * 8, size_t(64) * size_t(h->r.n_cep) * sizeof(float));
    if (rc) {
        mfcc_hip_stream_destroy(s);
        return rc;
    }
    *out = s;
    return MFCC_HIP_SUCCESS;
}

void mfcc_hip_stream_destroy(mfcc_hip_stream *s) {
    if (!s) return;
    mfcc_hip_handle *h = s->h;
    {
        DeviceGuard guard(h->device);
        (void)hipStreamSynchronize(h->stream);
        for (int i = 0; i < 2; ++i)
            if (s->buf[i]) (void)hipFree(s->buf[i]);
        if (s->d_out) (void)hipFree(s->d_out);
        delete s;
    }
    if (--h->n_sessions == 0 && h->destroy_pending) mfcc_hip_destroy(h);
}

int mfcc_hip_stream_reset(mfcc_hip_stream *s) {
    if (!s) return MFCC_HIP_ERROR_INVALID_PARAM;
    DeviceGuard guard(s->h->device);
    s->pending = 0;
    HIP_TRY(s->h, hipMemsetAsync(s->buf[s->cur], 0, sizeof(int16_t), s->h->stream));     // history := 0
    HIP_TRY(s->h, hipStreamSynchronize(s->h->stream));
    return MFCC_HIP_SUCCESS;
}

size_t mfcc_hip_stream_pending(const mfcc_hip_stream *s) { return s ? s->pending : 0; }

size_t mfcc_hip_stream_max_frames(const mfcc_hip_stream *s, size_t n) {
    return s ? (s->pending + n) / size_t(s->h->r.hop) + 1 : 0;
}

int mfcc_hip_stream_push(mfcc_hip_stream *s, const int16_t *samples, size_t n, void *out, size_t cap,
                         size_t *n_frames_out) {
    if (!s || (n && !samples)) return MFCC_HIP_ERROR_INVALID_PARAM;
    mfcc_hip_handle *h = s->h;
    const size_t nfft = size_t(h->r.nfft), hop = size_t(h->r.hop), ncep = size_t(h->r.n_cep);
    const size_t total = s->pending + n;
    const size_t nf = total >= nfft ? (total - nfft) / hop + 1 : 0;          // frames this push completes
    if (n_frames_out) *n_frames_out = nf;
    if (nf && (!out || cap < nf * ncep)) return MFCC_HIP_ERROR_BUFFER_SMALL;  // nothing consumed yet
    DeviceGuard guard(h->device);
    int rc = stream_reserve(s, total, nf * ncep * sizeof(float));
    if (rc) return rc;
    if (n)
        HIP_TRY(h, hipMemcpyAsync(s->buf[s->cur] + 1 + s->pending, samples, n * sizeof(int16_t), hipMemcpyHostToDevice,
                                  h->stream));
    if (nf) {
        rc = stream_emit(s, total, nf, out);
        if (rc) return rc;
        // the next frame starts nf hops further on: its history sample and what is already there move to the
        // front of the other buffer
        const size_t used = nf * hop;
        HIP_TRY(h, hipMemcpyAsync(s->buf[s->cur ^ 1], s->buf[s->cur] + used, (1 + total - used) * sizeof(int16_t),
                                  hipMemcpyDeviceToDevice, h->stream));
        s->cur ^= 1;
        s->pending = total - used;
    } else {
        s->pending = total;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));      // `samples` and `out` belong to the caller again
    return MFCC_HIP_SUCCESS;
}

int mfcc_hip_stream_flush(mfcc_hip_stream *s, void *out, size_t cap, size_t *n_frames_out) {
    if (!s) return MFCC_HIP_ERROR_INVALID_PARAM;
    mfcc_hip_handle *h = s->h;
    const size_t nf = h->r.pad_mode == MFCC_HIP_PAD_STREAM ? 1 : 0;
    if (n_frames_out) *n_frames_out = nf;
    if (nf && (!out || cap < size_t(h->r.n_cep))) return MFCC_HIP_ERROR_BUFFER_SMALL;
    DeviceGuard guard(h->device);
    if (nf) {
        // the zero-padded tail frame of main.c:134-144: the pending samples, then zeros (the kernels read
        // x[i] = 0 beyond the `total` samples they are given)
        int rc = stream_emit(s, s->pending, 1, out);
        if (rc) return rc;
    }
    return mfcc_hip_stream_reset(s);
}

int mfcc_hip_lift_file(const char *mfcc_in, const char *lift_out, int n_cep, double L, size_t *n_frames_out) {
    if (!mfcc_in || !lift_out || n_cep <= 0) return MFCC_HIP_ERROR_INVALID_PARAM;
    FILE *f = std::fopen(mfcc_in, "rb");
    if (!f) return MFCC_HIP_ERROR_IO;
    std::vector<int16_t> v;
    int16_t buf[4096];
    size_t got;
    while ((got = std::fread(buf, sizeof(int16_t), 4096, f)) > 0) v.insert(v.end(), buf, buf + got);
    std::fclose(f);
    if (v.size() % size_t(n_cep)) return MFCC_HIP_ERROR_INVALID_PARAM;       // np.reshape(raw, (-1, NCEPSTRUMS)) would raise
    std::vector<double> lift(n_cep, 1.0);
    if (L > 0.0)
        for (int n = 0; n < n_cep; ++n) lift[n] = 1.0 + (L / 2.0) * std::sin(mfcc_tables::kPi * double(n) / L);
    for (size_t i = 0; i < v.size(); ++i) {
        // lift.py:39: (lift * cepstra).astype(np.int16) -- truncation toward zero; a value beyond int16 keeps
        // its low 16 bits (C conversion through a wider integer, what NumPy does on x86-64)
        const double x = lift[i % size_t(n_cep)] * double(v[i]);
        v[i] = int16_t(uint16_t(int64_t(x)));
    }
    FILE *o = std::fopen(lift_out, "wb");
    if (!o) return MFCC_HIP_ERROR_IO;
    const size_t wr = v.empty() ? 0 : std::fwrite(v.data(), sizeof(int16_t), v.size(), o);
    std::fclose(o);
    if (wr != v.size()) return MFCC_HIP_ERROR_IO;
    if (n_frames_out) *n_frames_out = v.size() / size_t(n_cep);
    return MFCC_HIP_SUCCESS;
}

// ---- serial wire format + power gate (host only): magic.py:9-41, serial.c:89-122, cepstrum.c:15-71,161-183
size_t mfcc_hip_serial_packed_size(size_t n_frames, int n_cep) {
    return n_cep > 0 ? n_frames * 2 * (size_t(n_cep) + 1) : 0;
}

int mfcc_hip_serial_pack(const int16_t *cep, size_t n_frames, int n_cep, uint8_t *out, size_t out_capacity) {
    if (n_cep <= 0 || (n_frames && (!cep || !out))) return MFCC_HIP_ERROR_INVALID_PARAM;
    if (out_capacity < mfcc_hip_serial_packed_size(n_frames, n_cep)) return MFCC_HIP_ERROR_BUFFER_SMALL;
    uint8_t *p = out;
    for (size_t f = 0; f < n_frames; ++f) {
        *p++ = 0xa5;                                   // MagicInserter: 0xa55a first, high byte first on the wire
        *p++ = 0x5a;
        for (int c = 0; c < n_cep; ++c) {
            const uint16_t v = (uint16_t)cep[f * n_cep + c];
            *p++ = (uint8_t)(v >> 8);
            *p++ = (uint8_t)(v & 0xff);
        }
    }
    return MFCC_HIP_SUCCESS;
}

int mfcc_hip_serial_unpack(const uint8_t *bytes, size_t n_bytes, int n_cep, int16_t *cep, size_t max_frames,
                           size_t *n_frames_out, size_t *consumed_out) {
    if (n_cep <= 0 || (n_bytes && !bytes) || (max_frames && !cep)) return MFCC_HIP_ERROR_INVALID_PARAM;
    size_t pos = 0, frames = 0, consumed = 0;
    const size_t col = size_t(n_cep) * 2;
    while (frames < max_frames) {
        // expect_magic: skip to 0xa5; the byte after it must be 0x5a, otherwise both are dropped
        size_t q = pos;
        bool aligned = false;
        while (!aligned) {
            while (q < n_bytes && bytes[q] != 0xa5) ++q;
            if (q + 1 >= n_bytes) { q = n_bytes; break; }
            aligned = bytes[q + 1] == 0x5a;
            q += 2;
        }
        if (!aligned || q + col > n_bytes) break;
        for (int c = 0; c < n_cep; ++c)
            cep[frames * n_cep + c] = (int16_t)(uint16_t)((bytes[q + 2 * c] << 8) | bytes[q + 2 * c + 1]);
        pos = q + col;
        consumed = pos;
        ++frames;
    }
    if (n_frames_out) *n_frames_out = frames;
    if (consumed_out) *consumed_out = consumed;
    return MFCC_HIP_SUCCESS;
}

int mfcc_hip_eval_power(const int16_t *window, int n_cep, int n_frames, size_t head, long long *power_out) {
    if (!window || n_cep <= 0 || n_frames <= 0) return MFCC_HIP_ERROR_INVALID_PARAM;
    const size_t size = size_t(n_cep) * size_t(n_frames);
    if (head >= size) return MFCC_HIP_ERROR_INVALID_PARAM;
    const size_t first = 1 * size / 3, last = 2 * size / 3;
    long long power = 0;
    for (size_t i = first; i < last; i += size_t(n_cep)) {
        size_t k = head + i;
        if (k >= size) k -= size;
        power += (long long)window[k] * (long long)window[k];
    }
    if (power_out) *power_out = power;
    return power >= 100000000ll ? 1 : 0;               // POWER_THRESHOLD, cepstrum.c:13
}

#ifdef MFCC_W12_STAMPS
int mfcc_hip_debug_read_stamps12(unsigned long long *dst) {
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(mfcc_fused12::g_stamps12), sizeof(unsigned long long) * 48) != hipSuccess)
        return MFCC_HIP_ERROR_OTHER;
    unsigned long long z[48] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mfcc_fused12::g_stamps12), z, sizeof z) != hipSuccess) return MFCC_HIP_ERROR_OTHER;
    return MFCC_HIP_SUCCESS;
}
#endif

#ifdef MFCC_1K12_STAMPS
extern "C" int mfcc_hip_debug_read_stamps1k(unsigned long long *dst) {
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(mfcc_fused1024_w12::g_stamps1k), sizeof(unsigned long long) * 144) != hipSuccess)
        return MFCC_HIP_ERROR_OTHER;
    unsigned long long z[144] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mfcc_fused1024_w12::g_stamps1k), z, sizeof z) != hipSuccess) return MFCC_HIP_ERROR_OTHER;
    return MFCC_HIP_SUCCESS;
}
#endif
#ifdef MFCC_FUSED_STAMPS
// diagnostic build only: copy out and clear the per-phase cycle sums of the fused kernel
int mfcc_hip_debug_read_stamps(unsigned long long *dst) {
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(mfcc_fused::g_stamps), sizeof(unsigned long long) * 64) != hipSuccess)  // 4 x 12 phases + 16
        return MFCC_HIP_ERROR_OTHER;
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mfcc_fused::g_stamps), z, sizeof z) != hipSuccess) return MFCC_HIP_ERROR_OTHER;
    return MFCC_HIP_SUCCESS;
}
#endif

}  // extern "C"
