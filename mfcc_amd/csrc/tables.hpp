// Host-side builders for the constant tables of the MFCC kernels (no HIP in this file).
//
// Float contract  = notebook/MFCC.ipynb of the reference (cells cited per function).
// Fixed contract  = the RTL of mfcc/core/*.py + mfcc/misc/fft.py (lines cited per function).
// All tables are computed in double precision with the same operation order as the
// reference's NumPy expressions, then narrowed.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

namespace mfcc_tables {

static const double kPi = 3.141592653589793238462643383279502884;

inline int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}
inline bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// np.linspace(a, b, num)[i] for endpoint=True: a + i*step with step=(b-a)/(num-1), last = b
inline double linspace_at(double a, double b, int num, int i) {
    if (num == 1) return a;
    if (i == num - 1) return b;
    double step = (b - a) / double(num - 1);
    return double(i) * step + a;
}

// scipy.signal.get_window("hamm", n, fftbins=True)  (MFCC.ipynb cell 17 line 5,
// mfcc/core/window.py:24): general_cosine over linspace(-pi, pi, n + 1)[:-1]
inline std::vector<double> hamming_periodic(int n) {
    std::vector<double> w(n);
    for (int i = 0; i < n; ++i) {
        double fac = linspace_at(-kPi, kPi, n + 1, i);
        w[i] = 0.54 + 0.46 * std::cos(fac);       // a0*cos(0*fac) + a1*cos(1*fac)
    }
    return w;
}

// MFCC.ipynb cells 26-27 == mfcc/core/filterbank.py:9-20
inline std::vector<int> mel_points(int nfft, int n_mel, double sample_rate) {
    double fmin = 0.0, fmax = sample_rate / 2.0;
    double mel_lo = 2595.0 * std::log10(1.0 + fmin / 700.0);
    double mel_hi = 2595.0 * std::log10(1.0 + fmax / 700.0);
    std::vector<int> pts(n_mel + 2);
    for (int i = 0; i < n_mel + 2; ++i) {
        double m = linspace_at(mel_lo, mel_hi, n_mel + 2, i);
        double f = 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
        pts[i] = int(std::floor(double(nfft + 1) / sample_rate * f));
    }
    return pts;
}

// MFCC.ipynb cell 30 get_filters: rows x (nfft/2 + 1), linspace ramps, no normalisation
inline std::vector<double> mel_dense(int nfft, int n_mel, double sample_rate) {
    int nb = nfft / 2 + 1;
    std::vector<int> p = mel_points(nfft, n_mel, sample_rate);
    std::vector<double> w(size_t(n_mel) * nb, 0.0);
    for (int n = 0; n < n_mel; ++n) {
        int l1 = p[n + 1] - p[n];
        for (int j = 0; j < l1; ++j) {
            int k = p[n] + j;
            if (k >= 0 && k < nb) w[size_t(n) * nb + k] = linspace_at(0.0, 1.0, l1, j);
        }
        int l2 = p[n + 2] - p[n + 1];
        for (int j = 0; j < l2; ++j) {
            int k = p[n + 1] + j;
            if (k >= 0 && k < nb) w[size_t(n) * nb + k] = linspace_at(1.0, 0.0, l2, j);
        }
    }
    return w;
}

// MFCC.ipynb cell 38 dct(): orthonormal DCT-II basis, first n_cep rows; optional
// sinusoidal lifter of cell 43 / software/lift.py:12-26 folded in (L <= 0: off)
inline std::vector<double> dct_rows(int n_cep, int n_mel, double lifter_L) {
    std::vector<double> b(size_t(n_cep) * n_mel);
    for (int i = 0; i < n_cep; ++i) {
        double lift = 1.0;
        if (lifter_L > 0.0) lift = 1.0 + (lifter_L / 2.0) * std::sin(kPi * double(i) / lifter_L);
        for (int n = 0; n < n_mel; ++n) {
            double v;
            if (i == 0) {
                v = 1.0 / std::sqrt(double(n_mel));
            } else {
                double sample = double(2 * n + 1) * kPi / (2.0 * double(n_mel));
                v = std::cos(double(i) * sample) * std::sqrt(2.0 / double(n_mel));
            }
            b[size_t(i) * n_mel + n] = v * lift;
        }
    }
    return b;
}

// ------------------------------------------------------------------ fixed-point (RTL) tables

// mfcc/core/window.py:22-43 calc_coeffs + :53-123 curve reconstruction (precision = 8)
inline std::vector<int> fx_window_curve(int nfft, int precision = 8) {
    const int maxheight = (1 << (precision + 1)) - 1;
    std::vector<double> w = hamming_periodic(nfft);
    std::vector<int> winfull(nfft);
    for (int i = 0; i < nfft; ++i) winfull[i] = int(w[i] * double(maxheight));   // astype(int)
    int nmem = nfft / 8;
    std::vector<int> mem(nmem);
    for (int i = 0; i < nmem; ++i) mem[i] = winfull[2 * i + 1];
    int off_fst = mem[0];
    for (int i = 0; i < nmem; ++i) mem[i] -= off_fst;
    int off_lst = 2 * (winfull[nfft / 4] - off_fst);
    int nb = ilog2(nfft);
    int amask = (1 << (nb - 3)) - 1;
    int pmask = (1 << (precision + 1)) - 1;
    std::vector<int> curve(nfft);
    int point_r = 0;
    for (int c = 0; c < nfft; ++c) {
        int msb = (c >> (nb - 1)) & 1;
        int dir = (c >> (nb - 2)) & 1;
        int addr = (c >> 1) & amask;
        if (dir) addr = (~addr) & amask;
        int point = mem[addr];
        if (msb ^ dir) point = (off_lst - point) & pmask;
        if (c & 1) {
            curve[c] = (off_fst + point) & pmask;
            point_r = point;
        } else {
            curve[c] = (off_fst + ((point + point_r) >> 1)) & pmask;
        }
    }
    return curve;
}

// mfcc/misc/fft.py:28-36 (ROM) + :48-59 (second-quadrant decode), width 16: re,im for
// k in [0, size/2)
inline void fx_twiddles(int size, std::vector<int> &re, std::vector<int> &im) {
    int q = size / 4;
    re.assign(size / 2, 0);
    im.assign(size / 2, 0);
    for (int k = 0; k < q; ++k) {
        double p = linspace_at(0.0, kPi / 2.0, q + 1, k);      // endpoint=False: same step
        // np.round = rint (half to even)
        double r = std::nearbyint(16384.0 * std::cos(p));
        double i = std::nearbyint(16384.0 * -std::sin(p));
        re[k] = int(r);
        im[k] = int(i);
    }
    for (int k = 0; k < q; ++k) {
        re[q + k] = im[k];
        im[q + k] = -re[k];
    }
}

struct FxMel {
    int n_out;                       // values the streaming filterbank emits per frame (must be n_mel; fewer when the
                                     // filter points are too dense for its ramp logic)
    int shift;                       // right shift of the 64-bit accumulator
    std::vector<uint32_t> dense;     // [n_mel][nfft/2], weights x 2^-30
};

// Closed form of the streaming accumulators of mfcc/core/filterbank.py:88-142
// (width = width_mul = 30, gain 18, width_output 16; mfcc/core/mfcc.py:69-75):
// output r+1 = ((sum_k d_k * w[r][k]) >> shift) & 0xFFFF with
//   interior bin k of segment s (ramp b_k):  w[s-1][k] = 2^30 - b_k,  w[s][k] = b_k
//   event bin of segment s ("highest"):      w[s][k]   = 2^30
// (the `last` bin of the frame is never emitted).
inline FxMel fx_mel(int nfft, int n_mel, double sample_rate) {
    const int wsize = 30;
    std::vector<int> p = mel_points(nfft, n_mel, sample_rate);
    int nseg = n_mel + 1;
    std::vector<unsigned long long> steps(nseg);
    const unsigned long long max_acc = 1ull << (2 * wsize);
    for (int i = 0; i < nseg; ++i) {
        long long diff = (long long)p[i + 1] - p[i] - 1;
        if (diff != 0) {
            // Python floor division (diff may be negative for degenerate point sets)
            long long q = (long long)max_acc / diff;
            if (((long long)max_acc % diff != 0) && (diff < 0)) --q;
            steps[i] = (unsigned long long)(q - 1) & (max_acc - 1);
        } else {
            steps[i] = max_acc - 1;
        }
    }
    FxMel m;
    int span = p[n_mel + 1] - p[n_mel - 1];
    int lg = 0;
    while ((1 << (lg + 1)) <= span) ++lg;                // int(math.log2(span))
    int maxvalrange = lg + 30 + wsize;
    m.shift = maxvalrange - (18 + 16);
    int nb = nfft / 2;
    m.dense.assign(size_t(n_mel) * nb, 0u);
    unsigned long long acc = 0;
    int adr = 0;
    m.n_out = 0;
    const unsigned long long top = (1ull << wsize) - 1;
    for (int k = 0; k < nb; ++k) {
        bool last = (k == nb - 1);
        unsigned long long b = acc >> wsize;
        bool hi = (b == top);
        if (hi || last) {
            if (adr != 0) ++m.n_out;                     // filterbank.py:136-142: an event with adr != 0 emits
            if (!last && adr < n_mel) m.dense[size_t(adr) * nb + k] = 1u << wsize;
            adr = last ? 0 : adr + 1;
            acc = 0;
        } else {
            if (adr >= 1 && adr - 1 < n_mel)
                m.dense[size_t(adr - 1) * nb + k] = uint32_t((1ull << wsize) - b);
            if (adr < n_mel) m.dense[size_t(adr) * nb + k] = uint32_t(b);
            acc = (acc + steps[adr < nseg ? adr : nseg - 1]) & (max_acc - 1);
        }
    }
    return m;
}

}  // namespace mfcc_tables
