"""CPU oracle for the FIXED-POINT contract (the nMigen RTL of mfcc/core) -- TEST INFRASTRUCTURE ONLY.

A restatement, in NumPy integer arithmetic vectorised over frames, of what the
reference's RTL pipeline ``mfcc/core/mfcc.py:19-117`` computes for
``MFCC(width=16, nfft, samplerate, nfilters, nceptrums)``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import it.

PARITY STATUS: **partially pinned**.  The RTL cannot be simulated here (nmigen is
not installed and cannot be fetched), and the reference stores no integer outputs
for FFT / filterbank / log / DCT, so for those stages this oracle is "parity
unpinned" against the RTL: it is a careful reading of the source, cross-checked
by (a) an independent structural model (``oracle/mfcc_fixed_structural.py``: the
three-RAM scheduler, the streaming window counter, the filterbank registers, the
log FSM written the way the hardware is) and (b) closeness to the float model.
What IS pinned by data the reference holds (``tests/golden/notebook_known_answers.json``,
extracted from the stored outputs of ``notebook/MFCC.ipynb``): the 64-entry window ROM,
``offsetlast`` and the full 512-entry reconstructed window curve (cell 17), and the 34
mel filter points (cell 28 / MFCC-INT cell 8).

Conventions: ``>>`` on signed values is an arithmetic (floor) shift; ``wrap16`` keeps
the low 16 bits and reinterprets them as signed; all arrays are int64/uint64.
Each function cites the RTL lines it follows (paths relative to the reference root).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.signal import get_window as _get_window

from .mfcc_float import get_filter_points as _float_filter_points
from .mfcc_float import num_frames_stream, num_frames_notebook


def wrap16(v):
    v = np.asarray(v, dtype=np.int64)
    return ((v + 32768) & 0xFFFF) - 32768


def _log2_int(n):
    l = int(n).bit_length() - 1
    assert (1 << l) == n, "power of two expected"
    return l


# --------------------------------------------------------------------------- pre-emphasis

def preemph(x):
    """mfcc/core/preemph.py:18-28.  ``odata`` = previous accepted sample (reset 0);
    ``source.data = sink.data + (odata >> 5) - odata`` truncated to signed 16."""
    x = np.asarray(x).astype(np.int64)
    o = np.concatenate([np.zeros(1, np.int64), x[:-1]])
    return wrap16(x + (o >> 5) - o)


# --------------------------------------------------------------------------- framing

def frames_stream(y, n_frames, nfft=512, hop=170):
    """mfcc/core/frame.py:65-153 with windowlen == nfft (mfcc/core/mfcc.py:41-44):
    frame k = samples [hop*k, hop*k + nfft) of the pre-emphasised stream, no
    intra-frame padding (``padding`` at frame.py:77 is only active when windowlen < nfft)."""
    idx = hop * np.arange(n_frames)[:, None] + np.arange(nfft)[None, :]
    return np.asarray(y, dtype=np.int64)[idx]


# --------------------------------------------------------------------------- window

def window_coeffs(nfft=512, precision=8):
    """mfcc/core/window.py:22-43 ``calc_coeffs``: quarter-window ROM, every other point."""
    maxheight = 2 ** (precision + 1) - 1
    window = _get_window("hamm", nfft, fftbins=True)
    winfull = (window * maxheight).astype(int)
    mem = np.copy(winfull[:nfft // 4][1::2])
    off_fst = int(mem[0])
    mem -= off_fst
    assert max(mem) < 2 ** precision
    off_lst = int(2 * (winfull[nfft // 4] - off_fst))
    return mem.astype(np.int64), off_fst, off_lst


def window_curve(nfft=512, precision=8):
    """mfcc/core/window.py:53-123: the integer curve the RTL multiplies each sample by.

    count bits: msb = count[-1], dir = count[-2], addr = count[1:-2], odd = count[0]
    (:60-66); address inverted on ``dir`` (:94-98); ``point = off_lst - mem`` on
    ``msb ^ dir`` (:102-105); odd samples use ``off_fst + point`` and latch ``point_r``,
    even samples use ``off_fst + ((point + point_r) >> 1)`` (:109-115).  ``point_r`` is 0
    at reset and is 0 again at every frame start (index nfft-1 maps to mem[0] = 0)."""
    mem, off_fst, off_lst = window_coeffs(nfft, precision)
    nb = _log2_int(nfft)
    abits = nb - 3
    amask = (1 << abits) - 1
    pmask = (1 << (precision + 1)) - 1
    curve = np.zeros(nfft, dtype=np.int64)
    point_r = 0
    for c in range(nfft):
        msb = (c >> (nb - 1)) & 1
        dr = (c >> (nb - 2)) & 1
        addr = (c >> 1) & amask
        odd = c & 1
        if dr:
            addr = (~addr) & amask
        point = int(mem[addr])
        if msb ^ dr:
            point = (off_lst - point) & pmask
        if odd:
            curve[c] = (off_fst + point) & pmask
            point_r = point
        else:
            curve[c] = (off_fst + ((point + point_r) >> 1)) & pmask
    return curve


def window_apply(frames, curve, precision=8):
    """mfcc/core/window.py:84: ``mul.o.c[-width:]`` of the signed (16 x 9)-bit product,
    i.e. ``(x * curve) >> (precision + 1)`` (25-bit product, top 16 bits)."""
    return (np.asarray(frames, dtype=np.int64) * curve[None, :]) >> (precision + 1)


# --------------------------------------------------------------------------- FFT

def twiddle_rom(size, width=16):
    """mfcc/misc/fft.py:28-36 + :48-59 (non-inverted): (re, im) int for k in [0, size/2).

    First quadrant: ``np.round(2**(width-2) * exp(-1j * p))``, p = linspace(0, pi/2,
    size//4, endpoint=False); second quadrant (address MSB set): re = im(T[k - size/4]),
    im = -re(T[k - size/4])."""
    q = size // 4
    p = np.linspace(start=0, stop=np.pi / 2, num=int(q), endpoint=False)
    t = np.round((1 << (width - 2)) * np.exp(-1j * p))
    re = np.array([int(x.real) for x in t], dtype=np.int64)
    im = np.array([int(x.imag) for x in t], dtype=np.int64)
    return np.concatenate([re, im]), np.concatenate([im, -re])


def _bitrev(n_bits):
    n = 1 << n_bits
    r = np.zeros(n, dtype=np.int64)
    for i in range(n):
        v = 0
        for b in range(n_bits):
            if i & (1 << b):
                v |= 1 << (n_bits - 1 - b)
        r[i] = v
    return r


def fft_fixed(xr, size, width=16):
    """mfcc/misc/fft.py: ``FFT(size, i/o/m_width=16)`` on real input (imag = 0).

    Load (:413-424): sample at natural address a is stored at bit-reversed a.
    Scheduler (:216-344): stage s = 0..log2(size)-1, tap t = 0..size/2-1 is the standard
    in-place radix-2 DIT butterfly on logical indices i0 = ((t >> s) << (s+1)) | (t mod 2**s),
    i1 = i0 + 2**s (derived from the three-RAM addressing; the structural model in
    ``mfcc_fixed_structural.py`` checks the derivation), twiddle address accumulates
    2**(L-1-s) per tap modulo size/2 (:310-331) -> ((t mod 2**s) << (L-1-s)).
    Butterfly (:93-96, :140-192), bias_width = 14, scale_bit = 1:
        m0 = (x1r + x1i) * twr + 8191
        s1 = m0 - x1i * (twr + twi)      (= x1r*twr - x1i*twi + 8191)
        s2 = m0 - x1r * (twr - twi)      (= x1i*twr + x1r*twi + 8191)
        y0 = wrap16((x0 + (s >> 14)) >> 1),  y1 = wrap16((x0 - (s >> 14)) >> 1)
    Returns (re, im) of bins [0, size/2) like the streams read them out
    (fft_stream.py:24-38, dct_stream.py:36-37)."""
    L = _log2_int(size)
    bias = (1 << (width - 2 - 1)) - 1            # fft.py:94 -- precedence: 1 << (bias_width-1), then -1
    sh = width - 2
    twr_rom, twi_rom = twiddle_rom(size, width)
    xr = np.asarray(xr, dtype=np.int64)
    re = np.zeros_like(xr)
    re[..., _bitrev(L)] = xr
    im = np.zeros_like(re)
    t = np.arange(size // 2)
    for s in range(L):
        j = t & ((1 << s) - 1)
        g = t >> s
        i0 = (g << (s + 1)) | j
        i1 = i0 + (1 << s)
        ta = (j << (L - 1 - s)) & (size // 2 - 1)
        twr = twr_rom[ta]
        twi = twi_rom[ta]
        x0r, x0i = re[..., i0], im[..., i0]
        x1r, x1i = re[..., i1], im[..., i1]
        m0 = (x1r + x1i) * twr + bias
        a1 = (m0 - x1i * (twr + twi)) >> sh
        a2 = (m0 - x1r * (twr - twi)) >> sh
        nr = np.empty_like(re)
        ni = np.empty_like(im)
        nr[..., i0] = wrap16((x0r + a1) >> 1)
        ni[..., i0] = wrap16((x0i + a2) >> 1)
        nr[..., i1] = wrap16((x0r - a1) >> 1)
        ni[..., i1] = wrap16((x0i - a2) >> 1)
        re, im = nr, ni
    return re[..., :size // 2], im[..., :size // 2]


# --------------------------------------------------------------------------- power

def power_spectrum(re, im, width=16, width_output=30):
    """mfcc/core/pow2.py:32,64: ``(re*re + im*im)`` as unsigned 2*width bits, top
    ``width_output`` bits -> ``>> (2*width - width_output)``."""
    r = (re * re + im * im) & ((1 << (2 * width)) - 1)
    return r >> (2 * width - width_output)


# --------------------------------------------------------------------------- filterbank

def filter_points(nfft=512, ntap=32, sample_rate=16e3):
    """mfcc/core/filterbank.py:15-20 (identical arithmetic to NB cell 27); the RTL passes
    ``sample_rate`` as the float 16e3 (mfcc/core/mfcc.py:20,72)."""
    pts, _ = _float_filter_points(0, sample_rate / 2, ntap, nfft, sample_rate=sample_rate)
    return pts


def filter_steps(points, wsize=30):
    """mfcc/core/filterbank.py:22-34 ``calc_filters``."""
    out = []
    max_acc = 1 << (2 * wsize)
    for i in range(len(points) - 1):
        diff = int(points[i + 1]) - int(points[i]) - 1
        if diff:
            step = (max_acc // diff) - 1
        else:
            step = max_acc - 1
        out.append(step)
    return out


def filterbank_schedule(nfft=512, ntap=32, sample_rate=16e3, wsize=30):
    """Data-independent control of mfcc/core/filterbank.py:88-118 for one frame of nfft/2
    bins: per bin the 30-bit ramp ``b = acc >> wsize``, whether the bin is an event
    (``highest | last``) and the filter address at that bin."""
    points = filter_points(nfft, ntap, sample_rate)
    steps = filter_steps(points, wsize)
    nb = nfft // 2
    amask = (1 << (2 * wsize)) - 1
    top = (1 << wsize) - 1
    acc = 0
    adr = 0
    b = np.zeros(nb, dtype=np.uint64)
    event = np.zeros(nb, dtype=bool)
    adrs = np.zeros(nb, dtype=np.int64)
    for k in range(nb):
        last = (k == nb - 1)
        bb = acc >> wsize
        hi = (bb == top)
        b[k] = bb
        event[k] = hi or last
        adrs[k] = adr
        if hi or last:
            adr = 0 if last else adr + 1
            acc = 0
        else:
            acc = (acc + steps[adr]) & amask
    return points, b, event, adrs


def filterbank(power, nfft=512, ntap=32, sample_rate=16e3, width=30, gain=18, width_output=16):
    """mfcc/core/filterbank.py:37-144 with the parameters of mfcc/core/mfcc.py:69-75.

    Output side (:120-142): on a non-event bin ``rega += c; regb += (d << 30) - c`` with
    c = d * b; on an event bin the value emitted (when filter_adr != 0) is
    ``regb[-(gain + width_output):][:width_output]`` of the value *before* this update,
    then ``regb = rega + (d << 30); rega = 0``.  Arithmetic is modulo 2**64 here (the RTL's
    registers are ``maxvalrange`` >= 64 bits; only bits below 47 are read)."""
    wsize = width
    points, b, event, adrs = filterbank_schedule(nfft, ntap, sample_rate, wsize)
    maxvalrange = int(math.log2(int(points[-1]) - int(points[-3]))) + width + wsize
    shift = maxvalrange - (gain + width_output)
    assert shift >= 0 and shift + width_output <= 64
    d_all = np.asarray(power).astype(np.uint64)
    nfr = d_all.shape[0]
    rega = np.zeros(nfr, dtype=np.uint64)
    regb = np.zeros(nfr, dtype=np.uint64)
    out = []
    sh = np.uint64(wsize)
    omask = np.uint64((1 << width_output) - 1)
    with np.errstate(over="ignore"):
        for k in range(nfft // 2):
            d = d_all[:, k]
            if event[k]:
                if adrs[k] != 0:
                    out.append((regb >> np.uint64(shift)) & omask)
                regb = rega + (d << sh)
                rega = np.zeros(nfr, dtype=np.uint64)
            else:
                c = d * b[k]
                rega = rega + c
                regb = regb + (d << sh) - c
    assert len(out) == ntap, (len(out), ntap)
    return np.stack(out, axis=1).astype(np.int64)


# --------------------------------------------------------------------------- log2

def log2_fix(v, width=16, width_output=15):
    """mfcc/core/log.py:107-139 ``Log2Fix(width, width_output)`` wrapping
    ``Log2FixCalc`` (:8-104): Turner's binary logarithm in Q(width_output-precision).precision.

    precision = width_output - ceil(log2(width)) (:114); input ``(v or 1) << precision``
    (:123-126); SHIFT-RIGHT while x >= 2**(precision+1): x >>= 1, o += 2**precision (:57-62);
    then ``precision - 1`` squarings (cnt = precision-1 .. 1, :65, :80-101): c = z*z;
    if c[2*precision+1]: z = c >> (precision+1), o += b else z = c >> precision; b >>= 1,
    b starting at 2**(precision-1).  Output ``o[:width_output]``."""
    precision = width_output - math.ceil(math.log2(width))
    v = np.asarray(v, dtype=np.int64)
    x = np.where(v == 0, 1, v) << precision
    o = np.zeros_like(x)
    for _ in range(width + precision):
        m = x >= (1 << (precision + 1))
        if not m.any():
            break
        x = np.where(m, x >> 1, x)
        o = np.where(m, o + (1 << precision), o)
    z = x
    b = 1 << (precision - 1)
    for _ in range(precision - 1):
        zz = z & ((1 << (precision + 1)) - 1)          # mul inputs are precision+1 bits (:15)
        c = zz * zz
        hi = ((c >> (2 * precision + 1)) & 1) == 1
        z = np.where(hi, c >> (precision + 1), c >> precision)
        o = np.where(hi, o + b, o)
        b >>= 1
    return o & ((1 << width_output) - 1)


# --------------------------------------------------------------------------- DCT

def dct_fixed(x, nfilters=32, width=16):
    """mfcc/core/dct_stream.py:6-73: ``FFT(size = 4 * nfilters)`` of the sequence
    y[2n+1] = y[4*nfilters - 1 - 2n] = x[n], zeros elsewhere (:23-33, each input held four
    cycles :44); output Re of bins 0..nfilters-1 (:36-37)."""
    x = np.asarray(x, dtype=np.int64)
    size = 4 * nfilters
    y = np.zeros(x.shape[:-1] + (size,), dtype=np.int64)
    n = np.arange(nfilters)
    y[..., 2 * n + 1] = x
    y[..., size - 1 - 2 * n] = x
    re, _ = fft_fixed(y, size, width)
    return re[..., :nfilters]


# --------------------------------------------------------------------------- whole chain

def mfcc_fixed_ref(pcm, nfft=512, nfilters=32, nceptrums=13, sample_rate=16e3,
                   pad_mode="stream", return_stages=False):
    """The RTL pipeline mfcc/core/mfcc.py:90-104 on an int16 stream.

    ``pad_mode="stream"`` reproduces the reference host driver / bench feeding zeros after
    EOF (software/main.c:134-144, mfcc/core/mfcc.py:150-153): (n - nfft)//hop + 2 frames;
    the zeros enter *before* pre-emphasis.  ``pad_mode="notebook"`` keeps only full frames.
    hop = nfft // 3 (mfcc/core/mfcc.py:43).  Returns int16 (channels?, frames, nceptrums)."""
    pcm = np.asarray(pcm)
    if pcm.ndim == 2:
        return np.stack([mfcc_fixed_ref(c, nfft, nfilters, nceptrums, sample_rate, pad_mode)
                         for c in pcm])
    hop = nfft // 3
    n = len(pcm)
    if pad_mode == "stream":
        nf = num_frames_stream(n, nfft, hop)
    elif pad_mode == "notebook":
        nf = num_frames_notebook(n, nfft, hop)
    else:
        raise ValueError(pad_mode)
    need = (nf - 1) * hop + nfft if nf > 0 else 0
    x = pcm.astype(np.int64)
    if need > n:
        x = np.concatenate([x, np.zeros(need - n, dtype=np.int64)])
    y = preemph(x)
    fr = frames_stream(y, nf, nfft, hop)
    curve = window_curve(nfft)
    w = window_apply(fr, curve)
    re, im = fft_fixed(w, nfft)
    p = power_spectrum(re, im)
    mel = filterbank(p, nfft, nfilters, sample_rate)
    lg = log2_fix(mel)
    dct = dct_fixed(lg, nfilters)
    out = dct[:, :nceptrums].astype(np.int16)          # misc/discard.py:27-55
    if return_stages:
        return out, dict(preemph=y, framed=fr, curve=curve, windowed=w, fft_re=re, fft_im=im,
                         power=p, mel=mel, log=lg, dct=dct)
    return out
