"""Structural (hardware-shaped) scalar model of the reference RTL -- TEST INFRASTRUCTURE ONLY.

Purpose: an *independent second reading* of the nMigen source used to cross-check the
closed-form, vectorised oracle in ``oracle/mfcc_fixed.py``.  Where that file uses derived
closed forms (standard in-place DIT indices, a precomputed window curve, a filterbank
schedule), this file keeps the hardware's own structure: explicit registers with their
declared bit widths, the three FFT RAM banks with the scheduler's address equations and
bank-select multiplexers, the packed twiddle ROM words, the streaming window counter with
its look-ahead address, the filterbank's two-stage register pipeline, the log FSM and the
DCT fill counter.  It is slow (pure Python ints) and meant for a handful of frames.

It is not cycle accurate: stages are evaluated in dataflow order, one accepted sample per
step, with no back-pressure.  That is sufficient because every stage's result depends
only on the order of accepted samples (hazard analysis of the FFT scheduler: a stage-(s+1)
tap reads words written at least 256 - 2**s >= 128 taps earlier than the 8-cycle pipeline).

Paths cited are relative to the reference root.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.signal import get_window


def _s(v, w):
    """reinterpret the low w bits of v as a signed w-bit value (nMigen signed Signal assign)."""
    v &= (1 << w) - 1
    return v - (1 << w) if v >> (w - 1) else v


def _u(v, w):
    return v & ((1 << w) - 1)


# ----------------------------------------------------------------------------- preemph
class Preemph:
    """mfcc/core/preemph.py:15-30."""

    def __init__(self, width=16):
        self.w = width
        self.odata = 0                      # Signal(signed(width)), reset 0

    def push(self, x):
        y = _s(x + (self.odata >> 5) - self.odata, self.w)   # :24 comb, source.data is signed(width)
        self.odata = _s(x, self.w)                            # :20-21 sync
        return y


# ----------------------------------------------------------------------------- window
class Window:
    """mfcc/core/window.py:45-126, streaming: a counter, a synchronous ROM read port whose
    address is the look-ahead count when a sample is consumed, ``point_r`` latched on odd."""

    def __init__(self, width=16, nfft=512, precision=8):
        self.width, self.nfft, self.precision = width, nfft, precision
        maxheight = 2 ** (precision + 1) - 1
        window = get_window("hamm", nfft, fftbins=True)
        winfull = (window * maxheight).astype(int)
        mem = np.copy(winfull[:nfft // 4][1::2])
        self.off_fst = int(mem[0])
        mem -= self.off_fst
        self.off_lst = int(2 * (winfull[nfft // 4] - self.off_fst))
        self.mem = [int(v) for v in mem]
        self.nb = int(math.log2(nfft))
        self.count = 0
        self.point_r = 0
        # synchronous read port: data register holds mem[addr presented last cycle];
        # before the first sample the port has been presented bits_addr(count=0) -> mem[0 or ~0]
        self.rp_data = self.mem[self._addr(0)]

    def _addr(self, count):
        abits = self.nb - 3
        addr = (count >> 1) & ((1 << abits) - 1)          # count[1:-2]
        if (count >> (self.nb - 2)) & 1:                   # bit_dir
            addr = (~addr) & ((1 << abits) - 1)
        return addr

    def push(self, x, last):
        count = self.count
        pw = self.precision + 1
        msb = (count >> (self.nb - 1)) & 1
        dr = (count >> (self.nb - 2)) & 1
        if msb ^ dr:
            point = _u(self.off_lst - self.rp_data, pw)    # :102-103, point is 9 bits unsigned
        else:
            point = self.rp_data
        if not (count & 1):
            curve = _u(self.off_fst + ((point + self.point_r) >> 1), pw)   # :109-110
        else:
            curve = _u(self.off_fst + point, pw)
            self.point_r = point                                            # :114-115
        # counter (:118-123) and look-ahead ROM address (:93-98)
        nxt = 0 if last else _u(count + 1, self.nb)
        self.rp_data = self.mem[self._addr(nxt)]
        self.count = nxt
        prod = _s(x, self.width) * curve                   # Multiplier(signed(16), 9) -> signed 25
        prod = _s(prod, self.width + pw)
        return _s(prod >> pw, self.width), curve           # mul.o.c[-width:]  (:84)


# ----------------------------------------------------------------------------- FFT
class FFT:
    """mfcc/misc/fft.py:349-486 with TwiddleROM :18-61, Butterfly :64-194, Scheduler :197-346."""

    def __init__(self, size, width=16):
        self.size, self.w = size, width
        self.L = int(math.log2(size))
        p = np.linspace(start=0, stop=np.pi / 2, num=int(size // 4), endpoint=False)
        # packed ROM words exactly as :31-35 builds them (Python ints, then 2*width-bit words)
        self.rom = [_u(int(x.real) | int(x.imag) << width, 2 * width)
                    for x in np.round((1 << (width - 2)) * np.exp(-1j * p))]
        self.bias = (1 << (width - 2) - 1) - 1             # :94 verbatim precedence
        self.bias_width = width - 2
        self.mem = [[(0, 0)] * (size // 2) for _ in range(3)]

    def twiddle(self, addr):
        """:38-59 non-inverted decode of rp_addr (range(size//2))."""
        w = self.w
        sel = (addr >> (self.L - 2)) & 1                   # rp_addr[-1]
        word = self.rom[addr & ((1 << (self.L - 2)) - 1)]  # rp_addr[:-1]
        lo = word & ((1 << w) - 1)
        hi = (word >> w) & ((1 << w) - 1)
        real = _s(hi if sel else lo, w)                    # word_select(sel)
        imag = _s(-lo, w) if sel else _s(hi, w)
        return real, imag

    def butterfly(self, x0, x1, tw):
        w = self.w
        W = 2 * w + 1
        x0r, x0i = x0
        x1r, x1i = x1
        twr, twi = tw
        add_0 = _s(x1r + x1i, W)                           # :153
        mul_0 = _s(add_0 * twr, W)                         # :160
        mul_0b = _s(mul_0 + self.bias, W)                  # :166
        add_1 = _s(twr + twi, W)                           # :167
        sub_0 = _s(twr - twi, W)                           # :168
        mul_1 = _s(x1i * add_1, W)                         # :174
        mul_2 = _s(x1r * sub_0, W)                         # :175
        sub_1 = _s(mul_0b - mul_1, W)                      # :180
        sub_2 = _s(mul_0b - mul_2, W)                      # :181
        # slices of signed signals are UNSIGNED bit vectors (nMigen semantics)
        a1 = _u(sub_1, W) >> self.bias_width
        a2 = _u(sub_2, W) >> self.bias_width
        def fin(x, a, sign):
            t = x + a if sign > 0 else x - a               # signed + unsigned -> wide signed
            return _s(t >> 1, w)                           # [scale_bit:] then assign to signed(width)
        return (fin(x0r, a1, +1), fin(x0i, a2, +1)), (fin(x0r, a1, -1), fin(x0i, a2, -1))

    def load(self, data_real):
        """INIT state :408-425: bit-reversed address, mem0 <- even, mem1 & mem2 <- odd."""
        for a, v in enumerate(data_real):
            rev = int(format(a, "0%db" % self.L)[::-1], 2)
            word = (_s(v, self.w), 0)
            if rev & 1:
                self.mem[1][rev >> 1] = word
                self.mem[2][rev >> 1] = word
            else:
                self.mem[0][rev >> 1] = word

    def run(self):
        half = self.size // 2
        tbits = self.L - 1
        trom_addr = 0
        for stage in range(self.L):
            reads = []
            # all reads of a stage happen (>=128 taps) after the writes they depend on and the
            # writes of this stage never alias its own later reads except in-place mem0[tap]
            new = [list(m) for m in self.mem]
            for tap in range(half):
                pow2 = 1 << stage
                a0 = tap                                                   # :248
                a1 = (((tap << 1) ^ pow2) >> 1) & (half - 1)               # :249-250
                sel0 = ((tap >> (stage - 1)) & 1) if stage != 0 else 0     # :254
                sel1 = stage & 1                                           # :255
                m0 = self.mem[0][a0]
                m1 = self.mem[1][a1]
                m2 = self.mem[2][a1]
                if (sel1, sel0) == (0, 0):
                    x0, x1 = m0, m2
                elif (sel1, sel0) == (0, 1):
                    x0, x1 = m2, m0
                elif (sel1, sel0) == (1, 0):
                    x0, x1 = m0, m1
                else:
                    x0, x1 = m1, m0
                tw = self.twiddle(trom_addr)
                y0, y1 = self.butterfly(x0, x1, tw)
                ysel = (tap >> stage) & 1 if stage < tbits else 0          # :293 bit_select beyond width -> 0
                if ysel:
                    w0, w12 = y1, y0
                else:
                    w0, w12 = y0, y1
                new[0][tap] = w0
                if stage & 1:
                    new[2][tap] = w12                                      # :336-337
                else:
                    new[1][tap] = w12
                # twiddle address step: bit (L-1-stage) of a L-bit value, addr is (L-1) bits (:310-316)
                step = 1 << (self.L - 1 - stage)
                trom_addr = (trom_addr + step) & (half - 1)
            self.mem = new
        return self

    def read(self, addr):
        """:400-406: o.addr MSB clear -> mem0; else mem1 (odd log2 size) / mem2."""
        if not (addr >> (self.L - 1)) & 1:
            return self.mem[0][addr & (self.size // 2 - 1)]
        bank = 1 if self.L % 2 else 2
        return self.mem[bank][addr & (self.size // 2 - 1)]


# ----------------------------------------------------------------------------- power
def power(re, im, width=16, width_output=30):
    """mfcc/core/pow2.py:32 (32-bit unsigned sum of two signed 32-bit squares), :64."""
    r = _u(_s(re * re, 2 * width) + _s(im * im, 2 * width), 2 * width)
    return r >> (2 * width - width_output)


# ----------------------------------------------------------------------------- filterbank
class FilterBank:
    """mfcc/core/filterbank.py:37-144, registers and widths as declared."""

    def __init__(self, points, width=30, width_mul=30, gain=18, width_output=16):
        self.width, self.wm, self.gain, self.wo = width, width_mul, gain, width_output
        self.steps = []
        max_acc = 1 << (2 * width_mul)
        for i in range(len(points) - 1):
            diff = int(points[i + 1]) - int(points[i]) - 1
            self.steps.append((max_acc // diff) - 1 if diff else max_acc - 1)
        self.maxvalrange = int(math.log2(int(points[-1]) - int(points[-3]))) + width + width_mul
        self.i_acc = 0
        self.adr = 0
        self.rega = 0
        self.regb = 0

    def push(self, d, last):
        wm = self.wm
        b = self.i_acc >> wm                                     # i_acc[width_mul:]
        highest = (b == (1 << wm) - 1)
        adr = self.adr
        c = _u(d * b, self.width + wm)                           # Multiplier(width, width_mul)
        # input side (:107-115)
        if highest or last:
            self.adr = 0 if last else adr + 1
            self.i_acc = 0
        else:
            self.i_acc = _u(self.i_acc + self.steps[adr], 2 * wm)
        # output side (:120-142): o_data is combinational from o_regb before this edge
        o_data = (self.regb >> (self.maxvalrange - (self.gain + self.wo))) & ((1 << self.wo) - 1)
        emit = (highest or last) and adr != 0
        M = self.maxvalrange
        if highest or last:
            self.regb = _u(self.rega + (d << wm), M)
            self.rega = 0
        else:
            self.rega = _u(self.rega + c, M)
            self.regb = _u(self.regb + (d << wm) - c, M)
        return o_data if emit else None


# ----------------------------------------------------------------------------- log2
def log2fix(v, width=16, width_output=15):
    """mfcc/core/log.py:107-139 around the FSM of :32-104."""
    precision = width_output - math.ceil(math.log2(width))
    cw = width + precision                                        # Log2FixCalc(width=width+precision)
    x = ((v if v != 0 else 1) << precision) & ((1 << cw) - 1)     # Cat(Const(0, precision), data)
    b = 1 << (precision - 1)
    o = 0
    while x >> (precision + 1):                                   # SHIFT-RIGHT :57-62
        x >>= 1
        o = _u(o + (1 << precision), cw)
    cnt = precision - 1
    z = x
    while True:                                                   # CALC-1 / CALC-2
        a = z & ((1 << (precision + 1)) - 1)                      # mul inputs are precision+1 bits
        c = a * a
        if cnt == 0:
            break
        if (c >> (2 * precision + 1)) & 1:
            z = c >> (precision + 1)
            o = _u(o + b, cw)
        else:
            z = c >> precision
        cnt -= 1
        b >>= 1
    return o & ((1 << width_output) - 1)


# ----------------------------------------------------------------------------- DCT
def dct_stream(x, width=16):
    """mfcc/core/dct_stream.py:23-71: the 8-bit fill counter walks 4 cycles per input."""
    nf = len(x)
    f = FFT(4 * nf, width)
    abits = int(math.log2(4 * nf))
    data = [0] * (4 * nf)
    for cnt_fill in range(4 * nf):
        trig = (cnt_fill & 1) ^ ((cnt_fill >> 1) & 1)
        a = cnt_fill >> 1
        addr = (~a if trig else a) & ((1 << abits) - 1)
        val = x[cnt_fill >> 2] if (cnt_fill & 1) else 0
        data[addr] = val
    f.load(data)
    f.run()
    return [f.read(k)[0] for k in range(nf)]


# ----------------------------------------------------------------------------- chain
def mfcc_frames(pcm, n_frames, nfft=512, nfilters=32, nceptrums=13, sample_rate=16e3):
    """First ``n_frames`` frames of the pipeline mfcc/core/mfcc.py:90-104, zeros fed after EOF
    (mfcc/core/mfcc.py:150-153).  Returns a list of per-stage dicts."""
    from .mfcc_float import get_filter_points
    hop = nfft // 3
    points, _ = get_filter_points(0, sample_rate / 2, nfilters, nfft, sample_rate=sample_rate)
    pre = Preemph()
    need = (n_frames - 1) * hop + nfft
    stream = [pre.push(int(pcm[i]) if i < len(pcm) else 0) for i in range(need)]
    win = Window(nfft=nfft)
    fb = FilterBank(points)
    out = []
    for k in range(n_frames):
        fr = stream[hop * k: hop * k + nfft]
        wv, cv = zip(*[win.push(v, i == nfft - 1) for i, v in enumerate(fr)])
        f = FFT(nfft)
        f.load(wv)
        f.run()
        bins = [f.read(a) for a in range(nfft // 2)]
        pw = [power(r, i) for r, i in bins]
        mel = [m for m in (fb.push(d, i == nfft // 2 - 1) for i, d in enumerate(pw)) if m is not None]
        lg = [log2fix(m) for m in mel]
        dct = dct_stream(lg)
        out.append(dict(framed=list(fr), curve=list(cv), windowed=list(wv),
                        fft_re=[b[0] for b in bins], fft_im=[b[1] for b in bins],
                        power=pw, mel=mel, log=lg, dct=dct,
                        cep=[_s(v, 16) for v in dct[:nceptrums]]))
    return out
