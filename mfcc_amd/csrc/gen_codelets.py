#!/usr/bin/env python3
"""Generates codelets_gen.hpp: straight-line, register-resident FFT codelets for the fused
512-point MFCC kernel (gfx950).  Run:  python3 gen_codelets.py  (writes next to this file).

Two codelets, both radix-2 decimation-in-time with every trivial twiddle removed and every
non-trivial twiddle fused Linzer-Feig style (w = c (1 + i t): two FMAs form (1 + i t) b, four
FMAs add/subtract c times that to a -> 6 FMAs per butterfly instead of 4 mul/add + 4 add):

  rfft32_win : 32 real inputs e[n] (pre-emphasised samples) and 32 per-lane window constants
               w[n]  ->  Y[k] = sum_n e[n] w[n] exp(-2 pi i n k / 32), k = 0..16
               (Y[0], Y[16] real).  The window multiply is folded into the first butterfly
               layer (a w_a +- b w_b = one mul + two FMAs).
  cfft16     : 16 complex inputs -> 16 complex outputs, natural order in, natural order out
               (the bit reversal is register renaming).

The generator *traces*: every emitted statement is also evaluated numerically on random
inputs and the result is checked against numpy.fft before the header is written, so the
header cannot be stale or wrong without this script failing.
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Val:
    """a real SSA value with a free sign (neg modifiers cost nothing on the VALU)"""
    __slots__ = ("name", "sgn", "num")

    def __init__(self, name, sgn, num):
        self.name, self.sgn, self.num = name, sgn, num

    def neg(self):
        return Val(self.name, -self.sgn, -self.num)

    def ref(self):
        return self.name if self.sgn > 0 else "-" + self.name


ZERO = None     # exact zero


class Emit:
    def __init__(self):
        self.lines = []
        self.n = 0
        self.ops = 0

    def _new(self, expr, num):
        self.n += 1
        name = "t%d" % self.n
        self.lines.append("    const float %s = %s;" % (name, expr))
        self.ops += 1
        return Val(name, +1, num)

    @staticmethod
    def lit(c):
        s = "%.9g" % c
        if "." not in s and "e" not in s and "inf" not in s:
            s += ".0"
        return s + "f"

    def add(self, a, b):
        if a is ZERO:
            return b
        if b is ZERO:
            return a
        if a.sgn > 0 and b.sgn > 0:
            return self._new("%s + %s" % (a.name, b.name), a.num + b.num)
        if a.sgn > 0 and b.sgn < 0:
            return self._new("%s - %s" % (a.name, b.name), a.num + b.num)
        if a.sgn < 0 and b.sgn > 0:
            return self._new("%s - %s" % (b.name, a.name), a.num + b.num)
        v = self._new("%s + %s" % (a.name, b.name), -(a.num + b.num))
        return v.neg()

    def sub(self, a, b):
        return self.add(a, b.neg() if b is not ZERO else ZERO)

    def mulc(self, c, x):
        """literal constant times value"""
        if x is ZERO or c == 0.0:
            return ZERO
        if c == 1.0:
            return x
        if c == -1.0:
            return x.neg()
        return self._new("%s * %s" % (self.lit(c * x.sgn), x.name), c * x.num)

    def fmac(self, c, x, y):
        """c * x + y with a literal constant c"""
        if x is ZERO or c == 0.0:
            return y
        if y is ZERO:
            return self.mulc(c, x)
        if c == 1.0:
            return self.add(x, y)
        if c == -1.0:
            return self.sub(y, x)
        cc = c * x.sgn
        return self._new("fmaf(%s, %s, %s)" % (self.lit(cc), x.name, y.ref()), c * x.num + y.num)

    def fmav(self, cname, cnum, x, y, negc=False):
        """(+-cvar) * x + y with a lane-constant variable"""
        s = -1.0 if negc else 1.0
        if x is ZERO:
            return y
        sign = s * x.sgn
        cref = cname if sign > 0 else "-" + cname
        if y is ZERO:
            self.n += 1
            name = "t%d" % self.n
            self.lines.append("    const float %s = %s * %s;" % (name, cref, x.name))
            self.ops += 1
            return Val(name, +1, s * cnum * x.num)
        return self._new("fmaf(%s, %s, %s)" % (cref, x.name, y.ref()), s * cnum * x.num + y.num)


def lf_butterfly(E, a, b, w):
    """(a + w b, a - w b) for complex a, b (pairs of Val/ZERO) and a literal twiddle w"""
    ar, ai = a
    br, bi = b
    c, d = w.real, w.imag
    if abs(d) < 1e-15 and abs(c - 1.0) < 1e-15:                 # w = 1
        return (E.add(ar, br), E.add(ai, bi)), (E.sub(ar, br), E.sub(ai, bi))
    if abs(c) < 1e-15 and abs(d + 1.0) < 1e-15:                 # w = -i : w b = (bi, -br)
        return (E.add(ar, bi), E.sub(ai, br)), (E.sub(ar, bi), E.add(ai, br))
    if abs(c) < 1e-15 and abs(d - 1.0) < 1e-15:                 # w = +i : w b = (-bi, br)
        return (E.sub(ar, bi), E.add(ai, br)), (E.add(ar, bi), E.sub(ai, br))
    assert abs(c) > 0.15, "Linzer-Feig needs cos away from 0"
    t = d / c
    u = E.fmac(-t, bi, br)          # br - t bi
    v = E.fmac(t, br, bi)           # bi + t br
    return ((E.fmac(c, u, ar), E.fmac(c, v, ai)),
            (E.fmac(-c, u, ar), E.fmac(-c, v, ai)))


def cfft_dit(E, xs):
    """natural-order in, natural-order out radix-2 DIT on a list of complex pairs"""
    n = len(xs)
    if n == 1:
        return xs
    ev = cfft_dit(E, xs[0::2])
    od = cfft_dit(E, xs[1::2])
    out = [None] * n
    for k in range(n // 2):
        w = complex(math.cos(2 * math.pi * k / n), -math.sin(2 * math.pi * k / n))
        out[k], out[k + n // 2] = lf_butterfly(E, ev[k], od[k], w)
    return out


def rfft_dit(E, es, ws):
    """real-input DIT: es = values, ws = (name, num) window constants or None.
    Returns X[0..n/2] as complex pairs (imag ZERO where it is exactly 0)."""
    n = len(es)
    if n == 2:
        (a, b) = es
        if ws is None:
            return [(E.add(a, b), ZERO), (E.sub(a, b), ZERO)]
        (wa, wan), (wb, wbn) = ws
        m = E.fmav(wa, wan, a, ZERO)
        return [(E.fmav(wb, wbn, b, m), ZERO), (E.fmav(wb, wbn, b, m, negc=True), ZERO)]
    ev = rfft_dit(E, es[0::2], None if ws is None else ws[0::2])
    od = rfft_dit(E, es[1::2], None if ws is None else ws[1::2])
    X = [None] * (n // 2 + 1)
    for k in range(n // 4 + 1):
        er, ei = ev[k]
        orr, oi = od[k]
        if k == 0:
            X[0] = (E.add(er, orr), ZERO)
            X[n // 2] = (E.sub(er, orr), ZERO)
        elif k == n // 4:
            X[k] = (er, orr.neg() if orr is not ZERO else ZERO)        # E - i O, both real
        else:
            w = complex(math.cos(2 * math.pi * k / n), -math.sin(2 * math.pi * k / n))
            hi, lo = lf_butterfly(E, (er, ei), (orr, oi), w)           # E + wO, E - wO
            X[k] = hi
            X[n // 2 - k] = (lo[0], lo[1].neg() if lo[1] is not ZERO else ZERO)   # conj
    return X


def gen_rfft32():
    rng = np.random.default_rng(1)
    e_num = rng.standard_normal(32) * 1000
    w_num = rng.uniform(0.05, 1.0, 32)
    E = Emit()
    es = [Val("e[%d]" % i, +1, e_num[i]) for i in range(32)]
    ws = [("w[%d]" % i, w_num[i]) for i in range(32)]
    X = rfft_dit(E, es, ws)
    ref = np.fft.fft(e_num * w_num)[:17]
    got = np.array([complex(x[0].num if x[0] is not ZERO else 0.0,
                            x[1].num if x[1] is not ZERO else 0.0) for x in X])
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-12, err
    body = list(E.lines)
    for k, (re, im) in enumerate(X):
        body.append("    yr[%d] = %s;" % (k, re.ref()))
        if im is not ZERO:
            body.append("    yi[%d] = %s;" % (k, im.ref()))
    src = ("// real 32-point DFT of e[n] * w[n] (window folded into the first layer): %d VALU ops\n"
           "__device__ __forceinline__ void rfft32_win(const float (&e)[32], const float (&w)[32],\n"
           "                                           float (&yr)[17], float (&yi)[17]) {\n" % E.ops)
    src += "\n".join(body) + "\n    yi[0] = 0.0f;\n    yi[16] = 0.0f;\n}\n"
    return src, E.ops


def gen_cfft16():
    rng = np.random.default_rng(2)
    x_num = rng.standard_normal(16) + 1j * rng.standard_normal(16)
    E = Emit()
    xs = [(Val("xr[%d]" % i, +1, x_num[i].real), Val("xi[%d]" % i, +1, x_num[i].imag)) for i in range(16)]
    X = cfft_dit(E, xs)
    ref = np.fft.fft(x_num)
    got = np.array([complex(a.num, b.num) for a, b in X])
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-12, err
    body = list(E.lines)
    for k, (re, im) in enumerate(X):
        body.append("    zr[%d] = %s;" % (k, re.ref()))
        body.append("    zi[%d] = %s;" % (k, im.ref()))
    src = ("// complex 16-point DFT, natural order in and out: %d VALU ops\n"
           "__device__ __forceinline__ void cfft16(const float (&xr)[16], const float (&xi)[16],\n"
           "                                       float (&zr)[16], float (&zi)[16]) {\n" % E.ops)
    src += "\n".join(body) + "\n}\n"
    return src, E.ops


def main():
    a, na = gen_rfft32()
    b, nb = gen_cfft16()
    out = ("// GENERATED by gen_codelets.py -- do not edit.  Straight-line FFT codelets (see the\n"
           "// generator's docstring); every statement was traced numerically against numpy.fft.\n"
           "#pragma once\n#include <hip/hip_runtime.h>\n\nnamespace mfcc_codelets {\n\n"
           + a + "\n" + b + "\n}  // namespace mfcc_codelets\n")
    path = os.path.join(HERE, "codelets_gen.hpp")
    with open(path, "w") as f:
        f.write(out)
    print("rfft32_win: %d ops, cfft16: %d ops -> %s" % (na, nb, path))


if __name__ == "__main__":
    sys.exit(main())
