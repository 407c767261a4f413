#!/usr/bin/env python3
"""Generates codelets_gen.hpp: straight-line, register-resident FFT codelets for the fused
512-point MFCC kernel (gfx950), in PACKED fp32 form.  Run:  python3 gen_codelets.py

Why packed: the kernel is VALU-issue limited (two waves per SIMD, about one vector instruction
per 4-5 cycles per wave).  A v_pk_fma_f32 / v_pk_add_f32 issues at the cost of the scalar form and
does two lanes of work, so every complex value lives in one 64-bit register pair (re, im) and a
complex add, a multiplication by -i, a twiddle rotation ... is ONE instruction.  The re/im cross
terms use the VOP3P op_sel / neg_lo / neg_hi modifiers, which hipcc does not produce from vector
expressions -- those ops are emitted as inline asm (pure VALU: no memory counters, no hazards to
pad); constants go in SGPR pairs ("s" constraint).

Codelets (radix-2 decimation in time, trivial twiddles removed, non-trivial ones in the 3-op
Linzer-Feig form  u = b + (-t, t) * (b.y, b.x);  a +- c u):

  cfft16       16 complex in -> 16 complex out, natural order              (74 packed ops)
  rfft32_tw    real 32-point FFT of e[n] w[n] as a 16-point complex FFT of z[m] = (y[2m], y[2m+1])
               with the window folded into its first layer, the real-FFT split, and the per-lane
               twiddle W512^(n2 k1) of the 32 x 16 decomposition applied to columns k1 = 1..15.
               Outputs t[0..15] (column 0 as (Y0, 0)) and the real column 16.   (about 160 packed ops)

The generator *traces*: every emitted op is evaluated numerically and the results are checked
against numpy.fft before the header is written.
"""
import cmath
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def lit(c):
    s = "%.9g" % c
    if "." not in s and "e" not in s and "inf" not in s:
        s += ".0"
    return s + "f"


class CV:
    """a complex value held in a v2f register pair (re, im)"""
    __slots__ = ("name", "num")

    def __init__(self, name, num):
        self.name, self.num = name, complex(num)


class Gen:
    def __init__(self):
        self.lines = []
        self.consts = {}
        self.n = 0
        self.ops = 0

    def const(self, a, b):
        """constant pair (a, b) with |a| == |b|: returns (name of the (m, m) pair, neg_lo, neg_hi) so that
        signs ride on the operand's neg modifiers and the SGPR file holds one pair per magnitude"""
        m = abs(float(a))
        assert abs(abs(float(b)) - m) < 1e-12, (a, b)
        key = round(m, 12)
        if key not in self.consts:
            self.consts[key] = "k%d" % len(self.consts)
        return self.consts[key], int(a < 0), int(b < 0)

    def _new(self, num):
        self.n += 1
        return CV("p%d" % self.n, num)

    def _c(self, stmt_rhs, num):
        r = self._new(num)
        self.lines.append("    const v2f %s = %s;" % (r.name, stmt_rhs))
        self.ops += 1
        return r

    def _asm(self, op, ins, num, op_sel=None, op_sel_hi=None, neg_lo=None, neg_hi=None, prefix="p"):
        """ins: list of (constraint, expr); modifier lists have one entry per source operand"""
        n = len(ins)
        self.n += 1
        r = CV("%s%d" % (prefix, self.n), num)
        mods = ""
        if op_sel is not None and any(op_sel):
            mods += " op_sel:[%s]" % ",".join(str(v) for v in op_sel)
        if op_sel_hi is not None and not all(op_sel_hi):
            mods += " op_sel_hi:[%s]" % ",".join(str(v) for v in op_sel_hi)
        if neg_lo is not None and any(neg_lo):
            mods += " neg_lo:[%s]" % ",".join(str(v) for v in neg_lo)
        if neg_hi is not None and any(neg_hi):
            mods += " neg_hi:[%s]" % ",".join(str(v) for v in neg_hi)
        text = "%s %%0, %s%s" % (op, ", ".join("%%%d" % (i + 1) for i in range(n)), mods)
        ops = ", ".join('"%s"(%s)' % (c, e) for c, e in ins)
        self.lines.append('    v2f %s; asm("%s" : "=v"(%s) : %s);' % (r.name, text, r.name, ops))
        self.ops += 1
        return r

    # ---- plain elementwise ops (hipcc emits v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)
    def add(self, a, b):
        return self._c("%s + %s" % (a.name, b.name), a.num + b.num)

    def sub(self, a, b):
        return self._c("%s - %s" % (a.name, b.name), a.num - b.num)

    # ---- complex ops with cross terms
    def add_mi(self, a, b):          # a + (-i) b = (ar + bi, ai - br)
        return self._asm("v_pk_add_f32", [("v", a.name), ("v", b.name)], a.num - 1j * b.num,
                         op_sel=[0, 1], op_sel_hi=[1, 0], neg_hi=[0, 1])

    def sub_mi(self, a, b):          # a - (-i) b = (ar - bi, ai + br)
        return self._asm("v_pk_add_f32", [("v", a.name), ("v", b.name)], a.num + 1j * b.num,
                         op_sel=[0, 1], op_sel_hi=[1, 0], neg_lo=[0, 1])

    def add_conj(self, a, b):        # a + conj(b)
        return self._asm("v_pk_add_f32", [("v", a.name), ("v", b.name)], a.num + b.num.conjugate(), neg_hi=[0, 1])

    def sub_conj(self, a, b):        # a - conj(b)
        return self._asm("v_pk_add_f32", [("v", a.name), ("v", b.name)], a.num - b.num.conjugate(), neg_lo=[0, 1])

    def rot(self, b, t):             # (br - t bi, bi + t br) = b (1 + i t)
        k, nl, nh = self.const(-t, t)
        return self._asm("v_pk_fma_f32", [("v", b.name), ("s", k), ("v", b.name)], b.num * complex(1.0, t),
                         op_sel=[1, 0, 0], op_sel_hi=[0, 1, 1], neg_lo=[0, nl, 0], neg_hi=[0, nh, 0])

    def axpy(self, u, c, a):         # a + c u  (c real)
        k, nl, nh = self.const(c, c)
        return self._asm("v_pk_fma_f32", [("v", u.name), ("s", k), ("v", a.name)], a.num + c * u.num,
                         neg_lo=[0, nl, 0], neg_hi=[0, nh, 0])

    def axmy(self, u, c, a):         # a - c u
        k, nl, nh = self.const(-c, -c)
        return self._asm("v_pk_fma_f32", [("v", u.name), ("s", k), ("v", a.name)], a.num - c * u.num,
                         neg_lo=[0, nl, 0], neg_hi=[0, nh, 0])

    def mulc(self, d, g):            # g d for a complex literal g: 2 ops
        k1, nl1, nh1 = self.const(g.real, g.real)
        k2, nl2, nh2 = self.const(-g.imag, g.imag)
        m = self._asm("v_pk_mul_f32", [("v", d.name), ("s", k1)], g.real * d.num, neg_lo=[0, nl1], neg_hi=[0, nh1],
                      prefix="q")
        return self._asm("v_pk_fma_f32", [("v", d.name), ("s", k2), ("v", m.name)], g * d.num,
                         op_sel=[1, 0, 0], op_sel_hi=[0, 1, 1], neg_lo=[0, nl2, 0], neg_hi=[0, nh2, 0])

    def mul_mi(self, d):             # (-i) d = (di, -dr): 1 op
        k, _, _ = self.const(1.0, 1.0)
        return self._asm("v_pk_mul_f32", [("v", d.name), ("s", k)], -1j * d.num, op_sel=[1, 0], op_sel_hi=[0, 1],
                         neg_hi=[0, 1])

    def mulv(self, x, tw, twnum, conj=False):
        """x * tw (or conj(x) * tw) for a per-lane complex variable tw = (c, s): 2 ops"""
        m = self._asm("v_pk_mul_f32", [("v", x.name), ("v", tw)], x.num.real * twnum, op_sel=[0, 0], op_sel_hi=[0, 1],
                      prefix="q")
        xv = x.num.conjugate() if conj else x.num
        return self._asm("v_pk_fma_f32", [("v", x.name), ("v", tw), ("v", m.name)], xv * twnum,
                         op_sel=[1, 1, 0], op_sel_hi=[1, 0, 1],
                         neg_lo=[0, 0 if conj else 1, 0], neg_hi=[0, 1 if conj else 0, 0])

    # ---- the last layer in transposed packing: (Re(a + w b), Re(a - w b)) and (Im(a + w b), Im(a - w b))
    def bf_t(self, a, b, w):
        c, d = w.real, w.imag
        y0, y1 = a.num + w * b.num, a.num - w * b.num
        rn, im = complex(y0.real, y1.real), complex(y0.imag, y1.imag)
        if abs(d) < 1e-15 and abs(c - 1.0) < 1e-15:
            R = self._asm("v_pk_add_f32", [("v", a.name), ("v", b.name)], rn, op_sel=[0, 0], op_sel_hi=[0, 0], neg_hi=[0, 1])
            I = self._asm("v_pk_add_f32", [("v", a.name), ("v", b.name)], im, op_sel=[1, 1], op_sel_hi=[1, 1], neg_hi=[0, 1])
            return R, I
        if abs(c) < 1e-15 and abs(d + 1.0) < 1e-15:          # w = -i: (ar + bi, ai - br), (ar - bi, ai + br)
            R = self._asm("v_pk_add_f32", [("v", a.name), ("v", b.name)], rn, op_sel=[0, 1], op_sel_hi=[0, 1], neg_hi=[0, 1])
            I = self._asm("v_pk_add_f32", [("v", a.name), ("v", b.name)], im, op_sel=[1, 0], op_sel_hi=[1, 0], neg_lo=[0, 1])
            return R, I
        assert abs(c) > 0.15
        u = self.rot(b, d / c)
        k, nl, _ = self.const(c, c)
        R = self._asm("v_pk_fma_f32", [("v", u.name), ("s", k), ("v", a.name)], rn,
                      op_sel=[0, 0, 0], op_sel_hi=[0, 1, 0], neg_lo=[0, nl, 0], neg_hi=[0, 1 - nl, 0])
        I = self._asm("v_pk_fma_f32", [("v", u.name), ("s", k), ("v", a.name)], im,
                      op_sel=[1, 0, 1], op_sel_hi=[1, 1, 1], neg_lo=[0, nl, 0], neg_hi=[0, 1 - nl, 0])
        return R, I

    def power(self, R, I):
        """(re0^2 + im0^2, re1^2 + im1^2) with the rounding of fmaf(re, re, im * im)"""
        ew = lambda u, v: complex(u.num.real * v.num.real, u.num.imag * v.num.imag)
        m = self._c("%s * %s" % (I.name, I.name), ew(I, I))
        return self._c("__builtin_elementwise_fma(%s, %s, %s)" % (R.name, R.name, m.name), ew(R, R) + m.num)

    # ---- butterflies
    def bf(self, a, b, w):
        """(a + w b, a - w b)"""
        c, d = w.real, w.imag
        if abs(d) < 1e-15 and abs(c - 1.0) < 1e-15:
            return self.add(a, b), self.sub(a, b)
        if abs(c) < 1e-15 and abs(d + 1.0) < 1e-15:          # w = -i
            return self.add_mi(a, b), self.sub_mi(a, b)
        assert abs(c) > 0.15
        u = self.rot(b, d / c)
        return self.axpy(u, c, a), self.axmy(u, c, a)


def fft_dit(G, xs, leaf):
    n = len(xs)
    if n == 2:
        return list(leaf(xs[0], xs[1]))
    ev = fft_dit(G, xs[0::2], leaf)
    od = fft_dit(G, xs[1::2], leaf)
    out = [None] * n
    for k in range(n // 2):
        out[k], out[k + n // 2] = G.bf(ev[k], od[k], cmath.exp(-2j * math.pi * k / n))
    return out


def fft_dit_power(G, xs, leaf):
    """The same transform with its LAST layer in transposed packing -- a butterfly's two outputs X[k], X[k + n/2] come
    out as (Re X[k], Re X[k + n/2]) and (Im X[k], Im X[k + n/2]), at the same one-op-per-pair cost -- so that |X|^2 of
    two bins is one v_pk_mul + one v_pk_fma instead of two v_mul + two v_fma.  Returns ([(P[k], P[k + n/2])], reference
    pairs): the caller never sees the complex values."""
    n = len(xs)
    ev = fft_dit(G, xs[0::2], leaf)
    od = fft_dit(G, xs[1::2], leaf)
    out = []
    for k in range(n // 2):
        R, I = G.bf_t(ev[k], od[k], cmath.exp(-2j * math.pi * k / n))
        out.append(G.power(R, I))
    return out


ISSUE_DISTANCE = 5      # a lone wave issues a dependent packed op ~9 clocks after its producer and an
                        # independent one every ~6 (scratch/pkdep.hip on MI355X): keep >= 4 other ops between


def schedule(lines, first_use_order=None):
    """List-schedule straight-line statements `v2f NAME = f(operands)` so that every statement comes at
    least ISSUE_DISTANCE slots after the statements that define its operands whenever the DAG has that
    much parallelism; priority = longest path to a sink.  hipcc keeps the order of inline-asm
    statements unless registers force it not to, and it has no latency model for them."""
    import re
    defs, uses = [], []
    for ln in lines:
        m = re.match(r"\s*(?:const )?v2f (\w+)", ln)
        assert m, ln
        defs.append(m.group(1))
        rest = ln[m.end():]
        uses.append(set(re.findall(r"\b[pq]\d+\b", rest)) - {m.group(1)})
    idx = {d: i for i, d in enumerate(defs)}
    preds = [[idx[u] for u in us if u in idx] for us in uses]
    succs = [[] for _ in lines]
    for i, ps in enumerate(preds):
        for j in ps:
            succs[j].append(i)
    height = [0] * len(lines)
    for i in reversed(range(len(lines))):
        height[i] = 1 + max([height[j] for j in succs[i]], default=0)
    placed, slot_of, order = set(), {}, []
    remaining = set(range(len(lines)))
    slot = 0
    while remaining:
        ready = [i for i in remaining if all(j in placed for j in preds[i])]
        def earliest(i):
            return max([slot_of[j] + ISSUE_DISTANCE for j in preds[i]], default=0)
        ok = [i for i in ready if earliest(i) <= slot]
        if ok:
            pick = max(ok, key=lambda i: (height[i], -i))
        else:
            pick = min(ready, key=lambda i: (earliest(i), -height[i], i))
        order.append(pick)
        placed.add(pick)
        slot_of[pick] = slot
        remaining.discard(pick)
        slot += 1
    tight = sum(1 for i in order for j in preds[i] if slot_of[i] - slot_of[j] < ISSUE_DISTANCE)
    return [lines[i] for i in order], tight


def emit_consts(G):
    return "\n".join("    const v2f %s = {%s, %s};" % (name, lit(m), lit(m)) for m, name in G.consts.items())


def gen_cfft16():
    rng = np.random.default_rng(2)
    x = rng.standard_normal(16) + 1j * rng.standard_normal(16)
    G = Gen()
    X = fft_dit(G, [CV("x[%d]" % i, x[i]) for i in range(16)], lambda a, b: (G.add(a, b), G.sub(a, b)))
    ref = np.fft.fft(x)
    err = max(abs(v.num - r) for v, r in zip(X, ref)) / np.abs(ref).max()
    assert err < 1e-12, err
    lines, tight = schedule(G.lines)
    print("cfft16: %d dependences closer than %d slots" % (tight, ISSUE_DISTANCE))
    body = emit_consts(G) + "\n" + "\n".join(lines) + "\n" + \
        "\n".join("    z[%d] = %s;" % (k, v.name) for k, v in enumerate(X))
    src = ("// complex 16-point DFT, natural order in and out: %d packed VALU ops\n"
           "__device__ __forceinline__ void cfft16(const v2f (&x)[16], v2f (&z)[16]) {\n%s\n}\n" % (G.ops, body))
    return src, G.ops


def gen_cfft16_pow():
    rng = np.random.default_rng(12)
    x = rng.standard_normal(16) + 1j * rng.standard_normal(16)
    G = Gen()
    P = fft_dit_power(G, [CV("x[%d]" % i, x[i]) for i in range(16)], lambda a, b: (G.add(a, b), G.sub(a, b)))
    ref = np.abs(np.fft.fft(x)) ** 2
    err = max(max(abs(v.num.real - ref[k]), abs(v.num.imag - ref[k + 8])) for k, v in enumerate(P)) / ref.max()
    assert err < 1e-12, err
    lines, tight = schedule(G.lines)
    print("cfft16_pow: %d dependences closer than %d slots" % (tight, ISSUE_DISTANCE))
    body = emit_consts(G) + "\n" + "\n".join(lines) + "\n" + \
        "\n".join("    pp[%d] = %s;" % (k, v.name) for k, v in enumerate(P))
    src = ("// |X[k]|^2 of the complex 16-point DFT: pp[k] = (|X[k]|^2, |X[k + 8]|^2), k = 0..7 (last layer in transposed\n"
           "// packing, see the generator): %d packed VALU ops\n"
           "__device__ __forceinline__ void cfft16_pow(const v2f (&x)[16], v2f (&pp)[8]) {\n%s\n}\n" % (G.ops, body))
    return src, G.ops


def gen_cfft32_half_pow(h):
    rng = np.random.default_rng(13 + h)
    x = rng.standard_normal(32) + 1j * rng.standard_normal(32)
    G = Gen()
    xl = [CV("xl[%d]" % i, x[i]) for i in range(16)]
    xh = [CV("xh[%d]" % i, x[i + 16]) for i in range(16)]
    a = []
    for n in range(16):
        if h == 0:
            a.append(G.add(xl[n], xh[n]))
        else:
            d = G.sub(xl[n], xh[n])
            if n == 0:
                a.append(d)
            elif n == 8:
                a.append(G.mul_mi(d))
            else:
                a.append(G.mulc(d, cmath.exp(-2j * math.pi * n / 32)))
    P = fft_dit_power(G, a, lambda p, q: (G.add(p, q), G.sub(p, q)))
    ref = np.abs(np.fft.fft(x)[h::2]) ** 2
    err = max(max(abs(v.num.real - ref[k]), abs(v.num.imag - ref[k + 8])) for k, v in enumerate(P)) / ref.max()
    assert err < 1e-12, err
    lines, tight = schedule(G.lines)
    print("cfft32_h%d_pow: %d dependences closer than %d slots" % (h, tight, ISSUE_DISTANCE))
    body = emit_consts(G) + "\n" + "\n".join(lines) + "\n" + \
        "\n".join("    pp[%d] = %s;" % (k, v.name) for k, v in enumerate(P))
    src = ("// |F[2m + %d]|^2 of the complex 32-point DFT of x[n] = xl[n], x[n + 16] = xh[n]: pp[m] = (m, m + 8), m = 0..7\n"
           "// (one decimation-in-frequency step, a 16-point DFT with its last layer in transposed packing): %d packed VALU ops\n"
           "__device__ __forceinline__ void cfft32_h%d_pow(const v2f (&xl)[16], const v2f (&xh)[16], v2f (&pp)[8]) {\n%s\n}\n"
           % (h, G.ops, h, body))
    return src, G.ops


def gen_cfft32_half(h):
    """Half of a complex 32-point DFT by one decimation-in-frequency step: outputs F[2m + h], m = 0..15, of
    the 32 inputs xl[n] = x[n], xh[n] = x[n + 16]:  a[n] = xl[n] + xh[n] (h = 0) or
    (xl[n] - xh[n]) W32^n (h = 1), then the 16-point DFT of a."""
    rng = np.random.default_rng(3 + h)
    x = rng.standard_normal(32) + 1j * rng.standard_normal(32)
    G = Gen()
    xl = [CV("xl[%d]" % i, x[i]) for i in range(16)]
    xh = [CV("xh[%d]" % i, x[i + 16]) for i in range(16)]
    a = []
    for n in range(16):
        if h == 0:
            a.append(G.add(xl[n], xh[n]))
        else:
            d = G.sub(xl[n], xh[n])
            if n == 0:
                a.append(d)
            elif n == 8:
                a.append(G.mul_mi(d))
            else:
                a.append(G.mulc(d, cmath.exp(-2j * math.pi * n / 32)))
    X = fft_dit(G, a, lambda p, q: (G.add(p, q), G.sub(p, q)))
    ref = np.fft.fft(x)[h::2]
    err = max(abs(v.num - r) for v, r in zip(X, ref)) / np.abs(ref).max()
    assert err < 1e-12, err
    lines, tight = schedule(G.lines)
    print("cfft32_h%d: %d dependences closer than %d slots" % (h, tight, ISSUE_DISTANCE))
    body = emit_consts(G) + "\n" + "\n".join(lines) + "\n" + \
        "\n".join("    z[%d] = %s;" % (k, v.name) for k, v in enumerate(X))
    src = ("// outputs F[2m + %d], m = 0..15, of the complex 32-point DFT of x[n] = xl[n], x[n + 16] = xh[n]\n"
           "// (one decimation-in-frequency step, then a 16-point DFT): %d packed VALU ops\n"
           "__device__ __forceinline__ void cfft32_h%d(const v2f (&xl)[16], const v2f (&xh)[16], v2f (&z)[16]) {\n%s\n}\n"
           % (h, G.ops, h, body))
    return src, G.ops


def gen_rfft32_tw():
    rng = np.random.default_rng(1)
    e = rng.standard_normal(32) * 1000
    w = rng.uniform(0.05, 1.0, 32)              # stands for hamming / 64 (the split's 1/2 is in the table)
    twn = np.exp(-2j * np.pi * rng.uniform(0, 1, 16))
    G = Gen()
    eps = [(CV("ep[%d]" % m, complex(e[2 * m], e[2 * m + 1])), "wp[%d]" % m, complex(w[2 * m], w[2 * m + 1]))
           for m in range(16)]

    def leaf(a, b):                             # (a wa + b wb, a wa - b wb), elementwise on (y[2m], y[2m+1])
        (av, awn, aw), (bv, bwn, bw) = a, b
        ew = lambda v, ww: complex(v.num.real * ww.real, v.num.imag * ww.imag)
        m = G._c("%s * %s" % (av.name, awn), ew(av, aw))
        s = G._c("__builtin_elementwise_fma(%s, %s, %s)" % (bv.name, bwn, m.name), m.num + ew(bv, bw))
        d = G._c("__builtin_elementwise_fma(-%s, %s, %s)" % (bv.name, bwn, m.name), m.num - ew(bv, bw))
        return s, d

    Z = fft_dit(G, eps, leaf)                   # Z = FFT16 of z[m] = y[2m] + i y[2m+1], y = e w
    y = e * w
    zref = np.fft.fft(y[0::2] + 1j * y[1::2])
    assert max(abs(v.num - r) for v, r in zip(Z, zref)) / np.abs(zref).max() < 1e-12

    # real-FFT split (the common factor 1/2 lives in the window table, so every Y below is 2x the
    # textbook value; columns 0, 8, 16 are patched up explicitly):
    #   S = Z[k] + conj Z[16-k], D = Z[k] - conj Z[16-k], T = (-i W32^k) D,
    #   Y[k] = S + T,  Y[16-k] = conj(S - T)
    yref = 2.0 * np.fft.fft(y)[:17]
    T = [None] * 16
    for k in range(1, 8):
        S = G.add_conj(Z[k], Z[16 - k])
        D = G.sub_conj(Z[k], Z[16 - k])
        Tk = G.mulc(D, -1j * cmath.exp(-2j * math.pi * k / 32))
        Yk = G.add(S, Tk)
        Yc = G.sub(S, Tk)                       # conj of Y[16-k]
        assert abs(Yk.num - yref[k]) < 1e-9 * abs(yref).max()
        assert abs(Yc.num.conjugate() - yref[16 - k]) < 1e-9 * abs(yref).max()
        T[k] = G.mulv(Yk, "tw[%d]" % k, twn[k])
        T[16 - k] = G.mulv(Yc, "tw[%d]" % (16 - k), twn[16 - k], conj=True)
    # k = 8: Y[8] = conj Z[8] (textbook) -> 2 conj Z[8] in our scaling, so no patch is needed
    assert abs(2 * Z[8].num.conjugate() - 2 * yref[8] / 2) < 1e-9 * abs(yref).max()
    T[8] = G.mulv(Z[8], "tw[8]", twn[8], conj=True)
    T8x = G._c("%s * 2.0f" % T[8].name, 2 * T[8].num)
    T[8] = T8x
    # k = 0: (Y0, Y16) = (2a + 2b, 2a - 2b) * 2 ... textbook Y0 = a + b with Z = (a, b) unscaled; ours is 2x
    k2, _, _ = G.const(2.0, 2.0)
    a0, b0 = Z[0].num.real, Z[0].num.imag
    m0 = G._asm("v_pk_mul_f32", [("v", Z[0].name), ("s", k2)], complex(2 * a0, 2 * a0), op_sel=[0, 0], op_sel_hi=[0, 1],
                prefix="q")
    P = G._asm("v_pk_fma_f32", [("v", Z[0].name), ("s", k2), ("v", m0.name)], complex(2 * a0 + 2 * b0, 2 * a0 - 2 * b0),
               op_sel=[1, 0, 0], op_sel_hi=[1, 1, 1], neg_hi=[0, 1, 0])
    assert abs(P.num.real - yref[0].real) < 1e-9 * abs(yref).max()
    assert abs(P.num.imag - yref[16].real) < 1e-9 * abs(yref).max()
    T[0] = G._c("%s * (v2f){1.0f, 0.0f}" % P.name, complex(P.num.real, 0.0))
    # check the twiddled outputs
    for k in range(1, 16):
        assert abs(T[k].num - yref[k] * twn[k]) < 1e-9 * abs(yref).max(), k
    lines, tight = schedule(G.lines)
    print("rfft32_tw: %d dependences closer than %d slots" % (tight, ISSUE_DISTANCE))
    body = emit_consts(G) + "\n" + "\n".join(lines) + "\n" + \
        "\n".join("    t[%d] = %s;" % (k, v.name) for k, v in enumerate(T)) + "\n    v16 = %s.y;" % P.name
    src = ("// real 32-point DFT of y[n] = e[n] w[n] (x2: the table holds hamming/64), columns 1..15 times the\n"
           "// per-lane twiddle tw[k]; t[0] = (Y[0], 0), v16 = Y[16].  %d packed VALU ops\n"
           "__device__ __forceinline__ void rfft32_tw(const v2f (&ep)[16], const v2f (&wp)[16], const v2f (&tw)[16],\n"
           "                                          v2f (&t)[16], float &v16) {\n%s\n}\n" % (G.ops, body))
    return src, G.ops


def main():
    a, na = gen_rfft32_tw()
    b, nb = gen_cfft16()
    c0, n0 = gen_cfft32_half(0)
    c1, n1 = gen_cfft32_half(1)
    bp, nbp = gen_cfft16_pow()
    c0p, n0p = gen_cfft32_half_pow(0)
    c1p, n1p = gen_cfft32_half_pow(1)
    out = ("// GENERATED by gen_codelets.py -- do not edit.  Packed-fp32 straight-line FFT codelets (see the\n"
           "// generator's docstring); every op was traced numerically against numpy.fft.\n"
           "#pragma once\n#include <hip/hip_runtime.h>\n\nnamespace mfcc_codelets {\n\n"
           "typedef float v2f __attribute__((ext_vector_type(2)));\n\n" + a + "\n" + b + "\n" + c0 + "\n" + c1 + "\n" + bp + "\n" + c0p + "\n" + c1p +
           "\n}  // namespace mfcc_codelets\n")
    path = os.path.join(HERE, "codelets_gen.hpp")
    with open(path, "w") as f:
        f.write(out)
    print("rfft32_tw: %d packed ops, cfft16: %d, cfft32_h0: %d, cfft32_h1: %d; with |X|^2: cfft16_pow %d, cfft32_h0_pow %d, "
          "cfft32_h1_pow %d -> %s" % (na, nb, n0, n1, nbp, n0p, n1p, path))


if __name__ == "__main__":
    sys.exit(main())
