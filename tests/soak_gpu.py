"""Randomized differential soak (a tool; tests/test_gpu_soak_slice.py runs a fixed slice of it under pytest):
`python tests/soak_gpu.py [seed] [seconds]`.

Random channel counts, lengths, channel strides and base offsets (all alignment residues), history halo,
framing mode, n_cep 1..32 and six signal kinds (Gaussian at three levels, full-scale uniform, Gaussian
with a stretch of silence, DC, full-scale square, pure sine).  Fixed contract: the fused fixed-point
kernel against oracle/mfcc_fixed.py, bit for bit, EVERY signal kind.  Float contract: the fused 512 (random sample
rates, 16-filter banks) and 1024 (its per-rate schedules) kernels AND the generic kernel, each against the float64
oracle on every channel, and against each other.

How the float comparison is made (round 3; round 2 set aside whole FRAMES, 0.17 % of them).  Every handle also runs
with all n_mel coefficients, so the DCT (orthonormal) can be undone: log-mel = coefficients @ B.  The error is taken
band by band in the log-mel domain, the bands that are beyond fp32's reach are zeroed, and what is left is carried
back through the DCT and held to the contract: 1e-4 of the largest coefficient.  A band is beyond reach when its
energy is below 1e-7 of the frame's MEAN BIN POWER: a band that is a single complex FFT bin (44.1 / 48 kHz have
several) cancels that far about once in 1e7 band-frames, and an fp32 FFT's error there -- 1e-7 of the frame's rms --
is then as large as the band itself (tools/replay_band.py on case 10050240: bin 2 at 2.9e-12 next to 4e-4; both
kernels off by 0.04..0.06 in log2).  DC-only bands are NEVER set aside: that bin is summed exactly.
The summary counts the bands set aside and prints the largest log-mel error inside that set.  The kernel with the
drawn n_cep is compared with the first n_cep columns of the all-coefficient run (1e-6: the same log-mel values
through more DCT rows).  Frames with a silent band (-inf log-mel) are compared by their -inf / NaN pattern, exactly.
DC / square / sine inputs have mel bands at the fp32 noise floor where fp32 FFTs legitimately differ from float64
after the log: for the float contract their error is collected and its distribution printed,
not counted, unless FUZZ_STRICT is set.  No sample rate is masked.
Every case has its own seed, printed with the failure: `python tests/soak_gpu.py --case SEED` replays it.
FUZZ_FIXED=1 restricts the run to the fixed contract."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import torch, mfcc_amd
from oracle import mfcc_fixed as mx
from oracle import mfcc_float as mf


def signal(rng, n, kind):
    if kind == 0: x = rng.standard_normal(n) * rng.choice([30, 3000, 12000])
    elif kind == 1: x = rng.integers(-32768, 32768, n).astype(np.float64)
    elif kind == 2:
        x = rng.standard_normal(n) * 3000
        a, b = sorted(rng.integers(0, n + 1, 2)); x[a:b] = 0                     # a stretch of silence
    elif kind == 3: x = np.full(n, rng.integers(-32768, 32768), dtype=np.float64)  # DC
    elif kind == 4: x = 32767 * np.sign(np.sin(np.arange(n) * rng.uniform(0.01, 3.0)))   # full-scale square
    else: x = 20000 * np.sin(np.arange(n) * rng.uniform(0.001, 3.1))
    return np.clip(x, -32768, 32767).astype(np.int16)


ILL = [0, 0]       # bands of noise-like channels set aside / compared (see the docstring)
EXCL_MAX = [0.0]   # largest |log-mel error| inside the set-aside bands (noise-like channels)
HARD = []          # DC / square / sine channels, float contract: max coefficient error / max |coefficient|, per channel
FRAMES_1K = [0, 0]  # fused 1024 kernel (n_cep <= 32 < n_mel): frames with a band set aside / frames compared
REACH = 1e-7       # a band below this fraction of the frame's mean bin power is beyond fp32's reach


def pattern_ok(a, b):
    fin = np.isfinite(b)
    if not np.array_equal(np.isfinite(a), fin): return False, "finite pattern"
    if not np.array_equal(a[~fin], b[~fin], equal_nan=True): return False, "inf pattern"
    return True, ""


def band_compare(g_full, ref_full, st, B, noise_like):
    """g_full / ref_full: (frames, n_mel) coefficients; st: the oracle's stages.  Returns (ok, why, rel err, frames
    that have a band set aside)."""
    g = g_full.astype(np.float64)
    ok, why = pattern_ok(g, ref_full)
    if not ok: return False, why, np.inf, None
    rows = np.isfinite(ref_full).all(axis=1)
    if not rows.any(): return True, "", 0.0, np.zeros(len(ref_full), bool)
    mel, lm = st["mel"][rows], st["logmel"][rows]
    dl = g[rows] @ B - lm                                        # per-band log-mel error
    w = st["filters"]
    dc_only = (w[:, 0] > 0) & (np.count_nonzero(w[:, 1:], axis=1) == 0)
    out = (mel < REACH * st["power"][rows].mean(axis=1, keepdims=True)) & ~dc_only[None, :]
    if noise_like:
        ILL[0] += int(out.sum()); ILL[1] += out.size
        if out.any(): EXCL_MAX[0] = max(EXCL_MAX[0], float(np.abs(dl[out]).max()))
    dl = np.where(out, 0.0, dl)
    dc = dl @ B.T                                                # back through the DCT: the coefficient error of the kept bands
    rel = float(np.abs(dc).max() / max(np.abs(ref_full[rows]).max(), 1.0))
    aside = np.zeros(len(ref_full), bool)
    aside[np.flatnonzero(rows)[out.any(axis=1)]] = True
    return rel <= 1e-4, "err %.3g of the largest coefficient" % rel, rel, aside


def one_case(seed):
    """Returns a list of failure strings (empty: the case passed)."""
    rng = np.random.default_rng(seed)
    fails = []
    if rng.random() < 0.08:                      # ragged batch vs per-utterance calls, bit for bit
        nfft_r = int(rng.choice([512, 1024])); pad = str(rng.choice(["notebook", "stream"]))
        utts = [signal(rng, int(rng.integers(0, 6000)), int(rng.integers(0, 3))) for _ in range(int(rng.integers(1, 12)))]
        nfil_r = (16 if rng.random() < 0.3 else 32) if nfft_r == 512 else 40     # 512 / 16: both fused kernels' other form
        with mfcc_amd.MFCC(nfft=nfft_r, nfilters=nfil_r, nceptrums=int(rng.integers(1, 17)),
                           pad_mode=pad, power_scale=0) as m:
            fl = m.process_batch(utts)
            fx = m.process_batch(utts, fixed=True) if nfft_r == 512 else None
            for i, u in enumerate(utts):
                if not np.array_equal(fl[i], m.process(u), equal_nan=True): fails.append("RAGGED FLOAT %d %s utt %d len %d" % (nfft_r, pad, i, len(u)))
                if fx is not None and not np.array_equal(fx[i], m.process_fixed(u)): fails.append("RAGGED FIXED %s utt %d len %d" % (pad, i, len(u)))
        return fails
    big = rng.random() < 0.5
    cfg = "x512" if os.environ.get("FUZZ_FIXED") else rng.choice(["f512", "f1024", "x512"])
    nfft, hop = (1024, 341) if cfg == "f1024" else (512, 170)
    nch = int(rng.integers(1, 5)); halo = int(rng.integers(0, 2)); pad = str(rng.choice(["notebook", "stream"]))
    n = int(rng.integers(0, 40000 if big else 3 * nfft))
    ncep = int(rng.integers(1, 33))
    stride = n + halo + int(rng.integers(0, 9)); off = int(rng.integers(0, 8))
    flat = np.zeros(off + stride * nch + 16, dtype=np.int16)
    kinds = [int(rng.integers(0, 6)) for _ in range(nch)]
    for c in range(nch): flat[off + c * stride: off + c * stride + n + halo] = signal(rng, n + halo, kinds[c])
    dev = torch.from_numpy(flat).cuda()
    view = torch.as_strided(dev, (nch, n + halo), (stride, 1), storage_offset=off)
    tag = "%s nch %d n %d halo %d %s ncep %d stride %d off %d kinds %s" % (cfg, nch, n, halo, pad, ncep, stride, off, kinds)
    if cfg == "x512":
        nfil = 16 if rng.random() < 0.3 else 32               # 16: the constructor default, the kernel's other instantiation
        ncep = min(ncep, nfil)
        tag += " nfil %d" % nfil
        with mfcc_amd.MFCC(nfft=512, nfilters=nfil, nceptrums=ncep, pad_mode=pad) as m:
            got = m.process_fixed(view, halo=halo).cpu().numpy()
        for c in range(nch if n < 6000 else 1):
            x = flat[off + c * stride: off + c * stride + n + halo]
            if halo:            # the oracle has no halo argument: put the shard one hop into a longer stream
                ref = mx.mfcc_fixed_ref(np.concatenate([np.zeros(169, np.int16), x]), nfilters=nfil, nceptrums=ncep, pad_mode=pad)[1:]
                ok = np.array_equal(got[c][: len(ref)], ref[: len(got[c])])
            else:
                ok = np.array_equal(got[c], mx.mfcc_fixed_ref(x, nfilters=nfil, nceptrums=ncep, pad_mode=pad))
            if not ok:
                fails.append("FIXED %s ch %d" % (tag, c))
                break
        return fails
    nmel = (16 if rng.random() < 0.25 else 32) if cfg == "f512" else 40
    sr = int(rng.choice([16000, 16000, 8000, 22050, 44100, 48000] if cfg == "f512" else [16000, 16000, 8000, 11025, 22050, 32000, 44100, 48000]))
    ncep = min(ncep, nmel)
    kw = dict(nfft=nfft, nfilters=nmel, pad_mode=pad, samplerate=sr, power_scale=512.0 if cfg == "f512" else 0)
    tag += " nmel %d sr %d" % (nmel, sr)
    with mfcc_amd.MFCC(nceptrums=ncep, **kw) as a, mfcc_amd.MFCC(nceptrums=nmel, **kw) as af, \
            mfcc_amd.MFCC(impl="generic", nceptrums=nmel, **kw) as bf:
        ga = a.process(view, halo=halo).cpu().numpy()
        gaf = af.process(view, halo=halo).cpu().numpy(); gbf = bf.process(view, halo=halo).cpu().numpy()
        names = (af.kernel_name(), bf.kernel_name())
        name_a = a.kernel_name()
    if ga.shape[1] == 0:
        return fails                               # no frame: nothing to compare
    B = mf.dct_basis(nmel, nmel)
    for c in range(nch):
        x = flat[off + c * stride: off + c * stride + n + halo]
        xs = np.concatenate([np.zeros(hop - 1, np.int16), x]) if halo else x
        if pad == "stream":                        # the oracle's framing for the stages, like mfcc_float_ref
            nf = mf.num_frames_stream(len(xs), nfft, hop)
            xs = np.concatenate([xs, np.zeros((nf - 1) * hop + nfft - len(xs), dtype=xs.dtype)])
        ref, st = mf.mfcc_notebook(xs, nfft=nfft, hop=hop, n_mel=nmel, sample_rate=sr,
                                   power_scale=512.0 if cfg == "f512" else float(nfft), return_stages=True)
        if halo:
            ref = ref[1:]
            st = {k: (v[1:] if k in ("power", "mel", "logmel") else v) for k, v in st.items()}
        if halo and len(ref) == gaf.shape[1] - 1 and n < nfft:
            continue      # a shard shorter than a frame: the oracle's stream, one hop longer, has no frame of its own here
        noise_like = kinds[c] < 3
        worst, aside = 0.0, None
        for name, g in zip(names, (gaf[c], gbf[c])):
            if len(g) != len(ref):
                fails.append("FLOAT %s %s ch %d: %d frames, oracle %d" % (name, tag, c, len(g), len(ref)))
                continue
            ok, why, rel, asd = band_compare(g, ref, st, B, noise_like)
            worst = max(worst, rel)
            aside = asd if aside is None else aside
            if not ok and (noise_like or os.environ.get("FUZZ_STRICT")):
                fails.append("FLOAT %s vs float64 oracle %s ch %d: %s" % (name, tag, c, why))
        if not noise_like:
            HARD.append(worst)
        a64 = ga[c].astype(np.float64)
        if len(a64) != len(ref):
            continue
        if name_a == names[0]:
            # the drawn n_cep against the first columns of the all-coefficient run: the same log-mel values, fewer DCT rows
            f64 = gaf[c][:, :ncep].astype(np.float64)
            ok, why = pattern_ok(a64, f64)
            fin = np.isfinite(f64)
            if ok and fin.any() and np.abs(a64[fin] - f64[fin]).max() > 1e-6 * max(np.abs(f64[fin]).max(), 1.0):
                ok, why = False, "err %.3g of %.3g" % (np.abs(a64[fin] - f64[fin]).max(), np.abs(f64[fin]).max())
            if not ok:
                fails.append("FLOAT n_cep %d vs n_cep %d of %s %s ch %d: %s" % (ncep, nmel, names[0], tag, c, why))
        elif aside is not None:
            # the fused 1024 kernel keeps at most 32 of its 40 coefficients, so its DCT cannot be undone: it is held to
            # the contract in the coefficient domain on the frames that have no band set aside (counted in FRAMES_1K)
            r64 = ref[:, :ncep]
            ok, why = pattern_ok(a64, r64)
            keep = ~aside & np.isfinite(r64).all(axis=1)
            if noise_like:
                FRAMES_1K[0] += int(aside.sum()); FRAMES_1K[1] += len(aside)
            if ok and keep.any():
                rel = np.abs(a64[keep] - r64[keep]).max() / max(np.abs(r64[keep]).max(), 1.0)
                if rel > 1e-4: ok, why = False, "err %.3g of the largest coefficient" % rel
            if not ok and (noise_like or os.environ.get("FUZZ_STRICT")):
                fails.append("FLOAT %s vs float64 oracle %s ch %d: %s" % (name_a, tag, c, why))
    return fails


def summary():
    s = "bands beyond fp32's reach set aside %d of %d (%.4f %%) on noise-like channels, largest log-mel error inside " \
        "that set %.3g" % (ILL[0], ILL[1], 100.0 * ILL[0] / max(ILL[1], 1), EXCL_MAX[0])
    if FRAMES_1K[1]:
        s += "; fused 1024 kernel, coefficient domain: %d of %d frames not compared (a band set aside)" % tuple(FRAMES_1K)
    if HARD:
        h = np.sort(np.array(HARD)[np.isfinite(HARD)])
        if len(h):
            s += "; DC / square / sine channels (float contract, reported): %d, kept-band coefficient error / largest " \
                 "coefficient: median %.2g, p90 %.2g, p99 %.2g, max %.2g, above 1e-4: %d" % (
                     len(h), h[len(h) // 2], h[int(len(h) * 0.9)], h[int(len(h) * 0.99)], h[-1], int((h > 1e-4).sum()))
    return s


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--case":
        print(one_case(int(sys.argv[2])) or "case passes")
        sys.exit(0)
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    t0 = time.time(); cases = nfail = 0; t_said = t0
    while time.time() - t0 < budget:
        if time.time() - t_said > 60:                  # a line a minute: a silent GPU run is taken to be hung
            t_said = time.time(); print("...", cases, "cases,", nfail, "fails after %.0f s" % (t_said - t0), flush=True)
        seed = seed0 * 10_000_000 + cases
        cases += 1
        try:
            for f in one_case(seed):
                nfail += 1; print("MISMATCH [--case %d] %s" % (seed, f), flush=True)
        except Exception as e:
            nfail += 1; print("EXC [--case %d] %s" % (seed, repr(e)[:200]), flush=True)
    print("cases", cases, "fails", nfail, "in %.0f s;" % (time.time() - t0), summary())
