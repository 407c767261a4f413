"""Diagnostic: replay a float soak case (tests/soak_gpu.py seed) with all 32 coefficients and print the per-band
log-mel error of both kernels against the float64 oracle: python tools/replay_band.py SEED"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
import importlib.util, torch, mfcc_amd
from oracle import mfcc_float as mf
spec = importlib.util.spec_from_file_location("soak", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "soak_gpu.py"))
soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
assert rng.random() >= 0.08
big = rng.random() < 0.5
cfg = rng.choice(["f512", "f1024", "x512"]); assert cfg == "f512", cfg
nch = int(rng.integers(1, 5)); halo = int(rng.integers(0, 2)); pad = str(rng.choice(["notebook", "stream"]))
n = int(rng.integers(0, 40000 if big else 3 * 512)); ncep = int(rng.integers(1, 33))
stride = n + halo + int(rng.integers(0, 9)); off = int(rng.integers(0, 8))
flat = np.zeros(off + stride * nch + 16, dtype=np.int16)
kinds = [int(rng.integers(0, 6)) for _ in range(nch)]
for c in range(nch): flat[off + c * stride: off + c * stride + n + halo] = soak.signal(rng, n + halo, kinds[c])
nmel = 16 if rng.random() < 0.25 else 32
sr = int(rng.choice([16000, 16000, 8000, 22050, 44100, 48000]))
print("case", seed, dict(nch=nch, n=n, halo=halo, pad=pad, ncep=ncep, stride=stride, off=off, kinds=kinds, nmel=nmel, sr=sr))
x = flat[off: off + n + halo]
print("signal std %.1f" % x.astype(np.float64).std())
assert halo == 0 and nmel == 32
D = mf.dct_basis(32, 32)
ref, st = mf.mfcc_notebook(x, sample_rate=sr, return_stages=True)
lm = st["logmel"]
pts, _ = mf.get_filter_points(0, sr / 2, 32, 512, sample_rate=sr)
print("points", pts[:8])
xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
for impl in ("auto", "generic"):
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=32, samplerate=sr, impl=impl) as m:
        g = m.process(xd).cpu().numpy().astype(np.float64)
        name = m.kernel_name()
    err = np.abs(g @ D - lm)
    f, b = np.unravel_index(np.argmax(err), err.shape)
    print(name, "max coefficient err %.3g of %.3g; worst log-mel err %.3g at frame %d band %d (ref %.3f); per-band max:" %
          (np.abs(g - ref).max(), np.abs(ref).max(), err.max(), f, b, lm[f, b]), np.round(err.max(axis=0)[:8], 4))
    P = st["power"][f]
    print("   power around that band's bins:", P[max(0, pts[b] - 1): pts[b + 2] + 1], " frame rms power %.3g" % P.mean())
