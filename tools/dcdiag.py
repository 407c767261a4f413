"""Diagnostic (GPU): where do the fused (dense) and generic fp32 kernels differ from the float64 notebook at
44.1 / 48 kHz?  With nceptrums = 32 the DCT is orthonormal, so the per-band log-mel values can be recovered
from the GPU output (logmel = D^T c) and compared band by band."""
import os, sys, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mfcc_amd
from oracle import mfcc_float as mf
D = mf.dct_basis(32, 32)
res = {}
rng = np.random.default_rng(5)
for sr in (16000, 44100, 48000):
    pts, _ = mf.get_filter_points(0, sr / 2, 32, 512, sample_rate=sr)
    for kind, sigma in (("gauss", 3000), ("gauss", 30), ("uniform", 0)):
        n = 512 + 170 * 39999
        x = (rng.standard_normal(n) * sigma) if kind == "gauss" else rng.integers(-32768, 32768, n).astype(np.float64)
        x = np.clip(x, -32768, 32767).astype(np.int16)
        with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=32, samplerate=sr) as a, \
             mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=32, samplerate=sr, impl="generic") as b:
            ga, gb = a.process(x).astype(np.float64), b.process(x).astype(np.float64)
            ka, kb = a.kernel_name(), b.kernel_name()
        ref, st = mf.mfcc_notebook(x, sample_rate=sr, return_stages=True)
        lm = st["logmel"]                       # (frames, 32)
        la, lb = ga @ D, gb @ D                 # D orthonormal: logmel = c D
        ea, eb = np.abs(la - lm), np.abs(lb - lm)
        cmax = np.abs(ref).max()
        # DC bin of the float64 chain relative to its rms over frames
        x0 = np.array([f[0].real for f in st["fft"]]); rel = np.abs(x0) / x0.std()
        worst = np.argsort(ea[:, 0])[-3:][::-1]
        key = "%d/%s%d" % (sr, kind, sigma)
        res[key] = dict(points=pts[:4].tolist(), kernels=[ka, kb], frames=len(ref),
            cep_err_fused=float(np.abs(ga - ref).max() / cmax), cep_err_generic=float(np.abs(gb - ref).max() / cmax),
            band0_logmel_err=[float(ea[:, 0].max()), float(eb[:, 0].max())],
            other_bands_logmel_err=[float(ea[:, 1:].max()), float(eb[:, 1:].max())],
            worst_frames=[dict(frame=int(f), dc_over_rms=float(rel[f]), err_fused=float(ea[f, 0]), err_generic=float(eb[f, 0]),
                               logmel0=float(lm[f, 0])) for f in worst],
            frames_dc_below_1e_5_rms=int((rel < 1e-5).sum()))
        print(key, json.dumps(res[key]), flush=True)
json.dump(res, open("gpurun_out/dcdiag.json", "w"), indent=1)
