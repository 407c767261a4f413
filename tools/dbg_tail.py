"""Diagnostic: the twelve-wave 512 kernel against the four-wave form (fp32 DCT) on 10 min of noise: frames that differ by > 1e-3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mfcc_amd
from oracle import mfcc_float as mf
pcm = (mf.synth_pcm(9_600_000, seed=0) // 4).astype(np.int16)
with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m:
    a = np.asarray(m.process(pcm)); a2 = np.asarray(m.process(pcm)); name = m.kernel_name()
os.environ["MFCC_HIP_FUSED512"] = "w4"
with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m:
    r = np.asarray(m.process(pcm))
d = np.abs(a - r).max(axis=1)
bad = np.nonzero(d > 1e-3)[0]
print(os.environ.get("MFCC_HIP_LIB", "tree").split("/")[-1], name, "bad frames", len(bad), "of", len(d), "repeatable", np.array_equal(a, a2),
      "| frame-15-only tiles", int((np.bincount(bad // 16) == 1).sum()) if len(bad) else 0)
