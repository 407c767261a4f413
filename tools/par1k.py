"""Diagnostic: parity of whatever 1024 kernel the library picks (MFCC_HIP_LIB / MFCC_HIP_FUSED1024) against the oracle on a
short noise input with ragged edges -- a quick check for experimental builds before they are timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mfcc_amd
from oracle import mfcc_float as mf
x = mf.synth_pcm(16000 * 4 + 123, seed=5)
for sr in (16000, 44100):
    with mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, samplerate=sr, power_scale=0) as m:
        got = m.process(x)
        name = m.kernel_name()
    ref = mf.mfcc_float_ref(x, nfft=1024, hop=341, n_mel=40, sample_rate=sr, power_scale=1024.0)
    print(name, sr, "max err / max |ref| = %.2e" % (np.abs(np.asarray(got) - ref).max() / np.abs(ref).max()))
