"""Diagnostic: per-phase cycle breakdown of the fused kernel (stamps build). Not part of the product."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mfcc_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", (sys.argv[1] if len(sys.argv) > 1 else "stamps") + ".so")
import mfcc_amd
lib = L.load()
lib.mfcc_hip_debug_read_stamps.argtypes = [C.c_void_p]
nch = 64
pcm = (torch.randn((nch, 9_600_000), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
m = mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13)
out = m.process(pcm); torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
lib.mfcc_hip_debug_read_stamps(buf)
ms = m.time_launches(pcm, out, iters=5, warmup=0)
lib.mfcc_hip_debug_read_stamps(buf)
a = np.array(list(buf), dtype=np.float64)
names = ["Twrite", "Tread", "p2:cfft+power", "B1 wait", "park", "B2 wait", "top+fetch", "rfft32+tw", "melMFMA", "dct+store", "Qwrite", "-"]
tiles = 5 * (56468 + 15) // 16 * nch
print("kernel ms", ms, "tiles", tiles)
for w in range(4):
    pt = a[w*12:w*12+12] / tiles
    print("wave", w, "cycles/tile total %.0f:" % pt.sum(), " ".join("%s=%.0f" % (n, v) for n, v in zip(names, pt)))
