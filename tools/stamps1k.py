"""Diagnostic: per-phase cycle breakdown of the eight-frame-tile 1024 kernel (stamps build tools/variants/stamps1k.so)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MFCC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "stamps1k.so")
import torch, mfcc_amd
lib = mfcc_amd.load_library()
lib.mfcc_hip_debug_read_stamps1k.argtypes = [C.c_void_p]
pcm = (torch.randn((64, 9_600_000), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
m = mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, power_scale=0, samplerate=int(os.environ.get("STAMP_SR", "16000")))
out = m.process(pcm); torch.cuda.synchronize()
buf = (C.c_ulonglong * 52)()
lib.mfcc_hip_debug_read_stamps1k(buf)
ms = m.time_launches(pcm, out, iters=5, warmup=0)
lib.mfcc_hip_debug_read_stamps1k(buf)
a = np.array(list(buf), dtype=np.float64)
names = ["Sread+fetch", "tw+rfft32", "Twrite", "B1 wait", "Tread", "cfft32+pow", "split", "melMFMA", "role", "Q+park", "B2 wait", "-"]
print("kernel", m.kernel_name(), "ms", ms)
for w in range(4):
    pt = a[w * 12:w * 12 + 12] / max(a[48 + w], 1)
    print("wave", w, "clocks/tile total %.0f:" % pt.sum(), " ".join("%s=%.0f" % (n, v) for n, v in zip(names, pt)))
