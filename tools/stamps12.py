"""Diagnostic: per-wave work / wait split of the twelve-wave kernel (stamps build tools/variants/stamps12.so)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MFCC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "stamps12.so")
import torch, mfcc_amd
lib = mfcc_amd.load_library()
lib.mfcc_hip_debug_read_stamps12.argtypes = [C.c_void_p]
nch = 64
pcm = (torch.randn((nch, 9_600_000), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
m = mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=int(os.environ.get("STAMP_NCEP", "13")), samplerate=int(os.environ.get("STAMP_SR", "16000")))
out = m.process(pcm); torch.cuda.synchronize()
buf = (C.c_ulonglong * 48)()
lib.mfcc_hip_debug_read_stamps12(buf)
ms = m.time_launches(pcm, out, iters=5, warmup=0)
lib.mfcc_hip_debug_read_stamps12(buf)
a = np.array(list(buf), dtype=np.float64).reshape(12, 4)
print("kernel", m.kernel_name(), "ms", ms)
names = ["A0", "A1", "A2", "A3", "B0", "B1", "B2", "B3", "park0", "park1", "col16", "tail"]
for w in range(12):
    n = a[w, 3]
    print("%-6s work even-h %.0f  odd-h %.0f  per half-step total %.0f (clocks)" % (names[w], 2 * a[w, 0] / n, 2 * a[w, 1] / n, a[w, 2] / n))
