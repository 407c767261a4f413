"""Diagnostic: per-wave interval clocks of the producer / consumer 1024 kernel (stamps build tools/variants/stampspc.so)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MFCC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", os.environ.get("STAMP_LIB", "stampspc") + ".so")
import torch, mfcc_amd
lib = mfcc_amd.load_library()
lib.mfcc_hip_debug_read_stampspc.argtypes = [C.c_void_p]
pcm = (torch.randn((64, 9_600_000), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
m = mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, power_scale=0)
out = m.process(pcm); torch.cuda.synchronize()
buf = (C.c_ulonglong * 40)()
lib.mfcc_hip_debug_read_stampspc(buf)
ms = m.time_launches(pcm, out, iters=5, warmup=0)
lib.mfcc_hip_debug_read_stampspc(buf)
a = np.array(list(buf), dtype=np.float64).reshape(8, 5)
print("kernel", m.kernel_name(), "ms", ms)
for w in range(8):
    n = max(a[w, 4], 1)
    print("%s %d  A work %.0f  wait1 %.0f  B work %.0f  wait2 %.0f  (clocks per tile, total %.0f)" % (
        "producer" if w < 4 else "consumer", w, a[w, 0] / n, a[w, 1] / n, a[w, 2] / n, a[w, 3] / n, a[w, :4].sum() / n))
