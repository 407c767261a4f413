"""Diagnostic: per-wave segment times of the twelve-wave 1024 kernel (stamps build tools/variants/stamps1k.so,
tools/build_variant.sh stamps1k -DMFCC_1K12_STAMPS).  Workers: segments of pass 1 / pass 2; helpers: work / barrier."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MFCC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "stamps1k.so")
import torch, mfcc_amd
lib = mfcc_amd.load_library()
lib.mfcc_hip_debug_read_stamps1k.argtypes = [C.c_void_p]
pcm = (torch.randn((64, 9_600_000), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
m = mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, power_scale=0)
out = m.process(pcm); torch.cuda.synchronize()
buf = (C.c_ulonglong * 144)()
lib.mfcc_hip_debug_read_stamps1k(buf)
ms = m.time_launches(pcm, out, iters=5, warmup=0)
lib.mfcc_hip_debug_read_stamps1k(buf)
a = np.array(list(buf), dtype=np.float64).reshape(12, 12)
tiles = 5 * 64 * m.num_frames(9_600_000) / 16.0
print("kernel", m.kernel_name(), "ms", ms, "ticks per tile of the workgroup's two groups (a group's period = 2 half-steps)")
names = ["A0", "A1", "A2", "A3", "B0", "B1", "B2", "B3", "park0", "park1", "col16", "tail"]
seg = ["p1 operands", "p1 fft a", "p1 poll", "p1 store+fft b", "barrier 1", "p2 T reads", "p2 wait1+mfma1+Q", "barrier 2", "p2 fft h0", "p2 wait a0", "p2 mfma0", "p2 fft h1"]
for w in range(12):
    if w < 8:
        per = a[w] / (tiles / 2)
        print("%-6s " % names[w] + "  ".join("%s %.0f" % (seg[k], per[k]) for k in range(12)) + "  | period %.0f" % per.sum())
    else:
        per = a[w] / tiles
        print("%-6s work %.0f  barrier %.0f  | half-step %.0f" % (names[w], per[0], per[4], per.sum()))
