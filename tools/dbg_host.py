import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, mfcc_amd, ctypes as C
rng = np.random.default_rng(11)
m = mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13)
for nch, n in [(1, 21_000_003), (5, 4_100_001), (40, 300_007)]:
    x = (rng.standard_normal((nch, n)) * 3000).clip(-32768, 32767).astype(np.int16)
    try:
        host = m.process(x)
        dev = m.process(torch.from_numpy(x).cuda()).cpu().numpy()
        print(nch, n, "equal", np.array_equal(host, dev))
    except Exception as e:
        print(nch, n, "FAILED", e, "hip error", m._lib.mfcc_hip_last_hip_error(m._h))
