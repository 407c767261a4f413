"""Diagnostic: launches of the fused 512 kernel (config 2's shape, 64 channels x 10 min) timed by HIP events -- no result
check, so timing-only experimental builds (tools/variants/*.so through MFCC_HIP_LIB) can be compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mfcc_amd
torch.manual_seed(0)
pcm = (torch.randn((64, 9_600_000), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m:
    out = torch.empty((64, m.num_frames(9_600_000), 13), device="cuda")
    ms = m.time_launches(pcm, out, warmup=5, iters=int(os.environ.get("ITERS", "30")))
    print("kernel", m.kernel_name(), "frames", 64 * m.num_frames(9_600_000), "ms %.4f" % ms, flush=True)
