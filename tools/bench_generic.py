"""Kernel time of the parameter sets that run on the generic kernels (diagnostic): python tools/bench_generic.py [variant]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["MFCC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", sys.argv[1] + ".so")
import torch, mfcc_amd
torch.manual_seed(0)
nch, n = 16, 9_600_000
pcm = (torch.randn((nch, n), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
res = {}
only = os.environ.get("GEN_ONLY")
for fixed, nfft, nmel in ((True, 256, 16), (True, 256, 32), (True, 512, 16), (True, 1024, 32), (True, 1024, 64), (True, 512, 32),
                          (False, 256, 32), (False, 1024, 64)):
    if only and only != "%s%d/%d" % ("x" if fixed else "f", nfft, nmel):
        continue
    with mfcc_amd.MFCC(nfft=nfft, nfilters=nmel, nceptrums=13, pad_mode="stream", power_scale=0) as m:
        nf = m.num_frames(n)
        out = torch.empty((nch, nf, 13), device="cuda", dtype=torch.int16 if fixed else torch.float32)
        ms = m.time_launches(pcm, out, fixed=fixed, warmup=1, iters=3)
        key = "%s nfft %d / %d mel" % ("fixed" if fixed else "float", nfft, nmel)
        res[key] = dict(kernel=m.kernel_name(fixed=fixed), ms=round(ms, 3), frames=nf * nch, gframes_per_s=round(nf * nch / ms / 1e6, 4))
        print(key, res[key], flush=True)
json.dump(res, open("gpurun_out/bench_generic_%s.json" % (sys.argv[1] if len(sys.argv) > 1 else "product"), "w"), indent=1)
