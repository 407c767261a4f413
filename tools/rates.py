"""Kernel rates of the instantiations (diagnostic record for profiles/): python tools/rates.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mfcc_amd
torch.manual_seed(0)
nch, n = 64, 9_600_000
pcm = (torch.randn((nch, n), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
rows = {}
def run(tag, fixed=False, **kw):
    with mfcc_amd.MFCC(**kw) as m:
        nf = m.num_frames(n)
        out = torch.empty((nch, nf, m.nceptrums), device="cuda", dtype=torch.int16 if fixed else torch.float32)
        m.time_launches(pcm, out, fixed=fixed, warmup=20, iters=10)
        ms = m.time_launches(pcm, out, fixed=fixed, warmup=2, iters=20)
        rows[tag] = dict(kernel=m.kernel_name(fixed=fixed), frames=nf * nch, ms=round(ms, 4), gframes_per_s=round(nf * nch / ms / 1e6, 3))
        print(tag, rows[tag], flush=True)
base = dict(nfft=512, nfilters=32, nceptrums=13)
run("float 512/170/32/13 @16 kHz (banded, 12-wave)", **base)
os.environ["MFCC_HIP_FUSED512"] = "w4"
run("same, four-wave form (MFCC_HIP_FUSED512=w4)", **base)
run("float 512 @48 kHz, four-wave form", samplerate=48000, **base)
os.environ.pop("MFCC_HIP_FUSED512")
run("same, generic kernel", impl="generic", **base)
run("float 512 / 32 coefficients", nfft=512, nfilters=32, nceptrums=32)
run("float 512 @8 kHz (dense sets, 12-wave)", samplerate=8000, **base)
run("float 512 @22.05 kHz", samplerate=22050, **base)
run("float 512 @44.1 kHz (dense sets + exact integer DC bin, 12-wave)", samplerate=44100, **base)
run("float 512 @48 kHz (dense sets + exact integer DC bin, 12-wave)", samplerate=48000, **base)
run("float 512 / 16 filters (constructor default)", nfft=512, nfilters=16, nceptrums=16)
run("float 1024/341/40/13 (config 4 kernel)", nfft=1024, nfilters=40, nceptrums=13, power_scale=0)
run("float 1024/341/40/32", nfft=1024, nfilters=40, nceptrums=32, power_scale=0)
run("fixed 512/32/13 (config 3 kernel)", fixed=True, pad_mode="stream", **base)
json.dump(rows, open("gpurun_out/rates.json", "w"), indent=1)
