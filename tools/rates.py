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
k1 = dict(nfft=1024, nfilters=40, power_scale=0)
run("float 1024/341/40/13 @16 kHz (config 4 kernel: twelve waves, bf16-split)", nceptrums=13, **k1)
for form, what in (("w12", "twelve waves, fp32 lists"), ("f32", "eight waves in lockstep, fp32 lists"),
                   ("bf16", "eight waves in lockstep, bf16-split")):
    os.environ["MFCC_HIP_FUSED1024"] = form
    run("same, %s (MFCC_HIP_FUSED1024=%s)" % (what, form), nceptrums=13, **k1)
os.environ.pop("MFCC_HIP_FUSED1024")
run("float 1024/341/40/32", nceptrums=32, **k1)
run("float 1024/341/40/40 (all coefficients)", nceptrums=40, **k1)
run("float 1024 @8 kHz", nceptrums=13, samplerate=8000, **k1)
run("float 1024 @32 kHz", nceptrums=13, samplerate=32000, **k1)
run("float 1024 @44.1 kHz (round 2: generic kernel)", nceptrums=13, samplerate=44100, **k1)
run("float 1024 @48 kHz (round 2: generic kernel)", nceptrums=13, samplerate=48000, **k1)
run("float 1024 @48 kHz, generic kernel", nceptrums=13, samplerate=48000, impl="generic", **k1)
run("fixed 512/32/13 (config 3 kernel)", fixed=True, pad_mode="stream", **base)
run("fixed 512/16/16 (constructor default)", fixed=True, pad_mode="stream", nfft=512, nfilters=16, nceptrums=16)
run("fixed 256/16/16 (generic fixed kernel)", fixed=True, pad_mode="stream", nfft=256, nfilters=16, nceptrums=16)
json.dump(rows, open("gpurun_out/rates.json", "w"), indent=1)
