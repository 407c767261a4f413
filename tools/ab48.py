"""A/B of build variants at 48 kHz (the DC path): python tools/ab48.py variant..."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys
sys.path.insert(0, os.path.dirname(%r))
import torch, mfcc_amd
torch.manual_seed(0)
pcm = (torch.randn((64, 9_600_000), device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, samplerate=48000) as m:
    out = torch.empty((64, m.num_frames(9_600_000), 13), device="cuda")
    m.time_launches(pcm, out, warmup=20, iters=10)
    print("%%-10s %%.4f ms" %% (os.environ.get("VAR"), m.time_launches(pcm, out, warmup=2, iters=30)), flush=True)
''' % here
for rnd in range(2):
    for v in sys.argv[1:]:
        env = dict(os.environ, MFCC_HIP_LIB=os.path.join(here, "variants", v + ".so"), VAR=v)
        subprocess.run([sys.executable, "-c", code], env=env, timeout=200)
