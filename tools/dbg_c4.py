import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, mfcc_amd
n = int(os.environ.get("N", 57_600_000)); nch = 64
pcm = torch.empty((nch, n), dtype=torch.int16, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
for c in range(nch):
    pcm[c] = (torch.randn(n, generator=g, device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
m = mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, power_scale=0)
out = torch.empty((nch, m.num_frames(n), 13), device="cuda")
for _ in range(3): m.process(pcm, out=out)
torch.cuda.synchronize()
for k in (1, 10):
    t = time.perf_counter()
    for _ in range(k): m.process(pcm, out=out)
    torch.cuda.synchronize()
    print("process x%d: %.3f ms per call" % (k, (time.perf_counter() - t) / k * 1e3))
print("time_launches: %.3f ms" % m.time_launches(pcm, out, warmup=1, iters=10))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m.process(pcm, out=out)
e1.record(); torch.cuda.synchronize()
print("torch events around 10 process(): %.3f ms per call" % (e0.elapsed_time(e1) / 10))
