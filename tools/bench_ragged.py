"""Ragged vs uniform corpus (diagnostic): one launch over 10 000 utterances, equal lengths vs five different lengths."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mfcc_amd
n_utt, n = 10_000, 160_000
flat = (torch.randn(n_utt * n, device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, samplerate=int(os.environ.get("RAG_SR", "16000"))) as m:
    for name, lens in (("uniform", [n] * n_utt), ("ragged", [n - 997 * (u % 5) for u in range(n_utt)])):
        # ragged: utterances packed back to back (no gaps) in a fresh buffer
        offs = np.zeros(n_utt + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens, dtype=np.uint64)
        buf = torch.cat([flat[u * n:u * n + lens[u]] for u in range(n_utt)]) if name == "ragged" else flat
        fx = bool(int(os.environ.get('RAG_FIXED', '0')))
        out, fo = m.process_packed(buf, offs, fixed=fx)
        torch.cuda.synchronize()
        for _ in range(3): m.process_packed(buf, offs, fixed=fx, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): m.process_packed(buf, offs, fixed=fx, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(name, "frames", int(fo[-1]), "ms per corpus %.3f" % (dt * 1e3), "G frames/s %.3f" % (int(fo[-1]) / dt / 1e9))
