import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, mfcc_amd
n, nch = 9_600_000, 8
rng = np.random.default_rng(0)
hp = (rng.standard_normal((nch, n)) * 3000).clip(-32768, 32767).astype(np.int16)
m = mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13)
nf = m.num_frames(n)
m.process(hp)
ts = []
for rep in range(int(os.environ.get("REPS", "3"))):
    t = time.perf_counter(); out = m.process(hp); ts.append(time.perf_counter() - t)
dt = float(np.median(ts))
print("process(host, pageable): median %.2f ms  min %.2f ms  %.3f G frames/s  %.1f GB/s of input" % (dt * 1e3, min(ts) * 1e3, nch * nf / dt / 1e9, hp.nbytes / dt / 1e9))
if os.environ.get("ONLY_PROCESS"): sys.exit(0)
# raw copies
d = torch.empty((nch, n), dtype=torch.int16, device="cuda")
src = torch.from_numpy(hp)
pin = src.pin_memory()
for name, s in (("pageable", src), ("pinned", pin)):
    d.copy_(s); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): d.copy_(s, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("H2D %s: %.2f ms = %.1f GB/s" % (name, dt * 1e3, hp.nbytes / dt / 1e9))
o = torch.empty((nch, nf, 13), device="cuda")
ho = torch.empty((nch, nf, 13)); hop = ho.pin_memory()
for name, s in (("pageable", ho), ("pinned", hop)):
    s.copy_(o); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): s.copy_(o, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("D2H %s: %.2f ms = %.1f GB/s" % (name, dt * 1e3, o.numel() * 4 / dt / 1e9))
t = time.perf_counter(); x = hp.copy(); dt = time.perf_counter() - t
print("host memcpy of the input: %.2f ms = %.1f GB/s" % (dt * 1e3, hp.nbytes / dt / 1e9))
# does an async copy from / to pageable memory return before it is done?  and what does pinning in place cost?
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
st = torch.cuda.Stream()
for name, hostptr in (("pageable", hp.ctypes.data),):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = hip.hipMemcpyAsync(d.data_ptr(), hostptr, hp.nbytes, 1, st.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("hipMemcpyAsync H2D %s: call returned after %.2f ms, done after %.2f ms (rc %d)" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3, rc))
t0 = time.perf_counter(); rc = hip.hipHostRegister(hp.ctypes.data, hp.nbytes, 0); t1 = time.perf_counter()
print("hipHostRegister of %.0f MB: %.2f ms (rc %d)" % (hp.nbytes / 1e6, (t1 - t0) * 1e3, rc))
torch.cuda.synchronize()
t0 = time.perf_counter()
rc = hip.hipMemcpyAsync(d.data_ptr(), hp.ctypes.data, hp.nbytes, 1, st.cuda_stream)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("hipMemcpyAsync H2D registered: call returned after %.2f ms, done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
t0 = time.perf_counter(); hip.hipHostUnregister(hp.ctypes.data); t1 = time.perf_counter()
print("hipHostUnregister: %.2f ms" % ((t1 - t0) * 1e3))
