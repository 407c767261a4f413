"""The real MFCC handle wired into mfcc_amd.dist's frame-range sharding (every shard of every plan on one GPU),
small float transforms on the generic kernel, and the argument checks of the device path."""
import os

import numpy as np
import pytest

from oracle import mfcc_fixed as mx
from oracle import mfcc_float as mf

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def mfcc_amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import mfcc_amd
    return mfcc_amd


@pytest.mark.parametrize("pad_mode", ["notebook", "stream"])
@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_every_frame_shard_through_the_real_handle(mfcc_amd, wav_pcm, pad_mode, fixed, world):
    """plan_frames -> mfcc_compute(MFCC) per rank -> concatenation == the unsharded result, bit for bit
    (one-sample history halo; a non-final shard of a STREAM handle drops its spurious tail frame)."""
    from mfcc_amd import dist as md
    pcm = wav_pcm[:170 * 501 + 512 + 37]
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, pad_mode=pad_mode) as m:
        whole = m.process_fixed(pcm) if fixed else m.process(pcm)
        nf = m.num_frames(len(pcm))
        compute = md.mfcc_compute(m, fixed=fixed)
        parts = []
        for rank in range(world):
            shard, loc = md.process_frames_sharded(compute, pcm, rank, world, 13, n_frames=nf)
            assert loc.shape == (shard.n_frames, 13)
            parts.append(loc)
        got = np.concatenate(parts)
    assert got.shape == whole.shape and np.array_equal(got, whole)
    if fixed:
        assert np.array_equal(whole, mx.mfcc_fixed_ref(pcm, nceptrums=13, pad_mode=pad_mode))


@pytest.mark.parametrize("nfft,nmel", [(64, 8), (128, 16), (256, 20)])
def test_small_float_transforms_on_the_generic_kernel(mfcc_amd, nfft, nmel):
    """`MFCC(nfft=...)` takes any power of two (mfcc.py:20, fft.py:351-353); the float contract for 64 and 128
    runs on the generic kernel instead of failing at the first call."""
    x = mf.synth_pcm(20000, seed=nfft)
    hop = nfft // 3
    for pad in ("notebook", "stream"):
        with mfcc_amd.MFCC(nfft=nfft, nfilters=nmel, nceptrums=nmel, power_scale=0, pad_mode=pad) as m:
            assert m.kernel_name().endswith("generic_kernel")
            got = m.process(x)
        ref = mf.mfcc_float_ref(x, n_cep=nmel, nfft=nfft, hop=hop, n_mel=nmel, power_scale=float(nfft), pad_mode=pad)
        assert got.shape == ref.shape
        e = np.abs(got - ref).max() / np.abs(ref).max()
        assert e <= TOL, (nfft, pad, e)


def test_device_path_refuses_a_bad_out_tensor(mfcc_amd):
    import torch
    x = torch.from_numpy(mf.synth_pcm(5000, seed=1)).cuda()
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m:
        nf = m.num_frames(5000)
        good = torch.empty((1, nf, 13), device="cuda")
        assert m.process(x[None], out=good) is good
        ref = m.process(x)
        assert torch.equal(good[0], ref)
        for bad in (torch.empty((1, nf - 1, 13), device="cuda"), torch.empty((1, nf, 13), device="cuda", dtype=torch.float64),
                    torch.empty((1, nf, 26), device="cuda")[:, :, ::2], torch.empty((1, nf, 13))):
            with pytest.raises(ValueError):
                m.process(x[None], out=bad)
        with pytest.raises(ValueError):
            m.process(x, halo=2)
        # after a device-path call the handle is back on its own stream: a host-path call still works
        assert np.array_equal(m.process(x.cpu().numpy()), ref.cpu().numpy())


def test_four_wave_form_of_the_fused_512_kernel_still_agrees(mfcc_amd, wav_pcm, tmp_path):
    """`MFCC_HIP_FUSED512=w4` (read at handle creation) keeps the four-wave form of the fused 512 kernel, which otherwise
    only runs for the double-precision DC instantiation: same tables and codelets, another staging -- the results agree
    with the twelve-wave form to fp32 rounding and with the notebook to the contract."""
    import subprocess
    import sys
    out = tmp_path / "w4.npy"
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import mfcc_amd\n"
            "from scipy.io import wavfile\n"
            "sr, x = wavfile.read(%r)\n"
            "rows = []\n"
            "for kw in (dict(nfilters=32, nceptrums=32), dict(nfilters=32, nceptrums=13, samplerate=8000), dict(nfilters=16, nceptrums=16)):\n"
            "    with mfcc_amd.MFCC(nfft=512, pad_mode='stream', **kw) as m:\n"
            "        assert m.kernel_name() == 'mfcc_fused512_kernel', m.kernel_name()\n"
            "        rows.append(m.process(np.stack([x, x[::-1].copy()])))\n"
            "np.save(%r, np.concatenate([r.reshape(-1) for r in rows]))\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f2bjrop1.0.wav"), str(out)))
    env = dict(os.environ, MFCC_HIP_FUSED512="w4")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    w4 = np.load(out)
    rows, refs = [], []
    x2 = np.stack([wav_pcm, wav_pcm[::-1].copy()])
    for kw, okw in ((dict(nfilters=32, nceptrums=32), dict(n_cep=32)),
                    (dict(nfilters=32, nceptrums=13, samplerate=8000), dict(n_cep=13, sample_rate=8000)),
                    (dict(nfilters=16, nceptrums=16), dict(n_cep=16, n_mel=16))):
        with mfcc_amd.MFCC(nfft=512, pad_mode="stream", **kw) as m:
            assert m.kernel_name() == "mfcc_fused512_w12_kernel"
            rows.append(m.process(x2))
        refs.append(mf.mfcc_float_ref(x2, pad_mode="stream", **okw))
    w12 = np.concatenate([r.reshape(-1) for r in rows])
    ref = np.concatenate([r.reshape(-1) for r in refs])
    scale = np.abs(ref).max()
    assert w4.shape == w12.shape == ref.shape
    assert np.abs(w4 - w12).max() / scale < 2e-5
    assert np.abs(w4 - ref).max() / scale <= TOL and np.abs(w12 - ref).max() / scale <= TOL


@pytest.mark.gpu
def test_handles_and_sessions_give_their_device_memory_back(mfcc_amd):
    """create / use / destroy in a loop -- handles of every kernel family, ragged batches, streaming sessions: the
    device's free memory (hipMemGetInfo, which also sees the library's own hipMalloc) ends where it started."""
    import torch
    rng = np.random.default_rng(5)
    pcm = rng.integers(-3000, 3000, 60000).astype(np.int16)
    utts = [pcm[:n] for n in (700, 5000, 12345, 60000)]

    def one_round():
        for kw in (dict(nfft=512, nfilters=32, nceptrums=13), dict(nfft=512, nfilters=32, nceptrums=13, samplerate=48000),
                   dict(nfft=1024, nfilters=40, nceptrums=13, power_scale=0), dict(nfft=256, nfilters=16, nceptrums=8)):
            with mfcc_amd.MFCC(pad_mode="stream", **kw) as m:
                m.process(pcm)
                m.process_batch(utts)
                if kw["nfft"] != 1024:
                    m.process_fixed(pcm)
                    m.process_batch(utts, fixed=True)
                with m.stream() as st:
                    st.push(pcm[:4096]); st.push(pcm[4096:9000]); st.flush()

    one_round()                                    # first use: the runtime's own one-time allocations
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        one_round()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)


@pytest.mark.gpu
def test_distinct_handles_on_distinct_host_threads_are_independent(mfcc_amd):
    """include/mfcc_hip.h: a handle is not thread-safe, distinct handles are independent.  Four host threads, each
    with its own handle (own stream, own scratch) and its own input, both contracts, many calls in flight at once:
    every result equals the single-threaded one, bit for bit."""
    import threading
    rng = np.random.default_rng(17)
    inputs = [rng.integers(-8000, 8000, 20000 + 1111 * i).astype(np.int16) for i in range(4)]
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=16, pad_mode="stream") as m:
        want = [(m.process(x), m.process_fixed(x)) for x in inputs]
    errors = []

    def work(i):
        try:
            with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=16, pad_mode="stream") as h:
                for _ in range(40):
                    if not np.array_equal(h.process(inputs[i]), want[i][0], equal_nan=True):
                        errors.append(("float", i))
                    if not np.array_equal(h.process_fixed(inputs[i]), want[i][1]):
                        errors.append(("fixed", i))
        except Exception as e:           # noqa: BLE001 -- reported below, from the main thread
            errors.append((repr(e), i))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors[:4]


@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("pad_mode", ["notebook", "stream"])
def test_host_buffer_pipeline_chunks_equal_the_device_path(mfcc_amd, fixed, pad_mode, monkeypatch):
    """`mfcc_hip_process_i16` on host buffers is a copy pipeline: chunks of whole channels, or frame ranges of a long
    channel with a one-sample history halo, three in flight on two copy streams around the kernel, the caller's pages
    pinned block by block.  Whatever the chunking, the rows are those of ONE launch over the same samples resident on
    the device, bit for bit.  (Chunks are 64 MB by default; 4 MB here so that small inputs walk every path.)"""
    import torch
    monkeypatch.setenv("MFCC_HIP_HOST_CHUNK_MB", "4")
    rng = np.random.default_rng(11)
    shapes = [(1, 21_000_003),           # one long channel: frame-range chunks (41 MB -> 11 chunks), odd length
              (5, 4_100_001),            # channels of 8.2 MB: frame ranges inside every channel
              (7, 1_900_001),            # channels of 3.8 MB: one channel per chunk, more chunks than pipeline slots
              (40, 300_007)]             # many short channels: several channels per chunk
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, pad_mode=pad_mode) as m:
        for nch, n in shapes:
            x = (rng.standard_normal((nch, n)) * 3000).clip(-32768, 32767).astype(np.int16)
            host = m.process_fixed(x) if fixed else m.process(x)
            d = torch.from_numpy(x).cuda()
            dev = (m.process_fixed(d) if fixed else m.process(d)).cpu().numpy()
            assert host.shape == dev.shape == (nch, m.num_frames(n), 13)
            assert np.array_equal(host, dev), (nch, n)
            # a pinned caller buffer (already registered: the library must neither fail nor unpin it)
            xp = torch.from_numpy(x).pin_memory()
            again = m.process_fixed(xp.numpy()) if fixed else m.process(xp.numpy())
            assert np.array_equal(again, dev)
            del d, xp
        # the ragged host path runs through the same pipeline in chunks of whole utterances
        utts = [(rng.standard_normal(int(k)) * 3000).clip(-32768, 32767).astype(np.int16)
                for k in rng.integers(0, 600_000, 50)]
        many = m.process_batch(utts, fixed=fixed)
        for i in (0, 7, 23, 49):
            one = m.process_fixed(utts[i]) if fixed else m.process(utts[i])
            assert np.array_equal(many[i], one, equal_nan=True), i
        assert sum(len(r) for r in many) == sum(m.num_frames(len(u)) for u in utts)
