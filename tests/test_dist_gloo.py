"""N > 1 path on CPU: world_size-2 `gloo` processes shard one stream by frame ranges (one-sample
halo) and a batch by items, compute their part with the oracle injected as `compute`, gather, and
must reproduce the unsharded oracle result exactly.  (On the GPU box `compute` is MFCC.process and
the backend is nccl = RCCL; the sharding/halo/gather logic under test is the same code.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mfcc_amd import dist as md
from oracle import mfcc_fixed as mx
from oracle import mfcc_float as mf


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _float_compute(samples, halo, n_frames):
    """Oracle stand-in for MFCC.process(..., halo): samples[0] is history when halo == 1."""
    x = np.asarray(samples).astype(np.float64)
    y = mf.pre_emphasis(x)                         # y[j] uses x[j-1]; y[0] is wrong only if halo == 0 is false
    need = halo + (n_frames - 1) * 170 + 512
    if len(y) < need:
        y = np.concatenate([y, np.zeros(need - len(y))])
    fr = np.stack([y[halo + 170 * k: halo + 170 * k + 512] for k in range(n_frames)])
    win = fr * mf.hamming_window(512)
    p = np.abs(np.fft.rfft(win, axis=1) / 512.0) ** 2
    mel = p @ mf.mel_filterbank().T
    return (np.log2(mel) @ mf.dct_basis(32, 32).T)[:, :13]


def _fixed_compute(samples, halo, n_frames):
    x = np.asarray(samples).astype(np.int64)
    need = halo + (n_frames - 1) * 170 + 512
    if len(x) < need:
        x = np.concatenate([x, np.zeros(need - len(x), np.int64)])
    y = mx.preemph(x)
    if halo == 0:
        pass                                        # history 0 at the start of the stream
    fr = np.stack([y[halo + 170 * k: halo + 170 * k + 512] for k in range(n_frames)])
    w = mx.window_apply(fr, mx.window_curve())
    re, im = mx.fft_fixed(w, 512)
    lg = mx.log2_fix(mx.filterbank(mx.power_spectrum(re, im)))
    return mx.dct_fixed(lg)[:, :13].astype(np.float32)


def _worker(rank, world, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pcm = mf.synth_pcm(170 * 101 + 512 + 37, seed=4)
        if kind == "float":
            shard, loc = md.process_frames_sharded(_float_compute, pcm, rank, world, 13)
        elif kind == "fixed_stream":
            nf = mf.num_frames_stream(len(pcm))
            shard, loc = md.process_frames_sharded(_fixed_compute, pcm, rank, world, 13, n_frames=nf)
        elif kind == "corpus":                       # one batch call per rank (MFCC.process_batch in production)
            items = [mf.synth_pcm(3000 + 500 * i, seed=i) for i in range(5)]
            calls = []

            def batch(us):
                calls.append(len(us))
                return [mf.mfcc_float_ref(u) for u in us]
            idx, outs = md.process_corpus_sharded(batch, items, rank, world)
            assert calls == [len(idx)]               # exactly one launch for the whole shard
            loc = np.concatenate(outs) if outs else np.zeros((0, 13))
            shard = None
        else:                                        # items
            items = [mf.synth_pcm(3000 + 500 * i, seed=i) for i in range(5)]
            idx, outs = md.process_items_sharded(lambda u: mf.mfcc_float_ref(u), items, rank, world)
            loc = np.concatenate(outs) if outs else np.zeros((0, 13))
            shard = None
        t = torch.from_numpy(np.ascontiguousarray(loc, dtype=np.float64))
        full = md.gather_frames(t, 13)                      # all_gather
        root = md.gather_frames(t, 13, dst=0)               # gather to rank 0
        assert (root is None) == (rank != 0)
        if rank == 0:
            assert torch.equal(root, full)
        q.put((rank, full.numpy(), None if shard is None else (shard.frame_lo, shard.frame_hi, shard.halo)))
    finally:
        dist.destroy_process_group()


def _run(kind, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def test_frame_sharding_with_halo_float():
    res = _run("float")
    pcm = mf.synth_pcm(170 * 101 + 512 + 37, seed=4)
    ref = mf.mfcc_float_ref(pcm)
    assert ref.shape == (102, 13)
    for rank, full, sh in res:
        assert full.shape == ref.shape
        np.testing.assert_allclose(full, ref, rtol=0, atol=1e-9)
    assert res[0][2] == (0, 51, 0) and res[1][2] == (51, 102, 1)


def test_frame_sharding_fixed_stream_padding_is_bit_exact():
    res = _run("fixed_stream")
    pcm = mf.synth_pcm(170 * 101 + 512 + 37, seed=4)
    ref = mx.mfcc_fixed_ref(pcm, nceptrums=13)
    assert ref.shape == (103, 13)
    for rank, full, sh in res:
        assert np.array_equal(full.astype(np.int64), ref.astype(np.int64))


def test_item_sharding():
    res = _run("items")
    items = [mf.synth_pcm(3000 + 500 * i, seed=i) for i in range(5)]
    ref = np.concatenate([mf.mfcc_float_ref(u) for u in items])
    for rank, full, _ in res:
        np.testing.assert_array_equal(full, ref)


def test_corpus_sharding_one_batch_per_rank():
    res = _run("corpus")
    items = [mf.synth_pcm(3000 + 500 * i, seed=i) for i in range(5)]
    ref = np.concatenate([mf.mfcc_float_ref(u) for u in items])
    for rank, full, _ in res:
        np.testing.assert_array_equal(full, ref)


def test_plans():
    assert [len(r) for r in md.plan_items(10, 4)] == [3, 3, 2, 2]
    assert [len(r) for r in md.plan_items(2, 4)] == [1, 1, 0, 0]
    sh = md.plan_frames(9_600_000, 8)
    assert sum(s.n_frames for s in sh) == 56468
    assert sh[0].halo == 0 and all(s.halo == 1 for s in sh[1:])
    for a, b in zip(sh, sh[1:]):
        assert a.frame_hi == b.frame_lo
        assert b.sample_lo == 170 * b.frame_lo - 1
        assert a.sample_hi == 170 * (a.frame_hi - 1) + 512
    # overlap between neighbouring ranks = 342 samples + the history sample
    assert sh[0].sample_hi - sh[1].sample_lo == 343
    tiny = md.plan_frames(600, 4)
    assert [s.n_frames for s in tiny] == [1, 0, 0, 0]
