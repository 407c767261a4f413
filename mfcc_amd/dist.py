"""Frame / utterance sharding across the GPUs of one node (one process per GPU).

The MFCC path has no data-path exchange: a frame depends only on samples
``[hop*k - 1, hop*k + nfft)`` of its own stream (342 samples of overlap with the next frame plus
one pre-emphasis sample -- the RTL's only cross-frame state, mfcc/core/preemph.py:20-28 and the
ring buffer of mfcc/core/frame.py).  So:

* a batch of utterances / channels is split by item (``plan_items``), and
* one long stream is split into contiguous frame ranges; each rank reads its own sample span plus
  a ONE-sample history halo (``plan_frames``) -- nothing is exchanged between GPUs.

The only collective is the optional result gather (13 floats per frame) -- ``torch.distributed``
all_gather / gather, which is RCCL over xGMI with the ``nccl`` backend on the GPU box and ``gloo``
in the CPU test-suite.  The compute itself is injected (``compute``): in production it is
``MFCC.process`` / ``MFCC.process_fixed`` (HIP); the CPU tests inject the oracle, so the N > 1
logic is covered without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np


@dataclass(frozen=True)
class FrameShard:
    rank: int
    frame_lo: int          # frames [frame_lo, frame_hi) of the stream
    frame_hi: int
    sample_lo: int         # samples [sample_lo, sample_hi) are read (sample_lo includes the halo)
    sample_hi: int
    halo: int              # 1 if sample_lo is a history-only sample, 0 at the start of the stream

    @property
    def n_frames(self) -> int:
        return self.frame_hi - self.frame_lo


def split_even(n: int, parts: int) -> List[range]:
    """n items into `parts` contiguous ranges whose sizes differ by at most one."""
    base, extra = divmod(n, parts)
    out, lo = [], 0
    for r in range(parts):
        hi = lo + base + (1 if r < extra else 0)
        out.append(range(lo, hi))
        lo = hi
    return out


def plan_items(n_items: int, world: int) -> List[range]:
    """Utterances / channels per rank (configs 4 and 5 of BASELINE.json: embarrassingly parallel)."""
    return split_even(n_items, world)


def plan_frames(n_samples: int, world: int, nfft: int = 512, hop: int = 170,
                n_frames: Optional[int] = None) -> List[FrameShard]:
    """Contiguous frame ranges of ONE stream, with the sample span each rank must read.

    ``n_frames`` defaults to the notebook count ``(n - nfft)//hop + 1``; pass the STREAM count to
    include the zero-padded tail frame (the rank holding it simply runs past ``n_samples``; the
    kernels zero-pad).  Every rank but the first gets a one-sample history halo."""
    if n_frames is None:
        n_frames = 0 if n_samples < nfft else (n_samples - nfft) // hop + 1
    shards = []
    for rank, fr in enumerate(split_even(n_frames, world)):
        if len(fr) == 0:
            shards.append(FrameShard(rank, fr.start, fr.start, 0, 0, 0))
            continue
        first = fr.start * hop
        last = (fr.stop - 1) * hop + nfft                     # exclusive
        halo = 1 if first > 0 else 0
        shards.append(FrameShard(rank, fr.start, fr.stop, first - halo, min(last, n_samples), halo))
    return shards


def process_frames_sharded(compute: Callable, pcm: np.ndarray, rank: int, world: int, n_cep: int,
                           nfft: int = 512, hop: int = 170, n_frames: Optional[int] = None):
    """This rank's part of one stream: returns (shard, coefficients[shard.n_frames, n_cep]).

    ``compute(samples, halo, n_frames)`` must return ``n_frames`` rows for frames that start at
    ``samples[halo]``, treating ``samples[0]`` as history when ``halo == 1`` and zero-padding
    past the end -- :func:`mfcc_compute` wraps a real :class:`mfcc_amd.MFCC` handle into that shape."""
    shard = plan_frames(len(pcm), world, nfft, hop, n_frames)[rank]
    if shard.n_frames == 0:
        return shard, np.zeros((0, n_cep), dtype=np.float32)
    out = compute(pcm[shard.sample_lo:shard.sample_hi], shard.halo, shard.n_frames)
    return shard, np.asarray(out)


def mfcc_compute(m, fixed: bool = False, device=None) -> Callable:
    """The ``compute`` callback of :func:`process_frames_sharded` for a real ``MFCC`` handle ``m``:
    moves the shard to the GPU, runs ``m.process`` / ``m.process_fixed`` with the history halo, trims to the
    shard's frame count and returns a NumPy array.

    Trimming matters with ``pad_mode="stream"``: a non-final shard run through a STREAM handle yields one extra
    zero-padded tail frame that belongs to nobody (the final shard's last frame IS the stream's padded tail,
    because its sample span ends at the end of the stream)."""
    import torch

    def compute(samples, halo, n_frames):
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        x = torch.as_tensor(np.ascontiguousarray(samples, dtype=np.int16)).to(dev)
        need = (n_frames - 1) * m.hop + m.nfft + int(halo)
        if x.numel() < need and m.num_frames(x.numel() - int(halo)) < n_frames:
            # the zero-padded tail of a STREAM plan run through a NOTEBOOK handle: pad explicitly
            x = torch.cat([x, torch.zeros(need - x.numel(), dtype=torch.int16, device=dev)])
        out = m.process_fixed(x, halo=halo) if fixed else m.process(x, halo=halo)
        assert out.shape[0] >= n_frames, (out.shape, n_frames)
        return out[:n_frames].cpu().numpy()

    return compute


def gather_frames(local, n_cep: int, group=None, dst: Optional[int] = None):
    """Concatenate the per-rank coefficient blocks in rank order (the one collective of the path).

    ``local``: torch tensor [n_local, n_cep] on the backend's device (CUDA for nccl = RCCL, CPU for
    gloo).  Ragged sizes are handled by exchanging the row counts first.  ``dst=None``: every rank
    gets the result (all_gather); otherwise only ``dst`` does (others return None)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts) if counts else 0
    padded = torch.zeros((n_max, n_cep), device=local.device, dtype=local.dtype)
    padded[: local.shape[0]] = local
    if dst is None:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(bufs, padded, group=group)
    else:
        bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
        dist.gather(padded, bufs, dst=dst, group=group)
        if rank != dst:
            return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def process_items_sharded(compute: Callable, items: Sequence, rank: int, world: int):
    """This rank's utterances: returns (indices, [compute(item) for item in mine])."""
    mine = plan_items(len(items), world)[rank]
    return list(mine), [compute(items[i]) for i in mine]


def process_corpus_sharded(batch_compute: Callable, items: Sequence, rank: int, world: int):
    """Config 5 (a corpus of short utterances): this rank's utterances in ONE ragged launch.
    ``batch_compute`` maps a list of utterances to the list of their coefficient arrays --
    ``MFCC.process_batch`` (or ``lambda u: m.process_batch(u, fixed=True)``).  Returns (indices, results)."""
    mine = plan_items(len(items), world)[rank]
    outs = batch_compute([items[i] for i in mine]) if len(mine) else []
    assert len(outs) == len(mine)
    return list(mine), list(outs)
