#!/usr/bin/env python3
"""bench.py -- MFCC frames/sec (512-pt, 32 mel, 13 coeff) on N MI355X; % of HBM roofline.

    python bench.py --gpus N --steps K --warmup W        # N > 1: spawns one child process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     # the same ranks under torchrun

Default workload (BASELINE.json configs[1]): synthetic 16 kHz mono PCM, 10 min per channel
(9 600 000 samples -> 56 468 frames at 512/170), batched over 64 channels so one launch has
3.6 M frames (a single 19 MB channel cannot fill the chip).  float32 kernel, 13 coefficients.
A "step" = one pass of the hot path over the rank's batch, input and output resident in HBM.
Each rank owns its own batch (weak scaling, frames shard with no data-path collective --
SURVEY.md 8e); value = frames of all ranks / max-over-ranks time.

`--config 5` (BASELINE.json configs[4]): ONE fixed corpus of 10 000 utterances x 10 s (seed = utterance
id), sharded by utterance over the N ranks (mfcc_amd.dist.plan_items), one launch per rank per step:
strong scaling.  The optional result gather (13 floats per frame, RCCL over xGMI through
torch.distributed) is timed separately and reported as its own field, never inside `value`.  The default
run carries the same measurement as the `config5` sub-object of its line (skip with --no-config5).

Prints ONE JSON line (rank 0) with `roofline` (algorithmic bytes 392 B/frame over the kernel's
HIP-event duration, against the 8 TB/s HBM peak) and `cpu_baseline` (the oracle's restatement
of the reference notebook's NumPy path timed on this host, a bounded sample of the same workload).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
NFFT, HOP, NMEL, NCEP = 512, 170, 32, 13
SAMPLES_PER_CH = 9_600_000     # 10 min @ 16 kHz
BYTES_PER_FRAME = HOP * 2 + NCEP * 4      # 392 B: each sample read once, each output written once
C5_UTTS, C5_SAMPLES = 10_000, 160_000     # config 5: 10 000 utterances of 10 s
PREWARM_S = 0.4                # launches before anything is timed, whatever --warmup says: the clocks of an idle
                               # MI355X ramp over the first ~100 ms of work (round 1: 5 warm-up steps read 5 % slow)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 5],
                    help="2: configs[1], per-rank 64-channel batch, weak scaling (default); 5: configs[4], one fixed "
                         "10k-utterance corpus sharded by utterance, strong scaling")
    ap.add_argument("--channels", type=int, default=64, help="config 2: 10-min channels per GPU per step")
    ap.add_argument("--utterances", type=int, default=C5_UTTS, help="config 5: corpus size")
    ap.add_argument("--impl", default="auto", choices=["auto", "generic", "fused512"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="default run: leave out the config5 sub-object")
    ap.add_argument("--no-gather", action="store_true", help="config 5: do not time the result gather")
    ap.add_argument("--fixed", action="store_true", help="bench the fixed-point kernel instead (config 3)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                         "multi-rank path on a box with fewer GPUs than ranks: all ranks then share cuda:0)")
    ap.add_argument("--host-io", action="store_true",
                    help="also time the host-buffer entry point (H2D + kernel + D2H) on 8 channels; reported "
                         "as pcie_inclusive, never as value")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch

def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes BEFORE this process
    touches the GPU (it never does), relay rank 0's JSON line, fail if any rank fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print("bench.py: ranks failed: %s" % bad, file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------ helpers

class Ctx:
    """One rank: device, process group, timing helpers."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = 0 if args.backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0"))
        assert self.world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (self.world, args.gpus)
        self.dev = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.dev)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group("gloo")

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], device=self.dev if self.args.backend == "nccl" else "cpu",
                              dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, run, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        torch = self.torch
        for _ in range(warmup):
            run()
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        return self.max_over_ranks(time.perf_counter() - t0)

    def prewarm(self, run):
        torch = self.torch
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < PREWARM_S:
            for _ in range(8):
                run()
            torch.cuda.synchronize()

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def profile_figures(kernel_name, frames):
    """HBM bytes per launch and VALU instructions per frame of `kernel_name` from the newest committed rocprofv3
    PMC summary (profiles/summarize_rocprof.py; FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes) whose
    kernel-source stamp matches the kernels this run executes.  Per-frame figures, scaled to this launch."""
    import glob
    import mfcc_amd
    want = mfcc_amd.kernel_source_hash()
    stale = None
    for pj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            prof = json.load(open(pj))
        except Exception:
            continue
        d = prof.get("derived", {})
        if not prof.get("kernel", "").endswith(kernel_name) or "hbm_traffic_bytes_per_launch" not in d:
            continue
        if prof.get("kernel_source_hash") != want:
            stale = stale or os.path.relpath(pj, ROOT)
            continue
        per_frame = d["hbm_traffic_bytes_per_launch"] / d["frames_per_launch"]
        cnt = prof.get("counters", {})
        mf = cnt.get("SQ_INSTS_MFMA", {}).get("avg_per_launch")
        mc = cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("avg_per_launch")
        PROFILE["mfma_per_frame"] = mf / d["frames_per_launch"] if mf else None
        PROFILE["mfma_clk_per_frame"] = mc / d["frames_per_launch"] if mc else None
        return (round(per_frame * frames), os.path.relpath(pj, ROOT), d.get("valu_wave_instructions_per_frame"), None)
    return None, None, None, stale


PROFILE = {"mfma_per_frame": None, "mfma_clk_per_frame": None}


def roofline(frames, bytes_per_frame, kernel_ms, kernel_name, note_extra=""):
    achieved = frames * bytes_per_frame / (kernel_ms * 1e-3) / 1e9
    traffic, src, valu, stale = profile_figures(kernel_name, frames)
    r = {
        "bound": "hbm",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic,
        "traffic_unit": "bytes per launch (algorithmic: %d)" % (frames * bytes_per_frame),
        "traffic_source": src,
        "kernel_ms": round(kernel_ms, 4),
        "note": "algorithmic bytes = frames x %d B / HIP-event kernel time; the path is %s bound "
                "(DESIGN.md), so this fraction is reported as asked, not as the binding limit%s"
                % (bytes_per_frame, "integer-VALU (a 16-bit datapath emulated bit for bit)" if "fixed" in kernel_name
                   else "fp32-VALU/LDS", note_extra),
    }
    if traffic is None and stale:
        r["traffic_dropped"] = "%s was collected on other kernel sources (stamp mismatch)" % stale
    return r, valu, src


# fixed-point kernel: a third of its vector instructions (static count over the loop body) are plain add / shift / and
# ops that hold the pipe 2.4 clocks, the rest (dot2, SDWA, mul24, bfe, perm, v_mad_u64_u32 ...) 3.2 or more
# (tools/int_probe.hip, profiles/r02_int_probe.txt): 0.33 x 2.4 + 0.67 x 3.2
FIXED_PIPE_CLK = 2.94


def valu_roofline(torch, dev, frames, kernel_ms, valu_per_frame, src, fixed=False):
    # the binding limit: every VALU wave-instruction holds a SIMD for 4 clocks (round 1's definition,
    # kept for the float path next to alu_roofline); the integer kernel is priced at its measured instruction mix
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    clk = FIXED_PIPE_CLK if fixed else 4.0
    ceiling = n_cu * 4 * 2.4e9 / (valu_per_frame * clk)
    rate = frames / (kernel_ms * 1e-3)
    return {"bound": "valu", "achieved": round(rate, 1), "peak": round(ceiling, 1), "unit": "frames/s per GPU",
            "frac": round(rate / ceiling, 4),
            "note": "peak = CUs x 4 SIMDs x 2.4 GHz / (%.1f VALU wave-instructions per frame x %s clocks%s), "
                    "instruction count from %s" % (valu_per_frame, clk, " of pipe: a third plain ops at 2.4, the rest at "
                                                   "3.2, tools/int_probe.hip" if fixed else "", src)}


# Measured on MI355X with tools/alu_probe.hip (profiles/r02_alu_probe.txt), two waves per SIMD: a packed fp32 op
# (v_pk_fma/add/mul_f32) or v_dot2 holds the SIMD's vector pipe for 3.26 clocks, a plain fp32 op for 2.55, and
# v_mfma_f32_16x16x4_f32 for 32 -- during which NO vector instruction of either wave issues (MFMA + k VALU ops take
# 32 + 4.5 k clocks even inside one wave; SQ_VALU_MFMA_COEXEC_CYCLES reads 0): on gfx950 the fp32 matrix
# instruction is vector-pipe time, not a second pipe.  So the binding resource is ONE fp32 pipe per SIMD.
PIPE_CLK = {"pk": 3.26, "plain": 2.55, "mfma_f32_16x16x4": 32.0}
# packed ops per frame of the fused 512 kernel: (158 + 74) per lane pass (codelets_gen.hpp) x 16 lanes / 64
FUSED512_PK_PER_FRAME = (158 + 74) * 16 / 64.0


def alu_roofline(torch, dev, frames, kernel_ms, valu_per_frame, mfma_per_frame, mfma_clk_per_frame, src):
    """Every vector instruction priced at the pipe clocks alu_probe measured, the matrix instructions at their own busy
    cycles (SQ_VALU_MFMA_BUSY_CYCLES: 32 per fp32 16x16x4, 16 per bf16 16x16x32), all on ONE pipe per SIMD."""
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    plain = max(valu_per_frame - mfma_per_frame - FUSED512_PK_PER_FRAME, 0.0)
    clk = FUSED512_PK_PER_FRAME * PIPE_CLK["pk"] + plain * PIPE_CLK["plain"] + mfma_clk_per_frame
    ceiling = n_cu * 4 * 2.4e9 / clk
    rate = frames / (kernel_ms * 1e-3)
    return {"bound": "fp32 vector pipe (VALU and MFMA serialised)", "achieved": round(rate, 1), "peak": round(ceiling, 1),
            "unit": "frames/s per GPU", "frac": round(rate / ceiling, 4),
            "note": "peak = CUs x 4 SIMDs x 2.4 GHz / %.0f pipe clocks per frame (%.1f packed x %.2f + %.1f plain x %.2f + "
                    "%.1f MFMA busy clocks; pipe clocks measured by tools/alu_probe.hip, counts from %s)"
                    % (clk, FUSED512_PK_PER_FRAME, PIPE_CLK["pk"], plain, PIPE_CLK["plain"], mfma_clk_per_frame, src)}


def cpu_baseline(channels, gpu_rows, fixed, what):
    """The oracle timed on this host on a bounded sample (>= 10 s of CPU work or the whole sample list)."""
    import numpy as np
    from oracle import mfcc_fixed, mfcc_float
    tc, n_fr, n_done, ok = 0.0, 0, 0, True
    for chx, got in zip(channels, gpu_rows):
        if tc >= 10.0:
            break
        t1 = time.perf_counter()
        if fixed:
            ref = mfcc_fixed.mfcc_fixed_ref(chx, nceptrums=NCEP)
        else:
            ref = mfcc_float.mfcc_notebook(chx)[:, :NCEP]
        tc += time.perf_counter() - t1
        if fixed:
            ok = ok and bool(np.array_equal(ref, got))
        else:
            ok = ok and bool(np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max() < 1e-4)
        n_fr += len(ref)
        n_done += 1
    kind_note = ("NumPy restatement of the RTL arithmetic (oracle/mfcc_fixed.py), vectorised over frames" if fixed else
                 "restatement of notebook/MFCC.ipynb cells 7-39, float64, per-frame loops kept (oracle/mfcc_float.py), "
                 "single process")
    return {"value": round(n_fr / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%s: %d of them, %d frames in %.1f s; %s; host has %d logical CPUs; the GPU output of the sample "
                      "matches the CPU result: %s" % (what, n_done, n_fr, tc, kind_note, os.cpu_count() or 0, ok)}


# ------------------------------------------------------------------------------------------------ config 2

def bench_config2(cx, args):
    import mfcc_amd
    torch, dev, rank, world = cx.torch, cx.dev, cx.rank, cx.world
    nch = args.channels
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    pcm = torch.empty((nch, SAMPLES_PER_CH), dtype=torch.int16, device=dev)
    for c in range(nch):                 # white Gaussian, sigma 3000, generated on the device, HBM resident
        x = torch.randn(SAMPLES_PER_CH, generator=g, device=dev, dtype=torch.float32) * 3000.0
        pcm[c] = x.clamp_(-32768, 32767).to(torch.int16)
    del x
    pad = "stream" if args.fixed else "notebook"
    m = mfcc_amd.MFCC(nfft=NFFT, nfilters=NMEL, nceptrums=NCEP, pad_mode=pad, impl=args.impl, device=cx.local_rank)
    frames_per_ch = m.num_frames(SAMPLES_PER_CH)
    frames = frames_per_ch * nch
    out = torch.empty((nch, frames_per_ch, NCEP), device=dev, dtype=torch.int16 if args.fixed else torch.float32)
    run = (lambda: m.process_fixed(pcm, out=out)) if args.fixed else (lambda: m.process(pcm, out=out))

    cx.prewarm(run)
    dt = cx.timed(run, args.steps, args.warmup)
    # dominant kernel's launch duration, HIP events on the launch stream (rank-local)
    kernel_ms = m.time_launches(pcm, out, fixed=args.fixed, warmup=1, iters=max(3, min(args.steps, 10)))
    bpf = (HOP * 2 + NCEP * 2) if args.fixed else BYTES_PER_FRAME
    assert bool(torch.isfinite(out.float()).all()), "non-finite coefficients in bench output"

    pcie = None
    if rank == 0 and args.host_io:
        hp = pcm[:8].cpu().numpy()
        f = m.process_fixed if args.fixed else m.process
        f(hp)                                                  # warm the staging buffers
        t1 = time.perf_counter()
        for _ in range(3):
            f(hp)
        tp = (time.perf_counter() - t1) / 3
        pcie = {"value": round(8 * frames_per_ch / tp, 1), "unit": "frames/s",
                "what": "mfcc_hip_process_i16 on pageable host buffers, 8 channels: H2D + kernel + D2H"}

    line = None
    if rank == 0:
        kname = m.kernel_name(fixed=args.fixed)
        roof, valu, src = roofline(frames, bpf, kernel_ms, kname)
        line = {
            "metric": "MFCC frames/sec (512-pt, 32 mel, 13 coeff)",
            "value": round(frames * world * args.steps / dt, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32/int64 (RTL fixed-point)" if args.fixed else "f32",
            "data": "synthetic",
            "config": {
                "workload": ("synthetic 16 kHz mono, 10 min per channel, %d channels per GPU, nfft 512 / hop 170 / "
                             "32 mel / 13 coeff, %s" % (nch, "fixed-point int16 path (configs[2])" if args.fixed
                                                        else "float32 path (configs[1])")),
                "frames_per_step_per_gpu": frames,
                "bytes_per_frame": bpf,
                "parallelism": "frames sharded by channel across %d GPU(s), no data-path collective" % world,
                "kernel": kname,
                "prewarm_s": PREWARM_S,
            },
            "roofline": roof,
            "cpu_baseline": None,
        }
        if valu:
            line["valu_roofline"] = valu_roofline(torch, dev, frames, kernel_ms, valu, src, fixed=args.fixed)
            if PROFILE["mfma_per_frame"] and PROFILE["mfma_clk_per_frame"] and kname.startswith("mfcc_fused512"):
                line["alu_roofline"] = alu_roofline(torch, dev, frames, kernel_ms, valu, PROFILE["mfma_per_frame"],
                                                    PROFILE["mfma_clk_per_frame"], src)
        if pcie:
            line["pcie_inclusive"] = pcie
        if not args.no_cpu_baseline:
            n_s = min(nch, 48)
            line["cpu_baseline"] = cpu_baseline((pcm[c].cpu().numpy() for c in range(n_s)),
                                                (out[c].cpu().numpy() for c in range(n_s)), args.fixed,
                                                "whole 10-min channels of the batch")
    m.close()
    del pcm, out
    torch.cuda.empty_cache()
    return line


# ------------------------------------------------------------------------------------------------ config 5

def bench_config5(cx, args, steps, warmup, with_cpu):
    """ONE fixed corpus (seed = utterance id), sharded by utterance; one launch per rank per step."""
    import numpy as np
    import mfcc_amd
    from mfcc_amd import dist as md
    torch, dev, rank, world = cx.torch, cx.dev, cx.rank, cx.world
    n_utt, n = args.utterances, C5_SAMPLES
    mine = md.plan_items(n_utt, world)[rank]                       # contiguous range of utterance ids
    g = torch.Generator(device=dev)
    flat = torch.empty(max(len(mine), 1) * n, dtype=torch.int16, device=dev)
    for i, u in enumerate(mine):
        g.manual_seed(u)
        flat[i * n:(i + 1) * n] = (torch.randn(n, generator=g, device=dev) * 3000.0).clamp_(-32768, 32767).to(torch.int16)
    offsets = np.arange(len(mine) + 1, dtype=np.uint64) * n
    pad = "stream" if args.fixed else "notebook"
    m = mfcc_amd.MFCC(nfft=NFFT, nfilters=NMEL, nceptrums=NCEP, pad_mode=pad, impl=args.impl, device=cx.local_rank)
    per = m.num_frames(n)
    frames_local, frames_total = per * len(mine), per * n_utt
    odt = torch.int16 if args.fixed else torch.float32
    out = torch.empty((frames_local, NCEP), device=dev, dtype=odt)
    corpus = flat[:len(mine) * n]
    run = lambda: m.process_packed(corpus, offsets, fixed=args.fixed, out=out)

    cx.prewarm(run)
    dt = cx.timed(run, steps, warmup)
    kernel_ms = None
    if len(mine):
        kernel_ms = m.time_launches(flat[:len(mine) * n].view(len(mine), n), out.view(len(mine), per, NCEP),
                                    fixed=args.fixed, warmup=1, iters=max(3, min(steps, 10)))
    assert bool(torch.isfinite(out.float()).all()), "non-finite coefficients in bench output"

    # the one collective of the path, outside `value`: every rank's rows to rank 0 (RCCL over xGMI with nccl)
    gather = None
    if world > 1 and not args.no_gather:
        loc = out.float() if args.backend == "nccl" else out.float().cpu()
        md.gather_frames(loc, NCEP, dst=0)                         # warm-up (communicator setup)
        torch.cuda.synchronize()
        cx.barrier()
        t0 = time.perf_counter()
        full = md.gather_frames(loc, NCEP, dst=0)
        torch.cuda.synchronize()
        cx.barrier()
        tg = cx.max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            assert full.shape[0] == frames_total
            gather = {"ms": round(tg * 1e3, 3), "bytes": frames_total * NCEP * 4, "backend": args.backend,
                      "what": "gather_frames(dst=0): all ranks' [frames, 13] fp32 rows to rank 0, once, outside value"}

    res = None
    if rank == 0:
        bpf = (HOP * 2 + NCEP * 2) if args.fixed else BYTES_PER_FRAME
        kname = m.kernel_name(fixed=args.fixed)
        roof, valu, src = roofline(frames_local, bpf, kernel_ms, kname)
        res = {
            "metric": "MFCC frames/sec (512-pt, 32 mel, 13 coeff)",
            "value": round(frames_total * steps / dt, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32/int64 (RTL fixed-point)" if args.fixed else "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[4]: ONE corpus of %d utterances x 10 s (160 000 samples, seed = utterance id), "
                            "nfft 512 / hop 170 / 32 mel / 13 coeff, sharded by utterance over %d GPU(s), one launch "
                            "per rank per step" % (n_utt, world),
                "frames_per_step": frames_total,
                "frames_per_step_rank0": frames_local,
                "bytes_per_frame": bpf,
                "parallelism": "utterances %d..%d on rank 0 (plan_items), no data-path collective" % (mine.start, mine.stop - 1),
                "kernel": kname,
            },
            "roofline": roof,
            "gather": gather,
            "cpu_baseline": None,
        }
        if valu:
            res["valu_roofline"] = valu_roofline(torch, dev, frames_local, kernel_ms, valu, src, fixed=args.fixed)
        if with_cpu:
            k = min(len(mine), 256)
            rows = out.view(len(mine), per, NCEP)
            res["cpu_baseline"] = cpu_baseline((flat[i * n:(i + 1) * n].cpu().numpy() for i in range(k)),
                                               (rows[i].cpu().numpy() for i in range(k)), args.fixed,
                                               "10-s utterances of the corpus, ids %d.." % mine.start)
    m.close()
    del flat, out
    torch.cuda.empty_cache()
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    cx = Ctx(args)
    if args.config == 5:
        line = bench_config5(cx, args, args.steps, args.warmup, with_cpu=not args.no_cpu_baseline)
    else:
        line = bench_config2(cx, args)
        if not args.no_config5 and not args.host_io:
            c5 = bench_config5(cx, args, steps=max(5, min(args.steps, 20)), warmup=2, with_cpu=False)
            if cx.rank == 0:
                line["config5"] = {k: c5[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step", "scaling",
                                                      "config", "roofline", "gather")}
    if cx.rank == 0:
        print(json.dumps(line))
    cx.close()


if __name__ == "__main__":
    main()
