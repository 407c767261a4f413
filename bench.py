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

The default line also carries, as sub-objects measured the same way (barrier + synchronize around K steps, HIP events
for the kernel, skip each with --no-configN):
  `config3`  BASELINE.json configs[2]: the fixed-point kernel (RTL arithmetic, STREAM framing, 366 B/frame) on the same batch;
  `config4`  configs[3]: 64 channels x 1 h, nfft 1024 / hop 341 / 40 mel / 13 coeff (10.8 M frames, 734 B/frame);
  `config5`  configs[4]: ONE fixed corpus of 10 000 utterances x 10 s (seed = utterance id), sharded by utterance over the
             N ranks (mfcc_amd.dist.plan_items), one launch per rank per step: strong scaling.  The optional result
             gather (13 floats per frame, RCCL over xGMI through torch.distributed) is timed separately and reported
             as its own field, never inside `value`.
`--config 3|4|5` runs one of them as the headline instead.

Prints ONE JSON line (rank 0) with `roofline` (algorithmic bytes per frame over the kernel's HIP-event duration,
against the 8 TB/s HBM peak) and `cpu_baseline` (the oracle's restatement of the reference notebook's NumPy path
timed on this host, on a bounded sample of the same workload and on the reference's own wav, config 1's input).
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
SPEC_CLOCK_GHZ = 2.4           # MI355X_MICROARCH.md: max clock; the clock held under load comes from the PMC summary
C5_UTTS, C5_SAMPLES = 10_000, 160_000     # config 5: 10 000 utterances of 10 s
PREWARM_S = 0.4                # launches before anything is timed, whatever --warmup says: the clocks of an idle
                               # MI355X ramp over the first ~100 ms of work (round 1: 5 warm-up steps read 5 % slow)

# the batched configurations (BASELINE.json configs[1..3]); bytes per frame = hop x 2 B in + n_cep x 4 | 2 B out
BATCH = {
    2: dict(nfft=512, hop=170, nmel=32, ncep=13, fixed=False, pad="notebook", samples=9_600_000, power_scale=512.0,
            what="float32 path (configs[1])", per_ch="10 min"),
    3: dict(nfft=512, hop=170, nmel=32, ncep=13, fixed=True, pad="stream", samples=9_600_000, power_scale=512.0,
            what="fixed-point int16 path, RTL arithmetic, STREAM framing (configs[2])", per_ch="10 min"),
    4: dict(nfft=1024, hop=341, nmel=40, ncep=13, fixed=False, pad="notebook", samples=57_600_000, power_scale=0.0,
            what="float32 path, power scale 1/nfft, mel contraction on the matrix cores (configs[3])", per_ch="1 h"),
}
NFFT, HOP, NMEL, NCEP = 512, 170, 32, 13                  # config 5's parameters (= config 2's)


def bytes_per_frame(c):
    return c["hop"] * 2 + c["ncep"] * (2 if c["fixed"] else 4)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="2: configs[1], per-rank 64-channel batch, weak scaling (default, carries 3, 4 and 5 as "
                         "sub-objects); 3: the fixed-point kernel on that batch; 4: 64 channels x 1 h at 1024/341/40; "
                         "5: configs[4], one fixed 10k-utterance corpus sharded by utterance, strong scaling")
    ap.add_argument("--channels", type=int, default=64, help="configs 2-4: channels per GPU per step")
    ap.add_argument("--utterances", type=int, default=C5_UTTS, help="config 5: corpus size")
    ap.add_argument("--impl", default="auto", choices=["auto", "generic", "fused512"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="default run: leave out the config3 sub-object")
    ap.add_argument("--no-config4", action="store_true", help="default run: leave out the config4 sub-object")
    ap.add_argument("--no-config5", action="store_true", help="default run: leave out the config5 sub-object")
    ap.add_argument("--only", action="store_true", help="default run: no sub-objects at all (profiling passes)")
    ap.add_argument("--no-gather", action="store_true", help="config 5: do not time the result gather")
    ap.add_argument("--fixed", action="store_true", help="same as --config 3 (config 5: the fixed-point kernel)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                         "multi-rank path on a box with fewer GPUs than ranks: all ranks then share cuda:0)")
    ap.add_argument("--group", action="store_true",
                    help="create the process group even with one rank (world size 1 on RCCL: the only exercise of "
                         "the collective path one GPU allows)")
    ap.add_argument("--host-io", action="store_true",
                    help="also time the host-buffer entry point (H2D + kernel + D2H) on 8 channels; reported "
                         "as pcie_inclusive, never as value")
    a = ap.parse_args(argv)
    if a.fixed and a.config == 2:
        a.config = 3
    if a.config == 5 and a.utterances < a.gpus:
        ap.error("--utterances %d < --gpus %d: every rank needs at least one utterance" % (a.utterances, a.gpus))
    return a


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
        self.grouped = self.world > 1 or args.group
        if self.grouped:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1:
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group("gloo")

    def barrier(self):
        if self.grouped:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if not self.grouped:
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

    def enqueue_us(self, run, steps):
        """Host time of one step with nothing waited for: what the CPU needs to put a step into the stream.  If it
        exceeds the step's kernel time the rank is host-bound (the 8-GPU shard of config 5 runs 0.3 ms kernels)."""
        torch = self.torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (t1 - t0) / steps * 1e6

    def prewarm(self, run):
        torch = self.torch
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < PREWARM_S:
            for _ in range(8):
                run()
            torch.cuda.synchronize()

    def close(self):
        if self.grouped:
            self.dist.destroy_process_group()


def _probe_rows(name):
    """rows of a committed tools/*_probe output: label -> (A, B) cycles per instruction of the two co-resident waves"""
    import glob
    rows = {}
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s.txt" % name)))[-1:]:
        for ln in open(p):
            m = re.match(r"^(.*?)\s+(?:alone/)?A\s+([\d.]+)(?: cyc/instr)?\s+B\s+([\d.]+)", ln)
            if m:
                rows[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)))
        rows["__file__"] = os.path.relpath(p, ROOT)
    return rows


def _pipe(rows, label, default):
    """pipe clocks per instruction when two waves of a SIMD issue it against each other: 1 / (1/A + 1/B)"""
    ab = rows.get(label)
    if not ab or not ab[0] or not ab[1]:
        return default
    return round(1.0 / (1.0 / ab[0] + 1.0 / ab[1]), 2)


def pipe_clocks():
    """Measured on MI355X with tools/alu_probe.hip / tools/int_probe.hip (profiles/r*_alu_probe.txt, r*_int_probe.txt), two
    waves per SIMD: a packed fp32 op (v_pk_fma/add/mul_f32) or v_dot2 holds the SIMD's vector pipe ~3.2 clocks, a plain
    fp32 op ~2.6, v_mfma_f32_16x16x4_f32 32 -- during which NO vector instruction of either wave issues (SQ_VALU_MFMA_
    COEXEC_CYCLES reads 0): on gfx950 the fp32 matrix instruction is vector-pipe time.  Integer: add / shift ~2.4, dot2 /
    SDWA / mul24 / bfe / perm ~3.2.  Read from the committed probe outputs; the literals are those files' values and
    only stand in when a file is missing."""
    a, i = _probe_rows("alu_probe"), _probe_rows("int_probe")
    pk = _pipe(a, "pk_fma || pk_fma", 3.26)
    plain = _pipe(a, "fma32 || fma32", 2.55)
    iplain = _pipe(i, "v_add_u32 || same", 2.4)
    iother = _pipe(i, "v_dot2_i32_i16 +s || same", 3.2)
    return {"pk": pk, "plain": plain, "int_plain": iplain, "int_other": iother,
            # fixed-point kernel: a third of its vector instructions (static count over the loop body) are plain add /
            # shift / and ops, the rest dot2, SDWA, mul24, bfe, perm, v_mad_u64_u32 ...
            "fixed_mix": round(0.33 * iplain + 0.67 * iother, 2),
            "source": "%s, %s" % (a.get("__file__", "literals"), i.get("__file__", "literals"))}


# packed fp32 ops per frame (codelets_gen.hpp op counts x lanes per frame / 64): the 512 kernel runs rfft32_tw (158)
# and cfft16_pow (90: the transform and |X|^2) on 16 lanes per frame each; the 1024 kernel rfft32_tw on 32 lanes,
# cfft32_h0_pow (106) and _h1_pow (135) on 16 each
PK_PER_FRAME = {"mfcc_fused512": (158 + 90) * 16 / 64.0, "mfcc_fused1024": (158 * 32 + (106 + 135) * 16) / 64.0}


def profile_figures(kernel_name, frames):
    """Per-launch HBM bytes and per-frame instruction counts of `kernel_name` from the newest committed rocprofv3 PMC
    summary (profiles/summarize_rocprof.py; FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes) whose kernel-source
    stamp matches the kernels this run executes; a summary of other sources is dropped and named."""
    import glob
    import mfcc_amd
    want = mfcc_amd.kernel_source_hash()
    fig = {"traffic": None, "src": None, "valu": None, "mfma": None, "mfma_clk": None, "clock_ghz": None, "stale": None}
    for pj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            prof = json.load(open(pj))
        except Exception:
            continue
        d = prof.get("derived", {})
        if not prof.get("kernel", "").endswith(kernel_name) or "hbm_traffic_bytes_per_launch" not in d:
            continue
        if prof.get("kernel_source_hash") != want:
            fig["stale"] = fig["stale"] or os.path.relpath(pj, ROOT)
            continue
        n = d["frames_per_launch"]
        cnt = prof.get("counters", {})
        mf = cnt.get("SQ_INSTS_MFMA", {}).get("avg_per_launch") or cnt.get("SQ_INSTS_VALU_MFMA_F32", {}).get("avg_per_launch")
        mc = cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("avg_per_launch")
        fig.update(traffic=round(d["hbm_traffic_bytes_per_launch"] / n * frames), src=os.path.relpath(pj, ROOT),
                   valu=d.get("valu_wave_instructions_per_frame"), mfma=mf / n if mf else None,
                   mfma_clk=mc / n if mc else None, clock_ghz=d.get("shader_clock_ghz"), stale=None)
        return fig
    return fig


def roofline(frames, bpf, kernel_ms, kernel_name):
    achieved = frames * bpf / (kernel_ms * 1e-3) / 1e9
    fig = profile_figures(kernel_name, frames)
    r = {
        "bound": "hbm",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": fig["traffic"],
        "traffic_unit": "bytes per launch (algorithmic: %d)" % (frames * bpf),
        "traffic_source": fig["src"],
        "kernel_ms": round(kernel_ms, 4),
        "note": "algorithmic bytes = frames x %d B / HIP-event kernel time; the path is %s bound "
                "(DESIGN.md), so this fraction is reported as asked, not as the binding limit"
                % (bpf, "integer-VALU (a 16-bit datapath emulated bit for bit)" if "fixed" in kernel_name
                   else "fp32-VALU/LDS"),
    }
    if fig["traffic"] is None and fig["stale"]:
        r["traffic_dropped"] = "%s was collected on other kernel sources (stamp mismatch)" % fig["stale"]
    return r, fig


def compute_ceilings(torch, dev, frames, kernel_ms, kernel_name, fig, fixed):
    """The binding limits: `valu_roofline` = every VALU wave-instruction holds a SIMD for 4 clocks (round 1's
    definition; the integer kernel at its measured mix) and, for the float kernels, `alu_roofline` = every vector
    instruction at the pipe clocks tools/alu_probe.hip measured plus the matrix instructions' own busy cycles, all on ONE
    pipe per SIMD.  Clock: what the chip held under this kernel (GRBM_GUI_ACTIVE / 8 / kernel time, from the stamped
    PMC summary) when the summary has it, else the 2.4 GHz spec."""
    if not fig["valu"]:
        return {}
    pc = pipe_clocks()
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    ghz = fig["clock_ghz"] or SPEC_CLOCK_GHZ
    clk_note = ("%.2f GHz held under load (GRBM_GUI_ACTIVE / 8 / kernel time, %s)" % (ghz, fig["src"])
                if fig["clock_ghz"] else "%.1f GHz spec clock" % ghz)
    rate = frames / (kernel_ms * 1e-3)
    out = {}
    clk = pc["fixed_mix"] if fixed else 4.0
    ceiling = n_cu * 4 * ghz * 1e9 / (fig["valu"] * clk)
    out["valu_roofline"] = {
        "bound": "valu", "achieved": round(rate, 1), "peak": round(ceiling, 1), "unit": "frames/s per GPU",
        "frac": round(rate / ceiling, 4), "clock_ghz": ghz,
        "note": "peak = CUs x 4 SIMDs x %s / (%.1f VALU wave-instructions per frame x %s clocks%s), instruction count "
                "from %s" % (clk_note, fig["valu"], clk, " of pipe: a third plain ops at %.2f, the rest at %.2f, %s"
                             % (pc["int_plain"], pc["int_other"], pc["source"]) if fixed else "", fig["src"])}
    pk = next((v for k, v in PK_PER_FRAME.items() if k in kernel_name), None)
    if not fixed and pk and fig["mfma"] and fig["mfma_clk"]:
        plain = max(fig["valu"] - fig["mfma"] - pk, 0.0)
        clk = pk * pc["pk"] + plain * pc["plain"] + fig["mfma_clk"]
        ceiling = n_cu * 4 * ghz * 1e9 / clk
        out["alu_roofline"] = {
            "bound": "fp32 vector pipe (VALU and MFMA serialised)", "achieved": round(rate, 1),
            "peak": round(ceiling, 1), "unit": "frames/s per GPU", "frac": round(rate / ceiling, 4), "clock_ghz": ghz,
            "note": "peak = CUs x 4 SIMDs x %s / %.0f pipe clocks per frame (%.1f packed x %.2f + %.1f plain x %.2f + %.1f "
                    "MFMA busy clocks; pipe clocks from %s, counts from %s)"
                    % (clk_note, clk, pk, pc["pk"], plain, pc["plain"], fig["mfma_clk"], pc["source"], fig["src"])}
    return out


def cpu_baseline(channels, gpu_rows, cfg, what, budget_s=10.0):
    """The oracle timed on this host on a bounded sample (>= budget_s of CPU work or the whole sample list)."""
    import numpy as np
    from oracle import mfcc_fixed, mfcc_float
    tc, n_fr, n_done, ok = 0.0, 0, 0, True
    for chx, got in zip(channels, gpu_rows):
        if tc >= budget_s:
            break
        t1 = time.perf_counter()
        if cfg["fixed"]:
            ref = mfcc_fixed.mfcc_fixed_ref(chx, nceptrums=cfg["ncep"])
        elif cfg["nfft"] == 512:
            ref = mfcc_float.mfcc_notebook(chx)[:, :cfg["ncep"]]
        else:
            ref = mfcc_float.mfcc_float_ref(chx, nfft=cfg["nfft"], hop=cfg["hop"], n_mel=cfg["nmel"],
                                            power_scale=float(cfg["nfft"]))[:, :cfg["ncep"]]
        tc += time.perf_counter() - t1
        if got is not None:
            if cfg["fixed"]:
                ok = ok and bool(np.array_equal(ref, got))
            else:
                ok = ok and bool(np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max() < 1e-4)
        n_fr += len(ref)
        n_done += 1
    kind_note = ("NumPy restatement of the RTL arithmetic (oracle/mfcc_fixed.py), vectorised over frames" if cfg["fixed"]
                 else "restatement of notebook/MFCC.ipynb cells 7-39, float64, per-frame loops kept (oracle/mfcc_float.py), "
                 "single process")
    return {"value": round(n_fr / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%s: %d of them, %d frames in %.1f s; %s; host has %d logical CPUs; the GPU output of the sample "
                      "matches the CPU result: %s" % (what, n_done, n_fr, tc, kind_note, os.cpu_count() or 0, ok)}


def golden_wav():
    """config 1's input: the reference's own wav (tests/golden/, a fixture -- /root/reference does not exist on the box)"""
    import numpy as np
    p = os.path.join(ROOT, "tests", "golden", "f2bjrop1.0.wav")
    if not os.path.exists(p):
        return None
    raw = open(p, "rb").read()
    i = raw.index(b"data")
    n = int.from_bytes(raw[i + 4:i + 8], "little")
    return np.frombuffer(raw[i + 8:i + 8 + n], dtype="<i2").copy()


def noise_batch(cx, nch, n, seed):
    """white Gaussian, sigma 3000, generated on the device, HBM resident"""
    torch, dev = cx.torch, cx.dev
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pcm = torch.empty((nch, n), dtype=torch.int16, device=dev)
    for c in range(nch):
        x = torch.randn(n, generator=g, device=dev, dtype=torch.float32) * 3000.0
        pcm[c] = x.clamp_(-32768, 32767).to(torch.int16)
    del x
    return pcm


# ------------------------------------------------------------------------------------------------ configs 2, 3, 4

def bench_batch(cx, args, cid, steps, warmup, pcm=None, with_cpu=False, headline=False):
    """One batched configuration: every rank owns `--channels` channels (weak scaling), one launch per step."""
    import mfcc_amd
    torch, dev, rank, world = cx.torch, cx.dev, cx.rank, cx.world
    c = BATCH[cid]
    nch, n = args.channels, c["samples"]
    own = pcm is None
    if own:
        pcm = noise_batch(cx, nch, n, 1234 + rank + 1000 * (cid == 4))
    m = mfcc_amd.MFCC(nfft=c["nfft"], nfilters=c["nmel"], nceptrums=c["ncep"], pad_mode=c["pad"],
                      power_scale=c["power_scale"], impl=args.impl, device=cx.local_rank)
    fixed = c["fixed"]
    frames_per_ch = m.num_frames(n)
    frames = frames_per_ch * nch
    out = torch.empty((nch, frames_per_ch, c["ncep"]), device=dev, dtype=torch.int16 if fixed else torch.float32)
    run = (lambda: m.process_fixed(pcm, out=out)) if fixed else (lambda: m.process(pcm, out=out))

    cx.prewarm(run)
    dt = cx.timed(run, steps, warmup)
    # dominant kernel's launch duration, HIP events on the launch stream (rank-local)
    kernel_ms = m.time_launches(pcm, out, fixed=fixed, warmup=1, iters=max(3, min(steps, 10)))
    bpf = bytes_per_frame(c)
    assert bool(torch.isfinite(out.float()).all()), "non-finite coefficients in bench output"

    pcie = None
    if rank == 0 and args.host_io and headline:
        hp = pcm[:8].cpu().numpy()
        f = m.process_fixed if fixed else m.process
        for _ in range(3):
            f(hp)                                              # warm: the device staging buffers, and the first pinning
                                                               # of these pages is several times slower than the later ones
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            f(hp)
            ts.append(time.perf_counter() - t1)
        tp = sorted(ts)[len(ts) // 2]
        pcie = {"value": round(8 * frames_per_ch / tp, 1), "unit": "frames/s",
                "input_gbs": round(hp.nbytes / tp / 1e9, 1),
                "what": "mfcc_hip_process_i16 on pageable host buffers, 8 channels (154 MB), median of 5 calls: 64-MB chunks "
                        "pinned in place, H2D / kernel / D2H of neighbouring chunks overlapped"}

    line = None
    if rank == 0:
        kname = m.kernel_name(fixed=fixed)
        roof, fig = roofline(frames, bpf, kernel_ms, kname)
        line = {
            "metric": "MFCC frames/sec (%d-pt, %d mel, %d coeff)" % (c["nfft"], c["nmel"], c["ncep"]),
            "value": round(frames * world * steps / dt, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32/int64 (RTL fixed-point)" if fixed else "f32",
            "data": "synthetic",
            "config": {
                "workload": ("synthetic 16 kHz mono, %s per channel, %d channels per GPU, nfft %d / hop %d / "
                             "%d mel / %d coeff, %s" % (c["per_ch"], nch, c["nfft"], c["hop"], c["nmel"], c["ncep"],
                                                        c["what"])),
                "frames_per_step_per_gpu": frames,
                "bytes_per_frame": bpf,
                "parallelism": "frames sharded by channel across %d GPU(s), no data-path collective" % world,
                "kernel": kname,
                "fallback": m.is_fallback(fixed=fixed),      # a generic kernel runs: 3-6 x slower
                "prewarm_s": PREWARM_S,
            },
            "roofline": roof,
            "cpu_baseline": None,
        }
        line.update(compute_ceilings(torch, dev, frames, kernel_ms, kname, fig, fixed))
        if pcie:
            line["pcie_inclusive"] = pcie
        if with_cpu:
            n_s = min(nch, 48)
            # a 1-h channel of config 4 is 169 k frames (~10 s of CPU work): the sample is the first 10 min of channels
            cut = n if fixed else min(n, 9_600_000)
            nf_cut = m.num_frames(cut)
            line["cpu_baseline"] = cpu_baseline(
                (pcm[ch, :cut].cpu().numpy() for ch in range(n_s)),
                (out[ch, :nf_cut].cpu().numpy() for ch in range(n_s)),     # NOTEBOOK framing: a prefix's frames are a prefix
                c, "the first 10 min of channels of the batch" if cut < n else "whole 10-min channels of the batch")
            if cid in (2, 3):
                wav = golden_wav()
                if wav is not None:
                    got = (m.process_fixed if fixed else m.process)(wav)
                    line["cpu_baseline"]["config1"] = cpu_baseline(
                        [wav] * 400, [got] + [None] * 399, c,
                        "config 1's input, f2bjrop1.0.wav (178 240 samples), converted over and over", budget_s=2.0)
    m.close()
    del out
    if own:
        del pcm
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
    fixed = args.fixed
    mine = md.plan_items(n_utt, world)[rank]                       # contiguous range of utterance ids
    g = torch.Generator(device=dev)
    flat = torch.empty(max(len(mine), 1) * n, dtype=torch.int16, device=dev)
    for i, u in enumerate(mine):
        g.manual_seed(u)
        flat[i * n:(i + 1) * n] = (torch.randn(n, generator=g, device=dev) * 3000.0).clamp_(-32768, 32767).to(torch.int16)
    offsets = np.arange(len(mine) + 1, dtype=np.uint64) * n
    pad = "stream" if fixed else "notebook"
    m = mfcc_amd.MFCC(nfft=NFFT, nfilters=NMEL, nceptrums=NCEP, pad_mode=pad, impl=args.impl, device=cx.local_rank)
    per = m.num_frames(n)
    frames_local, frames_total = per * len(mine), per * n_utt
    odt = torch.int16 if fixed else torch.float32
    out = torch.empty((frames_local, NCEP), device=dev, dtype=odt)
    corpus = flat[:len(mine) * n]
    run = lambda: m.process_packed(corpus, offsets, fixed=fixed, out=out)

    cx.prewarm(run)
    dt = cx.timed(run, steps, warmup)
    kernel_ms = m.time_launches(flat[:len(mine) * n].view(len(mine), n), out.view(len(mine), per, NCEP),
                                fixed=fixed, warmup=1, iters=max(3, min(steps, 10)))
    assert bool(torch.isfinite(out.float()).all()), "non-finite coefficients in bench output"

    # what the host needs per step, against the kernel it feeds: this rank's shard, and -- on one GPU -- the shard an
    # 8-GPU run would give every rank (n_utt / 8 utterances: a kernel of ~0.3 ms), so that a host-bound step shows
    # before there is an 8-GPU node to show it on
    enq = {"this_rank": {"utterances": len(mine), "enqueue_us_per_step": round(cx.enqueue_us(run, 40), 1),
                         "kernel_us": round(kernel_ms * 1e3, 1)}}
    if world == 1 and n_utt >= 8:
        k8 = n_utt // 8
        c8, o8, out8 = flat[:k8 * n], offsets[:k8 + 1], out[:k8 * per]
        run8 = lambda: m.process_packed(c8, o8, fixed=fixed, out=out8)
        t8 = cx.timed(run8, 40, 5)
        ms8 = m.time_launches(c8.view(k8, n), out8.view(k8, per, NCEP), fixed=fixed, warmup=1, iters=10)
        enq["shard_of_8"] = {"utterances": k8, "enqueue_us_per_step": round(cx.enqueue_us(run8, 40), 1),
                             "kernel_us": round(ms8 * 1e3, 1), "step_us": round(t8 / 40 * 1e6, 1)}
    for v in enq.values():
        v["host_bound"] = v["enqueue_us_per_step"] > v["kernel_us"]

    # the one collective of the path, outside `value`: every rank's rows to rank 0 (RCCL over xGMI with nccl)
    gather = None
    if cx.grouped and not args.no_gather:
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
        bpf = HOP * 2 + NCEP * (2 if fixed else 4)
        kname = m.kernel_name(fixed=fixed)
        roof, fig = roofline(frames_local, bpf, kernel_ms, kname)
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
            "dtype": "int32/int64 (RTL fixed-point)" if fixed else "f32",
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
                "fallback": m.is_fallback(fixed=fixed),
            },
            "roofline": roof,
            "host_enqueue": enq,
            "gather": gather,
            "cpu_baseline": None,
        }
        res.update(compute_ceilings(torch, dev, frames_local, kernel_ms, kname, fig, fixed))
        if with_cpu:
            k = min(len(mine), 256)
            rows = out.view(len(mine), per, NCEP)
            res["cpu_baseline"] = cpu_baseline((flat[i * n:(i + 1) * n].cpu().numpy() for i in range(k)),
                                               (rows[i].cpu().numpy() for i in range(k)), BATCH[3 if fixed else 2],
                                               "10-s utterances of the corpus, ids %d.." % mine.start)
    m.close()
    del flat, out
    torch.cuda.empty_cache()
    return res


SUB_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "dtype", "config", "roofline",
            "valu_roofline", "alu_roofline", "host_enqueue", "gather")


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    cx = Ctx(args)
    with_cpu = not args.no_cpu_baseline
    if args.config == 5:
        line = bench_config5(cx, args, args.steps, args.warmup, with_cpu=with_cpu)
    elif args.config in (3, 4):
        line = bench_batch(cx, args, args.config, args.steps, args.warmup, with_cpu=with_cpu, headline=True)
    else:
        sub_steps = max(5, min(args.steps, 20))
        pcm = noise_batch(cx, args.channels, BATCH[2]["samples"], 1234 + cx.rank)
        line = bench_batch(cx, args, 2, args.steps, args.warmup, pcm=pcm, with_cpu=with_cpu, headline=True)
        subs = {}
        if not (args.only or args.host_io):
            if not args.no_config3:                   # the same batch through the fixed-point kernel
                subs["config3"] = bench_batch(cx, args, 3, sub_steps, 2, pcm=pcm)
            del pcm
            cx.torch.cuda.empty_cache()
            if not args.no_config4:
                subs["config4"] = bench_batch(cx, args, 4, sub_steps, 3)
            if not args.no_config5:
                subs["config5"] = bench_config5(cx, args, steps=sub_steps, warmup=2, with_cpu=False)
        if cx.rank == 0:
            for k, v in subs.items():
                line[k] = {kk: v[kk] for kk in SUB_KEYS if kk in v}
    if cx.rank == 0:
        print(json.dumps(line))
    cx.close()


if __name__ == "__main__":
    main()
