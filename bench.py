#!/usr/bin/env python3
"""bench.py -- MFCC frames/sec (512-pt, 32 mel, 13 coeff) on N MI355X; % of HBM roofline.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 16 kHz mono PCM, 10 min per channel
(9 600 000 samples -> 56 468 frames at 512/170), batched over 64 channels so one launch has
3.6 M frames (a single 19 MB channel cannot fill the chip).  float32 kernel, 13 coefficients.
A "step" = one pass of the hot path over the rank's batch, input and output resident in HBM.
Each rank owns its own batch (weak scaling, frames shard with no data-path collective --
SURVEY.md 8e); value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (algorithmic bytes 392 B/frame over the kernel's
HIP-event duration, against the 8 TB/s HBM peak) and `cpu_baseline` (the oracle's restatement
of the reference notebook's NumPy path timed on this host, one 10-min channel).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3       # vector fp32 peak, for the compute-side note
NFFT, HOP, NMEL, NCEP = 512, 170, 32, 13
SAMPLES_PER_CH = 9_600_000     # 10 min @ 16 kHz
BYTES_PER_FRAME = HOP * 2 + NCEP * 4      # 392 B: each sample read once, each output written once


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--channels", type=int, default=64, help="10-min channels per GPU per step")
    ap.add_argument("--impl", default="auto", choices=["auto", "generic", "fused512"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fixed", action="store_true", help="bench the fixed-point kernel instead (config 3)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                         "multi-rank path on a box with fewer GPUs than ranks: all ranks then share cuda:0)")
    ap.add_argument("--host-io", action="store_true",
                    help="also time the host-buffer entry point (H2D + kernel + D2H) on 8 channels; reported "
                         "as pcie_inclusive, never as value")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "gloo":
        local_rank = 0                       # rehearsal: every rank on the one GPU that is there
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import mfcc_amd

    # ---- synthetic input, generated on the device (white Gaussian, sigma 3000, int16), HBM resident
    nch = args.channels
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    pcm = torch.empty((nch, SAMPLES_PER_CH), dtype=torch.int16, device=dev)
    for c in range(nch):
        x = torch.randn(SAMPLES_PER_CH, generator=g, device=dev, dtype=torch.float32) * 3000.0
        pcm[c] = x.clamp_(-32768, 32767).to(torch.int16)
    del x

    pad = "stream" if args.fixed else "notebook"
    m = mfcc_amd.MFCC(nfft=NFFT, nfilters=NMEL, nceptrums=NCEP, pad_mode=pad, impl=args.impl,
                      device=local_rank)
    frames_per_ch = m.num_frames(SAMPLES_PER_CH)
    frames = frames_per_ch * nch
    out = torch.empty((nch, frames_per_ch, NCEP), device=dev,
                      dtype=torch.int16 if args.fixed else torch.float32)
    run = (lambda: m.process_fixed(pcm, out=out)) if args.fixed else (lambda: m.process(pcm, out=out))

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant kernel's launch duration, HIP events on the launch stream (rank-local)
    kernel_ms = m.time_launches(pcm, out, fixed=args.fixed, warmup=1, iters=max(3, min(args.steps, 10)))
    bytes_per_frame = (HOP * 2 + NCEP * 2) if args.fixed else BYTES_PER_FRAME
    achieved = frames * bytes_per_frame / (kernel_ms * 1e-3) / 1e9

    # HBM bytes per launch of this kernel, from the committed rocprofv3 PMC summary of the same
    # command (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes; profiles/summarize_rocprof.py)
    traffic, traffic_src, valu_per_frame = None, None, None
    try:
        import glob
        for pj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
            prof = json.load(open(pj))
            d = prof.get("derived", {})
            if prof.get("kernel", "").endswith(m.kernel_name(fixed=args.fixed)) and \
                    d.get("frames_per_launch") == frames and "hbm_traffic_bytes_per_launch" in d:
                traffic = round(d["hbm_traffic_bytes_per_launch"])
                traffic_src = os.path.relpath(pj, ROOT)
                valu_per_frame = d.get("valu_wave_instructions_per_frame")
                break
    except Exception:
        traffic = None

    # quick sanity on the timed output (not a parity test: tests/ do that)
    assert bool(torch.isfinite(out.float()).all()), "non-finite coefficients in bench output"

    pcie = None
    if rank == 0 and args.host_io:
        hp = pcm[:8].cpu().numpy()
        m.use_own_stream()
        m.process_fixed(hp) if args.fixed else m.process(hp)          # warm the staging buffers
        t1 = time.perf_counter()
        for _ in range(3):
            m.process_fixed(hp) if args.fixed else m.process(hp)
        tp = (time.perf_counter() - t1) / 3
        pcie = {"value": round(8 * frames_per_ch / tp, 1), "unit": "frames/s",
                "what": "mfcc_hip_process_i16 on pageable host buffers, 8 channels: H2D + kernel + D2H"}

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import mfcc_fixed, mfcc_float
        # bounded sample: whole 10-min channels of the same batch until >= 10 s of CPU work
        tc, n_done, n_fr, ok = 0.0, 0, 0, True
        while tc < 10.0 and n_done < min(nch, 48):
            chx = pcm[n_done].cpu().numpy()
            t1 = time.perf_counter()
            if args.fixed:
                ref = mfcc_fixed.mfcc_fixed_ref(chx, nceptrums=NCEP)
            else:
                ref = mfcc_float.mfcc_notebook(chx)[:, :NCEP]
            tc += time.perf_counter() - t1
            got = out[n_done].cpu().numpy()
            if args.fixed:
                ok = ok and bool(np.array_equal(ref, got))
            else:
                ok = ok and bool(np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max() < 1e-4)
            n_fr += len(ref)
            n_done += 1
        kind_note = ("NumPy restatement of the RTL arithmetic (oracle/mfcc_fixed.py), vectorised over frames"
                     if args.fixed else
                     "restatement of notebook/MFCC.ipynb cells 7-39, float64, per-frame loops kept "
                     "(oracle/mfcc_float.py), single process")
        cpu = {"value": round(n_fr / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "channels 0..%d of the batch (10 min synthetic PCM each): %d frames in %.1f s; %s; host "
                         "has %d logical CPUs; the GPU output of those channels matches the CPU result: %s"
                         % (n_done - 1, n_fr, tc, kind_note, os.cpu_count() or 0, ok)}

    if rank == 0:
        total_frames = frames * world
        value = total_frames * args.steps / dt
        line = {
            "metric": "MFCC frames/sec (512-pt, 32 mel, 13 coeff)",
            "value": round(value, 1),
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
                "bytes_per_frame": bytes_per_frame,
                "parallelism": "frames sharded by channel across %d GPU(s), no data-path collective" % world,
                "kernel": m.kernel_name(fixed=args.fixed),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_unit": "bytes per launch (algorithmic: %d)" % (frames * bytes_per_frame),
                "traffic_source": traffic_src,
                "kernel_ms": round(kernel_ms, 4),
                "note": "algorithmic bytes = frames x %d B / HIP-event kernel time; the path is fp32-VALU/LDS "
                        "bound (DESIGN.md), so this fraction is reported as asked, not as the binding limit"
                        % bytes_per_frame,
            },
            "cpu_baseline": cpu,
        }
        if valu_per_frame:
            # the binding limit: every VALU wave-instruction holds a SIMD for >= 4 clocks
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            ceiling = n_cu * 4 * 2.4e9 / (valu_per_frame * 4.0)
            rate = frames / (kernel_ms * 1e-3)
            line["valu_roofline"] = {
                "bound": "valu", "achieved": round(rate, 1), "peak": round(ceiling, 1), "unit": "frames/s per GPU",
                "frac": round(rate / ceiling, 4),
                "note": "peak = CUs x 4 SIMDs x 2.4 GHz / (%.1f VALU wave-instructions per frame x 4 clocks), "
                        "instruction count from %s" % (valu_per_frame, traffic_src)}
        if pcie:
            line["pcie_inclusive"] = pcie
        print(json.dumps(line))
    m.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
