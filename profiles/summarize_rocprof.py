#!/usr/bin/env python3
"""Condense rocprofv3 output directories (scratch, under gpurun_out/) into the small, tracked
summaries kept in profiles/.

    python profiles/summarize_rocprof.py gpurun_out/r1 profiles/r01

reads   <src>/kt/**/_kernel_stats.csv          (rocprofv3 --kernel-trace --stats)
        <src>/pmc_*/**/_counter_collection.csv (rocprofv3 --pmc ..., one pass per directory)
        <src>/bench.json                        (the bench line of the same build)
writes  <dst>_kernel_stats.csv   rows of this repo's kernels only (torch's RNG kernels dropped)
        <dst>_pmc.json           per-counter averages per launch of the dominant kernel + derived numbers

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE counts 128-byte requests as
64 bytes for wide (16 B/lane) coalesced streaming reads -- which is what the kernel's sample-window
loads are -- so the read side is doubled.  WRITE_SIZE is exact for 16-B-per-lane stores; the
kernel's stores are 4 B per lane, an access width the guide marks as uncalibrated.
"""
import collections
import csv
import glob
import json
import os
import sys


def main(src, dst):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mfcc_amd._lib import kernel_source_hash
    out = {"kernel_source_hash": kernel_source_hash()}     # bench.py drops counter figures of other kernel sources
    ks = glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True)
    rows = []
    if ks:
        ks.sort(key=os.path.getmtime)                      # gpurun merges additively: newest run wins
        for r in csv.DictReader(open(ks[-1])):
            if "mfcc" in r["Name"]:
                rows.append(r)
        with open(dst + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        import re
        # "void ns::kernel<false>(args...)" -> "ns::kernel": bench.py matches it against mfcc_hip_kernel_name()
        out["kernel"] = re.sub(r"<[^<>]*>", "", dom["Name"].split("(")[0]).replace("void ", "").strip()
        out["kernel_trace"] = {"calls": int(dom["Calls"]), "avg_ns": float(dom["AverageNs"]),
                               "min_ns": float(dom["MinNs"]), "max_ns": float(dom["MaxNs"])}
    counters = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        if not fs:
            continue
        fs.sort(key=os.path.getmtime)
        agg = collections.defaultdict(list)
        per_dispatch = collections.defaultdict(float)      # a counter may come as one row per XCD / SE: sum them per dispatch
        meta = {}
        for r in csv.DictReader(open(fs[-1])):
            if "fused512" in r["Kernel_Name"] or "mfcc_float_generic" in r["Kernel_Name"] or \
                    "mfcc_fixed" in r["Kernel_Name"] or "fused1024" in r["Kernel_Name"]:
                per_dispatch[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
                meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                           "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
        for (name, _disp), v in per_dispatch.items():
            agg[name].append(v)
        for k, v in agg.items():
            counters[k] = {"avg_per_launch": sum(v) / len(v), "launches": len(v), "pass": os.path.basename(d)}
        if meta:
            out["dispatch"] = meta
    out["counters"] = counters
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj):
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        out["bench"] = {k: b[k] for k in ("value", "unit", "ms_per_step", "roofline", "cpu_baseline", "config")}
        frames = b["config"]["frames_per_step_per_gpu"]
        bpf = b["config"]["bytes_per_frame"]
        d = {"frames_per_launch": frames, "algorithmic_bytes_per_launch": frames * bpf}
        if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
            rd = 2.0 * counters["FETCH_SIZE"]["avg_per_launch"] * 1024.0
            wr = counters["WRITE_SIZE"]["avg_per_launch"] * 1024.0
            d.update({"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                      "hbm_traffic_bytes_per_launch": rd + wr,
                      "traffic_over_algorithmic": (rd + wr) / (frames * bpf)})
        if "SQ_INSTS_VALU" in counters:
            d["valu_wave_instructions_per_frame"] = counters["SQ_INSTS_VALU"]["avg_per_launch"] / frames
            d["valu_lane_ops_per_frame"] = 64.0 * counters["SQ_INSTS_VALU"]["avg_per_launch"] / frames
        if "GRBM_GUI_ACTIVE" in counters and "kernel_trace" in out:
            # MI355X_MICROARCH.md, DVFS give-back: effective clock = GRBM_GUI_ACTIVE / 8 / kernel wall time (rocprofv3 reports
            # the sum over the 8 XCDs; reads high on dispatches shorter than ~0.3 ms).  The wall time is the kernel trace's
            # average of the same bench command.
            d["shader_clock_ghz"] = round(counters["GRBM_GUI_ACTIVE"]["avg_per_launch"] / 8.0 / out["kernel_trace"]["avg_ns"], 3)
        if "SQ_WAVE_CYCLES" in counters:
            wc = counters["SQ_WAVE_CYCLES"]["avg_per_launch"]
            for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if k in counters:
                    d["frac_" + k] = counters[k]["avg_per_launch"] / wc
        out["derived"] = d
    with open(dst + "_pmc.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out.get("derived", {}), indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
