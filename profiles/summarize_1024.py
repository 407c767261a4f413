#!/usr/bin/env python3
"""profiles/summarize_1024.py <gpurun_out/tag> -- condense tools/pmc_1024.sh's output into profiles/r02_fused1024_pmc.json
(same HBM arithmetic as summarize_rocprof.py: FETCH_SIZE in KiB, doubled for 16-byte-per-lane streaming reads on gfx950)."""
import csv, glob, json, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfcc_amd
src = sys.argv[1]
out = {"kernel_source_hash": mfcc_amd.kernel_source_hash(),
       "what": "tools/pmc_1024.sh: the fused 1024 kernel on config 4's shape at 64 channels x 10 min (1 801 600 frames per launch); "
               "counters from separate rocprofv3 --pmc passes"}
for r in csv.DictReader(open(glob.glob(src + '/kt/*/*kernel_stats.csv')[0])):
    if 'fused1024' in r['Name']:
        out["kernel"] = "mfcc_fused1024::mfcc_fused1024_kernel"
        out["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
cnt = {}
for d in sorted(glob.glob(src + '/pmc_*')):
    f = glob.glob(d + '/*/*counter_collection.csv') if os.path.isdir(d) else []
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if 'fused1024' in r['Kernel_Name']:
            per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
            if 'dispatch' not in out:
                out['dispatch'] = {k: r[k] for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size', 'Grid_Size', 'Workgroup_Size') if k in r}
    for c, dd in per.items():
        v = list(dd.values())
        cnt[c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v), "pass": os.path.basename(d)}
out["counters"] = cnt
frames = 1801600
rd = cnt["FETCH_SIZE"]["avg_per_launch"] * 1024 * 2
wr = cnt["WRITE_SIZE"]["avg_per_launch"] * 1024
wc = cnt["SQ_WAVE_CYCLES"]["avg_per_launch"]
out["derived"] = {
    "frames_per_launch": frames, "algorithmic_bytes_per_launch": frames * 734, "hbm_read_bytes_per_launch": rd,
    "hbm_write_bytes_per_launch": wr, "hbm_traffic_bytes_per_launch": rd + wr, "traffic_over_algorithmic": (rd + wr) / (frames * 734),
    "valu_wave_instructions_per_frame": cnt["SQ_INSTS_VALU"]["avg_per_launch"] / frames,
    "mfma_per_frame": cnt["SQ_INSTS_MFMA"]["avg_per_launch"] / frames,
    "mfma_busy_clocks_per_frame": cnt["SQ_VALU_MFMA_BUSY_CYCLES"]["avg_per_launch"] / frames,
    "lds_instructions_per_frame": cnt["SQ_INSTS_LDS"]["avg_per_launch"] / frames,
    "frac_SQ_WAIT_ANY": cnt["SQ_WAIT_ANY"]["avg_per_launch"] / wc, "frac_SQ_ACTIVE_INST_ANY": cnt["SQ_ACTIVE_INST_ANY"]["avg_per_launch"] / wc,
    "lds_bank_conflict_over_idx_active": cnt["SQ_LDS_BANK_CONFLICT"]["avg_per_launch"] / cnt["SQ_LDS_IDX_ACTIVE"]["avg_per_launch"]}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'r02_fused1024_pmc.json'), 'w'), indent=1)
print(json.dumps(out["kernel_trace"]), "traffic/algorithmic %.4f" % out["derived"]["traffic_over_algorithmic"])
