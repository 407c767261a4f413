#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmc_generic; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 tools/bench_generic.py > $OUT/p1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH --output-format csv -d $OUT/p2 -- python3 tools/bench_generic.py > $OUT/p2.log 2>&1
find $OUT -name '*.db' -delete
python3 - <<PY
import csv, glob, collections
for d in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if ("mfcc_fixed_kernel" in r["Kernel_Name"] or "generic" in r["Kernel_Name"]):
                agg[r["Grid_Size"] + "/" + r["LDS_Block_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for g, c in agg.items():
            print(g, {k: "%.4g" % (sum(v)/len(v)) for k, v in c.items()})
PY
