#!/bin/bash
# quick PMC comparison of the fused 512 forms: tools/pmc_quick.sh <tag> [env assignments passed to bench]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmcq_$1; rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config5 > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
find $OUT -name '*.db' -delete
python3 - <<PY
import csv, glob, collections
fr = 3613952
for d in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "fused512" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            a = sum(v)/len(v)
            print("%-28s %12.4g  per frame %8.2f" % (k, a, a/fr))
PY
