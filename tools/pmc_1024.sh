#!/bin/bash
# PMC passes over the fused 1024 kernel (config 4's shape): tools/pmc_1024.sh <tag>; summary by profiles/summarize_1024.py
set -o pipefail
TAG=${1:-k1k}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 200 python3 tools/run1024.py > $OUT/run.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/run1024.py > $OUT/kt.log 2>&1 || exit 2
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  ITERS=4 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$i -- python3 tools/run1024.py > $OUT/pmc_$i.log 2>&1 || echo "pass $i failed"
  echo pmc $i done
done
find $OUT -name '*.db' -delete
cat $OUT/run.log | tail -1
