#!/bin/bash
# Diagnostic driver (scratch): pipe-overlap counters for the fused 512 kernel.  Usage: tools/prof2.sh <tag>
set -o pipefail
TAG=${1:-r2a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $i ($C) failed"; tail -3 $OUT/pmc_$i.log; }
  echo pmc $i done
done
find $OUT -name "*.db" -delete
du -sh $OUT
