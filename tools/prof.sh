#!/bin/bash
# Diagnostic driver (scratch): bench + rocprofv3 passes for the round profile.  Usage: tools/prof.sh <tag>
set -o pipefail
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err || exit 1
tail -c 600 $OUT/bench.json; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --only $BENCH_ARGS > $OUT/kt.log 2>&1 || exit 2
echo kt done
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --only $BENCH_ARGS > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $i ($C) failed"; tail -3 $OUT/pmc_$i.log; }
  echo pmc $i done
done
# keep only the small csv files
find $OUT -name '*.db' -delete

du -sh $OUT
