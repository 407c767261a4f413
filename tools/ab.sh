#!/bin/bash
# A/B of experimental builds (tools/variants/*.so, same ABI) in one box session: kernel ms per variant, twice, interleaved
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  MFCC_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/$v.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-config5 $AB_ARGS > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.err; continue; }
  python3 -c "
import json; b=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); print('%-24s %.4f ms/step  kernel %.4f ms  %.3f G frames/s' % ('$v', b['ms_per_step'], b['roofline']['kernel_ms'], b['value']/1e9))"
done; done
