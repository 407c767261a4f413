#!/bin/bash
# A/B of experimental builds (tools/variants/*.so; "tree" = the in-tree library) on the headline shape (config 2: 64 x 10 min,
# 512 / 170 / 32 / 13) in ONE box session, interleaved, three times: bench.py --only, kernel time by HIP events
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in "$@"; do
    lib=$GRAFT_REPO_ROOT/tools/variants/$v.so; [ $v = tree ] && lib=$GRAFT_REPO_ROOT/mfcc_amd/libmfcc_hip.so
    echo -n "$v  "; MFCC_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --only --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernel_ms'], d['config']['kernel'])"
  done
done
