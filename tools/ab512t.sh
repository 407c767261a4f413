#!/bin/bash
# A/B of experimental builds (tools/variants/*.so) on config 2's shape, timing only (tools/run512.py), interleaved, twice
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in "$@"; do
    echo -n "$v  "; MFCC_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/$v.so timeout -k 10 200 python3 tools/run512.py 2>/dev/null | tail -1
  done
done
