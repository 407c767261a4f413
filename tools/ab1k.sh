#!/bin/bash
# A/B of experimental builds (tools/variants/*.so) on config 4's shape in ONE box session, interleaved, twice.
# A variant whose name starts with t16 runs round 2's sixteen-frame-tile kernel (MFCC_HIP_FUSED1024=t16), pc* the
# producer / consumer kernel, anything else the eight-frame-tile one.
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in "$@"; do
    sel=t8; case $v in t16*) sel=t16;; pc*) sel=pc;; esac
    echo -n "$v  "; MFCC_HIP_FUSED1024=$sel MFCC_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/$v.so timeout -k 10 200 python3 tools/run1024.py 2>/dev/null | tail -1
  done
done
