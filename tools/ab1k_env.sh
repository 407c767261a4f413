#!/bin/bash
# A/B of the 1024 kernel's forms (MFCC_HIP_FUSED1024) on config 4's shape in ONE box session, interleaved, twice.
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for sel in "$@"; do
    echo -n "$sel  "; MFCC_HIP_FUSED1024=$sel timeout -k 10 200 python3 tools/run1024.py 2>/dev/null | tail -1
  done
done
