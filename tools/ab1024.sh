#!/bin/bash
# A/B of the two fused 1024 kernels in ONE box session (config 4's shape at 64 channels x 10 min): eight-frame tiles (default)
# against round 2's sixteen-frame tiles (MFCC_HIP_FUSED1024=t16), interleaved, twice
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in t8 t16; do
    echo -n "$v  "; MFCC_HIP_FUSED1024=$v timeout -k 10 200 python3 tools/run1024.py | tail -1
  done
done
