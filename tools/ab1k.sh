#!/bin/bash
# A/B of experimental builds (tools/variants/*.so) on config 4's shape in ONE box session, interleaved, twice.
# The variant's name picks the form of the 1024 kernel (MFCC_HIP_FUSED1024): bf16* the set-list form, f32* the per-rate fp32
# lists, anything else what the library picks.
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in "$@"; do
    sel=auto; case $v in bf16*) sel=bf16;; f32*) sel=f32;; esac
    echo -n "$v  "; MFCC_HIP_FUSED1024=$sel MFCC_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/$v.so timeout -k 10 200 python3 tools/run1024.py 2>/dev/null | tail -1
  done
done
