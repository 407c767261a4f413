#!/bin/bash
# the host-buffer pipeline against its chunk size and with / without pinning in place (diagnostic)
cd $GRAFT_REPO_ROOT
for np in 0 1; do for mb in 16 32 64 256; do
  echo -n "nopin=$np chunk=${mb}MB: "; REPS=15 ONLY_PROCESS=1 MFCC_HIP_HOST_NOPIN=$np MFCC_HIP_HOST_CHUNK_MB=$mb python3 tools/hostio_probe.py 2>/dev/null | head -1
done; done
