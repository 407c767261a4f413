#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=VALUE ...]: an experimental build of the library (same ABI) for A/B runs
cd "$(dirname "$0")/.." && mkdir -p tools/variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -shared -o tools/variants/$name.so mfcc_amd/csrc/mfcc_hip.hip 2>&1 | grep -E "error|warning: v|spill" ; ls -la tools/variants/$name.so | awk '{print $5, $9}'
