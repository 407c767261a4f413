#!/bin/bash
# shader clock (GRBM_GUI_ACTIVE / 8 / kernel time) of the 512 kernel in an experimental build: tools/clock_of.sh <variant> ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
  out=gpurun_out/clk_$v; rm -rf $out; mkdir -p $out
  MFCC_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/$v.so timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -- python3 tools/run512.py > $out/log.txt 2>&1
  python3 - $out $v <<'PY'
import csv, glob, sys
out, v = sys.argv[1], sys.argv[2]
cc = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
kt = glob.glob(out + '/**/*kernel_trace.csv', recursive=True)
g = [float(r['Counter_Value']) for r in csv.DictReader(open(cc[0])) if 'fused512' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
import collections
per = collections.defaultdict(float)
for r in csv.DictReader(open(cc[0])):
    if 'fused512' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE': per[r['Dispatch_Id']] += float(r['Counter_Value'])
d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(kt[0])) if 'fused512' in r['Kernel_Name']]
n = min(len(per), len(d))
gv = list(per.values())[-n // 2:]; dv = d[-n // 2:]
print(v, "launches", n, "avg ms %.4f" % (sum(dv) / len(dv) / 1e6), "clock GHz %.3f" % (sum(gv) / len(gv) / 8 / (sum(dv) / len(dv))))
PY
done
