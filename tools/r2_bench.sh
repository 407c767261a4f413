#!/bin/bash
# bench lines of the round: driver-style default run, config 5, fixed, self-launched 2-rank gloo rehearsal
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || { tail -5 $OUT/bench_driver.err; exit 1; }
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
timeout -k 10 400 python3 bench.py --config 5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err || { tail -5 $OUT/bench_c5.err; exit 1; }
timeout -k 10 400 python3 bench.py --fixed --no-config5 > $OUT/bench_fixed.json 2> $OUT/bench_fixed.err || { tail -5 $OUT/bench_fixed.err; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --channels 16 --utterances 2000 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err || { tail -5 $OUT/bench_n2_gloo.err; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --config 5 --utterances 2000 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_n2_gloo_c5.json 2> $OUT/bench_n2_gloo_c5.err || { tail -5 $OUT/bench_n2_gloo_c5.err; exit 1; }
for f in bench_driver bench_default bench_c5 bench_fixed bench_n2_gloo bench_n2_gloo_c5; do echo "== $f"; python3 - <<PY
import json
b = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["scaling"], b["roofline"]["frac"], b["roofline"]["kernel_ms"], b["roofline"].get("traffic"), (b.get("cpu_baseline") or {}).get("value"))
if "config5" in b: print("  c5:", b["config5"]["value"], b["config5"]["ms_per_step"], b["config5"]["roofline"]["kernel_ms"], b["config5"]["gather"])
if b.get("gather"): print("  gather:", b["gather"])
PY
done
