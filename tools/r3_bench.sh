#!/bin/bash
# bench lines of the round: driver-style default run, defaults, configs 3 / 4 / 5 on their own, host buffers, the 2-rank gloo
# rehearsal and a 1-rank RCCL group
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 500 python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; }; }
run bench_driver --steps 20 --warmup 5
run bench_default
run bench_c3 --config 3
run bench_c4 --config 4
run bench_c5 --config 5
run bench_hostio --host-io --no-cpu-baseline
run bench_n1_nccl_c5 --config 5 --group --backend nccl --no-cpu-baseline
run bench_n2_gloo --gpus 2 --backend gloo --channels 16 --utterances 2000 --steps 10 --warmup 2 --no-cpu-baseline --no-config4
run bench_n2_gloo_c5 --gpus 2 --backend gloo --config 5 --utterances 2000 --steps 10 --warmup 2 --no-cpu-baseline
for f in bench_driver bench_default bench_c3 bench_c4 bench_c5 bench_hostio bench_n1_nccl_c5 bench_n2_gloo bench_n2_gloo_c5; do echo "== $f"; python3 - <<PY
import json
try:
    b = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
except Exception as e:
    print("  no line:", e); raise SystemExit
def row(tag, o):
    r = o["roofline"]
    print("  %-8s %.3f G frames/s  %.4f ms/step  kernel %.4f ms  hbm frac %.4f  traffic %s  valu %s alu %s" % (
        tag, o["value"] / 1e9, o["ms_per_step"], r["kernel_ms"], r["frac"], r.get("traffic"),
        (o.get("valu_roofline") or {}).get("frac"), (o.get("alu_roofline") or {}).get("frac")))
row("line", b)
for k in ("config3", "config4", "config5"):
    if k in b: row(k, b[k])
if b.get("cpu_baseline"): print("  cpu", b["cpu_baseline"]["value"], (b["cpu_baseline"].get("config1") or {}).get("value"))
if b.get("pcie_inclusive"): print("  pcie", b["pcie_inclusive"])
if b.get("gather"): print("  gather", b["gather"])
if b.get("host_enqueue"): print("  enqueue", b["host_enqueue"])
PY
done
