"""The bench line contract (driver prompt, section 4): checked on the committed output of the last
`python bench.py` run on the GPU box (profiles/r01_bench.json) and on bench.py's static text."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    txt = open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1]
    return json.loads(txt)


def test_committed_bench_line_has_the_contract_keys():
    b = _line("r01_bench.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["metric"].startswith("MFCC frames/sec") and b["unit"] == "frames/s"
    assert b["higher_is_better"] is True and b["scaling"] == "weak" and b["vs_baseline"] is None
    assert b["data"] == "synthetic" and b["dtype"] == "f32" and b["n_gpus"] == 1
    assert "workload" in b["config"] and "model" not in b["config"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # algorithmic bytes over kernel time, not wall time; traffic above the algorithmic bytes
    frames = b["config"]["frames_per_step_per_gpu"]
    assert abs(r["achieved"] - frames * b["config"]["bytes_per_frame"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.5
    assert r["traffic"] is None or r["traffic"] >= frames * b["config"]["bytes_per_frame"]
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == "frames/s"
    # whole-job value is consistent with the step time
    assert abs(b["value"] - frames * b["n_gpus"] / (b["ms_per_step"] * 1e-3)) / b["value"] < 1e-3


def test_fixed_bench_line_reports_the_integer_path():
    b = _line("r01_bench_fixed.json")
    assert b["dtype"].startswith("int") and b["config"]["bytes_per_frame"] == 366
    assert "fixed" in b["config"]["workload"]


def test_bench_only_uses_the_oracle_for_the_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("from oracle import") == 1 and src.count("import oracle") == 0
    i = src.index("from oracle import")
    j = src.rindex("\ndef ", 0, i)
    assert src[j:].startswith("\ndef cpu_baseline("), "the oracle import must sit inside the cpu_baseline leg"
    # and that leg only runs when asked for, on rank 0, outside the timed regions
    # the definition, two guarded call sites on the batch (the sample and config 1's wav) and one on the corpus
    assert src.count("cpu_baseline(") == 4
    assert "with_cpu = not args.no_cpu_baseline" in src and src.count("if with_cpu:") == 2
    for call in re.finditer(r"= cpu_baseline\(", src):
        head = src.rfind("if with_cpu:", 0, call.start())
        assert head >= 0 and src.count("\ndef ", head, call.start()) == 0, "unguarded cpu_baseline call"
