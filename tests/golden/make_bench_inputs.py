#!/usr/bin/env python3
"""Extracts the stimuli of the reference's own `__main__` benches (data, not code) into
tests/golden/ref_bench_inputs.json:

  mfcc/misc/fft.py:489-490      512 hard-coded int16 samples fed to FFT(size=512) (printed next to scipy's fft // 512)
  mfcc/core/dct_stream.py:75-   16 hard-coded samples fed to DCTStream (printed next to scipy's dct)
  mfcc/core/log.py:148          the single value 2207315 fed to Log2Fix(37, 20)
  mfcc/core/filterbank.py:147-  constant 1234 x 600 samples

Run in the build container (needs /root/reference); the JSON travels, the reference does not."""
import json
import os
import re

REF = "/root/reference/mfcc"
HERE = os.path.dirname(os.path.abspath(__file__))


def hex_list(path, after, name="data"):
    txt = open(path).read()
    i = txt.index(after)
    m = re.search(name + r"\s*=\s*\[(.*?)\]", txt[i:], re.S)
    return [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]


out = {}
vals = hex_list(os.path.join(REF, "misc", "fft.py"), 'if __name__ == "__main__":')
out["fft512_input_u16"] = vals
try:
    out["dct_input_u16"] = hex_list(os.path.join(REF, "core", "dct_stream.py"), 'if __name__ == "__main__":', "data_in")
except Exception:
    pass
out["log2fix_37_20_input"] = 2207315
out["filterbank_constant"] = {"value": 1234, "count": 600}
json.dump(out, open(os.path.join(HERE, "ref_bench_inputs.json"), "w"))
print({k: (len(v) if isinstance(v, list) else v) for k, v in out.items()})
