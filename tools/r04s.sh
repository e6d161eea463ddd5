#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "tiles or warp_specialised" 2>&1 | grep -a "passed\|failed\|Error\|error\|assert" | tail -6
export MFHIP_RETUNE=1
timeout 2400 python bench.py --no-cpu-baseline --no-parity-mode 2>&1 | grep -a '"metric"' | cut -c1-250
unset MFHIP_RETUNE
python - <<'PY'
import json, collections
j=json.load(open("gpurun_out/tune_cache_new.json"))
c=collections.Counter(v[0] for v in j["entries"].values())
print("tiles picked:", sorted(c.items()))
print({k:v for k,v in j["entries"].items() if v[0] in (49,50)})
PY
cp gpurun_out/tune_cache_new.json $MF_SESSION_OUT/
