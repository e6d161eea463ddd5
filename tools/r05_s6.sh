#!/bin/bash
# round 5, GPU session 6: fp16 mode parity (operator, layer, model, pipeline level), the attention MFMA-mix microbenchmark,
# the default bench line with its new legs
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 900 python -m pytest tests/test_fp16_gpu.py -x -q -m gpu > "$out/pytest_fp16_ops.txt" 2>&1; echo "pytest fp16 ops rc $?"; tail -n 8 "$out/pytest_fp16_ops.txt"
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_models_gpu.py tests/test_pipeline_gpu.py -q -m gpu -k "fp16" > "$out/pytest_fp16_models.txt" 2>&1; echo "pytest fp16 models rc $?"; tail -n 15 "$out/pytest_fp16_models.txt"
grep -a "ENVRATIO" "$out/pytest_fp16_models.txt" | head -5
./tools/micro/attn_mix > "$out/attn_mix.txt" 2>&1; cat "$out/attn_mix.txt"
timeout 1500 python bench.py --steps 4 --warmup 2 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc $?"
python - "$out/bench.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", r["value"], "step ms", r["roofline"]["denoise_step"]["ms"], "frac", r["roofline"]["denoise_step"]["frac"])
for k in ("parity_mode", "fp16_mode", "train_step", "sdxl"):
    v = r.get(k) or {}
    print(k, {a: v.get(a) for a in ("value", "unit", "denoise_step_ms", "ms_per_step", "error")})
PY
