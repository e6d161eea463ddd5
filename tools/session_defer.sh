#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_defer.sh — deferred split-K reduce: parity tests, then a same-box A/B of the step
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "deferred or groupnorm" 2>&1 | tail -n 15
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_models_gpu.py tests/test_program_gpu.py -x -q -m gpu 2>&1 | tail -n 6
for i in 1 2 3; do
  for v in 0 1; do
    MFHIP_DEFER_REDUCE=$v timeout 900 python bench.py --no-extra-legs --no-parity-mode --no-cpu-baseline --steps 4 > "$out/b_${v}_$i.json" 2> "$out/b_${v}_$i.err" || tail -n 5 "$out/b_${v}_$i.err"
    python - "$out/b_${v}_$i.json" "$v" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("defer", sys.argv[2], r["value"], r["roofline"]["denoise_step"]["ms"])
PY
  done
done
