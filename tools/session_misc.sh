#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_misc.sh — ops tests touched this round + XL pipeline switches + same-box A/B of the bench
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_xl_gpu.py -q -m gpu -k "groupnorm or partial_sums or group_sums or split or tiny or denoising" > "$out/pytest_misc.txt" 2>&1; echo "pytest rc $?"; tail -n 6 "$out/pytest_misc.txt"
for i in 1 2; do
  timeout 900 python bench.py --no-extra-legs --no-parity-mode --no-cpu-baseline --steps 3 --warmup 1 > "$out/bench_$i.json" 2> "$out/bench_$i.err"
  python - "$out/bench_$i.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"value {r['value']:.4f} img/s  ms_per_pass {r['ms_per_step']:.1f}  denoise step {r['roofline']['denoise_step']['ms']:.3f} ms  frac {r['roofline']['denoise_step']['frac']}")
PY
done
