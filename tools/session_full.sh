#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_full.sh — every GPU test, then same-box A/B of the round-6 switches
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
t0=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu --durations=10 > "$out/pytest_gpu.txt" 2>&1; echo "pytest -m gpu rc $? in $(( $(date +%s) - t0 )) s"; tail -n 25 "$out/pytest_gpu.txt"
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 900 python bench.py --no-extra-legs --no-parity-mode --no-cpu-baseline --steps 3 --warmup 1 > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
  python - "$out/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:28s} value {r['value']:.4f} img/s  ms_per_pass {r['ms_per_step']:.1f}  denoise step {r['roofline']['denoise_step']['ms']:.3f} ms  gemm family {r['roofline']['achieved']} TF/s ({r['roofline']['launches_per_denoise_step']} launches, {r['roofline']['avg_launch_us']} us)")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run base MFHIP_PREFER_PERS=0 MFHIP_GN_FROM_PARTS=0
run all_no_fffold MFHIP_NO_FF_LNFOLD=1
run all
run base2 MFHIP_PREFER_PERS=0 MFHIP_GN_FROM_PARTS=0
