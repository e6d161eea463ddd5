#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_ab.sh — model / pipeline parity tests, then same-box A/B of the round-6 switches on the default bench workload
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_models_gpu.py tests/test_pipeline_gpu.py tests/test_fp16_gpu.py -q -m gpu > "$out/pytest_models.txt" 2>&1; echo "pytest rc $?"; tail -n 6 "$out/pytest_models.txt"
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
run all
run all_no_fold MFHIP_NO_FF_LNFOLD=1
run base2 MFHIP_PREFER_PERS=0 MFHIP_GN_FROM_PARTS=0
run all2
