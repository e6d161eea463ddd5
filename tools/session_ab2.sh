#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_ab2.sh — same-box A/B of single switches on the default bench workload
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 900 python bench.py --no-extra-legs --no-parity-mode --no-cpu-baseline --steps 3 --warmup 1 > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
  python - "$out/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:28s} value {r['value']:.4f} img/s  ms_per_pass {r['ms_per_step']:.1f}  denoise step {r['roofline']['denoise_step']['ms']:.3f} ms")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run all
run no_groups MFHIP_GN_FROM_GROUPS=0
run no_parts MFHIP_GN_FROM_PARTS=0
run all2
run no_groups2 MFHIP_GN_FROM_GROUPS=0
