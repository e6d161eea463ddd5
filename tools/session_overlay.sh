#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_overlay.sh <overlay.json> — same-box A/B of a tune-cache overlay against the shipped cache
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
ov="${1:?usage: session_overlay.sh <overlay.json>}"
echo '{"_meta": {"tile_table": 1}, "entries": {}}' > "$out/empty.json"
cp "$ov" "$out/overlay.json"
for i in 1 2 3 4; do
  for v in empty overlay; do
    MFHIP_TUNE_CACHE="$out/$v.json" timeout 900 python bench.py --no-extra-legs --no-parity-mode --no-cpu-baseline --steps 4 > "$out/b_${v}_$i.json" 2> "$out/b_${v}_$i.err" || tail -n 5 "$out/b_${v}_$i.err"
    python - "$out/b_${v}_$i.json" "$v" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], r["value"], r["roofline"]["denoise_step"]["ms"])
PY
  done
done
