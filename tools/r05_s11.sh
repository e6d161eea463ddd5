#!/bin/bash
# round 5, GPU session 11: bisect the whole-pass slowdown against round 4 over this round's commits, same box; isolated launches A/B
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
B="--steps 3 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile"
run() { tag=$1; dir=$2; extra=$3; (cd $dir && MFHIP_TUNE_CACHE=/tmp/none_$tag.json timeout 400 python bench.py $B $extra > "$out/b_$tag.json" 2> "$out/b_$tag.err"); echo "$tag: $(grep -o '"value": [0-9.]*' "$out/b_$tag.json" | head -1) $(grep denoise "$out/b_$tag.err" | tail -1 | grep -o 'denoise [0-9.]* ms')"; }
run r04 _r04 ""
run c1 _c1 ""
run c2 _c2 ""
run c3 _c3 "--no-extra-legs"
run new . "--no-extra-legs"
run r04b _r04 ""
run newb . "--no-extra-legs"
for d in _r04 _c3 .; do echo "== ab_ops in $d"; (cd $d && timeout 300 python "$GRAFT_REPO_ROOT/tools/ab_ops.py" 2>&1 | tail -n 40); done | tee "$out/ab_ops.txt"
