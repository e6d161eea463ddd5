#!/bin/bash
# round-end measurement session (gpurun: bash tools/gpu_session.sh <tag> bash tools/measure_round.sh): every GPU test, smoke(), the default bench line, kernel stats of the same command, the counter passes, stamps
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 3000 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "pytest -m gpu rc $?"; tail -n 4 "$out/pytest_gpu.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.txt" 2>&1; echo "smoke rc $?"; tail -n 3 "$out/smoke.txt"
timeout 1800 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc $?"
python - "$out/bench_default.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", r["value"], r["unit"], "ms_per_step", r["ms_per_step"], "denoise step ms", r["roofline"]["denoise_step"]["ms"], "frac", r["roofline"]["frac"], r["roofline"]["denoise_step"]["frac"])
print("cpu_baseline", r.get("cpu_baseline"))
for k in ("parity_mode", "fp16_mode", "train_step", "sdxl"):
    v = r.get(k) or {}
    print(k, {a: v.get(a) for a in ("value", "unit", "denoise_step_ms", "ms_per_step", "error")})
PY
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o b -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-extra-legs --no-parity-mode --no-cpu-baseline > "$out/prof_bench.json" 2> "$out/prof_bench.err")
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*.db" -delete
head -n 12 "$out"/prof/*/b_kernel_stats.csv 2>/dev/null | cut -c1-200 || find "$out/prof" -name "*stats*" | head
bash tools/pmc_step.sh > "$out/pmc_step.txt" 2>&1; cp gpurun_out/pmc_gemm_family.json "$out/" 2>/dev/null; cat "$out/pmc_gemm_family.json"
bash tools/pmc_attn.sh > "$out/pmc_attn_d40.txt" 2>&1; tail -n 40 "$out/pmc_attn_d40.txt"
MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_stamps.so timeout 600 python tools/stamps.py 19,33,34,21,35,23,36,26,27,37,38,39 > "$out/stamps.txt" 2>&1
grep -v "^   ->\|^/opt" "$out/stamps.txt" | cut -c1-250
