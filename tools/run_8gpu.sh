#!/bin/bash
# One command for the first hour on an 8-GPU MI355X node: the scaling curve of BASELINE.json configs[1] / [2]
# (batch-sharded inference, no data-path collective) at 1 / 2 / 4 / 8 GPUs and the data-parallel training step of configs[3]
# (bucketed gradient all-reduce over RCCL / xGMI under the backward pass) at 1 and 8 GPUs, as one table:
#     images/s, scaling factor vs 1 GPU, samples/s, all-reduce ms / GB/s per rank, overlap fraction.
# Nothing here has run on more than one GPU: the build container and the gpurun boxes have one.  The per-rank code paths are
# covered by world-2 gloo tests (tests/test_distributed_cpu.py) and a 2-ranks-on-one-device run (tests/test_pipeline_gpu.py).
#
#   bash tools/run_8gpu.sh [outdir]        # from the repo root; ~10 minutes
set -u
root="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-$root/gpurun_out/scale8}"
mkdir -p "$out"
cd "$root" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=${NCCL_DEBUG:-INFO} NCCL_DEBUG_SUBSYS=${NCCL_DEBUG_SUBSYS:-INIT}
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
echo "[run_8gpu] $ngpu GPUs visible"
run() {   # n, tag, bench args...
  n=$1; tag=$2; shift 2
  if [ "$n" -gt "$ngpu" ]; then echo "[run_8gpu] skip $tag: needs $n GPUs"; return; fi
  # bench.py --gpus N launches its own ranks (distributed.torchrun_argv: c10d rendezvous on a port the store binds itself)
  python bench.py --gpus "$n" "$@" > "$out/$tag.json" 2> "$out/$tag.err"
  rc=$?
  # RCCL must have seen every rank: its INIT log names the communicator's size
  ranks=$(grep -ao "nranks [0-9]*" "$out/$tag.err" | sort -u | tail -n 1)
  echo "[run_8gpu] $tag: rc $rc ${ranks:+(RCCL: $ranks)}"
}
for n in 1 2 4 8; do
  run "$n" "infer_n$n" --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-extra-legs
done
for n in 1 8; do
  run "$n" "train_n$n" --mode train --precision bf16 --steps 10 --warmup 3
done
# the overlap fraction of the gradient exchange: the same 8-rank step with the exchange forced AFTER the backward pass
MF_GRAD_SYNC_SERIAL=1 run 8 train_n8_serial --mode train --precision bf16 --steps 10 --warmup 3
python - "$out" <<'PY'
import json, os, sys
out = sys.argv[1]
def rec(tag):
    try:
        with open(os.path.join(out, tag + ".json")) as f:
            lines = [l for l in f if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None
print(f"{'workload':34s} {'GPUs':>4s} {'value':>10s} {'unit':12s} {'x 1 GPU':>8s} {'ms/step':>9s}  notes")
base = {}
for kind, tags in (("infer", [1, 2, 4, 8]), ("train", [1, 8])):
    for n in tags:
        r = rec(f"{kind}_n{n}")
        if r is None:
            continue
        base.setdefault(kind, r["value"] if n == 1 else None)
        sc = f"{r['value'] / base[kind]:.2f}" if base.get(kind) else "-"
        note = ""
        ar = r.get("all_reduce")
        if ar:
            note = f"all-reduce {ar['ms']} ms for {ar['bytes'] / 1e9:.2f} GB = {ar['GB/s_per_rank']} GB/s per rank"
            s = rec("train_n8_serial") if n == 8 else None
            if s:
                hidden = s["ms_per_step"] - r["ms_per_step"]
                note += f"; serial exchange {s['ms_per_step']} ms/step -> {max(0.0, min(1.0, hidden / ar['ms'])):.2f} of the exchange hidden under backward"
        print(f"{('configs[1]/[2] inference' if kind == 'infer' else 'configs[3] training step'):34s} {n:4d} {r['value']:10.3f} {r['unit']:12s} {sc:>8s} {r['ms_per_step']:9.2f}  {note}")
t8, t1 = rec("infer_n8"), rec("infer_n1")
if t8 and t1:
    print(f"north star: >= 6x images/s from 1 -> 8 GPUs: {t8['value'] / t1['value']:.2f}x")
PY
