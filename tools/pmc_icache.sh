# Instruction-cache behaviour of the GEMM / conv kernels (round 4: are the once-executed prologue and epilogue of these 60-110 KB
# kernels instruction-fetch bound?).  rocprofv3 --pmc passes over single launches.  bash tools/pmc_icache.sh -> gpurun_out/pmc_icache.txt
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_icache.txt
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQ_INSTS_ALL|SQ_INST_LEVEL|SQ_WAVES\b|SQC_" | head -40 > $out
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY" "SQ_IFETCH_LEVEL SQC_ICACHE_MISSES_DUPLICATE SQ_BUSY_CYCLES"; do
  for shp in "8 16 16 1280 1280 1 44" "8 64 64 320 320 3 39" "8 64 64 320 320 1 26"; do
    tag=$(echo "$c$shp" | md5sum | cut -c1-8)
    timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmci_$tag -o p -- python3 tools/one_gemm.py bf16 $shp > gpurun_out/pmc_pass.log 2>&1 || echo "pass failed: $c / $shp" >> $out
    f=$(find gpurun_out/pmci_$tag -name "*counter_collection.csv" | head -1)
    echo "== counters [$c] shape [$shp]" >> $out
    if [ -n "$f" ]; then python3 - "$f" >> $out <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    if "gemm_conv" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
disp = collections.Counter((r["Kernel_Name"][:60], r["Dispatch_Id"]) for r in rows if "gemm_conv" in r["Kernel_Name"])
for k, d in acc.items():
    nd = len({dd for (kk, dd) in disp if kk == k})
    print("  ", k, "dispatches", nd, {c: round(v / max(nd, 1)) for c, v in d.items()})
PY
    fi
    rm -rf gpurun_out/pmci_$tag
  done
done
cat $out
