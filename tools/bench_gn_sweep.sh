# developer sweep of the two-launch GroupNorm's grid parameters (tools/bench_gn.py per setting)
cd "$GRAFT_REPO_ROOT"
for u in 4 8; do for ch in 64 32 16; do for bl in 2048 1024 512 256 128; do
  echo "== MFHIP_GN_UNROLL=$u MFHIP_GN_CHUNKS=$ch MFHIP_GN_APPLY_BLOCKS=$bl"
  MFHIP_GN_UNROLL=$u MFHIP_GN_CHUNKS=$ch MFHIP_GN_APPLY_BLOCKS=$bl python tools/bench_gn.py 2>&1 | grep -v amdgpu | head -4
done; done; done
