"""Merge tuned (tile, split-K) winners a GPU run left in gpurun_out/tune_cache_new.json into the cache shipped with the
package (reflecting-reality_amd/tune_cache.json).  Entries recorded under another tile-table version are dropped."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "reflecting-reality_amd", "tune_cache.json")
srcs = sys.argv[1:] or [os.path.join(ROOT, "gpurun_out", "tune_cache_new.json")]
with open(dst) as f:
    base = json.load(f)
ver = base["_meta"]["tile_table"]
n0 = len(base["entries"])
for src in srcs:
    with open(src) as f:
        new = json.load(f)
    if new.get("_meta", {}).get("tile_table") != ver:
        print(f"{src}: tile table version {new.get('_meta')} != {ver}, skipped")
        continue
    base["entries"].update(new["entries"])
with open(dst, "w") as f:
    json.dump({"_meta": base["_meta"], "entries": dict(sorted(base["entries"].items()))}, f, indent=0)
print(f"{dst}: {n0} -> {len(base['entries'])} entries")
