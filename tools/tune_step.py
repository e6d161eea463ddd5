"""Whole-step tuning of the (tile, split-K) choices of the denoise step (BASELINE configs[1]: batch 4 x 512 x 512, bf16).

The per-shape autotuner (hip._tuned_config) ranks candidates by the time of the launch ALONE.  In the step the launches run from
one hipGraph on two streams (BrushNet || UNet) between other kernels, where a tile that holds a whole CU (a 110-135 KB ring)
and one that leaves room for the other stream's blocks are not worth what they are worth alone.  This tool
  1. re-times every candidate of every GEMM key of the step in isolation (graph-timed) and keeps the whole ranking,
  2. walks the keys by their share of the step and tries each runner-up INSIDE the replayed step, keeping a change only when the
     step itself gets faster (coordinate descent on the metric the bench reports),
  3. writes the winners to gpurun_out/tune_cache_new.json (tools/merge_tune.py merges them into the shipped cache).
usage: python tools/tune_step.py [--precision bf16] [--top 3] [--within 0.10] [--max-evals 150] [--no-isolated]
"""
import argparse
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reflecting_reality_amd import hip, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--top", type=int, default=3)
ap.add_argument("--within", type=float, default=0.10, help="runner-ups within this fraction of the isolated best are tried")
ap.add_argument("--max-evals", type=int, default=150)
ap.add_argument("--no-isolated", action="store_true", help="skip the isolated re-tune (rankings from gpurun_out/tune_rankings.json)")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tune_cache_new.json"))
ap.add_argument("--overlay", default="", help="a tune_cache_new.json of an earlier run: its entries replace the isolated winners before the search")
ap.add_argument("--passes", type=int, default=1, help="sweeps over the keys (a later sweep re-tries alternatives against the changed neighbourhood)")
a = ap.parse_args()

dev = torch.device("cuda", 0)
hip.load()
pipe, _ = bench.build_pipeline(a.precision, dev)
inp = {k: v.to(dev) for k, v in synth.pipeline_inputs(4, 512, 512, seed=1234, cross_dim=768).items()}


def run(steps, timing=None):
    return pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], image=inp["image"],
                mask=inp["mask"], depth=inp["depth"], num_inference_steps=steps, guidance_scale=7.5, latents=inp["latents"],
                output_type="latent", brushnet_conditioning_scale=1.0, height=512, width=512, conditioning_noise=inp["vae_noise"],
                _timing=timing)


def step_ms(reps=2, steps=16):
    """ms per replayed denoise step with the cache as it stands: the graph is re-captured, then whole passes of replays are timed"""
    pipe._graph_state = None
    run(3)                                   # eager step + capture + one replay
    best = float("inf")
    for _ in range(reps):
        tm = {}
        run(steps, tm)
        torch.cuda.synchronize()
        best = min(best, tm["denoise_start"].elapsed_time(tm["denoise_end"]) / steps)
    return best


rank_path = os.path.join(ROOT, "gpurun_out", "tune_rankings.json")
t0 = time.time()
print(f"[tune_step] step with the shipped cache: {step_ms():.3f} ms", flush=True)      # what the result below has to beat on THIS box
if not a.no_isolated:
    hip.RETUNE, hip.TUNE_GRAPH, hip.TUNE_LOG = True, True, {}
    run(2)                                   # the eager first step re-times every candidate of every key it meets
    hip.RETUNE, hip.TUNE_GRAPH = False, False
    rankings = {k: sorted(v)[:8] for k, v in hip.TUNE_LOG.items()}
    hip.TUNE_LOG = None
    os.makedirs(os.path.dirname(rank_path), exist_ok=True)
    with open(rank_path, "w") as f:
        json.dump(rankings, f)
    print(f"[tune_step] isolated re-tune of {len(rankings)} keys in {time.time() - t0:.0f} s", flush=True)
else:
    with open(rank_path) as f:
        rankings = {k: [tuple(x) for x in v] for k, v in json.load(f).items()}

# which keys does one denoise step launch, and how often
hip.KEY_LOG = []
pipe._graph_state = None
pipe.use_hip_graph = False
run(2)
pipe.use_hip_graph = True
keys = collections.Counter(hip.KEY_LOG)
hip.KEY_LOG = None
n_steps_logged = 2
base_of = lambda k: k.split("@")[0]          # "<key>@<ctx>" (hip.TUNE_CTX): the isolated ranking is the plain key's
share = {k: (rankings[base_of(k)][0][0] / 8.0 * c / n_steps_logged) for k, c in keys.items() if base_of(k) in rankings}      # ms per step
order = sorted(share, key=lambda k: -share[k])
if os.environ.get("MF_TUNE_STEP_SHARES"):          # the step's GEMM time by key (count per step x best isolated time), largest first
    with open(os.environ["MF_TUNE_STEP_SHARES"], "w") as f:
        for k in order:
            r0 = rankings[base_of(k)][0]
            f.write(f"{share[k] * 1e3:9.1f} us/step  {keys[k] / n_steps_logged:5.1f} x {r0[0] / 8.0 * 1e3:7.1f} us  tile {r0[1]} sk {r0[2]}  {k}\n")
print(f"[tune_step] {len(order)} GEMM keys in the step; isolated sum {sum(share.values()):.2f} ms per step (graph-timed, 8 launches each)", flush=True)

cache = hip._tune_load()
base_of = lambda k: k.split("@")[0]
if a.overlay:
    with open(a.overlay) as f:
        ov = json.load(f)["entries"]
    n_ov = 0
    for k, v in ov.items():
        if base_of(k) in rankings and tuple(cache.get(k, ())) != tuple(v):
            cache[k] = tuple(v)
            n_ov += 1
    print(f"[tune_step] overlay {a.overlay}: {n_ov} entries differ from the isolated winners", flush=True)
base = step_ms()
print(f"[tune_step] step with the isolated winners: {base:.3f} ms", flush=True)
evals, changed = 0, {}
for k in [k for _ in range(a.passes) for k in order]:
    r = rankings[base_of(k)]
    best_iso = r[0][0]
    alts = [c for c in r[:a.top] if c[0] <= best_iso * (1.0 + a.within)]
    cur = tuple(cache.get(k) or cache.get(base_of(k)) or r[0][1:])       # (a key only ever looked up under its position tag has no plain entry)
    for (dt, t, s, mode) in alts:
        if evals >= a.max_evals:
            break
        cand = (t, s, mode)
        if cand == cur:
            continue
        cache[k] = cand
        ms = step_ms()
        evals += 1
        if ms < base * (1.0 - 0.0015):
            ms2 = step_ms()                  # confirm
            if ms2 < base * (1.0 - 0.001):
                print(f"  {k}: {cur} -> {cand} (isolated {best_iso * 1e3 / 8:.1f} -> {dt * 1e3 / 8:.1f} us): step {base:.3f} -> {min(ms, ms2):.3f} ms", flush=True)
                base, cur = min(ms, ms2), cand
                changed[k] = cand
                continue
        cache[k] = cur
    if evals >= a.max_evals:
        break
print(f"[tune_step] {evals} step evaluations, {len(changed)} keys changed, step {base:.3f} ms, {time.time() - t0:.0f} s", flush=True)
ver = hip.load().mf_gemm_tile_table_version()
entries = {k: list(cache[k]) for k in list(rankings) + list(changed) if k in cache}
with open(a.out, "w") as f:
    json.dump({"_meta": {"tile_table": ver}, "entries": dict(sorted(entries.items()))}, f, indent=0)
print(f"[tune_step] wrote {len(entries)} entries to {a.out}", flush=True)
