"""world_size-2 gloo run of the multi-GPU harness logic (sharding, barrier, max-over-ranks timing) on CPU."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    from reflecting_reality_amd import distributed as D
    rank, world, local = D.init_process_group("gloo")
    assert world == 2 and torch.distributed.get_backend() == "gloo"
    items = list(range(9))
    mine = D.shard(items, rank, world)
    D.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))                     # rank 1 is the slow one
    dt = D.max_over_ranks(time.perf_counter() - t0)
    total = D.sum_over_ranks(len(mine))
    D.barrier()
    # one file per rank: the two ranks share torchrun's stdout and their lines can interleave
    with open(os.path.join(os.environ["RESULT_DIR"], f"rank{rank}.json"), "w") as f:
        json.dump(dict(rank=rank, mine=mine, dt=dt, total=total), f)
""") % ROOT


def _torchrun(nproc: int):
    """torchrun prefix whose rendezvous store binds its own free port (no probe-then-release window)."""
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    from reflecting_reality_amd.distributed import torchrun_argv
    return torchrun_argv(nproc)


def test_two_rank_gloo_sharding_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RESULT_DIR=str(tmp_path))
    for attempt in range(3):             # (kept from the probed-port days; the c10d rendezvous needs no retry)
        cmd = _torchrun(2) + [str(script)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        if out.returncode == 0:
            break
        print(f"attempt {attempt}: rc={out.returncode}\n{out.stderr[-1500:]}")
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert res[0]["mine"] == [0, 1, 2, 3, 4] and res[1]["mine"] == [5, 6, 7, 8]      # contiguous, extra item to rank 0
    assert res[0]["total"] == res[1]["total"] == 9
    assert abs(res[0]["dt"] - res[1]["dt"]) < 1e-9 and res[0]["dt"] >= 0.1             # both report the slow rank's time


def test_bench_self_launch_two_ranks_stub():
    """`python bench.py --gpus 2` WITHOUT a torchrun environment starts its two ranks itself (child processes, before
    anything touches a GPU) and rank 0 prints one JSON line with n_gpus = 2; the time is the slowest rank's."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for attempt in range(3):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                              "--batch", "4", "--stub"], capture_output=True, text=True, timeout=300, env=env)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    # rank 1 sleeps 40 ms per pass: 3 passes >= 120 ms, and value = images of BOTH ranks / that time
    assert rec["ms_per_step"] >= 40.0
    assert abs(rec["value"] - 4 * 3 * 2 / (rec["ms_per_step"] * 3e-3)) < 0.05 * rec["value"]


def test_bench_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub"], capture_output=True,
                         text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (out.stderr + out.stdout)


GRAD_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import torch
    from collections import OrderedDict
    from reflecting_reality_amd import distributed as D
    from reflecting_reality_amd.autograd import Param
    rank, world, _ = D.init_process_group("gloo")

    class FakeModel:                       # the arena interface GradBuckets reads (models.HipModel.prepare_training)
        def __init__(self, sizes, seed):
            self._pmap, off = OrderedDict(), 0
            for i, n in enumerate(sizes):
                self._pmap[f"p{i}"] = (off, (n,), ("vec",))
                off += (n + 3) // 4 * 4
            self.n = off
            g = torch.Generator().manual_seed(seed)
            self.flat_g = torch.randn(off, generator=g)
        def num_arena_floats(self):
            return self.n
        def params(self):
            return [Param(k, None, self.flat_g[a:a + s[0]]) for k, (a, s, _) in self._pmap.items()]

    class FakeTape:
        on_param_grad = None

    sizes = [5, 1000, 37, 4096, 3, 250]
    models = [FakeModel(sizes, 100 + rank), FakeModel(sizes[:3], 200 + rank)]
    expect = [sum(FakeModel(s, seed + r).flat_g for r in range(world)) / world
              for s, seed in ((sizes, 100), (sizes[:3], 200))]
    gb = D.GradBuckets(models, bucket_floats=512)           # several buckets, parameters straddling bucket borders
    tape = FakeTape()
    gb.begin(tape)
    assert tape.on_param_grad is not None
    for m in models[::-1]:                                   # backward order: last model, last parameter first
        for p in m.params()[::-1]:
            if p.name != "p2":                               # a parameter that never reports: finish() flushes its bucket
                tape.on_param_grad(p)
    gb.finish()
    err = max(float((m.flat_g - e).abs().max()) for m, e in zip(models, expect))
    loss = D.gather_mean(torch.tensor(float(rank + 1)))
    with open(os.path.join(os.environ["RESULT_DIR"], f"grad{rank}.json"), "w") as f:
        json.dump(dict(err=err, loss=loss, nb=[p["nb"] for p in gb._plan]), f)
""") % ROOT


def test_bucketed_gradient_allreduce_equals_the_mean_of_the_ranks(tmp_path):
    """Data-parallel training's one exchange (train_brushnet_mirror.py:1267-1269, :1459): after GradBuckets.finish() every
    rank's gradient arena holds the mean of the ranks' gradients (= single-process summed gradients / world), whatever the
    order parameters finish in, with parameters straddling bucket borders and a parameter that never reports."""
    import json
    script = tmp_path / "grad_worker.py"
    script.write_text(GRAD_WORKER)
    env = dict(os.environ, RESULT_DIR=str(tmp_path))
    for attempt in range(3):
        cmd = _torchrun(2) + [str(script)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    res = [json.load(open(tmp_path / f"grad{r}.json")) for r in range(2)]
    for r in res:
        assert r["err"] < 1e-6 and r["nb"][0] > 5 and abs(r["loss"] - 1.5) < 1e-6


TUNE_WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    from reflecting_reality_amd import distributed as D, hip
    rank, world, local = D.init_process_group("gloo")
    out = os.environ["RESULT_DIR"]
    os.environ["MFHIP_TUNE_CACHE"] = os.path.join(out, "user_tune.json")
    log = []

    def warm():
        # stands in for a warm-up pass: a rank that finds the shape untuned "tunes" it (and would time every candidate tile)
        key = "9,9,7,77,777,1,1,0,0,1,0,0,1,1"
        hit = hip._tune_load().get(key)
        log.append(dict(rank=rank, t=time.time(), hit=hit is not None))
        if hit is None:
            time.sleep(0.3)
            hip._tune_load()[key] = (44, 6, 0)
            hip._tune_new[key] = (44, 6, 0)

    D.tuned_once(warm)
    with open(os.path.join(out, f"tune{rank}.json"), "w") as f:
        json.dump(log, f)
""") % ROOT


def test_cold_tune_cache_is_filled_by_rank_zero_only(tmp_path):
    """with several ranks and a cold tune cache only rank 0 tunes; it persists its winners, the other ranks
    re-read the cache behind a barrier and find the shape tuned (distributed.tuned_once, what bench.py's warm-up goes through)."""
    import json
    script = tmp_path / "tune_worker.py"
    script.write_text(TUNE_WORKER)
    env = dict(os.environ, RESULT_DIR=str(tmp_path))
    for attempt in range(3):
        cmd = _torchrun(2) + [str(script)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    r0, r1 = (json.load(open(tmp_path / f"tune{r}.json")) for r in range(2))
    assert len(r0) == 1 and len(r1) == 1
    assert r0[0]["hit"] is False and r1[0]["hit"] is True, (r0, r1)            # rank 1 found rank 0's winner: it did not tune
    assert r1[0]["t"] >= r0[0]["t"] + 0.3                                       # ... and only started after rank 0 had finished
    saved = json.load(open(tmp_path / "user_tune.json"))
    assert saved["entries"]["9,9,7,77,777,1,1,0,0,1,0,0,1,1"] == [44, 6, 0]
