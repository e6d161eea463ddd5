"""Export the denoise step of BASELINE configs[1] (batch 4 x 512 x 512, 50-step DDIM, CFG 7.5) as a step program, run the whole
loop from a C host with no Python in the process (examples/c_host/denoise_host.c), and compare its latents with the pipeline's.
usage: python tools/export_step.py [--precision bf16] [--steps 50] [--out /tmp/step.mfprog]"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reflecting_reality_amd import hip, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--out", default="/tmp/step.mfprog")
a = ap.parse_args()
dev = torch.device("cuda", 0)
hip.load()
pipe, _ = bench.build_pipeline(a.precision, dev)
inp = {k: v.to(dev) for k, v in synth.pipeline_inputs(4, 512, 512, seed=1234, cross_dim=768).items()}
kw = dict(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], image=inp["image"], mask=inp["mask"],
          depth=inp["depth"], num_inference_steps=a.steps, guidance_scale=7.5, latents=inp["latents"], output_type="latent",
          brushnet_conditioning_scale=1.0, height=512, width=512, conditioning_noise=inp["vae_noise"])
tm = {}
ref = pipe(**kw, _timing=tm).images.float().cpu()          # the pipeline's own loop: eager first step, capture, replays
tm = {}
ref2 = pipe(**kw, _timing=tm).images.float().cpu()
torch.cuda.synchronize()
print(f"pipeline (hipGraph on two streams): {tm['denoise_start'].elapsed_time(tm['denoise_end']) / a.steps:.3f} ms per denoise step; "
      f"repeatable: {torch.equal(ref, ref2)}")
pipe._graph_state = None
t0 = time.time()
info = pipe.export_denoise_step(a.out, **kw)
print(f"exported in {time.time() - t0:.1f} s: {info['calls']} calls, {info['buffers']} buffers, file {info['bytes'] / 1e9:.3f} GB "
      f"({info['const_bytes'] / 1e9:.3f} GB constants, {info['workspace_bytes'] / 1e9:.3f} GB workspace); entries {info['entries']}")
print("exporting run equals the plain run:", torch.equal(info["result"].images.float().cpu(), ref))
exe = "/tmp/denoise_host"
libdir = os.path.join(ROOT, "reflecting-reality_amd", "lib")
subprocess.run(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{os.path.join(ROOT, 'include')}",
                os.path.join(ROOT, "examples", "c_host", "denoise_host.c"), f"-L{libdir}", "-lmfhip", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe], check=True)
inp["latents"].float().cpu().contiguous().numpy().tofile("/tmp/lat_in.bin")
del pipe
torch.cuda.empty_cache()
torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
runtimes = {"ROCm 7.2 runtime (/opt/rocm/lib)": f"{libdir}:/opt/rocm/lib:", "the torch wheel's runtime": f"{libdir}:{torch_lib}:/opt/rocm/lib:"}
for (label, path), extra in [(r, e) for r in list(runtimes.items())[:1] for e in ([], ["--graph"])]:
    env = dict(os.environ, LD_LIBRARY_PATH=path + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, a.out, "/tmp/lat_in.bin", "/tmp/lat_out.bin"] + extra, capture_output=True, text=True, timeout=1200, env=env)
    print(f"[{label} {' '.join(extra)}]", out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stdout + out.stderr)
    if out.returncode == 0:
        got = torch.from_numpy(np.fromfile("/tmp/lat_out.bin", dtype=np.float32)).view(ref.shape)
        print("   C host latents == pipeline latents (bitwise):", torch.equal(got, ref), "max |diff|", float((got - ref).abs().max()))
# the same program replayed from THIS process (torch's streams, torch's graph capture): separates what the program's structure
# costs from what the C host's streams / instantiation cost
from reflecting_reality_amd import program  # noqa: E402
prog = program.Program(a.out, dev)
lat0 = prog.buffer("latents", torch.float32).clone()
for mode in ("eager", "graph"):
    prog.buffer("latents", torch.float32).copy_(lat0)
    g = None
    if mode == "graph":
        prog.run()
        prog.buffer("latents", torch.float32).copy_(lat0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            prog.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g.replay() if g is not None else prog.run()
    e1.record()
    torch.cuda.synchronize()
    print(f"[python host, {mode}] {e0.elapsed_time(e1) / 20:.3f} ms per program run")
prog.close()
os.remove(a.out)
