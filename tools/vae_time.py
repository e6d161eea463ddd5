import os, sys, time
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench
from reflecting_reality_amd import synth, hip
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
inp = synth.pipeline_inputs(4, 512, 512)
lat = inp["latents"].to(dev); img = inp["image"].to(dev)
for _ in range(2):
    pipe.vae.decode(lat, return_dict=False); pipe.vae._moments(img)
torch.cuda.synchronize()
for name, fn in (("decode", lambda: pipe.vae.decode(lat, return_dict=False)), ("encode", lambda: pipe.vae._moments(img))):
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(name, (time.perf_counter() - t0) / 5 * 1e3, "ms")
