"""VAE encode + decode (batch 4 x 512x512) for rocprofv3 --kernel-trace --stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflecting_reality_amd import synth
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
inp = synth.pipeline_inputs(4, 512, 512)
for _ in range(3):
    z = pipe.vae.decode(inp["latents"].to(dev), return_dict=False)[0]
    pipe.vae._moments(inp["image"])
torch.cuda.synchronize()
