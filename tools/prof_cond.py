"""Where the conditioning build spends its time (host preprocessing vs VAE encode vs small kernels)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflecting_reality_amd import hip, synth

dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
inp = synth.pipeline_inputs(4, 512, 512)
noise = torch.randn(8, 4, 64, 64)


def T(name, fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e3:7.2f} ms", flush=True)
    return r


img = T("prepare_image(image)", lambda: pipe.prepare_image(inp["image"], 512, 512, 4, 1))
m3 = T("prepare_image(mask)", lambda: pipe.prepare_image(inp["mask"], 512, 512, 4, 1))
om = T("original_mask (host)", lambda: (m3.sum(1)[:, None, :, :] < 0).to(torch.float32))
mom = T("vae._moments(img)", lambda: pipe.vae._moments(img))
imgd = img.to(dev)
T("vae._moments(img on device)", lambda: pipe.vae._moments(imgd))
T("noise.to(device)", lambda: noise.to(dev, torch.float32))
T("nearest_resize(mask.to(dev))", lambda: hip.nearest_resize(om.to(dev), 64, 64))
T("vae.decode", lambda: pipe.vae.decode(inp["latents"].to(dev), return_dict=False)[0])
out = pipe.vae.decode(inp["latents"].to(dev), return_dict=False)[0]
T("postprocess(pt)", lambda: pipe.image_processor.postprocess(out, output_type="pt", do_denormalize=[True] * 4))
T("whole build_conditioning", lambda: pipe.build_conditioning(inp["image"], inp["mask"], inp["depth"], 512, 512, 4, 1, True, noise))
print("--- inside build_conditioning ---")
sf = float(pipe.vae.config["scaling_factor"])
nd = noise.to(dev, torch.float32)
halves = T("vae_sample x2", lambda: [hip.vae_sample(mom, nd[i * 4:(i + 1) * 4].contiguous(), 4, sf) for i in range(2)])
mask_l = hip.nearest_resize(om.to(dev), 64, 64)
d = T("prepare_image(depth)", lambda: pipe.prepare_image(inp["depth"], 512, 512, 4, 1))
dl = T("nearest_resize(depth.to(dev))", lambda: hip.nearest_resize(d.to(dev), 64, 64))
extra = T("cat parts", lambda: torch.cat([mask_l, dl], 1))
T("final cat", lambda: torch.cat([torch.cat([h, extra], 1) for h in halves], 0).contiguous())
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    pipe.build_conditioning(inp["image"], inp["mask"], inp["depth"], 512, 512, 4, 1, True, noise)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
