"""Run a few BrushNet+UNet denoise steps (batch 4 x 512x512, CFG) for rocprofv3:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- python3 tools/profile_step.py --steps 3
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--vae", action="store_true", help="also run one VAE encode + decode")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    pipe, _ = bench.build_pipeline(a.precision, dev)
    from reflecting_reality_amd import synth
    inp = synth.pipeline_inputs(a.batch, a.size, a.size)
    x2 = torch.cat([inp["latents"].to(dev)] * 2)
    cond = torch.randn(2 * a.batch, 6, a.size // 8, a.size // 8, device=dev)
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]]).to(dev)
    for i in range(a.steps + 1):
        if i == 1:
            torch.cuda.synchronize()
            t0 = time.time()
        d, m, u = pipe.brushnet(x2, 981, encoder_hidden_states=pe, brushnet_cond=cond, return_dict=False)
        pipe.unet(x2, 981, pe, down_block_add_samples=d, mid_block_add_sample=m, up_block_add_samples=u)
    torch.cuda.synchronize()
    print(f"{(time.time() - t0) / a.steps * 1e3:.2f} ms per denoise step (eager, wall)")
    from reflecting_reality_amd import hip
    hip.tune_save()
    import shutil
    if os.path.isdir("gpurun_out"):
        hip.tune_save("gpurun_out/tune_cache_new.json")
    if a.vae:
        z = pipe.vae.decode(inp["latents"].to(dev), return_dict=False)[0]
        pipe.vae._moments(inp["image"])
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
