"""Per-shape table of the mf_gemm_conv launches of one denoise step (HIP events around each launch)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reflecting_reality_amd import hip, synth  # noqa: E402

dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
inp = synth.pipeline_inputs(B, 512, 512)
x2 = torch.cat([inp["latents"].to(dev)] * 2)
cond = torch.randn(2 * B, 6, 64, 64, device=dev)
pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]]).to(dev)
for rep in range(3):
    if rep == 2:
        hip.profile_begin()
    d, m, u = pipe.brushnet(x2, 981, encoder_hidden_states=pe, brushnet_cond=cond, return_dict=False)
    pipe.unet(x2, 981, pe, down_block_add_samples=d, mid_block_add_sample=m, up_block_add_samples=u)
n, secs, flops = hip.profile_end()
print(f"{n} launches, {secs * 1e3:.2f} ms, {flops / secs / 1e12:.1f} TFLOP/s overall")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for t, f, k in hip.LAST_PROFILE:
    a = agg[k]
    a[0] += 1; a[1] += t; a[2] += f
print(f"{'M':>7} {'N':>6} {'K':>6} kh s u nz tile sk |  n   total_us  avg_us  TF/s  %time")
for k, (c, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[0]:7d} {k[1]:6d} {k[2]:6d} {k[3]:2d} {k[4]} {k[5]} {k[6]:2d} {k[7]:4d} {k[8]:2d} | {c:3d} {t * 1e6:9.1f} {t / c * 1e6:7.1f} {f / t / 1e12:6.1f} {100 * t / secs:5.1f}")
hip.tune_save()
if os.path.isdir("gpurun_out"):          # launch-ordered shape keys, to be joined with a rocprofv3 kernel trace
    import json
    json.dump([[f, list(k)] for _, f, k in hip.LAST_PROFILE], open("gpurun_out/shape_seq.json", "w"))
