"""bf16 mixed-precision error envelope of the REFERENCE'S OWN TRAINING STEP: the first step of the 'frozen' case of
tools/make_golden.py (tiny_train backward half: BrushNet trainable, UNet frozen) run the way
train_brushnet_mirror.py runs with --mixed_precision=bf16 — the frozen UNet cast to bf16 (weight_dtype, :1127-1131,
1271-1273), fp32 BrushNet master weights, the forward under torch.autocast(bfloat16) (accelerate's prepare), the loss in
fp32 — and compared with the reference's fp32 gradients (tests/golden/tiny_train_backward.npz).

tests/ assert that the HIP 'bf16x1' mode (fp32 storage, operands rounded to bf16 for one MFMA per product) stays inside a
stated multiple of these deviations — a bound derived from the reference, not from our own kernels.

Runs only in the build container (needs /root/reference).  Output: tests/golden/bf16_train_envelope.json (numbers only).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as MG  # noqa: E402  (sets up the reference import path + shim)
from make_golden import BrushNetModel, R, GOLD  # noqa: E402
from diffusers import DDPMScheduler  # noqa: E402


def main():
    gold = np.load(os.path.join(GOLD, "tiny_train_backward.npz"))
    ucfg, vcfg = R.TINY_UNET, R.TINY_VAE
    torch.set_grad_enabled(True)
    (unet, _, _), _, _ = MG.models(ucfg, vcfg, 0)
    bn = BrushNetModel.from_unet(unet, conditioning_channels=5, load_weights_from_unet=False)
    MG.load_synth(bn, 21)
    bn.train()
    unet.requires_grad_(False)
    unet.to(torch.bfloat16)                                                       # weight_dtype (:1271-1273)
    # the inputs of step 0 of tiny_train(): same seeds, same order of draws
    g = torch.Generator().manual_seed(2024)
    bsz = 3
    latents = torch.randn(bsz, 4, 8, 8, generator=g) * 0.8
    noise = torch.randn(bsz, 4, 8, 8, generator=g)
    cond = torch.randn(bsz, 5, 8, 8, generator=g)
    ehs = torch.randn(bsz, 77, ucfg["cross_attention_dim"], generator=g)
    timesteps = torch.tensor([17, 480, 965]).long()
    ns = DDPMScheduler(num_train_timesteps=1000, beta_start=R.SD15_SCHED["beta_start"], beta_end=R.SD15_SCHED["beta_end"],
                       beta_schedule="scaled_linear")
    noisy = ns.add_noise(latents, noise, timesteps)
    wd = torch.bfloat16
    with torch.autocast("cpu", dtype=torch.bfloat16):
        down, mid, up = bn(noisy, timesteps, encoder_hidden_states=ehs, brushnet_cond=cond, return_dict=False)
        pred = unet(noisy.to(wd), timesteps, encoder_hidden_states=ehs.to(wd),
                    down_block_add_samples=[s.to(wd) for s in down], mid_block_add_sample=mid.to(wd),
                    up_block_add_samples=[s.to(wd) for s in up], return_dict=False)[0]            # MirrorFusionModel.forward (:858-888)
    loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    loss.backward()
    gn = float(torch.nn.utils.clip_grad_norm_(list(bn.parameters()), 1e30))
    ref_loss, ref_gn = float(gold["frozen_loss_0"]), float(gold["frozen_grad_norm_0"])
    # the fixture's inputs must be the golden's: the fp32 loss of this very batch is in the file
    out = {"_note": "reference (diffusers fork) under --mixed_precision=bf16 semantics vs its own fp32 step, tiny 'frozen' case, step 0",
           "loss": {"bf16": float(loss), "fp32": ref_loss, "abs_dev": abs(float(loss) - ref_loss)},
           "grad_norm": {"bf16": gn, "fp32": ref_gn, "rel_dev": abs(gn - ref_gn) / ref_gn}, "grads": {}}
    params = dict(bn.named_parameters())
    for k in gold.files:
        if not k.startswith("frozen_grad/"):
            continue
        name = k[len("frozen_grad/"):]
        ref = torch.from_numpy(gold[k]).float()
        got = params[name].grad.float()
        out["grads"][name] = {"rel_l2": float((got - ref).norm() / ref.norm().clamp_min(1e-30)),
                              "cos": float(F.cosine_similarity(got.flatten(), ref.flatten(), dim=0)),
                              "ref_norm": float(ref.norm())}
    print(json.dumps(out, indent=1))
    with open(os.path.join(GOLD, "bf16_train_envelope.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
