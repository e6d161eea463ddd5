"""Forward half of the reference's training step.

Mirrors examples/brushnet/train_brushnet_mirror.py: `MirrorFusionModel` (:836-888, BrushNet residuals injected
into the UNet, conditioning scale 1), `compute_snr` (src/diffusers/training_utils.py:50-73) and the loss block
(:1407-1449: add_noise at per-sample timesteps, epsilon / v_prediction target, plain or min-SNR-weighted MSE).
Everything here runs on the HIP kernels behind libmfhip; there is no autograd graph, so `backward()`, gradient
clipping and the optimizer step are NOT built (they raise) — this module gives the per-step loss the reference
logs, e.g. for validation-loss curves over a checkpoint, not training itself.
"""
from typing import Optional

import torch

from . import hip

__all__ = ["MirrorFusionModel", "compute_snr", "training_loss"]


class MirrorFusionModel:
    """train_brushnet_mirror.py:836-888 with `normal_proj_model=None` (the ip_adapter normals branch needs the
    IP-Adapter attention processors, which are outside the path)."""

    def __init__(self, unet, brushnet, normal_proj_model=None, adapter_modules=None, freq_encoder=None,
                 weight_dtype=torch.float32):
        if normal_proj_model is not None or adapter_modules is not None or freq_encoder is not None:
            raise NotImplementedError("normals_conditioning_mode='ip_adapter' is not built (SURVEY.md §8 f-4)")
        self.unet = unet
        self.brushnet = brushnet
        self.normal_proj_model = None
        self.adapter_modules = None
        self.freq_encoder = None
        self.weight_dtype = weight_dtype

    def get_trainable_modules(self, verbose: bool = True):
        raise NotImplementedError("no autograd on the HIP path: training (backward / optimizer) is not built")

    def forward(self, noisy_latents, timesteps, encoder_hidden_states=None, conditioning_latents=None, normal=None):
        if normal is not None:
            raise NotImplementedError("normals_conditioning_mode='ip_adapter' is not built (SURVEY.md §8 f-4)")
        down, mid, up = self.brushnet(noisy_latents, timesteps, encoder_hidden_states=encoder_hidden_states,
                                      brushnet_cond=conditioning_latents, return_dict=False)        # :860-866
        return self.unet(noisy_latents, timesteps, encoder_hidden_states=encoder_hidden_states,
                         down_block_add_samples=down, mid_block_add_sample=mid, up_block_add_samples=up,
                         return_dict=False)[0]                                                        # :874-886

    __call__ = forward


def compute_snr(noise_scheduler, timesteps: torch.Tensor) -> torch.Tensor:
    """training_utils.py:50-73: (sqrt(a_t) / sqrt(1 - a_t))**2 in fp32, one value per timestep (host tensor:
    `alphas_cumprod` is a 1000-entry host table, like the scheduler's own coefficient tables)."""
    a = noise_scheduler.alphas_cumprod
    ts = timesteps.cpu().long()
    alpha = (a ** 0.5)[ts].float()
    sigma = ((1.0 - a) ** 0.5)[ts].float()
    return (alpha / sigma) ** 2


def training_loss(model: MirrorFusionModel, noise_scheduler, latents: torch.Tensor, noise: torch.Tensor,
                  timesteps: torch.Tensor, encoder_hidden_states: torch.Tensor, conditioning_latents: torch.Tensor,
                  snr_gamma: Optional[float] = None):
    """train_brushnet_mirror.py:1407-1449 for already-encoded inputs: returns (loss [1] fp32 on the device,
    model_pred, target).  `latents` are the scaled VAE latents, `noise`/`timesteps` the script's random draws
    (:1408-1412), `conditioning_latents` the 5- (or 9-/8-) channel BrushNet conditioning (:1372-1402)."""
    noisy = noise_scheduler.add_noise(latents, noise, timesteps)                                     # :1416
    pred = model(noisy, timesteps, encoder_hidden_states, conditioning_latents)                      # :1422
    ptype = noise_scheduler.config["prediction_type"]
    if ptype == "epsilon":                                                                           # :1427-1432
        target = noise.to(pred.device, torch.float32)
    elif ptype == "v_prediction":
        target = noise_scheduler.get_velocity(latents, noise, timesteps)
    else:
        raise ValueError(f"Unknown prediction type {ptype}")
    weights = None
    if snr_gamma is not None:                                                                        # :1437-1449
        snr = compute_snr(noise_scheduler, timesteps)
        weights = torch.stack([snr, snr_gamma * torch.ones_like(snr)], dim=1).min(dim=1)[0]
        weights = weights / snr if ptype == "epsilon" else weights / (snr + 1)
        weights = hip.h2d(weights.float().contiguous(), pred.device)
    loss, _ = hip.mse_loss(pred.float(), target.float(), weights)
    return loss, pred, target
