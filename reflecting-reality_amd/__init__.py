"""reflecting-reality_amd — MI355X-native MirrorFusion (SD1.5 + BrushNet) denoising hot path.

The directory name carries a hyphen (it is the project name), so the package is imported under the
module name ``reflecting_reality_amd`` through the shim ``reflecting_reality_amd.py`` at the repo root.
Sub-modules:
  hip        ctypes binding of libmfhip.so (include/mfhip.h) — the only compute backend, no fallback
  ops        layer-level operators (conv2d / linear / attention / norms) over NHWC tensors
  models     BrushNetModel, UNet2DConditionModel, AutoencoderKL with the reference call surface
  schedulers DDIMScheduler, PNDMScheduler, UniPCMultistepScheduler, DDPMScheduler (forward process only)
  training   MirrorFusionModel, compute_snr, training_loss (forward + loss; no backward)
  pipeline   StableDiffusionBrushNetPipeline
"""
__version__ = "0.1.0"

_LAZY = {
    "BrushNetModel": "models", "UNet2DConditionModel": "models", "AutoencoderKL": "models",
    "BrushNetOutput": "models", "DDIMScheduler": "schedulers", "PNDMScheduler": "schedulers", "UniPCMultistepScheduler": "schedulers", "DDPMScheduler": "schedulers",
    "MirrorFusionModel": "training", "compute_snr": "training", "training_loss": "training",
    "MfhipAttnProcessor": "attn_processor", "StableDiffusionBrushNetPipeline": "pipeline", "StableDiffusionXLBrushNetPipeline": "pipeline", "StableDiffusionPipelineOutput": "pipeline",
    "VaeImageProcessor": "pipeline", "Precision": "ops",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f"{__name__}.{_LAZY[name]}"), name)
    raise AttributeError(name)
