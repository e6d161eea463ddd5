"""reflecting-reality_amd — MI355X-native MirrorFusion (SD1.5 + BrushNet) denoising hot path.

The directory name carries a hyphen (it is the project name), so the package is imported under the
module name ``reflecting_reality_amd`` through the shim ``reflecting_reality_amd.py`` at the repo root.
Sub-modules:
  hip        ctypes binding of libmfhip.so (include/mfhip.h) — the only compute backend, no fallback
  ops        layer-level operators (conv2d / linear / attention / norms) over NHWC tensors
  models     BrushNetModel, UNet2DConditionModel, AutoencoderKL with the reference call surface
  schedulers DDIMScheduler, PNDMScheduler, UniPCMultistepScheduler, DDPMScheduler (forward process only)
  autograd   the tape of hand-written vector-Jacobian products behind the training step
  training   MirrorFusionModel, training_loss, train_step (backward + clip + AdamW), checkpoint save / load hooks
  distributed  batch sharding for inference, bucketed gradient all-reduce (RCCL) for training
  inference  run_sharded: the examples/brushnet/test_brushnet.py harness (sample list split over ranks, N seeds each)
  pipeline   StableDiffusionBrushNetPipeline, StableDiffusionXLBrushNetPipeline
"""
__version__ = "0.1.0"

_LAZY = {
    "BrushNetModel": "models", "UNet2DConditionModel": "models", "AutoencoderKL": "models",
    "BrushNetOutput": "models", "DDIMScheduler": "schedulers", "PNDMScheduler": "schedulers", "UniPCMultistepScheduler": "schedulers", "DDPMScheduler": "schedulers",
    "MirrorFusionModel": "training", "compute_snr": "training", "training_loss": "training", "train_step": "training",
    "AdamW": "training", "save_state": "training", "load_state": "training", "run_sharded": "inference",
    "MfhipAttnProcessor": "attn_processor", "StableDiffusionBrushNetPipeline": "pipeline", "StableDiffusionXLBrushNetPipeline": "pipeline", "StableDiffusionPipelineOutput": "pipeline",
    "VaeImageProcessor": "pipeline", "Precision": "ops",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f"{__name__}.{_LAZY[name]}"), name)
    raise AttributeError(name)
