"""Model / scheduler configurations of the BASELINE workloads (the subset of config.json the hot path reads).

SD1.5 values: runwayml/stable-diffusion-v1-5 `unet/config.json`, `vae/config.json`, `scheduler/scheduler_config.json`
(those files are external to the reference tree).  BrushNet: BrushNetModel.from_unet with
conditioning_channels = 6 (4 masked-image latents + mask + depth; examples/brushnet/train_brushnet_mirror.py:968-971).
"""
SD15_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0)
SD15_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
SD15_SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                  steps_offset=1, set_alpha_to_one=False, clip_sample=False, skip_prk_steps=True)

TINY_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
    cross_attention_dim=32, attention_head_dim=4, norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0)
TINY_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64), layers_per_block=1,
                norm_num_groups=32, scaling_factor=0.18215)

# SDXL (stabilityai/stable-diffusion-xl-base-1.0 unet/config.json) and a tiny configuration of the
# same architecture: linear proj_in/proj_out, per-level transformer depth and head count, text_time embedding
SDXL_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    cross_attention_dim=2048, attention_head_dim=(5, 10, 20), transformer_layers_per_block=(1, 2, 10),
    use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=256,
    projection_class_embeddings_input_dim=2816, norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0)
SDXL_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.13025)
TINY_XL_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(32, 64, 64), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    cross_attention_dim=48, attention_head_dim=(2, 4, 8), transformer_layers_per_block=(1, 2, 3),
    use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=8,
    projection_class_embeddings_input_dim=6 * 8 + 24, norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0)


def brushnet_config(unet_cfg: dict, conditioning_channels: int = 6) -> dict:
    n = len(unet_cfg["block_out_channels"])
    cfg = dict(unet_cfg)
    cfg.pop("out_channels", None)
    cfg.update(conditioning_channels=conditioning_channels, down_block_types=("DownBlock2D",) * n,
               up_block_types=("UpBlock2D",) * n, mid_block_type="MidBlock2D")
    return cfg
