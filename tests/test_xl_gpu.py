"""SDXL architecture on the HIP path: BrushNet-XL residuals, UNet-XL with injection and the
StableDiffusionXLBrushNetPipeline loop against golden outputs of the reference (tools/make_golden.py::tiny_xl)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mirrorfusion_ref as R  # noqa: E402
from reflecting_reality_amd import DDIMScheduler, StableDiffusionXLBrushNetPipeline, synth  # noqa: E402
from reflecting_reality_amd import models as M  # noqa: E402
from util import check, golden, keys, report  # noqa: E402

DEV = "cuda"
SD_SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                steps_offset=1, set_alpha_to_one=False)
_cache = {}


def build_xl(prec):
    if prec not in _cache:
        shapes = keys("tiny_xl")
        unet = M.UNet2DConditionModel(dict(R.TINY_XL_UNET), precision=prec, device=DEV)
        unet.load_state_dict(synth.state_dict_for(shapes["unet"], 20))
        bn = M.BrushNetModel(dict(R.brushnet_config(R.TINY_XL_UNET, 5)), precision=prec, device=DEV)
        bn.load_state_dict(synth.state_dict_for(shapes["brushnet"], 21))
        vae = M.AutoencoderKL(dict(R.TINY_VAE), precision=prec, device=DEV)
        vae.load_state_dict(synth.state_dict_for(shapes["vae"], 2))
        _cache[prec] = (unet, bn, vae)
    return _cache[prec]


@pytest.mark.parametrize("prec,atol", [("fp32", 2e-4), ("f16x3", 2e-4), ("bf16", None)])
def test_tiny_xl_models(prec, atol):
    unet, bn, _ = build_xl(prec)
    G = golden("tiny_xl.npz")
    g = torch.Generator().manual_seed(43)
    x = torch.randn(2, 4, 8, 8, generator=g)
    cond = torch.randn(2, 5, 8, 8, generator=g)
    ehs = torch.randn(2, 77, 48, generator=g)
    added = dict(text_embeds=torch.randn(2, 24, generator=g),
                 time_ids=torch.tensor([[16., 16., 0., 0., 16., 16.], [32., 24., 4., 2., 16., 16.]]))
    d, m, u = bn(x, 401, encoder_hidden_states=ehs, brushnet_cond=cond, conditioning_scale=0.9, added_cond_kwargs=added,
                 return_dict=False)
    for i, t in enumerate(d):
        check(f"xl bn_down_{i}[{prec}]", t, G[f"bn_down_{i}"], prec, dict(atol=atol), f"tiny_xl/bn_down_{i}")
    check(f"xl bn_mid[{prec}]", m, G["bn_mid"], prec, dict(atol=atol), "tiny_xl/bn_mid")
    for i, t in enumerate(u):
        check(f"xl bn_up_{i}[{prec}]", t, G[f"bn_up_{i}"], prec, dict(atol=atol), f"tiny_xl/bn_up_{i}")
    eps = unet(x, 401, ehs, added_cond_kwargs=added, down_block_add_samples=d, mid_block_add_sample=m,
               up_block_add_samples=u, return_dict=False)[0]
    check(f"xl unet eps[{prec}]", eps, G["unet_eps_inj"], prec, dict(atol=atol), "tiny_xl/unet_eps_inj")
    with pytest.raises(ValueError):        # text_time needs its inputs (unet_2d_condition.py:973-981)
        unet(x, 401, ehs, added_cond_kwargs={"text_embeds": added["text_embeds"]})


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-3), ("f16x3", 1e-3), ("bf16", None)])
def test_tiny_xl_pipeline(prec, tol):
    unet, bn, vae = build_xl(prec)
    G = golden("tiny_xl.npz")
    pipe = StableDiffusionXLBrushNetPipeline(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None,
                                             tokenizer_2=None, unet=unet, brushnet=bn,
                                             scheduler=DDIMScheduler(**SD_SCHED, clip_sample=False))
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(1, 16, 16, seed=99, cross_dim=48, vae_scale=2)
    gp = torch.Generator().manual_seed(100)
    pooled, npooled = torch.randn(1, 24, generator=gp), torch.randn(1, 24, generator=gp)
    noise = torch.from_numpy(G["pipe_vae_noise"])
    kw = dict(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
              pooled_prompt_embeds=pooled, negative_pooled_prompt_embeds=npooled, image=inp["image"], mask=inp["mask"],
              num_inference_steps=3, guidance_scale=5.0, output_type="latent", brushnet_conditioning_scale=1.0, height=16,
              width=16, original_size=(24, 20), crops_coords_top_left=(2, 1), target_size=(16, 16), conditioning_noise=noise)
    outs = []
    for graph in (True, False):            # 3 steps: the captured graph replays steps 1-2
        pipe.use_hip_graph, pipe._graph_state = graph, None
        outs.append(pipe(latents=inp["latents"].clone(), **kw).images.float().cpu())
    assert torch.equal(outs[0], outs[1])
    ref = torch.from_numpy(G["pipe_latents"])
    check(f"xl 3-step latents[{prec}]", outs[0], ref, prec, dict(atol=tol), "tiny_xl/pipe_latents")
    with pytest.raises(ValueError):        # wrong pooled width: _get_add_time_ids' consistency check
        pipe(latents=inp["latents"].clone(), **{**kw, "pooled_prompt_embeds": torch.randn(1, 16),
                                                 "negative_pooled_prompt_embeds": torch.randn(1, 16)})
