"""fp8 (OCP e4m3) linear layers of the SDXL transformer blocks (BASELINE.json configs[4]): the
quantiser, the fp8 GEMM on v_mfma_scale_f32_32x32x64_f8f6f4, and the tiny-XL models / pipeline in precision "fp8"."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from reflecting_reality_amd import hip, ops  # noqa: E402

DEV = "cuda"


def deq(q, s):
    return q.float().cpu().double() * s.cpu().double().view(*q.shape[:-1], 1)


@pytest.mark.parametrize("rows,c", [(300, 320), (64, 1280), (5, 2048), (4096, 640), (9, 8), (130, 5120), (7, 8192)])
@pytest.mark.parametrize("ln", [False, True])
def test_quantize_rows_fp8(rows, c, ln):
    g = torch.Generator().manual_seed(61)
    x = torch.randn(rows, c, generator=g) * torch.rand(rows, 1, generator=g) * 4.0
    x[0] = 0.0                                                            # an all-zero row: scale 1, zeros out
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    for dt in (torch.float32, torch.bfloat16):
        xin = x.to(dt)
        y = F.layer_norm(xin.float(), (c,), gamma, beta, 1e-5) if ln else xin.float()
        q, s = hip.quantize_rows_fp8(xin.to(DEV), (gamma.to(DEV), beta.to(DEV)) if ln else None, 1e-5)
        assert q.dtype == hip.FP8 and s.shape == (rows,)
        amax = y.abs().amax(1)
        want_s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        assert torch.allclose(s.cpu(), want_s, rtol=2e-5 if ln else 1e-6, atol=0)
        err = (deq(q, s) - y.double()).abs()
        # e4m3: 3 mantissa bits -> half-ulp 2^-4 relative for normals; below 2^-6 * scale the spacing is 2^-9 * scale
        bound = y.double().abs() * 2.0 ** -4 + want_s.double().view(-1, 1) * 2.0 ** -10 + (1e-4 if ln else 0.0)
        assert bool((err <= bound).all()), float((err - bound).max())
        assert float(q.float().abs().max()) <= 448.0


@pytest.mark.parametrize("m,k,n", [(300, 320, 72), (2048, 640, 640), (64, 1280, 10240), (4096, 2048, 320), (77, 48, 32)])
def test_fp8_linear_matches_its_quantised_operands(m, k, n):
    """The fp8 GEMM is arithmetic on the quantised operands: compared with a float64 product of the DEQUANTISED inputs
    (the block-scaled matrix instruction accumulates its 64 products per step with a few bits less than a chain of fp32
    FMAs: measured 4e-5 ... 7e-5 of the result's rms, bound 2e-4); every fp8 tile, ragged M / N, bias + residual + alpha
    epilogue, fused GEGLU, V^T form."""
    prec = ops.Precision.get("fp8")
    g = torch.Generator().manual_seed(62)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    lw = ops.ConvWeight(w, b, prec, DEV, fp8=True)
    assert lw.fp8 and lw.w.dtype == hip.FP8 and lw.w_scale.shape == (n,)
    xq, xs = hip.quantize_rows_fp8(x.to(DEV))
    ref = deq(xq, xs) @ deq(lw.w, lw.w_scale).T + b.double()
    scale = float(ref.pow(2).mean().sqrt())
    for tile in (0, 1, 2, 3, 6, 7, 13, 14, 15):
        y = ops.linear((xq, xs), lw, out_dtype=torch.float32, tile=tile)
        e = float((y.double().cpu() - ref).abs().max()) / scale
        assert e < 2e-4, (tile, e)
    for bad_tile in (20, 4):          # a dx-reuse tile (3x3 convs only) and a tile fp8 is not instantiated for: refused
        with pytest.raises(hip.MfhipError, match="does not apply|not instantiated"):
            ops.linear((xq, xs), lw, out_dtype=torch.float32, tile=bad_tile)
    r0 = torch.randn(m, n, generator=g)
    y = ops.linear(x.to(DEV), lw, res0=r0.to(DEV), alpha=0.5, out_dtype=torch.float32)          # quantises x itself
    assert float((y.double().cpu() - (0.5 * ref + r0.double())).abs().max()) / scale < 2e-4
    # against the unquantised product: the fp8 error itself, ~2^-4 / sqrt(3) per operand averaged over K
    full = x.double() @ w.double().T + b.double()
    rel = float((ref - full).pow(2).mean().sqrt() / full.pow(2).mean().sqrt())
    print(f"fp8 linear {m}x{k}x{n}: rms error vs the unquantised product {rel:.3e}")
    assert rel < 0.06
    if n % 8 == 0:
        wg = ops.geglu_weight(w, b, prec, DEV, fp8=True)
        h, gate = (deq(xq, xs) @ deq(lw.w, lw.w_scale).T + b.double()).chunk(2, -1)
        yg = ops.linear_geglu((xq, xs), wg)
        assert yg.shape == (m, n // 2)
        assert float((yg.double().cpu() - h * F.gelu(gate)).abs().max()) < 3e-2 * float((h * F.gelu(gate)).abs().max())
    x3 = torch.randn(2, 77, k, generator=g)
    q3, s3 = hip.quantize_rows_fp8(x3.to(DEV))
    vt = ops.linear_t((q3, s3.view(2, 77)), ops.ConvWeight(w, b, prec, DEV, fp8=True), 80)
    ref_t = (deq(q3, s3.view(2, 77)) @ deq(lw.w, lw.w_scale).T + b.double()).transpose(1, 2)
    assert vt.shape == (2, n, 80) and float(vt[:, :, 77:].abs().max()) == 0.0
    assert float((vt[:, :, :77].double().cpu() - ref_t).abs().max()) / scale < 2e-2       # bf16 output rounding


def test_tiny_xl_models_and_pipeline_in_fp8():
    """precision "fp8" on the tiny SDXL configuration: BrushNet-XL has no transformer (identical to bf16), the UNet-XL's
    transformer Linears run in fp8.  Budget: fp8 e4m3 rounds operands to 2^-4 where bf16 rounds to 2^-9, but only the
    transformer Linears are affected and dot products average the rounding (over only 32 ... 128 terms in this tiny
    configuration); measured on this case the noise prediction and the 3-step latents move ~5x as far from the reference's
    fp32 result as the reference's own bf16 run does — asserted at 8x (mean) / 10x (max)."""
    from oracle import mirrorfusion_ref as R
    from reflecting_reality_amd import DDIMScheduler, StableDiffusionXLBrushNetPipeline, synth
    from reflecting_reality_amd import models as M
    from util import envelope, golden, keys
    shapes = keys("tiny_xl")
    unet = M.UNet2DConditionModel(dict(R.TINY_XL_UNET), precision="fp8", device=DEV)
    unet.load_state_dict(synth.state_dict_for(shapes["unet"], 20))
    n8 = sum(1 for v in unet.P.values() if getattr(v, "fp8", False))
    assert n8 > 10, "no fp8 layers were built"
    bn = M.BrushNetModel(dict(R.brushnet_config(R.TINY_XL_UNET, 5)), precision="fp8", device=DEV)
    bn.load_state_dict(synth.state_dict_for(shapes["brushnet"], 21))
    vae = M.AutoencoderKL(dict(R.TINY_VAE), precision="bf16", device=DEV)
    vae.load_state_dict(synth.state_dict_for(shapes["vae"], 2))
    G = golden("tiny_xl.npz")
    g = torch.Generator().manual_seed(43)
    x = torch.randn(2, 4, 8, 8, generator=g)
    cond = torch.randn(2, 5, 8, 8, generator=g)
    ehs = torch.randn(2, 77, 48, generator=g)
    added = dict(text_embeds=torch.randn(2, 24, generator=g),
                 time_ids=torch.tensor([[16., 16., 0., 0., 16., 16.], [32., 24., 4., 2., 16., 16.]]))
    d, m, u = bn(x, 401, encoder_hidden_states=ehs, brushnet_cond=cond, conditioning_scale=0.9, added_cond_kwargs=added, return_dict=False)
    eps = unet(x, 401, ehs, added_cond_kwargs=added, down_block_add_samples=d, mid_block_add_sample=m, up_block_add_samples=u,
               return_dict=False)[0]
    ref = torch.from_numpy(G["unet_eps_inj"])
    err = (eps.float().cpu() - ref).abs()
    env = envelope("tiny_xl/unet_eps_inj")
    print(f"tiny-XL eps in fp8: max {float(err.max()):.3e} mean {float(err.mean()):.3e} (reference bf16: {env['linf']:.3e} / {env['mean']:.3e})")
    assert float(err.mean()) <= 8.0 * env["mean"] and float(err.max()) <= 10.0 * env["linf"]
    pipe = StableDiffusionXLBrushNetPipeline(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None,
                                             unet=unet, brushnet=bn, scheduler=DDIMScheduler(
                                                 num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                                                 beta_schedule="scaled_linear", steps_offset=1, set_alpha_to_one=False, clip_sample=False))
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(1, 16, 16, seed=99, cross_dim=48, vae_scale=2)
    gp = torch.Generator().manual_seed(100)
    pooled, npooled = torch.randn(1, 24, generator=gp), torch.randn(1, 24, generator=gp)
    lat = pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], pooled_prompt_embeds=pooled,
               negative_pooled_prompt_embeds=npooled, image=inp["image"], mask=inp["mask"], num_inference_steps=3, guidance_scale=5.0,
               output_type="latent", brushnet_conditioning_scale=1.0, height=16, width=16, original_size=(24, 20),
               crops_coords_top_left=(2, 1), target_size=(16, 16), conditioning_noise=torch.from_numpy(G["pipe_vae_noise"]),
               latents=inp["latents"].clone()).images.float().cpu()
    e2 = (lat - torch.from_numpy(G["pipe_latents"])).abs()
    env2 = envelope("tiny_xl/pipe_latents")
    print(f"tiny-XL 3-step latents in fp8: max {float(e2.max()):.3e} mean {float(e2.mean()):.3e} (reference bf16: {env2['linf']:.3e} / {env2['mean']:.3e})")
    assert float(e2.mean()) <= 8.0 * env2["mean"] and float(e2.max()) <= 10.0 * env2["linf"]
