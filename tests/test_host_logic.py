"""Host-side logic that needs no GPU: scheduler tables / coefficients against the reference's golden vectors,
parameter enumeration against the reference's state-dict tables, input validation, image-processor semantics,
batch sharding."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mirrorfusion_ref as R
from reflecting_reality_amd import configs, distributed as D, synth
from reflecting_reality_amd.models import AutoencoderKL, BrushNetModel, UNet2DConditionModel
from reflecting_reality_amd.pipeline import StableDiffusionBrushNetPipeline, VaeImageProcessor
from reflecting_reality_amd.schedulers import DDIMScheduler, PNDMScheduler, UniPCMultistepScheduler
from util import golden, keys

SD = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
          steps_offset=1, set_alpha_to_one=False)


@pytest.mark.parametrize("size,ucfg,vcfg", [("tiny", configs.TINY_UNET, configs.TINY_VAE),
                                            ("sd15", configs.SD15_UNET, configs.SD15_VAE),
                                            ("tiny_xl", configs.TINY_XL_UNET, configs.TINY_VAE)])
def test_param_tables_match_reference_state_dicts(size, ucfg, vcfg):
    ref = keys(size)
    models = dict(unet=UNet2DConditionModel(dict(ucfg), device="cpu"),
                  brushnet=BrushNetModel(dict(configs.brushnet_config(ucfg, 5 if size == "tiny_xl" else 6)), device="cpu"),
                  vae=AutoencoderKL(dict(vcfg), device="cpu"))
    for name, m in models.items():
        mine = {k: tuple(v) for k, v in m.param_shapes().items()}
        assert mine == ref[name], f"{size}/{name}: parameter table differs from the reference state dict"
    if size == "sd15":   # BASELINE.md §2 parameter counts
        count = {n: sum(int(np.prod(s)) for s in m.param_shapes().values()) for n, m in models.items()}
        assert round(count["unet"] / 1e6, 1) == 859.5 and round(count["brushnet"] / 1e6, 1) == 618.8
        assert round(count["vae"] / 1e6, 1) == 83.7


def test_scheduler_tables_and_coefficients():
    G = golden("schedulers.npz")
    for n in (4, 50):
        d = DDIMScheduler(**SD, clip_sample=False)
        d.set_timesteps(n)
        assert d.timesteps.tolist() == G[f"ddim_timesteps_{n}"].tolist()
        p = PNDMScheduler(**SD, skip_prk_steps=True)
        p.set_timesteps(n)
        assert p.timesteps.tolist() == G[f"pndm_timesteps_{n}"].tolist()
        u = UniPCMultistepScheduler.from_config(p.config)     # examples/brushnet/test_brushnet.py:158
        u.set_timesteps(n)
        assert u.timesteps.tolist() == G[f"unipc_timesteps_{n}"].tolist()
        assert u.config["timestep_spacing"] == "linspace" and u.config["beta_schedule"] == "scaled_linear"
        assert len(u.sigmas) == n + 1
    assert np.allclose(d.alphas_cumprod.numpy(), G["alphas_cumprod"], rtol=0, atol=0)
    d = DDIMScheduler(**SD, clip_sample=False)
    d.set_timesteps(50)
    sa, sb, sp, dirc, std = d.step_coefficients(981)
    a_t, a_p = float(d.alphas_cumprod[981]), float(d.alphas_cumprod[961])
    assert abs(sa - a_t ** 0.5) < 1e-7 and abs(sb - (1 - a_t) ** 0.5) < 1e-7
    assert abs(sp - a_p ** 0.5) < 1e-7 and abs(dirc - (1 - a_p) ** 0.5) < 1e-7 and std == 0.0
    # last step uses final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one False)
    assert abs(d.step_coefficients(1)[2] - float(d.alphas_cumprod[0]) ** 0.5) < 1e-7
    # reference test_scheduler_ddim.py:110-120 (_get_variance) with its own config
    t = DDIMScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear")
    for (a, b), v in {(0, 0): 0.0, (420, 400): 0.14771, (980, 960): 0.32460, (487, 486): 0.00979, (999, 998): 0.02}.items():
        assert abs(float(t._get_variance(a, b)) - v) < 1e-5
    t.set_timesteps(5)
    t2 = DDIMScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, steps_offset=1)
    t2.set_timesteps(5)
    assert t2.timesteps.tolist() == [801, 601, 401, 201, 1]          # test_scheduler_ddim.py:63-67
    with pytest.raises(ValueError):
        DDIMScheduler().step_coefficients(10)                          # "run set_timesteps first" (:387-390)


def test_from_config_uses_target_defaults_for_unset_keys():
    """configuration_utils.py:458-459,649: a PNDM config carries no clip_sample -> DDIM's own default applies."""
    p = PNDMScheduler(**SD, skip_prk_steps=True)
    d = DDIMScheduler.from_config(p.config)
    assert d.config.steps_offset == 1 and d.config.beta_schedule == "scaled_linear" and d.config.clip_sample is True
    assert "skip_prk_steps" not in d.config


def test_image_processor_semantics():
    ip = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True)
    img = torch.rand(2, 3, 16, 16)
    assert torch.equal(ip.preprocess(img, 16, 16), 2.0 * img - 1.0)
    neg = torch.rand(1, 1, 16, 16) * 2 - 1                       # depth in [-1,1] passes through (image_processor.py:540-547)
    assert torch.equal(ip.preprocess(neg, 16, 16), neg)
    lat = torch.rand(1, 4, 8, 8)
    assert ip.preprocess(lat) is lat or torch.equal(ip.preprocess(lat), lat)   # 4 channels = latents (:532-533)
    # mask polarity (pipeline_brushnet.py:1139): white (1.0) -> 0 = hole, black -> 1 = keep
    m = torch.zeros(1, 3, 4, 4); m[:, :, :2] = 1.0
    pm = ip.preprocess(m, 4, 4)
    one_ch = (pm.sum(1)[:, None] < 0).float()
    assert one_ch[0, 0, 0, 0] == 0.0 and one_ch[0, 0, 3, 3] == 1.0
    arr = ip.postprocess(torch.zeros(1, 3, 4, 4), output_type="np")
    assert arr.shape == (1, 4, 4, 3) and np.allclose(arr, 0.5)
    with pytest.raises(ValueError):
        ip.preprocess("not an image")


def _pipe():
    mk = lambda cls, cfg: cls(dict(cfg), device="cpu")
    return StableDiffusionBrushNetPipeline(
        vae=mk(AutoencoderKL, configs.TINY_VAE), text_encoder=None, tokenizer=None,
        unet=mk(UNet2DConditionModel, configs.TINY_UNET),
        brushnet=mk(BrushNetModel, configs.brushnet_config(configs.TINY_UNET, 6)),
        scheduler=DDIMScheduler(**SD, clip_sample=False), safety_checker=None, feature_extractor=None,
        requires_safety_checker=False, depth_conditioning_mode="concat")


def test_memory_switches_of_the_reference_pipeline_are_accepted():
    """DiffusionPipeline's memory-saving switches (pipeline_utils.py:940-1683) never change a result: a script that calls
    them keeps working, the modules stay resident."""
    pipe = _pipe()
    for name in ("enable_model_cpu_offload", "enable_sequential_cpu_offload", "enable_attention_slicing", "disable_attention_slicing",
                 "enable_vae_slicing", "disable_vae_slicing", "enable_vae_tiling", "disable_vae_tiling",
                 "enable_xformers_memory_efficient_attention", "disable_xformers_memory_efficient_attention"):
        assert getattr(pipe, name)() is None
    pipe.set_progress_bar_config(disable=True)


def test_check_inputs_error_conventions():
    """pipeline_brushnet.py:573-693."""
    pipe = _pipe()
    img = torch.rand(1, 3, 16, 16)
    pe = torch.zeros(1, 77, 32)
    ok = dict(prompt=None, image=img, mask=img, callback_steps=None, prompt_embeds=pe, negative_prompt_embeds=pe,
              depth=img[:, :1])
    pipe.check_inputs(**ok)
    with pytest.raises(TypeError):
        pipe.check_inputs(**{**ok, "brushnet_conditioning_scale": 1})          # must be float (:649-650)
    with pytest.raises(ValueError):
        pipe.check_inputs(**{**ok, "prompt": "x"})                              # both prompt and embeds
    with pytest.raises(ValueError):
        pipe.check_inputs(**{**ok, "prompt_embeds": None})                      # neither
    with pytest.raises(ValueError):
        pipe.check_inputs(**{**ok, "negative_prompt_embeds": torch.zeros(1, 7, 32)})
    with pytest.raises(ValueError):
        pipe.check_inputs(**{**ok, "control_guidance_start": 0.6, "control_guidance_end": 0.5})
    with pytest.raises(ValueError):
        pipe.check_inputs(**{**ok, "callback_steps": 0})
    with pytest.raises(ValueError):
        pipe.check_inputs(**{**ok, "depth": None})
    with pytest.raises(ValueError):
        BrushNetModel(dict(configs.brushnet_config(configs.TINY_UNET, 6), brushnet_conditioning_channel_order="xyz"),
                      device="cpu").forward(img, 1, None, img)                  # brushnet.py:741 (after "no parameters")


def test_clip_skip_and_custom_timesteps_follow_the_reference():
    """clip_skip (pipeline_brushnet.py:352-370): the hidden state clip_skip layers before the last, through the text model's final
    LayerNorm.  Custom `timesteps` (retrieve_timesteps, :113-119): the reference's DDIM / PNDM / UniPC take none and it raises
    ValueError — so does this pipeline (before anything touches a device)."""
    import types
    pipe = _pipe()

    class Tok:
        model_max_length = 5

        def __call__(self, txt, **kw):
            return types.SimpleNamespace(input_ids=torch.arange(5)[None].repeat(len(txt), 1))

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(1))
            self.text_model = types.SimpleNamespace(final_layer_norm=lambda h: h * 10.0)

        def forward(self, ids, output_hidden_states=False):
            hs = tuple(torch.full((ids.shape[0], 5, 32), float(l)) for l in range(4))      # hidden states of layers 0 .. 3
            return (hs[-1], None, hs) if output_hidden_states else (hs[-1],)

    pipe.tokenizer, pipe.text_encoder = Tok(), Enc()
    pe, npe = pipe.encode_prompt(["a"], 1, True, None)
    assert float(pe[0, 0, 0]) == 3.0 and float(npe[0, 0, 0]) == 3.0                        # the last layer's output, as returned
    pe, _ = pipe.encode_prompt(["a"], 1, True, None, clip_skip=1)
    assert float(pe[0, 0, 0]) == 20.0                                                       # layer -(1 + 1) = 2, through the final norm
    pe, _ = pipe.encode_prompt(["a"], 1, False, None, clip_skip=2)
    assert float(pe[0, 0, 0]) == 10.0
    img = torch.rand(1, 3, 16, 16)
    with pytest.raises(ValueError, match="does not support custom"):
        pipe(prompt_embeds=torch.zeros(1, 77, 32), negative_prompt_embeds=torch.zeros(1, 77, 32), image=img, mask=img[:, :1],
             depth=img[:, :1], timesteps=[900, 500, 100], output_type="latent", height=16, width=16)


def test_models_fail_loudly_without_gpu_or_weights():
    u = UNet2DConditionModel(dict(configs.TINY_UNET), device="cpu")
    with pytest.raises(RuntimeError):
        u(torch.zeros(1, 4, 8, 8), 1, torch.zeros(1, 77, 32))                   # no parameters loaded
    with pytest.raises(RuntimeError):
        u.load_state_dict({})                                                    # strict key check
    with pytest.raises(NotImplementedError):
        UNet2DConditionModel(dict(configs.TINY_UNET, down_block_types=("AttnDownBlock2D", "DownBlock2D")), device="cpu")


def test_synth_is_key_seeded_and_order_independent():
    a = synth.fill("down_blocks.0.resnets.0.conv1.weight", (8, 4, 3, 3), 0)
    b = synth.fill("down_blocks.0.resnets.0.conv1.weight", (8, 4, 3, 3), 0)
    c = synth.fill("down_blocks.0.resnets.0.conv1.weight", (8, 4, 3, 3), 1)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(synth.fill("x.norm1.weight", (64,)).mean()) - 1.0) < 0.1
    assert float(synth.fill("brushnet_down_blocks.0.weight", (32, 32, 1, 1)).abs().max()) > 0   # zero-convs made live
    i1, i2 = synth.pipeline_inputs(2, 16, 16, vae_scale=2), synth.pipeline_inputs(2, 16, 16, vae_scale=2)
    assert all(torch.equal(i1[k], i2[k]) for k in i1)
    assert i1["vae_noise"].shape == (4, 4, 8, 8) and float(i1["mask"].mean()) == 0.25


def test_shard_range_matches_accelerate_split():
    # accelerate.PartialState.split_between_processes: first `n % world` ranks get one extra item, contiguous
    assert [D.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [D.shard_range(3, r, 8) for r in range(8)] == [(0, 1), (1, 2), (2, 3)] + [(3, 3)] * 5
    items = list(range(37))
    got = sum((D.shard(items, r, 8) for r in range(8)), [])
    assert got == items


def test_compat_diffusers_alias_exposes_the_hot_path_names():
    import importlib
    import os
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reflecting-reality_amd", "compat")
    saved = sys.modules.pop("diffusers", None)
    sys.path.insert(0, compat)
    try:
        d = importlib.import_module("diffusers")
        for name in ("BrushNetModel", "UNet2DConditionModel", "AutoencoderKL", "DDIMScheduler", "PNDMScheduler",
                     "UniPCMultistepScheduler", "StableDiffusionBrushNetPipeline"):
            assert hasattr(d, name)
        with pytest.raises(AttributeError):
            d.StableDiffusionXLPipeline
    finally:
        sys.path.remove(compat)
        sys.modules.pop("diffusers", None)
        if saved is not None:
            sys.modules["diffusers"] = saved


def test_pipeline_save_and_from_pretrained_roundtrip(tmp_path):
    """pipeline_utils.py save_pretrained / from_pretrained conventions on the reference's on-disk format."""
    import json
    from reflecting_reality_amd.schedulers import UniPCMultistepScheduler as U
    shapes = keys("tiny")
    mk = lambda klass, cfg, name, seed: klass(dict(cfg), precision="fp32", device="cpu").load_state_dict(
        synth.state_dict_for(shapes[name], seed))
    unet = mk(UNet2DConditionModel, configs.TINY_UNET, "unet", 0)
    bn = mk(BrushNetModel, configs.brushnet_config(configs.TINY_UNET, 6), "brushnet", 1)
    vae = mk(AutoencoderKL, configs.TINY_VAE, "vae", 2)
    sched = PNDMScheduler(**{k: v for k, v in configs.SD15_SCHED.items() if k != "clip_sample"})
    pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=bn,
                                           scheduler=sched, depth_conditioning_mode="concat")
    pipe.save_pretrained(str(tmp_path))
    idx = json.load(open(tmp_path / "model_index.json"))
    assert idx["unet"] == ["diffusers", "UNet2DConditionModel"] and idx["scheduler"] == ["diffusers", "PNDMScheduler"]
    for sub in ("unet", "brushnet", "vae"):
        assert (tmp_path / sub / "config.json").exists() and (tmp_path / sub / "diffusion_pytorch_model.safetensors").exists()
    # the scheduler config the way the SD1.5 hub repo ships it: no `timestep_spacing` key (it predates the option)
    (tmp_path / "scheduler" / "scheduler_config.json").write_text(json.dumps({
        "_class_name": "PNDMScheduler", "_diffusers_version": "0.6.0", "beta_end": 0.012, "beta_schedule": "scaled_linear",
        "beta_start": 0.00085, "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True,
        "steps_offset": 1, "trained_betas": None, "clip_sample": False}))
    # brushnet passed explicitly (test_brushnet.py:139-155), everything else from the directory
    bn2 = BrushNetModel.from_pretrained(str(tmp_path), subfolder="brushnet", torch_dtype=torch.float32, device="cpu")
    back = StableDiffusionBrushNetPipeline.from_pretrained(str(tmp_path), brushnet=bn2, torch_dtype=torch.float32,
                                                           device="cpu", safety_checker=None,
                                                           depth_conditioning_mode="concat", low_cpu_mem_usage=False)
    assert type(back.scheduler).__name__ == "PNDMScheduler" and back.scheduler.config["steps_offset"] == 1
    assert back.depth_conditioning_mode == "concat" and back.brushnet is bn2
    for a, b in ((unet, back.unet), (vae, back.vae), (bn, back.brushnet)):
        sa, sb = a.state_dict(), b.state_dict()
        assert sa.keys() == sb.keys() and all(torch.equal(sa[k], sb[k]) for k in sa)
    # the scheduler swap of test_brushnet.py:158 on the loaded pipeline
    back.scheduler = U.from_config(back.scheduler.config)
    back.scheduler.set_timesteps(50)
    assert back.scheduler.timesteps[:3].tolist() == [999, 979, 959]
    with pytest.raises(ValueError):             # a module that is neither passed nor on disk
        import shutil
        shutil.rmtree(tmp_path / "brushnet")
        StableDiffusionBrushNetPipeline.from_pretrained(str(tmp_path), device="cpu")
    # the training layout (flat fp32 arena in the kernels' layout, fused time_emb_proj matrix) exports the reference's
    # parameter tables unchanged — what the checkpoint hooks write
    for m, name in ((unet, "unet"), (bn, "brushnet")):
        before = m.state_dict()
        m.train()
        assert m.training and m.flat_g is not None and m.flat_w.numel() >= m.num_arena_floats() > 0
        after = m.state_dict()
        assert set(after) == set(shapes[name])
        for k in before:
            assert tuple(after[k].shape) == tuple(shapes[name][k]) and torch.equal(after[k], before[k]), k
        g = m.grad_state_dict()
        assert set(g) == set(after) and all(float(v.abs().max()) == 0.0 for v in g.values())
    with pytest.raises(NotImplementedError):
        vae.train()


def test_split_pack_layout_and_precision():
    """ops.split_pack: per block of 32 k the row holds [32 hi | 32 lo] 16-bit values, rows zero-padded to whole blocks;
    hi + lo reproduces the fp32 weight to 22 (fp16 halves) / 16 (bf16 halves) bits."""
    from reflecting_reality_amd import hip, ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(5, 72, generator=g) * 0.05
    for code, half, bits in ((hip.MF_F16X3, torch.float16, 21), (hip.MF_BF16X3, torch.bfloat16, 15)):
        packed, kp = ops.split_pack(w, code)
        assert kp == 96 and packed.shape == (5, 192) and packed.dtype == half
        blocks = packed.view(5, 3, 2, 32).float()
        rec = (blocks[:, :, 0] + blocks[:, :, 1]).reshape(5, 96)
        assert float(rec[:, 72:].abs().max()) == 0.0                          # zero padding
        assert float((rec[:, :72] - w).abs().max()) <= float(w.abs().max()) * 2.0 ** -bits
        assert torch.equal(blocks[:, :, 0].reshape(5, 96)[:, :72], w.to(half).float())
    cw = ops.ConvWeight(w[:, :64].reshape(5, 64, 1, 1).contiguous(), None, ops.Precision.get("f16x3"), "cpu")
    assert cw.w_split == 1 and cw.ldw == 64 and cw.w.shape == (5, 128)
    raw = ops.ConvWeight(w[:, :64].contiguous(), None, ops.Precision.get("f16x3"), "cpu", raw=True)
    assert raw.w_split == 0 and raw.w.dtype == torch.float32


def test_precision_names_and_the_reference_default_fp16():
    """torch_dtype=torch.float16 (the default of examples/brushnet/test_brushnet.py:122-126) selects the fp16 storage mode (round 5:
    MF_F16, the f16 MFMA forms) — never another precision silently; unknown names still raise."""
    from reflecting_reality_amd import hip, ops
    assert ops.Precision.get(torch.bfloat16).code == hip.MF_BF16 and ops.Precision.get("fp32").code == hip.MF_F32
    for alias in ("f16x3", "split", "parity"):
        p = ops.Precision.get(alias)
        assert p.split and p.code == hip.MF_F16X3 and p.act == torch.float32 and p.vec == 4
    for alias in ("fp16", "f16", torch.float16):
        p = ops.Precision.get(alias)
        assert p.name == "fp16" and p.code == hip.MF_F16 and p.act == p.compute == torch.float16 and p.vec == 8 and p.half and not p.split
    assert UNet2DConditionModel(dict(configs.TINY_UNET), precision=torch.float16, device="cpu").prec.name == "fp16"
    with pytest.raises(ValueError, match="unsupported precision"):
        ops.Precision.get(torch.float64)


def test_randn_tensor_follows_the_generator_device():
    """utils/torch_utils.py randn_tensor semantics: CPU generator -> host draw; list -> one draw per sample."""
    from reflecting_reality_amd.rng import randn_tensor
    a = randn_tensor((2, 4, 8, 8), torch.Generator().manual_seed(5))
    assert torch.equal(a, torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(5)))
    gs = [torch.Generator().manual_seed(s) for s in (1, 2)]
    b = randn_tensor((2, 4, 8, 8), gs)
    assert torch.equal(b[1], torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(2))[0])
    with pytest.raises(ValueError, match="list of generators"):
        randn_tensor((3, 4, 8, 8), gs)


def test_tune_cache_versioning_and_atomic_save(tmp_path, monkeypatch):
    from reflecting_reality_amd import hip
    ver = hip.load().mf_gemm_tile_table_version()
    path = tmp_path / "sub" / "tune.json"
    monkeypatch.setenv("MFHIP_TUNE_CACHE", str(path))
    monkeypatch.setattr(hip, "_tune", None)
    monkeypatch.setattr(hip, "_tune_new", {})
    shipped = hip._tune_read(hip._TUNE_PATH, ver)
    assert len(shipped) > 100, "the shipped cache must be readable under the library's tile-table version"
    assert hip._tune_read(hip._TUNE_PATH, ver + 1) == {}                 # another numbering: dropped, not trusted
    hip._tune_load()
    hip._tune_new["1,1,64,64,64,1,1,0,0,1,0,0,1,1"] = (3, 1)
    hip.tune_save()
    with open(path) as f:
        saved = json.load(f)
    assert saved["_meta"] == {"tile_table": ver} and saved["entries"] == {"1,1,64,64,64,1,1,0,0,1,0,0,1,1": [3, 1]}
    assert [p.name for p in path.parent.iterdir()] == ["tune.json"]        # no temp file left behind
    path.write_text("{ truncated")                                          # a corrupt user file is ignored
    assert hip._tune_read(str(path), ver) == {}
    hip._tune_forget("1,1,64,64,64,1,1,0,0,1,0,0,1,1")
    assert "1,1,64,64,64,1,1,0,0,1,0,0,1,1" not in hip._tune_new


def test_sharded_inference_harness_with_a_stub_pipeline(tmp_path):
    """inference.run_sharded (test_brushnet.py:163-168,247-266): contiguous split of the sample list, ONE generator per
    rank drawn from in sequence, `num_images_per_validation` images per sample; list_checkpoints orders checkpoint-N."""
    from reflecting_reality_amd import inference as I

    class Out:
        def __init__(self, x):
            self.images = [x]

    class StubPipe:
        device = "cpu"
        calls = []

        def __call__(self, generator=None, tag=None, **kw):
            assert kw["num_inference_steps"] == 7 and isinstance(kw["brushnet_conditioning_scale"], float)
            v = float(torch.randn(1, generator=generator))
            self.calls.append((tag, v))
            return Out((tag, v))

    samples = [dict(tag=i) for i in range(7)]
    got = {}
    for rank in range(3):
        pipe = StubPipe()
        res = I.run_sharded(pipe, samples, seed=11, num_images_per_validation=2, num_inference_steps=7, rank=rank, world=3)
        got.update(res)
        g = torch.Generator().manual_seed(11)
        want = [float(torch.randn(1, generator=g)) for _ in range(2 * len(res))]
        assert [v for imgs in res.values() for _, v in imgs] == want          # one seeded stream per rank, in order
    assert sorted(got) == list(range(7)) and all(len(v) == 2 and v[0][0] == i for i, v in got.items())
    assert [sorted(I.run_sharded(StubPipe(), samples, num_images_per_validation=1, num_inference_steps=7, rank=r, world=3))
            for r in range(3)] == [[0, 1, 2], [3, 4], [5, 6]]
    for n in (500, 1000, 90, 1500):
        (tmp_path / f"checkpoint-{n}").mkdir()
    (tmp_path / "logs").mkdir()
    assert [os.path.basename(p) for p in I.list_checkpoints(str(tmp_path))] == ["checkpoint-90", "checkpoint-500", "checkpoint-1000", "checkpoint-1500"]
    assert [os.path.basename(p) for p in I.list_checkpoints(str(tmp_path), 500)] == ["checkpoint-500", "checkpoint-1000", "checkpoint-1500"]


def test_set_attn_processor_surface():
    from reflecting_reality_amd import MfhipAttnProcessor
    unet = UNet2DConditionModel(dict(configs.TINY_UNET), precision="fp32", device="cpu")
    unet.load_state_dict(synth.state_dict_for(keys("tiny")["unet"], 0))
    procs = unet.attn_processors
    assert len(procs) == 2 * sum(1 for k in keys("tiny")["unet"] if k.endswith("attn1.to_q.weight"))
    assert all(k.endswith(".processor") and ".transformer_blocks." in k for k in procs)
    unet.set_attn_processor(MfhipAttnProcessor())
    unet.set_attn_processor(procs)
    with pytest.raises(ValueError, match="number of processors"):
        unet.set_attn_processor({"a": MfhipAttnProcessor()})
    with pytest.raises(NotImplementedError):
        unet.set_attn_processor(object())


def _frontend_inputs():
    """The seeded inputs of tools/make_golden.py::frontend_inputs (kept in step with it; the PIL pixels come from the fixture)."""
    g = torch.Generator().manual_seed(515)
    t01 = torch.rand(2, 3, 40, 56, generator=g)
    tneg = torch.rand(2, 3, 40, 56, generator=g) * 2.0 - 1.0
    mask = (torch.rand(2, 3, 40, 56, generator=g) > 0.6).float()
    arr = torch.rand(40, 56, 3, generator=g).numpy()
    _u8 = (torch.rand(40, 56, 3, generator=g) * 255).to(torch.uint8).numpy()
    post = torch.rand(2, 3, 24, 32, generator=g) * 2.4 - 1.2
    return dict(t01=t01, tneg=tneg, mask=mask, arr=arr, post=post)


def test_image_processor_host_path_matches_the_reference_vae_image_processor():
    """The package's VaeImageProcessor on host inputs (torch / numpy / PIL, with and without resize, default sizes) and its
    postprocess against the outputs of the REFERENCE's VaeImageProcessor (image_processor.py:446-610) recorded in
    tests/golden/frontend.npz by tools/make_golden.py::frontend."""
    import warnings
    import PIL.Image
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend.npz"))
    I = _frontend_inputs()
    ip = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name in ("t01", "tneg", "mask"):
            assert np.array_equal(ip.preprocess(I[name], height=40, width=56).numpy(), G[f"{name}_same"]), name
            assert np.array_equal(ip.preprocess(I[name], height=32, width=48).numpy(), G[f"{name}_resized"]), name
        assert np.array_equal(ip.preprocess(I["arr"], height=32, width=48).numpy(), G["np_resized"])
        assert np.array_equal(ip.preprocess([I["arr"], I["arr"][::-1].copy()], height=40, width=56).numpy(), G["np_list"])
        pil = PIL.Image.fromarray(G["pil_u8"])
        assert np.array_equal(ip.preprocess(pil, height=40, width=56).numpy(), G["pil_same"])
        assert np.array_equal(ip.preprocess(pil, height=32, width=48).numpy(), G["pil_resized"])
    assert np.array_equal(ip.postprocess(I["post"], output_type="pt", do_denormalize=[True, True]).numpy(), G["post_pt"])
    assert np.array_equal(ip.postprocess(I["post"], output_type="np", do_denormalize=[True, True]), G["post_np"])
    assert np.array_equal(np.stack([np.array(im) for im in ip.postprocess(I["post"], output_type="pil", do_denormalize=[True, True])]),
                          G["post_pil"])
    assert np.array_equal(ip.postprocess(I["post"], output_type="pt", do_denormalize=[True, False]).numpy(), G["post_pt_mixed"])


def test_lr_schedules_match_the_reference_get_scheduler():
    """optimization.get_scheduler (train_brushnet_mirror.py:1257-1263) against the learning rates the REFERENCE's
    diffusers.optimization.get_scheduler produced for every schedule name (tests/golden/lr_schedules.json,
    tools/make_golden_r04.py --lr-schedules): the value after construction and after each step, exactly."""
    import json
    from reflecting_reality_amd import optimization as O

    class Opt:
        def __init__(self, lr):
            self.lr = lr

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_schedules.json")) as f:
        g = json.load(f)
    for name, kw in g["_cases"].items():
        o = Opt(g["_lr"])
        sch = O.get_scheduler(name, o, **kw)
        seq = [sch.get_last_lr()[0]]
        for _ in range(g["_steps"]):
            sch.step()
            seq.append(o.lr)
        assert seq == g[name], name
    with pytest.raises(ValueError):
        O.get_scheduler("linear", Opt(1e-5), num_warmup_steps=3)               # needs num_training_steps
    with pytest.raises(ValueError):
        O.get_scheduler("cosine", Opt(1e-5))
    # a resumed schedule (state_dict round trip) continues where it stopped
    o = Opt(1e-5)
    a = O.get_scheduler("cosine", o, num_warmup_steps=2, num_training_steps=10)
    for _ in range(4):
        a.step()
    o2 = Opt(1e-5)
    b = O.get_scheduler("cosine", o2, num_warmup_steps=2, num_training_steps=10)
    b.load_state_dict(a.state_dict())
    a.step(); b.step()
    assert o.lr == o2.lr


def test_checkpoints_carry_the_lr_schedule(tmp_path):
    """accelerator.save_state / load_state persist the prepared lr_scheduler (scheduler.bin); without it a resumed
    run replays its warm-up.  The schedule-side of training.save_state / load_state on a stub model (no GPU): a run interrupted
    after 5 of 12 steps and resumed from its checkpoint follows the uninterrupted run's learning rates exactly, and AdamW's
    state carries `initial_lr` (its `lr` alone is the decayed value)."""
    from reflecting_reality_amd import optimization as O, training as T

    class StubModel:
        def get_trainable_modules(self):
            return []

    class StubOpt:            # AdamW's scheduler-facing surface (lr / initial_lr / state_dict), no arenas
        def __init__(self, lr):
            self.lr = lr
        state_dict = T.AdamW.state_dict
        step_count, betas, weight_decay, eps, exp_avg, exp_avg_sq, models = 0, (0.9, 0.999), 1e-2, 1e-8, [], [], []
        def load_state_dict(self, sd):
            self.lr, self.initial_lr = sd["lr"], sd["initial_lr"]

    def make():
        o = StubOpt(3e-4)
        return o, O.get_scheduler("cosine", o, num_warmup_steps=3, num_training_steps=12)

    o_ref, s_ref = make()
    ref = []
    for _ in range(12):
        s_ref.step()
        ref.append(o_ref.lr)
    o1, s1 = make()
    for _ in range(5):
        s1.step()
    sd = o1.state_dict()
    assert sd["initial_lr"] == 3e-4 and sd["lr"] == o1.lr != 3e-4
    path = T.save_state(str(tmp_path), 5, StubModel(), o1, lr_scheduler=s1)
    assert os.path.exists(os.path.join(path, "scheduler.bin"))
    o2, s2 = make()                                    # the resumed process builds its schedule first, then loads (script order)
    assert T.load_state(path, StubModel(), o2, lr_scheduler=s2) == 5
    got = []
    for _ in range(7):
        s2.step()
        got.append(o2.lr)
    assert got == ref[5:]
    # a schedule built AFTER the load with last_epoch continues too: initial_lr survived in the optimizer state
    o3 = StubOpt(1.0)
    o3.load_state_dict(sd)
    s3 = O.get_scheduler("cosine", o3, num_warmup_steps=3, num_training_steps=12, last_epoch=4)
    s3.step()
    assert o3.lr == ref[5]
    # a checkpoint without scheduler.bin warns instead of silently restarting the schedule
    p2 = T.save_state(str(tmp_path), 6, StubModel(), o1)
    with pytest.warns(UserWarning, match="no scheduler.bin"):
        T.load_state(p2, StubModel(), None, lr_scheduler=make()[1])


def test_gn_slab_predicate_mirrors_the_library():
    """hip.gn_slab_applies decides on the host whether a conv may leave its split-K reduce to the GroupNorm: it must agree with the
    dispatch of gn_slab_kernel in csrc/norm.hip for the channel counts of the path, and refuse what that form cannot take."""
    from reflecting_reality_amd import hip
    for c in (320, 640, 960, 1280, 1920, 2560):
        assert hip.gn_slab_applies(64, c, 32) and hip.gn_slab_applies(256, c, 32), c
        assert not hip.gn_slab_applies(1024, c, 32)
    assert not hip.gn_slab_applies(64, 3200, 32)       # 100 channels per group: the slab needs 25 x 64 threads
    assert not hip.gn_slab_applies(64, 324, 32) and not hip.gn_slab_applies(64, 320, 0)
