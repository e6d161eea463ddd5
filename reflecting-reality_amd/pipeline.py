"""StableDiffusionBrushNetPipeline with the reference call surface on the HIP models.

Reference: MirrorFusion/src/diffusers/pipelines/brushnet/pipeline_brushnet.py — __init__ :185-233,
check_inputs :573-693, prepare_image :741-774, prepare_latents :777-791, __call__ :848-1363 (hot loop
:1250-1332), and image_processor.py:446-610 (VaeImageProcessor).

Result-preserving shortcuts relative to the reference:
  * the masked image is VAE-encoded once for B images; the reference encodes the CFG-duplicated 2B batch
    (:771-772, :1188) whose two halves have identical moments and differ only in the posterior noise — the
    noise is still drawn for 2B so both halves match the reference;
  * classifier-free guidance and the DDIM update run as one kernel.
CLIP text encoding is outside the accelerated path: pass `prompt_embeds` / `negative_prompt_embeds`, or give
the pipeline a transformers `text_encoder` + `tokenizer` pair.
"""
from __future__ import annotations

import os
import time

from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch

from . import hip
from .models import AutoencoderKL, BrushNetModel, UNet2DConditionModel
from .rng import randn_tensor
from .schedulers import DDIMScheduler

try:
    import PIL.Image
except Exception:  # pragma: no cover
    PIL = None


@dataclass
class StableDiffusionPipelineOutput:
    images: Any
    nsfw_content_detected: Optional[List[bool]]


class VaeImageProcessor:
    """image_processor.py:446-610 for the formats the BrushNet pipeline feeds it (host side, once per image)."""

    def __init__(self, vae_scale_factor: int = 8, do_resize: bool = True, do_normalize: bool = True,
                 do_convert_rgb: bool = False):
        self.vae_scale_factor = vae_scale_factor
        self.do_resize, self.do_normalize, self.do_convert_rgb = do_resize, do_normalize, do_convert_rgb

    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        supported = (torch.Tensor, np.ndarray) + ((PIL.Image.Image,) if PIL is not None else ())
        if isinstance(image, supported):
            image = [image]
        elif not (isinstance(image, list) and all(isinstance(i, supported) for i in image)):
            raise ValueError(f"Input is in incorrect format: {[type(i) for i in image]}. Currently, we only support "
                             "PIL.Image.Image, np.ndarray, torch.Tensor")
        first = image[0]
        if PIL is not None and isinstance(first, PIL.Image.Image):
            if self.do_resize and height is not None:
                image = [i.resize((width, height), resample=PIL.Image.LANCZOS) for i in image]
            if self.do_convert_rgb:
                image = [i.convert("RGB") for i in image]
            arr = np.stack([np.array(i).astype(np.float32) / 255.0 for i in image], axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            t = torch.from_numpy(arr.transpose(0, 3, 1, 2))
        elif isinstance(first, np.ndarray):
            arr = np.concatenate(image, axis=0) if first.ndim == 4 else np.stack(image, axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            t = torch.from_numpy(arr.transpose(0, 3, 1, 2)).float()
        else:
            t = torch.cat(image, dim=0) if first.ndim == 4 else torch.stack(image, dim=0)
            if t.shape[1] == 4:                                             # already latents (:532-533)
                return t
        t = t.float()
        if t.is_cuda:
            # device tensors stay on the device and go through the front-end kernels (csrc/frontend.hip): nearest resize
            # (F.interpolate's default, :resize()), then 2x - 1 unless the tensor already holds negatives — the min is
            # reduced on the device and read by the normalising kernel, the host never waits for it
            if self.do_resize and height is not None and tuple(t.shape[-2:]) != (height, width):
                t = hip.nearest_resize(t, height, width)
            return hip.image_normalize(t) if self.do_normalize else t
        if self.do_resize and height is not None and tuple(t.shape[-2:]) != (height, width):
            t = torch.nn.functional.interpolate(t, size=(height, width))    # :resize() for tensors (host, once)
        if self.do_normalize and t.min() >= 0:                              # negatives pass through un-normalised
            t = 2.0 * t - 1.0
        return t

    def postprocess(self, image: torch.Tensor, output_type: str = "pil", do_denormalize=None):
        if output_type not in ("latent", "pt", "np", "pil"):
            output_type = "np"
        if output_type == "latent":
            return image
        if do_denormalize is None:
            do_denormalize = [self.do_normalize] * image.shape[0]
        if image.is_cuda and image.dtype == torch.float32 and (all(do_denormalize) or not any(do_denormalize)):
            if output_type == "pil":       # straight to uint8 HWC on the device: the host only copies bytes
                arr = hip.postprocess(image, denormalize=all(do_denormalize), uint8=True).cpu().numpy()
                if arr.shape[-1] == 1:
                    return [PIL.Image.fromarray(a.squeeze(), mode="L") for a in arr]
                return [PIL.Image.fromarray(a) for a in arr]
            image = hip.postprocess(image, denormalize=all(do_denormalize))
        else:
            image = torch.stack([(image[i] / 2 + 0.5).clamp(0, 1) if do_denormalize[i] else image[i]
                                 for i in range(image.shape[0])])
        if output_type == "pt":
            return image
        arr = image.cpu().permute(0, 2, 3, 1).float().numpy()
        if output_type == "np":
            return arr
        arr = (arr * 255).round().astype("uint8")
        if arr.shape[-1] == 1:
            return [PIL.Image.fromarray(a.squeeze(), mode="L") for a in arr]
        return [PIL.Image.fromarray(a) for a in arr]


class StableDiffusionBrushNetPipeline:
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]

    def __init__(self, vae: AutoencoderKL, text_encoder, tokenizer, unet: UNet2DConditionModel,
                 brushnet: BrushNetModel, scheduler, safety_checker=None, feature_extractor=None, image_encoder=None,
                 requires_safety_checker: bool = True, depth_conditioning_mode=None, normals_conditioning_mode=None):
        if safety_checker is not None and feature_extractor is None:
            raise ValueError("Make sure to define a feature extractor when loading the pipeline if you want to use "
                             "the safety checker. If you do not want to use the safety checker, you can pass "
                             "`'safety_checker=None'` instead.")
        if safety_checker is not None:
            raise NotImplementedError("the safety checker (a CLIP vision model) is outside the accelerated path; "
                                      "pass safety_checker=None as examples/brushnet/test_brushnet.py:150 does")
        for nm, mode in (("depth", depth_conditioning_mode), ("normals", normals_conditioning_mode)):
            if mode not in (None, "concat", "latents"):
                raise ValueError(f"{nm}_conditioning_mode must be None, 'concat' or 'latents', got {mode!r}")
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.brushnet, self.scheduler = unet, brushnet, scheduler
        self.safety_checker, self.feature_extractor, self.image_encoder = safety_checker, feature_extractor, image_encoder
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True)
        self.depth_conditioning_mode = depth_conditioning_mode
        self.normals_conditioning_mode = normals_conditioning_mode
        self.config = dict(requires_safety_checker=requires_safety_checker)
        self._progress_bar_config: Dict[str, Any] = {}
        self._guidance_scale = 7.5
        self._num_timesteps = 0
        self.use_hip_graph = True        # capture the denoise step into a hipGraph when the scheduler allows it
        self.overlap_brushnet = True     # BrushNet on a second HIP stream, ordered against the UNet by per-residual events
        self.share_brushnet_cfg = True   # BrushNet evaluated once when both CFG halves get identical inputs (_brushnet_shareable)
        # opt-in deviation from pipeline_brushnet.py:1188: draw ONE VAE-posterior sample of the conditioning latents and use
        # it for both classifier-free-guidance halves (the reference draws one per half); each half's distribution is
        # unchanged, BrushNet then runs once per image
        self.cfg_shared_conditioning_sample = False
        self._brushnet_once = False
        self._side_stream = None
        self.overlap_aux = False         # UNet shortcut convs / V projections on a third stream: measured 0.8 % slower
                                         # (18.15 vs 18.0 ms per step, tools/bench_aux.py), so off by default
        # both nets' time embeddings + fused time_emb_proj for the whole schedule in one batched pass before the loop
        # (graph path): 8 dependent launches fewer at the head of every step; A/B switch for tools/
        self.precompute_time_embedding = os.environ.get("MFHIP_NO_TEMB_TABLE") != "1"
        # opt-in: the 15 BrushNet zero-convs whose residual lands on a Transformer2DModel.proj_out run INSIDE that GEMM (models.LazyResidual)
        # measured on MI355X (gpurun_out/r03c): 16.75 ms per step with it, 16.56 without — the fused K = 2C GEMM lands on the
        # UNet's stream (the critical one) while the zero-conv it replaces ran in BrushNet's slack: OFF by default
        self.fold_zero_convs = os.environ.get("MFHIP_ZC_FOLD") == "1"
        self._added_cond = None          # SDXL: added_cond_kwargs of the (CFG-duplicated) batch, set by the XL subclass
        self._graph_state = None

    # ---- loading / saving (pipelines/pipeline_utils.py:148-296 save_pretrained, :465-919 from_pretrained) ---------
    _scheduler_classes = ("DDIMScheduler", "PNDMScheduler", "UniPCMultistepScheduler")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, torch_dtype=None, device="cuda", **kwargs):
        """The reference's calling convention (examples/brushnet/test_brushnet.py:146-155): `__init__` parameters without
        a default are modules — taken from kwargs, else loaded from the sub-folder `model_index.json` names (`brushnet`
        is never in an SD1.5 index, so it must be passed, like in the reference); parameters with a default
        (`depth_conditioning_mode`, ...) are plain kwargs.  `unet=None` means "load it from the base directory".
        Text encoder / tokenizer are outside the accelerated path: passed-in objects are kept, nothing is loaded."""
        import json
        import os
        from . import schedulers as S
        root = pretrained_model_name_or_path
        index = {}
        if os.path.exists(os.path.join(root, "model_index.json")):
            with open(os.path.join(root, "model_index.json")) as f:
                index = json.load(f)
        mods = {}
        for name, klass in (("vae", AutoencoderKL), ("unet", UNet2DConditionModel), ("brushnet", BrushNetModel)):
            obj = kwargs.pop(name, None)
            if obj is None:
                if not os.path.isdir(os.path.join(root, name)):
                    raise ValueError(f"Pipeline {cls.__name__} expected {{'{name}'}}, but it was neither passed nor "
                                     f"found under {root!r}.")
                obj = klass.from_pretrained(root, subfolder=name, torch_dtype=torch_dtype, device=device)
            mods[name] = obj
        sched = kwargs.pop("scheduler", None)
        if sched is None:
            with open(os.path.join(root, "scheduler", "scheduler_config.json")) as f:
                scfg = json.load(f)
            sname = scfg.get("_class_name") or (index.get("scheduler") or [None, "PNDMScheduler"])[1]
            if sname not in cls._scheduler_classes:
                raise NotImplementedError(f"scheduler {sname} is outside the MirrorFusion path (have {cls._scheduler_classes})")
            sched = getattr(S, sname).from_config(scfg)
        if kwargs.pop("safety_checker", None) is not None:
            raise NotImplementedError("the safety checker is outside the accelerated path; pass safety_checker=None")
        kwargs.pop("low_cpu_mem_usage", None)
        return cls(vae=mods["vae"], text_encoder=kwargs.pop("text_encoder", None), tokenizer=kwargs.pop("tokenizer", None),
                   unet=mods["unet"], brushnet=mods["brushnet"], scheduler=sched, safety_checker=None,
                   feature_extractor=kwargs.pop("feature_extractor", None),
                   requires_safety_checker=kwargs.pop("requires_safety_checker", False),
                   depth_conditioning_mode=kwargs.pop("depth_conditioning_mode", None),
                   normals_conditioning_mode=kwargs.pop("normals_conditioning_mode", None))

    def save_pretrained(self, save_directory: str, **unused):
        """`model_index.json` + one sub-folder per module in the reference's on-disk format (config.json +
        diffusion_pytorch_model.safetensors with the reference's key names; scheduler/scheduler_config.json)."""
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        index = {"_class_name": type(self).__name__, "_diffusers_version": "0.27.0.dev0"}
        for name in ("vae", "unet", "brushnet"):
            m = getattr(self, name)
            m.save_pretrained(os.path.join(save_directory, name))
            index[name] = ["diffusers", m._class_name]
        self.scheduler.save_config(os.path.join(save_directory, "scheduler"))
        index["scheduler"] = ["diffusers", type(self.scheduler).__name__]
        for name in ("text_encoder", "tokenizer", "safety_checker", "feature_extractor"):
            index[name] = [None, None]
        index.update(requires_safety_checker=False, depth_conditioning_mode=self.depth_conditioning_mode,
                     normals_conditioning_mode=self.normals_conditioning_mode)
        with open(os.path.join(save_directory, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2)

    # ---- DiffusionPipeline surface ----------------------------------------------------------------
    @property
    def device(self):
        return self.unet.device

    _execution_device = device

    def to(self, *args, **kwargs):
        for m in (self.vae, self.unet, self.brushnet):
            m.to(*args, **kwargs)
        return self

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    # Memory-saving switches of DiffusionPipeline (pipelines/pipeline_utils.py:940-1683).  None of them changes a result in
    # the reference; with 288 GB of HBM per GPU the modules simply stay resident, so they are accepted and do nothing
    # (test_brushnet.py:160 carries a commented-out enable_model_cpu_offload()).
    def enable_model_cpu_offload(self, gpu_id=None, device="cuda"):
        return None

    def enable_sequential_cpu_offload(self, gpu_id=None, device="cuda"):
        return None

    def enable_attention_slicing(self, slice_size="auto"):
        return None

    def disable_attention_slicing(self):
        return None

    def enable_vae_slicing(self):
        return None

    def disable_vae_slicing(self):
        return None

    def enable_vae_tiling(self):
        return None

    def disable_vae_tiling(self):
        return None

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        return None

    def disable_xformers_memory_efficient_attention(self):
        return None

    def progress_bar(self, total):
        try:
            from tqdm.auto import tqdm
            return tqdm(total=total, **self._progress_bar_config)
        except Exception:  # pragma: no cover
            class _N:
                def __enter__(s): return s
                def __exit__(s, *a): return False
                def update(s, *a): pass
            return _N()

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    @property
    def num_timesteps(self):
        return self._num_timesteps

    # ---- input validation (pipeline_brushnet.py:573-693) -------------------------------------------
    def check_inputs(self, prompt, image, mask, callback_steps, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None, brushnet_conditioning_scale=1.0, control_guidance_start=0.0,
                     control_guidance_end=1.0, callback_on_step_end_tensor_inputs=None, depth=None, normals=None):
        if callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if callback_on_step_end_tensor_inputs is not None and not all(
                k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None \
                and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, "
                             f"but got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")
        for nm, im in (("image", image), ("mask", mask)):
            if im is None:
                raise TypeError(f"`{nm}` must be passed (PIL image, numpy array, torch tensor or a list of those)")
        if self.depth_conditioning_mode is not None and depth is None:
            raise ValueError(f"depth_conditioning_mode={self.depth_conditioning_mode!r} needs a `depth` input")
        if self.normals_conditioning_mode is not None and normals is None:
            raise ValueError(f"normals_conditioning_mode={self.normals_conditioning_mode!r} needs a `normals` input")
        if not isinstance(brushnet_conditioning_scale, float):
            raise TypeError("For single brushnet: `brushnet_conditioning_scale` must be type `float`.")
        starts = control_guidance_start if isinstance(control_guidance_start, (tuple, list)) else [control_guidance_start]
        ends = control_guidance_end if isinstance(control_guidance_end, (tuple, list)) else [control_guidance_end]
        if len(starts) != len(ends):
            raise ValueError(f"`control_guidance_start` has {len(starts)} elements, but `control_guidance_end` has {len(ends)} elements.")
        for s, e in zip(starts, ends):
            if s >= e:
                raise ValueError(f"control guidance start: {s} cannot be larger or equal to control guidance end: {e}.")
            if s < 0.0:
                raise ValueError(f"control guidance start: {s} can't be smaller than 0.")
            if e > 1.0:
                raise ValueError(f"control guidance end: {e} can't be larger than 1.0.")

    # ---- prompt --------------------------------------------------------------------------------------
    def encode_prompt(self, prompt, num_images_per_prompt, do_cfg, negative_prompt=None, prompt_embeds=None,
                      negative_prompt_embeds=None, clip_skip: Optional[int] = None):
        """pipeline_brushnet.py:271-450 without LoRA / textual inversion.  clip_skip (:352-370): the hidden state clip_skip layers
        before the last one, through the text model's final LayerNorm (the text encoder is the caller's torch module: outside the
        accelerated path)."""
        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("pass `prompt_embeds` or construct the pipeline with a text_encoder and tokenizer")
            def enc(txt):
                ids = self.tokenizer(txt, padding="max_length", max_length=self.tokenizer.model_max_length,
                                     truncation=True, return_tensors="pt").input_ids
                ids = ids.to(next(self.text_encoder.parameters()).device)
                with torch.no_grad():
                    if clip_skip is None:
                        return self.text_encoder(ids)[0]
                    out = self.text_encoder(ids, output_hidden_states=True)
                    return self.text_encoder.text_model.final_layer_norm(out[-1][-(clip_skip + 1)])
            plist = [prompt] if isinstance(prompt, str) else prompt
            prompt_embeds = enc(plist)
            if do_cfg and negative_prompt_embeds is None:
                neg = negative_prompt if negative_prompt is not None else [""] * len(plist)
                neg = [neg] * len(plist) if isinstance(neg, str) else neg
                negative_prompt_embeds = enc(neg)
        b, s, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.float().repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, s, -1)
        if do_cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs `negative_prompt_embeds` when `prompt_embeds` is given")
            negative_prompt_embeds = negative_prompt_embeds.float().repeat(1, num_images_per_prompt, 1).view(
                b * num_images_per_prompt, s, -1)
        return prompt_embeds, negative_prompt_embeds

    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt, do_cfg=False):
        """pipeline_brushnet.py:741-774 — WITHOUT the CFG duplication (done after the VAE, see module docstring)."""
        image = self.image_processor.preprocess(image, height=height, width=width).to(dtype=torch.float32)
        repeat_by = batch_size if image.shape[0] == 1 else num_images_per_prompt
        return image.repeat_interleave(repeat_by, dim=0)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            noise = randn_tensor(shape, generator, self.device)      # on the generator's device, like randn_tensor
        else:
            noise = latents
        noise = noise.to(self.device, torch.float32)
        if self.scheduler.init_noise_sigma != 1.0:
            noise = hip.axpby_n([noise.contiguous()], [float(self.scheduler.init_noise_sigma)])
        return noise.contiguous(), noise

    def build_conditioning(self, image, mask, depth, height, width, batch, num_images_per_prompt, do_cfg,
                           conditioning_noise=None, normals=None):
        """pipeline_brushnet.py:1116-1215: [masked-image latents | mask | depth | normals] at latent resolution.
        `conditioning_noise`: the VAE posterior noise of the CFG-duplicated batch — one tensor (image) or a sequence
        (image, depth, normals) for the 'latents' modes; None draws them in the reference's order from the global RNG."""
        img = self.prepare_image(image, width, height, batch, num_images_per_prompt)
        m3 = self.prepare_image(mask, width, height, batch, num_images_per_prompt)
        original_mask = hip.mask_keep(m3) if m3.is_cuda else (m3.sum(1)[:, None, :, :] < 0).to(torch.float32)   # :1139 (1 = keep)
        height, width = img.shape[-2:]
        hl, wl = height // self.vae_scale_factor, width // self.vae_scale_factor
        dup = 2 if do_cfg else 1
        lat_c = self.vae.config["latent_channels"]
        sf = float(self.vae.config["scaling_factor"])
        noises = list(conditioning_noise) if isinstance(conditioning_noise, (list, tuple)) else [conditioning_noise]
        noises += [None] * (3 - len(noises))

        def encode_sample(x_host, noise):
            """vae.encode(x).latent_dist.sample() * scaling_factor for the CFG-duplicated batch: B images are encoded
            once and sampled once per CFG half (the reference encodes the duplicated batch, :1188)."""
            moments = self.vae._moments(x_host)
            if noise is None:
                if dup == 2 and self.cfg_shared_conditioning_sample:
                    noise = torch.randn(batch, lat_c, hl, wl, dtype=torch.float32).repeat(2, 1, 1, 1)
                else:
                    noise = torch.randn(dup * batch, lat_c, hl, wl, dtype=torch.float32)   # global RNG, like vae.py:782-791
                    self._cond_halves_identical = False     # two independent draws: never identical
            if noise.shape[0] != dup * batch:
                raise ValueError(f"conditioning_noise must have batch {dup * batch}")
            if dup == 2 and self._cond_halves_identical:
                # whether both CFG halves get the same sample is decided HERE, from how the noise was made (a host compare of a
                # host tensor; a device tensor costs one read-back) — not by comparing the built conditioning on the device
                self._cond_halves_identical = bool(torch.equal(noise[:batch], noise[batch:]))
            noise = noise.to(self.device, torch.float32)
            return [hip.vae_sample(moments, noise[i * batch:(i + 1) * batch].contiguous(), lat_c, sf) for i in range(dup)]

        self._cond_halves_identical = dup == 2
        halves = encode_sample(img, noises[0])
        parts = [[h] for h in halves]
        mask_l = hip.nearest_resize(hip.h2d(original_mask, self.device), hl, wl)                   # :1189-1195
        for ps in parts:
            ps.append(mask_l)
        if self.depth_conditioning_mode is not None:
            d = self.prepare_image(depth, width, height, batch, num_images_per_prompt)
            if self.depth_conditioning_mode == "concat":
                dl = hip.nearest_resize(hip.h2d(d, self.device), hl, wl)                           # :1198-1202
                for ps in parts:
                    ps.append(dl)
            else:
                for ps, h in zip(parts, encode_sample(d.repeat(1, 3, 1, 1), noises[1])):           # :1203-1206
                    ps.append(h)
        if self.normals_conditioning_mode is not None:
            nrm = self.prepare_image(normals, width, height, batch, num_images_per_prompt)
            if self.normals_conditioning_mode == "concat":
                nl = hip.nearest_resize(hip.h2d(nrm, self.device), hl, wl)                         # :1208-1212
                for ps in parts:
                    ps.append(nl)
            else:
                for ps, h in zip(parts, encode_sample(nrm, noises[2])):                            # :1213-1215
                    ps.append(h)
        # torch.cat along channels of each CFG half (:1196-1215) by the front-end kernel; stacking the halves is a copy
        return torch.cat([hip.concat_channels(ps, batch) for ps in parts], 0).contiguous()

    # ---- the call --------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt=None, image=None, mask=None, depth=None, normals=None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, timesteps: List[int] = None,
                 guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 ip_adapter_image=None, ip_adapter_image_embeds=None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, cross_attention_kwargs=None,
                 brushnet_conditioning_scale: Union[float, List[float]] = 1.0, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0,
                 control_guidance_end: Union[float, List[float]] = 1.0, clip_skip: Optional[int] = None,
                 callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                 conditioning_noise: Optional[torch.Tensor] = None, _timing: Optional[dict] = None, **kwargs):
        callback = kwargs.pop("callback", None)
        callback_steps = kwargs.pop("callback_steps", None)
        if ip_adapter_image is not None or ip_adapter_image_embeds is not None:
            raise NotImplementedError("IP-Adapter inputs are outside the BASELINE configs (SURVEY.md §2 #14)")
        if cross_attention_kwargs:
            raise NotImplementedError("cross_attention_kwargs (LoRA scale) are not built")
        if timesteps is not None:
            # retrieve_timesteps (pipeline_brushnet.py:113-119): the reference's DDIM / PNDM / UniPC `set_timesteps` take no custom
            # schedule either and the reference raises exactly this
            raise ValueError(f"The current scheduler class {self.scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" timestep schedules. Please check whether you are using the correct scheduler.")
        if guess_mode and type(self) is not StableDiffusionBrushNetPipeline:
            raise NotImplementedError("guess_mode is built for the SD1.5 pipeline")
        if isinstance(control_guidance_start, list) or isinstance(control_guidance_end, list):
            control_guidance_start = control_guidance_start[0] if isinstance(control_guidance_start, list) else control_guidance_start
            control_guidance_end = control_guidance_end[0] if isinstance(control_guidance_end, list) else control_guidance_end
        self.check_inputs(prompt, image, mask, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds,
                          brushnet_conditioning_scale, control_guidance_start, control_guidance_end,
                          callback_on_step_end_tensor_inputs, depth=depth, normals=normals)
        self._guidance_scale = guidance_scale
        guard = self.unet.prec.name == "f16x3" and self.device.type == "cuda"
        if guard:
            hip.split_overflow(reset=True)       # the range guard of the fp16 split precision counts this call only
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = self.do_classifier_free_guidance
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, num_images_per_prompt, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds, clip_skip=clip_skip)
        pe = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds         # :1103
        pe = pe.to(self.device)
        nb = batch_size * num_images_per_prompt

        first = image[0] if isinstance(image, list) else image
        if height is None or width is None:
            if torch.is_tensor(first) or isinstance(first, np.ndarray):
                hh, ww = (first.shape[-2:] if torch.is_tensor(first) else first.shape[-3:-1] if first.ndim == 4 else first.shape[:2])
            else:
                ww, hh = first.size
            height, width = height or int(hh), width or int(ww)
        # guess_mode (:771-772, 1260-1264): the conditioning is NOT doubled, BrushNet sees the conditional batch only
        cond = self.build_conditioning(image, mask, depth, height, width, nb, num_images_per_prompt, do_cfg and not guess_mode,
                                       conditioning_noise, normals)
        self._brushnet_once = (not guess_mode) and self._brushnet_shareable(cond, nb, do_cfg)

        self.scheduler.set_timesteps(num_inference_steps, device=self.device)                       # :1171
        ts = self.scheduler.timesteps
        dend = getattr(self, "_denoising_end", None)
        if dend is not None and isinstance(dend, float) and 0.0 < dend < 1.0:                        # pipeline_brushnet_sd_xl.py:1376-1391
            ntrain = int(self.scheduler.config["num_train_timesteps"])
            cutoff = int(round(ntrain - dend * ntrain))
            num_inference_steps = len([t for t in ts.tolist() if t >= cutoff])
            ts = ts[:num_inference_steps]
        self._num_timesteps = len(ts)
        latents, _ = self.prepare_latents(nb, self.unet.config["in_channels"], height, width, generator, latents)

        keep = [1.0 - float(i / len(ts) < control_guidance_start or (i + 1) / len(ts) > control_guidance_end)
                for i in range(len(ts))]                                                             # :1236-1242
        rescale = float(getattr(self, "_guidance_rescale", 0.0) or 0.0)
        fused_ddim = isinstance(self.scheduler, DDIMScheduler) and eta == 0.0 and rescale == 0.0
        num_warmup = len(ts) - num_inference_steps * self.scheduler.order
        if _timing is not None:          # HIP events on the launch stream around the denoise loop (bench.py)
            _timing["host_before_denoise"] = time.perf_counter()
            _timing["denoise_start"] = torch.cuda.Event(enable_timing=True)
            _timing["denoise_end"] = torch.cuda.Event(enable_timing=True)
            _timing["denoise_start"].record()
        use_graph = (self.use_hip_graph and do_cfg and callback is None and (fused_ddim or eta == 0.0) and rescale == 0.0
                     and all(k == 1.0 for k in keep) and len(ts) > 2 and not guess_mode)
        with self.progress_bar(total=num_inference_steps) as bar:
            if use_graph:
                latents = self._denoise_graph(latents, ts, pe, cond, nb, guidance_scale, float(brushnet_conditioning_scale),
                                              callback_on_step_end, callback_on_step_end_tensor_inputs, prompt_embeds,
                                              negative_prompt_embeds, bar, fused_ddim, eta, generator)
                ts = []
            for i, t in enumerate(ts):                                                               # :1250 HOT LOOP
                # step 0 autotunes GEMM tiles: keep its timings undisturbed; guess_mode re-packs the residuals on this
                # stream (the per-residual events are keyed by the tensors BrushNet returned)
                self._overlap(i > 0 and not guess_mode)
                x_in = torch.cat([latents] * 2) if do_cfg else latents                               # :1256
                x_in = self.scheduler.scale_model_input(x_in, t)
                cond_scale = float(brushnet_conditioning_scale) * keep[i]
                once = self._brushnet_once or (guess_mode and do_cfg)      # BrushNet on the conditional batch only
                down, mid, up = self.brushnet(x_in[nb:] if (guess_mode and do_cfg) else x_in[:nb] if once else x_in, t,
                                              encoder_hidden_states=pe[nb:] if (guess_mode and do_cfg) else pe[:nb] if once else pe,
                                              brushnet_cond=cond[:nb] if once else cond,
                                              conditioning_scale=cond_scale, added_cond_kwargs=self._added_cond,
                                              guess_mode=guess_mode, return_dict=False)              # :1277
                if guess_mode and do_cfg:           # zeros keep the unconditional half unchanged (:1287-1293); copies only
                    down = [torch.cat([torch.zeros_like(d), d]) for d in down]
                    mid = torch.cat([torch.zeros_like(mid), mid])
                    up = [torch.cat([torch.zeros_like(u), u]) for u in up]
                eps = self.unet(x_in, t, encoder_hidden_states=pe, down_block_add_samples=down,
                                mid_block_add_sample=mid, up_block_add_samples=up, added_cond_kwargs=self._added_cond,
                                return_dict=False)[0]                                                # :1296
                if do_cfg:
                    eu, ec = eps[:nb], eps[nb:]
                    if fused_ddim:
                        latents = self.scheduler.step(None, t, latents, return_dict=False, _cfg=(eu, ec, guidance_scale))[0]
                    else:
                        noise_pred = hip.cfg_combine(eu, ec, float(guidance_scale))                 # :1310-1312
                        if rescale > 0.0:
                            # rescale_noise_cfg (pipeline_brushnet_sd_xl.py:1478-1480; pipeline_stable_diffusion.py:59-70): one standard
                            # deviation per image and two scalings — a non-default switch outside the timed path, on torch's device ops
                            dims = list(range(1, ec.ndim))
                            std_text, std_cfg = ec.float().std(dim=dims, keepdim=True), noise_pred.float().std(dim=dims, keepdim=True)
                            noise_pred = (rescale * (noise_pred.float() * (std_text / std_cfg)) + (1.0 - rescale) * noise_pred.float()).to(noise_pred.dtype)
                        latents = self._sched_step(noise_pred, t, latents, eta, generator)
                else:
                    latents = self._sched_step(eps, t, latents, eta, generator)                     # :1315
                if callback_on_step_end is not None:
                    avail = dict(latents=latents, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds)
                    cb_kwargs = {k: avail[k] for k in callback_on_step_end_tensor_inputs}
                    outs = callback_on_step_end(self, i, t, cb_kwargs) or {}
                    latents = outs.pop("latents", latents)
                if i == len(ts) - 1 or ((i + 1) > num_warmup and (i + 1) % self.scheduler.order == 0):
                    bar.update()
                    if callback is not None and callback_steps and i % callback_steps == 0:
                        callback(i // getattr(self.scheduler, "order", 1), t, latents)

        self._overlap(False)
        if _timing is not None:
            _timing["denoise_end"].record()
            _timing["host_after_denoise"] = time.perf_counter()
        if output_type != "latent":
            sf = float(self.vae.config["scaling_factor"])
            z = hip.axpby_n([latents.contiguous()], [1.0 / sf])                                     # :1342
            img = self.vae.decode(z, return_dict=False)[0]
        else:
            img = latents
        img = self.image_processor.postprocess(img, output_type=output_type, do_denormalize=[True] * img.shape[0])
        if guard:
            raised = hip.split_overflow(reset=True)
            if raised:
                raise hip.SplitRangeError(
                    f"f16x3: an activation exceeded the fp16 range (|x| > 65504, flags {raised:#x}) during this call; its products "
                    "were computed from saturated halves, so the result is not returned.  Run this input with precision "
                    "'bf16x3' (fp32 range) or 'fp32'.")
        if not return_dict:
            return (img, None)
        return StableDiffusionPipelineOutput(images=img, nsfw_content_detected=None)

    def _brushnet_shareable(self, cond: torch.Tensor, nb: int, do_cfg: bool) -> bool:
        """Under classifier-free guidance the reference feeds BrushNet the duplicated batch (pipeline_brushnet.py:1256-1277:
        torch.cat([latents] * 2), [negative | positive] prompt embeddings, the conditioning latents of the duplicated image
        batch).  The BrushNet built by from_unet is attention-free (brushnet.py:484-486), so it never reads the prompt
        embeddings: its two halves differ ONLY through the conditioning latents, whose VAE posterior the reference samples
        independently per half (:1188 on the image doubled at :771-772).  When the two halves of `cond` are bit-identical
        (conditioning_noise with equal halves, or cfg_shared_conditioning_sample) both halves are the same function of the
        same inputs: BrushNet is evaluated once per image and the UNet's injection adds read each residual for both
        halves (mf_gemm_desc.res1_rows).  Never when an added embedding depends on the prompt (SDXL's pooled text
        embedding enters BrushNet-XL's time embedding)."""
        if not (do_cfg and self.share_brushnet_cfg and self.brushnet.config.get("addition_embed_type") is None
                and self.brushnet.config.get("class_embed_type") is None):
            return False
        # identical halves <=> every VAE posterior sample that went into `cond` was the same for both halves (mask, depth and
        # normals maps are deterministic functions of the inputs): recorded by build_conditioning, no device compare / sync
        return bool(getattr(self, "_cond_halves_identical", False))

    def _overlap(self, on: bool):
        """Turn the BrushNet || UNet stream overlap (models._RESIDUAL_EVENTS) on or off for the next forward calls."""
        if on and self.overlap_brushnet and self.device.type == "cuda":
            if self._side_stream is None or self._side_stream.device != self.device:
                self._side_stream = torch.cuda.Stream(device=self.device)
                self._aux_stream = torch.cuda.Stream(device=self.device)
            self.brushnet.side_stream = self._side_stream
            # an auxiliary stream for UNet launches off its critical path (shortcut convs, V projections).  BrushNet
            # keeps a single stream: a fork of the already forked side stream segfaults in hipStreamEndCapture
            # (ROCm 7.2), and its shortcut convs already overlap the UNet
            self.unet.aux_stream = self._aux_stream if self.overlap_aux else None
        else:
            self.brushnet.side_stream = None
            self.unet.aux_stream = self.brushnet.aux_stream = None

    def _denoise_graph(self, latents, ts, pe, cond, nb, guidance_scale, cond_scale, callback_on_step_end,
                       cb_inputs, prompt_embeds, negative_prompt_embeds, bar, fused_ddim=True, eta=0.0, generator=None):
        """The hot loop as ONE captured hipGraph replayed per timestep (BrushNet + UNet + CFG [+ DDIM update], ~800
        kernels on two streams): launch overhead disappears and the host only refreshes two tiny device buffers
        (timestep, scheduler coefficients) between replays.  DDIM's update is inside the graph; multistep schedulers
        (PNDM, UniPC: host-side state and per-step scalar coefficients) get the guided noise prediction from the graph
        and step eagerly (a handful of mf_axpby_n launches).  The first step runs eagerly: it autotunes GEMM tiles,
        binds the prompt (cross-attention K/V cache) and sizes every scratch buffer before anything is captured."""
        sched = self.scheduler
        dev = latents.device
        tvals = ts.to(torch.float32).to(dev)
        if fused_ddim:
            coefs = torch.tensor([sched.step_coefficients(int(t))[:4] for t in ts], dtype=torch.float32).to(dev)
            clip = float(sched.config["clip_sample_range"]) if sched.config["clip_sample"] else 0.0
            ptype = 0 if sched.config["prediction_type"] == "epsilon" else 1
        else:
            coefs, clip, ptype = None, 0.0, type(sched).__name__
        # The captured graph (and every buffer it reads) is kept across calls with the same shapes and scalars:
        # new inputs are copied INTO the static buffers, so repeated calls pay no capture / instantiate cost.  The key
        # carries each model's weights generation: load_state_dict / .to() rebuild every weight tensor (and the UNet's
        # cross-attention K/V cache), so a graph captured before that points at freed buffers and must not be replayed.
        added = self._added_cond
        key = (tuple(latents.shape), tuple(pe.shape), tuple(cond.shape), float(guidance_scale), cond_scale, ptype, clip,
               str(dev), id(self.unet), self.unet._weights_gen, id(self.brushnet), self.brushnet._weights_gen, self._brushnet_once,
               tuple((k, tuple(v.shape)) for k, v in sorted(added.items())) if added else None)
        st = self._graph_state if self._graph_state is not None and self._graph_state["key"] == key else None
        if st is None:
            st = dict(key=key, graph=None, lat=torch.empty_like(latents), pe=torch.empty_like(pe),
                      cond=torch.empty_like(cond), t_cur=torch.empty(1, dtype=torch.float32, device=dev),
                      coef_cur=torch.empty(4, dtype=torch.float32, device=dev))
            if added:
                st["added"] = {k: torch.empty(v.shape, dtype=torch.float32, device=dev) for k, v in added.items()}
            if not fused_ddim:
                st["eps"] = torch.empty_like(latents)            # guided noise prediction handed to scheduler.step
            self._graph_state = st
        if added:
            for k, v in added.items():
                st["added"][k].copy_(v)
            added = st["added"]
        lat, t_cur, coef_cur = st["lat"], st["t_cur"], st["coef_cur"]
        lat.copy_(latents)
        st["pe"].copy_(pe)
        st["cond"].copy_(cond)
        pe, cond = st["pe"], st["cond"]
        # every timestep is known here: both nets' time embeddings and the fused time_emb_proj GEMM run ONCE for
        # the whole schedule (4 batched launches per net) instead of 8 dependent launches at the head of every step; the
        # graph reads one row block per step from a static buffer.  Kept across calls with the same schedule.
        tkey = (tuple(float(t) for t in ts.tolist()), self.unet._weights_gen, self.brushnet._weights_gen)
        if self.precompute_time_embedding and (st.get("temb_key") != tkey or added):
            nrows_b = nb if self._brushnet_once else pe.shape[0]
            st["temb_tab"] = (self.unet.time_embedding_table(tvals, pe.shape[0], added),
                              self.brushnet.time_embedding_table(tvals, nrows_b, added))
            st["temb_key"] = tkey
            if "temb_cur" not in st or any(c.shape != t.shape[1:] for c, t in zip(st["temb_cur"], st["temb_tab"])):
                if st["graph"] is not None:
                    raise RuntimeError("time-embedding table changed shape under a captured graph")
                st["temb_cur"] = tuple(torch.empty_like(t[0]) for t in st["temb_tab"])
        temb_u, temb_b = st["temb_cur"] if self.precompute_time_embedding else (None, None)

        # residuals that land on a proj_out are never computed: their zero-conv becomes a second K segment of that GEMM
        lazy = self.unet.lazy_injection_names() if (self.fold_zero_convs and not self._brushnet_once) else None

        def one_step():
            x_in = torch.cat([lat] * 2)
            once = self._brushnet_once
            down, mid, up = self.brushnet(lat if once else x_in, t_cur, encoder_hidden_states=pe[:nb] if once else pe,
                                          brushnet_cond=cond[:nb] if once else cond,
                                          conditioning_scale=cond_scale, added_cond_kwargs=added, return_dict=False, _temb=temb_b,
                                          _lazy=lazy)
            eps = self.unet(x_in, t_cur, encoder_hidden_states=pe, down_block_add_samples=down,
                            mid_block_add_sample=mid, up_block_add_samples=up, added_cond_kwargs=added,
                            return_dict=False, _temb=temb_u)[0]
            if fused_ddim:
                hip.cfg_ddim_step_dev(eps[:nb], eps[nb:], float(guidance_scale), lat, coef_cur, ptype, clip, out=lat)
            else:
                st["eps"].copy_(hip.cfg_combine(eps[:nb], eps[nb:], float(guidance_scale)))         # :1310-1312

        # A graph captured by an earlier call with this key serves every step, the first included: what depends on the
        # prompt alone (its device copy, the cross-attention K / V^T) is recomputed into the buffers the graph reads.
        export = getattr(self, "_export_step_to", None)           # export_denoise_step(): the capture of step 1 is recorded as a program
        if export is not None:
            st["graph"] = None                                    # (a graph kept from an earlier call is captured again)
        replay_all = (export is None and st["graph"] is not None and os.environ.get("MFHIP_EAGER_FIRST") != "1"      # A/B switch
                      and self.unet.bind_prompt(pe))
        for i in range(len(ts)):
            t_cur.copy_(tvals[i:i + 1])
            if fused_ddim:
                coef_cur.copy_(coefs[i])
            if temb_u is not None:
                temb_u.copy_(st["temb_tab"][0][i])
                temb_b.copy_(st["temb_tab"][1][i])
            if i == 0 and not replay_all:
                one_step()                                   # eager: tunes GEMMs, binds the prompt K/V, sizes scratch
            else:
                if st["graph"] is None:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    # export_denoise_step(): the captured pass is ALSO written down as a step program (program.py)
                    rec = self._export_recorder(st, lat, coef_cur, temb_u, temb_b, cond, coefs, fused_ddim) if export is not None else None
                    self._overlap(True)                      # the side stream forks from / joins the capture stream
                    try:
                        # (thread_local: with an initialised process group its watchdog thread may poll events while this thread captures)
                        with torch.cuda.graph(graph, capture_error_mode="thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"):
                            if rec is not None:
                                with rec:
                                    one_step()
                            else:
                                one_step()
                    except Exception:
                        if rec is not None and rec.error is not None:
                            raise rec.error                  # (not the "unjoined work" the aborted capture reports on top of it)
                        raise
                    finally:
                        self._overlap(False)
                    st["graph"] = graph                      # capture does not execute: replay below runs this step
                    if rec is not None:                      # (so the buffers still hold what they held BEFORE the recorded step)
                        self._export_save(rec, graph, export, lat, temb_u, temb_b, len(ts), i, guidance_scale, cond_scale)
                st["graph"].replay()
            if not fused_ddim:
                # multistep schedulers keep references to their inputs (ets, last_sample): hand them copies, not the
                # graph's static buffers
                lat.copy_(self._sched_step(st["eps"].clone(), ts[i], lat.clone(), eta, generator))   # :1315
            if callback_on_step_end is not None:
                avail = dict(latents=lat, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds)
                outs = callback_on_step_end(self, i, ts[i], {k: avail[k] for k in cb_inputs}) or {}
                new = outs.pop("latents", None)
                if new is not None and new is not lat:
                    lat.copy_(new)
            bar.update()
        return lat.clone()

    def _export_recorder(self, st, lat, coef_cur, temb_u, temb_b, cond, coefs, fused_ddim):
        """The recorder of ONE denoise step (the loop body of pipeline_brushnet.py:1250-1332), entered inside the hipGraph capture
        of that step: the program carries the capture's forks and joins (BrushNet || UNet), replayed by mf_denoise_step_fused."""
        from . import program
        if temb_u is None:
            raise NotImplementedError("export_denoise_step needs precompute_time_embedding (the step reads one row block of the schedule's table)")
        named = dict(latents=lat, temb_unet=temb_u, temb_brushnet=temb_b, cond=cond)
        tables = {"table.temb_unet": st["temb_tab"][0].contiguous(), "table.temb_brushnet": st["temb_tab"][1].contiguous()}
        if fused_ddim:
            named["coef4"] = coef_cur
            tables["table.coef4"] = coefs.contiguous()
        else:
            # multistep schedulers (PNDM, UniPC: host-side state between steps, scheduling_pndm.py:321-390): the program ends with the
            # guided noise prediction in the io buffer "eps"; the update of the latents is the host's (mf_axpby_n)
            named["eps"] = st["eps"]
        return program.Recorder(named, tables, capture=True)

    def _export_save(self, rec, graph, path, lat, temb_u, temb_b, nsteps, step, guidance_scale, cond_scale):
        import json
        rec.finish(graph.pool())
        meta = dict(entry="mf_denoise_step_fused", reference="pipelines/brushnet/pipeline_brushnet.py:1250-1332", precision=self.unet.prec.name,
                    result="latents (DDIM update inside)" if "coef4" in rec.named else "eps (guided noise prediction; the scheduler update is the host's)",
                    latents=list(lat.shape), steps=nsteps, recorded_step=step, guidance_scale=float(guidance_scale),
                    conditioning_scale=cond_scale if isinstance(cond_scale, (int, float)) else list(cond_scale),
                    temb_unet=list(temb_u.shape), temb_brushnet=list(temb_b.shape), brushnet_once=bool(self._brushnet_once))
        self._export_info = rec.save(path, meta=json.dumps(meta))
        self._export_info["meta"] = meta

    def export_denoise_step(self, path: str, **call_kwargs) -> dict:
        """Run the pipeline once and write the denoise step's program to `path` (see program.py / include/mfhip.h "step programs").
        `call_kwargs`: what __call__ takes; the program is specialised on their shapes, the prompt (its cross-attention K / V^T are
        constants of the file), the guidance and conditioning scales, and the schedule (its tables travel as named constants)."""
        if self.device.type != "cuda":
            raise hip.MfhipError("export_denoise_step needs the device: a program is a recording of real launches")
        self._export_step_to, self._export_info = path, None
        try:
            call_kwargs.setdefault("output_type", "latent")
            out = self(**call_kwargs)
        finally:
            self._export_step_to = None
        if self._export_info is None:
            raise hip.MfhipError("export_denoise_step: the call did not reach a second denoise step on the graph path (num_inference_steps >= 2, "
                                 "use_graph left on)")
        self._export_info["result"] = out
        return self._export_info

    def _sched_step(self, noise_pred, t, latents, eta, generator):
        import inspect
        params = inspect.signature(self.scheduler.step).parameters
        extra = {}
        if "eta" in params:
            extra["eta"] = eta
        if "generator" in params:
            extra["generator"] = generator
        return self.scheduler.step(noise_pred, t, latents, **extra, return_dict=False)[0]


class StableDiffusionXLBrushNetPipeline(StableDiffusionBrushNetPipeline):
    """pipelines/brushnet/pipeline_brushnet_sd_xl.py:936-1535 on the same engine: the SDXL UNet / BrushNet-XL add the
    'text_time' embedding (pooled text embedding + Fourier features of original size / crop / target size) to the time
    embedding; conditioning is [masked-image latents | mask] (5 channels, :1301-1310); everything else is the
    SD1.5 loop.  Text encoders are outside the accelerated path: pass prompt_embeds and pooled_prompt_embeds."""

    def __init__(self, vae, text_encoder, text_encoder_2, tokenizer, tokenizer_2, unet, brushnet, scheduler,
                 force_zeros_for_empty_prompt: bool = True, add_watermarker=None, feature_extractor=None,
                 image_encoder=None):
        super().__init__(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, brushnet=brushnet,
                         scheduler=scheduler, safety_checker=None, feature_extractor=feature_extractor,
                         image_encoder=image_encoder, requires_safety_checker=False)
        if add_watermarker:
            raise NotImplementedError("the invisible watermark is outside the accelerated path")
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2
        self.config.update(force_zeros_for_empty_prompt=force_zeros_for_empty_prompt)

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, text_encoder_projection_dim):
        """pipeline_brushnet_sd_xl.py:_get_add_time_ids, including its consistency check against add_embedding."""
        ids = list(original_size) + list(crops_coords_top_left) + list(target_size)
        passed = self.unet.config["addition_time_embed_dim"] * len(ids) + text_encoder_projection_dim
        expected = self.unet.config["projection_class_embeddings_input_dim"]
        if expected != passed:
            raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} "
                             "was created. The model has an incorrect config. Please check "
                             "`unet.config.time_embedding_type` and `text_encoder_2.config.projection_dim`.")
        return torch.tensor([ids], dtype=torch.float32)

    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, image=None, mask=None, height=None, width=None,
                 num_inference_steps: int = 50, denoising_end=None, guidance_scale: float = 5.0, negative_prompt=None,
                 negative_prompt_2=None, num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, output_type="pil", return_dict: bool = True,
                 cross_attention_kwargs=None, guidance_rescale: float = 0.0, brushnet_conditioning_scale=1.0,
                 guess_mode: bool = False, control_guidance_start=0.0, control_guidance_end=1.0, original_size=None,
                 crops_coords_top_left=(0, 0), target_size=None, negative_original_size=None,
                 negative_crops_coords_top_left=(0, 0), negative_target_size=None, clip_skip=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",), conditioning_noise=None,
                 **kwargs):
        if prompt is not None or prompt_2 is not None or negative_prompt is not None or negative_prompt_2 is not None:
            raise NotImplementedError("the two CLIP text encoders are outside the accelerated path: pass prompt_embeds, "
                                      "negative_prompt_embeds, pooled_prompt_embeds and negative_pooled_prompt_embeds")
        if prompt_embeds is None or pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed. Make sure to "
                             "generate `pooled_prompt_embeds` from the same text encoder that was used to generate `prompt_embeds`.")
        do_cfg = guidance_scale > 1
        if do_cfg and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            raise ValueError("If `negative_prompt_embeds` are provided, `negative_pooled_prompt_embeds` also have to be passed.")
        ih, iw = self.image_processor.preprocess(image).shape[-2:]
        height, width = height or ih, width or iw
        original_size = original_size or (ih, iw)                                                   # :1330-1334
        target_size = target_size or (height, width)
        b = prompt_embeds.shape[0]
        nb = b * num_images_per_prompt
        pooled = pooled_prompt_embeds.float().repeat(1, num_images_per_prompt).view(nb, -1)         # encode_prompt :483-486
        proj_dim = int(pooled.shape[-1])
        ids = self._get_add_time_ids(original_size, crops_coords_top_left, target_size, proj_dim)
        if negative_original_size is not None and negative_target_size is not None:
            nids = self._get_add_time_ids(negative_original_size, negative_crops_coords_top_left, negative_target_size, proj_dim)
        else:
            nids = ids
        if do_cfg:
            npooled = negative_pooled_prompt_embeds.float().repeat(1, num_images_per_prompt).view(nb, -1)
            text = torch.cat([npooled, pooled], 0)
            ids = torch.cat([nids, ids], 0)
        else:
            text = pooled
        ids = ids.repeat(nb, 1)                                              # :1372 (sic: interleaves neg/pos rows)
        self._added_cond = dict(text_embeds=text.to(self.device), time_ids=ids.to(self.device))
        # denoising_end (pipeline_brushnet_sd_xl.py:1376-1391) and guidance_rescale (:1478-1480) are read by the shared loop
        self._denoising_end, self._guidance_rescale = denoising_end, float(guidance_rescale or 0.0)
        try:
            return super().__call__(image=image, mask=mask, height=height, width=width,
                                    num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                                    num_images_per_prompt=num_images_per_prompt, eta=eta, generator=generator,
                                    latents=latents, prompt_embeds=prompt_embeds,
                                    negative_prompt_embeds=negative_prompt_embeds, output_type=output_type,
                                    return_dict=return_dict, cross_attention_kwargs=cross_attention_kwargs,
                                    brushnet_conditioning_scale=brushnet_conditioning_scale, guess_mode=guess_mode,
                                    control_guidance_start=control_guidance_start,
                                    control_guidance_end=control_guidance_end, clip_skip=clip_skip,
                                    callback_on_step_end=callback_on_step_end,
                                    callback_on_step_end_tensor_inputs=list(callback_on_step_end_tensor_inputs),
                                    conditioning_noise=conditioning_noise, **kwargs)
        finally:
            self._added_cond = None
            self._denoising_end, self._guidance_rescale = None, 0.0
