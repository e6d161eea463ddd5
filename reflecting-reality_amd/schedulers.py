"""DDIMScheduler and PNDMScheduler with the reference call surface, stepping on the device.

Reference: MirrorFusion/src/diffusers/schedulers/scheduling_ddim.py (set_timesteps :299-342, step :344-470,
add_noise :473-497) and scheduling_pndm.py (set_timesteps :168-226, step_prk :261-319, step_plms :321-390,
_get_prev_sample :407-448).  The scalar coefficients are computed on the host exactly like the reference
does (fp32 0-dim tensor arithmetic on the alphas_cumprod table); the per-element update is one HIP kernel
(mf_cfg_ddim_step / mf_axpby_n).  Latents stay fp32 whatever the model precision.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import hip
from .models import FrozenConfig


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


def _betas(cfg) -> torch.Tensor:
    n = cfg["num_train_timesteps"]
    if cfg.get("trained_betas") is not None:
        return torch.tensor(cfg["trained_betas"], dtype=torch.float32)
    sch = cfg["beta_schedule"]
    if sch == "linear":
        return torch.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=torch.float32)
    if sch == "scaled_linear":
        return torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(f"{sch} is not implemented (reference supports it; outside the MirrorFusion path)")


class _SchedulerBase:
    order = 1
    _defaults: dict = {}

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        unknown = [k for k in kwargs if k not in cfg]
        cfg.update({k: v for k, v in kwargs.items() if k in cfg})
        # configuration_utils.py register_to_config: remember which keys were left at their defaults, so that
        # from_config into another scheduler class falls back to THAT class's defaults for them
        cfg["_use_default_values"] = [k for k in self._defaults if k not in kwargs]
        self.config = FrozenConfig(cfg)
        self._unknown = unknown
        self.betas = _betas(cfg)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg.get("set_alpha_to_one", True) else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, cfg["num_train_timesteps"])[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config, **kwargs):
        """configuration_utils.py:~250: keys the target class does not know are dropped; keys the source left at
        its default fall back to the TARGET's defaults (`_use_default_values`)."""
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        use_default = dict(config).get("_use_default_values", [])
        cfg = {k: v for k, v in cfg.items() if k not in use_default}
        cfg.update(kwargs)
        return cls(**{k: v for k, v in cfg.items() if k in cls._defaults})

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    config_name = "scheduler_config.json"

    def save_config(self, path: str):
        """configuration_utils.py save_config: `scheduler_config.json` with `_class_name` (explicitly set keys only
        are what a later from_config into another class keeps; the rest is re-defaulted there)."""
        import json
        import os
        os.makedirs(path, exist_ok=True)
        cfg = {k: v for k, v in dict(self.config).items() if not k.startswith("_")}
        cfg["_class_name"] = type(self).__name__
        cfg["_diffusers_version"] = "0.27.0.dev0"
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)

    save_pretrained = save_config

    @classmethod
    def from_pretrained(cls, path: str, subfolder=None, **kwargs):
        import json
        import os
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = json.load(f)
        return cls.from_config(cfg, **kwargs)

    def __len__(self):
        return self.config["num_train_timesteps"]

    def _alpha(self, t: int) -> torch.Tensor:
        return self.alphas_cumprod[t] if t >= 0 else self.final_alpha_cumprod

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """scheduling_ddim.py:473-497 (== DDPM): sqrt(a_t) x + sqrt(1 - a_t) eps, per-sample t."""
        ts = timesteps.cpu().long().reshape(-1)
        a = self.alphas_cumprod[ts]
        sa, sb = a ** 0.5, (1 - a) ** 0.5
        outs = []
        for i in range(original_samples.shape[0]):
            k = i if ts.numel() > 1 else 0
            outs.append(hip.axpby_n([original_samples[i].float().contiguous(), noise[i].float().contiguous()],
                                    [float(sa[k]), float(sb[k])]))
        return torch.stack(outs)

    def get_velocity(self, sample: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """scheduling_ddpm.py:527-546: v = sqrt(a_t) eps - sqrt(1 - a_t) x, per-sample t."""
        ts = timesteps.cpu().long().reshape(-1)
        a = self.alphas_cumprod[ts]
        sa, sb = a ** 0.5, (1 - a) ** 0.5
        outs = []
        for i in range(sample.shape[0]):
            k = i if ts.numel() > 1 else 0
            outs.append(hip.axpby_n([noise[i].float().contiguous(), sample[i].float().contiguous()],
                                    [float(sa[k]), -float(sb[k])]))
        return torch.stack(outs)


class DDPMScheduler(_SchedulerBase):
    """The training script's noise scheduler (train_brushnet_mirror.py:957): only the forward-process half
    (`add_noise`, `get_velocity`, `alphas_cumprod`) is used there; sampling with it is not part of the path."""
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                     thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                     sample_max_value=1.0, timestep_spacing="leading", steps_offset=0,
                     rescale_betas_zero_snr=False)

    def set_timesteps(self, num_inference_steps: int, device=None):
        raise NotImplementedError("DDPMScheduler here is the training-side noise scheduler; sample with DDIM/PNDM/UniPC")

    def step(self, *args, **kwargs):
        raise NotImplementedError("DDPMScheduler here is the training-side noise scheduler; sample with DDIM/PNDM/UniPC")


class DDIMScheduler(_SchedulerBase):
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                     prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                     clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                     rescale_betas_zero_snr=False)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c["thresholding"] or c["rescale_betas_zero_snr"]:
            raise NotImplementedError("thresholding / rescale_betas_zero_snr are outside the MirrorFusion path")
        if c["prediction_type"] not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"prediction_type {c['prediction_type']}")

    def set_timesteps(self, num_inference_steps: int, device=None):
        c = self.config
        if num_inference_steps > c["num_train_timesteps"]:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.config.train_timesteps`: {c['num_train_timesteps']}")
        self.num_inference_steps = num_inference_steps
        sp = c["timestep_spacing"]
        if sp == "linspace":
            ts = np.linspace(0, c["num_train_timesteps"] - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            ratio = c["num_train_timesteps"] // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += c["steps_offset"]
        elif sp == "trailing":
            ratio = c["num_train_timesteps"] / num_inference_steps
            ts = np.round(np.arange(c["num_train_timesteps"], 0, -ratio)).astype(np.int64) - 1
        else:
            raise ValueError(f"{sp} is not supported. Please make sure to choose one of 'leading' or 'trailing'.")
        self.timesteps = torch.from_numpy(ts)           # kept on the host: the loop indexes coefficient tables

    def _get_variance(self, timestep, prev_timestep):
        a_t, a_p = self._alpha(int(timestep)), self._alpha(int(prev_timestep))
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def step_coefficients(self, timestep: int, eta: float = 0.0) -> Tuple[float, float, float, float, float]:
        """(sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2), sigma) as the reference computes them."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        timestep = int(timestep)
        prev_t = timestep - self.config["num_train_timesteps"] // self.num_inference_steps
        a_t, a_p = self._alpha(timestep), self._alpha(prev_t)
        b_t = 1 - a_t
        std = eta * self._get_variance(timestep, prev_t) ** 0.5 if eta > 0 else torch.tensor(0.0)
        return (float(a_t ** 0.5), float(b_t ** 0.5), float(a_p ** 0.5), float((1 - a_p - std ** 2) ** 0.5), float(std))

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise: Optional[torch.Tensor] = None,
             return_dict: bool = True, _cfg: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None):
        """scheduling_ddim.py:344-470.  `_cfg=(eps_uncond, eps_cond, guidance)` fuses the guidance combine
        (pipeline_brushnet.py:1310-1312) into the same kernel."""
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output")
        sa, sb, sp, dirc, std = self.step_coefficients(timestep, eta)
        c = self.config
        clip = float(c["clip_sample_range"]) if c["clip_sample"] else 0.0
        ptype = 0 if c["prediction_type"] == "epsilon" else 1
        x = sample.float().contiguous()
        if _cfg is not None:
            eu, ec, g = _cfg
            prev = hip.cfg_ddim_step(eu, ec, float(g), x, sa, sb, sp, dirc, pred_type=ptype, clip=clip)
        else:
            prev = hip.cfg_ddim_step(model_output.float().contiguous(), None, -1.0, x, sa, sb, sp, dirc,
                                     pred_type=ptype, clip=clip)
        if eta > 0:
            if variance_noise is None:
                from .rng import randn_tensor
                variance_noise = randn_tensor(model_output.shape, generator, prev.device)   # scheduling_ddim.py:455-458
            prev = hip.axpby_n([prev, variance_noise.to(prev.device).float().contiguous()], [1.0, std])
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev)


class PNDMScheduler(_SchedulerBase):
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, skip_prk_steps=False, set_alpha_to_one=False, prediction_type="epsilon",
                     timestep_spacing="leading", steps_offset=0)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if self.config["prediction_type"] not in ("epsilon", "v_prediction"):
            raise ValueError(f"prediction_type given as {self.config['prediction_type']} must be one of `epsilon` or `v_prediction`")
        self.pndm_order = 4
        self.cur_model_output = None
        self.counter = 0
        self.cur_sample = None
        self.ets: List[torch.Tensor] = []
        self.prk_timesteps = None
        self.plms_timesteps = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        c = self.config
        self.num_inference_steps = num_inference_steps
        nt = c["num_train_timesteps"]
        sp = c["timestep_spacing"]
        if sp == "linspace":
            _ts = np.linspace(0, nt - 1, num_inference_steps).round().astype(np.int64)
        elif sp == "leading":
            _ts = (np.arange(0, num_inference_steps) * (nt // num_inference_steps)).round()
            _ts = _ts + c["steps_offset"]
        elif sp == "trailing":
            _ts = np.round(np.arange(nt, 0, -nt / num_inference_steps))[::-1].astype(np.int64) - 1
        else:
            raise ValueError(f"{sp} is not supported. Please make sure to choose one of 'linspace', 'leading' or 'trailing'.")
        if c["skip_prk_steps"]:
            self.prk_timesteps = np.array([])
            self.plms_timesteps = np.concatenate([_ts[:-1], _ts[-2:-1], _ts[-1:]])[::-1].copy()
        else:
            prk = np.array(_ts[-self.pndm_order:]).repeat(2) + np.tile(
                np.array([0, nt // num_inference_steps // 2]), self.pndm_order)
            self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = _ts[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self.ets = []
        self.counter = 0
        self.cur_model_output = None
        self.cur_sample = None

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True):
        if self.counter < len(self.prk_timesteps) and not self.config["skip_prk_steps"]:
            return self.step_prk(model_output, timestep, sample, return_dict)
        return self.step_plms(model_output, timestep, sample, return_dict)

    def _check_ready(self):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")

    def step_prk(self, model_output, timestep, sample, return_dict: bool = True):
        self._check_ready()
        timestep = int(timestep)
        ratio = self.config["num_train_timesteps"] // self.num_inference_steps
        diff_to_prev = 0 if self.counter % 2 else ratio // 2
        prev_t = timestep - diff_to_prev
        timestep = int(self.prk_timesteps[self.counter // 4 * 4])
        mo = model_output.float().contiguous()
        acc = self.cur_model_output
        if self.counter % 4 == 0:
            self.cur_model_output = hip.axpby_n([mo], [1 / 6]) if acc is None else hip.axpby_n([acc, mo], [1.0, 1 / 6])
            self.ets.append(mo)
            self.cur_sample = sample
        elif (self.counter - 1) % 4 == 0 or (self.counter - 2) % 4 == 0:
            self.cur_model_output = hip.axpby_n([acc, mo], [1.0, 1 / 3])
        else:
            mo = hip.axpby_n([acc, mo], [1.0, 1 / 6])
            self.cur_model_output = None
        cur_sample = self.cur_sample if self.cur_sample is not None else sample
        prev = self._get_prev_sample(cur_sample, timestep, prev_t, mo)
        self.counter += 1
        return (prev,) if not return_dict else SchedulerOutput(prev_sample=prev)

    def step_plms(self, model_output, timestep, sample, return_dict: bool = True):
        self._check_ready()
        if not self.config["skip_prk_steps"] and len(self.ets) < 3:
            raise ValueError(f"{self.__class__} can only be run AFTER scheduler has been run in 'prk' mode for at "
                             "least 12 iterations")
        timestep = int(timestep)
        ratio = self.config["num_train_timesteps"] // self.num_inference_steps
        prev_t = timestep - ratio
        mo = model_output.float().contiguous()
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(mo)
        else:
            prev_t = timestep
            timestep = timestep + ratio
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            mo = hip.axpby_n([mo, e[-1]], [0.5, 0.5])
            sample = self.cur_sample
            self.cur_sample = None
        elif len(e) == 2:
            mo = hip.axpby_n([e[-1], e[-2]], [3 / 2, -1 / 2])
        elif len(e) == 3:
            mo = hip.axpby_n([e[-1], e[-2], e[-3]], [23 / 12, -16 / 12, 5 / 12])
        else:
            mo = hip.axpby_n([e[-1], e[-2], e[-3], e[-4]], [55 / 24, -59 / 24, 37 / 24, -9 / 24])
        prev = self._get_prev_sample(sample, timestep, prev_t, mo)
        self.counter += 1
        return (prev,) if not return_dict else SchedulerOutput(prev_sample=prev)

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        """Formula (9) of PNDM (scheduling_pndm.py:407-448) as one fused linear combination."""
        a_t, a_p = self._alpha(int(timestep)), self._alpha(int(prev_timestep))
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        k = -(a_p - a_t) / denom
        x = sample.float().contiguous()
        if self.config["prediction_type"] == "v_prediction":
            # eps = sqrt(a_t) v + sqrt(b_t) x  ->  prev = (sample_coeff + k sqrt(b_t)) x + k sqrt(a_t) v
            return hip.axpby_n([x, model_output], [float(sample_coeff + k * b_t ** 0.5), float(k * a_t ** 0.5)])
        return hip.axpby_n([x, model_output], [float(sample_coeff), float(k)])


class UniPCMultistepScheduler(_SchedulerBase):
    """schedulers/scheduling_unipc_multistep.py (bh1/bh2, predict_x0, lower_order_final; what the reference's own
    scripts select: examples/brushnet/test_brushnet.py:158, train_brushnet_mirror.py:179).  Predictor (:455-582) and
    corrector (:584-719) are linear in {sample, last_sample, converted model outputs}: the host evaluates the scalar
    coefficients exactly like the reference (fp32 0-dim tensor arithmetic on lambda = log(alpha) - log(sigma), expm1)
    and each update is ONE fused mf_axpby_n launch."""

    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, solver_order=2, prediction_type="epsilon", thresholding=False,
                     dynamic_thresholding_ratio=0.995, sample_max_value=1.0, predict_x0=True, solver_type="bh2",
                     lower_order_final=True, disable_corrector=[], solver_p=None, use_karras_sigmas=False,
                     timestep_spacing="linspace", steps_offset=0, set_alpha_to_one=True)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c["thresholding"] or c["use_karras_sigmas"] or c["solver_p"] is not None or not c["predict_x0"]:
            raise NotImplementedError("thresholding / Karras sigmas / solver_p / noise-prediction UniPC are outside the MirrorFusion path")
        if c["solver_type"] not in ("bh1", "bh2"):
            raise NotImplementedError(f"{c['solver_type']} does is not implemented for {self.__class__}")
        if c["prediction_type"] not in ("epsilon", "v_prediction"):
            raise ValueError(f"prediction_type given as {c['prediction_type']} must be one of `epsilon` or `v_prediction`")
        if c["solver_order"] > 3:
            raise NotImplementedError("solver_order > 3")
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.model_outputs = [None] * c["solver_order"]
        self.lower_order_nums = 0
        self.last_sample = None
        self._step_index = None
        self.disable_corrector = list(c["disable_corrector"])

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int, device=None):
        c = self.config
        nt = c["num_train_timesteps"]
        if c["timestep_spacing"] == "linspace":
            ts = np.linspace(0, nt - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif c["timestep_spacing"] == "leading":
            ts = (np.arange(0, num_inference_steps + 1) * (nt // (num_inference_steps + 1))).round()[::-1][:-1].copy().astype(np.int64)
            ts += c["steps_offset"]
        elif c["timestep_spacing"] == "trailing":
            ts = np.arange(nt, 0, -nt / num_inference_steps).round().copy().astype(np.int64) - 1
        else:
            raise ValueError(f"{c['timestep_spacing']} is not supported. Please make sure to choose one of 'linspace', 'leading' or 'trailing'.")
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        last = ((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5
        new_sigmas = torch.from_numpy(np.concatenate([sigmas, [last]]).astype(np.float32))
        # the scalar coefficients depend only on (sigma table, step index, order): memoised across calls so that the
        # host arithmetic (0-dim tensor math + a small linear solve, ~0.9 ms per step) leaves the critical path
        if getattr(self, "_coef_cache", None) is None or self.sigmas.shape != new_sigmas.shape \
                or not torch.equal(self.sigmas, new_sigmas):
            self._coef_cache = {}
        self.sigmas = new_sigmas
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * c["solver_order"]
        self.lower_order_nums = 0
        self.last_sample = None
        self._step_index = None

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _coefs(self, order, idx_t, idx_s0, hist_offset):
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[idx_t])
        alpha_s0, sigma_s0 = self._alpha_sigma(self.sigmas[idx_s0])
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = (torch.log(alpha_t) - torch.log(sigma_t)) - lambda_s0
        rks = []
        for i in range(1, order):
            a_si, s_si = self._alpha_sigma(self.sigmas[self._step_index - (i + hist_offset)])
            rks.append((torch.log(a_si) - torch.log(s_si) - lambda_s0) / h)
        rks_t = torch.tensor([float(r) for r in rks] + [1.0])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.config["solver_type"] == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks_t, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, rks, torch.stack(R), torch.tensor([float(v) for v in b])

    def _index_for_timestep(self, timestep):
        cand = (self.timesteps == int(timestep)).nonzero()
        if len(cand) == 0:
            return len(self.timesteps) - 1
        return int(cand[1]) if len(cand) > 1 else int(cand[0])

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._step_index = self._index_for_timestep(timestep)
        c = self.config
        eps = model_output.float().contiguous()
        x = sample.float().contiguous()
        use_corrector = self._step_index > 0 and self._step_index - 1 not in self.disable_corrector \
            and self.last_sample is not None
        # convert_model_output (:385-453): x0 prediction, linear in (sample, model_output)
        a_c, s_c = self._alpha_sigma(self.sigmas[self._step_index])
        if c["prediction_type"] == "epsilon":
            m_t = hip.axpby_n([x, eps], [float(1 / a_c), float(-s_c / a_c)])
        else:
            m_t = hip.axpby_n([x, eps], [float(a_c), float(-s_c)])
        if use_corrector:                                                       # UniC (:584-719)
            order = self.this_order
            key = ("c", order, self._step_index)
            coefs = self._coef_cache.get(key)
            if coefs is None:
                alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, rks, R, b = self._coefs(order, self._step_index, self._step_index - 1, 1)
                rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
                k = -alpha_t * B_h
                coefs, cm0 = [float(sigma_t / sigma_s0), float(k * rhos[-1])], -alpha_t * h_phi_1 - k * rhos[-1]
                for i in range(1, order):
                    w = k * rhos[i - 1] / rks[i - 1]
                    coefs.append(float(w)); cm0 = cm0 - w
                coefs.append(float(cm0))
                self._coef_cache[key] = coefs
            terms = [self.last_sample, m_t] + [self.model_outputs[-(i + 1)] for i in range(1, order)] + [self.model_outputs[-1]]
            x = hip.axpby_n(terms, coefs)
        for i in range(c["solver_order"] - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = m_t
        this_order = min(c["solver_order"], len(self.timesteps) - self._step_index) if c["lower_order_final"] else c["solver_order"]
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = x
        order = self.this_order                                                 # UniP (:455-582)
        key = ("p", order, self._step_index)
        coefs = self._coef_cache.get(key)
        if coefs is None:
            alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, rks, R, b = self._coefs(order, self._step_index + 1, self._step_index, 0)
            coefs, cm0 = [float(sigma_t / sigma_s0)], -alpha_t * h_phi_1
            if order > 1:
                rhos_p = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
                k = -alpha_t * B_h
                for i in range(1, order):
                    w = k * rhos_p[i - 1] / rks[i - 1]
                    coefs.append(float(w)); cm0 = cm0 - w
            coefs.append(float(cm0))
            self._coef_cache[key] = coefs
        terms = [x] + [self.model_outputs[-(i + 1)] for i in range(1, order)] + [m_t]
        prev = hip.axpby_n(terms, coefs)
        if self.lower_order_nums < c["solver_order"]:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,) if not return_dict else SchedulerOutput(prev_sample=prev)

    def add_noise(self, original_samples, noise, timesteps):
        raise NotImplementedError("UniPC.add_noise is only used by the training script (SURVEY.md §8 f-2)")
