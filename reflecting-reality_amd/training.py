"""The reference's training step on the HIP path.

Mirrors examples/brushnet/train_brushnet_mirror.py: `MirrorFusionModel` (:836-888, BrushNet residuals injected
into the UNet, conditioning scale 1), `compute_snr` (src/diffusers/training_utils.py:50-73), the loss block
(:1407-1449: add_noise at per-sample timesteps, epsilon / v_prediction target, plain or min-SNR-weighted MSE), and the
step itself (:1459-1466): backward, `clip_grad_norm_(max_grad_norm)`, `AdamW.step()`, `zero_grad()`; plus the
`checkpoint-N/{brushnet,unet}` save / load hooks (:997-1069) and their rotation (:1474-1498).

Forward and backward run on the libmfhip kernels: the backward pass is a tape of hand-written vector-Jacobian products
(autograd.py) — there is no ATen autograd graph.  Parameters, gradients and the Adam moments live in flat fp32 arenas
(one launch of mf_adamw per model, one mf_sumsq for the norm; the clip coefficient never leaves the device); with more
than one rank the gradient arena is all-reduced in buckets over RCCL while the backward pass is still running
(distributed.GradBuckets).
"""
import json
import os
import shutil
from typing import List, Optional, Sequence

import torch

from . import autograd, hip, ops

__all__ = ["MirrorFusionModel", "compute_snr", "training_loss", "AdamW", "clip_grad_norm_", "train_step", "save_state",
           "load_state"]


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
        """train_brushnet_mirror.py:845-856: the children that hold trainable parameters."""
        return [m for m in (self.unet, self.brushnet) if m.training and m._requires_grad]

    def prepare_training(self, train_base_unet: bool = False):
        """:1072-1079: BrushNet trains; the UNet is frozen unless --train_base_unet (it still needs the training layout:
        the loss gradient reaches BrushNet's residuals through it)."""
        self.brushnet.prepare_training(requires_grad=True)
        self.unet.prepare_training(requires_grad=bool(train_base_unet))
        return self

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
                  snr_gamma: Optional[float] = None, _return_weights: bool = False):
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
    if _return_weights:
        return loss, pred, target, weights
    return loss, pred, target


# ---------------------------------------------------------------------------------------------------------------------
# optimizer, gradient clipping, the step
# ---------------------------------------------------------------------------------------------------------------------
class AdamW:
    """torch.optim.AdamW over the models' flat arenas (train_brushnet_mirror.py:1188-1200: lr 1e-5, betas 0.9 / 0.999,
    weight decay 1e-2, eps 1e-8; decay applies to every parameter, like the reference's single param group)."""

    def __init__(self, models: Sequence, lr: float = 1e-5, betas=(0.9, 0.999), weight_decay: float = 1e-2, eps: float = 1e-8):
        self.models = [m for m in models if m.training and m.flat_g is not None]
        if not self.models:
            raise ValueError("AdamW: no model with a gradient arena (call prepare_training(requires_grad=True) first)")
        self.lr, self.betas, self.weight_decay, self.eps = lr, tuple(betas), weight_decay, eps
        self.step_count = 0
        self.exp_avg = [torch.zeros(m.num_arena_floats(), dtype=torch.float32, device=m.device) for m in self.models]
        self.exp_avg_sq = [torch.zeros_like(t) for t in self.exp_avg]

    def zero_grad(self, set_to_none: bool = False, for_step: bool = False) -> None:
        """for_step (train_step / GraphedTrainStep): only the small accumulated parameters are cleared, the weights' first
        gradient of the step overwrites (models.zero_grad_for_step); otherwise the whole arena is."""
        for m in self.models:
            if for_step and FIRST_WRITE:
                m.zero_grad_for_step()
            else:
                m.flat_g.zero_()
                for prm, _ in getattr(m, "_plist", ()):
                    prm.fresh = False

    def finish_fresh(self) -> None:
        for m in self.models:
            m.finish_fresh()

    def step(self, grad_scale: Optional[torch.Tensor] = None) -> None:
        self.step_count += 1
        for m, ea, es in zip(self.models, self.exp_avg, self.exp_avg_sq):
            n = m.num_arena_floats()
            hip.adamw(m.flat_w[:n], m.flat_g[:n], ea, es, lr=self.lr, betas=self.betas, eps=self.eps,
                      weight_decay=self.weight_decay, step=self.step_count, grad_scale=grad_scale)
            m._weights_gen += 1            # derived layouts (dgrad weights, captured graphs) are stale now

    def state_dict(self) -> dict:
        # initial_lr (set by optimization.LambdaLR): the base a resumed LR schedule multiplies — `lr` alone is the DECAYED value
        return dict(step=self.step_count, lr=self.lr, initial_lr=getattr(self, "initial_lr", self.lr), betas=self.betas,
                    weight_decay=self.weight_decay, eps=self.eps,
                    exp_avg=[t.cpu() for t in self.exp_avg], exp_avg_sq=[t.cpu() for t in self.exp_avg_sq])

    def load_state_dict(self, sd: dict) -> None:
        """Restores the step count, the hyper-parameters and the moments.  NOT interchangeable with the reference's
        `optimizer.bin` (a torch.optim.AdamW state_dict keyed by parameter index, written by accelerate): the moments here are
        the flat arenas in the kernels' weight layout ([N][kh][kw][Cin_pad], fused time_emb_proj rows).  Model weights ARE
        interchangeable (checkpoint-N/{brushnet,unet}); accelerate's scheduler.bin / random_states are not written."""
        if not isinstance(sd, dict) or "exp_avg" not in sd or "step" not in sd:
            raise ValueError("optimizer.bin is not an mfhip AdamW state (a torch.optim / accelerate optimizer state cannot be "
                             "loaded: the moments are stored as flat arenas in the kernels' layout)")
        self.step_count = int(sd["step"])
        self.lr = float(sd.get("lr", self.lr))
        if "initial_lr" in sd:
            self.initial_lr = float(sd["initial_lr"])
        self.betas = tuple(sd.get("betas", self.betas))
        self.weight_decay = float(sd.get("weight_decay", self.weight_decay))
        self.eps = float(sd.get("eps", self.eps))
        for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(sd["exp_avg"]) + list(sd["exp_avg_sq"])):
            if dst.numel() != src.numel():
                raise ValueError("optimizer state does not match the models' arenas")
            dst.copy_(src)


class _ClipState:
    def __init__(self, device):
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=device)
        self.coef = torch.ones(1, dtype=torch.float32, device=device)
        self.norm = torch.zeros(1, dtype=torch.float32, device=device)


_clip_states: dict = {}
FIRST_WRITE = os.environ.get("MFHIP_ZERO_WHOLE_ARENA", "0") != "1"         # developer A/B: memset the whole gradient arena every step
DGRAD_PREFETCH = os.environ.get("MFHIP_NO_DGRAD_PREFETCH", "0") != "1"     # developer A/B


def clip_grad_norm_(models: Sequence, max_norm: float, loss_scale: float = 1.0):
    """torch.nn.utils.clip_grad_norm_ over every gradient arena (accelerator.clip_grad_norm_, :1463).  Returns
    (total_norm, coef): one-element DEVICE tensors — the gradients are not rescaled here, mf_adamw reads them times
    coef, and nothing synchronises with the host.  `loss_scale`: the arenas hold gradients x loss_scale (the norm and the
    coefficient account for it)."""
    models = [m for m in models if m.flat_g is not None]
    dev = models[0].device
    st = _clip_states.get(str(dev))
    if st is None:
        st = _clip_states[str(dev)] = _ClipState(dev)
    for i, m in enumerate(models):
        hip.sumsq(m.flat_g[: m.num_arena_floats()], st.sumsq, accumulate=i > 0)
    hip.clip_coef(st.sumsq, float(max_norm), st.coef, st.norm, unscale=1.0 / float(loss_scale))
    return st.norm, st.coef


def train_step(model: MirrorFusionModel, noise_scheduler, optimizer: AdamW, latents: torch.Tensor, noise: torch.Tensor,
               timesteps: torch.Tensor, encoder_hidden_states: torch.Tensor, conditioning_latents: torch.Tensor,
               snr_gamma: Optional[float] = None, max_grad_norm: float = 1.0, grad_sync=None, check_overflow: bool = True,
               gradient_accumulation_steps: int = 1, lr_scheduler=None):
    """One pass of the training loop's body (:1349, 1407-1466).  Returns (loss, grad_norm) as one-element device tensors.
    `grad_sync`: a distributed.GradBuckets when several ranks train data-parallel (the DDP wrap of :1267-1269).
    `check_overflow` (split precisions): read the range-guard flags after the backward pass and skip the optimizer step when
    an operand of the forward OR the backward pass left the fp16 range.
    `gradient_accumulation_steps` = G (--gradient_accumulation_steps, `accelerator.accumulate`, :1349): the gradients of G
    consecutive calls are summed in the arena (each loss scaled by 1 / G, accelerate's `backward`); only the G-th call
    exchanges gradients between ranks (DDP's no_sync on the others), clips, steps the optimizer and the learning-rate
    schedule and zeroes the arena — the other calls return (loss, None).
    `lr_scheduler`: an optimization.LambdaLR (optimization.get_scheduler, :1257); stepped once per rank per optimizer step,
    as accelerate's scheduler wrapper does (which is why the script scales its warm-up by num_processes, :1260)."""
    mods = model.get_trainable_modules()
    if not mods:
        raise RuntimeError("train_step: call model.prepare_training() first")
    accum = max(int(gradient_accumulation_steps), 1)
    micro = getattr(optimizer, "_micro", 0)
    sync_step = micro + 1 >= accum
    prec = model.brushnet.prec
    # bf16x1 runs the fp16 split flash attention only where the bf16 flash route does not apply (short sequences, MFHIP_NO_FLASH_BWD):
    # its guard — a host read-back — is armed BEFORE the step when the previous step launched such a kernel and AFTER it when this
    # step did although the previous one did not (the shapes changed between steps: ADVICE r5).  Several ranks always arm it: the
    # guard holds a collective there, and a decision taken from rank-local launch counts could differ between ranks.
    world = int(getattr(grad_sync, "world", 1) or 1) if grad_sync is not None else 1
    guard = check_overflow and (prec.code == hip.MF_F16X3 or (prec.code == hip.MF_BF16X1 and (world > 1 or getattr(model, "_split_kernels_per_step", 1) > 0)))
    if micro == 0:
        optimizer._window_split0 = ops.SPLIT_KERNEL_CALLS
        optimizer.zero_grad(for_step=True)
        if guard:
            hip.split_overflow(reset=True)     # flags raised by earlier, unrelated work do not count against this step; the
                                               # forward pass that follows DOES (the flags are sticky until the read below)
    tape = autograd.Tape(prec.tape_code)
    if grad_sync is not None and sync_step:
        grad_sync.begin(tape)
    # The data-gradient layouts of the weights that train (transposed, tap-flipped, split: ~540 small launches) depend only on
    # the weights the optimizer left behind: rebuild them on a side stream under the forward pass instead of inside the backward
    # (the list is what the previous step's backward had to rebuild; the first step builds them lazily).
    dgrad_ready = None
    arena_key = tuple(m.flat_w.data_ptr() for m in (model.brushnet, model.unet) if getattr(m, "flat_w", None) is not None)
    if getattr(model, "_dgrad_prefetch_key", None) != arena_key:        # prepare_training() rebuilt an arena: the list is stale
        model._dgrad_prefetch, model._dgrad_prefetch_key = None, arena_key
    prefetch = getattr(model, "_dgrad_prefetch", None)
    if prefetch and DGRAD_PREFETCH:
        dev = mods[0].device
        side = getattr(model, "_side_stream", None)
        if side is None:
            side = model._side_stream = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for cw in prefetch:
                autograd._dgrad_weight(cw, prec.tape_code)
            dgrad_ready = side.record_event()
    ops.TAPE = tape
    try:
        loss, pred, target, weights = training_loss(model, noise_scheduler, latents, noise, timesteps, encoder_hidden_states,
                                                    conditioning_latents, snr_gamma, _return_weights=True)
    finally:
        ops.TAPE = None
    d_pred = hip.mse_grad(pred.contiguous(), target.contiguous(), weights)
    # Split precision (f16x3) multiplies 16-bit halves: a gradient of 1e-5 would keep only the 11 bits of its high half
    # (its low half falls below the fp16 subnormals).  A power-of-two loss scale that brings d loss / d pred (= 2 (pred -
    # target) w / (n B)) to O(1) keeps every backward GEMM operand in the range where hi + lo carries 22 bits; it is
    # exact in fp32 and undone inside the clip coefficient.  fp32 MFMA needs none.
    scale = 1.0
    if prec.split or prec.code == hip.MF_BF16X1:
        # (bf16x1 too: its d = 8 / 40 attention runs forward and backward on fp16 halves (ops.attention), and an unscaled dO of
        # 1e-6 sits in the fp16 subnormals, where the toward-zero split loses it — found by the full-size reference gradients of
        # round 4: every gradient behind the 64 x 64 attention layers came out 0.3-0.5 % short.  bf16 rounding is invariant under
        # a power-of-two scale, so the mode's own arithmetic does not change.)
        scale = float(2 ** int(pred.numel() - 1).bit_length())
        # one scale per accumulation window: the arena SUMS the micro-batches' scaled gradients and only the last
        # call's scale is undone in the clip coefficient, so a ragged micro-batch whose element count crosses a power of two
        # must not change it — the window keeps the scale of its first micro-batch
        if micro == 0:
            optimizer._window_scale = scale
        else:
            scale = float(getattr(optimizer, "_window_scale", scale))
    if scale != 1.0 or accum > 1:
        d_pred = hip.axpby_n([d_pred], [scale / accum], out=d_pred)      # 1 / G: accelerator.backward under accumulate()
    model.loss_scale = scale
    tape.add(pred, d_pred)
    if dgrad_ready is not None:
        torch.cuda.current_stream(mods[0].device).wait_event(dgrad_ready)
    tape.backward()
    optimizer.finish_fresh()
    if tape.dgrad_rebuilt:
        known = {id(c) for c in (prefetch or [])}
        model._dgrad_prefetch = list(prefetch or []) + [c for c in tape.dgrad_rebuilt if id(c) not in known]
    if not sync_step:
        optimizer._micro = micro + 1
        return loss, None
    optimizer._micro = 0
    if grad_sync is not None:
        grad_sync.finish()
    norm, coef = clip_grad_norm_(mods, max_grad_norm, loss_scale=scale)
    model._split_kernels_per_step = ops.SPLIT_KERNEL_CALLS - getattr(optimizer, "_window_split0", ops.SPLIT_KERNEL_CALLS)
    if not guard and check_overflow and prec.code == hip.MF_BF16X1 and model._split_kernels_per_step > 0:
        guard = True        # armed late: the flags were not reset before this window, so work before it may raise them too — at
                            # worst one step skipped that need not have been, never a saturated gradient applied
    if guard:
        # fp16 halves saturate above 65504 without producing inf / NaN (mfhip.h, mf_split_overflow): a step whose forward or
        # (loss-scaled) backward operands left that range has silently wrong gradients — it is SKIPPED, like a GradScaler
        # step with inf gradients.  One 12-byte read-back per step; every rank skips together (the flag is all-reduced).
        raised = hip.split_overflow(reset=True)
        if grad_sync is not None and getattr(grad_sync, "world", 1) > 1:
            from . import distributed as D
            raised = int(D.max_over_ranks(float(raised), device=mods[0].device))
        if raised:
            import warnings
            model.overflow_steps = getattr(model, "overflow_steps", 0) + 1
            warnings.warn(f"train_step: a split-precision operand exceeded the fp16 range (flags {raised:#x}); optimizer step "
                          f"skipped ({model.overflow_steps} so far) — train with precision 'fp32' if this persists")
            optimizer.zero_grad()
            return loss, norm
    optimizer.step(grad_scale=coef)
    if lr_scheduler is not None:
        for _ in range(max(int(getattr(grad_sync, "world", 1) or 1), 1) if grad_sync is not None else 1):
            lr_scheduler.step()
    return loss, norm


class GraphedTrainStep:
    """train_step() with zero_grad + forward + loss + backward + clip captured ONCE in a hipGraph and replayed every step.

    A training step is ~5000 kernel launches; each costs ~25 us of Python + ctypes on the host, so the host's launch rate
    (not the GPU) bounds the bf16x1 step and eats into the f16x3 one.  What depends on host values stays outside the graph
    and runs eagerly around the replay: the noising of the latents (per-sample schedule coefficients looked up by timestep on
    the host, :1416), the SNR weights (:1437-1449), the range-guard read-back, and the optimizer (its bias corrections are
    host scalars).  Inputs are copied into static device buffers; shapes are fixed at the first call.  The first `warmup`
    calls run the eager train_step (they tune GEMM tiles, build the frozen network's derived weights and learn which
    data-gradient layouts to prefetch); the next call captures.  With a GradBuckets gradient sync (several ranks) the step is
    captured as a CHAIN of graphs cut at every point where the backward pass releases a gradient bucket; at replay the bucket's
    RCCL all-reduce is issued eagerly on the side stream between two segments (RCCL calls are never captured), the last segment
    (mean over ranks, clip) runs once every exchange has been awaited.  The arithmetic and its order are the eager step's: weights are
    bit-identical after the same number of steps (tests/test_training_gpu.py).  The returned (loss, grad_norm) tensors live
    in the graph's memory pool and are overwritten by the next call: read them (float(), .clone()) before stepping again.
    Memory: the graph keeps one step's activations resident between calls (they are reused, not reallocated)."""

    def __init__(self, model: MirrorFusionModel, noise_scheduler, optimizer: AdamW, snr_gamma: Optional[float] = None,
                 max_grad_norm: float = 1.0, check_overflow: bool = True, warmup: int = 2, lr_scheduler=None, grad_sync=None):
        self.model, self.ns, self.opt = model, noise_scheduler, optimizer
        # a distributed.GradBuckets: the step is then captured as a CHAIN of graphs cut wherever the backward pass releases a
        # gradient bucket, and the bucket's all-reduce is issued eagerly between two replays (RCCL calls are never captured)
        self.grad_sync = grad_sync
        self.segments = None
        self.snr_gamma, self.max_grad_norm, self.check_overflow = snr_gamma, max_grad_norm, check_overflow
        self.warmup, self.calls, self.graph = max(int(warmup), 1), 0, None
        self.lr_scheduler = lr_scheduler
        self._arena_key = None
        self.split_kernels = 1          # fp16-split attention launches in the captured step (bf16x1: arms the range guard; set at capture)

    def _arenas(self):
        """What the captured kernels point into: the weight / gradient arenas of both networks."""
        return tuple(t.data_ptr() for m in (self.model.brushnet, self.model.unet)
                     for t in (getattr(m, "flat_w", None), getattr(m, "flat_g", None)) if t is not None)

    # -- eager, host-dependent staging ----------------------------------------------------------------------------------
    def _stage(self, latents, noise, timesteps, ehs, cond):
        dev = self.noisy.device
        self.noisy.copy_(self.ns.add_noise(latents, noise, timesteps))                                   # :1416
        ptype = self.ns.config["prediction_type"]
        if ptype == "epsilon":                                                                           # :1427-1432
            self.target.copy_(noise.to(dev, torch.float32))
        elif ptype == "v_prediction":
            self.target.copy_(self.ns.get_velocity(latents, noise, timesteps))
        else:
            raise ValueError(f"Unknown prediction type {ptype}")
        self.t_dev.copy_(hip.h2d(timesteps.reshape(-1).float().contiguous(), dev))
        self.ehs.copy_(ehs.to(dev, torch.float32))
        self.cond.copy_(cond.to(dev, torch.float32))
        if self.snr_gamma is not None:                                                                   # :1437-1449
            snr = compute_snr(self.ns, timesteps)
            w = torch.stack([snr, self.snr_gamma * torch.ones_like(snr)], dim=1).min(dim=1)[0]
            w = w / snr if ptype == "epsilon" else w / (snr + 1)
            self.weights.copy_(hip.h2d(w.float().contiguous(), dev))

    def _body(self):
        model, prec = self.model, self.model.brushnet.prec
        split_calls0 = ops.SPLIT_KERNEL_CALLS
        mods = model.get_trainable_modules()
        self.opt.zero_grad(for_step=True)
        tape = autograd.Tape(prec.tape_code)
        if self.grad_sync is not None:
            self.grad_sync.begin(tape)
        dgrad_ready = None
        prefetch = getattr(model, "_dgrad_prefetch", None)
        cur = torch.cuda.current_stream(mods[0].device)
        if prefetch and DGRAD_PREFETCH:
            side = getattr(model, "_side_stream", None)
            if side is None:
                side = model._side_stream = torch.cuda.Stream(device=mods[0].device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for cw in prefetch:
                    autograd._dgrad_weight(cw, prec.tape_code)
                dgrad_ready = side.record_event()
        ops.TAPE = tape
        try:
            pred = model(self.noisy, self.t_dev, self.ehs, self.cond)
            loss, _ = hip.mse_loss(pred.float(), self.target, self.weights if self.snr_gamma is not None else None)
        finally:
            ops.TAPE = None
        d_pred = hip.mse_grad(pred.contiguous(), self.target, self.weights if self.snr_gamma is not None else None)
        scale = 1.0
        if prec.split or prec.code == hip.MF_BF16X1:
            scale = float(2 ** int(pred.numel() - 1).bit_length())
            d_pred = hip.axpby_n([d_pred], [scale], out=d_pred)
        model.loss_scale = scale
        tape.add(pred, d_pred)
        if dgrad_ready is not None:
            cur.wait_event(dgrad_ready)
        tape.backward()
        self.opt.finish_fresh()
        if self.grad_sync is not None and self.segments is not None:
            gs = self.grad_sync
            gs.flush()                      # (capturing: reported to _cut like the buckets released during the backward pass)
            self._cut(None, None)           # everything issued so far ends a segment; the exchanges are awaited between replays
            gs.scale()
        norm, coef = clip_grad_norm_(mods, self.max_grad_norm, loss_scale=scale)
        self.split_kernels = ops.SPLIT_KERNEL_CALLS - split_calls0
        return loss, norm, coef

    # -- a chain of graphs with the gradient exchange between them ----------------------------------------------------------
    def _cut(self, pi, b):
        """Capture hook of GradBuckets: end the graph segment being captured here, remember that bucket (pi, b) is sent after
        it (pi None: wait for every exchange instead), and begin the next segment in the same memory pool."""
        self._cur.capture_end()
        self.segments.append((self._cur, (pi, b)))
        self._cur = torch.cuda.CUDAGraph()
        self._cur.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def _capture_chain(self):
        dev = self.model.get_trainable_modules()[0].device
        gs = self.grad_sync
        self.segments, self._pool = [], torch.cuda.graph_pool_handle()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        gs.capture_hook = self._cut
        try:
            with torch.cuda.stream(side):
                self._cur = torch.cuda.CUDAGraph()
                # thread_local: an initialised process group's watchdog thread polls its events with hipEventQuery while this thread
                # captures; in the default "global" mode that call invalidates the capture (SIGABRT from the watchdog)
                self._cur.capture_begin(pool=self._pool, capture_error_mode="thread_local")
                try:
                    out = self._body()
                finally:
                    self._cur.capture_end()
                self.segments.append((self._cur, None))
        finally:
            gs.capture_hook = None
        torch.cuda.current_stream(dev).wait_stream(side)
        return out

    def _replay_chain(self):
        gs = self.grad_sync
        gs.begin(autograd.Tape(self.model.brushnet.prec.tape_code))       # fresh per-step exchange state (handles, sent flags)
        for graph, after in self.segments:
            graph.replay()
            if after is None:
                continue
            pi, b = after
            if pi is None:
                gs.wait()
            else:
                gs._state[pi]["sent"][b] = True
                gs.send_bucket(pi, b)

    def __call__(self, latents, noise, timesteps, encoder_hidden_states, conditioning_latents):
        model = self.model
        mods = model.get_trainable_modules()
        if not mods:
            raise RuntimeError("GraphedTrainStep: call model.prepare_training() first")
        if self.graph is None and self.calls < self.warmup:
            self.calls += 1
            return train_step(model, self.ns, self.opt, latents, noise, timesteps, encoder_hidden_states, conditioning_latents,
                              snr_gamma=self.snr_gamma, max_grad_norm=self.max_grad_norm, check_overflow=self.check_overflow,
                              lr_scheduler=self.lr_scheduler, grad_sync=self.grad_sync)
        dev = mods[0].device
        prec = model.brushnet.prec
        if self.graph is not None and self._arenas() != self._arena_key:
            # prepare_training() rebuilt an arena since the capture: the graph's kernels point at freed memory
            self.graph, self.segments = None, None
        if self.graph is None:
            f32 = dict(dtype=torch.float32, device=dev)
            self.noisy, self.target = torch.empty(latents.shape, **f32), torch.empty(latents.shape, **f32)
            self.t_dev = torch.empty(latents.shape[0], **f32)
            self.ehs, self.cond = torch.empty(encoder_hidden_states.shape, **f32), torch.empty(conditioning_latents.shape, **f32)
            self.weights = torch.ones(latents.shape[0], **f32)
            self._stage(latents, noise, timesteps, encoder_hidden_states, conditioning_latents)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            # The per-step re-layouts of the weights that train (ConvWeight.operand: split-pack; autograd._dgrad_weight:
            # transpose + split-pack) must be IN the graph, whatever their generation stamps say: a warm-up step skipped by the
            # range guard leaves the stamps current, and a graph captured then would replay on frozen copies while AdamW keeps
            # updating the arena.  The token makes each of them rebuild once during this capture.
            ops.CAPTURE_TOKEN = object()
            try:
                if self.grad_sync is not None and (self.grad_sync.world > 1 or self.grad_sync.force):
                    self.loss, self.norm, self.coef = self._capture_chain()
                else:
                    mode = "thread_local" if torch.distributed.is_available() and torch.distributed.is_initialized() else "global"
                    with torch.cuda.graph(self.graph, capture_error_mode=mode):
                        self.loss, self.norm, self.coef = self._body()
            finally:
                ops.CAPTURE_TOKEN = None
            self._arena_key = self._arenas()
        else:
            for t, ref in ((latents, self.noisy), (encoder_hidden_states, self.ehs), (conditioning_latents, self.cond)):
                if tuple(t.shape) != tuple(ref.shape):
                    raise ValueError(f"GraphedTrainStep: input shape {tuple(t.shape)} != the captured {tuple(ref.shape)}")
            self._stage(latents, noise, timesteps, encoder_hidden_states, conditioning_latents)
        guard = self.check_overflow and (prec.code == hip.MF_F16X3 or (prec.code == hip.MF_BF16X1 and self.split_kernels > 0))
        if guard:
            hip.split_overflow(reset=True)
        if self.segments is not None:
            self._replay_chain()
        else:
            self.graph.replay()
        self.calls += 1
        raised = hip.split_overflow(reset=True) if guard else 0
        if guard and self.grad_sync is not None and getattr(self.grad_sync, "world", 1) > 1:
            from . import distributed as D
            raised = int(D.max_over_ranks(float(raised), device=dev))      # every rank skips together
        if raised:
            import warnings
            model.overflow_steps = getattr(model, "overflow_steps", 0) + 1
            warnings.warn(f"GraphedTrainStep: a split-precision operand exceeded the fp16 range; optimizer step skipped "
                          f"({model.overflow_steps} so far) — train with precision 'fp32' if this persists")
            self.opt.zero_grad()
            return self.loss, self.norm
        self.opt.step(grad_scale=self.coef)
        if self.lr_scheduler is not None:
            for _ in range(max(int(getattr(self.grad_sync, "world", 1) or 1), 1) if self.grad_sync is not None else 1):
                self.lr_scheduler.step()
        return self.loss, self.norm


# ---------------------------------------------------------------------------------------------------------------------
# checkpoints: accelerator.save_state with the script's hooks (:997-1069) and its rotation (:1474-1498)
# ---------------------------------------------------------------------------------------------------------------------
def save_state(output_dir: str, global_step: int, model: MirrorFusionModel, optimizer: Optional[AdamW] = None,
               checkpoints_total_limit: Optional[int] = None, is_main_process: bool = True, lr_scheduler=None) -> Optional[str]:
    """Writes `output_dir/checkpoint-<global_step>/{brushnet,unet}` (config.json + diffusion_pytorch_model.safetensors in
    the reference's layout; `unet` only when it trains), `optimizer.bin` and — when a schedule is passed — `scheduler.bin`
    (accelerate saves the prepared lr_scheduler with the state, train_brushnet_mirror.py:1267-1269, 1496); before that, removes
    the oldest checkpoints so that at most `checkpoints_total_limit` remain — the script's order of operations."""
    if not is_main_process:
        return None
    os.makedirs(output_dir, exist_ok=True)
    if checkpoints_total_limit is not None:
        cps = sorted((d for d in os.listdir(output_dir) if d.startswith("checkpoint")), key=lambda x: int(x.split("-")[1]))
        if len(cps) >= checkpoints_total_limit:
            for d in cps[: len(cps) - checkpoints_total_limit + 1]:
                shutil.rmtree(os.path.join(output_dir, d))
    path = os.path.join(output_dir, f"checkpoint-{global_step}")
    os.makedirs(path, exist_ok=True)
    for m in model.get_trainable_modules():
        m.save_pretrained(os.path.join(path, "brushnet" if m is model.brushnet else "unet"))
    if optimizer is not None:
        torch.save(optimizer.state_dict(), os.path.join(path, "optimizer.bin"))
    if lr_scheduler is not None:
        torch.save(lr_scheduler.state_dict(), os.path.join(path, "scheduler.bin"))
    with open(os.path.join(path, "trainer_state.json"), "w") as f:
        json.dump(dict(global_step=global_step), f)
    return path


def load_state(path: str, model: MirrorFusionModel, optimizer: Optional[AdamW] = None, lr_scheduler=None) -> int:
    """load_model_hook (:1035-1066): each trainable module reloads its weights from its sub-folder; returns the step.
    `lr_scheduler` (built BEFORE this call, like the script builds and prepares it before accelerator.load_state, :1232-1300)
    continues from the saved epoch: without it a non-constant schedule would replay its warm-up after a resume."""
    from safetensors.torch import load_file
    for m in model.get_trainable_modules():
        sub = "brushnet" if m is model.brushnet else "unet"
        m.load_state_dict(load_file(os.path.join(path, sub, m.weights_name)))
    if optimizer is not None:
        if not os.path.exists(os.path.join(path, "optimizer.bin")):
            import warnings
            warnings.warn(f"load_state: {path} has no optimizer.bin — the weights are restored, AdamW restarts from zero moments "
                          "(the next steps will differ from the uninterrupted run)")
        else:
            if [m for m in model.get_trainable_modules()] != optimizer.models:
                optimizer.models = model.get_trainable_modules()
            optimizer.load_state_dict(torch.load(os.path.join(path, "optimizer.bin")))
    if lr_scheduler is not None:
        sp = os.path.join(path, "scheduler.bin")
        if os.path.exists(sp):
            lr_scheduler.load_state_dict(torch.load(sp))
        else:
            import warnings
            warnings.warn(f"load_state: {path} has no scheduler.bin — the LR schedule restarts from its first step "
                          "(checkpoints written before round 5, or without lr_scheduler=, do not carry it)")
    with open(os.path.join(path, "trainer_state.json")) as f:
        return int(json.load(f)["global_step"])


def latest_checkpoint(output_dir: str) -> Optional[str]:
    """--resume_from_checkpoint latest (:1285-1290)."""
    if not os.path.isdir(output_dir):
        return None
    cps = sorted((d for d in os.listdir(output_dir) if d.startswith("checkpoint")), key=lambda x: int(x.split("-")[1]))
    return os.path.join(output_dir, cps[-1]) if cps else None
