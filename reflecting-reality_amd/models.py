"""BrushNetModel, UNet2DConditionModel and AutoencoderKL with the reference call surface, running on
the libmfhip kernels.

Reference: MirrorFusion/src/diffusers/models/brushnet.py (BrushNetModel.forward :678-925, from_unet
:452-530), models/unets/unet_2d_condition.py (:1039-1348, BrushNet injection :1202-1324),
models/unets/unet_2d_blocks.py, models/resnet.py:329-405, models/transformers/transformer_2d.py:257-469,
models/attention.py:291-412, models/autoencoders/{autoencoder_kl,vae}.py.  State-dict keys, config.json
fields and the forward signatures are the reference's; the arithmetic is not ATen: every op is a HIP
kernel behind include/mfhip.h, and nothing here falls back to PyTorch math.

Data layout: activations live as NHWC.  Tensors crossing the public API keep the reference's logical
NCHW shape but are channels-last *views* of that memory (``nhwc.permute(0, 3, 1, 2)``), so the 28
BrushNet residuals flow into the UNet without a single layout copy.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import autograd, hip, ops
from .autograd import Param
from .ops import ConvWeight, Precision

F32 = torch.float32


def _scoped_tune_ctx(fn):
    """hip.TUNE_CTX (the position tag of the tuner's keys) is a module global the forwards set as they go: whatever a forward leaves
    behind — also when it raises half-way — must not tag the NEXT model's lookups.  Restores the caller's value on every exit;
    with `fresh` the body starts untagged (the VAE: its keys have no position)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        prev = hip.TUNE_CTX
        try:
            return fn(*args, **kwargs)
        finally:
            hip.TUNE_CTX = prev
    return wrapped


class FrozenConfig(dict):
    """config.json view with attribute access (reference: configuration_utils.FrozenDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


@dataclass
class BrushNetOutput:
    up_block_res_samples: List[torch.Tensor]
    down_block_res_samples: List[torch.Tensor]
    mid_block_res_sample: torch.Tensor


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


@dataclass
class DecoderOutput:
    sample: torch.Tensor


@dataclass
class AutoencoderKLOutput:
    latent_dist: "DiagonalGaussianDistribution"


def _as_tuple(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


def to_nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
    return x_nhwc.permute(0, 3, 1, 2)


def from_nchw(x: torch.Tensor, prec: Precision, c_pad: Optional[int] = None) -> torch.Tensor:
    """NCHW tensor (any memory format) -> contiguous NHWC in the activation dtype (zero-copy when the
    tensor already is a channels-last view of the right dtype)."""
    b, c, h, w = x.shape
    cp = c_pad or c
    nhwc = x.permute(0, 2, 3, 1)
    if nhwc.is_contiguous() and x.dtype == prec.act and cp == c:
        return nhwc
    if x.dtype != F32 or not x.is_contiguous():
        x = x.contiguous().float() if not nhwc.is_contiguous() else nhwc.float().permute(0, 3, 1, 2).contiguous()
    return hip.pack_nhwc(x, None, cp, prec.act)


class HipModel:
    """Minimal ModelMixin/ConfigMixin counterpart (modeling_utils.py, configuration_utils.py)."""

    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"
    _class_name = "HipModel"
    training = False            # class defaults: objects assembled piecewise (tests) are inference models
    flat_w = flat_g = None
    _weights_gen = 0

    def __init__(self, config: Dict[str, Any], precision: Union[str, Precision, torch.dtype] = "bf16",
                 device: Union[str, torch.device] = "cuda"):
        self.config = FrozenConfig(config)
        self.prec = Precision.get(precision)
        self.device = torch.device(device)
        self._src: Optional[Dict[str, torch.Tensor]] = None   # fp32 master copy (CPU), for save_pretrained
        self._ready = False
        self._weights_gen = 0       # bumped whenever the device weights are rebuilt (captured hipGraphs key on it)
        # training layout (prepare_training): every parameter is a view of flat fp32 arenas — master weights in the
        # kernels' layout, gradients, and (owned by training.AdamW) the Adam moments
        self.training = False
        self._requires_grad = True
        self.flat_w: Optional[torch.Tensor] = None
        self.flat_g: Optional[torch.Tensor] = None
        self._arena_used = 0
        self._pmap: "OrderedDict[str, tuple]" = OrderedDict()
        self.gradient_checkpointing = False

    # -- dtype / device surface the callers touch ----------------------------------------------
    @property
    def dtype(self) -> torch.dtype:
        return self.prec.act

    def to(self, *args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (str, torch.device)) and torch.device(a) != self.device:
                self.device = torch.device(a)
                if self._src is not None:
                    self.load_state_dict(self._src)
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag: bool = True):
        """Whether prepare_training() gives this model a gradient arena (train_brushnet_mirror.py:1072-1079: the UNet is
        frozen unless --train_base_unet, BrushNet trains)."""
        if self.training and flag != self._requires_grad:
            self._requires_grad = bool(flag)
            self.prepare_training()
        self._requires_grad = bool(flag)
        return self

    # -- parameters -----------------------------------------------------------------------------
    def param_shapes(self) -> "OrderedDict[str, Tuple[int, ...]]":
        raise NotImplementedError

    def state_dict(self) -> Dict[str, torch.Tensor]:
        if self.training:
            return self._export(self.flat_w)
        if self._src is None:
            raise RuntimeError("no parameters loaded")
        return dict(self._src)

    def grad_state_dict(self) -> Dict[str, torch.Tensor]:
        """The gradient arena in the reference's parameter layout (what `p.grad` holds there)."""
        if not (self.training and self.flat_g is not None):
            raise RuntimeError("no gradients: call prepare_training() with requires_grad")
        return self._export(self.flat_g)

    # -- training layout ------------------------------------------------------------------------------
    def prepare_training(self, requires_grad: Optional[bool] = None):
        """Rebuild the device parameters as views of one flat fp32 arena in the kernels' layout (raw fp32 weights — the
        optimizer updates them in place every step, so nothing is pre-split or fused) plus, when the model trains, a
        gradient arena of the same layout.  Precision must keep fp32 storage ('fp32', 'f16x3', or 'bf16x1': the arithmetic
        of the reference's --mixed_precision=bf16)."""
        if self.prec.act != F32:
            raise NotImplementedError("training runs with fp32 master weights and activations: build the model with "
                                      "precision='fp32', 'f16x3' or 'bf16x1' (bf16 products, fp32 storage)")
        if requires_grad is not None:
            self._requires_grad = bool(requires_grad)
        src = self.state_dict() if (self.training or self._src is not None) else None
        if src is None:
            raise RuntimeError("no parameters loaded")
        cap = 0
        for shp in self.param_shapes().values():
            cap += (int(torch.Size(shp).numel()) // shp[1] * ((shp[1] + 7) // 8 * 8) if len(shp) == 4 else int(torch.Size(shp).numel())) + 24
        self.flat_w = torch.zeros(cap, dtype=F32, device=self.device)
        self.flat_g = torch.zeros(cap, dtype=F32, device=self.device) if self._requires_grad else None
        self._arena_used = 0
        self._pmap = OrderedDict()
        self._plist = []               # (Param, arena offset) in allocation order
        self._fw_params = []           # conv / linear weights of _conv_param: one ConvWeight each, whole [N][K] written by its wgrad
        self._zero_tbl = None
        self.training = True
        self._src = None
        self._prepare({k: v.to("cpu", F32) for k, v in src.items()})
        self._weights_gen += 1
        self._ready = True
        if self.flat_g is not None:
            self._zero_table()                 # built here (host -> device copies): never inside a graph capture
        return self

    def _alloc(self, name: str, t: torch.Tensor, meta: tuple) -> Param:
        """Bump-allocate t (kernel layout, fp32) in the arena; 32-byte aligned, so that the same offsets are 16-byte aligned in the
        bf16 twin arena (weights_bf16)."""
        n = t.numel()
        a = self._arena_used
        if a + n > self.flat_w.numel():
            raise RuntimeError("training arena overflow")
        w = self.flat_w[a:a + n].view(t.shape)
        w.copy_(t.to(self.device, F32))
        g = self.flat_g[a:a + n].view(t.shape) if self.flat_g is not None else None
        self._arena_used = a + (n + 7) // 8 * 8
        p = Param(name, w, g)
        self._pmap[name] = (a, tuple(t.shape), meta)
        self._plist.append((p, a))
        return p

    def zero_grad_for_step(self) -> None:
        """optimizer.zero_grad() (train_brushnet_mirror.py:1466) without clearing the whole arena: the conv / linear weights — all
        but a few thousand of its floats — are marked `fresh`, so that their first mf_conv_wgrad of the step WRITES the gradient
        (autograd._conv_grads), and only the small accumulated parameters (biases, norm scales / shifts) are zeroed, in one launch
        (mf_zero_ranges).  finish_fresh() clears whatever weight no backward pass reached."""
        if self.flat_g is None:
            return
        if not hasattr(self, "_fw_params"):          # a model with its own arena set-up: clear everything
            self.flat_g.zero_()
            return
        tbl = self._zero_table()
        if tbl[1].numel():
            hip.zero_ranges(self.flat_g, tbl[1], tbl[2])
        for prm in tbl[3]:
            prm.fresh = True

    def _zero_table(self):
        """(arena address, offsets, lengths of the parameters zero_grad_for_step clears, the weights it marks fresh instead)."""
        tbl = self._zero_tbl
        if tbl is None or tbl[0] != self.flat_g.data_ptr():
            conv = {id(prm): prm for prm in self._fw_params}
            offs, lens = [], []
            chunk = 16384                     # floats per block of mf_zero_ranges: a fused matrix (temb_proj: 13 M floats) spreads over the chip
            for prm, a in self._plist:
                if prm.grad is not None and id(prm) not in conv:
                    n = prm.grad.numel()
                    for o in range(0, n, chunk):
                        offs.append(a + o)
                        lens.append(min(chunk, n - o))
            dev = self.flat_g.device
            tbl = self._zero_tbl = (self.flat_g.data_ptr(), torch.tensor(offs, dtype=torch.int64, device=dev),
                                    torch.tensor(lens, dtype=torch.int64, device=dev), list(conv.values()))
        return tbl

    def finish_fresh(self) -> None:
        """After the backward pass: a weight still marked fresh received no gradient this step — clear it (it holds stale values)."""
        tbl = getattr(self, "_zero_tbl", None)
        if tbl is None:
            return
        for prm in tbl[3]:
            if prm.fresh:
                prm.grad.zero_()
                prm.fresh = False

    def num_arena_floats(self) -> int:
        return self._arena_used

    def weights_bf16(self) -> torch.Tensor:
        """The bf16 twin of the weight arena (same offsets), refreshed by ONE mf_cast_bf16 launch per weight generation — per
        optimizer step for a network that trains, once for a frozen one: the forward GEMM operands of the bf16x1 mode on
        pre-rounded copies (ops.ConvWeight.operand_bf16) are views of it."""
        n = self._arena_used
        tok = ops.CAPTURE_TOKEN            # capturing a training graph: the refresh must be IN the graph (training.GraphedTrainStep)
        recapture = tok is not None and self._requires_grad and getattr(self, "_w16_tok", None) is not tok
        w16 = getattr(self, "flat_w16", None)
        if w16 is None or w16.numel() != self.flat_w.numel() or getattr(self, "_w16_src", None) != self.flat_w.data_ptr():
            w16 = self.flat_w16 = torch.empty(self.flat_w.numel(), dtype=torch.bfloat16, device=self.device)
            self._w16_gen, self._w16_src = None, self.flat_w.data_ptr()
        if self._w16_gen != self._weights_gen or recapture:
            hip.cast_bf16(self.flat_w[:n], out=w16[:n])
            self._w16_gen, self._w16_tok = self._weights_gen, tok
        return w16

    def _export(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        host = flat[: self._arena_used].cpu()
        for name, (a, shape, meta) in self._pmap.items():
            t = host[a:a + int(torch.Size(shape).numel())].view(shape)
            if meta[0] == "conv":                      # [N][kh][kw][cin_pad] -> [N, cin, kh, kw]
                _, n, cin, kh, kw, cp, is_linear = meta
                t = t.view(n, kh, kw, cp)[..., :cin].permute(0, 3, 1, 2).contiguous()
                out[name] = t.view(n, cin) if is_linear else t
            elif meta[0] == "rows":                    # row blocks of one fused matrix: (names, row counts)
                off = 0
                for nm, rows in zip(meta[1], meta[2]):
                    out[nm] = t[off:off + rows].clone()
                    off += rows
            elif meta[0] == "pad_rows":                # leading rows of a row-padded matrix
                _, nm, rows, inner = meta
                tt = t[:rows]
                if inner is not None:
                    _, n, cin, kh, kw, cp, is_linear = inner
                    tt = tt.reshape(rows, kh, kw, cp)[..., :cin].permute(0, 3, 1, 2).contiguous()
                    tt = tt.view(rows, cin) if is_linear else tt
                out[nm] = tt.clone()
            else:
                out[name] = t.clone()
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        shapes = self.param_shapes()
        sd = self._convert_deprecated_keys(dict(sd))
        missing = [k for k in shapes if k not in sd]
        unexpected = [k for k in sd if k not in shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"{type(self).__name__}.load_state_dict: missing {missing[:5]}... ({len(missing)}), "
                               f"unexpected {unexpected[:5]}... ({len(unexpected)})")
        for k, shp in shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shp)}")
        self._src = {k: sd[k].detach().to("cpu", F32) for k in shapes if k in sd}
        if self.training:                  # keep the training layout: rebuild the arenas from the new weights
            self.training = False
            return self.prepare_training()
        self._prepare(self._src)
        self._weights_gen += 1
        self._ready = True
        return self

    def _convert_deprecated_keys(self, sd):
        return sd

    def _prepare(self, sd: Dict[str, torch.Tensor]) -> None:
        raise NotImplementedError

    @classmethod
    def from_config(cls, config, **kw):
        return cls(dict(config), **kw)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype=None, precision=None,
                        device="cuda", **unused):
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            config = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        # torch_dtype=None -> the fast mode; torch.float16 (what test_brushnet.py:124 selects by default) raises in
        # Precision.get instead of being run silently as something else
        prec = precision or (torch_dtype if torch_dtype is not None else "bf16")
        model = cls(config, precision=prec, device=device)
        model.load_state_dict(load_file(os.path.join(d, cls.weights_name)))
        return model

    def train(self, mode: bool = True):
        """nn.Module.train(): the MirrorFusion nets have no dropout / batch-norm state (dropout 0.0, resnet.py:393), so the
        only effect is switching the parameters to the training layout the first time."""
        if mode and not self.training:
            self.prepare_training()
        return self

    def enable_gradient_checkpointing(self):
        """brushnet.py:674-676 / unet_2d_condition.py `enable_gradient_checkpointing`, called by train_brushnet_mirror.py:1153-1155
        under `--gradient_checkpointing`.  Built since round 6 at the reference's granularity (every ResnetBlock2D and every
        Transformer2DModel of the down / mid / up blocks, unet_2d_blocks.py:1167-1196, 2597-2622): the block's forward runs on a
        throw-away tape, only its inputs are kept, and the backward pass runs the block again (autograd.checkpoint).  Gradients
        are bit-identical to the un-checkpointed step; a step costs one more forward of the checkpointed blocks."""
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    def _ckpt(self, fn):
        """fn() under activation recomputation when this model trains with gradient checkpointing on (else just fn())."""
        if getattr(self, "gradient_checkpointing", False) and ops.TAPE is not None:
            return autograd.checkpoint(fn)
        return fn()

    def save_pretrained(self, path: str, **unused):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = self._class_name
        cfg["_diffusers_version"] = "0.27.0.dev0"
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, os.path.join(path, self.weights_name))

    # -- helpers shared by the three models -------------------------------------------------------
    def _conv(self, sd, name, prec=None, cin_pad=None, fp8: bool = False) -> ConvWeight:
        if self.training:
            return self._conv_param(name, sd[name + ".weight"], sd.get(name + ".bias"), prec or self.prec, cin_pad)
        # to_v is consumed by ops.linear_t with the weight as the A operand: never pre-split
        return ConvWeight(sd[name + ".weight"], sd.get(name + ".bias"), prec or self.prec, self.device, cin_pad,
                          raw=name.endswith(".to_v"), fp8=fp8)

    def _conv_param(self, name, weight, bias, prec, cin_pad=None, n_pad: Optional[int] = None) -> ConvWeight:
        """A conv / linear weight allocated in the training arena ([N][kh][kw][cin_pad] fp32); n_pad zero-pads the rows
        (the VAE's 8-aligned moments / latents)."""
        is_linear = weight.dim() == 2
        w4 = weight[:, :, None, None] if is_linear else weight
        n, cin, kh, kw = w4.shape
        cp = cin_pad if cin_pad is not None else (cin + prec.vec - 1) // prec.vec * prec.vec
        wk = torch.nn.functional.pad(w4.float().permute(0, 2, 3, 1), (0, cp - cin)).reshape(n, kh * kw * cp)
        meta = ("conv", n, cin, kh, kw, cp, is_linear)
        rows = n
        if n_pad is not None and n_pad != n:
            wk = torch.nn.functional.pad(wk, (0, 0, 0, n_pad - n))
            meta = ("pad_rows", name + ".weight", n, meta)
            rows = n_pad
        p_w = self._alloc(name + ".weight", wk, meta)
        if rows == n and p_w.grad is not None:
            self._fw_params.append(p_w)        # eligible for zero_grad_for_step's write-first (never the row-padded / fused matrices)
        p_b = None
        if bias is not None:
            bk = torch.nn.functional.pad(bias.float(), (0, rows - n))
            p_b = self._alloc(name + ".bias", bk, ("vec",) if rows == n else ("pad_rows", name + ".bias", n, None))
        return ConvWeight.from_params(p_w, p_b, prec, rows, cin, cp, kh, kw, self)

    def _norm(self, sd, name):
        if self.training:
            return (self._alloc(name + ".weight", sd[name + ".weight"].float(), ("vec",)),
                    self._alloc(name + ".bias", sd[name + ".bias"].float(), ("vec",)))
        return (sd[name + ".weight"].to(self.device, F32).contiguous(), sd[name + ".bias"].to(self.device, F32).contiguous())


def _resnet_shapes(out, p, cin, cout, temb_ch):
    out[p + "norm1.weight"] = (cin,); out[p + "norm1.bias"] = (cin,)
    out[p + "conv1.weight"] = (cout, cin, 3, 3); out[p + "conv1.bias"] = (cout,)
    if temb_ch:
        out[p + "time_emb_proj.weight"] = (cout, temb_ch); out[p + "time_emb_proj.bias"] = (cout,)
    out[p + "norm2.weight"] = (cout,); out[p + "norm2.bias"] = (cout,)
    out[p + "conv2.weight"] = (cout, cout, 3, 3); out[p + "conv2.bias"] = (cout,)
    if cin != cout:
        out[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); out[p + "conv_shortcut.bias"] = (cout,)


def _transformer_shapes(out, p, c, cross, depth=1, linear=False):
    proj = (c, c) if linear else (c, c, 1, 1)                    # use_linear_projection (transformer_2d.py:376-385)
    out[p + "norm.weight"] = (c,); out[p + "norm.bias"] = (c,)
    out[p + "proj_in.weight"] = proj; out[p + "proj_in.bias"] = (c,)
    for i in range(depth):
        b = f"{p}transformer_blocks.{i}."
        for n in ("norm1", "norm2", "norm3"):
            out[b + n + ".weight"] = (c,); out[b + n + ".bias"] = (c,)
        for a, kv in (("attn1", c), ("attn2", cross)):
            out[b + a + ".to_q.weight"] = (c, c)
            out[b + a + ".to_k.weight"] = (c, kv)
            out[b + a + ".to_v.weight"] = (c, kv)
            out[b + a + ".to_out.0.weight"] = (c, c); out[b + a + ".to_out.0.bias"] = (c,)
        out[b + "ff.net.0.proj.weight"] = (8 * c, c); out[b + "ff.net.0.proj.bias"] = (8 * c,)
        out[b + "ff.net.2.weight"] = (c, 4 * c); out[b + "ff.net.2.bias"] = (c,)
    out[p + "proj_out.weight"] = proj; out[p + "proj_out.bias"] = (c,)


def _add_embedding_shapes(out, cfg, temb):
    """SDXL's text_time embedding (unet_2d_condition.py:~560, brushnet.py:303-305)."""
    if cfg.get("addition_embed_type") == "text_time":
        d = cfg["projection_class_embeddings_input_dim"]
        out["add_embedding.linear_1.weight"] = (temb, d); out["add_embedding.linear_1.bias"] = (temb,)
        out["add_embedding.linear_2.weight"] = (temb, temb); out["add_embedding.linear_2.bias"] = (temb,)


class _UNetCore(HipModel):
    """What BrushNetModel and UNet2DConditionModel share: time embedding, resnets, samplers."""

    def _common_defaults(self):
        c = self.config
        c.setdefault("in_channels", 4)
        c.setdefault("block_out_channels", (320, 640, 1280, 1280))
        c.setdefault("layers_per_block", 2)
        c.setdefault("norm_num_groups", 32)
        c.setdefault("norm_eps", 1e-5)
        c.setdefault("cross_attention_dim", 1280)
        c.setdefault("attention_head_dim", 8)
        c.setdefault("num_attention_heads", None)
        c.setdefault("flip_sin_to_cos", True)
        c.setdefault("freq_shift", 0)
        c.setdefault("transformer_layers_per_block", 1)
        c.setdefault("downsample_padding", 1)
        c.setdefault("mid_block_scale_factor", 1)
        c.setdefault("act_fn", "silu")
        c.setdefault("use_linear_projection", False)
        c.setdefault("resnet_time_scale_shift", "default")
        c.setdefault("addition_embed_type", None)
        c.setdefault("addition_time_embed_dim", None)
        c.setdefault("projection_class_embeddings_input_dim", None)
        c["block_out_channels"] = tuple(c["block_out_channels"])
        unsupported = []
        if c["act_fn"] not in ("silu", "swish"): unsupported.append("act_fn")
        if c["addition_embed_type"] not in (None, "text_time"): unsupported.append("addition_embed_type")
        if c["resnet_time_scale_shift"] != "default": unsupported.append("resnet_time_scale_shift")
        if c["mid_block_scale_factor"] != 1: unsupported.append("mid_block_scale_factor")
        if c["downsample_padding"] != 1: unsupported.append("downsample_padding")
        for k in ("class_embed_type", "encoder_hid_dim_type", "num_class_embeds"):
            if c.get(k) is not None: unsupported.append(k)
        if unsupported:
            raise NotImplementedError(f"{type(self).__name__}: config options outside the SD1.5 / SDXL hot path: {unsupported}")

    def _heads(self, level: int) -> int:
        c = self.config
        h = c["num_attention_heads"] or c["attention_head_dim"]     # unet_2d_condition.py:~300 (sic)
        return h if isinstance(h, int) else h[level]

    # ---- parameter preparation ------------------------------------------------------------------
    def _prepare_time(self, sd):
        f32 = Precision.get("fp32")
        self.te1 = self._conv(sd, "time_embedding.linear_1", f32)
        self.te2 = self._conv(sd, "time_embedding.linear_2", f32)
        self.add1 = self.add2 = None
        if self.config["addition_embed_type"] == "text_time":
            self.add1 = self._conv(sd, "add_embedding.linear_1", f32)
            self.add2 = self._conv(sd, "add_embedding.linear_2", f32)
        # all time_emb_proj layers as ONE [sum(Cout), temb] GEMM; each resnet reads a column slice
        names = [k[: -len(".time_emb_proj.weight")] for k in self.param_shapes() if k.endswith(".time_emb_proj.weight")]
        self.temb_slices = {}
        off = 0
        ws, bs = [], []
        for n in names:
            w = sd[n + ".time_emb_proj.weight"]
            self.temb_slices[n + "."] = (off, off + w.shape[0])
            off += w.shape[0]
            ws.append(w); bs.append(sd[n + ".time_emb_proj.bias"])
        if self.training:          # the fused matrix IS the master copy: its row blocks are the resnets' time_emb_proj
            rows = [w.shape[0] for w in ws]
            p_w = self._alloc("time_emb_proj.weight*", torch.cat(ws, 0).float(),
                              ("rows", [n + ".time_emb_proj.weight" for n in names], rows))
            p_b = self._alloc("time_emb_proj.bias*", torch.cat(bs, 0).float(),
                              ("rows", [n + ".time_emb_proj.bias" for n in names], rows))
            k = ws[0].shape[1]
            self.temb_proj = ConvWeight.from_params(p_w, p_b, f32, sum(rows), k, k, 1, 1, self)
            if p_w.grad is not None:
                self._fw_params.append(p_w)        # one Linear, one weight gradient over all its rows: write-first like _conv_param's
        else:
            self.temb_proj = ConvWeight(torch.cat(ws, 0), torch.cat(bs, 0), f32, self.device)

    def _prepare_resnet(self, sd, p):
        self.P[p + "norm1"] = self._norm(sd, p + "norm1")
        self.P[p + "conv1"] = self._conv(sd, p + "conv1")
        self.P[p + "norm2"] = self._norm(sd, p + "norm2")
        self.P[p + "conv2"] = self._conv(sd, p + "conv2")
        if p + "conv_shortcut.weight" in sd:
            self.P[p + "conv_shortcut"] = self._conv(sd, p + "conv_shortcut")

    def _prepare_transformer(self, sd, p):
        # precision "fp8": every Linear of the transformer blocks whose input is a token row of the model width runs on fp8
        # e4m3 operands (per-row activation scales, per-output-channel weight scales); the prompt's K / V projections (77
        # tokens, computed once per prompt) and everything outside the transformer blocks stay bf16
        f8 = self.prec.fp8_linear and not self.training
        ok8 = lambda name: f8 and sd[name + ".weight"].shape[1] % 16 == 0
        self.P[p + "norm"] = self._norm(sd, p + "norm")
        self.P[p + "proj_in"] = self._conv(sd, p + "proj_in", fp8=ok8(p + "proj_in") and sd[p + "proj_in.weight"].dim() == 2)
        self.P[p + "proj_out"] = self._conv(sd, p + "proj_out", fp8=ok8(p + "proj_out") and sd[p + "proj_out.weight"].dim() == 2)
        i = 0
        while f"{p}transformer_blocks.{i}.norm1.weight" in sd:
            b = f"{p}transformer_blocks.{i}."
            for n in ("norm1", "norm2", "norm3"):
                self.P[b + n] = self._norm(sd, b + n)
            for a in ("attn1", "attn2"):
                for l in ("to_q", "to_k", "to_v", "to_out.0"):
                    nm = b + a + "." + l
                    self.P[nm] = self._conv(sd, nm, fp8=ok8(nm) and not (a == "attn2" and l in ("to_k", "to_v")))
            if self.training:      # no fused / interleaved copies: the arena holds each parameter once
                self.P[b + "ff.net.0.proj"] = self._conv(sd, b + "ff.net.0.proj")
            else:
                # self-attention: q and k read the same tokens -> one GEMM with N = 2C
                self.P[b + "attn1.to_qk"] = ConvWeight(torch.cat([sd[b + "attn1.to_q.weight"], sd[b + "attn1.to_k.weight"]], 0),
                                                       None, self.prec, self.device, fp8=ok8(b + "attn1.to_q"))
                self.P[b + "ff.net.0.proj"] = ops.geglu_weight(sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"],
                                                               self.prec, self.device, fp8=ok8(b + "ff.net.0.proj"))
            self.P[b + "ff.net.2"] = self._conv(sd, b + "ff.net.2", fp8=ok8(b + "ff.net.2"))
            c = sd[b + "norm1.weight"].shape[0]
            if self.prec.name in ("bf16", "fp16") and not self.training and c % 320 == 0:
                # bf16 inference: the three LayerNorms of the block are FOLDED into the Linears that consume them (W diag(gamma),
                # bias + W beta; the GEMM gathers the row statistics itself and normalises in its epilogue), and q | k | v of the
                # self-attention are one GEMM whose V third is stored transposed: 3 LayerNorm launches and the V^T launch less
                # per block, the normalised tensors are never written (attention.py:203,233,261,291-412)
                ln = lambda n: (sd[b + n + ".weight"], sd[b + n + ".bias"], 1e-5)
                self.P[b + "attn1.to_qkv_ln"] = ConvWeight(torch.cat([sd[b + "attn1.to_q.weight"], sd[b + "attn1.to_k.weight"],
                                                                      sd[b + "attn1.to_v.weight"]], 0), None, self.prec, self.device,
                                                           ln=ln("norm1"))
                self.P[b + "attn2.to_q_ln"] = ConvWeight(sd[b + "attn2.to_q.weight"], None, self.prec, self.device, ln=ln("norm2"))
                # norm3 -> GEGLU projection (round 6): on the ring tiles the fold lost at every level (sixteen column tiles repeat the row
                # statistics, and the GEGLU epilogue wants small tiles); the persistent tile 70 gathers them once per block from the A
                # tiles it stages anyway (tools/bench_ff1.py: 32768 x 320 -> 2560 +4 us against a 12 us LayerNorm launch)
                self.P[b + "ff.net.0.proj_ln"] = ops.geglu_weight(sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"],
                                                                  self.prec, self.device, ln=ln("norm3"))
            i += 1
        self.tdepth[p] = i

    # ---- forward pieces ---------------------------------------------------------------------------
    def _time_embedding(self, timestep, batch: int, added_cond_kwargs=None) -> torch.Tensor:
        """Timesteps + TimestepEmbedding (+ SDXL's text_time add_embedding, unet_2d_condition.py:971-987) + every
        resnet's time_emb_proj(SiLU(emb)) -> [batch, sum(Cout)] fp32.
        (embeddings.py:27-67,225-254; resnet.py:369-376).  Always fp32, like the reference."""
        if not torch.is_tensor(timestep):
            t = torch.full((batch,), float(timestep), dtype=F32, device=self.device)
        else:
            t = timestep.to(self.device, F32).reshape(-1)
            if t.numel() == 1:
                t = t.expand(batch)
            t = t.contiguous()
        c0 = self.config["block_out_channels"][0]
        e = hip.timestep_embedding(t, c0, self.config["flip_sin_to_cos"], float(self.config["freq_shift"]))
        if ops.TAPE is not None:           # training: every activation is its own differentiated operator
            if self.add1 is not None:
                raise NotImplementedError("training the SDXL text_time embedding is not built")
            ops.TAPE.no_grad(e)
            e = ops.silu(ops.linear(e, self.te1, out_dtype=F32))
            e = ops.silu(ops.linear(e, self.te2, out_dtype=F32))
            return ops.linear(e, self.temb_proj, out_dtype=F32)
        e = ops.linear(e, self.te1, act=hip.ACT_SILU, out_dtype=F32)
        if self.add1 is None:
            if added_cond_kwargs:
                raise NotImplementedError("added_cond_kwargs without addition_embed_type='text_time'")
            e = ops.linear(e, self.te2, act=hip.ACT_SILU, out_dtype=F32)   # SiLU(emb): every consumer applies it first
            return ops.linear(e, self.temb_proj, out_dtype=F32)
        for k in ("text_embeds", "time_ids"):
            if not added_cond_kwargs or k not in added_cond_kwargs:
                raise ValueError(f"{self.__class__} has the config param `addition_embed_type` set to 'text_time' which "
                                 f"requires the keyword argument `{k}` to be passed in `added_cond_kwargs`")
        text = added_cond_kwargs["text_embeds"].to(self.device, F32)
        tid = added_cond_kwargs["time_ids"].to(self.device, F32).reshape(-1).contiguous()
        te = hip.timestep_embedding(tid, self.config["addition_time_embed_dim"], self.config["flip_sin_to_cos"],
                                    float(self.config["freq_shift"])).view(text.shape[0], -1)
        a = torch.cat([text, te], -1).contiguous()                       # tiny: [batch, 2816]
        a = ops.linear(a, self.add1, act=hip.ACT_SILU, out_dtype=F32)
        # emb = time_embedding(t) + add_embedding(...): the second GEMM adds the first one's output as a residual
        e = ops.linear(e, self.te2, out_dtype=F32)
        e = ops.linear(a, self.add2, res0=e, act=hip.ACT_SILU, out_dtype=F32)
        return ops.linear(e, self.temb_proj, out_dtype=F32)

    def time_embedding_table(self, timesteps: torch.Tensor, batch: int, added_cond_kwargs=None) -> torch.Tensor:
        """Every resnet's time_emb_proj(SiLU(time_embedding(t))) for ALL timesteps of a schedule in one batched pass
        (the timesteps are known after set_timesteps; resnet.py:369-376) -> [steps, rows, sum(Cout)] fp32.
        rows = 1 when the embedding depends on the timestep alone (SD1.5: every image of the batch reads the same row,
        `forward(..., _temb=table[i])`), rows = batch with SDXL's per-image text_time embedding."""
        t = timesteps.to(self.device, F32).reshape(-1)
        steps = t.numel()
        if self.add1 is None:
            return self._time_embedding(t, steps, None).view(steps, 1, -1)
        added = {k: v.to(self.device, F32).repeat(steps, *([1] * (v.dim() - 1))) for k, v in added_cond_kwargs.items()}
        return self._time_embedding(t.repeat_interleave(batch), steps * batch, added).view(steps, batch, -1)

    def _temb_rows(self, temb: Optional[torch.Tensor], timestep, bsz: int, added_cond_kwargs) -> torch.Tensor:
        """The [bsz, sum(Cout)] time-embedding rows of one forward pass: computed here, or a precomputed row block of
        time_embedding_table (one shared row is broadcast with a zero row stride: mf_gemm_desc.ld_temb = 0)."""
        if temb is None:
            return self._time_embedding(timestep, bsz, added_cond_kwargs)
        if ops.TAPE is not None:
            raise hip.MfhipError("training differentiates its own time embedding: _temb is an inference input")
        if temb.shape[0] not in (1, bsz):
            raise hip.MfhipError(f"_temb has {temb.shape[0]} rows for a batch of {bsz}")
        return temb.expand(bsz, -1) if temb.shape[0] == 1 else temb

    def _temb(self, temb_all: Optional[torch.Tensor], p: str) -> Optional[torch.Tensor]:
        if temb_all is None:
            return None
        a, b = self.temb_slices[p]
        return temb_all[:, a:b]

    def _operand_dtype(self) -> torch.dtype:
        """Storage dtype of a norm / activation output whose ONLY consumers are conv / linear operands (resnet norms, the transformer's
        GroupNorm, LayerNorms and GEGLU).  bf16x1 training on pre-rounded operands (ops.ConvWeight.fast16): the producer writes bf16 — the
        very rounding the GEMM would apply, done once, with half the bytes written, read and kept for the backward pass."""
        if self.training and ops.BF16X1_FAST and self.prec.code == hip.MF_BF16X1:
            return torch.bfloat16
        return self.prec.act

    def _resnet(self, p: str, x: torch.Tensor, temb_all, x1: Optional[torch.Tensor] = None,
                inj: Optional[torch.Tensor] = None, eps: Optional[float] = None) -> torch.Tensor:
        return self._ckpt(lambda: self._resnet_impl(p, x, temb_all, x1, inj, eps))

    def _resnet_impl(self, p: str, x: torch.Tensor, temb_all, x1: Optional[torch.Tensor] = None,
                     inj: Optional[torch.Tensor] = None, eps: Optional[float] = None) -> torch.Tensor:
        """ResnetBlock2D (resnet.py:329-405) on NHWC x (or the never-materialised cat([x, x1], C)), with the
        BrushNet injection add fused into conv2's epilogue."""
        g = self.config["norm_num_groups"]
        eps = self.config["norm_eps"] if eps is None else eps
        P = self.P
        join = None
        if p + "conv_shortcut" in P:
            # the 1x1 shortcut only needs the block input: on the auxiliary stream it overlaps norm1 / conv1 / norm2
            sc, join = self._on_aux(lambda: ops.conv2d(x, P[p + "conv_shortcut"], padding=0, x1=x1))
        else:
            sc = x
        h = ops.groupnorm(x, P[p + "norm1"], groups=g, eps=eps, silu=True, out_dtype=self._operand_dtype(), x1=x1)
        # conv1's output feeds norm2 and nothing else: above 16 x 16 the launch leaves norm2's statistics (gn_part), up to 16 x 16 —
        # where it runs split-K — it leaves the summing of its K slices to norm2 (defer_reduce: the reduce launch disappears)
        h = ops.conv2d(h, P[p + "conv1"], temb=self._temb(temb_all, p), gn_part=self.config["norm_num_groups"], defer_reduce=g)
        h = ops.groupnorm(h, P[p + "norm2"], groups=g, eps=eps, silu=True, out_dtype=self._operand_dtype())
        if join is not None:
            join()
        return ops.conv2d(h, P[p + "conv2"], res0=sc, res1=inj, gn_part=self.config["norm_num_groups"])

    aux_stream: Optional["torch.cuda.Stream"] = None

    def _on_aux(self, fn):
        """Run `fn` (launches whose inputs are ready now and whose result is needed later) on this model's auxiliary
        HIP stream; returns (result, join) where join() orders the current stream after it.  Without an auxiliary
        stream (the default outside the pipeline's denoise loops) fn runs inline."""
        aux = self.aux_stream
        if aux is None:
            return fn(), None
        cur = torch.cuda.current_stream(self.device)
        aux.wait_stream(cur)
        with torch.cuda.stream(aux):
            out = fn()
            done = torch.cuda.Event()
            done.record(aux)
        for t in (out if isinstance(out, (tuple, list)) else (out,)):
            t.record_stream(cur)

        def join():
            cur.wait_event(done)
        return out, join

    ln_fold = os.environ.get("MFHIP_NO_LNFOLD") != "1"      # A/B switch for the folded LayerNorms / fused q | k | v (bf16 inference)
    ff_ln_fold = os.environ.get("MFHIP_NO_FF_LNFOLD") != "1"   # A/B switch: norm3 folded into the GEGLU projection where tile 70 runs it

    def _attention(self, b: str, x: torch.Tensor, ctx: Optional[torch.Tensor], heads: int, residual: torch.Tensor,
                   fold: bool = False) -> torch.Tensor:
        """Attention + AttnProcessor2_0 (attention_processor.py:1213-1286) + the block's residual add.
        Self-attention projects q and k with one GEMM; cross-attention K / V^T depend only on the prompt
        embeddings (attention_processor.py:1253-1254) and are cached across denoise steps."""
        P = self.P
        xt = x[0] if isinstance(x, tuple) else x          # fp8: x is the (quantised rows, row scales) pair of ops.layernorm
        c = xt.shape[-1]
        d = c // heads
        if self.training:
            # training layout: separate q / k / v projections and the unfused, differentiated attention; the prompt's
            # K / V are recomputed every step (their weights may be training)
            src = x if ctx is None else ctx
            # bf16x1 on the bf16 flash kernels: q / k / v leave their projections in bf16 (as under the reference's autocast)
            qdt = torch.bfloat16 if (ops.flash_bf16_train(self.prec, xt.shape[1], d) and c % 8 == 0 and P[b + "to_q"].fast16(xt.shape[-1])
                                     and P[b + "to_k"].fast16(src.shape[-1]) and P[b + "to_v"].fast16(src.shape[-1])) else None
            q = ops.linear(x, P[b + "to_q"], out_dtype=qdt)
            k = ops.linear(src, P[b + "to_k"], out_dtype=qdt)
            v = ops.linear(src, P[b + "to_v"], out_dtype=qdt)
            o = ops.attention_train(q, k, v, heads, 1.0 / (d ** 0.5), self.prec)
            return ops.linear(o, P[b + "to_out.0"], res0=residual)
        if fold and ctx is None:
            # x is the UN-normalised residual stream: norm1 lives inside the fused q | k | v GEMM
            skv = xt.shape[1]
            qk, vt = ops.linear_qkv(x, P[b + "to_qkv_ln"])
            q, k = qk[..., :c], qk[..., c:]
        elif ctx is None:
            skv = xt.shape[1]
            # V^T on the auxiliary stream while q | k is projected on this one
            vt, join = self._on_aux(lambda: ops.linear_t(x, P[b + "to_v"], (skv + 7) // 8 * 8, out=self._vt_buffer(xt, c, skv)))
            qk = ops.linear(x, P[b + "to_qk"])
            q, k = qk[..., :c], qk[..., c:]
            if join is not None:
                join()
        else:
            skv = ctx.shape[1]
            q = ops.linear(x, P[b + ("to_q_ln" if fold else "to_q")])
            kv = self._cross_kv.get(b)
            if kv is None or kv[2] != self._ehs_gen:
                # (re)compute into persistent buffers: a captured hipGraph keeps reading the same addresses
                reuse = kv is not None and kv[0].shape[:2] == ctx.shape[:2]
                k = ops.linear(ctx, P[b + "to_k"], out=kv[0] if reuse else None)
                vt = ops.linear_t(ctx, P[b + "to_v"], (skv + 7) // 8 * 8, out=kv[1] if reuse else None)
                self._cross_kv[b] = (k, vt, self._ehs_gen)
            else:
                k, vt = kv[0], kv[1]
        o = ops.attention(q, k, vt, heads, skv, 1.0 / (d ** 0.5), self.prec, c=c)
        return ops.linear(o, P[b + "to_out.0"], res0=residual)

    def _vt_buffer(self, x: torch.Tensor, c: int, skv: int) -> Optional[torch.Tensor]:
        """Self-attention V^T needs no zero padding when the token count is a multiple of 8."""
        if skv % 8 != 0:
            return None
        return torch.empty(x.shape[0], c, skv, dtype=self.prec.act, device=x.device)

    def _bind_prompt(self, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        """Convert the prompt embeddings once and invalidate the cross-attention K/V cache when they change."""
        # identity + version of the caller's tensor (a data_ptr alone can be recycled by the allocator)
        ref = getattr(self, "_ehs_ref", None)
        hit = ref is not None and ref() is encoder_hidden_states and self._ehs_key == encoder_hidden_states._version
        if not hit:
            import weakref
            self._ehs_ref = weakref.ref(encoder_hidden_states)
            self._ehs_key = encoder_hidden_states._version
            val = self._ehs(encoder_hidden_states)
            old = getattr(self, "_ehs_val", None)
            if old is not None and old.shape == val.shape and old.device == val.device:
                old.copy_(val)               # keep the address stable for captured graphs
            else:
                self._ehs_val = val
            self._ehs_gen = getattr(self, "_ehs_gen", 0) + 1
        return self._ehs_val

    def bind_prompt(self, encoder_hidden_states: torch.Tensor) -> bool:
        """(Re)compute everything that depends only on the prompt embeddings — their device copy and the K / V^T of every
        cross-attention layer — into the persistent buffers the forward pass (and a captured denoise graph) reads,
        without running a forward pass.  Returns False when the buffers do not exist yet or no longer fit (first call,
        new prompt shape): the caller then runs one eager forward, which creates them."""
        if not self._cross_kv:
            return False
        ctx = self._bind_prompt(encoder_hidden_states)
        for b, kv in list(self._cross_kv.items()):
            if kv[2] == self._ehs_gen:
                continue
            if kv[0].shape[:2] != ctx.shape[:2]:
                return False
            skv = ctx.shape[1]
            ops.linear(ctx, self.P[b + "to_k"], out=kv[0])
            ops.linear_t(ctx, self.P[b + "to_v"], (skv + 7) // 8 * 8, out=kv[1])
            self._cross_kv[b] = (kv[0], kv[1], self._ehs_gen)
        return True

    def _transformer(self, p: str, x: torch.Tensor, ehs: torch.Tensor, heads: int,
                     inj: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self._ckpt(lambda: self._transformer_impl(p, x, ehs, heads, inj))

    def _transformer_impl(self, p: str, x: torch.Tensor, ehs: torch.Tensor, heads: int,
                          inj: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Transformer2DModel + BasicTransformerBlock(s) (transformer_2d.py:334-430, attention.py:291-412)."""
        P = self.P
        bsz, hh, ww, c = x.shape
        g = self.config["norm_num_groups"]
        h = ops.groupnorm(x, P[p + "norm"], groups=g, eps=1e-6, silu=False, out_dtype=self._operand_dtype())
        if P[p + "proj_in"].fp8:           # use_linear_projection (SDXL): a Linear over tokens, on the fp8 path
            h = ops.linear(h.view(bsz, hh * ww, c), P[p + "proj_in"])
        else:
            h = ops.conv2d(h, P[p + "proj_in"], padding=0).view(bsz, hh * ww, c)
        for i in range(self.tdepth[p]):
            b = f"{p}transformer_blocks.{i}."
            if self.ln_fold and (b + "attn1.to_qkv_ln") in P and (hh * ww) % 8 == 0 and ops.TAPE is None:
                # which LayerNorms fold into their consumer is decided per level from measurements on MI355X (tools/bench_fold.py,
                # gpurun_out/r03d: batch 8): the in-kernel row statistics cost ~5 us per GEMM and run on the warp-specialised
                # ring tiles only, so the fold pays where the LayerNorm launch it removes costs more than that
                #   norm2 -> to_q (N = C):           64x64 35.4 -> 27.6 us, 32x32 22.7 -> 18.3, 16x16 23.2 -> 21.2, 8x8 14.8 -> 20.2 (no)
                #   norm1 -> q | k | v^T (N = 3C):   64x64 68.9 -> 72.2 (no), 32x32 51.0 -> 49.6, 16x16 45.9 -> 38.0, 8x8 26.7 -> 21.4
                #   norm3 -> GEGLU (N = 8C):         slower at every level (16 column tiles repeat the statistics, and the
                #                                    GEGLU epilogue prefers the small tiles): stays a LayerNorm launch
                tokens = hh * ww
                f_qkv, f_q = tokens <= 1024, tokens >= 256
                # (round 6: on the persistent tile 70 the fused q | k | v^T also pays at 64 x 64 — the row statistics are gathered once
                # per block, and the V^T launch and the LayerNorm launch both go)
                f_qkv = f_qkv or (self.ff_ln_fold and hip.pers_linear(bsz * tokens, 3 * c, c, self.prec.act))
                if f_qkv:
                    h = self._attention(b + "attn1.", h, None, heads, h, fold=True)
                else:
                    h = self._attention(b + "attn1.", ops.layernorm(h, P[b + "norm1"], 1e-5, self.prec.act), None, heads, h)
                if f_q:
                    h = self._attention(b + "attn2.", h, ehs, heads, h, fold=True)
                else:
                    h = self._attention(b + "attn2.", ops.layernorm(h, P[b + "norm2"], 1e-5, self.prec.act), ehs, heads, h)
                ffl = P.get(b + "ff.net.0.proj_ln")
                if ffl is not None and self.ff_ln_fold and hip.pers_linear(bsz * tokens, ffl.n, c, self.prec.act):
                    gg = ops.linear_geglu(h, ffl)                 # norm3 folded: the persistent GEMM gathers the row statistics
                else:
                    gg = ops.linear_geglu(ops.layernorm(h, P[b + "norm3"], 1e-5, self.prec.act), P[b + "ff.net.0.proj"])
                h = ops.linear(gg, P[b + "ff.net.2"], res0=h)
                continue
            n = ops.layernorm(h, P[b + "norm1"], 1e-5, self._operand_dtype(), fp8=P[b + "attn1.to_out.0"].fp8)
            h = self._attention(b + "attn1.", n, None, heads, h)
            n = ops.layernorm(h, P[b + "norm2"], 1e-5, self._operand_dtype(), fp8=P[b + "attn2.to_q"].fp8)
            h = self._attention(b + "attn2.", n, ehs, heads, h)
            n = ops.layernorm(h, P[b + "norm3"], 1e-5, self._operand_dtype(), fp8=P[b + "ff.net.2"].fp8)
            if self.training:      # GEGLU as its own (differentiated) launch on the plain, un-interleaved weight
                ff0 = P[b + "ff.net.0.proj"]
                if ops.bf16x1_operands(self.prec) and ff0.n % 16 == 0 and ff0.fast16(n.shape[-1]):
                    # the [rows, 2 inner] pre-activation — the largest activation of the network — in bf16, its gradient too
                    gg = ops.geglu(ops.linear(n, ff0, out_dtype=torch.bfloat16), torch.bfloat16, bias=ff0.p_bias)
                else:
                    gg = ops.geglu(ops.linear(n, ff0), self._operand_dtype())
            else:
                gg = ops.linear_geglu(n, P[b + "ff.net.0.proj"])
            h = ops.linear(gg, P[b + "ff.net.2"], res0=h)
        if isinstance(inj, LazyResidual):
            fused = self._fused_proj_out(p, inj)
            if fused is not None and inj.feature.shape == x.shape:
                inj.wait()
                return ops.conv2d(h.view(bsz, hh, ww, c), fused, padding=0, x1=inj.feature, res0=x, gn_part=self.config["norm_num_groups"])
            inj = inj.materialize()
        if P[p + "proj_out"].fp8:
            return ops.linear(h, P[p + "proj_out"], res0=x.view(bsz, hh * ww, c),
                              res1=inj.view(-1, hh * ww, c) if inj is not None else None).view(bsz, hh, ww, c)
        return ops.conv2d(h.view(bsz, hh, ww, c), P[p + "proj_out"], padding=0, res0=x, res1=inj, gn_part=self.config["norm_num_groups"])

    def _fused_proj_out(self, p: str, lz: "LazyResidual") -> Optional[ConvWeight]:
        """[W_proj_out | scale * W_zero_conv] over K = 2C with bias b_po + scale * b_zc, built once per (layer, BrushNet weights,
        scale) from the two models' fp32 master copies; None where the projection is not a plain 1x1 conv of this precision."""
        bn = lz.owner
        key = (p, lz.name, lz.scale, id(bn), bn._weights_gen, self._weights_gen)
        cache = self.__dict__.setdefault("_fused_po", {})
        if key in cache:
            return cache[key]
        cw = None
        if (self._src is not None and bn._src is not None and not self.P[p + "proj_out"].fp8 and not self.training
                and self._src[p + "proj_out.weight"].dim() == 4 and bn.prec.name == self.prec.name):
            w = torch.cat([self._src[p + "proj_out.weight"].float(), lz.scale * bn._src[lz.name + ".weight"].float()], 1)
            b = self._src[p + "proj_out.bias"].float() + lz.scale * bn._src[lz.name + ".bias"].float()
            cw = ConvWeight(w, b, self.prec, self.device)
        for k in [k for k in cache if k[0] == p and k != key]:     # weights or scale changed: drop the stale entry
            del cache[k]
        cache[key] = cw
        return cw

    # ---- the reference's operator plug-in point (attention_processor.py:216; brushnet.py:558-590;
    # unet_2d_condition.py:716-748) ----------------------------------------------------------------------------
    @property
    def attn_processors(self) -> Dict[str, Any]:
        """One entry per attention layer, keyed like the reference ("...attn1.processor"); every layer runs the HIP
        attention of this library."""
        from .attn_processor import MfhipAttnProcessor
        return {k[: -len("to_q")] + "processor": MfhipAttnProcessor() for k in self.P if k.endswith(".to_q")}

    def set_attn_processor(self, processor) -> None:
        """The reference lets a caller swap the attention arithmetic per layer.  Here attention is fused into the HIP graph
        (q|k projection in one GEMM, V produced transposed, flash kernel): the only processor these models run is
        `MfhipAttnProcessor` (the same kernel, exposed on the reference's processor ABI for use inside the reference's own
        modules).  Passing it — or a dict of it with exactly the reference's keys — is accepted; anything else is refused
        instead of being silently ignored."""
        from .attn_processor import MfhipAttnProcessor
        keys = set(self.attn_processors)
        if isinstance(processor, dict):
            if len(processor) != len(keys):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                                 f" number of attention layers: {len(keys)}. Please make sure to pass {len(keys)} processor classes.")
            procs = list(processor.values())
        else:
            procs = [processor]
        if not all(isinstance(p, MfhipAttnProcessor) for p in procs):
            raise NotImplementedError("the HIP models run their own fused attention: only MfhipAttnProcessor is accepted "
                                      "(use it with the reference's modules to run this kernel there)")

    def set_default_attn_processor(self) -> None:
        return None

    def _ehs(self, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        return encoder_hidden_states.to(self.device, self.prec.act).contiguous()

    def _inj(self, t, keep_lazy: bool = False):
        if t is None:
            return None
        if isinstance(t, LazyResidual):
            return t if keep_lazy else t.materialize()
        ev = _RESIDUAL_EVENTS.pop(t.data_ptr(), None)          # produced on BrushNet's side stream: order after it
        if ev is not None:
            torch.cuda.current_stream(t.device).wait_event(ev)
        return from_nchw(t, self.prec)


class LazyResidual:
    """A BrushNet residual that has NOT been computed: the feature its zero-conv reads (NHWC), the zero-conv's name and scale.
    The UNet consumes it where the injection lands on a 1x1 proj_out (unet_2d_blocks.py:1389,1484,2627,2752 after a
    Transformer2DModel): proj_out(h) + x + zero_conv(f) = [W_po | s W_zc] . [h | f] + (b_po + s b_zc) + x — ONE GEMM over two K
    segments, so the zero-conv launch, its output and the residual read disappear.  Anywhere else it is materialised."""

    def __init__(self, feature: torch.Tensor, name: str, scale: float, owner: "BrushNetModel"):
        self.feature, self.name, self.scale, self.owner = feature, name, float(scale), owner

    def wait(self):
        ev = _RESIDUAL_EVENTS.pop(self.feature.data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream(self.feature.device).wait_event(ev)

    def materialize(self) -> torch.Tensor:
        self.wait()
        return ops.conv2d(self.feature, self.owner.P[self.name], padding=0, alpha=self.scale)


# BrushNet || UNet overlap.  BrushNet has no data dependence on the UNet, and the UNet needs BrushNet's residual k
# only at its k-th injection point (unet_2d_condition.py:1218 ff.).  With `BrushNetModel.side_stream` set (the
# pipeline does that inside its denoise loops), BrushNet runs on that HIP stream, records an event after every
# zero-conv and the UNet waits on that event right before the add.  Neither network fills 256 CUs during its
# low-resolution / short-K launches, and every launch has a ramp and a drain: two streams fill those holes.
_RESIDUAL_EVENTS: Dict[int, "torch.cuda.Event"] = {}


# =================================================================================================
class BrushNetModel(_UNetCore):
    """models/brushnet.py — attention-free UNet clone emitting 12 down + 1 mid + 15 up residuals."""

    _class_name = "BrushNetModel"

    def __init__(self, config=None, precision="bf16", device="cuda", **kwargs):
        cfg = dict(config or {})
        cfg.update(kwargs)
        cfg.setdefault("conditioning_channels", 5)
        n = len(cfg.get("block_out_channels", (320, 640, 1280, 1280)))
        cfg.setdefault("down_block_types", ("DownBlock2D",) * n)
        cfg.setdefault("up_block_types", ("UpBlock2D",) * n)
        cfg.setdefault("mid_block_type", "MidBlock2D")
        cfg.setdefault("brushnet_conditioning_channel_order", "rgb")
        cfg.setdefault("global_pool_conditions", False)
        super().__init__(cfg, precision, device)
        self._common_defaults()
        c = self.config
        if any(t != "DownBlock2D" for t in c["down_block_types"]) or any(t != "UpBlock2D" for t in c["up_block_types"]) \
                or c["mid_block_type"] != "MidBlock2D":
            raise NotImplementedError("BrushNetModel: only the attention-free layout produced by from_unet "
                                      "(DownBlock2D / MidBlock2D / UpBlock2D, brushnet.py:484-486) is built")
        if c["global_pool_conditions"]:
            raise NotImplementedError("global_pool_conditions is off in every MirrorFusion config")

    @classmethod
    def from_unet(cls, unet: "UNet2DConditionModel", brushnet_conditioning_channel_order: str = "rgb",
                  conditioning_embedding_out_channels=(16, 32, 96, 256), load_weights_from_unet: bool = True,
                  conditioning_channels: int = 5) -> "BrushNetModel":
        """brushnet.py:452-530: same widths as the UNet, attention-free blocks; optionally clone its weights."""
        uc = unet.config
        n = len(uc["block_out_channels"])
        cfg = {k: uc[k] for k in ("in_channels", "flip_sin_to_cos", "freq_shift", "block_out_channels",
                                  "layers_per_block", "downsample_padding", "mid_block_scale_factor", "act_fn",
                                  "norm_num_groups", "norm_eps", "cross_attention_dim", "attention_head_dim",
                                  "num_attention_heads", "use_linear_projection", "resnet_time_scale_shift",
                                  "transformer_layers_per_block", "addition_embed_type", "addition_time_embed_dim",
                                  "projection_class_embeddings_input_dim")}
        cfg.update(conditioning_channels=conditioning_channels, down_block_types=("DownBlock2D",) * n,
                   mid_block_type="MidBlock2D", up_block_types=("UpBlock2D",) * n,
                   brushnet_conditioning_channel_order=brushnet_conditioning_channel_order,
                   conditioning_embedding_out_channels=tuple(conditioning_embedding_out_channels))
        bn = cls(cfg, precision=unet.prec, device=unet.device)
        if load_weights_from_unet:
            usd = unet.state_dict()
            sd = {}
            for k, shp in bn.param_shapes().items():
                if k.startswith("brushnet_"):
                    sd[k] = torch.zeros(shp)                                  # zero_module (brushnet.py:928-931)
                elif k == "conv_in_condition.weight":
                    w = torch.zeros(shp)
                    w[:, :4] = usd["conv_in.weight"]
                    w[:, 4:8] = usd["conv_in.weight"]                          # brushnet.py:514-518
                    sd[k] = w
                elif k == "conv_in_condition.bias":
                    sd[k] = usd["conv_in.bias"].clone()
                else:
                    sd[k] = usd[k].clone()
            bn.load_state_dict(sd)
        return bn

    def param_shapes(self):
        c = self.config
        out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        boc = c["block_out_channels"]
        temb = boc[0] * 4
        cin = c["in_channels"] + c["conditioning_channels"]
        out["conv_in_condition.weight"] = (boc[0], cin, 3, 3); out["conv_in_condition.bias"] = (boc[0],)
        out["time_embedding.linear_1.weight"] = (temb, boc[0]); out["time_embedding.linear_1.bias"] = (temb,)
        out["time_embedding.linear_2.weight"] = (temb, temb); out["time_embedding.linear_2.bias"] = (temb,)
        _add_embedding_shapes(out, c, temb)
        lpb = c["layers_per_block"]
        n = len(boc)
        zero = [boc[0]]
        ch = boc[0]
        for i in range(n):
            for j in range(lpb):
                _resnet_shapes(out, f"down_blocks.{i}.resnets.{j}.", ch if j == 0 else boc[i], boc[i], temb)
                zero.append(boc[i])
            ch = boc[i]
            if i != n - 1:
                out[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (ch, ch, 3, 3)
                out[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (ch,)
                zero.append(ch)
        for k, z in enumerate(zero):
            out[f"brushnet_down_blocks.{k}.weight"] = (z, z, 1, 1); out[f"brushnet_down_blocks.{k}.bias"] = (z,)
        out["brushnet_mid_block.weight"] = (boc[-1], boc[-1], 1, 1); out["brushnet_mid_block.bias"] = (boc[-1],)
        for j in range(2):
            _resnet_shapes(out, f"mid_block.resnets.{j}.", boc[-1], boc[-1], temb)
        rev = list(reversed(boc))
        skip_ch = list(zero)
        uz = []
        prev = rev[0]
        for i in range(n):
            oc = rev[i]
            for j in range(lpb + 1):
                sk = skip_ch.pop()
                _resnet_shapes(out, f"up_blocks.{i}.resnets.{j}.", (prev if j == 0 else oc) + sk, oc, temb)
                uz.append(oc)
            prev = oc
            if i != n - 1:
                out[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (oc, oc, 3, 3)
                out[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (oc,)
                uz.append(oc)
        for k, z in enumerate(uz):
            out[f"brushnet_up_blocks.{k}.weight"] = (z, z, 1, 1); out[f"brushnet_up_blocks.{k}.bias"] = (z,)
        return out

    def _prepare(self, sd):
        c = self.config
        self.P: Dict[str, Any] = {}
        self.tdepth: Dict[str, int] = {}
        cin = c["in_channels"] + c["conditioning_channels"]
        self.cin_pad = (cin + 7) // 8 * 8
        self.P["conv_in_condition"] = self._conv(sd, "conv_in_condition", cin_pad=self.cin_pad)
        self._prepare_time(sd)
        for k in self.param_shapes():
            if k.endswith(".norm1.weight") and ".resnets." in k:
                self._prepare_resnet(sd, k[: -len("norm1.weight")])
            elif k.endswith("samplers.0.conv.weight"):
                self.P[k[: -len(".weight")]] = self._conv(sd, k[: -len(".weight")])
            elif k.startswith("brushnet_") and k.endswith(".weight"):
                self.P[k[: -len(".weight")]] = self._conv(sd, k[: -len(".weight")])

    @_scoped_tune_ctx
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: Optional[torch.Tensor] = None,
                brushnet_cond: torch.Tensor = None, conditioning_scale: float = 1.0, class_labels=None,
                timestep_cond=None, attention_mask=None, added_cond_kwargs=None, cross_attention_kwargs=None,
                guess_mode: bool = False, return_dict: bool = True, *, _temb: Optional[torch.Tensor] = None,
                _lazy: Optional[set] = None):
        """brushnet.py:678-925.  Returns NCHW-shaped channels-last views (see module docstring).
        `_temb` (not in the reference): a row block of time_embedding_table for this timestep.  `_lazy` (not in the reference):
        names of zero-convs to hand over as LazyResidual objects instead of tensors (the pipeline passes
        UNet2DConditionModel.lazy_injection_names(): the residuals that land on a proj_out)."""
        c = self.config
        order = c["brushnet_conditioning_channel_order"]
        if order == "bgr":
            brushnet_cond = torch.flip(brushnet_cond, dims=[1])
        elif order != "rgb":
            raise ValueError(f"unknown `brushnet_conditioning_channel_order`: {order}")        # brushnet.py:741
        if not self._ready:
            raise RuntimeError("BrushNetModel has no parameters loaded")
        if guess_mode and ops.TAPE is not None:
            raise NotImplementedError("guess_mode is an inference feature")
        if class_labels is not None or timestep_cond is not None or attention_mask is not None:
            raise NotImplementedError("class/timestep_cond/attention_mask inputs are outside the SD1.5 / SDXL hot path")
        side = self.side_stream
        if side is None:
            d, m, u = self._forward_impl(sample, timestep, brushnet_cond, conditioning_scale, None, added_cond_kwargs, guess_mode,
                                         _temb, _lazy)
        else:
            main = torch.cuda.current_stream(self.device)
            _RESIDUAL_EVENTS.clear()
            side.wait_stream(main)                                                                # inputs are ready

            def publish(t: torch.Tensor):
                ev = torch.cuda.Event()
                ev.record(side)
                t.record_stream(main)                     # consumed on `main`: the allocator must not recycle it early
                _RESIDUAL_EVENTS[t.data_ptr()] = ev

            with torch.cuda.stream(side):
                d, m, u = self._forward_impl(sample, timestep, brushnet_cond, conditioning_scale, publish, added_cond_kwargs,
                                             guess_mode, _temb, _lazy)
        hip.TUNE_CTX = None
        if not return_dict:
            return d, m, u
        return BrushNetOutput(down_block_res_samples=d, mid_block_res_sample=m, up_block_res_samples=u)

    side_stream: Optional["torch.cuda.Stream"] = None

    def _forward_impl(self, sample, timestep, brushnet_cond, conditioning_scale, publish, added_cond_kwargs=None,
                      guess_mode: bool = False, temb_rows: Optional[torch.Tensor] = None, lazy: Optional[set] = None):
        """Each zero-conv (brushnet.py:889-894) runs right after the feature it reads is produced — the same
        arithmetic as the reference's end-of-forward loops, but residual k is final as early as possible.
        guess_mode (brushnet.py:896-902): residual k of the [down | mid | up] list is scaled by logspace(-1, 0)[k] *
        conditioning_scale (fp32, like the reference's tensor arithmetic) instead of conditioning_scale."""
        c = self.config
        bsz = sample.shape[0]
        n = len(c["block_out_channels"])
        lpb = c["layers_per_block"]
        n_down, n_up = 1 + n * lpb + (n - 1), n * (lpb + 1) + (n - 1)
        hip.TUNE_CTX = "b" if ops.TAPE is None else None      # inference: position-dependent tile choices (hip.TUNE_CTX)
        if guess_mode:
            sc = (torch.logspace(-1, 0, n_down + 1 + n_up) * conditioning_scale).tolist()
        else:
            sc = [float(conditioning_scale)] * (n_down + 1 + n_up)
        scale_of = {f"brushnet_down_blocks.{k}": sc[k] for k in range(n_down)}
        scale_of["brushnet_mid_block"] = sc[n_down]
        scale_of.update({f"brushnet_up_blocks.{k}": sc[n_down + 1 + k] for k in range(n_up)})
        temb = self._temb_rows(temb_rows, timestep, bsz, added_cond_kwargs)
        x = hip.pack_nhwc(sample.to(self.device).float().contiguous(), brushnet_cond.to(self.device).float().contiguous(),
                          self.cin_pad, self.prec.act)                                            # :810 cat + pad
        if ops.TAPE is not None:
            ops.TAPE.no_grad(x)            # the batch's own inputs need no gradient
        x = ops.conv2d(x, self.P["conv_in_condition"], gn_part=self.config["norm_num_groups"])

        def zero_conv(name: str, r: torch.Tensor):
            if lazy and name in lazy and ops.TAPE is None and not guess_mode:
                if publish is not None:
                    publish(r)                 # the UNet reads the FEATURE: ready one launch earlier than the residual was
                return LazyResidual(r, name, scale_of[name], self)
            y = ops.conv2d(r, self.P[name], padding=0, alpha=scale_of[name])
            if publish is not None:
                publish(y)
            return to_nchw_view(y)

        down = [x]
        d = [zero_conv("brushnet_down_blocks.0", x)]
        for i in range(n):                                                                        # :815-828
            for j in range(lpb):
                x = self._resnet(f"down_blocks.{i}.resnets.{j}.", x, temb)
                down.append(x)
                d.append(zero_conv(f"brushnet_down_blocks.{len(d)}", x))
            if i != n - 1:
                x = ops.conv2d(x, self.P[f"down_blocks.{i}.downsamplers.0.conv"], stride=2, padding=1, gn_part=self.config["norm_num_groups"])
                down.append(x)
                d.append(zero_conv(f"brushnet_down_blocks.{len(d)}", x))
        for j in range(2):                                                                        # MidBlock2D
            x = self._resnet(f"mid_block.resnets.{j}.", x, temb)
        m = zero_conv("brushnet_mid_block", x)
        u: List[torch.Tensor] = []
        skips = list(down)
        for i in range(n):                                                                        # :856-887
            for j in range(lpb + 1):
                x = self._resnet(f"up_blocks.{i}.resnets.{j}.", x, temb, x1=skips.pop())
                u.append(zero_conv(f"brushnet_up_blocks.{len(u)}", x))
            if i != n - 1:
                x = ops.conv2d(x, self.P[f"up_blocks.{i}.upsamplers.0.conv"], upsample=True, gn_part=self.config["norm_num_groups"])
                u.append(zero_conv(f"brushnet_up_blocks.{len(u)}", x))
        return d, m, u

    __call__ = forward


# =================================================================================================
class UNet2DConditionModel(_UNetCore):
    """models/unets/unet_2d_condition.py with the BrushNet injection kwargs."""

    _class_name = "UNet2DConditionModel"

    def __init__(self, config=None, precision="bf16", device="cuda", **kwargs):
        cfg = dict(config or {})
        cfg.update(kwargs)
        cfg.setdefault("out_channels", 4)
        cfg.setdefault("down_block_types", ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",))
        cfg.setdefault("up_block_types", ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3)
        cfg.setdefault("mid_block_type", "UNetMidBlock2DCrossAttn")
        super().__init__(cfg, precision, device)
        self._common_defaults()
        c = self.config
        ok_d = {"CrossAttnDownBlock2D", "DownBlock2D"}
        ok_u = {"CrossAttnUpBlock2D", "UpBlock2D"}
        if not set(c["down_block_types"]) <= ok_d or not set(c["up_block_types"]) <= ok_u \
                or c["mid_block_type"] != "UNetMidBlock2DCrossAttn":
            raise NotImplementedError("UNet2DConditionModel: only the SD1.5 block zoo is built "
                                      "(CrossAttn{Down,Up}Block2D, {Down,Up}Block2D, UNetMidBlock2DCrossAttn)")

    def param_shapes(self):
        c = self.config
        out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        boc = c["block_out_channels"]
        temb = boc[0] * 4
        cross = c["cross_attention_dim"]
        depth = _as_tuple(c["transformer_layers_per_block"], len(boc))
        out["conv_in.weight"] = (boc[0], c["in_channels"], 3, 3); out["conv_in.bias"] = (boc[0],)
        out["time_embedding.linear_1.weight"] = (temb, boc[0]); out["time_embedding.linear_1.bias"] = (temb,)
        out["time_embedding.linear_2.weight"] = (temb, temb); out["time_embedding.linear_2.bias"] = (temb,)
        _add_embedding_shapes(out, c, temb)
        lpb = c["layers_per_block"]
        n = len(boc)
        skip = [boc[0]]
        ch = boc[0]
        for i, bt in enumerate(c["down_block_types"]):
            for j in range(lpb):
                _resnet_shapes(out, f"down_blocks.{i}.resnets.{j}.", ch if j == 0 else boc[i], boc[i], temb)
                if bt == "CrossAttnDownBlock2D":
                    _transformer_shapes(out, f"down_blocks.{i}.attentions.{j}.", boc[i], cross, depth[i], c["use_linear_projection"])
                skip.append(boc[i])
            ch = boc[i]
            if i != n - 1:
                out[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (ch, ch, 3, 3)
                out[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (ch,)
                skip.append(ch)
        _resnet_shapes(out, "mid_block.resnets.0.", boc[-1], boc[-1], temb)
        _transformer_shapes(out, "mid_block.attentions.0.", boc[-1], cross, depth[-1], c["use_linear_projection"])
        _resnet_shapes(out, "mid_block.resnets.1.", boc[-1], boc[-1], temb)
        rev = list(reversed(boc))
        rdepth = list(reversed(depth))
        prev = rev[0]
        for i, bt in enumerate(c["up_block_types"]):
            oc = rev[i]
            for j in range(lpb + 1):
                sk = skip.pop()
                _resnet_shapes(out, f"up_blocks.{i}.resnets.{j}.", (prev if j == 0 else oc) + sk, oc, temb)
                if bt == "CrossAttnUpBlock2D":
                    _transformer_shapes(out, f"up_blocks.{i}.attentions.{j}.", oc, cross, rdepth[i], c["use_linear_projection"])
            prev = oc
            if i != n - 1:
                out[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (oc, oc, 3, 3)
                out[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (oc,)
        out["conv_norm_out.weight"] = (boc[0],); out["conv_norm_out.bias"] = (boc[0],)
        out["conv_out.weight"] = (c["out_channels"], boc[0], 3, 3); out["conv_out.bias"] = (c["out_channels"],)
        return out

    def lazy_injection_names(self) -> set:
        """The BrushNet zero-convs whose residual this UNet adds in a Transformer2DModel's proj_out epilogue (the blocks with
        attention): the pipeline asks BrushNet to hand those over as LazyResidual objects (see there)."""
        c = self.config
        if c["use_linear_projection"]:
            return set()
        n, lpb = len(c["block_out_channels"]), c["layers_per_block"]
        names, k = set(), 1                                            # down residual 0 is added to conv_in's output
        for i, bt in enumerate(c["down_block_types"]):
            for j in range(lpb):
                if bt == "CrossAttnDownBlock2D":
                    names.add(f"brushnet_down_blocks.{k}")
                k += 1
            if i != n - 1:
                k += 1
        k = 0
        for i, bt in enumerate(c["up_block_types"]):
            for j in range(lpb + 1):
                if bt == "CrossAttnUpBlock2D":
                    names.add(f"brushnet_up_blocks.{k}")
                k += 1
            if i != n - 1:
                k += 1
        return names

    def _prepare(self, sd):
        c = self.config
        self.P = {}
        self.tdepth = {}
        self.cin_pad = (c["in_channels"] + 7) // 8 * 8
        self.P["conv_in"] = self._conv(sd, "conv_in", cin_pad=self.cin_pad)
        self._prepare_time(sd)
        for k in self.param_shapes():
            if k.endswith(".norm1.weight") and ".resnets." in k:
                self._prepare_resnet(sd, k[: -len("norm1.weight")])
            elif k.endswith("samplers.0.conv.weight"):
                self.P[k[: -len(".weight")]] = self._conv(sd, k[: -len(".weight")])
            elif k.endswith(".proj_in.weight"):
                self._prepare_transformer(sd, k[: -len("proj_in.weight")])
        self.P["conv_norm_out"] = self._norm(sd, "conv_norm_out")
        self.P["conv_out"] = self._conv(sd, "conv_out")
        self._cross_kv, self._ehs_key, self._ehs_gen, self._ehs_val, self._ehs_ref = {}, None, 0, None, None

    @_scoped_tune_ctx
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, class_labels=None,
                timestep_cond=None, attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                down_intrablock_additional_residuals=None, encoder_attention_mask=None, return_dict: bool = True,
                down_block_add_samples: Optional[List[torch.Tensor]] = None,
                mid_block_add_sample: Optional[torch.Tensor] = None,
                up_block_add_samples: Optional[List[torch.Tensor]] = None, *, _temb: Optional[torch.Tensor] = None):
        """unet_2d_condition.py:1039-1348.  `down_block_add_samples` / `up_block_add_samples` are consumed with
        pop(0) exactly like the reference does (the caller's lists are emptied)."""
        if not self._ready:
            raise RuntimeError("UNet2DConditionModel has no parameters loaded")
        if any(v is not None for v in (class_labels, timestep_cond, attention_mask,
                                       down_block_additional_residuals, mid_block_additional_residual,
                                       down_intrablock_additional_residuals, encoder_attention_mask)) \
                or (cross_attention_kwargs not in (None, {})):
            raise NotImplementedError("ControlNet / T2I-adapter / mask / LoRA-scale inputs are outside the MirrorFusion hot path")
        c = self.config
        n = len(c["block_out_channels"])
        lpb = c["layers_per_block"]
        is_brushnet = down_block_add_samples is not None and mid_block_add_sample is not None \
            and up_block_add_samples is not None                                                    # :1202
        bsz = sample.shape[0]
        temb = self._temb_rows(_temb, timestep, bsz, added_cond_kwargs)
        ehs = self._bind_prompt(encoder_hidden_states)
        x = from_nchw(sample.to(self.device), self.prec, self.cin_pad)
        if ops.TAPE is not None:
            ops.TAPE.no_grad(x, ehs)
        hip.TUNE_CTX = "e" if ops.TAPE is None else None
        x = ops.conv2d(x, self.P["conv_in"], gn_part=self.config["norm_num_groups"])
        skips = [x]                                                                                 # :1215 pre-add
        if is_brushnet:
            x = ops.add(x, self._inj(down_block_add_samples.pop(0)), self.prec.act)                 # :1218

        def take(lst, keep_lazy=False):
            return self._inj(lst.pop(0), keep_lazy) if (is_brushnet and len(lst) > 0) else None

        for i, bt in enumerate(c["down_block_types"]):
            has_attn = bt == "CrossAttnDownBlock2D"
            for j in range(lpb):
                inj = take(down_block_add_samples, has_attn) if is_brushnet else None
                if has_attn:
                    x = self._resnet(f"down_blocks.{i}.resnets.{j}.", x, temb)
                    x = self._transformer(f"down_blocks.{i}.attentions.{j}.", x, ehs, self._heads(i), inj)
                else:
                    x = self._resnet(f"down_blocks.{i}.resnets.{j}.", x, temb, inj=inj)
                skips.append(x)                                                                     # post-add (:1388-1391)
            if i != n - 1:
                inj = take(down_block_add_samples) if is_brushnet else None
                x = ops.conv2d(x, self.P[f"down_blocks.{i}.downsamplers.0.conv"], stride=2, padding=1, res1=inj, gn_part=self.config["norm_num_groups"])
                skips.append(x)
        x = self._resnet("mid_block.resnets.0.", x, temb)
        x = self._transformer("mid_block.attentions.0.", x, ehs, self._heads(n - 1))
        x = self._resnet("mid_block.resnets.1.", x, temb,
                         inj=self._inj(mid_block_add_sample) if is_brushnet else None)              # :1288-1289
        if hip.TUNE_CTX is not None:
            hip.TUNE_CTX = "d"
        for i, bt in enumerate(c["up_block_types"]):
            has_attn = bt == "CrossAttnUpBlock2D"
            for j in range(lpb + 1):
                inj = take(up_block_add_samples, has_attn) if is_brushnet else None
                sk = skips.pop()
                if has_attn:
                    x = self._resnet(f"up_blocks.{i}.resnets.{j}.", x, temb, x1=sk)
                    x = self._transformer(f"up_blocks.{i}.attentions.{j}.", x, ehs, self._heads(n - 1 - i), inj)
                else:
                    x = self._resnet(f"up_blocks.{i}.resnets.{j}.", x, temb, x1=sk, inj=inj)
            if i != n - 1:
                inj = take(up_block_add_samples) if is_brushnet else None
                x = ops.conv2d(x, self.P[f"up_blocks.{i}.upsamplers.0.conv"], upsample=True, res1=inj, gn_part=self.config["norm_num_groups"])
        x = ops.groupnorm(x, self.P["conv_norm_out"], groups=c["norm_num_groups"], eps=c["norm_eps"], silu=True,
                          out_dtype=self.prec.act)
        y = ops.conv2d(x, self.P["conv_out"], out_dtype=F32)
        hip.TUNE_CTX = None
        out = hip.unpack_nchw(y, c["out_channels"])
        if ops.TAPE is not None:           # d eps (NCHW) -> d y (NHWC, the conv's own channel count)
            autograd.record_pointwise(ops.TAPE, (y,), out,
                                      lambda g: (hip.pack_nhwc(g.view(out.shape), None, y.shape[-1], F32),))
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    __call__ = forward


# =================================================================================================
class DiagonalGaussianDistribution:
    """vae.py:769-822, moments kept NHWC on the device; `sample` needs explicit or generator noise."""

    def __init__(self, moments_nhwc: torch.Tensor, latent_channels: int):
        self._m = moments_nhwc
        self._c = latent_channels

    @property
    def parameters(self) -> torch.Tensor:
        return hip.unpack_nchw(self._m, 2 * self._c)

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, h, w, _ = self._m.shape
        if noise is None:
            # vae.py:782-791 (randn_tensor): on the generator's device; without one, the host's global RNG
            from .rng import randn_tensor
            noise = randn_tensor((b, self._c, h, w), generator, self._m.device)
        return hip.vae_sample(self._m, noise.to(self._m.device), self._c, 1.0)

    def mode(self) -> torch.Tensor:
        return hip.unpack_nchw(self._m, self._c)


class AutoencoderKL(HipModel):
    """models/autoencoders/autoencoder_kl.py:238-309 + vae.py Encoder/Decoder."""

    _class_name = "AutoencoderKL"

    def __init__(self, config=None, precision="bf16", device="cuda", **kwargs):
        cfg = dict(config or {})
        cfg.update(kwargs)
        cfg.setdefault("in_channels", 3); cfg.setdefault("out_channels", 3); cfg.setdefault("latent_channels", 4)
        cfg.setdefault("block_out_channels", (128, 256, 512, 512)); cfg.setdefault("layers_per_block", 2)
        cfg.setdefault("norm_num_groups", 32); cfg.setdefault("scaling_factor", 0.18215)
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        super().__init__(cfg, precision, device)

    def prepare_training(self, requires_grad=None):
        raise NotImplementedError("the VAE is frozen in MirrorFusion training (train_brushnet_mirror.py:1072) and outside the "
                                  "loss graph: it has no training layout")

    def _convert_deprecated_keys(self, sd):
        # modeling_utils.py:929-971: query/key/value/proj_attn -> to_q/to_k/to_v/to_out.0
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        out = {}
        for k, v in sd.items():
            parts = k.split(".")
            if len(parts) >= 2 and parts[-2] in ren and "attentions" in k:
                parts[-2] = ren[parts[-2]]
                k = ".".join(parts)
            out[k] = v
        return out

    def param_shapes(self):
        c = self.config
        out = OrderedDict()
        boc = c["block_out_channels"]
        n, lpb, lat = len(boc), c["layers_per_block"], c["latent_channels"]

        def mid(p, ch):
            _resnet_shapes(out, p + "resnets.0.", ch, ch, 0)
            a = p + "attentions.0."
            out[a + "group_norm.weight"] = (ch,); out[a + "group_norm.bias"] = (ch,)
            for l in ("to_q", "to_k", "to_v", "to_out.0"):
                out[a + l + ".weight"] = (ch, ch); out[a + l + ".bias"] = (ch,)
            _resnet_shapes(out, p + "resnets.1.", ch, ch, 0)

        out["encoder.conv_in.weight"] = (boc[0], c["in_channels"], 3, 3); out["encoder.conv_in.bias"] = (boc[0],)
        ch = boc[0]
        for i in range(n):
            for j in range(lpb):
                _resnet_shapes(out, f"encoder.down_blocks.{i}.resnets.{j}.", ch if j == 0 else boc[i], boc[i], 0)
            ch = boc[i]
            if i != n - 1:
                out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (ch, ch, 3, 3)
                out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (ch,)
        mid("encoder.mid_block.", boc[-1])
        out["encoder.conv_norm_out.weight"] = (boc[-1],); out["encoder.conv_norm_out.bias"] = (boc[-1],)
        out["encoder.conv_out.weight"] = (2 * lat, boc[-1], 3, 3); out["encoder.conv_out.bias"] = (2 * lat,)
        out["decoder.conv_in.weight"] = (boc[-1], lat, 3, 3); out["decoder.conv_in.bias"] = (boc[-1],)
        mid("decoder.mid_block.", boc[-1])
        rev = list(reversed(boc))
        ch = rev[0]
        for i in range(n):
            for j in range(lpb + 1):
                _resnet_shapes(out, f"decoder.up_blocks.{i}.resnets.{j}.", ch if j == 0 else rev[i], rev[i], 0)
            ch = rev[i]
            if i != n - 1:
                out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (ch, ch, 3, 3)
                out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (ch,)
        out["decoder.conv_norm_out.weight"] = (boc[0],); out["decoder.conv_norm_out.bias"] = (boc[0],)
        out["decoder.conv_out.weight"] = (c["out_channels"], boc[0], 3, 3); out["decoder.conv_out.bias"] = (c["out_channels"],)
        out["quant_conv.weight"] = (2 * lat, 2 * lat, 1, 1); out["quant_conv.bias"] = (2 * lat,)
        out["post_quant_conv.weight"] = (lat, lat, 1, 1); out["post_quant_conv.bias"] = (lat,)
        return out

    def _prepare(self, sd):
        c = self.config
        self.P = {}
        v = self.prec.vec
        for k in self.param_shapes():
            if k.endswith(".norm1.weight") and ".resnets." in k:
                p = k[: -len("norm1.weight")]
                self.P[p + "norm1"] = self._norm(sd, p + "norm1"); self.P[p + "conv1"] = self._conv(sd, p + "conv1")
                self.P[p + "norm2"] = self._norm(sd, p + "norm2"); self.P[p + "conv2"] = self._conv(sd, p + "conv2")
                if p + "conv_shortcut.weight" in sd:
                    self.P[p + "conv_shortcut"] = self._conv(sd, p + "conv_shortcut")
            elif k.endswith("samplers.0.conv.weight"):
                self.P[k[: -len(".weight")]] = self._conv(sd, k[: -len(".weight")])
            elif k.endswith("group_norm.weight"):
                a = k[: -len("group_norm.weight")]
                self.P[a + "group_norm"] = self._norm(sd, a + "group_norm")
                for l in ("to_q", "to_k", "to_v", "to_out.0"):
                    self.P[a + l] = self._conv(sd, a + l)
        self.cin_pad = (c["in_channels"] + 7) // 8 * 8
        lat = c["latent_channels"]
        self.lat_pad = (lat + 7) // 8 * 8
        self.P["encoder.conv_in"] = self._conv(sd, "encoder.conv_in", cin_pad=self.cin_pad)
        self.P["encoder.conv_norm_out"] = self._norm(sd, "encoder.conv_norm_out")
        # encoder.conv_out -> quant_conv are two linear maps with nothing in between: keep them separate (exact
        # reference order) but give quant_conv an 8-aligned input by padding conv_out's N to 8 with zero rows
        w, b = sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"]
        mom_pad = (2 * lat + 7) // 8 * 8
        wp = torch.zeros(mom_pad, *w.shape[1:]); wp[: 2 * lat] = w
        bp = torch.zeros(mom_pad); bp[: 2 * lat] = b
        self.P["encoder.conv_out"] = ConvWeight(wp, bp, self.prec, self.device)
        self.P["quant_conv"] = self._conv(sd, "quant_conv", cin_pad=mom_pad)
        # post_quant_conv output feeds decoder.conv_in: pad its N to 8 the same way
        w, b = sd["post_quant_conv.weight"], sd["post_quant_conv.bias"]
        wp = torch.zeros(self.lat_pad, *w.shape[1:]); wp[:lat] = w
        bp = torch.zeros(self.lat_pad); bp[:lat] = b
        self.P["post_quant_conv"] = ConvWeight(wp, bp, self.prec, self.device, cin_pad=self.lat_pad)
        self.P["decoder.conv_in"] = self._conv(sd, "decoder.conv_in", cin_pad=self.lat_pad)
        self.P["decoder.conv_norm_out"] = self._norm(sd, "decoder.conv_norm_out")
        self.P["decoder.conv_out"] = self._conv(sd, "decoder.conv_out")

    # ---- blocks -------------------------------------------------------------------------------------
    def _resnet(self, p, x):
        g = self.config["norm_num_groups"]
        P = self.P
        h = ops.groupnorm(x, P[p + "norm1"], groups=g, eps=1e-6, silu=True, out_dtype=self.prec.act)
        h = ops.conv2d(h, P[p + "conv1"], gn_part=self.config["norm_num_groups"])
        h = ops.groupnorm(h, P[p + "norm2"], groups=g, eps=1e-6, silu=True, out_dtype=self.prec.act)
        sc = ops.conv2d(x, P[p + "conv_shortcut"], padding=0) if p + "conv_shortcut" in P else x
        return ops.conv2d(h, P[p + "conv2"], res0=sc, gn_part=self.config["norm_num_groups"])

    def _mid(self, p, x):
        """UNetMidBlock2D (unet_2d_blocks.py:601-753): resnet, 1-head spatial self-attention, resnet."""
        P = self.P
        x = self._resnet(p + "resnets.0.", x)
        b, hh, ww, c = x.shape
        a = p + "attentions.0."
        n = ops.groupnorm(x, P[a + "group_norm"], groups=self.config["norm_num_groups"], eps=1e-6, silu=False,
                          out_dtype=self.prec.act).view(b, hh * ww, c)
        q = ops.linear(n, P[a + "to_q"])
        k = ops.linear(n, P[a + "to_k"])
        s = hh * ww
        vt = ops.linear_t(n, P[a + "to_v"], (s + 7) // 8 * 8)
        o = ops.attention(q, k, vt, 1, s, 1.0 / (c ** 0.5), self.prec)
        x = ops.linear(o, P[a + "to_out.0"], res0=x.view(b, s, c)).view(b, hh, ww, c)
        return self._resnet(p + "resnets.1.", x)

    def _moments(self, x: torch.Tensor) -> torch.Tensor:
        c = self.config
        n = len(c["block_out_channels"])
        hip.TUNE_CTX = None                                      # the VAE's tune keys carry no position tag
        h = from_nchw(hip.h2d(x, self.device).float(), self.prec, self.cin_pad)
        h = ops.conv2d(h, self.P["encoder.conv_in"], gn_part=self.config["norm_num_groups"])
        for i in range(n):
            for j in range(c["layers_per_block"]):
                h = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}.", h)
            if i != n - 1:   # Downsample2D(padding=0): asymmetric (0,1,0,1) pad (downsampling.py:140-142)
                h = ops.conv2d(h, self.P[f"encoder.down_blocks.{i}.downsamplers.0.conv"], stride=2, padding=(0, 0, 1, 1), gn_part=self.config["norm_num_groups"])
        h = self._mid("encoder.mid_block.", h)
        h = ops.groupnorm(h, self.P["encoder.conv_norm_out"], groups=c["norm_num_groups"], eps=1e-6, silu=True,
                          out_dtype=self.prec.act)
        h = ops.conv2d(h, self.P["encoder.conv_out"])
        return ops.conv2d(h, self.P["quant_conv"], padding=0, out_dtype=F32)

    @_scoped_tune_ctx
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        d = DiagonalGaussianDistribution(self._moments(x), self.config["latent_channels"])
        return AutoencoderKLOutput(latent_dist=d) if return_dict else (d,)

    @_scoped_tune_ctx
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        c = self.config
        n = len(c["block_out_channels"])
        hip.TUNE_CTX = None                                      # the VAE's tune keys carry no position tag
        h = from_nchw(z.to(self.device).float(), self.prec, self.lat_pad)
        h = ops.conv2d(h, self.P["post_quant_conv"], padding=0)
        h = ops.conv2d(h, self.P["decoder.conv_in"], gn_part=self.config["norm_num_groups"])
        h = self._mid("decoder.mid_block.", h)
        for i in range(n):
            for j in range(c["layers_per_block"] + 1):
                h = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}.", h)
            if i != n - 1:
                h = ops.conv2d(h, self.P[f"decoder.up_blocks.{i}.upsamplers.0.conv"], upsample=True, gn_part=self.config["norm_num_groups"])
        h = ops.groupnorm(h, self.P["decoder.conv_norm_out"], groups=c["norm_num_groups"], eps=1e-6, silu=True,
                          out_dtype=self.prec.act)
        y = ops.conv2d(h, self.P["decoder.conv_out"], out_dtype=F32)
        img = hip.unpack_nchw(y, c["out_channels"])
        return DecoderOutput(sample=img) if return_dict else (img,)
