"""Layer-level operators of the MirrorFusion hot path on top of the libmfhip C ABI.

Tensors are NHWC ([B, H, W, C]; the same memory as token-major [B, H*W, C]).  ``Precision`` picks the
compute mode of a whole model:
  "bf16"   bf16 MFMA operands and activation storage, fp32 accumulate / statistics — the fast path;
  "fp32"   fp32 MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate) — the exact parity path;
  "f16x3"  fp32 storage, every GEMM operand split into two fp16 halves and multiplied with three fp16 MFMAs
           (22 significant bits): the parity mode that runs on the fast matrix pipe.  "split" / "parity" are
           aliases.  "bf16x3" is the same scheme on bf16 halves (16 bits, fp32 exponent range).
  "bf16x1" fp32 storage, every GEMM operand rounded to bf16 (nearest-even) for ONE bf16 MFMA per product: the
           arithmetic of the reference's --mixed_precision=bf16 training (no loss scaling needed); training only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple, Union

import os

import torch

from . import autograd, hip

# While a tape is installed (training.train_step does that around the forward pass) every operator below also records its
# backward closure on it (autograd.py).  None = inference: nothing is recorded, nothing is kept alive.
TAPE: Optional["autograd.Tape"] = None


@dataclass(frozen=True)
class Precision:
    name: str                     # "bf16" | "fp32" | "f16x3" | "bf16x3" | "bf16x1" | "fp8"
    compute: torch.dtype          # dtype of the GEMM operands in memory
    act: torch.dtype              # activation storage dtype
    code: int = -1                # mf_gemm_desc.dtype (MF_BF16 / MF_F32 / MF_F16X3 / MF_BF16X3 / MF_BF16X1)
    fp8_linear: bool = False      # "fp8": the transformer blocks' Linear layers run on fp8 e4m3 operands (the rest bf16)

    @staticmethod
    def get(name: Union[str, "Precision", torch.dtype]) -> "Precision":
        if isinstance(name, Precision):
            return name
        if name in ("bf16", torch.bfloat16):
            return Precision("bf16", torch.bfloat16, torch.bfloat16, hip.MF_BF16)
        if name in ("fp32", "f32", torch.float32):
            return Precision("fp32", torch.float32, torch.float32, hip.MF_F32)
        if name in ("f16x3", "split", "parity"):
            return Precision("f16x3", torch.float32, torch.float32, hip.MF_F16X3)
        if name == "bf16x3":
            return Precision("bf16x3", torch.float32, torch.float32, hip.MF_BF16X3)
        if name in ("bf16x1", "mixed-bf16"):
            return Precision("bf16x1", torch.float32, torch.float32, hip.MF_BF16X1)
        if name == "fp8":
            return Precision("fp8", torch.bfloat16, torch.bfloat16, hip.MF_BF16, True)
        if name in ("fp16", "f16", torch.float16):
            # the reference's scripts default to fp16 (examples/brushnet/test_brushnet.py:122-126, --mixed_precision fp16 of the
            # training script's validation): fp16 storage, one f16 MFMA per product — the bf16 kernels' byte layout and speed
            # at 11 significant bits instead of 8 (range |x| < 65504, like the reference's own fp16 run)
            return Precision("fp16", torch.float16, torch.float16, hip.MF_F16)
        raise ValueError(f"unsupported precision {name!r} (use 'bf16', 'fp16', 'fp8', 'fp32', 'f16x3', 'bf16x3' or 'bf16x1')")

    @property
    def vec(self) -> int:         # elements per 16-byte vector: channel counts must be multiples of this
        return 8 if self.compute in (torch.bfloat16, torch.float16) else 4

    @property
    def half(self) -> bool:       # 16-bit storage and operands, one MFMA per product ("bf16", "fp16", "fp8"'s non-Linear layers)
        return self.code in (hip.MF_BF16, hip.MF_F16)

    @property
    def split(self) -> bool:      # three MFMAs per product on (hi, lo) halves: weights are pre-split, gradients loss-scaled
        return self.code in (hip.MF_F16X3, hip.MF_BF16X3)

    @property
    def tape_code(self) -> int:   # contraction mode of the backward GEMMs (fp32 storage modes only)
        return self.code if self.code in (hip.MF_F16X3, hip.MF_BF16X3, hip.MF_BF16X1) else hip.MF_F32


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class ConvWeight:
    """Conv2d / Linear parameters re-laid-out once for the implicit-GEMM kernel: [N][kh][kw][Cin_pad]."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], prec: Precision, device,
                 cin_pad: Optional[int] = None, raw: bool = False, fp8: bool = False, ln=None):
        """`ln` = (gamma, beta, eps) of a LayerNorm whose output this Linear consumes (attention.py:203,233,261): the norm is
        FOLDED into the GEMM — W' = W diag(gamma), bias' = bias + W beta, ln_colsum[n] = sum_k W'[n][k] (of the rounded W', so the
        epilogue's mean correction cancels what the matrix pipe accumulated) — and mf_gemm_conv normalises in its epilogue."""
        self.fp8, self.w_scale = False, None
        self.ln_colsum, self.ln_eps = None, 0.0
        if weight.dim() == 2:
            weight = weight[:, :, None, None]
        if ln is not None:
            if fp8 or raw or prec.name not in ("bf16", "fp16") or weight.shape[2:] != (1, 1):
                raise hip.MfhipError("a folded LayerNorm needs a plain bf16 / fp16 Linear")
            g, b, eps = ln
            w2 = weight.detach().float().reshape(weight.shape[0], -1)
            bias = (bias.detach().float() if bias is not None else 0.0) + w2 @ b.detach().float().to(w2.device)
            weight = (w2 * g.detach().float().to(w2.device)[None, :])[:, :, None, None]
            self.ln_eps = float(eps)
        n, cin, kh, kw = weight.shape
        cp = cin_pad if cin_pad is not None else _round_up(cin, prec.vec)
        w = weight.detach().to(device=device, dtype=torch.float32).permute(0, 2, 3, 1)   # [N, kh, kw, Cin]
        if cp != cin:
            w = torch.nn.functional.pad(w, (0, cp - cin))
        w = w.reshape(n, kh * kw * cp)
        self.w_split, self.ldw = 0, kh * kw * cp
        if fp8:
            # per-output-channel symmetric quantisation to OCP e4m3: w ~ q * w_scale[n], |q| <= 448
            if (kh * kw * cp) % 16 != 0:
                raise hip.MfhipError("fp8 weights need K % 16 == 0")
            amax = w.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
            self.w_scale = (amax / 448.0).reshape(n).contiguous()
            self.w = (w / (amax / 448.0)).to(hip.FP8).contiguous()
            self.fp8 = True
        elif prec.split and not raw:
            # split ahead of time: per block of 32 k, [32 high halves | 32 low halves] (the same 128 bytes as 32
            # floats); rows zero-padded to whole blocks.  `raw` keeps fp32 (the weight is the A operand: linear_t)
            self.w, self.ldw = split_pack(w, prec.code)
            self.w_split = 1
        else:
            self.w = w.to(prec.compute).contiguous()
        if ln is not None:
            self.ln_colsum = self.w.float().sum(1).contiguous()
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.n, self.cin, self.cin_pad, self.kh, self.kw = n, cin, cp, kh, kw
        self.prec = prec
        self.p_w = self.p_bias = None          # autograd.Param views when the weight lives in a training arena
        self._gen_src = None

    @classmethod
    def from_params(cls, p_w: "autograd.Param", p_bias: Optional["autograd.Param"], prec: Precision, n: int, cin: int,
                    cin_pad: int, kh: int, kw: int, gen_src) -> "ConvWeight":
        """A weight whose storage is a view of a model's flat fp32 arena ([N][kh*kw*cin_pad], never pre-split: the
        optimizer updates it in place every step)."""
        self = cls.__new__(cls)
        self.fp8, self.w_scale = False, None
        self.ln_colsum, self.ln_eps = None, 0.0
        self.w, self.bias = p_w.data, (p_bias.data if p_bias is not None else None)
        self.w_split, self.ldw = 0, kh * kw * cin_pad
        self.n, self.cin, self.cin_pad, self.kh, self.kw = n, cin, cin_pad, kh, kw
        self.prec = prec
        self.p_w, self.p_bias, self._gen_src = p_w, p_bias, gen_src
        return self

    def operand(self):
        """(weight tensor, w_split, ldw) for mf_gemm_conv.  A weight that lives in a training arena is fp32 (the optimizer updates
        it in place); under a split precision its (hi, lo) halves are packed on the device once per weight generation — every
        step for a network that trains, once for a frozen one — because the pre-split GEMM forms are 25-30 % faster than
        splitting the weight tile in registers (profiles/r03_presplit_weights_microbench.txt)."""
        if self.p_w is None or not PRESPLIT_TRAINING or self.prec.code not in (hip.MF_F16X3, hip.MF_BF16X3):
            return self.w, self.w_split, self.ldw
        gen = self.generation()
        recapture = CAPTURE_TOKEN is not None and self.trains() and getattr(self, "_wp_tok", None) is not CAPTURE_TOKEN
        if getattr(self, "_wp_gen", None) != gen or getattr(self, "_wp", None) is None or recapture:
            self._wp, self._wp_ld = hip.split_pack(self.w.view(self.n, self.ldw), self.prec.code, out=getattr(self, "_wp", None))
            self._wp_gen, self._wp_tok = gen, CAPTURE_TOKEN
        return self._wp, 1, self._wp_ld

    def fast16(self, *channels: int) -> bool:
        """MF_BF16X1 (fp32 storage, every GEMM operand rounded to bf16, fp32 accumulate) on PRE-ROUNDED COPIES: the activation is
        cast once (mf_cast_bf16), the weight once per generation, and the product runs on the LDS-DMA, warp-specialised bf16
        kernels of the inference path instead of the register-staged forms that round in the main loop.  Same arithmetic per
        product (nearest-even bf16 operands, fp32 accumulate and output).  Needs 16-byte rows: channel counts that are multiples
        of 8 (everything but the 4- / 10-channel stems)."""
        return (BF16X1_FAST and self.prec.code == hip.MF_BF16X1 and not self.fp8 and self.w.dtype == torch.float32 and self.ldw % 8 == 0
                and all(c % 8 == 0 for c in channels))

    def operand_bf16(self) -> torch.Tensor:
        """The bf16 copy of the fp32 weight ([N][ldw]), rebuilt when the weights changed (per step for a network that trains)."""
        src = self._gen_src
        if self.p_w is not None and getattr(src, "flat_w", None) is not None and hasattr(src, "weights_bf16"):
            # a weight of a model's training arena: a view of the arena's bf16 twin (one cast launch per generation for ALL weights)
            off = (self.w.data_ptr() - src.flat_w.data_ptr()) // 4
            return src.weights_bf16()[off: off + self.n * self.ldw].view(self.n, self.ldw)
        gen = self.generation()
        recapture = CAPTURE_TOKEN is not None and self.trains() and getattr(self, "_wb_tok", None) is not CAPTURE_TOKEN
        if getattr(self, "_wb_gen", None) != gen or getattr(self, "_wb", None) is None or recapture:
            self._wb = hip.cast_bf16(self.w.reshape(self.n, self.ldw), out=getattr(self, "_wb", None))
            self._wb_gen, self._wb_tok = gen, CAPTURE_TOKEN
        return self._wb

    def trains(self) -> bool:
        """The weight lives in an arena the optimizer updates (its derived layouts go stale every step)."""
        return self.p_w is not None and self.p_w.grad is not None

    def generation(self) -> int:
        """Changes whenever the underlying weights do (layouts derived from them are rebuilt then)."""
        return self._gen_src._weights_gen if self._gen_src is not None else 0


PRESPLIT_TRAINING = os.environ.get("MFHIP_NO_PRESPLIT", "0") != "1"      # developer A/B: split arena weights in registers
GN_KEEP_STATS = os.environ.get("MFHIP_GN_RECOMPUTE_STATS", "0") != "1"    # developer A/B: the backward pass recomputes the statistics
BF16X1_FAST = os.environ.get("MFHIP_BF16X1_SLOW", "0") != "1"             # developer A/B: bf16x1 on pre-rounded operand copies
# Set (to a fresh object) by training.GraphedTrainStep while it captures: every per-step re-layout of a weight that trains
# (operand() here, autograd._dgrad_weight) is rebuilt once under the capture regardless of its generation stamp, so that the
# rebuild is part of the graph.
CAPTURE_TOKEN = None


def split_pack(w: torch.Tensor, code: int):
    """fp32 [N, K] -> (packed 16-bit tensor [N, 2*Kp], Kp) in mf_gemm_desc's w_split layout: hi = round(w) to the
    16-bit type, lo = round(w - hi); per block of 32 k the 32 hi values precede the 32 lo values."""
    n, k = w.shape
    kp = _round_up(k, 32)
    if kp != k:
        w = torch.nn.functional.pad(w, (0, kp - k))
    half = torch.float16 if code == hip.MF_F16X3 else torch.bfloat16
    hip.split_pack_check(w, code)
    hi = w.to(half)
    lo = (w - hi.float()).to(half)
    packed = torch.cat([hi.view(n, kp // 32, 32), lo.view(n, kp // 32, 32)], dim=2).reshape(n, 2 * kp).contiguous()
    return packed, kp


def _shared_rows(res1: Optional[torch.Tensor], m_out: int, n: int) -> int:
    """mf_gemm_desc.res1_rows for a residual that has fewer rows than the output: one residual shared by batch
    replicas (BrushNet's residual for both halves of a classifier-free-guidance batch).  0 = one row per output row."""
    if res1 is None:
        return 0
    rows = res1.numel() // n
    if rows == m_out:
        return 0
    if rows <= 0 or m_out % rows:
        raise hip.MfhipError(f"residual with {rows} rows cannot be shared by an output of {m_out} rows")
    if TAPE is not None:
        raise hip.MfhipError("training: shared (batch-replicated) residuals are inference only")
    return rows


def _want_gn_part(asked: bool, hw: int, n: int) -> bool:
    """GroupNorm statistics from the producing launch: inference only, where GroupNorm runs as separate statistics / apply
    launches (hw > 256: below that one launch keeps the rows in registers), shapes the C ABI takes."""
    # (hw <= 4096: the finalize launch walks an image's hw / rows partial blocks serially per channel — 16 to 32 of them at the UNet's
    # 64 x 64 / 32 x 32 levels; the VAE's 512 x 512 level would be 2048, measured 8 ms slower per pass)
    return bool(asked) and TAPE is None and hip.GN_FROM_PARTS and 256 < hw <= 4096 and hw % 32 == 0 and n % 8 == 0


def conv2d(x: torch.Tensor, cw: ConvWeight, *, stride: int = 1,
           padding: Union[int, Tuple[int, int, int, int]] = 1, upsample: bool = False,
           x1: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None,
           res0: Optional[torch.Tensor] = None, res1: Optional[torch.Tensor] = None,
           alpha: float = 1.0, act: int = hip.ACT_NONE, out_dtype: Optional[torch.dtype] = None,
           splitk: int = 0, tile: int = 0, sk_fused: bool = False, gn_part: Union[bool, int] = False,
           defer_reduce: Union[bool, int] = False) -> torch.Tensor:
    """Convolution over NHWC x (optionally cat([x, x1], C) and/or nearest-2x upsampled), fused epilogue
    alpha*(conv + bias + temb[b]) + res0 + res1.  padding = int or (top, left, bottom, right).
    defer_reduce (the consumer's group count): the caller passes the result to groupnorm() with that many groups and to nothing else (a
    resnet's conv1 -> norm2, resnet.py:381-393): on the levels that run split-K (images up to 16 x 16) the launch that sums the K slices is
    then that GroupNorm (mf_gemm_desc.defer_reduce).
    gn_part: the output feeds a GroupNorm — the launch also leaves per-channel partial sums of its output (mf_gemm_desc.gn_part),
    attached to the returned tensor, and groupnorm() then skips its statistics pass (inference, images above 16 x 16)."""
    b, h, w, c0 = x.shape
    c1 = x1.shape[-1] if x1 is not None else 0
    if c0 + c1 != cw.cin_pad:
        raise hip.MfhipError(f"conv2d: input channels {c0}+{c1} != weight channels {cw.cin_pad}")
    if not x.is_contiguous() or (x1 is not None and not x1.is_contiguous()):
        raise hip.MfhipError("conv2d: inputs must be contiguous NHWC")
    pt, pl, pb, pr = (padding,) * 4 if isinstance(padding, int) else padding
    hu, wu = (h * 2, w * 2) if upsample else (h, w)
    ho = (hu + pt + pb - cw.kh) // stride + 1
    wo = (wu + pl + pr - cw.kw) // stride + 1
    out = torch.empty(b, ho, wo, cw.n, dtype=out_dtype or cw.prec.act, device=x.device)
    wt, wsp, wld = cw.operand()
    code, xa, x1a = cw.prec.code, x, x1
    fast = cw.fast16(c0, c1)
    if fast:
        b16 = lambda t: t if (t is None or t.dtype == torch.bfloat16) else hip.cast_bf16(t)
        code, wt, wsp, wld, xa, x1a = hip.MF_BF16, cw.operand_bf16(), 0, cw.ldw, b16(x), b16(x1)
    elif x.dtype == torch.bfloat16 and cw.prec.code == hip.MF_BF16X1:
        raise hip.MfhipError("conv2d: a bf16 activation in the bf16x1 mode needs channel counts that are multiples of 8")
    hip.gemm_conv(xa, wt, out, dtype=code, w_split=wsp, ldw=wld, c0=c0, lda0=c0, a1=x1a, c1=c1, lda1=c1,
                  batch=b, h_in=h, w_in=w, h_out=ho, w_out=wo, kh=cw.kh, kw=cw.kw, stride=stride,
                  pad_t=pt, pad_l=pl, upsample=upsample, n=cw.n, bias=cw.bias,
                  temb=temb, ld_temb=(temb.stride(0) if temb is not None else 0),
                  res0=res0, res1=res1, res1_rows=_shared_rows(res1, b * ho * wo, cw.n), alpha=alpha, act=act, splitk=splitk,
                  tile=tile, sk_fused=sk_fused, gn_part=(gn_part if _want_gn_part(gn_part, ho * wo, cw.n) else False),
                  defer_reduce=bool(defer_reduce and TAPE is None and res0 is None and res1 is None and act == hip.ACT_NONE and not sk_fused
                                    and hip.gn_slab_applies(ho * wo, cw.n, 32 if defer_reduce is True else int(defer_reduce))))
    if TAPE is not None:
        autograd.record_conv(TAPE, x, x1, cw, out, batch=b, h_in=h, w_in=w, h_out=ho, w_out=wo, stride=stride, pad_t=pt, pad_l=pl,
                             upsample=upsample, temb=temb, res0=res0, res1=res1, alpha=alpha, act=act, x16=(xa, x1a) if fast else None)
    return out


def linear(x: torch.Tensor, lw: ConvWeight, *, res0: Optional[torch.Tensor] = None,
           res1: Optional[torch.Tensor] = None, alpha: float = 1.0, act: int = hip.ACT_NONE,
           out_dtype: Optional[torch.dtype] = None, splitk: int = 0, tile: int = 0,
           out: Optional[torch.Tensor] = None, sk_fused: bool = False) -> torch.Tensor:
    """y = x @ W^T + b over the last dim of x ([..., K] contiguous).  An fp8 weight takes x as an (fp8, row scales) pair
    from hip.quantize_rows_fp8 / ops.layernorm(..., fp8=True), or quantises a bf16 / fp32 x itself."""
    if lw.fp8:
        return _linear_fp8(x, lw, res0, res1, alpha, act, out_dtype, out, tile, ldc=None)
    k = x.shape[-1]
    if k != lw.cin_pad:
        raise hip.MfhipError(f"linear: K={k} != weight K={lw.cin_pad}")
    m = x.numel() // k
    if out is None:
        out = torch.empty(*x.shape[:-1], lw.n, dtype=out_dtype or lw.prec.act, device=x.device)
    wt, wsp, wld = lw.operand()
    code, xa = lw.prec.code, x
    fast = x.is_contiguous() and lw.ln_colsum is None and lw.fast16(k)
    if fast:
        code, wt, wsp, wld, xa = hip.MF_BF16, lw.operand_bf16(), 0, lw.ldw, (x if x.dtype == torch.bfloat16 else hip.cast_bf16(x))
    elif x.dtype == torch.bfloat16 and lw.prec.code == hip.MF_BF16X1:
        raise hip.MfhipError("linear: a bf16 activation in the bf16x1 mode needs K % 8 == 0 and a contiguous input")
    hip.gemm_conv(xa, wt, out, dtype=code, w_split=wsp, ldw=wld, c0=k, lda0=k, batch=m, h_in=1, w_in=1,
                  h_out=1, w_out=1, n=lw.n, bias=lw.bias, res0=res0, res1=res1, res1_rows=_shared_rows(res1, m, lw.n), alpha=alpha,
                  act=act, splitk=splitk, tile=tile, ln_colsum=lw.ln_colsum, ln_eps=lw.ln_eps, sk_fused=sk_fused)
    if TAPE is not None:
        if lw.ln_colsum is not None:
            raise hip.MfhipError("training: folded LayerNorms are inference only")
        autograd.record_conv(TAPE, x, None, lw, out, batch=m, h_in=1, w_in=1, h_out=1, w_out=1, stride=1, pad_t=0, pad_l=0,
                             upsample=False, temb=None, res0=res0, res1=res1, alpha=alpha, act=act, x16=(xa, None) if fast else None)
    return out


def _linear_fp8(x, lw: ConvWeight, res0, res1, alpha, act, out_dtype, out, tile, ldc):
    if TAPE is not None:
        raise hip.MfhipError("training: fp8 layers are inference only")
    xq, xs = x if isinstance(x, tuple) else hip.quantize_rows_fp8(x)
    k = xq.shape[-1]
    if k != lw.cin_pad:
        raise hip.MfhipError(f"linear: K={k} != weight K={lw.cin_pad}")
    m = xq.numel() // k
    n_out = lw.n // 2 if act == hip.ACT_GEGLU4 else lw.n
    if out is None:
        out = torch.empty(*xq.shape[:-1], n_out, dtype=out_dtype or lw.prec.act, device=xq.device)
    hip.gemm_conv(xq, lw.w, out, dtype=hip.MF_FP8, c0=k, lda0=k, batch=m, h_in=1, w_in=1, h_out=1, w_out=1, n=lw.n, ldc=ldc or n_out,
                  bias=lw.bias, res0=res0, res1=res1, res1_rows=_shared_rows(res1, m, lw.n), alpha=alpha, act=act, a_scale=xs,
                  w_scale=lw.w_scale,
                  splitk=1 if act == hip.ACT_GEGLU4 else 0, tile=tile)
    return out


def linear_qkv(x: torch.Tensor, lw: ConvWeight, tile: int = 0):
    """Self-attention's three projections as ONE GEMM over x [B, S, C] with lw = cat([to_q, to_k, to_v]) ([3C, C], usually with
    the block's norm1 folded in): returns (qk [B, S, 2C] — q and k as column slices — and V^T [B, C, S]); the V third leaves
    the epilogue transposed (mf_gemm_desc.vt_out), so attention reads keys contiguously and no second launch re-reads x."""
    b, s, k = x.shape
    c = lw.n // 3
    if TAPE is not None or k != lw.cin_pad or lw.n != 3 * c or s % 8 or (2 * c) % 640:
        raise hip.MfhipError("linear_qkv: inference only, [3C, C] weight, tokens % 8 == 0, C % 320 == 0")
    qk = torch.empty(b, s, 2 * c, dtype=lw.prec.act, device=x.device)
    vt = torch.empty(b, c, s, dtype=lw.prec.act, device=x.device)
    hip.gemm_conv(x, lw.w, qk, dtype=lw.prec.code, ldw=lw.ldw, c0=k, lda0=k, batch=b * s, h_in=1, w_in=1, h_out=1, w_out=1,
                  n=lw.n, ldc=2 * c, bias=lw.bias, tile=tile, ln_colsum=lw.ln_colsum, ln_eps=lw.ln_eps, vt_out=vt, vt_n0=2 * c,
                  vt_tokens=s)
    return qk, vt


def geglu_weight(weight: torch.Tensor, bias: torch.Tensor, prec: Precision, device, fp8: bool = False, ln=None) -> ConvWeight:
    """GEGLU.proj ([2*inner, dim]: value rows then gate rows, activations.py:92,100-103) with rows interleaved
    [4 value | 4 gate] so that mf_gemm_conv's 8-channel epilogue lanes hold matching value/gate pairs."""
    inner = weight.shape[0] // 2
    assert inner % 4 == 0
    idx = torch.arange(inner).view(-1, 4)
    order = torch.cat([idx, idx + inner], dim=1).reshape(-1)
    return ConvWeight(weight[order], bias[order], prec, device, fp8=fp8, ln=ln)


def linear_geglu(x: torch.Tensor, lw: ConvWeight, tile: int = 0) -> torch.Tensor:
    """hidden * gelu(gate) of FeedForward's GEGLU in the GEMM epilogue: [..., K] -> [..., inner]."""
    if TAPE is not None:
        raise hip.MfhipError("training: use ops.linear + ops.geglu (the fused epilogue keeps no pre-activation)")
    if lw.fp8:
        return _linear_fp8(x, lw, None, None, 1.0, hip.ACT_GEGLU4, None, None, tile, ldc=lw.n // 2)
    k = x.shape[-1]
    m = x.numel() // k
    inner = lw.n // 2
    out = torch.empty(*x.shape[:-1], inner, dtype=lw.prec.act, device=x.device)
    hip.gemm_conv(x, lw.w, out, dtype=lw.prec.code, w_split=lw.w_split, ldw=lw.ldw, c0=k, lda0=k, batch=m, h_in=1, w_in=1,
                  h_out=1, w_out=1, n=lw.n, ldc=inner, bias=lw.bias, act=hip.ACT_GEGLU4, splitk=1, tile=tile,
                  ln_colsum=lw.ln_colsum, ln_eps=lw.ln_eps)
    return out


def linear_t(x: torch.Tensor, lw: ConvWeight, ld_out: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Transposed projection per batch: out[b][n][s] = sum_k W[n][k] x[b][s][k] (+ bias[n]).

    The weight plays the A operand and the tokens the [N][K] operand, so V^T comes out of the GEMM
    with keys contiguous and coalesced stores (attention wants V^T; no transpose kernel exists).
    x: [B, S, K]; returns [B, n, ld_out] with columns [S, ld_out) left untouched (callers zero them once).
    """
    if lw.fp8:           # weight rows (output channels) are the A operand: a_scale = the weight's scales, w_scale = the tokens'
        xq, xs = x if isinstance(x, tuple) else hip.quantize_rows_fp8(x)
        b, s, k = xq.shape
        if out is None:
            out = torch.zeros(b, lw.n, ld_out, dtype=lw.prec.act, device=xq.device)
        hip.gemm_conv(lw.w, xq, out, dtype=hip.MF_FP8, c0=k, lda0=k, batch=lw.n, h_in=1, w_in=1, h_out=1, w_out=1, ldw=k, n=s,
                      ldc=ld_out, bias=lw.bias, bias_mode=1, nz=b, zdiv=1, a_zs=(0, 0), w_zs=(s * k, 0), o_zs=(lw.n * ld_out, 0),
                      a_scale=lw.w_scale, w_scale=xs, w_scale_zs=s, splitk=1)
        return out
    b, s, k = x.shape
    if TAPE is not None:
        raise hip.MfhipError("training: use ops.linear + ops.transpose_tokens")
    if lw.w_split:
        raise hip.MfhipError("linear_t: the weight is the A operand here, build it with ConvWeight(..., raw=True)")
    if out is None:
        out = torch.zeros(b, lw.n, ld_out, dtype=lw.prec.act, device=x.device)
    hip.gemm_conv(lw.w, x, out, dtype=lw.prec.code, c0=k, lda0=k, batch=lw.n, h_in=1, w_in=1, h_out=1, w_out=1,
                  ldw=k, n=s, ldc=ld_out, bias=lw.bias, bias_mode=1, nz=b, zdiv=1,
                  a_zs=(0, 0), w_zs=(s * k, 0), o_zs=(lw.n * ld_out, 0))
    return out


# ---- operators that exist as separate launches only on the training path (inference fuses them away) -----------------
def groupnorm(x0: torch.Tensor, norm, *, groups: int, eps: float, silu: bool, out_dtype: torch.dtype,
              x1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """hip.groupnorm over `norm` = (gamma, beta) tensors or autograd.Params."""
    g, b = norm
    pg = g if isinstance(g, autograd.Param) else None
    # training: the forward pass keeps every group's (mean, rstd) for the backward pass (one read of x less there)
    stats = torch.empty(x0.shape[0], groups, 2, dtype=torch.float32, device=x0.device) if (TAPE is not None and GN_KEEP_STATS) else None
    out = hip.groupnorm(x0, g.data if pg else g, b.data if pg else b, groups=groups, eps=eps, silu=silu, out_dtype=out_dtype, x1=x1,
                        stats_out=stats)
    if TAPE is not None:
        if pg is None:
            raise hip.MfhipError("training: GroupNorm parameters must be autograd.Params (model built with train=True)")
        autograd.record_groupnorm(TAPE, x0, x1, g, b, out, groups, eps, silu, stats)
    return out


def layernorm(x: torch.Tensor, norm, eps: float, out_dtype: torch.dtype, fp8: bool = False):
    """fp8=True: LayerNorm and per-row fp8 quantisation in one pass; returns the (fp8, row scales) pair an fp8 linear takes."""
    g, b = norm
    pg = g if isinstance(g, autograd.Param) else None
    if fp8 and TAPE is None and x.shape[-1] % 8 == 0 and x.shape[-1] <= 8192:
        return hip.quantize_rows_fp8(x, (g.data if pg else g, b.data if pg else b), eps)
    out = hip.layernorm(x, g.data if pg else g, b.data if pg else b, eps, out_dtype)
    if TAPE is not None:
        if pg is None:
            raise hip.MfhipError("training: LayerNorm parameters must be autograd.Params (model built with train=True)")
        autograd.record_layernorm(TAPE, x, g, b, out, eps)
    return out


def add(a: torch.Tensor, b: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """a + b; b may hold 1/r of a's leading (batch) dimension: it is then shared by the r batch replicas of a."""
    if b.numel() != a.numel():
        r = a.shape[0] // max(b.shape[0], 1)
        if TAPE is not None or b.shape[0] * r != a.shape[0] or b.shape[1:] != a.shape[1:]:
            raise hip.MfhipError(f"add: shapes {tuple(a.shape)} and {tuple(b.shape)} do not match")
        out = torch.empty(a.shape, dtype=out_dtype, device=a.device)
        for i in range(r):
            hip.add(a[i * b.shape[0]:(i + 1) * b.shape[0]], b, out_dtype, out=out[i * b.shape[0]:(i + 1) * b.shape[0]])
        return out
    out = hip.add(a, b, out_dtype)
    if TAPE is not None:
        autograd.record_pointwise(TAPE, (a, b), out, lambda g: (g, g))
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    out = hip.silu_f32(x)
    if TAPE is not None:
        autograd.record_pointwise(TAPE, (x,), out, lambda g: (hip.silu_bwd(x, g.view(x.shape)),))
    return out


def geglu(h: torch.Tensor, out_dtype: torch.dtype, bias: Optional["autograd.Param"] = None) -> torch.Tensor:
    """a * gelu(g) of h = [a | g].  A bf16 `h` (the MF_BF16X1 mode's FeedForward pre-activation) is differentiated into a bf16
    gradient; `bias` = the bias parameter of the Linear that produced h: its gradient (the column sums of dh) leaves the same pass."""
    out = hip.geglu(h, out_dtype)
    tape = TAPE
    if tape is not None and h.dtype == torch.bfloat16:
        def bwd():
            g = tape.take(out)
            if g is None:
                return
            bg = bias.grad if (bias is not None and bias.grad is not None) else None
            dh = hip.geglu_bwd_bf16(h, g.view(-1, out.shape[-1]).contiguous(), bg)
            if bg is not None:
                tape.colsum_done[dh.data_ptr()] = dh
            tape.add(h, dh)
        tape.record(bwd)
    elif tape is not None:
        autograd.record_pointwise(tape, (h,), out, lambda g: (hip.geglu_bwd(h, g.view(out.shape)),))
    return out


def bf16x1_operands(prec: Precision) -> bool:
    """The MF_BF16X1 mode on pre-rounded bf16 operand copies (training)."""
    return TAPE is not None and BF16X1_FAST and prec.code == hip.MF_BF16X1


def flash_bf16_train(prec: Precision, sq: int, d: int) -> bool:
    """Whether ops.attention_train takes the bf16 flash route for these sizes (its q / k / v may then be produced in bf16)."""
    return (bf16x1_operands(prec) and FLASH_BWD and d in FLASH_BWD_BF16_HEAD_DIMS and sq % 4 == 0 and sq >= FLASH_BWD_MIN_TOKENS)


def transpose_tokens(v: torch.Tensor, ld: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """[B, S, C] -> V^T [B, C, ld] (columns past S zero): the training path's stand-in for linear_t.  dtype bf16 rounds."""
    b, s, c = v.shape
    vt = torch.zeros(b, c, ld, dtype=dtype, device=v.device) if ld != s else torch.empty(b, c, ld, dtype=dtype, device=v.device)
    hip.transpose(v, s, c, nz=b, ldx=c, ldy=ld, zsx=s * c, zsy=c * ld, out=vt)
    return vt


def attention_train(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float, prec: Precision) -> torch.Tensor:
    """Unfused attention on [B, S, C] fp32 q / k / v, differentiated as one operator (autograd.record_attention)."""
    b, sq, c = q.shape
    skv = k.shape[1]
    d = c // heads
    ld = (skv + 7) // 8 * 8
    tape = TAPE

    def probs():
        scores = torch.empty(b * heads, sq, ld, dtype=torch.float32, device=q.device)
        hip.gemm_conv(q, k, scores, dtype=prec.code, c0=d, lda0=c, batch=sq, h_in=1, w_in=1, h_out=1, w_out=1, ldw=c, n=skv, ldc=ld,
                      alpha=scale, nz=b * heads, zdiv=heads, a_zs=(sq * c, d), w_zs=(skv * c, d), o_zs=(heads * sq * ld, sq * ld),
                      splitk=1)
        return hip.softmax_rows(scores, skv, torch.float32)

    flash = sq % 4 == 0 and sq >= FLASH_BWD_MIN_TOKENS and tape is not None and FLASH_BWD
    if flash_bf16_train(prec, sq, d):
        # the bf16x1 mode on pre-rounded operands: q / k / v in bf16 (produced so by their projections, or rounded here: what
        # the reference's autocast hands to F.scaled_dot_product_attention, attention_processor.py:1266), the inference flash
        # kernel with the row statistics, and the single-plane flash backward (autograd.record_attention_flash_bf16)
        if not (q.dtype == k.dtype == v.dtype):
            raise hip.MfhipError("attention_train: q / k / v of one dtype")
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        q16, k16, v16 = (q, k, v) if q.dtype == torch.bfloat16 else (hip.cast_bf16(q), hip.cast_bf16(k), hip.cast_bf16(v))
        vt16 = transpose_tokens(v, ld, torch.bfloat16)
        o16 = torch.empty(b, sq, c, dtype=torch.bfloat16, device=q.device)
        lse = torch.empty(b, heads, sq, dtype=torch.float32, device=q.device)
        hip.attention_bf16(q16, k16, vt16, o16, ldq=c, ldk=c, ldvt=ld, ldo=c, batch=b, heads=heads, sq=sq, skv=skv, head_dim=d,
                           scale=scale, lse=lse)
        autograd.record_attention_flash_bf16(tape, q, k, v, o16, lse, heads, scale, q16, k16, v16)
        return o16
    if q.dtype != torch.float32:
        raise hip.MfhipError("attention_train: bf16 q / k / v only on the bf16 flash route (ops.flash_bf16_train)")
    vt = transpose_tokens(v, ld)
    if flash and prec.code in (hip.MF_F16X3, hip.MF_BF16X1) and d in FLASH_BWD_HEAD_DIMS:
        # flash forward WITH the row statistics, flash backward (autograd.record_attention_flash): nothing of size Sq x Skv is
        # ever written in either direction.  The bf16x1 mode (fp32 storage, bf16 products) takes the same split-precision kernels:
        # finer than its own arithmetic, and the S x S tensors of the unfused form are what bounds it at 64 x 64 latents.
        global SPLIT_KERNEL_CALLS
        SPLIT_KERNEL_CALLS += 1
        qs, ks, vs = hip.split_halves(q.contiguous()), hip.split_halves(k.contiguous()), hip.split_halves(vt)
        out = torch.empty(b, sq, c, dtype=torch.float32, device=q.device)
        lse = torch.empty(b, heads, sq, dtype=torch.float32, device=q.device)
        hip.attention_f16x3(qs, ks, vs, out, ldq=c, ldk=c, ldvt=vt.shape[-1], ldo=c, batch=b, heads=heads, sq=sq, skv=skv, head_dim=d,
                            scale=scale, lse=lse)
        autograd.record_attention_flash(tape, q, k, v, out, lse, heads, scale, qs, ks)
        return out
    if prec.code == hip.MF_F16X3 and d in FLASH_SPLIT_HEAD_DIMS:
        # forward on the flash kernel (split precision): P is not materialised here at all; backward recomputes it once
        out = attention(q, k, vt, heads, skv, scale, prec, c=c)
    else:
        p = probs()
        out = torch.empty(b, sq, c, dtype=torch.float32, device=q.device)
        hip.gemm_conv(p, vt, out, dtype=prec.code, c0=ld, lda0=ld, batch=sq, h_in=1, w_in=1, h_out=1, w_out=1, ldw=ld, n=d, ldc=c,
                      nz=b * heads, zdiv=heads, a_zs=(heads * sq * ld, sq * ld), w_zs=(c * ld, d * ld), o_zs=(sq * c, d), splitk=1)
    if tape is not None:
        autograd.record_attention(tape, q, k, v, out, heads, skv, scale, probs)
    return out


def attention_unfused(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, skv: int, scale: float,
                      prec: Precision) -> torch.Tensor:
    """softmax(q k^T * scale) v through two strided-batched GEMMs and a row softmax (scores in fp32).

    q: [B, Sq, C], k: [B, Skv, C], vt: [B, C, ldv] (V^T, pad columns zero).  Used by the fp32 parity
    mode and by head dims the fused kernel does not cover (the VAE's single 512-wide head).
    """
    b, sq, c = q.shape
    d = c // heads
    ldv = vt.shape[-1]
    scores = torch.empty(b * heads, sq, ldv, dtype=torch.float32, device=q.device)
    hip.gemm_conv(q, k, scores, dtype=prec.code, c0=d, lda0=c, batch=sq, h_in=1, w_in=1, h_out=1, w_out=1,
                  ldw=c, n=skv, ldc=ldv, alpha=scale, nz=b * heads, zdiv=heads,
                  a_zs=(sq * c, d), w_zs=(k.shape[1] * c, d), o_zs=(heads * sq * ldv, sq * ldv), splitk=1)
    p = hip.softmax_rows(scores, skv, prec.act)
    out = torch.empty(b, sq, c, dtype=prec.act, device=q.device)
    hip.gemm_conv(p, vt, out, dtype=prec.code, c0=ldv, lda0=ldv, batch=sq, h_in=1, w_in=1, h_out=1, w_out=1,
                  ldw=ldv, n=d, ldc=c, nz=b * heads, zdiv=heads,
                  a_zs=(heads * sq * ldv, sq * ldv), w_zs=(c * ldv, d * ldv), o_zs=(sq * c, d), splitk=1)
    return out


FLASH_BWD_HEAD_DIMS = (8, 40)                    # mf_attention_bwd_f16x3
SPLIT_KERNEL_CALLS = 0      # attention launches on fp16 halves so far (training.py: the bf16x1 mode only needs its fp16 range guard when this moved)
FLASH_BWD_BF16_HEAD_DIMS = (8, 40, 80)           # mf_attention_bwd_bf16 (single planes: 80 fits in LDS)
FLASH_BWD_MIN_TOKENS = 256                       # shorter sequences keep the unfused backward (its S x S tensors are small there)
import os as _os
FLASH_BWD = _os.environ.get("MFHIP_NO_FLASH_BWD") != "1"       # A/B switch
FLASH_HEAD_DIMS = (8, 40, 64, 80, 160)
FLASH_SPLIT_HEAD_DIMS = (8, 40, 64, 80)          # 160 (two split K / V^T planes, double buffered) does not fit in LDS


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, skv: int, scale: float,
              prec: Precision, c: Optional[int] = None) -> torch.Tensor:
    """q: [B, Sq, ldq] and k: [B, Skv, ldk] may be column slices of a fused projection (`c` = model width)."""
    b, sq, ldq = q.shape
    c = c or ldq
    d = c // heads
    if prec.half and d in FLASH_HEAD_DIMS:           # bf16, or fp16 on the f16 MFMA forms (hip.attention_bf16 dispatches on q.dtype)
        out = torch.empty(b, sq, c, dtype=prec.compute, device=q.device)
        return hip.attention_bf16(q, k, vt, out, ldq=q.stride(1), ldk=k.stride(1), ldvt=vt.shape[-1], ldo=c, batch=b,
                                  heads=heads, sq=sq, skv=skv, head_dim=d, scale=scale)
    if prec.code == hip.MF_F16X3 and d in FLASH_SPLIT_HEAD_DIMS and vt.shape[-1] % 8 == 0:
        # the parity mode runs the same flash kernel: operands as (hi, lo) fp16 planes, fp32 output.  q and k may be
        # column slices of one fused projection: split the parent once and slice the planes
        if q.stride(1) != c and q.storage_offset() + c == k.storage_offset() and q.stride(1) == k.stride(1) == 2 * c:
            qk = q.as_strided((b, sq, 2 * c), (sq * 2 * c, 2 * c, 1), q.storage_offset())
            ph, pl = hip.split_halves(qk)
            qs, ks = (ph[..., :c], pl[..., :c]), (ph[..., c:], pl[..., c:])
        else:
            qs, ks = hip.split_halves(q.contiguous()), hip.split_halves(k.contiguous())
        vs = hip.split_halves(vt)
        out = torch.empty(b, sq, c, dtype=torch.float32, device=q.device)
        return hip.attention_f16x3(qs, ks, vs, out, ldq=qs[0].stride(1), ldk=ks[0].stride(1), ldvt=vt.shape[-1], ldo=c,
                                   batch=b, heads=heads, sq=sq, skv=skv, head_dim=d, scale=scale)
    if q.stride(1) != c or k.stride(1) != c:
        q, k = q.contiguous(), k.contiguous()
    return attention_unfused(q, k, vt, heads, skv, scale, prec)
