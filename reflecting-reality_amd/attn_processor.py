"""`MfhipAttnProcessor`: the HIP attention kernels behind the reference's own operator ABI.

The one operator plug-in point the reference has is `Attention.set_processor` / `model.set_attn_processor(proc)`
(models/attention_processor.py:216; brushnet.py:558-590; unets/unet_2d_condition.py:716-748): any callable
`proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0) -> Tensor` replaces
`AttnProcessor2_0` (attention_processor.py:1213-1286).  A maintainer who keeps the reference's torch modules on a ROCm
device installs this class to run the `softmax(q k^T / sqrt(d)) v` core on `mf_attention_bf16` (bf16 inputs, head dims
8 / 40 / 64 / 80 / 160) or on the strided-batched `mf_gemm_conv` + `mf_softmax_rows` path (fp32, other head dims); the
projections, norms and the residual stay the module's own torch layers, exactly as in the reference processor.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .ops import Precision


class MfhipAttnProcessor:
    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None, scale: float = 1.0
                 ) -> torch.Tensor:
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed on the MirrorFusion path (SURVEY.md §8 a-9)")
        if not hidden_states.is_cuda:
            raise RuntimeError("MfhipAttnProcessor needs tensors on a ROCm device: there is no CPU fallback")
        residual = hidden_states
        if getattr(attn, "spatial_norm", None) is not None:
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:                                                  # the VAE's spatial attention (:1228-1230)
            b, c, hh, ww = hidden_states.shape
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        if getattr(attn, "group_norm", None) is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        q = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        elif getattr(attn, "norm_cross", None):
            encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
        k = attn.to_k(encoder_hidden_states)
        v = attn.to_v(encoder_hidden_states)
        heads = attn.heads
        inner = k.shape[-1]
        d = inner // heads
        skv = k.shape[1]
        prec = Precision.get("bf16" if q.dtype == torch.bfloat16 else "fp16" if q.dtype == torch.float16 else "fp32")
        ld = (skv + 7) // 8 * 8
        vt = torch.zeros(v.shape[0], inner, ld, dtype=prec.act, device=v.device)      # V^T, keys contiguous
        vt[:, :, :skv] = v.to(prec.act).transpose(1, 2)
        o = ops.attention(q.to(prec.act).contiguous(), k.to(prec.act).contiguous(), vt, heads, skv, 1.0 / (d ** 0.5), prec)
        o = o.to(q.dtype)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
        if input_ndim == 4:
            o = o.transpose(-1, -2).reshape(b, c, hh, ww)
        if getattr(attn, "residual_connection", False):
            o = o + residual
        return o / getattr(attn, "rescale_output_factor", 1.0)
