"""ctypes binding of libmfhip.so (C ABI in include/mfhip.h) on top of torch-ROCm tensors.

PyTorch is used only to own device memory and the HIP stream; every arithmetic op below is a
hand-written gfx950 kernel.  There is NO fallback: if the library is missing or a call fails this
module raises, it never computes on the host or through ATen.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple, Union

import torch

from . import _build

MF_F32, MF_BF16, MF_F16X3, MF_BF16X3, MF_FP8, MF_BF16X1, MF_F16 = 0, 1, 2, 3, 4, 5, 6
FP8 = torch.float8_e4m3fn          # OCP e4m3 (gfx950's fp8), 1 byte per element
ACT_NONE, ACT_SILU, ACT_GEGLU4 = 0, 1, 2
ABI_VERSION = 20


class MfhipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("a0", C.c_void_p), ("a1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("lda0", C.c_int64), ("lda1", C.c_int64),
        ("a_dtype", C.c_int32),
        ("batch", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
        ("h_out", C.c_int32), ("w_out", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("upsample", C.c_int32),
        ("w", C.c_void_p), ("ldw", C.c_int64), ("w_split", C.c_int32),
        ("n", C.c_int32),
        ("nz", C.c_int32), ("zdiv", C.c_int32),
        ("a_zs_o", C.c_int64), ("a_zs_i", C.c_int64), ("w_zs_o", C.c_int64), ("w_zs_i", C.c_int64),
        ("o_zs_o", C.c_int64), ("o_zs_i", C.c_int64),
        ("bias", C.c_void_p), ("bias_mode", C.c_int32),
        ("temb", C.c_void_p), ("ld_temb", C.c_int64),
        ("a_scale", C.c_void_p), ("w_scale", C.c_void_p), ("a_scale_zs", C.c_int64), ("w_scale_zs", C.c_int64),
        ("res0", C.c_void_p), ("res0_dtype", C.c_int32), ("ld_res0", C.c_int64),
        ("res1", C.c_void_p), ("res1_dtype", C.c_int32), ("ld_res1", C.c_int64), ("res1_rows", C.c_int32),
        ("alpha", C.c_float), ("act", C.c_int32),
        ("out", C.c_void_p), ("out_dtype", C.c_int32), ("ldc", C.c_int64),
        ("splitk", C.c_int32), ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("tile", C.c_int32),
        ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float),
        ("vt_out", C.c_void_p), ("vt_n0", C.c_int32), ("vt_tokens", C.c_int32), ("vt_ld", C.c_int64),
        ("sk_tickets", C.c_void_p), ("sk_ticket_cap", C.c_int32),
        ("gn_part", C.c_void_p), ("gn_part_floats", C.c_int64), ("gn_part_rows", C.c_void_p),
        ("gn_groups", C.c_int32), ("gn_grouped", C.c_void_p),
        ("defer_reduce", C.c_int32), ("deferred_splits", C.c_void_p),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("a0", C.c_void_p), ("a1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("lda0", C.c_int64), ("lda1", C.c_int64),
        ("batch", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("upsample", C.c_int32),
        ("dy", C.c_void_p), ("lddy", C.c_int64),
        ("n", C.c_int32),
        ("dw", C.c_void_p), ("lddw", C.c_int64),
        ("accumulate", C.c_int32), ("splitm", C.c_int32),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
    ]


class GroupNormBwdDesc(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p), ("c0", C.c_int32), ("c1", C.c_int32),
        ("dy", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("dx0", C.c_void_p), ("dx1", C.c_void_p),
        ("dgamma_part", C.c_void_p), ("dbeta_part", C.c_void_p),
        ("batch", C.c_int32), ("hw", C.c_int32), ("groups", C.c_int32), ("silu", C.c_int32),
        ("eps", C.c_float),
        ("ws", C.c_void_p),
        ("dgamma_acc", C.c_void_p), ("dbeta_acc", C.c_void_p),
        ("add0", C.c_void_p), ("add1", C.c_void_p),
        ("stats_in", C.c_void_p),
    ]


class AttnBwdDesc(C.Structure):
    _fields_ = [
        ("q_hi", C.c_void_p), ("q_lo", C.c_void_p), ("ldq", C.c_int64),
        ("k_hi", C.c_void_p), ("k_lo", C.c_void_p), ("ldk", C.c_int64),
        ("v_hi", C.c_void_p), ("v_lo", C.c_void_p), ("ldv", C.c_int64),
        ("do_hi", C.c_void_p), ("do_lo", C.c_void_p), ("lddo", C.c_int64),
        ("qt_hi", C.c_void_p), ("qt_lo", C.c_void_p), ("ldqt", C.c_int64),
        ("kt_hi", C.c_void_p), ("kt_lo", C.c_void_p), ("ldkt", C.c_int64),
        ("dot_hi", C.c_void_p), ("dot_lo", C.c_void_p), ("lddot", C.c_int64),
        ("lse", C.c_void_p), ("dd", C.c_void_p),
        ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("ldo", C.c_int64),
        ("batch", C.c_int32), ("heads", C.c_int32), ("sq", C.c_int32), ("skv", C.c_int32), ("head_dim", C.c_int32),
        ("scale", C.c_float), ("out_dtype", C.c_int32),
    ]


class GroupNormDesc(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("in_dtype", C.c_int32),
        ("batch", C.c_int32), ("hw", C.c_int32),
        ("groups", C.c_int32),
        ("eps", C.c_float),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("silu", C.c_int32),
        ("out", C.c_void_p), ("out_dtype", C.c_int32),
        ("ws", C.c_void_p), ("stats_out", C.c_void_p),
        ("part0", C.c_void_p), ("part0_rows", C.c_int32), ("part1", C.c_void_p), ("part1_rows", C.c_int32),
        ("grp0", C.c_void_p), ("grp0_rows", C.c_int32),
        ("sk_ws", C.c_void_p), ("sk_splits", C.c_int32),
        ("sk_bias", C.c_void_p), ("sk_temb", C.c_void_p), ("sk_ld_temb", C.c_int64), ("sk_alpha", C.c_float),
    ]


# every symbol include/mfhip.h declares (tests/test_abi.py checks the header against this list)
EXPORTS = [
    "mf_abi_version", "mf_last_error", "mf_sizeof_gemm_desc", "mf_sizeof_groupnorm_desc",
    "mf_gemm_conv", "mf_gemm_num_tiles", "mf_gemm_tile_shape", "mf_gemm_tile_table_version",
    "mf_groupnorm", "mf_groupnorm_ws_floats", "mf_layernorm", "mf_softmax_rows", "mf_attention_bf16", "mf_attention_f16",
    "mf_attention_f16x3", "mf_attention_f16x3_lse", "mf_sizeof_attn_bwd_desc", "mf_attention_bwd_f16x3", "mf_rowdot_heads",
    "mf_attention_bwd_bf16", "mf_attention_bf16_lse", "mf_rowdot_heads_bf16", "mf_cast_bf16_colsum", "mf_cast_bf16_colsum_ws_floats",
    "mf_transpose_bf16_bf16", "mf_geglu_bwd_bf16", "mf_geglu_bwd_bf16_ws_floats", "mf_rowdot_heads_cast", "mf_debug_set_wgrad_dma", "mf_zero_ranges",
    "mf_split_halves", "mf_split_overflow", "mf_quantize_rows_fp8",
    "mf_pack_nhwc", "mf_unpack_nchw", "mf_add", "mf_cast_bf16", "mf_geglu", "mf_timestep_embedding", "mf_silu_f32",
    "mf_cfg_ddim_step", "mf_cfg_ddim_step_dev", "mf_cfg_combine", "mf_axpby_n", "mf_mse_loss", "mf_vae_sample", "mf_nearest_resize",
    # image front-end (csrc/frontend.hip)
    "mf_minmax_ws_floats", "mf_minmax", "mf_image_normalize", "mf_mask_keep", "mf_concat_channels", "mf_postprocess",
    "mf_depth_normalize", "mf_select_ws_bytes", "mf_select_ranks", "mf_depth_percentile_normalize", "mf_bicubic_resize_crop",
    "mf_bicubic_aa_resize_crop",
    "mf_hwc_to_chw_affine",
    # training (csrc/train.hip)
    "mf_sizeof_wgrad_desc", "mf_conv_wgrad_ws_floats", "mf_conv_wgrad", "mf_split_pack", "mf_transpose", "mf_transpose_bf16", "mf_colsum_ws_floats", "mf_colsum",
    "mf_sizeof_groupnorm_bwd_desc", "mf_groupnorm_bwd", "mf_groupnorm_bwd_ws_floats", "mf_groupnorm_bwd_streams", "mf_layernorm_bwd", "mf_layernorm_bwd_parts", "mf_softmax_bwd", "mf_silu_bwd", "mf_geglu_bwd",
    "mf_zero_insert2x", "mf_sumpool2x2", "mf_mse_grad", "mf_sumsq_ws_doubles", "mf_sumsq", "mf_clip_coef", "mf_adamw",
    # step programs (csrc/program.cpp; program.py records them)
    "mf_memcpy2d", "mf_memset", "mf_program_load", "mf_program_destroy", "mf_program_num_buffers", "mf_program_buffer_info",
    "mf_program_find_buffer", "mf_program_bind", "mf_program_num_calls", "mf_program_meta", "mf_program_run",
    "mf_denoise_step_fused", "mf_unet_forward", "mf_brushnet_forward", "mf_vae_decode", "mf_vae_encode_moments",
]

_lib: Optional[C.CDLL] = None
_RECORDER = None        # program.Recorder's proxy while a step program is being recorded: every launch goes through it


def lib_path() -> str:
    # MFHIP_LIB: a developer build of the same sources (e.g. the stamped one of tools/stamps.py); never set in production
    return os.environ.get("MFHIP_LIB") or _build.LIB_PATH


def load() -> C.CDLL:
    """Load libmfhip.so (after torch, so that its libamdhip64.so.7 is the one HIP runtime in-process)."""
    global _lib
    if _RECORDER is not None:
        return _RECORDER
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise MfhipError(
            f"{path} not found: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). There is no CPU fallback for the MirrorFusion hot path.")
    lib = C.CDLL(path)
    lib.mf_last_error.restype = C.c_char_p
    lib.mf_groupnorm_ws_floats.restype = C.c_int64
    lib.mf_groupnorm_bwd_ws_floats.restype = C.c_int64
    lib.mf_layernorm_bwd_parts.restype = C.c_int64
    lib.mf_layernorm_bwd_parts.argtypes = [C.c_int64]
    for fn in ("mf_conv_wgrad_ws_floats", "mf_colsum_ws_floats", "mf_sumsq_ws_doubles", "mf_minmax_ws_floats", "mf_select_ws_bytes",
               "mf_cast_bf16_colsum_ws_floats", "mf_geglu_bwd_bf16_ws_floats"):
        getattr(lib, fn).restype = C.c_int64
    if lib.mf_abi_version() != ABI_VERSION:
        raise MfhipError(f"libmfhip ABI {lib.mf_abi_version()} != binding ABI {ABI_VERSION}: rebuild the library")
    if lib.mf_sizeof_gemm_desc() != C.sizeof(GemmDesc) or lib.mf_sizeof_groupnorm_desc() != C.sizeof(GroupNormDesc):
        raise MfhipError("descriptor struct layout mismatch between mfhip.h and the ctypes binding")
    if lib.mf_sizeof_attn_bwd_desc() != C.sizeof(AttnBwdDesc):
        raise MfhipError("mf_attn_bwd_desc layout mismatch between mfhip.h and the ctypes binding")
    if lib.mf_sizeof_wgrad_desc() != C.sizeof(WgradDesc) or lib.mf_sizeof_groupnorm_bwd_desc() != C.sizeof(GroupNormBwdDesc):
        raise MfhipError("training descriptor struct layout mismatch between mfhip.h and the ctypes binding")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise MfhipError(f"{what} failed (rc={rc}): {load().mf_last_error().decode()}")


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return MF_F32
    if dtype == torch.bfloat16:
        return MF_BF16
    if dtype == torch.float16:
        return MF_F16
    if dtype == FP8:
        return MF_FP8
    raise MfhipError(f"unsupported dtype {dtype}")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req_cuda(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MfhipError("libmfhip ops need device tensors (there is no CPU path)")
        if t is not None and getattr(t, "_sk_pending", None) is not None:
            raise MfhipError("a tensor whose split-K reduce was deferred (gemm_conv(defer_reduce=True)) reached an op other than the "
                             "groupnorm() it was promised to: its memory has not been written")


# ---- scratch buffers (split-K slabs, GroupNorm statistics) -------------------------------------
_scratch: dict = {}


def scratch(name: str, nfloats: int, device) -> torch.Tensor:
    # one buffer per (purpose, device, stream): launches on different streams may run concurrently
    key = (name, torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(nfloats, 1), dtype=torch.float32, device=device)
        _scratch[key] = buf
    return buf


# tiles whose kernels carry the in-launch combine (csrc/gemm_conv.hip: tile_has_skf), per compute code
SK_FUSED_TILES = {MF_BF16: (1, 2, 3, 6, 41, 43, 44, 48), MF_F16X3: (1, 2, 3, 6, 41, 44)}
SK_TICKETS = 8192          # arrival counters of the in-launch split-K combine (mf_gemm_desc.sk_tickets), per stream
SK_STREAMS = 64
_tickets: dict = {}        # device index -> ([SK_STREAMS, SK_TICKETS] zeroed int32, {stream id: row})


def sk_tickets(device) -> torch.Tensor:
    """The zeroed ticket row of the current stream.  Launches on different streams may run concurrently, so every stream owns
    a row of one per-device table; the kernels leave their tickets zeroed, so the table is cleared once, when it is allocated —
    outside any graph capture (a capture stream only picks a row: no allocation, no memset node)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ent = _tickets.get(idx)
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            raise MfhipError("the split-K ticket table must exist before a graph capture (run the step eagerly once)")
        ent = _tickets[idx] = (torch.zeros(SK_STREAMS, SK_TICKETS, dtype=torch.int32, device=dev), {})
    table, rows = ent
    sid = torch.cuda.current_stream(dev).cuda_stream
    row = rows.get(sid)
    if row is None:
        if len(rows) >= SK_STREAMS:
            raise MfhipError(f"more than {SK_STREAMS} streams issued split-K GEMMs on device {idx}")
        row = rows[sid] = len(rows)
    return table[row]


_staging: dict = {}


def h2d(t: torch.Tensor, device) -> torch.Tensor:
    """Host tensor -> device through a persistent pinned staging buffer.  A freshly allocated pageable tensor
    (every preprocessing result is one) pays page faults + driver pinning on a direct .to(device): 12.6 MB took
    31 ms on the MI355X host, 10x the rest of the conditioning build.  Device tensors pass through."""
    if t.device.type != "cpu":
        return t.to(device)
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes < (1 << 16):
        return t.to(device)
    key = torch.device(device).index
    buf = _staging.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 24), dtype=torch.uint8, pin_memory=True)
        _staging[key] = buf
    stage = buf[:nbytes].view(t.dtype).view(t.shape)
    stage.copy_(t)
    out = stage.to(device, non_blocking=True)
    torch.cuda.current_stream(device).synchronize()      # the staging buffer is reused by the next call
    return out


SPLITK_WS_FLOATS = 16 * 1024 * 1024  # 64 MiB of fp32 slabs

# Optional live profiler used by bench.py: when PROFILE is a list, every mf_gemm_conv launch is bracketed by
# HIP events on the launch stream and appended as (start_event, end_event, algorithmic_flops).
PROFILE = None


PROFILE_ATTN_FLOPS = 0.0      # 4 * batch * heads * Sq * Skv * d of every flash-attention launch between profile_begin / profile_end


def profile_begin():
    global PROFILE, PROFILE_ATTN_FLOPS
    PROFILE = []
    PROFILE_ATTN_FLOPS = 0.0


def profile_end():
    """Returns (n_launches, total_seconds, total_flops) for the bracketed mf_gemm_conv launches."""
    global PROFILE
    rec, PROFILE = PROFILE, None
    torch.cuda.synchronize()
    global LAST_PROFILE
    LAST_PROFILE = [(a.elapsed_time(b) * 1e-3, f, k) for a, b, f, k in rec]
    secs = sum(t for t, _, _ in LAST_PROFILE)
    return len(rec), secs, float(sum(f for _, f, _ in LAST_PROFILE))


LAST_PROFILE = []


# ---- per-shape autotuning of (tile, split-K) ---------------------------------------------------------
# The library's heuristic is only a prior; the host times every instantiated tile (and a few split-K factors
# for grids that cannot fill 256 CUs) the first time a GEMM shape is seen and remembers the winner.  Winners
# are persisted in tune_cache.json next to this file so later processes (and graph capture) start tuned.
AUTOTUNE = os.environ.get("MFHIP_AUTOTUNE", "1") != "0"
# Offer the in-launch split-K combine (mf_gemm_desc.sk_tickets) to the tuner.  OFF by default: measured on MI355X (tools/bench_skf.py,
# profiles/r04_splitk_in_launch.txt) the last-arriver form with an agent-scope release per block is 8-15 us SLOWER than the reduce
# launch it replaces on every small-M shape of the step (each block's release writes back its XCD's L2: ~45 ns per block, serialised).
SK_FUSED = os.environ.get("MFHIP_SK_FUSED", "0") == "1"
# Route the 1x1 GEMMs whose blocks would walk >= 2 output tiles to the persistent tile 70 instead of the cached tile.  A/B switch.
PREFER_PERS = os.environ.get("MFHIP_PREFER_PERS", "1") != "0"
PERS_TILE = 70
PERS_MIN_NLOOP = int(os.environ.get("MFHIP_PERS_MIN_NLOOP", "2"))


def gn_slab_applies(hw: int, c: int, groups: int) -> bool:
    """Whether mf_groupnorm runs its one-launch form on [*, hw, c] in `groups` groups, one segment (csrc/norm.hip, the dispatch of
    gn_slab_kernel) — the form a deferred split-K reduce needs (mf_groupnorm_desc.sk_ws)."""
    if groups <= 0 or c % groups or c % 8:
        return False
    cpg = c // groups
    sc = cpg
    while sc % 8:
        sc += cpg
    p = hw if hw < 64 else 64
    return hw <= 4 * p and c % sc == 0 and sc // cpg <= 8 and (sc // 8) * p <= 1024 and p * sc * 8 <= 64 * 1024


def _pers_applies(m, n, k, kh, kw, stride, pad_t, pad_l, upsample, h_in, h_out, w_in, w_out, c1, nz, temb, res0, res1, out, a0, bias_mode,
                  a_scale, w_scale) -> bool:
    """The call is one tile 70 serves (mf_gemm_conv's own check, gemm_conv.cpp kPersTile) AND its grid gives every block at least
    PERS_MIN_NLOOP output tiles (the library's range choice: double the column ranges while there are fewer than 256 blocks)."""
    if not (kh == 1 and kw == 1 and stride == 1 and pad_t == 0 and pad_l == 0 and not upsample and h_in == h_out and w_in == w_out):
        return False
    if c1 or nz != 1 or temb is not None or res1 is not None or bias_mode or a_scale is not None or w_scale is not None:
        return False
    if m % 128 or n % 160 or k % 64 or k < 128 or out.dtype not in (torch.bfloat16, torch.float16) or a0.dtype != out.dtype:
        return False
    if res0 is not None and res0.dtype != out.dtype:
        return False
    tiles_m, tiles_n, ranges = m // 128, n // 160, 1
    while tiles_m * ranges < 256 and tiles_n % (ranges * 2) == 0:
        ranges *= 2
    return tiles_n // ranges >= PERS_MIN_NLOOP


def pers_linear(m: int, n: int, k: int, dtype: torch.dtype) -> bool:
    """Would a plain Linear of this shape (16-bit, no time embedding, at most one residual) go to the persistent tile 70?  models.py asks
    before it picks a form only that tile serves well (a LayerNorm folded into the GEGLU projection)."""
    if not PREFER_PERS or dtype not in (torch.bfloat16, torch.float16) or m % 128 or n % 160 or k % 64 or k < 128:
        return False
    tiles_m, tiles_n, ranges = m // 128, n // 160, 1
    while tiles_m * ranges < 256 and tiles_n % (ranges * 2) == 0:
        ranges *= 2
    return tiles_n // ranges >= PERS_MIN_NLOOP


# GroupNorm statistics from the producing GEMM's epilogue (mf_gemm_desc.gn_part -> mf_groupnorm_desc.part0 / part1).  A/B switch.
DEFER_REDUCE = os.environ.get("MFHIP_DEFER_REDUCE", "1") != "0"     # A/B switch: a resnet's conv1 leaves its split-K reduce to norm2
GN_FROM_PARTS = os.environ.get("MFHIP_GN_FROM_PARTS", "1") != "0"
GN_FROM_GROUPS = os.environ.get("MFHIP_GN_FROM_GROUPS", "1") != "0"     # ... per-group sums: no finalize launch either.  A/B switch.
RETUNE = os.environ.get("MFHIP_RETUNE", "0") == "1"      # developer switch: re-measure every shape once (new tiles were added)
TUNE_GRAPH = os.environ.get("MFHIP_TUNE_GRAPH", "0") == "1"   # developer switch: time candidates from a hipGraph (see _tuned_config)
TUNE_LOG: Optional[dict] = None     # developer hook (tools/tune_step.py): every candidate's time of every key tuned while it is a dict
KEY_LOG: Optional[list] = None      # developer hook: the tune key of every autotuned mf_gemm_conv call while it is a list
# Where in the denoise step a call sits ("b": BrushNet, on the side stream under the UNet's encoder; "e": UNet encoder + mid block;
# "d": UNet decoder, alone on the chip once BrushNet has finished).  The best tile for one shape differs between them — a tile that
# leaves room for the other stream's blocks wins under overlap, a whole-CU ring tile wins alone (tools/tune_step.py) — so the tune
# cache may hold "<key>@<ctx>" entries beside the plain one; a missing tagged entry falls back to the plain key.
TUNE_CTX: Optional[str] = None
_NO_TUNE_CTX = os.environ.get("MFHIP_NO_TUNE_CTX", "0") == "1"     # A/B switch: ignore the position-tagged entries
# The package ships a cache tuned on MI355X (read-only); new winners go to a per-user file (MFHIP_TUNE_CACHE, default
# ~/.cache/mfhip/tune_cache.json) that is overlaid on it.  Both carry the library's tile-table version: when tiles are
# renumbered (mf_gemm_tile_table_version changes) stale indices are dropped instead of being trusted.
_TUNE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_cache.json")
_tune: Optional[dict] = None
_tune_new: dict = {}
_tune_misses = 0


def tune_user_path() -> str:
    return os.environ.get("MFHIP_TUNE_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "mfhip", "tune_cache.json")


def _tune_read(path: str, version: int) -> dict:
    import json
    try:
        with open(path) as f:
            raw = json.load(f)
        if not isinstance(raw, dict) or raw.get("_meta", {}).get("tile_table") != version:
            return {}
        return {k: tuple(v) for k, v in raw.get("entries", {}).items()}
    except (OSError, ValueError, TypeError, AttributeError):
        return {}


def _tune_load() -> dict:
    global _tune
    if _tune is None:
        version = load().mf_gemm_tile_table_version()
        _tune = _tune_read(_TUNE_PATH, version)
        _tune.update(_tune_read(tune_user_path(), version))
    return _tune


def tune_save(path: Optional[str] = None) -> None:
    """Persist the winners found by this process (merged over what the user file already holds): written to a temporary
    file and renamed, so a concurrent reader never sees half a file."""
    if not _tune_new:
        return
    import json
    import tempfile
    path = path or tune_user_path()
    version = load().mf_gemm_tile_table_version()
    merged = _tune_read(path, version)
    merged.update(_tune_new)
    try:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=os.path.dirname(path) or ".", suffix=".tmp")
        with os.fdopen(fd, "w") as f:
            json.dump({"_meta": {"tile_table": version}, "entries": {k: list(v) for k, v in sorted(merged.items())}}, f, indent=0)
        os.replace(tmp, path)
    except OSError:
        pass


def tune_reload() -> None:
    """Drop the in-memory cache: the next lookup re-reads the shipped file and the per-user file (another rank of this node may
    just have written its winners there: distributed.tuned_once)."""
    global _tune
    _tune = None


def _tune_forget(ks: str) -> None:
    _tune_load().pop(ks, None)
    _tune_new.pop(ks, None)


def _tune_key(key: tuple) -> str:
    return ",".join(x if isinstance(x, str) else str(int(x)) for x in key)


def _tuned_config(d: "GemmDesc", key: tuple):
    cache = _tune_load()
    ks = _tune_key(key)
    hit = cache.get(ks)
    if hit is not None and not (RETUNE and ks not in _tune_new):
        return hit
    if torch.cuda.is_current_stream_capturing() or _RECORDER is not None:
        return (0, 0)           # (a recorded pass, like a capture, must not contain the tuner's trial launches: warm up first)
    lib = load()
    global _tune_misses
    _tune_misses += 1
    if _tune_misses == 1 and os.environ.get("RANK") is not None:
        # multi-rank cold start: every rank measures its own misses (the shipped tune_cache.json covers the benchmark's shapes,
        # so this only shows for new shapes); harmless for results — tiles differ in speed, not in arithmetic order per tile
        import sys
        print(f"[mfhip] rank {os.environ['RANK']}: GEMM shape {ks} is not in the tune cache; autotuning on this rank "
              f"(further misses are tuned silently; winners go to {tune_user_path()})", file=sys.stderr, flush=True)
    m, n, k = key[2], key[3], key[4]
    es = 2 if key[0] in (MF_BF16, MF_F16) else 1 if key[0] == MF_FP8 else 4
    nkt = (k * es + 127) // 128
    cands = []
    ntiles = lib.mf_gemm_num_tiles()
    bm, bn = C.c_int(), C.c_int()
    for t in range(1, ntiles + 1):
        lib.mf_gemm_tile_shape(t, C.byref(bm), C.byref(bn))
        blocks = -(-m // bm.value) * -(-n // bn.value) * key[9]
        sks = [1]
        if blocks < 512 and nkt >= 8 and not key[10]:
            sks += [s for s in (2, 3, 4, 6, 8, 12, 16, 24) if s <= nkt // 4 and blocks * s <= 2048
                    and s * key[9] * m * n <= d.ws_floats]
        cands += [(t, s, 0) for s in sks]
        # the in-launch combine (mode 1: the last-arriving K slice reduces; no second launch) makes a split of 2-4 cheap
        # enough for grids of up to 512 tiles and K of 4+ tiles
        if SK_FUSED and not key[10] and t in SK_FUSED_TILES.get(key[0], ()):
            cands += [(t, s, 1) for s in (2, 3, 4, 6, 8, 12, 16) if s <= max(nkt // 2, 1) and blocks <= SK_TICKETS and blocks * s <= 2048
                      and nkt >= 4 and s * key[9] * m * n <= d.ws_floats]
    best, best_t = (0, 0, 0), float("inf")
    st = _stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tk = sk_tickets(torch.device("cuda", torch.cuda.current_device()))
    for t, s, mode in cands:
        d.tile, d.splitk = t, s
        d.sk_tickets, d.sk_ticket_cap = (tk.data_ptr(), tk.numel()) if mode else (None, 0)
        if lib.mf_gemm_conv(C.byref(d), st) != 0:
            continue
        dt = float("inf")
        if TUNE_GRAPH:
            # developer re-tune: 8 launches replayed from a hipGraph, so that launches shorter than a ctypes call
            # (~16 us) are ranked by their kernel time, not by the host's launch rate (slow: a capture per candidate)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                cs = _stream()
                for _ in range(8):
                    lib.mf_gemm_conv(C.byref(d), cs)
            g.replay()
            for _trial in range(2):
                e0.record()
                g.replay()
                e1.record()
                e1.synchronize()
                dt = min(dt, e0.elapsed_time(e1))
            del g
        else:
            for _trial in range(2):      # best of two bursts: one burst alone mis-ranks candidates that are within a few %
                e0.record()
                for _ in range(4):
                    lib.mf_gemm_conv(C.byref(d), st)
                e1.record()
                e1.synchronize()
                dt = min(dt, e0.elapsed_time(e1))
        if TUNE_LOG is not None:
            TUNE_LOG.setdefault(ks, []).append((dt, t, s, mode))
        if dt < best_t:
            best, best_t = (t, s, mode), dt
    cache[ks] = best
    _tune_new[ks] = best
    return best


def gemm_conv(a0: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, dtype, w_split: int = 0,
              c0: int, lda0: int, batch: int, h_in: int, w_in: int, h_out: int, w_out: int,
              kh: int = 1, kw: int = 1, stride: int = 1, pad_t: int = 0, pad_l: int = 0, upsample: bool = False,
              a1: Optional[torch.Tensor] = None, c1: int = 0, lda1: int = 0,
              ldw: Optional[int] = None, n: int, ldc: Optional[int] = None,
              bias: Optional[torch.Tensor] = None, bias_mode: int = 0,
              temb: Optional[torch.Tensor] = None, ld_temb: int = 0,
              res0: Optional[torch.Tensor] = None, ld_res0: Optional[int] = None,
              res1: Optional[torch.Tensor] = None, ld_res1: Optional[int] = None, res1_rows: int = 0,
              alpha: float = 1.0, act: int = ACT_NONE,
              nz: int = 1, zdiv: int = 1, a_zs=(0, 0), w_zs=(0, 0), o_zs=(0, 0),
              a_scale: Optional[torch.Tensor] = None, w_scale: Optional[torch.Tensor] = None, a_scale_zs: int = 0,
              w_scale_zs: int = 0, splitk: int = 0, tile: int = 0, ln_colsum: Optional[torch.Tensor] = None, ln_eps: float = 1e-5,
              vt_out: Optional[torch.Tensor] = None, vt_n0: int = 0, vt_tokens: int = 0, sk_fused: bool = False,
              gn_part: Union[bool, int] = False, defer_reduce: bool = False) -> torch.Tensor:
    """Raw descriptor-level call of mf_gemm_conv (see include/mfhip.h). All strides in elements.  `dtype`: a torch
    dtype (bf16 / fp32 compute) or an MF_* compute code (the split codes take fp32 a0 and, with w_split=1, a weight
    from ops.split_pack).  `defer_reduce`: the caller hands `out` to groupnorm() next and to nothing else — a split-K launch may
    then leave its reduce to that GroupNorm (mf_gemm_desc.defer_reduce): `out` comes back UNWRITTEN with out._sk_pending set."""
    _req_cuda(a0, a1, w, out, bias, temb, res0, res1)
    d = GemmDesc()
    code = dtype if isinstance(dtype, int) else dt_code(dtype)
    d.dtype = code
    d.a0, d.a1 = _ptr(a0), _ptr(a1)
    d.c0, d.c1, d.lda0, d.lda1 = c0, c1, lda0, lda1
    d.a_dtype = dt_code(a0.dtype)
    d.batch, d.h_in, d.w_in, d.h_out, d.w_out = batch, h_in, w_in, h_out, w_out
    d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.upsample = kh, kw, stride, pad_t, pad_l, int(upsample)
    want_w = ({MF_F16X3: torch.float16, MF_BF16X3: torch.bfloat16}[code] if w_split else
              torch.bfloat16 if code == MF_BF16 else torch.float16 if code == MF_F16 else FP8 if code == MF_FP8 else torch.float32)
    if w.dtype != want_w:
        raise MfhipError(f"weight dtype {w.dtype} != {want_w} expected by compute code {code} (w_split={w_split})")
    d.w = _ptr(w)
    d.w_split = w_split
    d.ldw = ldw if ldw is not None else kh * kw * (c0 + c1)
    d.n = n
    d.nz, d.zdiv = nz, zdiv
    d.a_zs_o, d.a_zs_i = a_zs
    d.w_zs_o, d.w_zs_i = w_zs
    d.o_zs_o, d.o_zs_i = o_zs
    for t in (bias, temb):
        if t is not None and t.dtype != torch.float32:
            raise MfhipError("bias / temb must be fp32")
    d.bias, d.bias_mode = _ptr(bias), bias_mode
    d.temb, d.ld_temb = _ptr(temb), ld_temb
    for t in (a_scale, w_scale):
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
            raise MfhipError("a_scale / w_scale must be fp32 device tensors")
    d.a_scale, d.w_scale, d.a_scale_zs, d.w_scale_zs = _ptr(a_scale), _ptr(w_scale), a_scale_zs, w_scale_zs
    d.res0, d.res0_dtype = _ptr(res0), (dt_code(res0.dtype) if res0 is not None else 0)
    d.ld_res0 = ld_res0 if ld_res0 is not None else n
    d.res1, d.res1_dtype = _ptr(res1), (dt_code(res1.dtype) if res1 is not None else 0)
    d.ld_res1 = ld_res1 if ld_res1 is not None else n
    d.res1_rows = int(res1_rows)
    d.alpha, d.act = alpha, act
    d.out, d.out_dtype = _ptr(out), dt_code(out.dtype)
    d.ldc = ldc if ldc is not None else n
    ws = scratch("splitk", SPLITK_WS_FLOATS, out.device)
    d.splitk, d.ws, d.ws_floats = splitk, ws.data_ptr(), ws.numel()
    d.tile = tile
    fused = 0
    if ln_colsum is not None:
        _req_cuda(ln_colsum)
        if ln_colsum.dtype != torch.float32 or ln_colsum.numel() != n:
            raise MfhipError("ln_colsum must be n fp32 values")
        d.ln_colsum, d.ln_eps = _ptr(ln_colsum), ln_eps
        fused |= 1
    if vt_out is not None:
        _req_cuda(vt_out)
        if vt_out.dtype not in (torch.bfloat16, torch.float16) or vt_out.dim() != 3 or not vt_out.is_contiguous():
            raise MfhipError("vt_out must be a contiguous bf16 / fp16 [images][n - vt_n0][ld] tensor")
        d.vt_out, d.vt_n0, d.vt_tokens, d.vt_ld = _ptr(vt_out), vt_n0, vt_tokens, vt_out.shape[-1]
        fused |= 2
    tkey = None
    use_pers = (PREFER_PERS and tile == 0 and splitk in (0, 1) and code in (MF_BF16, MF_F16) and not gn_part
                and (vt_out is None or (vt_n0 % 160 == 0 and vt_tokens % 8 == 0 and res0 is None and act == ACT_NONE))
                and _pers_applies(batch * h_out * w_out, n, kh * kw * (c0 + c1), kh, kw, stride, pad_t, pad_l, upsample, h_in, h_out, w_in, w_out, c1, nz,
                                  temb, res0, res1, out, a0, bias_mode, a_scale, w_scale))
    if use_pers:
        # the persistent 128-row GEMM (tile 70) where a block walks two or more output tiles: the epilogue of every tile but the
        # last runs under the next main loop (csrc/gemm_pers.hip; tools/bench_ff1.py for the per-shape table).  Not a tuner candidate
        # for these calls: it wins them by 5-35 % in isolation and the step agrees (same-box A/B, profiles/r06_ab_switches.txt)
        d.tile, d.splitk = PERS_TILE, 1
    elif tile == 0 and splitk in (0, 1) and AUTOTUNE:
        tkey = (code, d.a_dtype, batch * h_out * w_out, n, kh * kw * (c0 + c1), kh, stride, int(upsample), int(c1 > 0),
                nz, int(splitk == 1) if not fused else 1, act, h_out, w_out) + ((w_split,) if code in (MF_F16X3, MF_BF16X3) else ()) \
            + ((("ln", "vt", "lnvt")[fused - 1],) if fused else ())
        cfg = None
        if TUNE_CTX is not None and not _NO_TUNE_CTX:
            ks_ctx = _tune_key(tkey) + "@" + TUNE_CTX
            cfg = _tune_load().get(ks_ctx)
            if KEY_LOG is not None:
                KEY_LOG.append(ks_ctx)
        elif KEY_LOG is not None:
            KEY_LOG.append(_tune_key(tkey))
        if cfg is None:
            cfg = _tuned_config(d, tkey)
        d.tile, d.splitk = cfg[0], cfg[1]
        if len(cfg) > 2 and cfg[2]:
            tk = sk_tickets(out.device)
            d.sk_tickets, d.sk_ticket_cap = tk.data_ptr(), tk.numel()
        else:
            d.sk_tickets, d.sk_ticket_cap = None, 0
    elif sk_fused and splitk > 1:
        tk = sk_tickets(out.device)
        d.sk_tickets, d.sk_ticket_cap = tk.data_ptr(), tk.numel()
    part = part_rows = None
    if gn_part:
        # GroupNorm statistics from this launch (mf_gemm_desc.gn_part): per-channel partial sums of the output, attached to `out`
        # as out._gn_part = (fp32 buffer, rows per block) for groupnorm() to pick up.  Set after the tuner ran (it times plain launches).
        m_rows = batch * h_out * w_out
        part = torch.empty(2 * n * (m_rows // 32), dtype=torch.float32, device=out.device)
        part_rows, grouped = C.c_int32(0), C.c_int32(0)
        d.gn_part, d.gn_part_floats, d.gn_part_rows = part.data_ptr(), part.numel(), C.addressof(part_rows)
        # gn_part = the consumer GroupNorm's group count (an int > 1): per-group sums too where the tile allows (no finalize launch then)
        d.gn_groups, d.gn_grouped = (int(gn_part) if (gn_part is not True and int(gn_part) > 1) else 0), C.addressof(grouped)
    deferred = C.c_int32(0)
    if defer_reduce and DEFER_REDUCE and not gn_part:
        d.defer_reduce, d.deferred_splits = 1, C.addressof(deferred)

    def pending():
        if deferred.value > 0:      # (the slabs stay in this stream's split-K scratch until the GroupNorm that follows has read them)
            out._sk_pending = (ws, int(deferred.value), bias, temb, ld_temb, float(alpha))
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(load().mf_gemm_conv(C.byref(d), _stream()), "mf_gemm_conv")
        e1.record()
        pending()
        PROFILE.append((e0, e1, 2.0 * batch * h_out * w_out * n * kh * kw * (c0 + c1) * nz,
                        (batch * h_out * w_out, n, kh * kw * (c0 + c1), kh, stride, int(upsample), nz, d.tile, d.splitk, int(code))))
        if part is not None:
            out._gn_part = (part, int(part_rows.value), int(d.gn_groups) if grouped.value else 0)
        return out
    rc = load().mf_gemm_conv(C.byref(d), _stream())
    if rc != 0 and tkey is not None and d.tile != 0:
        # a cached (tile, split-K) the library no longer accepts for this call: forget it and let the heuristic choose
        _tune_forget(_tune_key(tkey))
        d.tile, d.splitk, d.sk_tickets, d.sk_ticket_cap = 0, splitk, None, 0
        rc = load().mf_gemm_conv(C.byref(d), _stream())
    _check(rc, "mf_gemm_conv")
    pending()
    if part is not None:
        # (buffer, rows per block, G): G > 0 = per-group sums of G groups follow the per-channel ones at float 2 * n * (M / rows)
        out._gn_part = (part, int(part_rows.value), int(d.gn_groups) if grouped.value else 0)
    return out


def groupnorm(x0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, groups: int, eps: float, silu: bool,
              out_dtype: torch.dtype, x1: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              stats_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x0/x1: NHWC [B, H, W, C] (or [B, HW, C]); returns the normalised tensor over cat([x0, x1], -1).  stats_out (fp32
    [B, groups, 2], written): every group's (mean, rstd), which groupnorm_bwd takes as `stats`."""
    pend = getattr(x0, "_sk_pending", None)
    if pend is not None:
        x0._sk_pending = None              # consumed here (x0 itself stays unwritten: nothing else may read it)
    _req_cuda(x0, x1, gamma, beta)
    b = x0.shape[0]
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    hw = x0.numel() // (b * c0)
    if not x0.is_contiguous() or (x1 is not None and not x1.is_contiguous()):
        raise MfhipError("groupnorm inputs must be contiguous NHWC")
    if out is None:
        out = torch.empty(*x0.shape[:-1], c0 + c1, dtype=out_dtype, device=x0.device)
    d = GroupNormDesc()
    d.x0, d.x1, d.c0, d.c1 = _ptr(x0), _ptr(x1), c0, c1
    d.in_dtype = dt_code(x0.dtype)
    d.batch, d.hw, d.groups, d.eps = b, hw, groups, eps
    d.gamma, d.beta, d.silu = _ptr(gamma), _ptr(beta), int(silu)
    d.out, d.out_dtype = _ptr(out), dt_code(out.dtype)
    lib = load()
    ws = scratch("gn", int(lib.mf_groupnorm_ws_floats(b, groups, c0 + c1)), x0.device)
    d.ws = ws.data_ptr()
    if stats_out is not None:
        _f32(stats_out)
        if stats_out.numel() != b * groups * 2 or not stats_out.is_contiguous():
            raise MfhipError("groupnorm: stats_out is a contiguous [batch, groups, 2] tensor")
        d.stats_out = stats_out.data_ptr()
    if pend is not None:
        if x1 is not None or stats_out is not None:
            raise MfhipError("groupnorm: a deferred split-K input is a single segment without stats_out")
        ws_t, splits, sk_bias, sk_temb, sk_ld, sk_alpha = pend
        d.sk_ws, d.sk_splits, d.sk_bias, d.sk_temb, d.sk_ld_temb, d.sk_alpha = ws_t.data_ptr(), splits, _ptr(sk_bias), _ptr(sk_temb), sk_ld, sk_alpha
    if GN_FROM_PARTS and hw > 256:
        # statistics handed over by the producing GEMMs (gemm_conv(..., gn_part=True) attached them to its output tensor)
        p0, p1 = getattr(x0, "_gn_part", None), (getattr(x1, "_gn_part", None) if x1 is not None else None)
        if p0 is not None and (x1 is None or p1 is not None):
            d.part0, d.part0_rows = p0[0].data_ptr(), p0[1]
            if p1 is not None:
                d.part1, d.part1_rows = p1[0].data_ptr(), p1[1]
            if x1 is None and len(p0) > 2 and p0[2] == groups and GN_FROM_GROUPS:
                d.grp0, d.grp0_rows = p0[0].data_ptr() + 4 * 2 * c0 * (b * hw // p0[1]), p0[1]
    _check(lib.mf_groupnorm(C.byref(d), _stream()), "mf_groupnorm")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out_dtype: torch.dtype
              ) -> torch.Tensor:
    _req_cuda(x, gamma, beta)
    c = x.shape[-1]
    rows = x.numel() // c
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _check(load().mf_layernorm(C.c_void_p(x.data_ptr()), dt_code(x.dtype), C.c_void_p(out.data_ptr()),
                               dt_code(out_dtype), C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()),
                               C.c_int64(rows), c, C.c_float(eps), _stream()), "mf_layernorm")
    return out


def softmax_rows(scores: torch.Tensor, cols: int, out_dtype: torch.dtype) -> torch.Tensor:
    """scores: [..., ld] fp32; softmax over the first `cols` entries of each row, pad written as 0."""
    _req_cuda(scores)
    ld = scores.shape[-1]
    rows = scores.numel() // ld
    out = torch.empty(scores.shape, dtype=out_dtype, device=scores.device)
    _check(load().mf_softmax_rows(C.c_void_p(scores.data_ptr()), C.c_void_p(out.data_ptr()), dt_code(out_dtype),
                                  C.c_int64(rows), cols, ld, _stream()), "mf_softmax_rows")
    return out


def attention_bf16(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, *, ldq: int, ldk: int,
                   ldvt: int, ldo: int, batch: int, heads: int, sq: int, skv: int, head_dim: int, scale: float,
                   lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`lse` (fp32 [batch, heads, sq], written): the row statistic of the flash backward (mf_attention_bf16_lse)."""
    _req_cuda(q, k, vt, out, lse)
    if PROFILE is not None:               # bench.py's FLOP census (no timing here: the GEMM family is the timed one)
        global PROFILE_ATTN_FLOPS
        PROFILE_ATTN_FLOPS += 4.0 * batch * heads * sq * skv * head_dim
    if q.dtype == torch.float16:          # the fp16 storage mode: the same kernel on the f16 MFMA forms (inference: no row statistics)
        if lse is not None or not (k.dtype == vt.dtype == out.dtype == torch.float16):
            raise MfhipError("attention_bf16: fp16 operands take fp16 k / vt / out and no lse")
        _check(load().mf_attention_f16(C.c_void_p(q.data_ptr()), C.c_int64(ldq), C.c_void_p(k.data_ptr()), C.c_int64(ldk),
                                       C.c_void_p(vt.data_ptr()), C.c_int64(ldvt), C.c_void_p(out.data_ptr()),
                                       C.c_int64(ldo), batch, heads, sq, skv, head_dim, C.c_float(scale), _stream()),
               "mf_attention_f16")
        return out
    if lse is None:
        _check(load().mf_attention_bf16(C.c_void_p(q.data_ptr()), C.c_int64(ldq), C.c_void_p(k.data_ptr()), C.c_int64(ldk),
                                        C.c_void_p(vt.data_ptr()), C.c_int64(ldvt), C.c_void_p(out.data_ptr()),
                                        C.c_int64(ldo), batch, heads, sq, skv, head_dim, C.c_float(scale), _stream()),
               "mf_attention_bf16")
        return out
    _f32(lse)
    if lse.numel() != batch * heads * sq or not lse.is_contiguous():
        raise MfhipError("attention_bf16: lse is a contiguous [batch, heads, sq] tensor")
    _check(load().mf_attention_bf16_lse(C.c_void_p(q.data_ptr()), C.c_int64(ldq), C.c_void_p(k.data_ptr()), C.c_int64(ldk),
                                        C.c_void_p(vt.data_ptr()), C.c_int64(ldvt), C.c_void_p(out.data_ptr()), C.c_int64(ldo),
                                        C.c_void_p(lse.data_ptr()), batch, heads, sq, skv, head_dim, C.c_float(scale), _stream()),
           "mf_attention_bf16_lse")
    return out


def quantize_rows_fp8(x: torch.Tensor, norm=None, eps: float = 1e-5):
    """Per-row dynamic fp8 quantisation of [..., C] (optionally LayerNorm(x; gamma, beta) first): returns
    (q fp8 [..., C], scale fp32 [rows]) with x ~ q * scale[row]."""
    _req_cuda(x)
    if not x.is_contiguous():
        raise MfhipError("quantize_rows_fp8: contiguous input")
    c = x.shape[-1]
    rows = x.numel() // c
    q = torch.empty(x.shape, dtype=FP8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    g, b = norm if norm is not None else (None, None)
    _check(load().mf_quantize_rows_fp8(C.c_void_p(x.data_ptr()), dt_code(x.dtype), C.c_void_p(q.data_ptr()), C.c_void_p(sc.data_ptr()),
                                       C.c_int64(rows), c, C.c_void_p(_ptr(g)), C.c_void_p(_ptr(b)), C.c_float(eps), _stream()),
           "mf_quantize_rows_fp8")
    return q, sc


def split_overflow(reset: bool = True) -> int:
    """Bit mask of the fp16 split precision's range-guard flags (0 = every f16x3 operand since the last reset was inside
    |x| <= 65504; bit 0 GEMM / conv, bit 1 attention operands, bit 2 weight gradients).  Synchronises the current stream."""
    raised = C.c_int32(0)
    _check(load().mf_split_overflow(int(reset), C.byref(raised), _stream()), "mf_split_overflow")
    return int(raised.value)


class SplitRangeError(MfhipError):
    """An f16x3 operand left the fp16 range: the products computed from it are wrong (saturated, not inf)."""


def split_pack_check(w: torch.Tensor, code: int) -> None:
    """Weights are split once on the host: refuse an fp16 split of a weight outside the fp16 range."""
    if code == MF_F16X3 and w.numel() and float(w.abs().max()) > 65504.0:
        raise SplitRangeError("f16x3: a weight exceeds the fp16 range (|w| > 65504); use precision 'bf16x3' or 'fp32'")


def split_pack(w: torch.Tensor, code: int, out: Optional[torch.Tensor] = None):
    """fp32 [rows, k] (row stride w.stride(0)) -> (16-bit [rows, 2 * kp] in mf_gemm_desc's w_split layout, kp) on the device
    (mf_split_pack; ops.split_pack is the host-side twin the inference weights go through once)."""
    _req_cuda(w, out)
    if w.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1:
        raise MfhipError("split_pack: fp32 [rows, k] with unit column stride")
    rows, k = w.shape
    kp = (k + 31) // 32 * 32
    half = torch.float16 if code == MF_F16X3 else torch.bfloat16
    if out is None:
        out = torch.empty(rows, 2 * kp, dtype=half, device=w.device)
    elif out.dtype != half or out.numel() != rows * 2 * kp or not out.is_contiguous():
        raise MfhipError("split_pack: out must be a contiguous 16-bit [rows, 2 * kp] tensor")
    _check(load().mf_split_pack(C.c_void_p(w.data_ptr()), C.c_int64(w.stride(0)), C.c_void_p(out.data_ptr()), C.c_int64(rows), k, code,
                                _stream()), "mf_split_pack")
    return out, kp


def split_halves(x: torch.Tensor):
    """fp32 tensor -> (hi, lo) fp16 tensors of the same shape with hi + lo = x to 22 bits."""
    _req_cuda(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise MfhipError("split_halves: contiguous fp32 input")
    hi = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _check(load().mf_split_halves(C.c_void_p(x.data_ptr()), C.c_void_p(hi.data_ptr()), C.c_void_p(lo.data_ptr()),
                                  C.c_int64(x.numel()), _stream()), "mf_split_halves")
    return hi, lo


def attention_f16x3(q, k, vt, out: torch.Tensor, *, ldq: int, ldk: int, ldvt: int, ldo: int, batch: int, heads: int,
                    sq: int, skv: int, head_dim: int, scale: float, lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q / k / vt: (hi, lo) pairs from split_halves; out fp32.  lse (optional, fp32 [batch, heads, sq]): the row statistics the
    flash backward needs."""
    _req_cuda(*q, *k, *vt, out, lse)
    _check(load().mf_attention_f16x3_lse(C.c_void_p(q[0].data_ptr()), C.c_void_p(q[1].data_ptr()), C.c_int64(ldq),
                                         C.c_void_p(k[0].data_ptr()), C.c_void_p(k[1].data_ptr()), C.c_int64(ldk),
                                         C.c_void_p(vt[0].data_ptr()), C.c_void_p(vt[1].data_ptr()), C.c_int64(ldvt),
                                         C.c_void_p(out.data_ptr()), C.c_int64(ldo), C.c_void_p(_ptr(lse)), batch, heads, sq, skv, head_dim,
                                         C.c_float(scale), _stream()), "mf_attention_f16x3")
    return out


def rowdot_heads(a: torch.Tensor, b: torch.Tensor, heads: int) -> torch.Tensor:
    """[B, S, C] x [B, S, C] -> [B, heads, S]: per-head dot products of the rows (the D term of the attention backward).
    `b` may be bf16 (the bf16 forward's output)."""
    _f32(a)
    bsz, s, c = a.shape
    out = torch.empty(bsz, heads, s, dtype=torch.float32, device=a.device)
    if b.dtype == torch.bfloat16:
        _req_cuda(b)
        if b.shape != a.shape or not (a.is_contiguous() and b.is_contiguous()):
            raise MfhipError("rowdot_heads: contiguous operands of one shape")
        _check(load().mf_rowdot_heads_bf16(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), bsz, s, heads,
                                           c // heads, C.c_int64(c), _stream()), "mf_rowdot_heads_bf16")
        return out
    _f32(b)
    _check(load().mf_rowdot_heads(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), bsz, s, heads, c // heads,
                                  C.c_int64(c), _stream()), "mf_rowdot_heads")
    return out


def rowdot_heads_cast(a: torch.Tensor, b16: torch.Tensor, heads: int):
    """(D [B, heads, S] = per-head row dots of a (fp32) with b16 (bf16), a16 = bf16(a)) from one read of a (mf_rowdot_heads_cast)."""
    _f32(a)
    _req_cuda(b16)
    bsz, s, c = a.shape
    if b16.dtype != torch.bfloat16 or b16.shape != a.shape or not (a.is_contiguous() and b16.is_contiguous()):
        raise MfhipError("rowdot_heads_cast: contiguous fp32 / bf16 operands of one shape")
    out = torch.empty(bsz, heads, s, dtype=torch.float32, device=a.device)
    a16 = torch.empty_like(b16)
    _check(load().mf_rowdot_heads_cast(C.c_void_p(a.data_ptr()), C.c_void_p(b16.data_ptr()), C.c_void_p(a16.data_ptr()), C.c_void_p(out.data_ptr()),
                                       bsz, s, heads, c // heads, _stream()), "mf_rowdot_heads_cast")
    return out, a16


def attention_bwd_f16x3(q, k, v, do, qt, kt, dot, lse: torch.Tensor, dd: torch.Tensor, dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, *,
                        heads: int, scale: float, _entry: str = "mf_attention_bwd_f16x3") -> None:
    """Flash attention backward (mf_attention_bwd_f16x3).  q / k / v / do: (hi, lo) planes [B, S, C]; qt / kt / dot: (hi, lo) planes of
    the transposed tensors [B, C, ld]; lse / dd fp32 [B, heads, Sq]; dq / dk / dv fp32 [B, S, C] (written)."""
    b, sq, c = q[0].shape
    skv = k[0].shape[1]
    d = AttnBwdDesc()
    d.q_hi, d.q_lo, d.ldq = q[0].data_ptr(), q[1].data_ptr(), c
    d.k_hi, d.k_lo, d.ldk = k[0].data_ptr(), k[1].data_ptr(), c
    d.v_hi, d.v_lo, d.ldv = v[0].data_ptr(), v[1].data_ptr(), c
    d.do_hi, d.do_lo, d.lddo = do[0].data_ptr(), do[1].data_ptr(), c
    d.qt_hi, d.qt_lo, d.ldqt = qt[0].data_ptr(), qt[1].data_ptr(), qt[0].shape[-1]
    d.kt_hi, d.kt_lo, d.ldkt = kt[0].data_ptr(), kt[1].data_ptr(), kt[0].shape[-1]
    d.dot_hi, d.dot_lo, d.lddot = dot[0].data_ptr(), dot[1].data_ptr(), dot[0].shape[-1]
    d.lse, d.dd = lse.data_ptr(), dd.data_ptr()
    d.dq, d.dk, d.dv, d.ldo = dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), c
    d.batch, d.heads, d.sq, d.skv, d.head_dim, d.scale = b, heads, sq, skv, c // heads, scale
    d.out_dtype = dt_code(dq.dtype)
    _check(getattr(load(), _entry)(C.byref(d), _stream()), _entry)


def attention_bwd_bf16(q, k, v, do, qt, kt, dot, lse: torch.Tensor, dd: torch.Tensor, dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, *,
                       heads: int, scale: float) -> None:
    """mf_attention_bwd_bf16: the flash backward on single bf16 planes.  q / k / v / do bf16 [B, S, C]; qt / kt / dot bf16 [B, C, ld]."""
    for t in (q, k, v, do, qt, kt, dot):
        if t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise MfhipError("attention_bwd_bf16: contiguous bf16 operands")
    _f32(lse, dd)
    if not (dq.dtype == dk.dtype == dv.dtype and dq.dtype in (torch.float32, torch.bfloat16)):
        raise MfhipError("attention_bwd_bf16: dq / dk / dv are all fp32 or all bf16")
    attention_bwd_f16x3((q, q), (k, k), (v, v), (do, do), (qt, qt), (kt, kt), (dot, dot), lse, dd, dq, dk, dv, heads=heads, scale=scale,
                        _entry="mf_attention_bwd_bf16")


def pack_nhwc(src0: torch.Tensor, src1: Optional[torch.Tensor], c_pad: int, out_dtype: torch.dtype) -> torch.Tensor:
    """NCHW fp32 (optionally two tensors concatenated on C) -> NHWC out_dtype with channels zero-padded."""
    _req_cuda(src0, src1)
    b, c0, h, w = src0.shape
    c1 = src1.shape[1] if src1 is not None else 0
    src0 = src0.contiguous().float()
    if src1 is not None:
        src1 = src1.contiguous().float()
    out = torch.empty(b, h, w, c_pad, dtype=out_dtype, device=src0.device)
    _check(load().mf_pack_nhwc(C.c_void_p(src0.data_ptr()), c0, C.c_void_p(_ptr(src1)), c1,
                               C.c_void_p(out.data_ptr()), dt_code(out_dtype), c_pad, b, h * w, _stream()),
           "mf_pack_nhwc")
    return out


def unpack_nchw(src: torch.Tensor, c: int) -> torch.Tensor:
    """NHWC [B, H, W, ld] -> NCHW fp32 [B, c, H, W] (first c channels)."""
    _req_cuda(src)
    b, h, w, ld = src.shape
    out = torch.empty(b, c, h, w, dtype=torch.float32, device=src.device)
    _check(load().mf_unpack_nchw(C.c_void_p(src.data_ptr()), dt_code(src.dtype), C.c_int64(ld),
                                 C.c_void_p(out.data_ptr()), c, b, h * w, _stream()), "mf_unpack_nchw")
    return out


def add(a: torch.Tensor, b: torch.Tensor, out_dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req_cuda(a, b, out)
    if a.shape != b.shape:
        raise MfhipError(f"mf_add shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    if out is None:
        out = torch.empty(a.shape, dtype=out_dtype, device=a.device)
    elif out.shape != a.shape or out.dtype != out_dtype or not out.is_contiguous():
        raise MfhipError("mf_add: `out` must be a contiguous tensor of the operands' shape and the output dtype")
    _check(load().mf_add(C.c_void_p(a.data_ptr()), dt_code(a.dtype), C.c_void_p(b.data_ptr()), dt_code(b.dtype),
                         C.c_void_p(out.data_ptr()), dt_code(out_dtype), C.c_int64(a.numel()), _stream()), "mf_add")
    return out


def cast_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 -> bf16 (nearest-even) copy of a contiguous tensor (mf_cast_bf16)."""
    _req_cuda(x, out)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise MfhipError("cast_bf16: contiguous fp32 input")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    elif out.dtype != torch.bfloat16 or out.numel() != x.numel() or not out.is_contiguous():
        raise MfhipError("cast_bf16: `out` must be a contiguous bf16 tensor of the same size")
    _check(load().mf_cast_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int64(x.numel()), _stream()), "mf_cast_bf16")
    return out


def cast_bf16_colsum(x: torch.Tensor, n: int, *, segs: int = 1, seg_out: Optional[torch.Tensor] = None, ldo: Optional[int] = None,
                     tot_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 copy of the contiguous fp32 [rows, n] gradient `x` plus, from the same read, its column sums ADDED into
    seg_out[s][:n] (row stride ldo; one row per segment of rows / segs rows) and / or tot_out[:n] (mf_cast_bf16_colsum)."""
    _f32(x, seg_out, tot_out)
    if not x.is_contiguous() or x.numel() % n or (x.numel() // n) % segs:
        raise MfhipError("cast_bf16_colsum: contiguous [segs * rows_per_seg, n] input")
    rps = x.numel() // n // segs
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lib = load()
    ws = scratch("cast_colsum", int(lib.mf_cast_bf16_colsum_ws_floats(segs, C.c_int64(rps), n)), x.device)
    _check(lib.mf_cast_bf16_colsum(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), segs, C.c_int64(rps), n, C.c_void_p(_ptr(seg_out)),
                                   C.c_int64(n if ldo is None else ldo), 1, C.c_void_p(_ptr(tot_out)), 1, C.c_void_p(ws.data_ptr()), _stream()),
           "mf_cast_bf16_colsum")
    return out


def geglu(h: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    _req_cuda(h)
    c = h.shape[-1] // 2
    rows = h.numel() // (2 * c)
    out = torch.empty(*h.shape[:-1], c, dtype=out_dtype, device=h.device)
    _check(load().mf_geglu(C.c_void_p(h.data_ptr()), dt_code(h.dtype), C.c_void_p(out.data_ptr()), dt_code(out_dtype),
                           C.c_int64(rows), c, _stream()), "mf_geglu")
    return out


def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos: bool, freq_shift: float) -> torch.Tensor:
    _req_cuda(t)
    t = t.float().contiguous()
    out = torch.empty(t.numel(), dim, dtype=torch.float32, device=t.device)
    _check(load().mf_timestep_embedding(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), t.numel(), dim,
                                        int(flip_sin_to_cos), C.c_float(freq_shift), _stream()),
           "mf_timestep_embedding")
    return out


def silu_f32(x: torch.Tensor) -> torch.Tensor:
    _req_cuda(x)
    out = torch.empty_like(x)
    _check(load().mf_silu_f32(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int64(x.numel()), _stream()),
           "mf_silu_f32")
    return out


def cfg_ddim_step(eps_u: torch.Tensor, eps_c: Optional[torch.Tensor], g: float, x: torch.Tensor, sqrt_at: float,
                  sqrt_1m_at: float, sqrt_ap: float, dir_coef: float, eps_out: Optional[torch.Tensor] = None,
                  pred_type: int = 0, clip: float = 0.0) -> torch.Tensor:
    _req_cuda(eps_u, eps_c, x, eps_out)
    xp = torch.empty_like(x)
    _check(load().mf_cfg_ddim_step(C.c_void_p(eps_u.data_ptr()), C.c_void_p(_ptr(eps_c)), C.c_float(g),
                                   C.c_void_p(x.data_ptr()), C.c_void_p(xp.data_ptr()), C.c_float(sqrt_at),
                                   C.c_float(sqrt_1m_at), C.c_float(sqrt_ap), C.c_float(dir_coef), pred_type,
                                   C.c_float(clip), C.c_void_p(_ptr(eps_out)), C.c_int64(x.numel()), _stream()),
           "mf_cfg_ddim_step")
    return xp


def cfg_ddim_step_dev(eps_u: torch.Tensor, eps_c: Optional[torch.Tensor], g: float, x: torch.Tensor, coef4: torch.Tensor,
                      pred_type: int = 0, clip: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DDIM update with device-resident coefficients (graph replay); `out` may be `x` itself (in place)."""
    _req_cuda(eps_u, eps_c, x, coef4, out)
    xp = out if out is not None else torch.empty_like(x)
    _check(load().mf_cfg_ddim_step_dev(C.c_void_p(eps_u.data_ptr()), C.c_void_p(_ptr(eps_c)), C.c_float(g),
                                       C.c_void_p(x.data_ptr()), C.c_void_p(xp.data_ptr()),
                                       C.c_void_p(coef4.data_ptr()), pred_type, C.c_float(clip),
                                       C.c_int64(x.numel()), _stream()), "mf_cfg_ddim_step_dev")
    return xp


def cfg_combine(eps_u: torch.Tensor, eps_c: torch.Tensor, g: float) -> torch.Tensor:
    _req_cuda(eps_u, eps_c)
    out = torch.empty_like(eps_u)
    _check(load().mf_cfg_combine(C.c_void_p(eps_u.data_ptr()), C.c_void_p(eps_c.data_ptr()), C.c_float(g),
                                 C.c_void_p(out.data_ptr()), C.c_int64(eps_u.numel()), _stream()), "mf_cfg_combine")
    return out


def axpby_n(xs, coefs, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = sum_i coefs[i] * xs[i] over fp32 tensors of equal shape (at most 6); `out` may be one of the inputs."""
    _req_cuda(*xs)
    n = len(xs)
    arr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    cf = (C.c_float * n)(*[float(c) for c in coefs])
    if out is None:
        out = torch.empty_like(xs[0])
    _check(load().mf_axpby_n(arr, cf, n, C.c_void_p(out.data_ptr()), C.c_int64(out.numel()), _stream()), "mf_axpby_n")
    return out


def mse_loss(pred: torch.Tensor, target: torch.Tensor, weights: Optional[torch.Tensor] = None):
    """(loss[1], per_sample[B]) with per_sample[b] = mean((pred[b] - target[b])**2) * weights[b]; fp32 tensors."""
    _req_cuda(pred, target)
    if pred.shape != target.shape or pred.dtype != torch.float32 or target.dtype != torch.float32:
        raise ValueError("mse_loss: pred and target must be fp32 tensors of the same shape")
    pred, target = pred.contiguous(), target.contiguous()
    rows = pred.shape[0]
    per = torch.empty(rows, dtype=torch.float32, device=pred.device)
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    w = None
    if weights is not None:
        w = weights.to(pred.device, torch.float32).contiguous()
        if w.numel() != rows:
            raise ValueError("mse_loss: one weight per sample")
    _check(load().mf_mse_loss(C.c_void_p(pred.data_ptr()), C.c_void_p(target.data_ptr()),
                              C.c_void_p(w.data_ptr() if w is not None else None), C.c_void_p(per.data_ptr()),
                              C.c_void_p(loss.data_ptr()), C.c_int32(rows), C.c_int64(pred.numel() // rows), _stream()),
           "mf_mse_loss")
    return loss, per


def vae_sample(moments: torch.Tensor, noise: torch.Tensor, c: int, scaling: float) -> torch.Tensor:
    """moments NHWC [B, H, W, ld >= 2c]; noise NCHW fp32 [B, c, H, W] -> z NCHW fp32."""
    _req_cuda(moments, noise)
    b, h, w, ld = moments.shape
    z = torch.empty(b, c, h, w, dtype=torch.float32, device=moments.device)
    noise = noise.float().contiguous()
    _check(load().mf_vae_sample(C.c_void_p(moments.data_ptr()), dt_code(moments.dtype), C.c_int64(ld),
                                C.c_void_p(noise.data_ptr()), C.c_void_p(z.data_ptr()), c, b, h * w,
                                C.c_float(scaling), _stream()), "mf_vae_sample")
    return z


def nearest_resize(src: torch.Tensor, h_out: int, w_out: int) -> torch.Tensor:
    """F.interpolate(src, size=(h_out, w_out)) (default mode='nearest') for NCHW fp32."""
    _req_cuda(src)
    b, c, h, w = src.shape
    src = src.float().contiguous()
    out = torch.empty(b, c, h_out, w_out, dtype=torch.float32, device=src.device)
    _check(load().mf_nearest_resize(C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()), b * c, h, w, h_out, w_out,
                                    _stream()), "mf_nearest_resize")
    return out


# ---- training (csrc/train.hip): fp32 tensors ---------------------------------------------------------------------
def _f32(*ts):
    for t in ts:
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
            raise MfhipError("training kernels take fp32 device tensors")


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, *, code: int, c0: int, batch: int, h_in: int, w_in: int,
               h_out: int, w_out: int, kh: int = 1, kw: int = 1, stride: int = 1, pad_t: int = 0, pad_l: int = 0,
               upsample: bool = False, x1: Optional[torch.Tensor] = None, c1: int = 0, n: int, accumulate: bool = True) -> None:
    """dw[n][kh*kw*(c0+c1)] (+)= sum_m dy[m][n] * im2col(x | x1)[m][:] (mf_conv_wgrad).  code MF_BF16: x / x1 / dy are bf16 tensors."""
    if code == MF_BF16:
        _req_cuda(x, x1, dy, dw)
        if any(t is not None and t.dtype != torch.bfloat16 for t in (x, x1, dy)) or dw.dtype != torch.float32:
            raise MfhipError("conv_wgrad(MF_BF16): bf16 x / dy, fp32 dw")
    else:
        _f32(x, x1, dy, dw)
    d = WgradDesc()
    d.dtype = code if code in (MF_F16X3, MF_BF16X1, MF_BF16) else MF_F32
    d.a0, d.a1, d.c0, d.c1, d.lda0, d.lda1 = _ptr(x), _ptr(x1), c0, c1, c0, c1
    d.batch, d.h_in, d.w_in, d.h_out, d.w_out = batch, h_in, w_in, h_out, w_out
    d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.upsample = kh, kw, stride, pad_t, pad_l, int(upsample)
    d.dy, d.lddy, d.n = _ptr(dy), n, n
    d.dw, d.lddw = _ptr(dw), kh * kw * (c0 + c1)
    d.accumulate, d.splitm = int(accumulate), 0
    ws = scratch("wgrad", WGRAD_WS_FLOATS, x.device)
    d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    _check(load().mf_conv_wgrad(C.byref(d), _stream()), "mf_conv_wgrad")


def zero_ranges(base: torch.Tensor, offs: torch.Tensor, lens: torch.Tensor) -> None:
    """base[offs[i] : offs[i] + lens[i]] = 0 in one launch (mf_zero_ranges; offs / lens: int64 device tensors)."""
    _f32(base)
    _req_cuda(offs, lens)
    if offs.dtype != torch.int64 or lens.dtype != torch.int64 or offs.numel() != lens.numel() or not (offs.is_contiguous() and lens.is_contiguous()):
        raise MfhipError("zero_ranges: two contiguous int64 device tensors of one length")
    _check(load().mf_zero_ranges(C.c_void_p(base.data_ptr()), C.c_void_p(offs.data_ptr()), C.c_void_p(lens.data_ptr()), offs.numel(), _stream()),
           "mf_zero_ranges")


def set_wgrad_dma(on: bool) -> None:
    """Developer / test switch (mf_debug_set_wgrad_dma): bf16-input weight gradients on the LDS-DMA kernel (default) or the
    register-staged one; bit-identical."""
    load().mf_debug_set_wgrad_dma(int(bool(on)))


WGRAD_WS_FLOATS = 64 * 1024 * 1024      # 256 MiB of split-M slabs


def transpose(x: torch.Tensor, rows: int, cols: int, *, nz: int = 1, ldx: Optional[int] = None, ldy: Optional[int] = None,
              zsx: int = 0, zsy: int = 0, out: Optional[torch.Tensor] = None, y_offset: int = 0) -> torch.Tensor:
    """out[z][c][r] = x[z][r][c] (element strides; see mf_transpose).  A bf16 `out` takes the rounding variant (mf_transpose_bf16)."""
    ldx = cols if ldx is None else ldx
    ldy = rows if ldy is None else ldy
    if x.dtype == torch.bfloat16:
        _req_cuda(x, out)
        if out is None or out.dtype != torch.bfloat16 or y_offset:
            raise MfhipError("transpose: a bf16 input takes a bf16 `out`")
        _check(load().mf_transpose_bf16_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), nz, rows, cols, C.c_int64(ldx), C.c_int64(ldy),
                                             C.c_int64(zsx), C.c_int64(zsy), _stream()), "mf_transpose_bf16_bf16")
        return out
    if out is not None and out.dtype == torch.bfloat16:
        _f32(x)
        _req_cuda(out)
        _check(load().mf_transpose_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr() + 2 * y_offset), nz, rows, cols, C.c_int64(ldx),
                                        C.c_int64(ldy), C.c_int64(zsx), C.c_int64(zsy), _stream()), "mf_transpose_bf16")
        return out
    _f32(x, out)
    if out is None:
        out = torch.empty(nz, cols, ldy, dtype=torch.float32, device=x.device)
    _check(load().mf_transpose(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr() + 4 * y_offset), nz, rows, cols, C.c_int64(ldx),
                               C.c_int64(ldy), C.c_int64(zsx), C.c_int64(zsy), _stream()), "mf_transpose")
    return out


def colsum(x: torch.Tensor, n: int, *, segs: int = 1, rows_per_seg: Optional[int] = None, ldx: Optional[int] = None,
           out: Optional[torch.Tensor] = None, ldo: Optional[int] = None, accumulate: bool = False) -> torch.Tensor:
    _f32(x, out)
    ldx = n if ldx is None else ldx
    rows_per_seg = x.numel() // ldx // segs if rows_per_seg is None else rows_per_seg
    if out is None:
        out = torch.empty(segs, n, dtype=torch.float32, device=x.device)
    lib = load()
    ws = scratch("colsum", int(lib.mf_colsum_ws_floats(segs, C.c_int64(rows_per_seg), n)), x.device)
    _check(lib.mf_colsum(C.c_void_p(x.data_ptr()), C.c_int64(ldx), C.c_void_p(out.data_ptr()), C.c_int64(n if ldo is None else ldo), segs,
                         C.c_int64(rows_per_seg), n, int(accumulate), C.c_void_p(ws.data_ptr()), _stream()), "mf_colsum")
    return out


def groupnorm_bwd(x0: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, groups: int, eps: float, silu: bool,
                  x1: Optional[torch.Tensor] = None, want_param_grads: bool = True, streaming: bool = True,
                  grad_acc: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, add0: Optional[torch.Tensor] = None,
                  add1: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None):
    """`stats`: the [B, groups, 2] (mean, rstd) hip.groupnorm(..., stats_out=) kept for this input (the streaming form skips its
    statistics pass).  Returns (dx0, dx1 or None, dgamma_part [B, C] or None, dbeta_part).  streaming=False withholds the workspace, which keeps
    the one-block-per-(image, group) kernel at every size (tests compare the two).  grad_acc = (dgamma, dbeta) fp32 [C]: where
    the shape runs the streaming form the parameter gradients are ADDED there by the kernel itself and the partials come back
    None; elsewhere it is ignored (sum the partials with colsum)."""
    _f32(x0, x1, dy, gamma, beta)
    b, c0 = x0.shape[0], x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    hw = x0.numel() // (b * c0)
    dx0 = torch.empty_like(x0)
    dx1 = torch.empty_like(x1) if x1 is not None else None
    fused = bool(grad_acc is not None and want_param_grads and streaming and load().mf_groupnorm_bwd_streams(b, hw, c0, c1))
    dg = torch.empty(b, c0 + c1, dtype=torch.float32, device=x0.device) if (want_param_grads and not fused) else None
    db = torch.empty_like(dg) if dg is not None else None
    d = GroupNormBwdDesc()
    if fused:
        _f32(*grad_acc)
        d.dgamma_acc, d.dbeta_acc = _ptr(grad_acc[0]), _ptr(grad_acc[1])
    for t, ref in ((add0, x0), (add1, x1)):       # dx = gradient + add: add is laid out like the matching input
        if t is not None and (ref is None or t.numel() != ref.numel() or not t.is_contiguous()):
            raise MfhipError("groupnorm_bwd: add0 / add1 must be contiguous and sized like x0 / x1")
    _f32(add0, add1)
    d.add0, d.add1 = _ptr(add0), _ptr(add1)
    d.x0, d.x1, d.c0, d.c1, d.dy = _ptr(x0), _ptr(x1), c0, c1, _ptr(dy)
    d.gamma, d.beta, d.dx0, d.dx1, d.dgamma_part, d.dbeta_part = _ptr(gamma), _ptr(beta), _ptr(dx0), _ptr(dx1), _ptr(dg), _ptr(db)
    d.batch, d.hw, d.groups, d.silu, d.eps = b, hw, groups, int(silu), eps
    if stats is not None:
        _f32(stats)
        if stats.numel() != b * groups * 2 or not stats.is_contiguous():
            raise MfhipError("groupnorm_bwd: stats is a contiguous [batch, groups, 2] tensor")
        d.stats_in = stats.data_ptr()
    if streaming:
        d.ws = _ptr(scratch("gn_bwd", int(load().mf_groupnorm_bwd_ws_floats(b, hw, c0 + c1, groups)), x0.device))
    _check(load().mf_groupnorm_bwd(C.byref(d), _stream()), "mf_groupnorm_bwd")
    return dx0, dx1, dg, db


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, eps: float, want_param_grads: bool = True,
                  add: Optional[torch.Tensor] = None):
    """add (optional, sized like x): dx = gradient + add."""
    _f32(x, dy, gamma, add)
    if add is not None and (add.numel() != x.numel() or not add.is_contiguous()):
        raise MfhipError("layernorm_bwd: add must be contiguous and sized like x")
    c = x.shape[-1]
    rows = x.numel() // c
    dx = torch.empty_like(x)
    nb = int(load().mf_layernorm_bwd_parts(rows))
    dg = torch.empty(nb, c, dtype=torch.float32, device=x.device) if want_param_grads else None
    db = torch.empty_like(dg) if want_param_grads else None
    _check(load().mf_layernorm_bwd(C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(gamma.data_ptr()),
                                   C.c_void_p(dx.data_ptr()), C.c_void_p(_ptr(dg)), C.c_void_p(_ptr(db)), C.c_int64(rows), c,
                                   C.c_float(eps), C.c_void_p(_ptr(add)), _stream()), "mf_layernorm_bwd")
    return dx, dg, db


def softmax_bwd(p: torch.Tensor, dp: torch.Tensor, cols: int, scale: float) -> torch.Tensor:
    _f32(p, dp)
    ld = p.shape[-1]
    ds = torch.empty_like(p)
    _check(load().mf_softmax_bwd(C.c_void_p(p.data_ptr()), C.c_void_p(dp.data_ptr()), C.c_void_p(ds.data_ptr()),
                                 C.c_int64(p.numel() // ld), cols, ld, C.c_float(scale), _stream()), "mf_softmax_bwd")
    return ds


def silu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _f32(x, dy)
    dx = torch.empty_like(x)
    _check(load().mf_silu_bwd(C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(dx.data_ptr()), C.c_int64(x.numel()),
                              _stream()), "mf_silu_bwd")
    return dx


def geglu_bwd(h: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    _f32(h, dout)
    c = h.shape[-1] // 2
    dh = torch.empty_like(h)
    _check(load().mf_geglu_bwd(C.c_void_p(h.data_ptr()), C.c_void_p(dout.data_ptr()), C.c_void_p(dh.data_ptr()),
                               C.c_int64(h.numel() // (2 * c)), c, _stream()), "mf_geglu_bwd")
    return dh


def geglu_bwd_bf16(h16: torch.Tensor, dout: torch.Tensor, bias_grad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mf_geglu_bwd_bf16: h16 bf16 [..., 2c], dout fp32 [..., c] -> dh bf16 [..., 2c]; the column sums of dh are ADDED into bias_grad."""
    _req_cuda(h16)
    _f32(dout, bias_grad)
    c = h16.shape[-1] // 2
    rows = h16.numel() // (2 * c)
    if h16.dtype != torch.bfloat16 or not h16.is_contiguous() or not dout.is_contiguous() or dout.numel() != rows * c:
        raise MfhipError("geglu_bwd_bf16: contiguous bf16 [rows, 2c] pre-activation and fp32 [rows, c] gradient")
    dh = torch.empty_like(h16)
    lib = load()
    ws = scratch("geglu_bwd", int(lib.mf_geglu_bwd_bf16_ws_floats(C.c_int64(rows), c)), h16.device) if bias_grad is not None else None
    _check(lib.mf_geglu_bwd_bf16(C.c_void_p(h16.data_ptr()), C.c_void_p(dout.data_ptr()), C.c_void_p(dh.data_ptr()), C.c_int64(rows), c,
                                 C.c_void_p(_ptr(bias_grad)), C.c_void_p(_ptr(ws)), _stream()), "mf_geglu_bwd_bf16")
    return dh


def zero_insert2x(x: torch.Tensor) -> torch.Tensor:
    _f32(x)
    b, h, w, c = x.shape
    y = torch.empty(b, 2 * h, 2 * w, c, dtype=torch.float32, device=x.device)
    _check(load().mf_zero_insert2x(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), b, h, w, c, _stream()), "mf_zero_insert2x")
    return y


def sumpool2x2(x: torch.Tensor) -> torch.Tensor:
    _f32(x)
    b, h2, w2, c = x.shape
    y = torch.empty(b, h2 // 2, w2 // 2, c, dtype=torch.float32, device=x.device)
    _check(load().mf_sumpool2x2(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), b, h2 // 2, w2 // 2, c, _stream()), "mf_sumpool2x2")
    return y


def mse_grad(pred: torch.Tensor, target: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    _f32(pred, target, weights)
    rows = pred.shape[0]
    d = torch.empty_like(pred)
    _check(load().mf_mse_grad(C.c_void_p(pred.data_ptr()), C.c_void_p(target.data_ptr()), C.c_void_p(_ptr(weights)),
                              C.c_void_p(d.data_ptr()), rows, C.c_int64(pred.numel() // rows), _stream()), "mf_mse_grad")
    return d


def sumsq(x: torch.Tensor, out: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """out[0] (float64, device) (+)= sum x^2."""
    _f32(x)
    lib = load()
    key = ("sumsq", torch.device(x.device).index, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _scratch.get(key)
    if ws is None:
        ws = torch.empty(int(lib.mf_sumsq_ws_doubles()), dtype=torch.float64, device=x.device)
        _scratch[key] = ws
    _check(lib.mf_sumsq(C.c_void_p(x.data_ptr()), C.c_int64(x.numel()), C.c_void_p(out.data_ptr()), int(accumulate),
                        C.c_void_p(ws.data_ptr()), _stream()), "mf_sumsq")
    return out


def clip_coef(sumsq_t: torch.Tensor, max_norm: float, coef: torch.Tensor, norm_out: Optional[torch.Tensor] = None,
              unscale: float = 1.0) -> None:
    _check(load().mf_clip_coef(C.c_void_p(sumsq_t.data_ptr()), C.c_float(max_norm), C.c_float(unscale), C.c_void_p(coef.data_ptr()),
                               C.c_void_p(_ptr(norm_out)), _stream()), "mf_clip_coef")


def adamw(w: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
          weight_decay: float = 1e-2, step: int, grad_scale: Optional[torch.Tensor] = None) -> None:
    _f32(w, g, m, v, grad_scale)
    _check(load().mf_adamw(C.c_void_p(w.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()),
                           C.c_int64(w.numel()), C.c_float(lr), C.c_float(betas[0]), C.c_float(betas[1]), C.c_float(eps),
                           C.c_float(weight_decay), step, C.c_void_p(_ptr(grad_scale)), _stream()), "mf_adamw")


# ---- image front-end (csrc/frontend.hip) ------------------------------------------------------------------------------
def _mm_ws(device) -> torch.Tensor:
    return scratch("minmax", int(load().mf_minmax_ws_floats()) + 2, device)


def minmax(x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[min, max] of x (where mask > 0) as a 2-element DEVICE tensor."""
    _f32(x, mask)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    _check(load().mf_minmax(C.c_void_p(x.data_ptr()), C.c_void_p(_ptr(mask)), C.c_int64(x.numel()), C.c_void_p(out.data_ptr()),
                            C.c_void_p(_mm_ws(x.device).data_ptr()), _stream()), "mf_minmax")
    return out


def image_normalize(x: torch.Tensor) -> torch.Tensor:
    """VaeImageProcessor.preprocess for a device tensor: 2x - 1 unless the tensor already holds negatives."""
    _f32(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    _check(load().mf_image_normalize(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_int64(x.numel()),
                                     C.c_void_p(minmax(x).data_ptr()), _stream()), "mf_image_normalize")
    return y


def mask_keep(m: torch.Tensor) -> torch.Tensor:
    _f32(m)
    b, c, h, w = m.shape
    out = torch.empty(b, 1, h, w, dtype=torch.float32, device=m.device)
    _check(load().mf_mask_keep(C.c_void_p(m.contiguous().data_ptr()), C.c_void_p(out.data_ptr()), b, c, C.c_int64(h * w), _stream()),
           "mf_mask_keep")
    return out


def concat_channels(srcs, batch: int) -> torch.Tensor:
    """torch.cat(srcs, 1) for NCHW fp32 tensors whose batch divides `batch` (smaller ones are repeated)."""
    _f32(*srcs)
    srcs = [s.contiguous() for s in srcs]
    h, w = srcs[0].shape[-2:]
    n = len(srcs)
    out = torch.empty(batch, sum(s.shape[1] for s in srcs), h, w, dtype=torch.float32, device=srcs[0].device)
    arr = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    ch = (C.c_int32 * n)(*[s.shape[1] for s in srcs])
    bs = (C.c_int32 * n)(*[s.shape[0] for s in srcs])
    _check(load().mf_concat_channels(arr, ch, bs, n, C.c_void_p(out.data_ptr()), batch, C.c_int64(h * w), _stream()), "mf_concat_channels")
    return out


def postprocess(x: torch.Tensor, denormalize: bool = True, uint8: bool = False) -> torch.Tensor:
    """clamp(x / 2 + 0.5, 0, 1) as fp32 NCHW, or (uint8=True) round(. * 255) as uint8 NHWC."""
    _f32(x)
    x = x.contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c) if uint8 else (b, c, h, w), dtype=torch.uint8 if uint8 else torch.float32, device=x.device)
    _check(load().mf_postprocess(C.c_void_p(x.data_ptr()), C.c_void_p(None if uint8 else out.data_ptr()),
                                 C.c_void_p(out.data_ptr() if uint8 else None), b, c, C.c_int64(h * w), int(denormalize), _stream()),
           "mf_postprocess")
    return out


def _sel_ws(device) -> torch.Tensor:
    return scratch("select", int(load().mf_select_ws_bytes()) // 4 + 8, device)


def depth_percentile_normalize(depth: torch.Tensor, signed_range: bool = True) -> torch.Tensor:
    """apply_transforms_depth(normalization_method="percentile"): clip to the [2 %, 98 %] percentiles and map to the range."""
    import math
    _f32(depth)
    depth = depth.contiguous()
    n = depth.numel()
    ranks, ts = [], []
    for q in (2.0, 98.0):
        pos = q / 100.0 * (n - 1)
        lo = math.floor(pos)
        ranks += [lo, min(lo + 1, n - 1)]
        ts.append(pos - lo)
    rk = torch.tensor(ranks, dtype=torch.int64).to(depth.device)
    vals = torch.empty(4, dtype=torch.float32, device=depth.device)
    out = torch.empty_like(depth)
    _check(load().mf_depth_percentile_normalize(C.c_void_p(depth.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int64(n), C.c_void_p(rk.data_ptr()),
                                                C.c_float(ts[0]), C.c_float(ts[1]), int(signed_range), C.c_void_p(vals.data_ptr()),
                                                C.c_void_p(_sel_ws(depth.device).data_ptr()), _stream()), "mf_depth_percentile_normalize")
    return out


def select_ranks(x: torch.Tensor, ranks) -> torch.Tensor:
    """The order statistics x_(r) (0-based, ascending) for up to four ranks, on the device."""
    _f32(x)
    x = x.contiguous()
    rk = torch.tensor(list(ranks), dtype=torch.int64).to(x.device)
    vals = torch.empty(len(ranks), dtype=torch.float32, device=x.device)
    _check(load().mf_select_ranks(C.c_void_p(x.data_ptr()), C.c_int64(x.numel()), C.c_void_p(rk.data_ptr()), len(ranks), C.c_void_p(vals.data_ptr()),
                                  C.c_void_p(_sel_ws(x.device).data_ptr()), _stream()), "mf_select_ranks")
    return vals


def bicubic_resize_crop(x: torch.Tensor, resized: tuple, crop: tuple, out_hw: tuple, a: float = 1.0, b: float = 0.0,
                        antialias: bool = False) -> torch.Tensor:
    """x [planes, H, W] fp32 -> a * bicubic(x -> resized)[crop window of out_hw at (top, left) = crop] + b.  antialias: PyTorch's
    antialiased kernel (F.interpolate(..., antialias=True): torchvision 0.18's Resize on tensors) instead of the plain one."""
    _f32(x)
    x = x.contiguous()
    planes, h, w = x.shape
    out = torch.empty(planes, out_hw[0], out_hw[1], dtype=torch.float32, device=x.device)
    fn, name = (load().mf_bicubic_aa_resize_crop, "mf_bicubic_aa_resize_crop") if antialias else (load().mf_bicubic_resize_crop, "mf_bicubic_resize_crop")
    _check(fn(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), planes, h, w, int(resized[0]), int(resized[1]),
              int(crop[0]), int(crop[1]), int(out_hw[0]), int(out_hw[1]), C.c_float(a), C.c_float(b), _stream()), name)
    return out


def axpby_affine(x: torch.Tensor, a: float, b: float) -> torch.Tensor:
    """a * x + b on fp32 planes (the bicubic kernel at scale 1 is the identity: reuse it as the affine map)."""
    planes, h, w = x.shape
    return bicubic_resize_crop(x, (h, w), (0, 0), (h, w), a, b)


def hwc_to_chw_affine(x: torch.Tensor, a: float, b: float) -> torch.Tensor:
    _f32(x)
    x = x.contiguous()
    h, w, c = x.shape
    out = torch.empty(c, h, w, dtype=torch.float32, device=x.device)
    _check(load().mf_hwc_to_chw_affine(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int64(h * w), c, C.c_float(a), C.c_float(b),
                                       _stream()), "mf_hwc_to_chw_affine")
    return out


def depth_normalize(depth: torch.Tensor, mask: Optional[torch.Tensor] = None, max_scene_depth: float = 5.0, delta: float = 0.5,
                    signed_range: bool = True) -> torch.Tensor:
    _f32(depth, mask)
    depth = depth.contiguous()
    out = torch.empty_like(depth)
    _check(load().mf_depth_normalize(C.c_void_p(depth.data_ptr()), C.c_void_p(_ptr(mask.contiguous()) if mask is not None else None),
                                     C.c_void_p(out.data_ptr()), C.c_int64(depth.numel()), C.c_float(max_scene_depth), C.c_float(delta),
                                     int(signed_range), C.c_void_p(_mm_ws(depth.device).data_ptr()), _stream()), "mf_depth_normalize")
    return out
