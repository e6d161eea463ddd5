"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLD, name))


def keys(name):
    with open(os.path.join(GOLD, f"keys_{name}.json")) as f:
        return {m: {k: tuple(v) for k, v in d.items()} for m, d in json.load(f).items()}


def report(name, got, ref, atol, rtol=0.0, fail=True):
    got = torch.as_tensor(np.asarray(got) if not torch.is_tensor(got) else got).float().cpu()
    ref = torch.as_tensor(ref).float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = int((err > tol).sum())
    print(f"{name}: max_abs_err={err.max().item():.3e} mean_abs_err={err.mean().item():.3e} "
          f"ref_absmax={ref.abs().max().item():.3e} bad={bad}/{ref.numel()}")
    if fail:
        assert bad == 0, f"{name}: {bad} elements exceed atol={atol} rtol={rtol}; max err {err.max().item():.3e}"
    return err.max().item()


def strided_sample(t, stride, n=256):
    return t.float().cpu().double().flatten()[::int(stride)][:n].float()


_ENV = {}
# The HIP bf16 path must stay inside this multiple of the REFERENCE's own bf16 deviation from its fp32 results
# (tests/golden/bf16_envelope.json, tools/make_bf16_envelope.py).  Two independent bf16 evaluations of the same net draw
# their rounding errors independently: the means agree closely, the maxima (an extreme-value statistic) less so.
# Round 3: cut from 2.0 / 1.5 to 1.25 / 1.1 — every ratio observed on MI355X is <= 0.85 (printed by report_env as
# "ratio linf / mean"), so a regression that doubles the bf16 error now fails.
ENV_K_LINF, ENV_K_MEAN = 1.25, 1.1
# L-inf over a few hundred elements is an extreme-value statistic of two INDEPENDENT rounding-error draws (ours and the
# reference's): on the tiny fixtures' 2 x 2-pixel tensors (512 elements) it scatters up to 1.53 x while the mean stays at 0.89 x
# (gpurun_out/r03f).  Small tensors get 1.75 x on the maximum; the mean bound — the one a doubled error cannot pass — is the
# same 1.1 x everywhere.
ENV_K_LINF_SMALL, ENV_SMALL_NUMEL = 1.75, 8192


def envelope(key, prec="bf16"):
    """The reference's own deviation from its fp32 results when run in `prec` (bf16: tools/make_bf16_envelope.py; fp16, the
    reference scripts' default torch_dtype: the same tool with --dtype fp16 -> tests/golden/fp16_envelope.json)."""
    if prec not in _ENV:
        with open(os.path.join(GOLD, f"{prec}_envelope.json")) as f:
            _ENV[prec] = json.load(f)
    return _ENV[prec][key]


def report_env(name, got, ref, key, k_linf=None, k_mean=ENV_K_MEAN, prec="bf16"):
    """bf16 / fp16 mode: |got - ref| (ref = the reference's fp32 result) against the reference's own envelope in that dtype for this case."""
    got = torch.as_tensor(np.asarray(got) if not torch.is_tensor(got) else got).float().cpu()
    ref = torch.as_tensor(ref).float()
    err = (got - ref).abs()
    env = envelope(key, prec)
    if k_linf is None:
        k_linf = ENV_K_LINF if ref.numel() >= ENV_SMALL_NUMEL else ENV_K_LINF_SMALL
    linf, mean = err.max().item(), err.mean().item()
    print(f"{name}: max_abs_err={linf:.3e} (reference {prec}: {env['linf']:.3e}) mean_abs_err={mean:.3e} "
          f"(reference {prec}: {env['mean']:.3e}) ref_absmax={ref.abs().max().item():.3e} "
          f"ENVRATIO linf {linf / max(env['linf'], 1e-30):.3f} mean {mean / max(env['mean'], 1e-30):.3f}")
    assert linf == linf, f"{name}: NaN"
    assert linf <= k_linf * env["linf"], f"{name}: L-inf {linf:.3e} > {k_linf} x the reference's {prec} envelope {env['linf']:.3e}"
    assert mean <= k_mean * env["mean"], f"{name}: mean error {mean:.3e} > {k_mean} x the reference's {prec} envelope {env['mean']:.3e}"
    return linf


def check(name, got, ref, prec, tol, key):
    """fp32-class precisions: atol/rtol `tol`; bf16: the reference-derived envelope `key`."""
    if prec in ("bf16", "fp16"):
        return report_env(name, got, ref, key, prec=prec)
    return report(name, got, ref, **tol)
