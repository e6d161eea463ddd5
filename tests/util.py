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
