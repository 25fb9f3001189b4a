"""Shared helpers for the parity tests (test infrastructure)."""
import os

import torch

import restatement as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_fixture(name):
    fx = torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)
    if fx.get("state_dict") is None and fx.get("shapes") is not None:
        fx["state_dict"] = R.synthetic_state_dict(fx["shapes"], seed=fx["synthetic_seed"])
    return fx


def cast_sd(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def maxabs(a, b):
    return (a.double() - b.double()).abs().max().item()


def assert_close(a, b, atol, rtol=0.0, what=""):
    a, b = a.double().cpu(), b.double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max|err|={err.max().item():.3e} "
        f"(atol={atol:g}, rtol={rtol:g}), ref absmax={b.abs().max().item():.3e}")
