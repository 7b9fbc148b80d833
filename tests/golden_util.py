"""Helpers shared by the golden-vector tests."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def layout(name):
    with open(os.path.join(GOLD, name)) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


def sub(t, cs=4, ys=3, xs=3):
    return t[:, ::cs, ::ys, ::xs]


def rel_err(got, ref):
    """max |got-ref| / max(|ref|)  -- the "1e-3 rel fp32" bar of BASELINE.json is read as relative to the
    tensor's scale (element-wise relative error is meaningless next to zero crossings)."""
    got = torch.as_tensor(got, dtype=torch.float32)
    ref = torch.as_tensor(ref, dtype=torch.float32)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()
