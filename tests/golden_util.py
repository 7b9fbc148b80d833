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


def err_norms(got, ref):
    """The three error figures a parity assertion checks, all relative to the reference tensor:
      inf   max|d| / max|ref|                      (error against the tensor's scale)
      l2    ||d||_2 / ||ref||_2                    (sees a small error spread over a small-magnitude tensor that `inf` cannot)
      elem  max(|d| - 1e-3 |ref|) / max|ref|       (what is left of an element's error after its own 1e-3 relative allowance)"""
    got = torch.as_tensor(got, dtype=torch.float32).double()
    ref = torch.as_tensor(ref, dtype=torch.float32).double()
    d = (got - ref).abs()
    scale = ref.abs().max().clamp_min(1e-12)
    return {"inf": (d.max() / scale).item(),
            "l2": (d.square().sum().sqrt() / ref.square().sum().sqrt().clamp_min(1e-12)).item(),
            "elem": ((d - 1e-3 * ref.abs()).max() / scale).item()}


def rel_err(got, ref):
    """The figure every golden / oracle comparison asserts `< tol` on: the LARGEST of
      * max|got-ref| / max|ref|  -- BASELINE.json's "1e-3 rel fp32" read relative to the tensor's scale (element-wise relative error
        is meaningless next to zero crossings);
      * the relative L2 error ||got-ref||_2 / ||ref||_2  (second norm);
      * twice the element-wise excess max(|got-ref| - 1e-3 |ref|) / max|ref|, i.e. `< tol` asserts
        |got-ref| <= 1e-3 |ref| + (tol / 2) max|ref| for EVERY element.
    With the tolerances in use (2e-4 on the GPU path, 5e-5 on the oracle; contract 1e-3) the 1e-3 bar is therefore held in the
    max norm, in the L2 norm and element-wise at once."""
    n = err_norms(got, ref)
    return max(n["inf"], n["l2"], 2.0 * n["elem"])
