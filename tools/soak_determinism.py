#!/usr/bin/env python3
"""Soak test (GPU box): the same recurrent loop run twice must give bit-identical frames and states at every step
(a hand-off race in a kernel would show up as a mismatch somewhere in a few hundred steps x ~280 launches)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import weights_util as wu  # noqa: E402
from bench import model_args  # noqa: E402
from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--height", type=int, default=180)
ap.add_argument("--width", type=int, default=240)
a = ap.parse_args()
dev = torch.device("cuda", 0)
m = DCEIFlowCistaNet(model_args(a.height, a.width)).eval()
wu.fill_module(m, 1234)
m = m.to(dev)
evs = [wu.synth_events(a.batch, 5, a.height, a.width, 77 + i).to(dev) for i in range(8)]


def flat(x):
    """tensors of a nested state structure, in order"""
    if torch.is_tensor(x):
        return [x]
    if isinstance(x, (list, tuple)):
        return [t for e in x for t in flat(e)]
    if isinstance(x, dict):
        return [t for k in sorted(x) for t in flat(x[k])]
    return []


def loop():
    prev, states, sums = torch.zeros(a.batch, 1, a.height, a.width, device=dev), None, []
    with torch.no_grad():
        for i in range(a.steps):
            prev, bf, states = m({"event_voxel": evs[i % 8], "rec_img0": prev}, states, {})
            sums.append(torch.stack([prev.double().sum(), bf["flow_final"].double().abs().sum()] + [s.double().abs().sum() for s in flat(states)]))
    torch.cuda.synchronize()
    return torch.stack(sums).cpu(), prev.cpu(), [s.cpu() for s in flat(states)]


s1, p1, st1 = loop()
s2, p2, st2 = loop()
bad = (s1 != s2).any(dim=1).nonzero().flatten().tolist()
same = torch.equal(p1, p2) and all(torch.equal(x, y) for x, y in zip(st1, st2))
print("steps %d: per-step checksums differ at %d steps%s; final frames/states bit-identical: %s; finite: %s" % (
    a.steps, len(bad), (" (first %d)" % bad[0]) if bad else "", same, bool(torch.isfinite(p1).all())))
sys.exit(0 if (not bad and same) else 1)
