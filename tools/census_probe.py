#!/usr/bin/env python3
"""How many workgroups of a convolution kernel does a CU really hold at once?  (tuning tool, GPU box only)

Needs a -DCF_CENSUS build of conv_igemm.hip / conv_wino4.hip (tools/build_variant.sh); wave 0 of every workgroup records
[HW_ID, XCC_ID, start, end] and this script sweeps the intervals per CU.
    CF_LIB_PATH=$PWD/build_var/census.so CASES=gates:40,gates:42 python tools/census_probe.py
"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

buf = torch.zeros(4 * 400000, dtype=torch.int64, device="cuda")
os.environ["CF_STAMP_BUF"] = str(buf.data_ptr())
import conv_bench  # noqa: E402

for case in os.environ.get("CASES", "gates:40,gates:42,cista.P:40,cista.P:42").split(","):
    name, tile = case.split(":")
    shape = [s for s in conv_bench.SHAPES if s[0].startswith(name)][0]
    buf.zero_()
    r = conv_bench.run(shape, int(tile), 1)
    torch.cuda.synchronize()
    d = buf.view(-1, 4).cpu()
    d = d[d[:, 3] > 0]
    if r is None or d.shape[0] == 0:
        print("%-34s tile %2d: no census records" % (shape[0], int(tile)))
        continue
    per_cu = defaultdict(list)
    for hw, xcc, t0, t1 in d.tolist():
        cu = (xcc & 0xF, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF)        # xcc, se, sh, cu
        per_cu[cu].append((t0, t1))
    peak, life = [], []
    for cu, iv in per_cu.items():
        ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
        cur = best = 0
        for _, s in ev:
            cur += s
            best = max(best, cur)
        peak.append(best)
        life += [b - a for a, b in iv]
    span = (d[:, 3].max() - d[:, 2].min()).item() / 100.0
    print("%-34s tile %2d: %5d workgroups on %3d CUs | resident workgroups per CU: max %d, median %d | lifetime %.1f us (median) | span %.1f us"
          % (shape[0], int(tile), d.shape[0], len(per_cu), max(peak), sorted(peak)[len(peak) // 2], sorted(life)[len(life) // 2] / 100.0, span), flush=True)
