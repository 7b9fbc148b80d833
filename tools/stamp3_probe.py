#!/usr/bin/env python3
"""Per-group cycle stamps INSIDE one step of conv_wino4_kernel's chunk loop (tuning tool, GPU box only).

Needs a -DCF_STAMP -DW4_STAMP3=<class> build (class 0 = the step after the hand-off barrier, which also issues the next
chunk's DMA; 1 = the first step of a chunk):   CF_LIB_PATH=build_var/stamp3_0.so python tools/stamp3_probe.py
Prints, per wave row (i = 0..5) and averaged, the cumulative cycles at the end of each of the step's six MFMA groups
and at the end of the step (row transform + descriptor advance).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

buf = torch.zeros(8 * 12 * 40000, dtype=torch.int64, device="cuda")
os.environ["CF_STAMP_BUF"] = str(buf.data_ptr())
import conv_bench  # noqa: E402

for name in os.environ.get("CASES", "gates,cista.P").split(","):
    shape = [s for s in conv_bench.SHAPES if s[0].startswith(name)][0]
    buf.zero_()
    r = conv_bench.run(shape, 42, 3)
    torch.cuda.synchronize()
    d = buf.view(-1, 12, 8).cpu().double()
    d = d[d[:, 0, 4] > 0]
    n = d[0, 0, 4].item()
    cols = [0, 1, 2, 3, 5, 6, 7]
    print("%s  %.1f us  chunks %d  workgroups %d" % (shape[0], r[0], n, d.shape[0]))
    for w in range(12):
        print("  wave %2d: " % w + " ".join("%6.0f" % (d[:, w, c].mean().item() / n) for c in cols))
    print("  mean   : " + " ".join("%6.0f" % (d[:, :, c].mean().item() / n) for c in cols), flush=True)
