#!/usr/bin/env python3
"""What an HBM-class kernel of the CISTA + warp step can reach on this machine (tuning tool, GPU box only).

north_star asks for >= 0.9 of the 8 TB/s HBM roofline on that step; the ceiling a kernel can be held to is what a plain device copy of the
same footprint achieves.  Prints, for the footprints of the step's HBM-class kernels at 180x240 B=8 (and B=4, the half-batch chains):
torch's own copy_ (read + write), a read-only reduction and a write-only fill, each from flushed caches (a 512 MiB memset in between)
and back to back (Infinity-Cache resident), next to the library's warp of the sparse code.  -> profiles/r03_hbm_ceiling_probe.txt
"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from cista_flow_amd.utils.flow_utils import FrameWarp

dev = torch.device("cuda:0")
junk = torch.empty(512 * 1024 * 1024 // 4, device=dev)


def timed(fn, cold, n=15):
    ts = []
    for _ in range(n):
        if cold:
            junk.zero_()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


print("# median of 15, HIP events; 'cold' = caches flushed by a 512 MiB memset before every sample, 'warm' = back to back")
print("%-46s %9s %9s %9s %9s" % ("operation", "MB moved", "cold us", "cold TB/s", "warm TB/s"))
for name, shape in (("sparse code z  [8,128,90,120]", (8, 128, 90, 120)), ("sparse code z, half batch [4,128,90,120]", (4, 128, 90, 120)),
                    ("up-sampled features [4,64,180,240]", (4, 64, 180, 240))):
    x = torch.randn(shape, device=dev)
    y = torch.empty_like(x)
    mb = x.numel() * 4 / 1e6
    for op, fn, moved in (("copy_ (read + write)", lambda: y.copy_(x), 2 * mb), ("sum (read only)", lambda: x.sum(), mb),
                          ("fill_ (write only)", lambda: y.fill_(1.0), mb)):
        c, w = timed(fn, True), timed(fn, False)
        print("%-46s %9.1f %9.1f %9.2f %9.2f" % (name[:24] + " " + op, moved, c, moved / c, moved / w))
B, C, h, w = 8, 128, 90, 120
z = torch.randn(B, C, h, w, device=dev).contiguous(memory_format=torch.channels_last)
flow = torch.nn.functional.avg_pool2d(3.0 * torch.randn(B, 2, 2 * h, 2 * w, device=dev), 9, 1, 4)
fw = FrameWarp("forward")
mb = 4.0 * B * h * w * (2 * C + 2) / 1e6
c, wv = timed(lambda: fw.warp_frame(z, flow), True), timed(lambda: fw.warp_frame(z, flow), False)
print("%-46s %9.1f %9.1f %9.2f %9.2f" % ("cf_warp of z (smooth 3 px flow)", mb, c, mb / c, mb / wv))
