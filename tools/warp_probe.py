"""Stand-alone timing of the sparse-code warp (tuning tool, GPU box only): python tools/warp_probe.py
Under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` it gives the fabric traffic of warp_kernel alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cista_flow_amd.utils.flow_utils import FrameWarp

B, C, h, w = 8, 128, 90, 120
g = torch.Generator().manual_seed(1)
z = torch.randn(B, C, h, w, generator=g).cuda().contiguous(memory_format=torch.channels_last)
flow = (3.0 * torch.randn(B, 2, 2 * h, 2 * w, generator=g)).cuda()
# smooth flow like a real field: low-pass it
flow = torch.nn.functional.avg_pool2d(flow, 9, 1, 4)
junk = torch.empty(512 * 1024 * 1024 // 4, device="cuda")      # flushes L2 / Infinity Cache between launches
fw = FrameWarp("forward")
for _ in range(3):
    out = fw.warp_frame(z, flow)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    junk.zero_()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fw.warp_frame(z, flow)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
def cold(fn, n=10):
    out = []
    for _ in range(n):
        junk.zero_()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return sorted(out)[len(out) // 2]


dst = torch.empty_like(z)
print("reference points, cold caches: z.clone-like copy_ %.1f us | zero-flow warp %.1f us | warm warp %.1f us" % (
    cold(lambda: dst.copy_(z)), cold(lambda: fw.warp_frame(z, torch.zeros_like(flow))),
    sorted([(lambda a, b: (a.record(), fw.warp_frame(z, flow), b.record(), torch.cuda.synchronize(), a.elapsed_time(b) * 1e3)[-1])(
        torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)])[5]))
nbytes = 4.0 * B * h * w * (2 * C + 2)
print("Z-warp %dx%dx%dx%d cold caches: median %.1f us (min %.1f) -> %.2f TB/s of %.1f MB algorithmic" % (
    B, C, h, w, sorted(ts)[len(ts) // 2], min(ts), nbytes / (sorted(ts)[len(ts) // 2] * 1e-6) / 1e12, nbytes / 1e6))
