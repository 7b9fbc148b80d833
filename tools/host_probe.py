#!/usr/bin/env python3
"""Host side of a step (round 4, small-batch regime): how long does the host take to ENQUEUE one steady-state frame, and how much of
that is inside HIP?  GPU box:
    LD_PRELOAD=$PWD/tools/probes/hip_time_shim.so python tools/host_probe.py [B]
Prints host enqueue time per step (the call returns; nothing is waited for), end-to-end time per step, and -- with the shim -- the time inside
hipLaunchKernel / hipEventRecord / hipStreamWaitEvent."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import weights_util as wu  # noqa: E402
from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W, N = 180, 240, 200
dev = torch.device("cuda:0")
a = argparse.Namespace(image_dim=[H, W], num_bins=5, warp_mode="forward", base_channels=64, depth=5, ds=8, is_bi=False)
m = DCEIFlowCistaNet(a).eval()
wu.fill_module(m, 1234)
m = m.to(dev)
evs = [wu.synth_events(B, 5, H, W, 7 + t).to(dev) for t in range(4)]
shim = None
try:
    g = ctypes.CDLL(None)
    g.hip_shim_reset
    shim = g
except (OSError, AttributeError):
    pass
states, prev = None, torch.zeros(B, 1, H, W, device=dev)
with torch.no_grad():
    for t in range(20):
        prev, bf, states = m({"event_voxel": evs[t & 3], "rec_img0": prev}, states, {})
    torch.cuda.synchronize()
    if shim:
        shim.hip_shim_reset()
    # (a) free-running: the host runs ahead of the GPU until the runtime's limit on commands in flight, then every launch waits: "host"
    #     time here is the GPU's time whenever the GPU is the slower of the two
    host = 0.0
    t0 = time.perf_counter()
    for t in range(N):
        h0 = time.perf_counter()
        prev, bf, states = m({"event_voxel": evs[t & 3], "rec_img0": prev}, states, {})
        host += time.perf_counter() - h0
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d free-running: forward calls %.3f ms per step, drained %.3f ms per step" % (B, host / N * 1e3, (t2 - t0) / N * 1e3), flush=True)
    if shim:
        shim.hip_shim_report()
        shim.hip_shim_reset()
    # (b) one step at a time into an EMPTY queue: what the host needs to enqueue a step when nothing holds it back
    host, e2e = 0.0, 0.0
    for t in range(N):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        prev, bf, states = m({"event_voxel": evs[t & 3], "rec_img0": prev}, states, {})
        h1 = time.perf_counter()
        torch.cuda.synchronize()
        host += h1 - h0
        e2e += time.perf_counter() - h0
print("B=%d one step at a time: host enqueue %.3f ms per step, step end to end %.3f ms" % (B, host / N * 1e3, e2e / N * 1e3), flush=True)
if shim:
    print("(shim totals below are over %d steps)" % N, flush=True)
    shim.hip_shim_report()
