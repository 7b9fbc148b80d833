#!/usr/bin/env python3
"""Tuning probe (GPU box): does running the B sequences as P independent part-batch pipelines (own handle, own stream, own host
thread each) beat one lock-step batch?  Sequences are independent, and the phases of a step differ in what bounds them
(the 1/8-resolution update iterations under-fill the chip, the CISTA convs saturate it), so staggered pipelines can overlap
one pipeline's latency-bound phase with another's MFMA-bound one.

    python tools/dual_pipeline_probe.py [--batch 8] [--pipes 1,2,4] [--steps 40]
prints aggregate reconstructed frames/s per pipeline count.
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import weights_util as wu  # noqa: E402
from bench import model_args  # noqa: E402
from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet  # noqa: E402


def run(P, B, H, W, steps, warmup, stagger):
    dev = torch.device("cuda", 0)
    b = B // P
    models, evs, streams = [], [], []
    for p in range(P):
        m = DCEIFlowCistaNet(model_args(H, W)).eval()
        wu.fill_module(m, 1234)
        m = m.to(dev)
        m.event_flownet.return_flow_preds = True
        models.append(m)
        evs.append([wu.synth_events(b, 5, H, W, 1234 + 10 * p + i).to(dev) for i in range(4)])
        streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    barrier = threading.Barrier(P + 1)

    def worker(p):
        torch.cuda.set_device(dev)
        prev, states = torch.zeros(b, 1, H, W, device=dev), None
        with torch.no_grad(), torch.cuda.stream(streams[p]):
            for i in range(warmup):
                prev, _, states = models[p]({"event_voxel": evs[p][i % 4], "rec_img0": prev}, states, {})
            streams[p].synchronize()
            barrier.wait()
            if stagger and p:
                time.sleep(stagger * p * 1e-3)
            for i in range(steps):
                prev, _, states = models[p]({"event_voxel": evs[p][i % 4], "rec_img0": prev}, states, {})
            streams[p].synchronize()
            barrier.wait()

    th = [threading.Thread(target=worker, args=(p,)) for p in range(P)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    return B * steps / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--width", type=int, default=240)
    ap.add_argument("--pipes", default="1,2,4")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--stagger-ms", type=float, default=0.0)
    a = ap.parse_args()
    for P in [int(x) for x in a.pipes.split(",")]:
        if a.batch % P:
            continue
        v = run(P, a.batch, a.height, a.width, a.steps, a.warmup, a.stagger_ms)
        print("pipelines %d x B=%d: %.1f frames/s aggregate" % (P, a.batch // P, v), flush=True)


if __name__ == "__main__":
    main()
