#!/usr/bin/env python3
"""Tuning probe (GPU box only): ONE model instance with B sequences against L independent instances ("lanes") with B/L
sequences each, every lane on its own stream and never joined between steps -- how much of the step is latency (small
launches that do not fill 256 CUs) rather than throughput.   python tools/lanes_probe.py [B] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import weights_util as wu  # noqa: E402
from bench import model_args  # noqa: E402
from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H, W = 180, 240
dev = torch.device("cuda", 0)


def make(b, seed):
    m = DCEIFlowCistaNet(model_args(H, W)).eval()
    wu.fill_module(m, 1234)
    m = m.to(dev)
    m.event_flownet.return_flow_preds = True
    evs = [wu.synth_events(b, 5, H, W, seed + i).to(dev) for i in range(4)]
    return {"m": m, "evs": evs, "prev": torch.zeros(b, 1, H, W, device=dev), "st": None, "i": 0}


def step(l):
    I, bf, st = l["m"]({"event_voxel": l["evs"][l["i"] % 4], "rec_img0": l["prev"]}, l["st"], {})
    l["prev"], l["st"] = I, st
    l["i"] += 1


with torch.no_grad():
    for lanes in (1, 2, 4):
        if B % lanes:
            continue
        ls = [make(B // lanes, 100 * j) for j in range(lanes)]
        ss = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
        for order in ("interleaved", "staggered"):
            for _ in range(4):
                for l, s in zip(ls, ss):
                    with torch.cuda.stream(s):
                        step(l)
            if order == "staggered" and lanes > 1:      # lane j runs j/lanes of a step behind lane 0
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                for l, s in zip(ls, ss):
                    with torch.cuda.stream(s):
                        step(l)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print("lanes %d x B %d (%s): %.3f ms per step of %d frames = %.1f frames/s" % (lanes, B // lanes, order, el / K * 1e3, B, B * K / el), flush=True)
            if lanes == 1:
                break
        del ls
