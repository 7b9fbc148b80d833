#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernel over tile shapes (tuning tool, GPU box only).

    python tools/conv_bench.py            # the layer shapes of cista-eiflow at 180x240, B=8
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cista_flow_amd import lib  # noqa: E402

L = lib.load()
dev = torch.device("cuda:0")

# name, B, Cin, H, W, Cout, KH, KW, stride, padT, padL, pad_mode
SHAPES = [
    ("gru.zr    384->256 1x5 @24x32", 8, 384, 24, 32, 256, 1, 5, 1, 0, 2, 0),
    ("gru.q     384->128 1x5 @24x32", 8, 384, 24, 32, 128, 1, 5, 1, 0, 2, 0),
    # the per-iteration GRU convolutions as the model runs them (the context third of K is precomputed once per step: gru.pre)
    ("zr.h      256->256 1x5 @23x30", 8, 256, 23, 30, 256, 1, 5, 1, 0, 2, 0),
    ("q.h       256->128 1x5 @23x30", 8, 256, 23, 30, 128, 1, 5, 1, 0, 2, 0),
    ("zr.v      256->256 5x1 @23x30", 8, 256, 23, 30, 256, 5, 1, 1, 2, 0, 0),
    ("q.v       256->128 5x1 @23x30", 8, 256, 23, 30, 128, 5, 1, 1, 2, 0, 0),
    ("convc1    336->256 1x1 @23x30", 8, 336, 23, 30, 256, 1, 1, 1, 0, 0, 0),
    ("convf2    128->64  3x3 @23x30", 8, 128, 23, 30, 64, 3, 3, 1, 1, 1, 0),
    ("l2.0.c1    64->96  3x3/2 @90x120", 8, 64, 90, 120, 96, 3, 3, 2, 1, 1, 0),
    ("l3.0.c1    96->128 3x3/2 @45x60", 8, 96, 45, 60, 128, 3, 3, 2, 1, 1, 0),
    ("l2.ds      64->96  1x1/2 @90x120", 8, 64, 90, 120, 96, 1, 1, 2, 0, 0, 0),
    ("convc2    256->192 3x3 @24x32", 8, 256, 24, 32, 192, 3, 3, 1, 1, 1, 0),
    ("menc      320->128 3x3 @24x32", 8, 320, 24, 32, 128, 3, 3, 1, 1, 1, 0),
    ("fh.conv1  128->256 3x3 @24x32", 8, 128, 24, 32, 256, 3, 3, 1, 1, 1, 0),
    ("layer3    128->128 3x3 @24x32", 8, 128, 24, 32, 128, 3, 3, 1, 1, 1, 0),
    ("layer2     96->96  3x3 @48x64", 8, 96, 48, 64, 96, 3, 3, 1, 1, 1, 0),
    ("layer1     64->64  3x3 @96x128", 8, 64, 96, 128, 64, 3, 3, 1, 1, 1, 0),
    ("cista.D   128->64  3x3 @90x120", 8, 128, 90, 120, 64, 3, 3, 1, 1, 1, 1),
    ("cista.P    64->128 3x3 @90x120", 8, 64, 90, 120, 128, 3, 3, 1, 1, 1, 1),
    ("gates     192->256 3x3 @90x120", 8, 192, 90, 120, 256, 3, 3, 1, 1, 1, 1),
    ("out_gates 256->128 3x3 @90x120", 8, 256, 90, 120, 128, 3, 3, 1, 1, 1, 1),
    ("Gates     128->256 3x3 @90x120", 8, 128, 90, 120, 256, 3, 3, 1, 1, 1, 1),
    ("W0         64->64  3x3/2 @180x240", 8, 64, 180, 240, 64, 3, 3, 2, 1, 1, 1),
    # the same CISTA layers at cista-idnet 260x346 B=16 (config 5) and cista-eiflow 480x640 B=4 (config 4)
    ("big.gates 192->256 3x3 @130x173 B16", 16, 192, 130, 173, 256, 3, 3, 1, 1, 1, 1),
    ("big.out_g 256->128 3x3 @130x173 B16", 16, 256, 130, 173, 128, 3, 3, 1, 1, 1, 1),
    ("big.P      64->128 3x3 @130x173 B16", 16, 64, 130, 173, 128, 3, 3, 1, 1, 1, 1),
    ("big.D     128->64  3x3 @130x173 B16", 16, 128, 130, 173, 64, 3, 3, 1, 1, 1, 1),
    ("hs.gates  192->256 3x3 @240x320 B4", 4, 192, 240, 320, 256, 3, 3, 1, 1, 1, 1),
    ("hs.P       64->128 3x3 @240x320 B4", 4, 64, 240, 320, 128, 3, 3, 1, 1, 1, 1),
]


PREC = int(os.environ.get("PREC", "0"))


def run(shape, tile, iters=20):
    name, B, Cin, H, W, Cout, KH, KW, stride, pT, pL, pm = shape
    B = int(os.environ.get("BATCH", B))
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, KH, KW, device=dev) / (Cin * KH * KW) ** 0.5
    b = torch.randn(Cout, device=dev)
    z = os.environ.get("ZERO", "")   # DVFS probe: all-zero operands draw less power (MI355X_MICROARCH.md, DVFS give-back item 1)
    if z in ("1", "x"): x.zero_()    # ZERO=1: everything, ZERO=x: activations only, ZERO=w: weights only
    if z in ("1", "w"): w.zero_(); b.zero_()
    Ho = (H + 2 * pT - KH) // stride + 1
    Wo = (W + 2 * pL - KW) // stride + 1
    out = torch.empty(B, Ho, Wo, Cout, device=dev)
    ms = C.c_float(0)
    rc = L.cf_op_conv2d_bench(lib.ptr(x), B, Cin, H, W, lib.ptr(w), lib.ptr(b), Cout, KH, KW, stride, pT, pL, pm, 0, 0,
                              tile, lib.ptr(out), lib.current_stream_ptr(), iters, C.byref(ms), PREC)
    if rc != 0:
        return None
    flops = 2.0 * B * Ho * Wo * Cout * Cin * KH * KW
    return ms.value * 1e3, flops / (ms.value * 1e-3) / 1e12


def main():
    tiles = [int(t) for t in os.environ.get("TILES", "0,1,2,4,8,9,10,11,12,13,14").split(",")]
    only = os.environ.get("SHAPES")
    for sh in SHAPES:
        if only and not any(o in sh[0] for o in only.split(",")):
            continue
        res = []
        for t in tiles:
            r = run(sh, t)
            res.append("t%d: %s" % (t, "n/a" if r is None else "%7.1fus %5.1fTF" % r))
        print("%-32s %s" % (sh[0], " | ".join(res)), flush=True)


if __name__ == "__main__":
    main()
