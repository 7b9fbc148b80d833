#!/usr/bin/env python3
"""In-kernel cycle accounting of conv_dma_kernel (tuning tool, GPU box only).

Needs a -DCF_STAMP build of the library:
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DCF_STAMP -o build_var/lib_stamp.so cista_flow_amd/csrc/*.hip
    CF_LIB_PATH=$PWD/build_var/lib_stamp.so python tools/stamp_probe.py
Every wave records (s_memtime) the cycles it spent waiting for its DMA (vmcnt), at the stage barrier, issuing the
next stage, and the lengths of its prologue / main loop / tail.  The stamps themselves cost ~100 cycles each.
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

CASES = os.environ.get("CASES", "cista.D:23,cista.P:23,cista.P:28,gates:28,gates:25,gru.zr:20,gru.q:22,convc2:20,layer1:23")
for case in CASES.split(","):
    name, tile = case.split(":")
    shape = [s for s in conv_bench.SHAPES if s[0].startswith(name)][0]
    buf.zero_()
    r = conv_bench.run(shape, int(tile), 3)
    torch.cuda.synchronize()
    d = buf.view(-1, 8).cpu()
    d = d[d[:, 4] > 0].double()
    if r is None or d.shape[0] == 0:
        print("%-34s tile %2d  no stamps (tile not available for this shape)" % (shape[0], int(tile)), flush=True)
        continue
    n = d[:, 4].mean().item()
    m = d.mean(0)
    comp = (m[5] - m[0] - m[1] - m[2]) / n
    if int(tile) == 42:     # conv_wino4_kernel: [vmcnt wait + barrier, DMA issue, steps 0..2, prologue, chunks, loop, exchange, MHz]
        print("%-34s tile 42  %7.1f us %5.1f TF | chunks %3d | per chunk: steps0-2 %6.0f  wait+barrier %6.0f  dma issue %5.0f  step3 %6.0f"
              " | prologue %6.0f  loop %7.0f  exchange %6.0f cycles | shader clock %4.0f MHz" % (
                  shape[0], r[0], r[1], n, m[2] / n, m[0] / n, m[1] / n, (m[5] - m[0] - m[1] - m[2]) / n, m[3], m[5], m[6], m[7]), flush=True)
        continue
    if int(tile) in (47, 50):     # conv_wino16_kernel<0 | 1>: [wait + barrier, tail: drain + dead loads, tail: exchange, prologue, chunks, loop, tail, MHz]
        print("%-34s tile %2d  %7.1f us %5.1f TF | chunks %3d | per chunk: dma-wait+barrier %4.0f  rest %4.0f | prologue %6.0f  loop %7.0f  tail %6.0f = drain + dead loads + barrier %5.0f, exchange %5.0f, fused epilogue %5.0f cycles | shader clock %4.0f MHz" % (
                  shape[0], int(tile), r[0], r[1], n, m[0] / n, (m[5] - m[0]) / n, m[3], m[5], m[6], m[1], m[2], m[6] - m[1] - m[2], m[7]), flush=True)
        continue
    if int(tile) == 47:     # (old format)
        print("%-34s tile 47  %7.1f us %5.1f TF | chunks %3d | per chunk: dma-wait+barrier %4.0f  issue + reads + transform + 16 mfma(16x16x4) %4.0f"
              " | prologue %6.0f  loop %7.0f  tail %6.0f cycles | shader clock %4.0f MHz" % (
                  shape[0], r[0], r[1], n, m[0] / n, (m[5] - m[0]) / n, m[3], m[5], m[6], m[7]), flush=True)
        continue
    if int(tile) == 46:     # conv_wino1d_kernel: [DMA wait + barrier, -, -, prologue, chunks, loop, tail, MHz]
        print("%-34s tile 46  %7.1f us %5.1f TF | chunks %3d | per chunk: dma-wait+barrier %4.0f  issue + reads + transform + 12 mfma %4.0f"
              " | prologue %6.0f  loop %7.0f  tail %6.0f cycles | shader clock %4.0f MHz" % (
                  shape[0], r[0], r[1], n, m[0] / n, (m[5] - m[0]) / n, m[3], m[5], m[6], m[7]), flush=True)
        continue
    if int(tile) in (48, 49):     # conv_wino_p_kernel<0 | 1>: [DMA wait + barrier, items, issue + transform, begin -> first chunk step, chunks (all items), loops, tails, MHz]
        items = d[:, 1].mean().item()
        print("%-34s tile %2d  %7.1f us %5.1f TF | items per workgroup %4.2f (max %d) | per chunk: dma-wait+barrier %4.0f  issue+transform %4.0f  frag reads + 16 mfma %4.0f"
              " | begin -> first chunk step %6.0f (once per workgroup)  per item: loop %7.0f  tail + hand-over %6.0f cycles | shader clock %4.0f MHz" % (
                  shape[0], int(tile), r[0], r[1], items, int(d[:, 1].max().item()), m[0] / n, m[2] / n, (m[5] - m[0] - m[2]) / n, m[3], m[5] / items, m[6] / items, m[7]), flush=True)
        continue
    if int(tile) == 40:     # conv_wino_kernel: [DMA wait + first barrier, second barrier, issue + input transform]
        print("%-34s tile 40  %7.1f us %5.1f TF | chunks %3d | per chunk: dma-wait+barrier %4.0f  issue+transform %4.0f  barrier %4.0f  "
              "frag reads + 16 mfma %4.0f | prologue %6.0f  loop %7.0f  tail %6.0f cycles | shader clock %4.0f MHz" % (
                  shape[0], r[0], r[1], n, m[0] / n, m[2] / n, m[1] / n, comp, m[3], m[5], m[6], m[7]), flush=True)
        continue
    print("%-34s tile %2d  %7.1f us %5.1f TF | stages %3d | per stage: vmcnt-wait %4.0f  barrier %4.0f  issue %4.0f  lds+mfma %4.0f"
          " | prologue %6.0f  loop %7.0f  tail %6.0f cycles | shader clock %4.0f MHz (median %4.0f)" % (
              shape[0], int(tile), r[0], r[1], n, m[0] / n, m[1] / n, m[2] / n, comp, m[3], m[5], m[6], m[7], d[:, 7].median().item()), flush=True)
