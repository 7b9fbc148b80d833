#!/usr/bin/env python3
"""Launch ONE conv shape/tile a few times (PMC target for rocprofv3).  usage: conv_one.py <shape_idx> <tile> [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_bench  # noqa: E402

idx, tile = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
print(conv_bench.SHAPES[idx][0], "tile", tile, conv_bench.run(conv_bench.SHAPES[idx], tile, iters))
