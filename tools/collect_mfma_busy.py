#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python bench.py ...
    python tools/collect_mfma_busy.py <dir> profiles/r01_mfma_busy.txt

SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles the matrix pipe of a SIMD is occupied (64 per v_mfma_f32_32x32x2_f32),
summed over the chip; GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active cycles (MI355X_MICROARCH.md, DVFS note), so
   busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)       effective clock = GUI_ACTIVE / 8 / duration
(the clock quotient reads high on dispatches shorter than ~0.3 ms; the busy fraction does not depend on it).
"""
import collections
import csv
import glob
import re
import sys


def main():
    d, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("cf::", "").replace(" ", "")
            if n.startswith("conv_dma_kernel<"):       # ring depth: default 3 is not part of bench.py's names
                n = re.sub(r",4>$", ",nbuf4>", re.sub(r",3>$", ">", n))
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[n] += 1
    rows = []
    for n, c in acc.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui <= 0:
            continue
        rows.append((busy, n, cnt[n], busy / (gui / 8.0 * 1024.0), gui / 8.0 / max(cnt[n], 1)))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write("# MFMA pipe busy fraction per kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; bench.py --steps 4 --warmup 2)\n")
        f.write("# busy = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs): share of the kernels' own cycles in which a SIMD's matrix pipe is occupied\n")
        f.write("%-52s %8s %10s %16s\n" % ("kernel", "launches", "mfma_busy", "cycles/launch"))
        for busy, n, k, frac, cyc in rows:
            if busy <= 0:
                continue
            f.write("%-52s %8d %9.1f%% %16.0f\n" % (n[:52], k, 100.0 * frac, cyc))
    print(open(out).read())


if __name__ == "__main__":
    main()
