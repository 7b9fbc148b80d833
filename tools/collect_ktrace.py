#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of a SERIALISED run (CF_SERIAL=1) grouped by (kernel, grid) and joined with the library's
per-launch-site table of algorithmic work, so that every roofline fraction of the bench line can be recomputed from
rocprof's own durations.

    python tools/collect_ktrace.py <dir with *kernel_trace.csv> <layers.txt.json> <out.txt>
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

PEAK_TF, PEAK_TBS = 157.3, 8.0


def norm(name):
    n = re.sub(r"\(.*$", "", name)          # drop the argument list
    n = n.replace("void ", "").replace("cf::", "").replace(" ", "")
    return n


def main():
    src, layers, out = sys.argv[1:4]
    traces = sorted(glob.glob(src + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime, reverse=True)
    assert traces, "no kernel_trace.csv under " + src
    groups = defaultdict(list)
    for r in csv.DictReader(open(traces[0])):
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        groups[(norm(r["Kernel_Name"]), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lib = json.load(open(layers))
    nsteps = lib["steps"]
    # library rows: (kernel, grid) -> class, work per launch, tags
    site = defaultdict(lambda: {"work": 0.0, "launches": 0, "cls": None, "tags": set(), "ms": 0.0})
    for r in lib["rows"]:
        k = (r["kernel"].replace(" ", ""), r["grid"])
        s = site[k]
        s["work"] += r["work"]; s["launches"] += r["launches"]; s["cls"] = r["class"]; s["tags"].add(r["tag"]); s["ms"] += r["ms"]
    lines = []
    tot = {"mfma": [0.0, 0.0], "hbm": [0.0, 0.0]}
    for (kname, grid), durs in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        match = None
        for (lk, lg), s in site.items():
            if lg == grid and (kname == lk or kname.startswith(lk[:-1] + ",") or kname.startswith(lk[:-1] + ">")):
                match = s
                break
        avg = sum(durs) / len(durs)
        if match:
            wpl = match["work"] / match["launches"]
            rate = wpl / (avg * 1e-6) / 1e12
            frac = rate / (PEAK_TF if match["cls"] == "mfma" else PEAK_TBS)
            tot[match["cls"]][0] += wpl * len(durs)
            tot[match["cls"]][1] += sum(durs)
            lines.append("%-46s %9d %6d %9.2f %9.2f  %-4s %10.3f %8.3f %6.3f  %s" % (
                kname[:46], grid, len(durs), avg, match["ms"] * 1e3 / match["launches"], match["cls"], wpl / 1e6, rate, frac,
                ",".join(sorted(match["tags"]))[:60]))
        else:
            lines.append("%-46s %9d %6d %9.2f %9s  %-4s" % (kname[:46], grid, len(durs), avg, "-", "-"))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace of `CF_SERIAL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-roofline`\n")
        f.write("# (side streams folded into one stream: every kernel alone on the chip, SAME grids as the timed step), grouped by (kernel, grid =\n")
        f.write("# work-items) and joined with the library's per-launch-site algorithmic work (bench.py roofline pass, %d steps).\n" % nsteps)
        f.write("# rate = work per launch / rocprof avg duration: TFLOP/s for class mfma (peak %.1f), TB/s for class hbm (peak %.1f).\n" % (PEAK_TF, PEAK_TBS))
        f.write("%-46s %9s %6s %9s %9s  %-4s %10s %8s %6s  %s\n" % ("kernel", "grid", "calls", "avg_us", "hipev_us", "cls", "Mwork/call", "rate", "frac", "launch sites"))
        f.write("\n".join(lines) + "\n")
        for c, (w, us) in tot.items():
            if us:
                r = w / (us * 1e-6) / 1e12
                f.write("# class %s: %.1f us per run of the trace, time-weighted rate %.3f = %.4f of peak\n" % (c, us, r, r / (PEAK_TF if c == "mfma" else PEAK_TBS)))
    print(open(out).read())


if __name__ == "__main__":
    main()
