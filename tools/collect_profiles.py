#!/usr/bin/env python3
"""Copy the summaries tools/make_profiles.sh left under gpurun_out/prof into profiles/ (tracked).

    python tools/collect_profiles.py r01
"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof")
    dst = os.path.join(ROOT, "profiles")
    # gpurun merges into the local gpurun_out/, so files of earlier runs linger: take the newest
    stats = sorted(glob.glob(src + "/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime, reverse=True)
    assert stats, "no kernel_stats.csv under gpurun_out/prof/stats"
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, tag + "_kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary, round %s, build of the last commit touching csrc/\n" % tag)
        f.write("# command: rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt --no-roofline\n")
        f.write("# 4 warm-up + 20 timed steps of cista-eiflow 180x240 B=8, side streams concurrent (normal operation); the serialised trace\n")
        f.write("# the roofline fractions can be recomputed from is %s_ktrace_serial.txt\n" % tag)
        f.write("%-100s %8s %12s %10s %7s\n" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in rows:
            f.write("%-100s %8s %12.3f %10.2f %7s\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       float(r["AverageNs"]) / 1e3, r["Percentage"]))
    shutil.copy(os.path.join(src, "layers.txt"), os.path.join(dst, tag + "_launch_sites_hip_events.txt"))
    for name in ("ktrace_serial.txt", "stamps.txt", "mfma_clock_probe.txt", "setup_launches.txt"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, tag + "_" + name))
    shutil.copy(os.path.join(src, "hbm_traffic.json"), os.path.join(dst, "hbm_traffic.json"))
    if os.path.exists(os.path.join(src, "mfma_busy.txt")):
        shutil.copy(os.path.join(src, "mfma_busy.txt"), os.path.join(dst, tag + "_mfma_busy.txt"))
    line = open(os.path.join(src, "bench.log")).read().strip().splitlines()[-1]
    open(os.path.join(dst, tag + "_bench_line.json"), "w").write(line + "\n")
    print("profiles/ updated from", src)


if __name__ == "__main__":
    main()
