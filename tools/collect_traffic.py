#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
bench.py into profiles/hbm_traffic.json: fabric bytes per launch for every kernel.

gfx950 correction (guide, section HBM): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled;
WRITE_SIZE is exact.  Both counters are in KiB and include Infinity-Cache hits (they count L2's fabric side).

    python tools/collect_traffic.py <dir_with_FETCH_SIZE_pass> <dir_with_WRITE_SIZE_pass> profiles/hbm_traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter, by_grid=False):
    """by_grid: key every kernel as name@grid (work-items), the split a per-launch-site comparison needs."""
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"]
            m = re.match(r"void cf::(conv_igemm_kernel)<(\d+), (\d+), (\d+), (\d+), (\d+), \d+, \d+, (\d+)>", n)
            if m:   # fold the A-mode / precision instantiations of one tile shape together (bench.py's names)
                g = m.groups()
                n = "%s<%s,%s,%s,%s,%s%s>" % (g[0], g[1], g[2], g[3], g[4], g[5], ",kcw32" if g[6] == "32" else "")
            elif re.match(r"void cf::conv_dma_kernel<", n):
                n = re.sub(r"^void cf::", "", n).split("(")[0].replace(" ", "")
                # last template argument = ring depth: 3 is the default and is not part of bench.py's names
                n = re.sub(r",3>$", ">", n)
                n = re.sub(r",4>$", ",nbuf4>", n)
            else:
                n = re.sub(r"^void ", "", n).split("(")[0].replace("cf::", "")
            n = n.replace(" ", "")
            if by_grid:
                n = "%s@%d" % (n, int(r["Grid_Size"]) if "Grid_Size" in r else
                               int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
            acc[n][0] += float(r["Counter_Value"])
            acc[n][1] += 1
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, [0.0, 1])
        w = wr.get(k, [0.0, 1])
        res[k] = {"fetch_bytes_per_launch": round(2.0 * 1024.0 * f[0] / max(f[1], 1)),
                  "write_bytes_per_launch": round(1024.0 * w[0] / max(w[1], 1)),
                  "launches_sampled": int(max(f[1], w[1]))}
        res[k]["bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    # the HBM-class kernels run with several grids (image / sparse-code warp, per-resolution norms): split those by grid
    fg, wg = per_kernel(fd, "FETCH_SIZE", True), per_kernel(wd, "WRITE_SIZE", True)
    bg = {}
    for k in sorted(set(fg) | set(wg)):
        if k.startswith("conv_dma") or k.startswith("conv_igemm"):
            continue
        f = fg.get(k, [0.0, 1])
        w = wg.get(k, [0.0, 1])
        bg[k] = {"fetch_bytes_per_launch": round(2.0 * 1024.0 * f[0] / max(f[1], 1)),
                 "write_bytes_per_launch": round(1024.0 * w[0] / max(w[1], 1)), "launches_sampled": int(max(f[1], w[1]))}
    res["_by_grid"] = bg
    import datetime
    res["_meta"] = {"collected": "%s, tools/make_profiles.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py "
                                 "--steps 4 --warmup 2; FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64), both include Infinity-Cache hits"
                                 % datetime.date.today().isoformat()}
    json.dump(res, open(out, "w"), indent=1)
    res = {k: v for k, v in res.items() if not k.startswith("_")}
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches_sampled"])[:12]:
        print("%-44s %8.1f MB/launch (fetch %7.1f write %7.1f) x %d" % (k, v["bytes_per_launch"] / 1e6, v["fetch_bytes_per_launch"] / 1e6,
                                                                          v["write_bytes_per_launch"] / 1e6, v["launches_sampled"]))


if __name__ == "__main__":
    main()
