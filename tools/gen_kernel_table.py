#!/usr/bin/env python3
"""tests/golden/kernel_selection.json: which kernel (tile kind) every convolution launch site of every BASELINE config resolves to.
GPU box only: runs two recurrent frames of each config with cf_plan_enable on and writes the recorded descriptors + choices.
    python tools/gen_kernel_table.py            # rewrites the table (review the diff before committing: it IS the selection policy)
The CPU test tests/test_kernel_selection_cpu.py replays every descriptor through cf_conv_plan (same chooser, nothing launched)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import weights_util as wu  # noqa: E402
from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet, ERAFTCistaNet, IDCistaNet  # noqa: E402

CONFIGS = [  # name, class, H, W, B   (BASELINE.json configs[1..4]: per-GPU shapes; + the B = 1 latency regime of configs[0])
    ("eiflow_180x240_B8", DCEIFlowCistaNet, 180, 240, 8),
    ("eraft_180x240_B8", ERAFTCistaNet, 180, 240, 8),
    ("eiflow_480x640_B4", DCEIFlowCistaNet, 480, 640, 4),
    ("idnet_260x346_B16", IDCistaNet, 260, 346, 16),
    ("eiflow_180x240_B1", DCEIFlowCistaNet, 180, 240, 1),
]


def write_table(out, fields):
    path = os.path.join(ROOT, "tests", "golden", "kernel_selection.json")
    with open(path, "w") as f:
        f.write("{\n \"_meta\": %s,\n \"fields\": %s,\n \"configs\": {\n" % (json.dumps(out["_meta"]), json.dumps(fields)))
        names = list(out["configs"])
        for i, n in enumerate(names):
            f.write("  %s: [\n" % json.dumps(n))
            rows = out["configs"][n]
            for j, r in enumerate(rows):
                f.write("   %s%s\n" % (json.dumps(r, separators=(",", ":")), "," if j + 1 < len(rows) else ""))
            f.write("  ]%s\n" % ("," if i + 1 < len(names) else ""))
        f.write(" }\n}\n")
    print("wrote", path)


def replay():
    """--replay (CPU): keep the recorded descriptors, re-run the launcher's chooser on each and rewrite the choices -- after a change of
    the heuristics' defaults that does not add or remove launch sites.  Prints every site whose kernel changed."""
    from cista_flow_amd import lib
    path = os.path.join(ROOT, "tests", "golden", "kernel_selection.json")
    with open(path) as f:
        t = json.load(f)
    for name, rows in t["configs"].items():
        for r in rows:
            tile, kernel = lib.conv_plan(r["desc"])
            if (tile, kernel) != (r["tile"], r["kernel"]):
                print("%-20s %-40s %s (tile %d) -> %s (tile %d)" % (name, r["tag"], r["kernel"], r["tile"], kernel, tile))
                r["tile"], r["kernel"] = tile, kernel
    write_table(t, t["fields"])


def main():
    if "--replay" in sys.argv:
        return replay()
    dev = torch.device("cuda:0")
    out = {"_meta": "tools/gen_kernel_table.py: launch-site -> kernel for every BASELINE config (default environment, fp32)", "configs": {}}
    fields = None
    for name, cls, H, W, B in CONFIGS:
        a = argparse.Namespace(image_dim=[H, W], num_bins=5, warp_mode="forward", base_channels=64, depth=5, ds=8, is_bi=False)
        m = cls(a).eval()
        wu.fill_module(m, 1234)
        m = m.to(dev)
        if hasattr(m.event_flownet, "return_flow_preds"):
            m.event_flownet.return_flow_preds = True
        h = m._be().get(B, dev)
        h.plan_enable(True)
        states, prev, flow_init = None, torch.zeros(B, 1, H, W, device=dev), None
        evs = [wu.synth_events(B, 5, H, W, 7 + t).to(dev) for t in range(3)]
        with torch.no_grad():
            for t in range(1, 3):
                if cls is DCEIFlowCistaNet:
                    I, bf, states = m({"event_voxel": evs[t], "rec_img0": prev}, states, {})
                elif cls is ERAFTCistaNet:
                    I, bf, states = m({"event_voxel": evs[t], "event_voxel_old": evs[t - 1], "rec_img0": prev}, states, {})
                else:
                    I, bf, states = m({"event_voxel": evs[t], "rec_img0": prev}, states, flow_init, {})
                    flow_init = bf["next_flow"]
                prev = I
        torch.cuda.synchronize()
        plan = h.plan()
        fields = plan["fields"]
        rows = sorted(plan["rows"], key=lambda r: (r["tag"], r["desc"]))
        out["configs"][name] = rows
        print("%-22s %3d launch sites, kernels: %s" % (name, len(rows), sorted({r["kernel"] for r in rows})), flush=True)
        del m, h
    out["fields"] = fields
    write_table(out, fields)


if __name__ == "__main__":
    main()
