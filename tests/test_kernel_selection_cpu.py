"""Which kernel does each convolution launch site of each BASELINE config resolve to?  The committed table
tests/golden/kernel_selection.json (recorded on the GPU by tools/gen_kernel_table.py) is replayed, descriptor by descriptor, through
cf_conv_plan -- the launcher's own chooser run dry, pure host logic -- so a change of the launcher's heuristics
(CF_WINO_MIN / CF_WINO16_MAX / ... defaults, tile rules) shows up HERE as a named diff instead of as a silent slowdown."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "golden", "kernel_selection.json")
KNOBS = ("CF_WINO_MIN", "CF_WINO_SK", "CF_WINO_SK2_MAX", "CF_WINO_SK4_MAX", "CF_WINO16_MAX", "CF_WINO16_MIN", "CF_WINO16_KMIN",
         "CF_WINO1D_MIN", "CF_WINO4_MIN", "CF_WINOP", "CF_PATCH", "CF_DMA", "CF_SCHED")


def _table():
    with open(TABLE) as f:
        return json.load(f)


@pytest.mark.parametrize("config", sorted(_table()["configs"]) if os.path.exists(TABLE) else [])
def test_launch_sites_resolve_to_the_committed_kernels(config):
    if any(k in os.environ for k in KNOBS):
        pytest.skip("a launcher knob is set in the environment: the table is for the defaults")
    from cista_flow_amd import lib
    t = _table()
    assert len(t["fields"]) == len(t["configs"][config][0]["desc"])
    diffs = []
    for row in t["configs"][config]:
        tile, kernel = lib.conv_plan(row["desc"])
        if (tile, kernel) != (row["tile"], row["kernel"]):
            diffs.append("%s: committed %s (tile %d), launcher now picks %s (tile %d)" % (row["tag"], row["kernel"], row["tile"], kernel, tile))
    assert not diffs, "\n".join(diffs)


def test_table_covers_every_baseline_config():
    t = _table()
    assert {"eiflow_180x240_B8", "eraft_180x240_B8", "eiflow_480x640_B4", "idnet_260x346_B16", "eiflow_180x240_B1"} <= set(t["configs"])
    # the production kernels of the headline config, by name: a table regenerated with a knob set by accident would lose one of them
    kernels = {r["kernel"] for r in t["configs"]["eiflow_180x240_B8"]}
    for k in ("conv_wino_kernel", "conv_wino16_kernel", "conv_wino1d_kernel", "conv_wino_sk_kernel<2>"):
        assert k in kernels, k
