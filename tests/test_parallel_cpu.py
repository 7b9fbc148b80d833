"""N>1 path on CPU: world_size-2 gloo run of the sequence sharding + frame collation used by bench.py."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cista_flow_amd.parallel import collate_frames, shard_range   # noqa: E402


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_config3_split_is_four_sequences_per_rank():
    """BASELINE configs[3]: 32 sequences of 480x640 over 8 ranks (`bench.py --gpus 8 --strong 32`): every rank holds exactly 4
    consecutive sequences, in rank order, and a padded all-gather of 4-row shards collates them without a gap."""
    spans = [shard_range(32, r, 8) for r in range(8)]
    assert spans == [(4 * r, 4 * r + 4) for r in range(8)]
    # ragged totals near it keep the contiguous, at-most-one-apart property bench.py's padded all-gather relies on
    for n in (31, 33, 35):
        sizes = [shard_range(n, r, 8)[1] - shard_range(n, r, 8)[0] for r in range(8)]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_bench_pins_each_rank_to_its_own_cores():
    """bench.pin_rank_to_cores: disjoint, equal slices of the cores the process may use, one per local rank; a no-op for one rank."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    if not hasattr(os, "sched_getaffinity"):
        import pytest
        pytest.skip("no sched_setaffinity on this platform")
    before = os.sched_getaffinity(0)
    try:
        assert bench.pin_rank_to_cores(0, 1) is None and os.sched_getaffinity(0) == before
        world = 2 if len(before) >= 2 else 1
        seen = []
        for r in range(world):
            os.sched_setaffinity(0, before)
            n = bench.pin_rank_to_cores(r, world)
            if world > 1:
                mine = os.sched_getaffinity(0)
                assert n == len(mine) == len(before) // world and mine <= before
                assert all(not (mine & o) for o in seen)
                seen.append(mine)
    finally:
        os.sched_setaffinity(0, before)
    model, phys = bench.cpu_info()
    assert isinstance(model, str) and phys >= 1


def _worker(rank, world, port, n_seq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = shard_range(n_seq, rank, world)
    # a stand-in for the per-rank reconstruction: frame b is filled with its global sequence id
    local = torch.stack([torch.full((1, 4, 6), float(i)) for i in range(s, e)]) if e > s else torch.zeros(0, 1, 4, 6)
    out = collate_frames(local, n_seq)
    ok = out.shape[0] == n_seq and all(float(out[i].mean()) == float(i) for i in range(n_seq))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_seq, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_seq, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_collate_equal_shards_gloo():
    _run(8, 29611)


def test_collate_ragged_shards_gloo():
    _run(7, 29612)


def test_c_abi_library_exports_every_declared_symbol():
    """include/cistaflow.h <-> libcistaflow.so: every declared entry point is exported (no compute, no GPU)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "cistaflow.h")).read()
    declared = sorted(set(re.findall(r"\b(cf_[a-z0-9_]+)\s*\(", hdr)))
    from cista_flow_amd import lib
    so = lib.load()
    assert sorted(lib.SYMBOLS) == declared
    for sym in declared:
        assert hasattr(so, sym), sym


def test_no_gpu_means_loud_failure():
    """The product path must not fall back to anything when there is no GPU."""
    import argparse
    import pytest
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet
    a = argparse.Namespace(image_dim=[128, 128], num_bins=5, warp_mode='forward', base_channels=64, depth=5, ds=8, is_bi=False)
    m = DCEIFlowCistaNet(a).eval()
    with pytest.raises(RuntimeError):
        m({"event_voxel": torch.zeros(1, 5, 128, 128), "rec_img0": torch.zeros(1, 1, 128, 128)}, None, {})


def test_traffic_table_uses_the_library_kernel_names():
    """bench.py looks the dominant kernel up in profiles/hbm_traffic.json by the name cf_conv_tile_name() returns:
    every conv entry of the committed table must be such a name (a mismatch silently turns roofline.traffic to null)."""
    import json
    from cista_flow_amd import lib
    L = lib.load()
    names = {L.cf_conv_tile_name(t).decode() for t in range(1, 51)}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json")))
    conv = [k for k in table if k.startswith("conv_dma_kernel") or k.startswith("conv_igemm_kernel")]
    assert conv, "no conv kernels in profiles/hbm_traffic.json"
    for k in conv:
        assert k in names, k


def test_bench_self_launch_relays_child_exit_code():
    """`python bench.py --gpus 2` with no torchrun environment must start the two-rank job itself instead
    of raising.  Without a GPU every rank stops at "needs an MI355X": the launcher has to relay that failure as its own
    non-zero exit code and must not print a JSON line."""
    import subprocess
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by tests/test_bench_gpu.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr, r.stderr[-1500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
