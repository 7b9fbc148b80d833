"""bench.py control flow with more than one rank.  The real N > 1 run (RCCL over xGMI) only happens on the driver's
8-GPU node; this rehearsal drives the same code with two ranks sharing the one GPU and gloo collectives
(CF_BENCH_REHEARSE=1), so that a rank-asymmetric collective (a hang on the real node) is caught here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    """A rendezvous port nobody holds right now: the GPU host's network namespace is shared with other jobs, and a fixed port that
    happens to be taken turns these tests into a silent 7-minute wait."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_bench_two_rank_rehearsal(gpu):
    env = dict(os.environ, CF_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "2", "--height", "128", "--width", "128"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # rank 0 prints exactly one JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["roofline"] is not None and out["cpu_baseline"] is None


def test_bench_single_rank_rccl_path(gpu):
    """RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the real backend can only be exercised here
    with a world of one: CF_BENCH_FORCE_DIST=1 runs bench.py's whole N > 1 code path -- init_process_group("nccl",
    device_id), all_gather_into_tensor of the reconstructed frames on the side stream with record_stream, the barriers
    and the MAX all-reduce of the step time -- and checks that the gathered frames equal the local ones."""
    env = dict(os.environ, CF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port())
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "2", "--height", "128",
           "--width", "128", "--no-cpu-baseline", "--no-alt", "--no-roofline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0


def test_bench_self_launch_two_ranks(gpu):
    """`python bench.py --gpus 2` without a torchrun environment starts the two-rank job itself (child process, before any HIP
    call) and relays rank 0's single JSON line; here the ranks share the one GPU over gloo (CF_BENCH_REHEARSE=1).  Also the
    strong-scaling split: 3 sequences over 2 ranks (ragged shards 2 + 1, padded all-gather)."""
    env = dict(os.environ, CF_BENCH_REHEARSE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--strong", "3",
           "--height", "128", "--width", "128", "--no-alt"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["scaling"] == "strong"
    assert out["config"]["total_sequences"] == 3 and out["value"] > 0
