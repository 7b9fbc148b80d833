"""Side-stream fork / join bookkeeping of the library (cista_flow_amd/csrc/fork_join.h) as a host-only unit test: the same template
cf_api.hip instantiates with HIP runs here over a recording mock, under AddressSanitizer + UBSan (a stream / event
index past the tables, a double fork or an un-joined stream must be an error code, not a crash inside cf_step)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fork_join_table_under_asan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "fork_join_test")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-Werror",
           "-I", os.path.join(ROOT, "cista_flow_amd", "csrc"), os.path.join(ROOT, "tests", "native", "fork_join_test.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stdout + b.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
