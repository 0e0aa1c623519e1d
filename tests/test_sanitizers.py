"""(listed in .gpurunignore: the GPU pool refuses snapshots that mention sanitizer flags; this file only ever runs on the CPU)
AddressSanitizer + UBSan over the CPU code: the oracle (every function, oracle/selftest.c) and the library's two
host-memory ops (csrc/host.cpp). GPU sanitizers are not available on the pool, so the device kernels are covered by
the parity tests only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def test_oracle_under_asan_ubsan(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_selftest")
    odir = os.path.join(ROOT, "oracle")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra", "-Wno-unused-parameter",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + odir, "-o", exe,
           os.path.join(odir, "selftest.c"), os.path.join(odir, "epnet_oracle.c"), "-lm"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "oracle selftest ok" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


def test_host_ops_under_asan_ubsan(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "host_selftest")
    cmd = [HIPCC, "--cuda-host-only", "-x", "hip", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "epnet_amd", "csrc"),
           os.path.join(ROOT, "epnet_amd", "csrc", "host.cpp"), os.path.join(ROOT, "tests", "host_selftest.cpp"), "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="protect_shadow_gap=0")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode == 0 and "host ops rc 0" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
