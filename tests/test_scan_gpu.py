"""The single-pass scan (opus_amd/csrc/scan.h: decoupled look-back, DPP inside the wave) against the three-kernel scan and a host
loop: additions, the non-commutative "last defined value", exclusive and inclusive, in place and not, 1 .. 216 M elements (the token
count of the 1 GB workload).  tools/scan_check.hip is built by __graft_entry__.build() (tools/Makefile) and runs as a child process."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "_build", "scan_check")


@pytest.mark.gpu
def test_single_pass_scan_equals_three_kernel_scan_and_host_loop():
    assert os.path.exists(EXE), "tools/_build/scan_check is missing: run __graft_entry__.build()"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.strip().endswith("all ok"), r.stdout[-4000:] + r.stderr[-2000:]
    assert r.stdout.count("mismatches 0 ") == r.stdout.count("excl="), r.stdout[-4000:]
