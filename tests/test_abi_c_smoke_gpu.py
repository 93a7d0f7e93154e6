"""integration/c/smoke.c — a plain-C program that dlopen()s libmaelsim.so, runs echo x 8 (BASELINE configs[0]) on device 0 through
the C-ABI alone and checks KAT-1 and the checker verdicts; compiled with gcc and executed here."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_plain_c_caller_runs_echo(lib, tmp_path):
    exe = str(tmp_path / "smoke")
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "c", "smoke.c"),
                    "-ldl", "-o", exe], check=True)
    r = subprocess.run([exe, os.path.join(ROOT, "maelstrom_amd", "libmaelsim.so")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "smoke ok" in r.stdout and r.stdout.count("valid? true") == 8
