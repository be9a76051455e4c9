"""Listing-level check of the hand-waited prefetch loads of csrc/igemm3.hip (igemm3s_kernel).

The kernel issues its patch prefetch through inline assembly and waits for it with a hand-written `s_waitcnt vmcnt(N)`:
the compiler does not know those registers are in flight, so a register copy placed between a load and the wait would
read stale data.  tools/asm_load_check.py walks every path from an inline load to the wait on the compiler's listing;
this test compiles the file for gfx950 (no GPU needed) and requires zero violations -- a compiler or source change that
breaks the assumption fails here, not as a silent wrong result on the GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_instruction_touches_an_in_flight_prefetch_register(tmp_path):
    src = os.path.join(ROOT, "bird-sound-event-detecion_amd", "csrc", "igemm3.hip")
    lst = str(tmp_path / "igemm3.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
                    "--cuda-device-only", "-S", src, "-o", lst], check=True, timeout=900,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_load_check.py"), lst],
                       capture_output=True, text=True, timeout=600)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0, r.stdout[-2000:]
    kernels = int(last.split()[0])
    assert kernels >= 16, last          # every igemm3s instance carries the inline loads
    assert last.endswith("0 violations"), last
