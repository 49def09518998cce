"""-m gpu: the standalone reproducer of the compiler fault (tests/sweeps/canary/REPORT.md: SGPR split copies in front of a join block's exec
restore let the VGPR allocator's copies in) — the wrong sums an over-sized window kernel returns under the default register allocator
(tests/sweeps/canary/: one recorded launch of exa_hprodw — 256 VGPRs + 84 AGPRs — replayed WITHOUT libexahip against code objects
hipcc builds from the recorded source).  What must hold whatever the compiler does: the build with the library's fallback flags
(exa_build.cpp safe_flags) reproduces the recorded output, register poison included.  What is reported: whether the default build
still shows the fault (ROCm 7.2: 983 of 3008 entries differ) — the day it does not, the fallback has become unnecessary for this
kernel, not wrong."""
import os
import subprocess

import pytest

from conftest import PARITY_LINES, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
HERE = os.path.dirname(os.path.abspath(__file__))


def test_the_fallback_flags_make_the_oversized_kernel_right():
    script = os.path.join(HERE, "sweeps", "canary", "run_canary.sh")
    out = subprocess.run(["bash", script, os.path.join(HERE, "sweeps", "canary"), "default safe sgpr-basic"], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if "exa_hprodw" in ln]
    safe = [ln for ln in lines if ln.startswith("safe")]
    default = [ln for ln in lines if ln.startswith("default")]
    guard = [ln for ln in lines if ln.startswith("sgpr-basic")]       # the library's base flags since round 5: the guard on the cause
    assert len(guard) == 2 and all("equal to the recorded output" in ln for ln in guard), (out.stdout[-2000:], out.stderr[-2000:])
    assert len(safe) == 2 and all("equal to the recorded output" in ln for ln in safe), (out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    assert len(default) == 2
    fault = any("DIFFERENT" in ln for ln in default)
    PARITY_LINES.append("canary (tests/sweeps/canary): default allocator flags -> " + ("DIFFERENT from the recorded output: the fault is present, the fallback is needed"
                                                                                       if fault else "equal: this compiler no longer shows the fault on this kernel"))
