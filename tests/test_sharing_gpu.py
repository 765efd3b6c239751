"""Reproducibility when the GPU is SHARED (round 5).  Three processes running the same kernels side by side change which workgroups meet on a CU and
how far the wavefronts of a workgroup drift apart: a hazard in a kernel's LDS schedule that a lone process never shows turns up as launches that
differ from the first.  Found this way: the 128 x 128 interleaved-request tap tile on the wide LDS image with its early request schedule (1-5 of
150 launches of ANY layer differed; `conv_taps_il.hip`, ASCHED) — every bit-equality test of a single process had passed on it for a round."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _side_by_side(cmd, n=3, timeout=600):
    procs = [subprocess.Popen([sys.executable] + cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(n)]
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=timeout)
        assert p.returncode == 0, out[-2000:]
        outs.append(out)
    return outs


def test_tap_tiles_reproducible_with_three_processes_on_the_gpu():
    outs = _side_by_side(["tools/layer_race_check.py", "--reps", "60"])
    lines = [l for o in outs for l in o.splitlines() if " launches differ from the first" in l]
    assert len(lines) >= 3 * 20, "every process reports every layer x tile"
    bad = [l for l in lines if not re.search(r": 0 of \d+ launches", l)]
    assert not bad, "\n".join(bad[:10])


def test_fgt_step_reproducible_with_three_processes_on_the_gpu():
    outs = _side_by_side(["tools/determinism_check.py", "--passes", "12"])
    sums = set()
    for o in outs:
        m = re.search(r"(\d+) passes, (\d+) differ from the first; checksum ([\d.]+)", o)
        assert m, o[-1500:]
        assert int(m.group(2)) == 0, o[-1500:]
        sums.add(m.group(3))
    assert len(sums) == 1, f"the processes disagree: {sums}"
