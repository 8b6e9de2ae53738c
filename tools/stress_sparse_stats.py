#!/usr/bin/env python3
"""Which paths of map_sparse_kernel a set of parity cases takes (stats build: `make -C pyani_plus_amd/csrc stats`): runs
tests/test_gpu_fragani.py::test_sparse_segments_next_to_low_complexity_sequence and two seeds of tests/tools/fragani_stress.py
with libpyani_hip_stats.so and sums the sparse kernel's event counters the library prints under PA_FRAGANI_TRACE=1.
    python3 tools/stress_sparse_stats.py        (GPU box)"""
import os
import re
import subprocess
import sys

PATCH = """
import sys, os, runpy
from pathlib import Path
sys.path.insert(0, os.getcwd())
from pyani_plus_amd import _capi
_capi.LIB_PATH = Path('pyani_plus_amd/_lib/libpyani_hip_stats.so').resolve()
_capi.TOOLS_LIB_PATH = _capi.LIB_PATH  # (the stats build is a tools build: the switches exist)
"""


def run(label, code, args):
    env = dict(os.environ, PA_FRAGANI_TRACE="1")
    r = subprocess.run([sys.executable, "-c", PATCH + code, *args], env=env, capture_output=True, text=True)
    past = groups = cands = 0
    for line in (r.stderr + r.stdout).splitlines():
        m = re.search(r"sparse stats: (\d+) segments with a candidate, (\d+) candidates, (\d+) groups.* (\d+) candidates whose first hit lies past", line)
        if m:
            cands += int(m.group(2)); groups += int(m.group(3)); past += int(m.group(4))
    if r.returncode:
        print((r.stderr + r.stdout)[-1500:])
    print(f"{label}: rc {r.returncode}, sparse candidates {cands}, groups evaluated {groups}, candidates whose first hit lies past the batch of 512 window ids {past}")


run("low-complexity test", "import pytest\nsys.exit(pytest.main(['-q', '-x', '-s', 'tests/test_gpu_fragani.py', '-k', 'next_to_low_complexity']))\n", [])
for seed in ("3", "5"):
    run(f"stress seed {seed}", "sys.argv = ['tests/tools/fragani_stress.py', '150', sys.argv[1]]\nrunpy.run_path('tests/tools/fragani_stress.py', run_name='__main__')\n", [seed])
