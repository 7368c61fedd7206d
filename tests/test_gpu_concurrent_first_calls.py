"""The first concurrent calls against a scene (tests/concurrent_first_calls.py has the story): fresh processes, 8 threads, a dense
scene, rows of 120 ids (straight to a batch launch + the wide kernel) and rows of 64 (through the open batch and its stage-100
fallback).  Every call equals the patch of the one batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rows", [120, 64])
def test_first_concurrent_single_patch_calls_equal_the_batch(rows):
    for _ in range(4):   # (fresh processes: the workspaces are grown once per process; the old clear failed about one process in six)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "concurrent_first_calls.py"), str(rows), "8", "1"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert "mismatches 0" in r.stdout
