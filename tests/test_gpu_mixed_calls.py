"""Every query entry point of one handle in random order (tests/tools/mix_calls.py): the accumulators
that are zero at rest -- replica counters, shared-hash flags, exclusive sums -- must stay consistent
whatever ran before.  Runs in a child: a fresh process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mixed_calls_on_one_handle(hip_lib):
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "mix_calls.py"), "60", "7"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mixed calls ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
