"""GPU: the forms of the sample-driven lookup that the small randomized cases of tests/tools/fuzz_parity.py do not
reach by themselves -- the 1024-lane tiles (samples of 4e5+ hashes), the presence filter in front of the buckets
(databases of 1e6+ distinct hashes) -- forced onto them, where the CPU oracle checks every count.  (Their default is
the 256-lane aggregating form.)  The knobs are tuning switches behind YH_DEBUG_TUNING=1, read once per process, hence
the subprocess."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [
    {"YH_INDEX_TILE": "2", "YH_FILTER_MIN": "1", "YH_FILTER_BPH": "2"},   # crowded filter: many false positives
    {"YH_INDEX_TILE": "1", "YH_FILTER_MIN": "1", "YH_FILTER_BPH": "16"},
], ids=["tile2-filter2", "tile1-filter16"])
def test_forced_lookup_forms_against_oracle(hip_lib, env):
    e = dict(os.environ, YH_DEBUG_TUNING="1")
    e.update(env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_parity.py"), "--seconds", "6", "--seed", "31", "--no-batch"],
                       env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert p.returncode == 0 and res["fuzz"] == "ok", line + p.stderr[-2000:]
    assert res["rounds"] >= 3
