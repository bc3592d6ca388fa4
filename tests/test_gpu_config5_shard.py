"""GPU: BASELINE.json configs[4] (GTDB full, ~400 k genomes at scaled=100 over 8 GPUs) at ONE GPU's share:
56 000 references x ~39 000 hashes = 2.2e9 reference hashes -- every posting / stream position array is
indexed beyond 2^31 -- against a 10^7-hash sample.  tests/tools/scale_probe.py does the work: streaming
kernel == sample-driven kernel == the independent one-wave-per-reference bsearch kernel == a torch count,
and overlap + exclusive counts == the CPU oracle (the box has the host memory: 18 GB of hashes)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


def test_one_gpu_shard_of_gtdb_full(hip_lib, capsys):
    import scale_probe

    rc = scale_probe.main(["--refs", "56000", "--median", "33000", "--sample", "10000000", "--steps", "3", "--oracle", "auto"])
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    print(line)
    assert rc == 0, line
    assert res["positions_exceed_2^31"] and res["overlap_equals_bsearch_kernel"] and res["overlap_sum_equals_torch_count"]
    assert res["indexed_equals_stream"]
    assert res["equals_cpu_oracle"] is True, "the GPU box is expected to have the host memory for the oracle leg"
