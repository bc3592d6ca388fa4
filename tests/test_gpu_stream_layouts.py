"""The streaming lookup (k_stream_lookup over the hash-sorted delta stream) on edge cases against the oracle -- FORCED:
the library's own choice for databases this small is the sample-driven kernel.  Second pass: the run step through the
general exclusive pass (YH_NO_FUSED_RUN behind the YH_DEBUG_TUNING gate, read once per process, hence the child)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from oracle import oracle
from yacht_amd import _lib, synth
from yacht_amd.engine import RefDB
from tests.test_gpu_keys import _colliding_case

def check(values, offsets, sample, **kw):
    with RefDB(values, offsets, **kw) as db:
        assert db.info()["stream_layout"] == 1, db.info()
        db.set_lookup(_lib.YH_LOOKUP_STREAM)
        assert db.lookup_choice(max(sample.size, 1)) == _lib.YH_LOOKUP_STREAM
        ov, e, m = db.run_counts(sample)
        ov_only = db.overlap(sample)
    w_ov = oracle.overlap(values, offsets, sample)
    mask = (w_ov > 0).astype(np.uint8)
    w_e, w_m = oracle.exclusive(values, offsets, mask, sample)
    assert np.array_equal(ov, w_ov) and np.array_equal(ov_only, w_ov)
    assert np.array_equal(e, np.where(mask, w_e, 0)) and np.array_equal(m, np.where(mask, w_m, 0))

v, o, s = _colliding_case()
check(v, o, s)
with RefDB(v, o, flags=1) as db_noidx:       # YH_DB_NO_INDEX: the stream without the inverted index
    assert np.array_equal(db_noidx.overlap(s), oracle.overlap(v, o, s))
v, o, s = synth.config2(seed=5)
check(v, o, s)
v, o = synth.config4(seed=6, n_clusters=40, size=500)   # many equal hashes: runs of delta 0
rng = np.random.default_rng(1)
s = np.unique(np.concatenate([rng.choice(v, 4000), rng.integers(0, 2**63, 3000, dtype=np.uint64)]))
check(v, o, s)
# full-range hashes, a gap far above 255 truncated units, an empty reference, a one-hash database
refs = [np.array([0, 1, 2**40, 2**63, 2**64 - 1], dtype=np.uint64), np.zeros(0, np.uint64),
        np.array([5, 2**40, 2**40 + 1], dtype=np.uint64)]
offs = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.uint64)
check(np.concatenate(refs), offs, np.array([0, 5, 2**40, 2**64 - 1], dtype=np.uint64))
check(np.array([77], dtype=np.uint64), np.array([0, 1], dtype=np.uint64), np.array([3, 77, 99], dtype=np.uint64))
print("stream layout ok")
"""


@pytest.mark.parametrize("extra", [{}, {"YH_DEBUG_TUNING": "1", "YH_NO_FUSED_RUN": "1"}], ids=["fused-run", "general-run"])
def test_stream_lookup_matches_oracle(hip_lib, extra):
    """(second case: the run step through the general exclusive pass -- shared-hash flags, k_excl_worklist,
    k_excl_pieces over the postings, k_excl_final -- instead of the fused three-launch form)"""
    env = dict(os.environ, **extra)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "stream layout ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


CHILD_WIDE_TILE = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from oracle import oracle
from yacht_amd.engine import RefDB

# One workgroup (YH_STREAM_WGS=1, a tuning switch behind YH_DEBUG_TUNING) over a database whose truncated-key range exceeds 2^32: the sample keys
# of a tile are staged as 32-bit offsets from its first key, so a tile must end where the offset would
# overflow -- a sample of a few hashes at the two ends and in the middle of the range forces that.
rng = np.random.default_rng(11)
H = 70_000_000                       # the range of truncated keys is 32..64 x H: 4.4e9 needs ~62 x H
top = int(4.4e9) << 31               # hashes uniform below this -> stream_shift 31, key range 4.4e9 > 2^32
vals = np.unique(rng.integers(0, top, size=H, dtype=np.uint64))
refs = [vals[0::2].copy(), vals[1::2].copy()]                 # two interleaved references
values = np.concatenate(refs)
offsets = np.array([0, refs[0].size, refs[0].size + refs[1].size], dtype=np.uint64)
picks = np.concatenate([vals[:3], vals[vals.size // 2 - 1: vals.size // 2 + 2], vals[-3:]])
sample = np.unique(np.concatenate([picks, np.array([5, top // 2 + 12345, top - 2, 2**64 - 2], dtype=np.uint64)]))
with RefDB(values, offsets, flags=1 | 16) as db:               # overlap only, no bucket table: every query streams
    info = db.info()
    assert info["stream_layout"] == 1
    span = int(vals[-1] >> info["stream_shift"]) - int(vals[0] >> info["stream_shift"])
    assert span > 2**32, (span, info["stream_shift"])
    got = db.overlap(sample)
want = oracle.overlap(values, offsets, sample)
assert np.array_equal(got, want), (got, want)
assert int(want.sum()) >= 7
print("wide tile ok", span, got.tolist())
"""


def test_tile_ends_where_32_bit_key_offsets_would_overflow(hip_lib):
    env = dict(os.environ, YH_DEBUG_TUNING="1", YH_STREAM_WGS="1")
    r = subprocess.run([sys.executable, "-c", CHILD_WIDE_TILE % ROOT], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0 and "wide tile ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
