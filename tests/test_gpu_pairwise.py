"""`yacht train`'s pairwise counts (yh_pairwise.hip: reference-major records, one row in LDS per workgroup) against the
oracle: short, inline and whole-wave posting lists, several column blocks, handles with and without posting ranks,
row ranges, and more survivors than the first output buffer holds.

The column-block cases need a tuning variable (YH_DEBUG_TUNING=1 YH_PAIR_COLS=...), read when the library is first
used, so they run in a child process.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle
from yacht_amd import synth
from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY, train_select

rng = np.random.default_rng(int(sys.argv[2]))
mh = synth.max_hash_for_scaled(1000)
refs = synth.clustered_refs(rng, 60, (1.0, 0.9, 0.5, 0.25, 0.1), 300)
n0 = len(refs)
# posting lists of every kind: a hash in 5..32 sketches (walked by its lane), in 33..300 (by the wave), and in all
extra = [[] for _ in range(n0)]
for m in (5, 8, 9, 31, 32, 33, 64, 65, 200, n0):
    for _ in range(3):
        h = int(rng.integers(1, mh))
        for r in rng.choice(n0, size=m, replace=False):
            extra[int(r)].append(h)
refs = [np.unique(np.concatenate([r, np.array(e, np.uint64)])) for r, e in zip(refs, extra)]
refs.insert(7, np.zeros(0, np.uint64))
refs.append(np.array([3], np.uint64))
values, offsets = synth.pack(refs)
sizes = np.diff(offsets).astype(np.uint32)
n = len(refs)
out = {"n": n}
for c in (0.0, 0.95 ** 31, 0.9):
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c, threads=4)
    for flags in (0, YH_DB_PAIRWISE_ONLY):
        with RefDB(values, offsets, flags=flags) as db:
            gi, gj, gc = db.pairwise(c)
            ok = bool(np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc) and db.index_stats() == wstats)
            cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n, 3)]))
            parts = [db.pairwise(c, a, b) for a, b in zip(cuts, cuts[1:])]
            ok_rows = all(np.array_equal(np.concatenate([p[k] for p in parts]), w) for k, w in ((0, wi), (1, wj), (2, wc)))
            ok_sel = bool(np.array_equal(train_select(sizes, gi, gj), oracle.train_select(sizes, wi, wj)))
            if flags == YH_DB_PAIRWISE_ONLY:
                db.pairwise(c)
                out["sparse_rows"], out["dense_rows"] = db.pairwise_row_stats()
        out[f"c{c:.3f}_flags{flags}"] = [ok, bool(ok_rows), ok_sel, int(wi.size)]
print(json.dumps(out))
"""


def _run(seed, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(seed)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("cols", [None, 64, 128, "unfused", "lists", "sparse", "sparse_lists"])
def test_pairwise_rows_against_the_oracle(hip_lib, cols):
    """YH_DB_PAIRWISE_ONLY handles take the fused path (records written by the sort's last pass, yh_sort.hip), the others the
    posting arrays + k_pair_transpose; "unfused": the train handle without the fused path (posting ranks); "lists": every fused
    record in list form (what >= 2^21 - 1 references get)."""
    env = {} if cols is None else {"YH_DEBUG_TUNING": "1"}
    if cols == "unfused":
        env["YH_NO_FUSED_TRAIN"] = "1"
    elif cols == "lists":
        env["YH_FZ_NO_INLINE"] = "1"
    elif cols == "sparse":         # round 6: the row pass of large N (k_pair_rows_sparse: a hash table over the touched columns) forced onto this one
        env["YH_PAIR_SPARSE"] = "1"
    elif cols == "sparse_lists":
        env["YH_PAIR_SPARSE"] = "1"
        env["YH_FZ_NO_INLINE"] = "1"
    elif cols is not None:
        env["YH_PAIR_COLS"] = str(cols)
    out = _run(cols if isinstance(cols, int) else 5, env)
    if isinstance(cols, str) and cols.startswith("sparse"):
        assert out["sparse_rows"] > 300 and out["dense_rows"] == 0, out
    assert out["n"] > 300
    checks = {k: v for k, v in out.items() if k not in ("n", "sparse_rows", "dense_rows")}
    assert len(checks) == 6
    for k, v in checks.items():
        assert v[:3] == [True, True, True], (k, v)
    assert any(v[3] > 1000 for v in checks.values())  # (the lists shared by all sketches: a dense corner)


@pytest.mark.parametrize("shares", ["0.4,0.3,0.2,0.1", "0.5,0.5", "0.05,0.05,0.9", "0.25,0.25,0.25,0.125,0.125"])
def test_database_uploaded_in_chunks(hip_lib, shares):
    """yh_db_create on host arrays above a size bar uploads the references in chunks, each sorted and merged into the
    sorted prefix while the next one crosses the bus (yh_build_upload_sorted).  Forced onto a small database here; the
    merged (hash, reference) order is checked on the device (YH_CHECK_SORT) and every pair result against the oracle."""
    out = _run(11, {"YH_DEBUG_TUNING": "1", "YH_UPLOAD_CHUNK_MIN": "1", "YH_UPLOAD_SHARES": shares, "YH_CHECK_SORT": "1"})
    checks = {k: v for k, v in out.items() if k not in ("n", "sparse_rows", "dense_rows")}
    for k, v in checks.items():
        assert v[:3] == [True, True, True], (k, v)


@pytest.mark.parametrize("n", [1500, 400])
def test_more_survivors_than_the_first_buffer(hip_lib, n):
    """n sketches that all hold one common hash: n (n - 1) ordered pairs at C = 0, all behind the segments' fixed slots.
    1 500: more than the million entries the output starts with, so the row pass is repeated with the size it counted;
    400: more than the 65 536 that come back with the counts in the first copy, so the survivors take a second one."""
    from oracle import oracle
    from yacht_amd import synth
    from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY

    rng = np.random.default_rng(77)
    mh = synth.max_hash_for_scaled(1000)
    common = np.array([int(rng.integers(1, mh))], np.uint64)
    refs = [np.unique(np.concatenate([common, synth.random_sketch(rng, 20, mh)])) for _ in range(n)]
    values, offsets = synth.pack(refs)
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, 0.0, threads=4)
    assert wi.size == n * (n - 1)
    with RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY) as db:
        gi, gj, gc = db.pairwise(0.0)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
        assert db.index_stats() == wstats


SORT_WORKER = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle
from yacht_amd import synth
from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY

rng = np.random.default_rng(3)
mh = synth.max_hash_for_scaled(1000)
base = [synth.random_sketch(rng, 400, mh) for _ in range(2000)]
for a in range(0, 2000, 5):           # clusters: neighbours share half of a sketch
    base[a + 1] = np.unique(np.concatenate([base[a + 1], base[a][::2]]))
heavy = int(sys.argv[2])              # one hash held by this many references
if heavy:
    h = np.array([int(rng.integers(1, mh))], np.uint64)
    for r in rng.choice(2000, size=heavy, replace=False):
        base[int(r)] = np.unique(np.concatenate([base[int(r)], h]))
values, offsets = synth.pack(base)
wi, wj, wc, wstats = oracle.train_pairs(values, offsets, 0.2, threads=4)
ok = True
for flags in (YH_DB_PAIRWISE_ONLY, 0):
    with RefDB(values, offsets, flags=flags) as db:
        gi, gj, gc = db.pairwise(0.2)
        ok = ok and bool(np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc) and db.index_stats() == wstats)
print(json.dumps({"ok": ok, "pairs": int(wi.size)}))
"""


@pytest.mark.parametrize("mode", ["pieces", "two_level"])
@pytest.mark.parametrize("heavy,verdict", [(0, "taken"), (800, "taken"), (2000, "REFUSED")])
def test_distribution_sort_takes_uniform_keys_and_refuses_the_rest(hip_lib, heavy, verdict, mode):
    """yh_sort.hip on 2 000 sketches (8e5 hashes, ~310 buckets), the train handle and the full handle: uniform keys, a hash
    held by 800 references, a hash held by all 2 000 (it overflows its bucket).
    mode "pieces" (the default since round 5: regions read in place, overflowed buckets through a side list): all three are
    TAKEN by both handles -- the verdict lines "[yh pieces] ... -> taken", and "side list" says when it was used;
    mode "two_level" (YH_NO_PIECES=1 YH_NO_PIECES_SORT=1: round 4's two-level sort, still the path of inputs the piece geometry
    does not take): the overflowing one is REFUSED on the device and sorted by rocPRIM ("[yh sort] ... -> REFUSED").
    Either way the order is checked on the device (YH_CHECK_SORT) and pairs and statistics equal the oracle's."""
    env = dict(os.environ)
    env.update({"YH_DEBUG_TUNING": "1", "YH_TRACE_BUILD": "1", "YH_CHECK_SORT": "1"})
    if mode == "two_level":
        env.update({"YH_NO_PIECES": "1", "YH_NO_PIECES_SORT": "1"})
    r = subprocess.run([sys.executable, "-c", SORT_WORKER, ROOT, str(heavy)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["pairs"] > 300
    verdicts = [ln for ln in r.stderr.splitlines() if (ln.startswith("[yh sort]") or ln.startswith("[yh pieces]")) and " -> " in ln]
    pieces = [ln for ln in verdicts if ln.startswith("[yh pieces]")]
    sorts = [ln for ln in verdicts if ln.startswith("[yh sort]")]
    if mode == "pieces":
        assert len(pieces) == 2 and not sorts and all(ln.rstrip().endswith("taken") for ln in pieces), verdicts
        used_side_list = ["side list: " in ln or ("side list " in ln and " side list 0 pairs" not in ln) for ln in r.stderr.splitlines() if ln.startswith("[yh pieces]")]
        assert any(used_side_list) == (heavy == 2000), r.stderr[-1500:]
    else:
        assert not pieces and len(sorts) == 2 and all(ln.rstrip().endswith(verdict) for ln in sorts), verdicts


def test_tiny_sketches_and_shared_counts_on_the_train_handle(hip_lib):
    """6 000 sketches of 0..6 hashes drawn from a small pool (many share): every block of the position -> reference look-up
    of the fused path (one entry per 256 CSR positions) spans dozens of references, some of them empty.  Pairs, statistics and
    the per-reference shared-hash counts (counted from the records on demand) against the oracle / the full handle."""
    import torch

    from oracle import oracle
    from yacht_amd import synth
    from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY

    rng = np.random.default_rng(91)
    mh = synth.max_hash_for_scaled(1000)
    pool = np.unique(rng.integers(1, mh, size=9000, dtype=np.uint64))
    refs = [np.unique(rng.choice(pool, size=int(rng.integers(0, 7)), replace=False)) if rng.random() < 0.9 else np.zeros(0, np.uint64)
            for _ in range(6000)]
    values, offsets = synth.pack(refs)
    n = len(refs)
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, 0.3, threads=4)
    assert wi.size > 1000
    with RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY) as tdb, RefDB(values, offsets) as db:
        gi, gj, gc = tdb.pairwise(0.3)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
        assert tdb.index_stats() == wstats
        ns_t = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        ns_d = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        tdb.nshared_device(ns_t.data_ptr())
        tdb.synchronize()
        db.nshared_device(ns_d.data_ptr())
        db.synchronize()
        assert torch.equal(ns_t, ns_d) and int(ns_d.sum()) > 0


def test_neighbouring_hashes_crowd_one_slot_of_the_grouping_table(hip_lib):
    """1 200 CONSECUTIVE hash values, each held by two of 2 000 sketches: all of them start at the same slot of their bucket's
    hash table in the fused train path (k_bucket_group probes linearly from a linear function of the key) -- long probe
    sequences, same answers.  Next to them one hash held by 1 100 sketches: more holders than the counting sort of the other
    handles ranks (it refuses them and rocPRIM sorts), no limit for the grouping."""
    from oracle import oracle
    from yacht_amd import synth
    from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY

    rng = np.random.default_rng(123)
    mh = synth.max_hash_for_scaled(1000)
    refs = [synth.random_sketch(rng, 400, mh) for _ in range(2000)]
    h0 = int(rng.integers(mh // 4, mh // 2))
    extra = [[] for _ in range(2000)]
    for d in range(1200):
        for r in rng.choice(2000, size=2, replace=False):
            extra[int(r)].append(h0 + d)
    heavy = int(rng.integers(mh // 2, mh - 1))
    for r in rng.choice(2000, size=1100, replace=False):
        extra[int(r)].append(heavy)
    refs = [np.unique(np.concatenate([a, np.array(e, np.uint64)])) for a, e in zip(refs, extra)]
    values, offsets = synth.pack(refs)
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, 0.001, threads=4)
    assert wi.size > 1100 * 1099
    for flags in (YH_DB_PAIRWISE_ONLY, 0):
        with RefDB(values, offsets, flags=flags) as db:
            gi, gj, gc = db.pairwise(0.001)
            assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc), flags
            assert db.index_stats() == wstats


@pytest.mark.parametrize("size", [65535, 65536])
def test_sixteen_bit_row_counts_at_their_limit(hip_lib, size):
    """Three identical sketches of 65 535 hashes: every count of the row pass is 65 535 -- the largest a 16-bit count holds, in
    both halves of one LDS word (columns 0 and 1 of row 2), without a carry between them; with 65 536 hashes the handle
    must take the 32-bit rows.  Two smaller relatives keep the threshold filter in play."""
    from oracle import oracle
    from yacht_amd import synth
    from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY

    rng = np.random.default_rng(5)
    mh = synth.max_hash_for_scaled(1000)
    big = synth.random_sketch(rng, size + 64, mh)[:size]  # (exactly `size` distinct hashes)
    assert big.size == size
    refs = [big, big.copy(), big.copy(), big[::3].copy(), np.unique(np.concatenate([big[::7], synth.random_sketch(rng, 500, mh)]))]
    values, offsets = synth.pack(refs)
    for c in (0.0, 0.3):
        wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c, threads=4)
        with RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY) as db:
            gi, gj, gc = db.pairwise(c)
            assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc), (size, c)
            assert db.index_stats() == wstats
    assert int(wc.max()) == big.size


def test_database_created_from_its_packed_form(hip_lib):
    """yh_db_create_packed (ABI 7): the chunks of a packed CSR expanded in HBM under the upload -- the train handle and the full
    handle equal to those made from the plain arrays (pairs, statistics, run counts), a small database through the host-side
    unpacking, forged blobs refused."""
    from yacht_amd import _lib, synth
    from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY, csr_pack

    values, offsets = synth.config4(seed=77, n_clusters=520, size=5000)  # 1.3e7 hashes: above the chunked upload's bar
    assert values.size >= 12 << 20
    blob = csr_pack(values, offsets)
    c = 0.95 ** 31
    with RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY) as a, RefDB.from_packed(blob, flags=YH_DB_PAIRWISE_ONLY) as b:
        pa, pb = a.pairwise(c), b.pairwise(c)
        assert all(np.array_equal(x, y) for x, y in zip(pa, pb)) and pa[0].size > 1000
        assert a.index_stats() == b.index_stats()
        assert b.info()["n_hashes"] == values.size
    rng = np.random.default_rng(3)
    n = offsets.size - 1
    parts = [values[int(offsets[j]):int(offsets[j + 1])][::3] for j in rng.choice(n, size=20, replace=False)]
    sample = np.unique(np.concatenate(parts + [synth.random_sketch(rng, 20000, synth.max_hash_for_scaled(1000))]))
    with RefDB(values, offsets) as a, RefDB.from_packed(blob) as b:
        ra, rb = a.run_counts(sample), b.run_counts(sample)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb)) and int(ra[0].max()) > 0
    # forged: a block that points outside the payload; a block whose first hash lies below its predecessor's hashes
    first_block = 8 + (n + 1)
    bad = blob.copy()
    bad[first_block + 3 * 5 + 1] = np.uint64(2 ** 41)
    with pytest.raises(_lib.YachtHipError, match="packed CSR"):
        RefDB.from_packed(bad, flags=YH_DB_PAIRWISE_ONLY)
    bad = blob.copy()
    bad[first_block + 3 * 7] = np.uint64(1)  # (block 7 is not the first block of its sketch: sketches have ~20)
    with pytest.raises(_lib.YachtHipError, match="ascending"):
        RefDB.from_packed(bad, flags=YH_DB_PAIRWISE_ONLY)
    # a small database: unpacked on the host, created as ever
    sv, so = synth.config4(seed=5, n_clusters=30, size=200)
    with RefDB(sv, so, flags=YH_DB_PAIRWISE_ONLY) as a, RefDB.from_packed(csr_pack(sv, so), flags=YH_DB_PAIRWISE_ONLY) as b:
        assert all(np.array_equal(x, y) for x, y in zip(a.pairwise(0.1), b.pairwise(0.1)))


HOT_WORKER = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle
from yacht_amd import synth
from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY, train_select

rng = np.random.default_rng(606)
mh = synth.max_hash_for_scaled(1000)
n_hot, n = 2400, 2600
refs = [synth.random_sketch(rng, int(rng.integers(8, 30)), mh) for _ in range(n)]
hot = [int(rng.integers(1, mh)) for _ in range(3)]          # three hashes that the first 2 400 sketches all hold
for i in range(n_hot):
    refs[i] = np.unique(np.concatenate([refs[i], np.array(hot, np.uint64)]))
for i in range(n_hot, n - 1, 2):                             # ... and pairs of near-duplicates among the rest (sparse rows)
    refs[i + 1] = np.unique(np.concatenate([refs[i][: refs[i].size * 3 // 4], refs[i + 1][:4]]))
values, offsets = synth.pack(refs)
sizes = np.diff(offsets).astype(np.uint32)
out = {}
for c in (0.0, 0.2):
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c, threads=8)
    with RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY) as db:
        gi, gj, gc = db.pairwise(c)
        sp, de = db.pairwise_row_stats()
        part = db.pairwise(c, 100, 2500)
        keep = (wi >= 100) & (wi < 2500)
        ok_part = bool(np.array_equal(part[0], wi[keep]) and np.array_equal(part[1], wj[keep]) and np.array_equal(part[2], wc[keep]))
    out[f"c{c}"] = [bool(np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)), ok_part, int(wi.size), int(sp), int(de),
                    bool(np.array_equal(train_select(sizes, gi, gj), oracle.train_select(sizes, wi, wj)))]
print(json.dumps(out))
"""


@pytest.mark.parametrize("cols", [None, 1024])
def test_sparse_rows_hand_hot_rows_back_to_the_dense_pass(hip_lib, cols):
    """k_pair_rows_sparse keeps a row's counts in a hash table over the columns it touches (2 048 at most); a row that touches
    more -- here 2 400 sketches that all hold three common hashes: 2 399 columns each -- is handed back and takes the dense pass
    (k_pair_rows with a row list; `cols`: in several column blocks), the other rows stay sparse.  Pairs, row ranges and the
    selection against the oracle."""
    env = dict(os.environ)
    env.update({"YH_DEBUG_TUNING": "1", "YH_PAIR_SPARSE": "1"})
    if cols:
        env["YH_PAIR_COLS"] = str(cols)
    r = subprocess.run([sys.executable, "-c", HOT_WORKER, ROOT], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for k, v in out.items():
        assert v[0] and v[1] and v[5], (k, v)
        assert v[3] == 200 and v[4] == 2400, (k, v)   # (the last full-range call: 200 rows sparse, 2 400 handed back)
    assert out["c0.0"][2] > 2400 * 2399
