"""Host-side logic that needs no GPU: signature I/O, the mirrored helper functions, the C ABI's
symbol table, the loud failure without a device, and the vectorised hypothesis test."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

from yacht_amd import _lib, sigio
from yacht_amd import hypothesis_recovery_src as hr
from yacht_amd import utils
from yacht_amd.train_core import format_pair_line, row_ranges

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FX = os.path.join(GOLD, "fixtures")


# ---- C ABI ---------------------------------------------------------------------------------------------
def header_functions():
    text = open(os.path.join(ROOT, "include", "yacht_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yh_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    """libyacht_hip.so loads (also without a GPU) and exports exactly the header's entry points."""
    names = header_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.lib_path())
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/yacht_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert _lib.load().yh_abi_version() == 8


def test_no_device_fails_loudly():
    """No CPU fallback: without a HIP device every handle creation is an error, never a result."""
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    from yacht_amd.engine import RefDB

    with pytest.raises(_lib.YachtHipError) as ei:
        RefDB(np.array([1, 2, 3], np.uint64), np.array([0, 3], np.uint64))
    assert ei.value.code == _lib.YH_ERR_NO_DEVICE


def test_train_select_host_entry_matches_oracle_on_goldens():
    """yh_train_select is host code: it runs here and must reproduce the reference's walk."""
    from oracle import oracle
    from yacht_amd.engine import train_select

    cases = json.load(open(os.path.join(GOLD, "golden_train.json")))
    arrays = np.load(os.path.join(GOLD, "golden_train.npz"))
    for c in cases:
        values, offsets = arrays[c["tag"] + "_values"], arrays[c["tag"] + "_offsets"]
        sizes = np.diff(offsets).astype(np.uint32)
        pi, pj, _pc, _ = oracle.train_pairs(values, offsets, c["c"], threads=2)
        assert train_select(sizes, pi, pj).tolist() == c["selected"], c["tag"]


# ---- signature I/O --------------------------------------------------------------------------------------
def test_signature_metadata_known_answers():
    """tests/unittests_data/test_collect_signature_info_data.json of the reference:
    name -> (md5sum, mean abundance, sketch size, scaled) for the 20 fixture genomes."""
    want = json.load(open(os.path.join(FX, "test_collect_signature_info_data.json")))
    sigs = sigio.load_file_as_signatures(os.path.join(FX, "20_genomes_sketches.zip"), ksize=31)
    assert len(sigs) == 20
    got = {s.name: [s.md5sum(), s.minhash.mean_abundance, len(s.minhash), s.minhash.scaled] for s in sigs}
    assert got == want


def test_sample_and_empty_fixtures(tmp_path):
    s = utils.load_signature_with_ksize(os.path.join(FX, "sample.sig.zip"), 31)
    assert len(s.minhash) == 49821 and s.minhash.scaled == 1000
    assert s.minhash.mean_abundance == pytest.approx(2.4032636839886794, rel=1e-15)
    assert sigio.zip_has_manifest(os.path.join(FX, "sample.sig.zip"))
    with pytest.raises(ValueError, match="Empty sketch in signature"):   # reference bug YAC-13 fixture
        utils.load_signature_with_ksize(os.path.join(FX, "extract_empty_hash.sig.zip"), 31)
    with pytest.raises(ValueError, match="Expected exactly one signature with ksize 21"):
        utils.load_signature_with_ksize(os.path.join(FX, "sample.sig.zip"), 21)


def test_sig_roundtrip_and_md5(tmp_path):
    rng = np.random.default_rng(0)
    mins = np.unique(rng.integers(0, sigio.max_hash_for_scaled(1000), 500, dtype=np.uint64))
    sig = sigio.make_signature(mins, 31, 1000, name="g1", filename="g1.fa", abundances=np.arange(mins.size) % 3 + 1)
    z = str(tmp_path / "db.sig.zip")
    sigio.write_sig_zip([sig], z)
    back = sigio.load_file_as_signatures(z, ksize=31)[0]
    assert back.name == "g1" and np.array_equal(back.minhash.mins, mins) and back.md5sum() == sig.md5sum()
    assert back.minhash.scaled == 1000 and sigio.zip_has_manifest(z)
    plain = str(tmp_path / "g1.sig")
    sigio.write_sig(sig, plain)
    assert np.array_equal(sigio.read_mins_first_signature(plain), mins)
    assert sigio.read_mins_first_signature(str(tmp_path / "missing.sig")).size == 0


def test_get_num_kmers_reference_values():
    assert utils.get_num_kmers(2.5, 100, 10000) == 2_500_000          # reference tests/test_utils.py
    assert utils.get_num_kmers(None, 100, 10, True) == 1000
    assert utils.get_num_kmers(1.0036337209302326, 1376, 1000, False) == 1381


# ---- train-core file contract ---------------------------------------------------------------------------
def test_pair_line_format_matches_iostream():
    assert format_pair_line(0, 1, 250, 1000, 2000) == "0,1,0.0909091,0.25,0.125"
    assert format_pair_line(2, 3, 500, 500, 500) == "2,3,1,1,1"
    assert format_pair_line(4, 2, 100, 100, 500) == "4,2,0.2,1,0.2"


def test_row_ranges_match_reference_split():
    assert row_ranges(7, 3, 1) == [(0, 0, 0, 2), (0, 1, 2, 4), (0, 2, 4, 7)]
    assert row_ranges(3, 8, 1)[-1] == (0, 7, 0, 3) and row_ranges(3, 8, 1)[0] == (0, 0, 0, 0)
    r = row_ranges(10, 2, 3)  # ceil(10/3)=4 rows per pass
    assert [(p, a, b) for p, _t, a, b in r] == [(0, 0, 2), (0, 2, 4), (1, 4, 6), (1, 6, 8), (2, 8, 9), (2, 9, 10)]


# ---- hypothesis test --------------------------------------------------------------------------------------
def test_hyp_test_batch_equals_scalar_and_golden():
    g = json.load(open(os.path.join(GOLD, "golden_hyp.json")))["single_hyp_test"]
    groups = {}
    for row in g:
        groups.setdefault((row["k"], row["sig"], row["ani"], row["cov"]), []).append(row)
    for (k, sig, ani, cov), rows in groups.items():
        e = [r["e"] for r in rows]
        m = [r["m"] for r in rows]
        cols = hr.hyp_test_batch(e, m, k, sig, ani, cov)
        for i, row in enumerate(rows):
            w = row["out"]
            s = hr.single_hyp_test((row["e"], row["m"]), k, sig, ani, cov)
            assert bool(cols[0][i]) == bool(s[0]) == w[0]
            assert int(cols[3][i]) == s[3] == w[3] and float(cols[5][i]) == float(s[5]) == w[5]
            for c, sv, wv in ((cols[1][i], s[1], w[1]), (cols[6][i], s[6], w[6]), (cols[7][i], s[7], w[7])):
                assert float(c) == pytest.approx(wv, rel=1e-12, abs=1e-15)
                assert float(sv) == pytest.approx(wv, rel=1e-12, abs=1e-15)


def test_single_hyp_test_return_types():
    """Shape/type contract asserted by the reference's tests (test_hypothesis_recovery_src.py:77-81)."""
    r = hr.single_hyp_test((100, 50), 31, 0.99, 0.95, 1)
    assert len(r) == 8 and isinstance(r[0], (bool, np.bool_)) and isinstance(r[2], int) and isinstance(r[3], int)
    assert hr.single_hyp_test((2829, 0), 31, 0.99, 0.95, 0.2)[3] == 565   # int(e * cov) truncates


# ---- drivers ------------------------------------------------------------------------------------------------
def test_coverage_plan_matches_reference_rules():
    from yacht_amd.run_YACHT import coverage_plan

    assert coverage_plan([1, 0.5, 0.1, 0.05, 0.01]) == ([1.0, 0.5, 0.1, 0.05, 0.01], True)
    assert coverage_plan([0.001]) == ([1.0, 0.001], False)           # 1.0 is forced in front
    assert coverage_plan([0.1, 0.5, 0.1]) == ([1.0, 0.5, 0.1], False)  # de-duplicated, descending


def test_cli_parsers_and_error_paths(tmp_path):
    from yacht_amd import cli, make_training_data_from_sketches

    p = cli.build_parser()
    a = p.parse_args(["train", "--ref_file", "x.zip", "--ksize", "31"])
    assert (a.num_threads, a.ani_thresh, a.prefix, a.force) == (16, 0.95, "yacht", False)
    r = p.parse_args(["run", "--json", "c.json", "--sample_file", "s.sig.zip"])
    assert r.min_coverage_list == [1, 0.5, 0.1, 0.05, 0.01] and r.significance == 0.99
    assert cli.main(["download", "demo"]) == 2
    k = p.parse_args(["sketch", "ref", "--infile", "g.fa", "--outfile", "o.zip"])
    assert (k.kmer, k.scaled) == (31, 1000)
    # reference behaviour (tests/test_make_training_data_from_sketches.py): a non-zip reference file is rejected
    bad = p.parse_args(["train", "--ref_file", str(tmp_path / "refs.sig"), "--ksize", "31", "--outdir", str(tmp_path)])
    with pytest.raises(ValueError, match="is not a zip file"):
        make_training_data_from_sketches.main(bad)
    missing = p.parse_args(["run", "--json", str(tmp_path / "nope.json"), "--sample_file", "s.sig.zip",
                            "--outdir", str(tmp_path)])
    with pytest.raises(ValueError, match="does not exist"):
        missing.func(missing)


def test_sig_batch_reader_matches_the_python_reader(tmp_path):
    """yh_sig_batch_* (host threads, no GPU needed): record 0 / signature 0 / "mins" of every file, an
    unreadable file is an empty sketch, unsorted writers are sorted and de-duplicated --
    the same sketches as the json-module reader."""
    import json

    from yacht_amd import train_core

    rng = np.random.default_rng(5)
    paths, want = [], []
    for i in range(200):
        m = np.unique(rng.integers(0, 2 ** 64 - 1, size=int(rng.integers(0, 400)), dtype=np.uint64))
        p = tmp_path / f"s{i}.sig"
        rec = [{"class": "sourmash_signature", "name": f"g{i}", "signatures": [
            {"num": 0, "ksize": 31, "seed": 42, "max_hash": 18446744073709552, "mins": [int(x) for x in m],
             "abundances": [1] * len(m), "molecule": "dna"}], "version": 0.4}]
        p.write_text(json.dumps(rec) if i % 2 else json.dumps(rec, indent=2))
        paths.append(str(p))
        want.append(m)
    (tmp_path / "unsorted.sig").write_text('[{"signatures":[{"mins":[5, 3, 3, 9]}]}]')
    paths.append(str(tmp_path / "unsorted.sig"))
    want.append(np.array([3, 5, 9], dtype=np.uint64))
    paths.append(str(tmp_path / "missing.sig"))
    values, offsets = train_core.read_sketches_csr(paths, threads=3)
    assert offsets.size == len(paths) + 1 and int(offsets[-1]) == values.size
    for i, m in enumerate(want):
        assert np.array_equal(values[int(offsets[i]):int(offsets[i + 1])], m)
    py = train_core.read_sketches(paths[:len(want) - 1], 1)
    assert all(np.array_equal(a, b) for a, b in zip(py, want))
    assert int(offsets[-1]) - int(offsets[-2]) == 0  # missing file: empty sketch
    # round 6: the same files straight into the packed form (yh_sig_batch_pack: no CSR in between) -- byte for byte what packing
    # the CSR gives, for any number of packing threads; and its rows cut out again (yh_csr_subset)
    from yacht_amd.engine import csr_pack, csr_subset, csr_unpack

    for t in (1, 3):
        packed, poff = train_core.read_sketches_packed(paths, threads=t)
        assert np.array_equal(poff, offsets) and np.array_equal(packed, csr_pack(values, offsets))
    rows = [200, 0, 199, 201, 7]
    sv, so = csr_unpack(csr_subset(packed, rows))
    assert np.array_equal(so, np.concatenate([[0], np.cumsum([int(offsets[r + 1] - offsets[r]) for r in rows])]).astype(np.uint64))
    assert np.array_equal(sv, np.concatenate([values[int(offsets[r]):int(offsets[r + 1])] for r in rows]))


def test_batch_reader_reports_missing_and_malformed_files(tmp_path, capfd):
    """The reference's train core goes on with an empty sketch (and a message) when a file cannot be opened
    and dies when one does not parse (src/cpp/main.cpp:62-84); the threaded reader says which is which."""
    from yacht_amd import sigio, train_core

    good = tmp_path / "good.sig"
    sigio.write_sig(sigio.make_signature(np.array([5, 9, 11], np.uint64), name="g"), str(good))
    missing = tmp_path / "missing.sig"
    values, offsets = train_core.read_sketches_csr([str(good), str(missing), str(good)], threads=2)
    assert offsets.tolist() == [0, 3, 3, 6] and values.tolist() == [5, 9, 11, 5, 9, 11]
    assert "Could not open the file!" in capfd.readouterr().err
    bad = tmp_path / "bad.sig"
    bad.write_text('[{"signatures": [{"mins": [1, 2, oops]}]}]')
    with pytest.raises(ValueError, match="could not be parsed"):
        train_core.read_sketches_csr([str(good), str(bad)], threads=1)
    cut = tmp_path / "cut.sig"
    cut.write_text('[{"signatures":[{"mins":[1, 2')
    with pytest.raises(ValueError, match="could not be parsed"):
        train_core.read_sketches_csr([str(cut)], threads=1)
    nokey = tmp_path / "nokey.sig"
    nokey.write_text('[{"signatures": [{"ksize": 31}]}]')
    with pytest.raises(ValueError, match="could not be parsed"):
        train_core.read_sketches_csr([str(nokey)], threads=1)


def test_native_metadata_pass_equals_reference_known_answers_and_python_reader(tmp_path):
    """utils.collect_signature_info / decompress_all_sig_files on the library's threaded reader (yh_gunzip_files,
    yh_sig_meta_*): the reference's own known answers for the 20 fixture genomes
    (tests/unittests_data/test_collect_signature_info_data.json), and record-for-record the same as the Python
    reader (get_info_from_single_sig) on files that exercise the scanner: escapes and non-ASCII in names, "name"
    behind "signatures", several k-mer sizes in one file, no abundances, null abundances, an empty sketch, a file
    without that k-mer size, a truncated file, unsorted mins (deferred to the Python reader), a .sig.gz left
    compressed."""
    import gzip as gz
    import zipfile

    work = tmp_path / "work"
    (work / "signatures").mkdir(parents=True)
    with zipfile.ZipFile(os.path.join(FX, "20_genomes_sketches.zip")) as z:
        z.extractall(work)
    packed = sorted(str(p) for p in (work / "signatures").glob("*.sig.gz"))
    assert len(packed) == 20
    utils.decompress_all_sig_files(packed, 4)
    assert not list((work / "signatures").glob("*.gz")) and len(list((work / "signatures").glob("*.sig"))) == 20
    want = json.load(open(os.path.join(FX, "test_collect_signature_info_data.json")))
    info = utils.collect_signature_info(4, 31, str(work))
    assert {k: [v[0], v[1], v[2], v[3]] for k, v in info.items()} == want
    assert all(os.path.exists(v[4]) for v in info.values())

    odd = tmp_path / "odd"
    (odd / "signatures").mkdir(parents=True)

    def sig(ksize, mins, ab="omit", max_hash=18446744073709552):
        s = {"num": 0, "ksize": ksize, "seed": 42, "max_hash": max_hash, "mins": mins, "md5sum": "x", "molecule": "dna"}
        if ab != "omit":
            s["abundances"] = ab
        return s

    files = {
        "escapes.sig": json.dumps([{"class": "sourmash_signature", "name": 'E. coli "K-12" \\ tab\there é中\U0001f9ec', "filename": "f",
                                    "signatures": [sig(31, [3, 9, 27], [2, 3, 4])], "version": 0.4}]),
        "name_last.sig": '[{"signatures":[' + json.dumps(sig(31, [5, 6])) + '],"filename":"x","name":"behind the signatures"}]',
        "two_ksizes.sig": json.dumps([{"name": "two k", "signatures": [sig(21, [1, 2, 3], [1, 1, 1]), sig(31, [10, 20, 30, 40], [5, 1, 1, 1]),
                                                                       sig(51, [7])]}]),
        "null_ab.sig": json.dumps([{"name": "null abundances", "signatures": [sig(31, [11, 12, 13], None)]}]),
        "scaled100.sig": json.dumps([{"name": "scaled 100", "signatures": [sig(31, [1, 2], [1, 2], 184467440737095520)]}]),
        "empty.sig": json.dumps([{"name": "empty sketch", "signatures": [sig(31, [], [])]}]),
        "other_k.sig": json.dumps([{"name": "no k31", "signatures": [sig(21, [1, 2])]}]),
        "twice.sig": json.dumps([{"name": "two of k31", "signatures": [sig(31, [1]), sig(31, [2])]}]),
        "cut.sig": '[{"name":"cut","signatures":[{"ksize":31,"mins":[1,2',
        "unsorted.sig": json.dumps([{"name": "unsorted mins", "signatures": [sig(31, [9, 3, 3, 5], [1, 2, 2, 7])]}]),
    }
    for name, text in files.items():
        (odd / "signatures" / name).write_text(text, encoding="utf-8")
    with gz.open(odd / "signatures" / "still_packed.sig.gz", "wt", encoding="utf-8") as f:
        f.write(json.dumps([{"name": "read through gzip", "signatures": [sig(31, [100, 200], [4, 6])]}]))
    got = utils.collect_signature_info(3, 31, str(odd))
    want = {}
    for f in os.listdir(odd / "signatures"):
        rec = utils.get_info_from_single_sig(str(odd / "signatures" / f), 31)
        if rec:
            want[rec[1]] = (rec[2], rec[3], rec[4], rec[5], rec[0])
    assert got == want
    assert set(got) == {'E. coli "K-12" \\ tab\there é中\U0001f9ec', "behind the signatures", "two k", "null abundances",
                        "scaled 100", "unsorted mins", "read through gzip"}
    assert got["two k"][1:4] == (2.0, 4, 1000) and got["null abundances"][1] is None and got["scaled 100"][3] == 100
    assert got["unsorted mins"][2] == 3


def test_metadata_pass_hands_its_sketches_to_the_core(tmp_path):
    """`yacht train` reads its signature files once: utils.collect_signature_info keeps what the train core reads from
    each file (yh_sig_meta_read_keep / yh_sig_meta_take_batch) and train_core.read_sketches_csr over the SAME path
    list takes it -- equal to reading the files, statuses included; any other list reads the files."""
    import zipfile

    from yacht_amd import train_core

    work = tmp_path / "work"
    (work / "signatures").mkdir(parents=True)
    with zipfile.ZipFile(os.path.join(FX, "20_genomes_sketches.zip")) as z:
        z.extractall(work)
    utils.decompress_all_sig_files(sorted(str(p) for p in (work / "signatures").glob("*.sig.gz")), 2)
    (work / "signatures" / "zz_two_ksizes.sig").write_text(json.dumps([{"name": "two", "signatures": [
        {"num": 0, "ksize": 21, "seed": 42, "max_hash": 18446744073709552, "mins": [5, 7, 9], "molecule": "dna"},
        {"num": 0, "ksize": 31, "seed": 42, "max_hash": 18446744073709552, "mins": [1, 2], "molecule": "dna"}]}]))
    paths = [os.path.join(str(work), "signatures", f) for f in os.listdir(work / "signatures")]
    train_core.drop_parsed_sketches()
    want_v, want_o = train_core.read_sketches_csr(paths, threads=2)  # from the files
    info = utils.collect_signature_info(3, 31, str(work))
    assert len(info) == 21 and train_core._PARSED.get("paths") == paths
    got_v, got_o = train_core.read_sketches_csr(paths, threads=2)     # from the metadata pass
    assert not train_core._PARSED and np.array_equal(got_o, want_o) and np.array_equal(got_v, want_v)
    j = paths.index(os.path.join(str(work), "signatures", "zz_two_ksizes.sig"))
    assert got_v[int(got_o[j]):int(got_o[j + 1])].tolist() == [5, 7, 9]  # the core takes signature 0 whatever its k-mer size
    # another list (here: reversed) is read from the files, and the offer is dropped
    utils.collect_signature_info(3, 31, str(work))
    rev_v, rev_o = train_core.read_sketches_csr(paths[::-1], threads=2)
    assert not train_core._PARSED and int(rev_o[-1]) == int(want_o[-1])
    assert np.array_equal(rev_v[: int(rev_o[1])], want_v[int(want_o[-2]):])


def test_extract_members_concatenated_gzip_and_traversal(tmp_path):
    """A .sig.gz of several gzip members inflates whole (as gzip / yh_gunzip_files read it); a member -- file OR
    directory entry -- that would land outside the working directory is refused."""
    import gzip
    import zipfile

    from yacht_amd.make_training_data_from_sketches import _extract_members

    z = tmp_path / "a.zip"
    work = tmp_path / "work"
    work.mkdir()
    with zipfile.ZipFile(z, "w") as a:
        a.writestr("signatures/", b"")
        a.writestr("signatures/x.sig.gz", gzip.compress(b"[1,") + gzip.compress(b"2]"))
        a.writestr("signatures/broken.sig.gz", b"not gzip")
    assert _extract_members((str(z), str(work), ["signatures/", "signatures/x.sig.gz", "signatures/broken.sig.gz"])) == 3
    assert (work / "signatures" / "x.sig").read_bytes() == b"[1,2]"
    assert (work / "signatures" / "broken.sig.gz").read_bytes() == b"not gzip"  # left for the gunzip pass to complain about
    for bad in ("../../evil/", "../evil.sig"):
        with zipfile.ZipFile(z, "w") as a:
            a.writestr(bad, b"")
        with pytest.raises(ValueError, match="outside the working directory"):
            _extract_members((str(z), str(work), [bad]))


def test_zip_ingest_one_pass_equals_the_three_passes(tmp_path):
    """utils.ingest_zip_database (yh_zip_sig_ingest: central directory -> host threads pread, inflate, parse and write the
    members) against the reference's three passes as this build had them (unzip, gunzip, metadata): the reference's own
    known answers for the 20 fixture genomes, the same files on disk byte for byte, the sketches the train core reads from
    them, the archive's member order -- on the fixture (stored members) and on a re-packed archive with DEFLATED members,
    zip64 records (more than 65 535 members), a member that is not a signature and one that does not inflate."""
    import gzip as gz
    import zipfile

    from yacht_amd import train_core

    want = json.load(open(os.path.join(FX, "test_collect_signature_info_data.json")))
    src = os.path.join(FX, "20_genomes_sketches.zip")
    # the three passes, into `ref`
    ref = tmp_path / "ref"
    with zipfile.ZipFile(src) as z:
        z.extractall(ref)
        order = [n for n in z.namelist() if n.startswith("signatures/") and n.endswith(".sig.gz")]
    utils.decompress_all_sig_files(sorted(str(p) for p in (ref / "signatures").glob("*.sig.gz")), 4)
    one = tmp_path / "one"
    info = utils.ingest_zip_database(src, str(one), 31, 4)
    assert {k: [v[0], v[1], v[2], v[3]] for k, v in info.items()} == want
    assert [os.path.relpath(v[4], one) for v in info.values()] == [n[:-3] for n in order]  # the archive's order
    for p in sorted((ref / "signatures").glob("*.sig")):
        assert (one / "signatures" / p.name).read_bytes() == p.read_bytes()
    assert (one / "SOURMASH-MANIFEST.csv").read_bytes() == (ref / "SOURMASH-MANIFEST.csv").read_bytes()
    # the sketches were kept for the core: the next read of exactly these paths takes them (no file is opened)
    paths = train_core.parsed_paths()
    assert paths == [v[4] for v in info.values()]
    values, offsets = train_core.read_sketches_csr(paths, threads=2)
    assert train_core.parsed_paths() is None
    v2, o2 = train_core.read_sketches_csr(paths, threads=2)  # (from the files this time)
    assert np.array_equal(values, v2) and np.array_equal(offsets, o2) and offsets.size == 21
    # the files written in the BACKGROUND while the archive is read without writing (what `yacht train` does)
    bg_dir = tmp_path / "bg"
    bg = utils.BackgroundExtraction(src, str(bg_dir), 3)
    info_bg = utils.ingest_zip_database(src, str(bg_dir), 31, 4, write_files=False, background=bg)
    train_core.drop_parsed_sketches()
    assert bg.wait() == 21 and bg.wait() == 0  # (20 signatures + the manifest; a second wait is a no-op)
    assert {k: v[:4] for k, v in info_bg.items()} == {k: v[:4] for k, v in info.items()}
    for p_ in sorted((ref / "signatures").glob("*.sig")):
        assert (bg_dir / "signatures" / p_.name).read_bytes() == p_.read_bytes()
    assert (bg_dir / "SOURMASH-MANIFEST.csv").read_bytes() == (ref / "SOURMASH-MANIFEST.csv").read_bytes()
    with pytest.raises(_lib.YachtHipError):
        utils.BackgroundExtraction(str(tmp_path / "missing.zip"), str(tmp_path / "w5"), 2)
    # nothing written: the same records, no directory
    lean = tmp_path / "lean"
    info2 = utils.ingest_zip_database(src, str(lean), 31, 4, write_files=False)
    train_core.drop_parsed_sketches()
    assert {k: v[:4] for k, v in info2.items()} == {k: v[:4] for k, v in info.items()} and not lean.exists()

    # a harder archive: deflated members, > 65 535 entries (zip64 end records), zip64 extra fields, odd members
    big = tmp_path / "big.zip"
    with zipfile.ZipFile(src) as z, zipfile.ZipFile(big, "w", compression=zipfile.ZIP_DEFLATED) as out:
        out.writestr("SOURMASH-MANIFEST.csv", z.read("SOURMASH-MANIFEST.csv"))
        for i, n in enumerate(order):
            data = z.read(n)
            if i % 3 == 0:   # stored, as sourmash writes them
                out.writestr(zipfile.ZipInfo(n), data, compress_type=zipfile.ZIP_STORED)
            elif i % 3 == 1:  # deflated on top of the gzip
                out.writestr(n, data)
            else:             # a plain .sig member, deflated, written through the zip64 code path
                with out.open(zipfile.ZipInfo(n[:-3]), "w", force_zip64=True) as f:
                    f.write(gz.decompress(data))
        out.writestr("signatures/broken.sig.gz", b"\x1f\x8bnot gzip at all")
        out.writestr("notes/readme.txt", "not a signature")
        for i in range(66_000):
            out.writestr(zipfile.ZipInfo(f"filler/{i}"), b"")
    two = tmp_path / "two"
    info3 = utils.ingest_zip_database(str(big), str(two), 31, 4)
    train_core.drop_parsed_sketches()
    assert {k: [v[0], v[1], v[2], v[3]] for k, v in info3.items()} == want
    assert (two / "notes" / "readme.txt").read_text() == "not a signature"
    assert (two / "signatures" / "broken.sig.gz").exists()  # (does not inflate: left as it is, and reported, not fatal)
    assert len(list((two / "filler").iterdir())) == 66_000
    for p in sorted((ref / "signatures").glob("*.sig")):
        assert (two / "signatures" / p.name).read_bytes() == p.read_bytes()
    # members may not leave the working directory
    evil = tmp_path / "evil.zip"
    with zipfile.ZipFile(evil, "w") as out:
        out.writestr("../outside.sig", "[]")
    with pytest.raises(_lib.YachtHipError, match="outside the working directory"):
        utils.ingest_zip_database(str(evil), str(tmp_path / "w3"), 31, 2)
    assert not (tmp_path / "outside.sig").exists()
    with pytest.raises(_lib.YachtHipError):
        utils.ingest_zip_database(os.path.join(FX, "test_collect_signature_info_data.json"), str(tmp_path / "w4"), 31, 2)


def test_zip_ingest_refuses_corrupt_archives_without_taking_the_process_down(tmp_path):
    """ADVICE r04 on the native archive reader: (1) a member whose bytes no longer match the CRC-32 of the central directory
    is refused (Python's zipfile raises BadZipFile there) although it still inflates; (2) end records that claim 2^60
    entries / a central directory larger than the file come back as an error code -- no bad_alloc through extern "C", no
    std::terminate in the extraction thread; (3) an archive with bytes put in front of it (every stated offset shifted) is
    read like Python's zipfile reads it; (4) a compression method the reader does not know is an error code -- and
    `yacht train` then reads the archive with Python's zipfile by itself."""
    import struct
    import subprocess
    import zipfile

    from yacht_amd import train_core

    src = os.path.join(FX, "20_genomes_sketches.zip")
    want = json.load(open(os.path.join(FX, "test_collect_signature_info_data.json")))
    raw = bytearray(open(src, "rb").read())
    with zipfile.ZipFile(src) as z:
        infos = z.infolist()
    victim = next(i for i in infos if i.filename.endswith(".sig.gz"))
    # (1) one payload byte of a stored member flipped, inside the gzip stream's 8-byte trailer-free part: pick a byte of
    # the deflate data whose change still inflates for SOME members -- whatever it does, the CRC of the zip member differs
    data_off = victim.header_offset + 30 + len(victim.filename.encode()) + struct.unpack("<H", raw[victim.header_offset + 28: victim.header_offset + 30])[0]
    bad = bytearray(raw)
    bad[data_off + victim.compress_size // 2] ^= 0x01
    p1 = tmp_path / "crc.zip"
    p1.write_bytes(bad)
    with pytest.raises(zipfile.BadZipFile):
        with zipfile.ZipFile(p1) as z:
            z.read(victim.filename)
    with pytest.raises(_lib.YachtHipError):
        utils.ingest_zip_database(str(p1), str(tmp_path / "w1"), 31, 3, write_files=False)
    train_core.drop_parsed_sketches()
    bg = utils.BackgroundExtraction(str(p1), str(tmp_path / "w1b"), 2)
    with pytest.raises(_lib.YachtHipError):
        bg.wait()
    # (2) forged end records.  Plain end record: entry count / directory size / offset fields
    eocd = raw.rfind(b"PK\x05\x06")
    for field_off, value in ((12, 0x7fffffff), (16, 0x7ffffff0)):
        f = bytearray(raw)
        f[eocd + field_off: eocd + field_off + 4] = struct.pack("<I", value)
        q = tmp_path / f"forged_{field_off}.zip"
        q.write_bytes(f)
        with pytest.raises(_lib.YachtHipError):
            utils.ingest_zip_database(str(q), str(tmp_path / "wf"), 31, 2, write_files=False)
        train_core.drop_parsed_sketches()
    # ... and a zip64 pair that claims 2^60 entries in a 2^61-byte directory (built by hand behind the real directory)
    cd_size, cd_off = struct.unpack("<II", raw[eocd + 12: eocd + 20])
    z64 = struct.pack("<IQHHIIQQQQ", 0x06064b50, 44, 45, 45, 0, 0, 1 << 60, 1 << 60, 1 << 61, cd_off)
    loc = struct.pack("<IIQI", 0x07064b50, 0, cd_off + cd_size, 1)
    end = struct.pack("<IHHHHIIH", 0x06054b50, 0, 0, 0xffff, 0xffff, 0xffffffff, 0xffffffff, 0)
    q = tmp_path / "forged_zip64.zip"
    q.write_bytes(bytes(raw[:eocd]) + z64 + loc + end)
    # (in a child process: what used to happen here was std::terminate -- the interpreter gone, not an exception)
    code = ("import sys; sys.path.insert(0, %r)\nfrom yacht_amd import utils, _lib\n"
            "for how in ('ingest', 'extract'):\n"
            "    try:\n"
            "        if how == 'ingest': utils.ingest_zip_database(%r, %r, 31, 2, write_files=False)\n"
            "        else: utils.BackgroundExtraction(%r, %r, 2).wait()\n"
            "        print(how, 'accepted')\n"
            "    except _lib.YachtHipError as e:\n"
            "        print(how, 'refused')\n" % (ROOT, str(q), str(tmp_path / "wz"), str(q), str(tmp_path / "wz2")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == ["ingest", "refused", "extract", "refused"], r.stdout + r.stderr[-2000:]
    # (3) 1 000 bytes in front of the archive: Python reads it (offsets are taken relative to the directory's real place)
    pre = tmp_path / "prefixed.zip"
    pre.write_bytes(b"#!stub\n" + b"x" * 993 + bytes(raw))
    with zipfile.ZipFile(pre) as z:
        assert z.testzip() is None
    info = utils.ingest_zip_database(str(pre), str(tmp_path / "w3"), 31, 3, write_files=False)
    train_core.drop_parsed_sketches()
    assert {k: [v[0], v[1], v[2], v[3]] for k, v in info.items()} == want
    # (4) a method the native reader does not know (bzip2): an error code from the library ...
    odd = tmp_path / "bz.zip"
    with zipfile.ZipFile(src) as z, zipfile.ZipFile(odd, "w") as out:
        for i in z.infolist():
            out.writestr(i.filename, z.read(i.filename), compress_type=zipfile.ZIP_BZIP2 if i.filename.endswith(".sig.gz") else zipfile.ZIP_STORED)
    with pytest.raises(_lib.YachtHipError):
        utils.ingest_zip_database(str(odd), str(tmp_path / "w4"), 31, 2, write_files=False)
    train_core.drop_parsed_sketches()
    # ... and a stale offer of another working directory is not taken for this one's file list (utils.py:364)
    utils.ingest_zip_database(src, str(tmp_path / "elsewhere"), 31, 2)
    assert train_core.parsed_paths() is not None
    wd = tmp_path / "here"
    (wd / "signatures").mkdir(parents=True)
    listed = []
    orig_run = train_core.run
    try:
        def fake_run(file_list, *a, **k):
            listed.extend(open(file_list).read().split())
            raise RuntimeError("stop here")
        train_core.run = fake_run
        (wd / "signatures" / "only.sig").write_text("[]")
        with pytest.raises(ValueError, match="stop here"):
            utils.run_yacht_train_core(1, 0.95, 31, str(wd), {})
    finally:
        train_core.run = orig_run
    assert listed == [str(wd / "signatures" / "only.sig")] and train_core.parsed_paths() is None


def test_sample_archive_through_the_native_scanner_equals_the_python_reader(tmp_path):
    """utils.load_signature_with_ksize reads a one-signature archive -- a sample -- through the library's scanner
    (yh_zip_sig_ingest): the same name, md5, sizes, mins, mean abundance and (lazily) abundances as the general reader,
    which still handles every other shape: several signature files, several k-mer sizes with the wanted one not first,
    no signature of that size, an empty one (the reference's messages)."""
    import json as js

    from yacht_amd import sigio

    f = os.path.join(FX, "sample.sig.zip")
    a = utils.load_signature_with_ksize(f, 31)
    b = sigio.load_file_as_signatures(f, ksize=31)[0]
    assert type(a.minhash).__name__ == "_NativeMinHash"
    assert a.name == b.name and a.md5sum() == b.md5sum() and len(a.minhash) == len(b.minhash) == 49821
    assert a.minhash.scaled == b.minhash.scaled == 1000 and a.minhash.max_hash == b.minhash.max_hash
    assert a.minhash.mean_abundance == b.minhash.mean_abundance and a.minhash.track_abundance
    assert np.array_equal(a.minhash.mins, b.minhash.mins) and np.array_equal(a.minhash.abundances, b.minhash.abundances)
    assert a.minhash.hashes == b.minhash.hashes
    assert utils._load_single_signature_native(os.path.join(FX, "20_genomes_sketches.zip"), 31) is None   # 20 files
    assert utils._load_single_signature_native(f, 21) is None                                              # no such k
    with pytest.raises(ValueError, match="Expected exactly one signature with ksize 21"):
        utils.load_signature_with_ksize(f, 21)
    with pytest.raises(ValueError, match="Empty sketch"):
        utils.load_signature_with_ksize(os.path.join(FX, "extract_empty_hash.sig.zip"), 31)
    # two k-mer sizes in one file, the wanted one second: the core's "first signature" is another sketch -> general reader
    two = [sigio.make_signature(np.array([1, 5, 9], np.uint64), ksize=21, scaled=1000, name="two"),
           sigio.make_signature(np.array([2, 6, 10, 14], np.uint64), ksize=31, scaled=1000, name="two")]
    rec = sigio.signature_record(two[0])
    rec["signatures"].append(sigio.signature_record(two[1])["signatures"][0])
    import zipfile
    z = tmp_path / "two.sig.zip"
    with zipfile.ZipFile(z, "w") as out:
        out.writestr(sigio.MANIFEST_NAME, sigio.MANIFEST_HEADER + "\n")
        out.writestr("signatures/x.sig", js.dumps([rec]))
    assert utils._load_single_signature_native(str(z), 31) is None
    got = utils.load_signature_with_ksize(str(z), 31)
    assert got.minhash.mins.tolist() == [2, 6, 10, 14] and type(got.minhash).__name__ == "MinHash"
    first = utils.load_signature_with_ksize(str(z), 21)     # the first signature IS the wanted one: native
    assert first.minhash.mins.tolist() == [1, 5, 9] and type(first.minhash).__name__ == "_NativeMinHash" and first.minhash.mean_abundance is None


def test_path_lists_are_what_the_reference_writes(tmp_path):
    """hypothesis_recovery_src._write_path_list (and the list of run_yacht_train_core): byte for byte what the reference's
    `pd.DataFrame(paths).to_csv(path, header=False, index=False)` writes, including the paths a CSV reader needs quoted."""
    import pandas as pd

    from yacht_amd.hypothesis_recovery_src import _write_path_list

    paths = ["/a/b.sig", "/with,comma/x.sig", '/with"quote/y.sig', "/plain/z", "/sp ace/w.sig", "/uni\u00e9/q.sig"]
    _write_path_list(str(tmp_path / "mine.txt"), paths)
    pd.DataFrame(paths).to_csv(tmp_path / "theirs.txt", header=False, index=False)
    assert (tmp_path / "mine.txt").read_bytes() == (tmp_path / "theirs.txt").read_bytes()


def test_pool_release_is_exported_and_harmless_without_a_gpu():
    """yh_pool_release exists (idle blocks of the device buffer cache back to the driver on demand); with nothing
    cached -- and no GPU -- it releases zero bytes."""
    assert _lib.pool_release() == 0
    assert _lib.alloc_stats()["bytes_idle"] == 0


def test_traffic_files_were_taken_from_this_source_of_the_kernels():
    """profiles/traffic_r06_*.json carry the sha256 of the kernel sources their PMC passes ran on; bench.py / bench_train.py attach
    `roofline.traffic` only when it matches.  A mismatch is not an error of the code -- the kernels were edited since the last
    measurement session -- so it skips, loudly, instead of failing: retake with scripts/measure_session.sh + adopt_session.sh."""
    import hashlib
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def src(files):
        h = hashlib.sha256()
        for f in files:
            with open(os.path.join(root, "yacht_amd", "csrc", f), "rb") as fh:
                h.update(fh.read())
        return h.hexdigest()[:16]

    want = {"fused": src(("yh_query.hip", "yh_common.h")), "train": src(("yh_sort.hip", "yh_pairwise.hip", "yh_common.h"))}
    stale = []
    for n, tag in want.items():
        with open(os.path.join(root, "profiles", f"traffic_r06_{n}.json")) as f:
            if json.load(f)["source_tag"] != tag:
                stale.append(n)
    if stale:
        pytest.skip(f"PMC traffic files are older than the kernels they describe: {stale} (bench lines will carry traffic = null)")


def test_native_md5_of_a_sketch_equals_hashlib(tmp_path):
    """Round 6: the scanner's MD5 (RFC 1321, unrolled) over str(ksize) + every hash in decimal -- sourmash's md5sum of a sketch -- against
    hashlib for sketches of many sizes (every alignment of the 64-byte block buffer; the digits arrive a number at a time)."""
    import hashlib

    from yacht_amd import sigio, utils

    rng = np.random.default_rng(66)
    d = tmp_path / "signatures"
    d.mkdir()
    want = {}
    for i, n in enumerate([1, 2, 3, 4, 5, 7, 13, 64, 65, 127, 128, 129, 1000, 3900, 6000] + [int(x) for x in rng.integers(1, 300, size=25)]):
        m = np.unique(rng.integers(0, 18446744073709552 if i % 3 else 2 ** 64 - 1, size=n, dtype=np.uint64))
        want[f"g{i}"] = hashlib.md5(("31" + "".join(str(int(x)) for x in m)).encode()).hexdigest()
        sigio.write_sig(sigio.make_signature(m, ksize=31, scaled=1000, name=f"g{i}", abundances=np.ones(m.size, np.int64)), str(d / f"s{i}.sig"))
    info = utils.collect_signature_info(3, 31, str(tmp_path))
    assert {k: v[0] for k, v in info.items()} == want
