#!/usr/bin/env python3
"""Generate tests/golden/*.json|npz from the GENUINE reference, in the build container.

    python tests/golden/make_golden.py            (needs /root/reference and `make -C oracle ref`)
    python tests/golden/make_golden.py cfg3       (only golden_train_cfg3.json: configs[3] at full size; `all` = everything)
    python tests/golden/make_golden.py hypreal    (only golden_hyp_real.*: the hypothesis test on the (e, m) of real runs)

Two sources, both the reference's own code, neither of which travels to the GPU box:

  * the reference train core, compiled unmodified from /root/reference/src/cpp/main.cpp into
    oracle/_ref/run_yacht_train_core (recipe: oracle/Makefile) and run here on seeded sketches;
  * the reference's Python, imported from /root/reference/src/yacht with two module stubs for
    packages this image lacks (`sourmash`: only named in type annotations and in
    load_signature_with_ksize, which is replaced by an in-memory loader below; `loguru`: logging
    only).  scipy / numpy / pandas are the real ones.  The functions exercised —
    get_exclusive_hashes, single_hyp_test, get_alt_mut_rate — run unmodified.

Only numbers are written out: inputs (seeded CSR arrays) and the outputs the reference produced.
The fixtures under tests/golden/fixtures/ are DATA files the reference's own tests hold
(tests/testdata/*.zip, tests/unittests_data/*.json), copied byte for byte.
"""
from __future__ import annotations

import importlib
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("YACHT_REFERENCE", "/root/reference")

from oracle import oracle  # noqa: E402  (run_ref_exe wrapper only)
from yacht_amd import sigio, synth  # noqa: E402


# ---------------------------------------------------------------------------------------------------
def import_reference():
    sm = types.ModuleType("sourmash")
    sm.SourmashSignature = object

    def _no_sourmash(*a, **k):
        raise RuntimeError("sourmash is not installed; loaders are patched by make_golden.py")

    sm.load_file_as_signatures = _no_sourmash
    lg = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    lg.logger = _Logger()
    sys.modules["sourmash"] = sm
    sys.modules["loguru"] = lg
    pkg = types.ModuleType("yacht")
    pkg.__path__ = [os.path.join(REF, "src", "yacht")]  # bypass yacht/__init__.py (needs biom, pytaxonkit)
    sys.modules["yacht"] = pkg
    hr = importlib.import_module("yacht.hypothesis_recovery_src")
    utils = importlib.import_module("yacht.utils")
    return hr, utils


class _MH:
    def __init__(self, mins):
        self.hashes = {int(h): 1 for h in mins}


class _Sig:
    def __init__(self, mins):
        self.minhash = _MH(mins)


def tolist(a):
    return [int(x) for x in a]


# ---------------------------------------------------------------------------------------------------
def golden_hyp(hr) -> dict:
    # known answers held by the reference's own tests (tests/test_unit.py:11-20,
    # tests/test_unittests.py:86-111): [nu, thresh, ksize, significance, expected]
    alt_kat = [
        [100, 10000, 21, 0.99, -1.0],
        [10, 0, 21, 0.99, 0.28015945851802826],
        [10, 0, 31, 0.99, 0.19963312102481723],
        [10, 5, 21, 0.99, 0.0698992155957967],
        [10, 5, 31, 0.99, 0.047902071848511696],
        [10, 9, 21, 0.99, 0.02169068099465221],
        [100, 10, 11, 0.99, 0.2397729973308742],
        [1000, 0, 1, 0.99, 0.9999899497147453],
        [0, 5, 31, 0.99, -1.0],
        [10, 20, 31, 0.99, -1.0],
    ]
    alt_now = [[*row[:4], float(hr.get_alt_mut_rate(row[0], row[1], row[2], row[3]))] for row in alt_kat]
    grid = []
    params = [(31, 0.99, 0.95), (31, 0.90, 0.90), (51, 0.95, 0.95), (21, 0.99, 0.9995)]
    for (k, sig, ani) in params:
        for cov in (1, 0.5, 0.1, 0.05, 0.01, 0.001):
            for e in (0, 1, 2, 10, 100, 1000, 2829, 5000, 50000):
                thr = int(hr.single_hyp_test((e, 0), k, sig, ani, cov)[5])
                for m in sorted({0, 1, max(thr - 1, 0), thr, thr + 1, e}):
                    r = hr.single_hyp_test((e, m), k, sig, ani, cov)
                    grid.append({"e": e, "m": m, "k": k, "sig": sig, "ani": ani, "cov": cov,
                                 "out": [bool(r[0]), float(r[1]), int(r[2]), int(r[3]), int(r[4]), float(r[5]),
                                         float(r[6]), float(r[7])]})
    return {"alt_mut_rate_reference_tests": alt_kat, "alt_mut_rate_here": alt_now, "single_hyp_test": grid}


def golden_exclusive(hr) -> tuple:
    import pandas as pd

    arrays = {}
    cases = []
    rng = np.random.default_rng(20240501)

    def run_case(tag, refs, names, nontrivial, sample):
        md5s = [f"{tag}_{i:04d}" for i in range(len(refs))]
        store = {m: r for m, r in zip(md5s, refs)}
        hr.load_signature_with_ksize = lambda path, ksize: _Sig(store[os.path.basename(path)[: -len(".sig")]])
        manifest = pd.DataFrame({"organism_name": names, "md5sum": md5s})
        info, sub = hr.get_exclusive_hashes(manifest, nontrivial, _Sig(sample), 31, "/nonexistent")
        values, offsets = synth.pack(refs)
        arrays[f"{tag}_values"] = values
        arrays[f"{tag}_offsets"] = offsets
        arrays[f"{tag}_sample"] = np.asarray(sample, dtype=np.uint64)
        cases.append({"tag": tag, "names": names, "nontrivial": nontrivial,
                      "sub_rows": [md5s.index(m) for m in sub["md5sum"]],
                      "info": [[int(a), int(b)] for a, b in info]})

    # (1) clustered genomes, subset = a hand-picked list
    refs = synth.clustered_refs(rng, 6, (1.0, 0.9, 0.5, 0.25, 0.1), 200)
    names = [f"org{i}" for i in range(len(refs))]
    sample = synth.sample_from_refs(rng, refs, [0, 1, 7, 12, 29], 0.6, 3000)
    run_case("clusters", refs, names, ["org0", "org1", "org2", "org7", "org12", "org13", "org29"], sample)
    # (2) duplicate organism names (both rows selected), a subset of size 1, a hash held by 3 references
    base = synth.random_sketch(rng, 300, synth.max_hash_for_scaled(1000))
    r0 = base[:200].copy()
    r1 = base[100:300].copy()
    r2 = np.union1d(base[150:180], synth.random_sketch(rng, 100, synth.max_hash_for_scaled(1000)))
    r3 = synth.random_sketch(rng, 150, synth.max_hash_for_scaled(1000))
    refs2 = [r0, r1, r2, r3]
    names2 = ["dup", "dup", "third", "lonely"]
    sample2 = np.unique(np.concatenate([base[::3], r3[::2]]))
    run_case("dupnames", refs2, names2, ["dup", "third"], sample2)
    run_case("single", refs2, ["a", "b", "c", "d"], ["c"], sample2)
    run_case("all", refs2, ["a", "b", "c", "d"], ["a", "b", "c", "d"], sample2)
    return arrays, cases


def golden_train() -> tuple:
    arrays = {}
    cases = []
    rng = np.random.default_rng(20240502)

    def run_case(tag, refs, c, threads):
        with tempfile.TemporaryDirectory() as d:
            selected, lines, stdout = oracle.run_ref_exe(refs, c, d, threads=threads)
        stats = {}
        for ln in stdout.splitlines():
            if ln.startswith("Total number of distinct hashes:"):
                stats["distinct"] = int(ln.split(":")[1])
            elif "appear in only one sketch" in ln:
                stats["singletons"] = int(ln.split(":")[1])
            elif ln.startswith("Size of the index:"):
                stats["index"] = int(ln.split(":")[1])
            elif ln.startswith("Number of empty sketches:"):
                stats["empty"] = int(ln.split(":")[1])
        values, offsets = synth.pack(refs)
        arrays[f"{tag}_values"] = values
        arrays[f"{tag}_offsets"] = offsets
        cases.append({"tag": tag, "c": c, "threads": threads, "selected": selected, "pair_lines": lines,
                      "stats": stats})

    # the worked example of SURVEY.md §8c: threshold hit exactly, ties, duplicate, subset, empty
    A = np.arange(1, 1001, dtype=np.uint64)
    B = np.concatenate([np.arange(1, 251), np.arange(5000, 6750)]).astype(np.uint64)
    X = np.arange(10000, 10500, dtype=np.uint64)
    Z = np.arange(10000, 10100, dtype=np.uint64)
    micro = [A, B, X, X.copy(), Z, np.zeros(0, np.uint64), np.array([20000], np.uint64)]
    run_case("micro", micro, 0.25, 3)
    # N = 17: the first size at which libstdc++'s introsort is no longer a plain insertion sort;
    # many equal sizes so that tie order matters
    fam = synth.random_sketch(rng, 400, synth.max_hash_for_scaled(1000))
    n17 = [fam[rng.random(fam.size) < 0.8][:250].copy() for _ in range(17)]
    run_case("n17_ties", n17, 0.95 ** 31, 2)
    c64 = synth.clustered_refs(rng, 13, (1.0, 0.9, 0.5, 0.25, 0.1), 300)[:64]
    run_case("n64", c64, 0.95 ** 31, 4)
    # N < T
    run_case("n3_t8", c64[:3], 0.95 ** 31, 8)
    # 512 small sketches, equal sizes inside clusters (private_fill tops everyone up to `size`)
    c512 = synth.clustered_refs(rng, 103, (1.0, 0.9, 0.5, 0.25, 0.1), 60)[:512]
    run_case("n512", c512, 0.95 ** 31, 8)
    run_case("n512_c05", c512, 0.5, 1)
    return arrays, cases


def xlsx_columns(path: str, sheet: str, columns) -> dict:
    """Numeric columns of one sheet of an .xlsx (a zip of XML; openpyxl is not in this image): {header: [values]}."""
    import re
    import zipfile

    z = zipfile.ZipFile(path)
    wb = z.read("xl/workbook.xml").decode()
    names = re.findall(r'<sheet name="([^"]+)" sheetId="(\d+)"', wb)
    sid = {n: i for n, i in names}[sheet]
    xml = z.read(f"xl/worksheets/sheet{sid}.xml").decode()
    rows = re.findall(r"<row [^>]*>(.*?)</row>", xml, flags=re.S)
    cell = re.compile(r'<c r="([A-Z]+)\d+"[^>]*?(?:/>|>(?:<is><t>(.*?)</t></is>|<v>(.*?)</v>)</c>)', flags=re.S)
    header = {col: (txt if txt is not None else val) for col, txt, val in cell.findall(rows[0])}
    want = {col: name for col, name in header.items() if name in columns}
    out = {name: [] for name in want.values()}
    for r in rows[1:]:
        got = {col: val for col, _txt, val in cell.findall(r) if col in want}
        for col, name in want.items():
            out[name].append(float(got[col]))
    return out


def golden_hyp_real(hr) -> dict:
    """(n_exclusive, n_matches) pairs of REAL runs -- the raw_result sheets of the reference's shipped
    use_case_examples/**/result_*.xlsx -- through the reference's single_hyp_test at three coverages each, with the
    parameters those workbooks were made with.  Every 12th row: ~17 000 tuples."""
    files = [("use_case_examples/low_abundance_samples/result_k31_ani0.90.xlsx", 31, 0.90, 0.90),
             ("use_case_examples/low_abundance_samples/result_k51_ani0.95.xlsx", 51, 0.95, 0.95),
             ("use_case_examples/MAG_fishing/result_k51_ani0.95_SRR32008482.xlsx", 51, 0.95, 0.95)]
    cols = {k: [] for k in ("file", "e", "m", "cov", "present", "p_val", "n_cov", "thr", "conf", "alt")}
    params = []
    for fi, (rel, k, sig, ani) in enumerate(files):
        tab = xlsx_columns(os.path.join(REF, rel), "raw_result", ("num_exclusive_kmers_to_genome", "num_matches"))
        e_all = [int(x) for x in tab["num_exclusive_kmers_to_genome"]]
        m_all = [int(x) for x in tab["num_matches"]]
        params.append({"file": rel, "ksize": k, "significance": sig, "ani_thresh": ani, "rows_in_sheet": len(e_all)})
        for e, m in list(zip(e_all, m_all))[fi::12]:
            for cov in (1.0, 0.1, 0.01):
                r = hr.single_hyp_test((e, m), k, sig, ani, cov)
                for name, v in zip(("file", "e", "m", "cov", "present", "p_val", "n_cov", "thr", "conf", "alt"),
                                   (fi, e, m, cov, bool(r[0]), float(r[1]), int(r[3]), float(r[5]), float(r[6]), float(r[7]))):
                    cols[name].append(v)
    arrays = {"file_index": np.asarray(cols["file"], np.uint8), "e": np.asarray(cols["e"], np.uint32),
              "m": np.asarray(cols["m"], np.uint32), "cov": np.asarray(cols["cov"], np.float64),
              "present": np.asarray(cols["present"], np.bool_), "p_val": np.asarray(cols["p_val"], np.float64),
              "n_cov": np.asarray(cols["n_cov"], np.uint32), "thr": np.asarray(cols["thr"], np.float64),
              "conf": np.asarray(cols["conf"], np.float64), "alt": np.asarray(cols["alt"], np.float64)}
    return arrays, params


def golden_train_cfg3() -> dict:
    """BASELINE.json configs[3] at its REAL size (synth.config4(): 10 000 sketches of ~5 000 hashes) through the
    genuine reference executable, `-t 8 -c 0.95**31 -p 1` (src/cpp/main.cpp:215-420 end to end).  The input is
    regenerated from its seed by the test (a digest of it is stored); stored outputs: the selected ids in walk
    order, the three printed statistics, and a sha256 over the sorted pair lines (the lines themselves: 1.4 MB)."""
    import hashlib

    values, offsets = synth.config4()
    refs = [values[int(offsets[i]):int(offsets[i + 1])] for i in range(offsets.size - 1)]
    c = 0.95 ** 31
    with tempfile.TemporaryDirectory(dir=os.environ.get("YACHT_GOLDEN_TMP")) as d:
        selected, lines, stdout = oracle.run_ref_exe(refs, c, d, threads=8)
    stats = {}
    for ln in stdout.splitlines():
        if ln.startswith("Total number of distinct hashes:"):
            stats["distinct"] = int(ln.split(":")[1])
        elif "appear in only one sketch" in ln:
            stats["singletons"] = int(ln.split(":")[1])
        elif ln.startswith("Size of the index:"):
            stats["index"] = int(ln.split(":")[1])
        elif ln.startswith("Number of empty sketches:"):
            stats["empty"] = int(ln.split(":")[1])
    return {"generator": "yacht_amd.synth.config4()", "n_refs": int(offsets.size - 1), "n_hashes": int(values.size),
            "input_sha256": hashlib.sha256(values.tobytes() + offsets.tobytes()).hexdigest(),
            "c": c, "threads": 8, "selected": selected, "n_pair_lines": len(lines),
            "pair_lines_sha256": hashlib.sha256("\n".join(lines).encode()).hexdigest(),
            "first_pair_lines": lines[:5], "stats": stats}


def golden_fixture(hr) -> dict:
    """The reference's end-to-end known answer (tests/test_workflow.py:62-66) reproduced from the
    raw fixture JSON: which genomes overlap the sample, their (n_exclusive, n_matches), and the
    hypothesis-test row at min_coverage 0.001."""
    import pandas as pd

    fx = os.path.join(HERE, "fixtures")
    refs = sigio.load_file_as_signatures(os.path.join(fx, "20_genomes_sketches.zip"), ksize=31)
    refs.sort(key=lambda s: s.md5sum())  # a fixed order for the vector below
    sample = sigio.load_file_as_signatures(os.path.join(fx, "sample.sig.zip"), ksize=31)[0]
    sset = set(int(h) for h in sample.minhash.mins)
    overlap = [len(sset.intersection(int(h) for h in r.minhash.mins)) for r in refs]  # python sets, not the plugin
    store = {r.md5sum(): r.minhash.mins for r in refs}
    hr.load_signature_with_ksize = lambda path, ksize: _Sig(store[os.path.basename(path)[: -len(".sig")]])
    manifest = pd.DataFrame({"organism_name": [r.name for r in refs], "md5sum": [r.md5sum() for r in refs]})
    names = [r.name for r, o in zip(refs, overlap) if o > 0]
    info, sub = hr.get_exclusive_hashes(manifest, names, _Sig(sample.minhash.mins), 31, "/nonexistent")
    rows = []
    for (e, m), name in zip(info, sub["organism_name"]):
        r = hr.single_hyp_test((e, m), 31, 0.99, 0.95, 0.001)
        rows.append({"organism_name": name, "n_exclusive": int(e), "n_matches": int(m),
                     "hyp_cov_0.001": [bool(r[0]), float(r[1]), int(r[2]), int(r[3]), int(r[4]), float(r[5]),
                                       float(r[6]), float(r[7])]})
    return {"md5_order": [r.md5sum() for r in refs], "names": [r.name for r in refs],
            "sizes": [len(r.minhash) for r in refs], "overlap_python_sets": overlap,
            "sample_hashes": len(sample.minhash), "sample_mean_abundance": sample.minhash.mean_abundance,
            "rows": rows}


def copy_fixtures() -> None:
    fx = os.path.join(HERE, "fixtures")
    os.makedirs(fx, exist_ok=True)
    for rel in ("tests/testdata/20_genomes_sketches.zip", "tests/testdata/sample.sig.zip",
                "tests/testdata_bug_YAC13/extract_empty_hash.sig.zip",
                "tests/unittests_data/test_collect_signature_info_data.json",
                "demo/ref_genomes/GCF_018918235.1_genomic.fna.gz", "demo/ref_genomes/GCF_018918045.1_genomic.fna.gz"):
        dst = os.path.join(fx, os.path.basename(rel))
        shutil.copyfile(os.path.join(REF, rel), dst)
        os.chmod(dst, 0o644)


def main() -> None:
    assert os.path.isdir(REF), f"{REF} not found"
    oracle.build(with_ref=True)
    assert oracle.have_ref_exe()
    if "cfg3" in sys.argv[1:] or "all" in sys.argv[1:]:  # minutes of the reference exe on 10 000 files: on request
        with open(os.path.join(HERE, "golden_train_cfg3.json"), "w") as f:
            json.dump(golden_train_cfg3(), f)
        print("golden_train_cfg3.json written")
        if "cfg3" in sys.argv[1:]:
            return
    copy_fixtures()
    hr, _utils = import_reference()
    if "hypreal" in sys.argv[1:] or "all" in sys.argv[1:]:
        import scipy

        arrays, params = golden_hyp_real(hr)
        np.savez_compressed(os.path.join(HERE, "golden_hyp_real.npz"), **arrays)
        with open(os.path.join(HERE, "golden_hyp_real.json"), "w") as f:
            json.dump({"files": params, "n": int(arrays["e"].size), "scipy": scipy.__version__}, f)
        print("golden_hyp_real written:", int(arrays["e"].size), "tuples")
        if "hypreal" in sys.argv[1:]:
            return
    with open(os.path.join(HERE, "golden_hyp.json"), "w") as f:
        json.dump(golden_hyp(hr), f)
    arrays, cases = golden_exclusive(hr)
    np.savez_compressed(os.path.join(HERE, "golden_exclusive.npz"), **arrays)
    with open(os.path.join(HERE, "golden_exclusive.json"), "w") as f:
        json.dump(cases, f)
    arrays, cases = golden_train()
    np.savez_compressed(os.path.join(HERE, "golden_train.npz"), **arrays)
    with open(os.path.join(HERE, "golden_train.json"), "w") as f:
        json.dump(cases, f)
    with open(os.path.join(HERE, "golden_fixture.json"), "w") as f:
        json.dump(golden_fixture(hr), f, indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
