"""GPU: the HIP path against the committed golden vectors of the genuine reference, the
reference's end-to-end known answer through the mirrored host functions, and full-size
properties at BASELINE.json's GTDB-rs214 scale."""
import json
import os
import shutil
import zipfile

import numpy as np
import pandas as pd
import pytest

from yacht_amd import hypothesis_recovery_src as hr
from yacht_amd import sigio, utils
from yacht_amd.engine import RefDB, YH_DB_KEEP_CSR, train_select
from yacht_amd.train_core import format_pair_line

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FX = os.path.join(GOLD, "fixtures")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_train_goldens(hip_lib):
    """pair lines, index statistics and selection order equal to what the reference executable
    produced (exact-threshold pair, ties, duplicates, empty sketch, N=17/64/512)."""
    cases = _load("golden_train.json")
    arrays = np.load(os.path.join(GOLD, "golden_train.npz"))
    for c in cases:
        values, offsets = arrays[c["tag"] + "_values"], arrays[c["tag"] + "_offsets"]
        sizes = np.diff(offsets).astype(np.uint32)
        with RefDB(values, offsets) as db:
            pi, pj, pc = db.pairwise(c["c"])
            stats = db.index_stats()
        lines = [format_pair_line(int(i), int(j), int(k), int(sizes[i]), int(sizes[j])) for i, j, k in zip(pi, pj, pc)]
        assert lines == c["pair_lines"], c["tag"]
        assert stats == (c["stats"]["distinct"], c["stats"]["singletons"], c["stats"]["index"]), c["tag"]
        assert train_select(sizes, pi, pj).tolist() == c["selected"], c["tag"]


def test_exclusive_goldens(hip_lib):
    cases = _load("golden_exclusive.json")
    arrays = np.load(os.path.join(GOLD, "golden_exclusive.npz"))
    for c in cases:
        values, offsets = arrays[c["tag"] + "_values"], arrays[c["tag"] + "_offsets"]
        sample = arrays[c["tag"] + "_sample"]
        mask = np.array([n in set(c["nontrivial"]) for n in c["names"]])
        with RefDB(values, offsets) as db:
            e, m = db.exclusive(mask, sample)
        assert [[int(e[j]), int(m[j])] for j in c["sub_rows"]] == c["info"], c["tag"]


def test_train_core_run_writes_reference_files(hip_lib, tmp_path):
    """The in-process train core leaves the files the reference executable leaves: per-(pass,
    thread) comparison files with identical lines and the selected paths in walk order."""
    from oracle import oracle

    cases = {c["tag"]: c for c in _load("golden_train.json")}
    arrays = np.load(os.path.join(GOLD, "golden_train.npz"))
    from yacht_amd import train_core

    for tag in ("micro", "n64", "n3_t8"):
        c = cases[tag]
        values, offsets = arrays[tag + "_values"], arrays[tag + "_offsets"]
        refs = [values[int(offsets[j]):int(offsets[j + 1])] for j in range(offsets.size - 1)]
        work = tmp_path / tag
        work.mkdir()
        paths = oracle.write_minimal_sigs(refs, str(work / "sigs"))
        flist = work / "filelist.txt"
        flist.write_text("\n".join(paths) + "\n")
        out = work / "selected.txt"
        train_core.run(str(flist), str(work), str(out), threads=c["threads"], passes=1, containment_threshold=c["c"])
        assert [paths.index(p) for p in out.read_text().split("\n") if p] == c["selected"]
        names = sorted(f for f in os.listdir(work) if f.endswith(".txt") and f[0].isdigit())
        assert names == [f"0_{t:03d}.txt" for t in range(c["threads"])]
        lines = [ln for n in names for ln in (work / n).read_text().split("\n") if ln]
        assert sorted(lines, key=lambda s: tuple(int(x) for x in s.split(",")[:2])) == c["pair_lines"]
    with pytest.raises(ValueError, match="containment threshold must be between 0.0 and 1.0"):
        train_core.run(str(flist), str(work), str(out), containment_threshold=1.5)


def test_dropin_executable(hip_lib, tmp_path):
    """yacht_amd/lib/run_yacht_train_core: the reference executable's argv and files."""
    import subprocess

    from oracle import oracle
    from yacht_amd import build

    exe = build.EXE_PATH
    assert os.path.exists(exe), "run `python -m yacht_amd.build` (build() does) before the GPU tests"
    cases = {c["tag"]: c for c in _load("golden_train.json")}
    arrays = np.load(os.path.join(GOLD, "golden_train.npz"))
    for tag in ("micro", "n17_ties", "n512"):
        c = cases[tag]
        values, offsets = arrays[tag + "_values"], arrays[tag + "_offsets"]
        refs = [values[int(offsets[j]):int(offsets[j + 1])] for j in range(offsets.size - 1)]
        work = tmp_path / tag
        work.mkdir()
        paths = oracle.write_minimal_sigs(refs, str(work / "sigs"))
        (work / "filelist.txt").write_text("\n".join(paths) + "\n")
        out = work / "selected.txt"
        proc = subprocess.run([exe, "-t", str(c["threads"]), "-c", repr(c["c"]), "-p", "1", str(work / "filelist.txt"),
                               str(work), str(out)], capture_output=True, text=True)
        assert proc.returncode == 0, proc.stderr
        assert [paths.index(p) for p in out.read_text().split("\n") if p] == c["selected"]
        names = sorted(f for f in os.listdir(work) if f.endswith(".txt") and f[0].isdigit())
        assert names == [f"0_{t:03d}.txt" for t in range(c["threads"])]
        lines = [ln for n in names for ln in (work / n).read_text().split("\n") if ln]
        assert sorted(lines, key=lambda s: tuple(int(x) for x in s.split(",")[:2])) == c["pair_lines"]
        assert f"Total number of distinct hashes: {c['stats']['distinct']}" in proc.stdout
        assert f"Size of the index: {c['stats']['index']}" in proc.stdout
        assert f"Number of empty sketches: {c['stats']['empty']}" in proc.stdout
    bad = subprocess.run([exe, "-c", "1.5", "a", "b", "c"], capture_output=True, text=True)
    assert bad.returncode == 1 and "containment threshold must be between 0.0 and 1.0" in bad.stderr
    assert "Usage:" in bad.stdout


@pytest.fixture()
def trained_fixture(tmp_path):
    """`yacht train` on the reference's 20-genome fixture, through the mirrored helpers
    (make_training_data_from_sketches.py:107-155 of the reference)."""
    work = tmp_path / "gtdb_ani_thresh_0.95_intermediate_files"
    with zipfile.ZipFile(os.path.join(FX, "20_genomes_sketches.zip")) as z:
        z.extractall(work)
    import glob

    utils.decompress_all_sig_files(glob.glob(str(work / "signatures" / "*.sig.gz")), 2)
    info = utils.collect_signature_info(2, 31, str(work))
    manifest = utils.run_yacht_train_core(4, 0.95, 31, str(work), info)
    return work, info, manifest


def test_reference_workflow_known_answer(hip_lib, trained_fixture, tmp_path):
    """tests/test_workflow.py::test_full_workflow of the reference, on the HIP path: files left by
    train, then CP032507.1 present with 2 matches and threshold 0 at min_coverage 0.001."""
    work, info, manifest = trained_fixture
    assert len(info) == 20 and len(manifest) == 20  # only 9 hashes are shared: nothing is removed
    for f in ("training_sig_files.tsv", "selected_result.tsv", "SOURMASH-MANIFEST.csv",
              "signatures/04212e93c2172d4df49dc5d8c2973d8b.sig", "comparison_files/0_000.txt"):
        assert (work / f).exists(), f
    want = _load("golden_fixture.json")
    assert sorted(manifest["md5sum"]) == sorted(want["md5_order"])
    row = manifest[manifest["organism_name"].str.startswith("CP032507.1")].iloc[0]
    assert int(row["num_unique_kmers_in_genome_sketch"]) >= 3741 and int(row["genome_scale_factor"]) == 1000

    sample_zip = tmp_path / "sample.sig.zip"
    shutil.copyfile(os.path.join(FX, "sample.sig.zip"), sample_zip)
    sample_sig = utils.load_signature_with_ksize(str(sample_zip), 31)
    results = hr.hypothesis_recovery(manifest, (str(sample_zip), sample_sig), str(work), [1.0, 0.001], 1000, 31,
                                     0.99, 0.95, 2)
    hr.release_reference_dbs()
    assert len(results) == 2
    df = results[1]
    assert list(df.columns[-8:]) == hr.GIVEN_COLUMNS and float(df["min_coverage"].iloc[0]) == 0.001
    assert len(df) == 1
    r = df.iloc[0]
    assert r["organism_name"] == "CP032507.1 Ectothiorhodospiraceae bacterium BW-2 chromosome, complete genome"
    assert bool(r["in_sample_est"]) is True and int(r["num_matches"]) == 2
    assert float(r["acceptance_threshold_with_coverage"]) == 0.0
    assert int(r["num_exclusive_kmers_to_genome"]) == 3741 and int(r["num_exclusive_kmers_to_genome_coverage"]) == 3
    g = want["rows"][0]["hyp_cov_0.001"]
    assert float(r["p_vals"]) == pytest.approx(g[1], rel=1e-12)
    assert float(r["actual_confidence_with_coverage"]) == pytest.approx(g[6], rel=1e-12)
    assert float(r["alt_confidence_mut_rate_with_coverage"]) == pytest.approx(g[7], rel=1e-12)
    sdir = tmp_path / "sample_sample_intermediate_files"
    assert (sdir / "sample_multisearch_result.csv").exists() and (sdir / "organism_sig_file.txt").exists()
    assert (work / hr.DB_CACHE_NAME).exists()  # the packed database next to the training output


def test_full_scale_properties(hip_lib):
    """BASELINE.json configs[2] scale (85 205 references, ~3.3e8 hashes).  Properties that need no oracle:
    two independent kernels agree bit for bit; a reference queried as the sample overlaps itself
    completely; overlap counts are bounded by sketch sizes; the sum of overlaps equals the number
    of (sample hash, reference) incidences counted through the inverted index."""
    import torch

    from yacht_amd import synth

    values, offsets, sample = synth.config3_device(seed=4242, n_refs=85_205, n_sample=1_000_000, device="cuda:0")
    n = offsets.numel() - 1
    sizes = (offsets[1:] - offsets[:-1]).cpu().numpy().astype(np.uint32)
    db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n, flags=YH_DB_KEEP_CSR)
    try:
        a = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        b = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        db.overlap_device(sample.data_ptr(), sample.numel(), a.data_ptr())
        db.overlap_bsearch_device(sample.data_ptr(), sample.numel(), b.data_ptr())
        db.synchronize()
        ov = a.cpu().numpy().view(np.uint32)
        assert np.array_equal(ov, b.cpu().numpy().view(np.uint32))
        assert (ov <= sizes).all() and int((ov > 0).sum()) >= 200
        # independent count of incidences: every sample hash that is in the database, times the
        # number of references holding it (torch ops on the raw CSR, no yacht_amd kernel)
        pos = torch.searchsorted(sample, values).clamp_(max=sample.numel() - 1)
        assert int((sample[pos] == values).sum().item()) == int(ov.astype(np.int64).sum())
        del pos
        # self query: reference j as the sample
        for j in (0, 12345, n - 1):
            lo, hi = int(offsets[j]), int(offsets[j + 1])
            one = values[lo:hi].contiguous()
            db.overlap_device(one.data_ptr(), one.numel(), a.data_ptr())
            db.synchronize()
            assert int(a[j].item()) == hi - lo
        # exclusive counts: e_j <= |R_j|, m_j <= min(e_j, overlap_j); unmasked rows are zero
        e = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        m = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        db.run_device(sample.data_ptr(), sample.numel(), a.data_ptr(), e.data_ptr(), m.data_ptr())
        db.synchronize()
        ov2, e, m = (t.cpu().numpy().view(np.uint32) for t in (a, e, m))
        assert np.array_equal(ov2, ov)
        assert (e <= sizes).all() and (m <= e).all() and (m <= ov).all()
        assert not e[ov == 0].any() and not m[ov == 0].any()
        # ... and the same sample against the CPU oracle on the WHOLE database (all host threads: ~2 s on the GPU box),
        # through the plain step, the one-launch pipelined step and the packed upload + compact rows
        from oracle import oracle

        h_values = values.cpu().numpy().view(np.uint64)
        h_offsets = offsets.cpu().numpy().view(np.uint64)
        h_sample = sample.cpu().numpy().view(np.uint64)
        want_ov = oracle.overlap(h_values, h_offsets, h_sample, threads=oracle.hardware_threads())
        want_e, want_m = oracle.exclusive(h_values, h_offsets, want_ov > 0, h_sample)
        assert np.array_equal(ov, want_ov) and np.array_equal(e, want_e) and np.array_equal(m, want_m)
        bufs = [torch.zeros(3, n, dtype=torch.int32, device="cuda:0") for _ in range(3)]
        for k in range(5):
            c = bufs[k % 3]
            db.run_device_pipelined(sample.data_ptr(), sample.numel(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())
        db.run_device_join()
        db.synchronize()
        for c in bufs:
            got = c.cpu().numpy().view(np.uint32)
            assert np.array_equal(got[0], want_ov) and np.array_equal(got[1], want_e) and np.array_equal(got[2], want_m)
        rows = db.run_rows(h_sample)  # (packs the sample itself)
        keep = np.flatnonzero(want_ov)
        assert np.array_equal(rows["ref"], keep) and np.array_equal(rows["overlap"], want_ov[keep])
        assert np.array_equal(rows["n_excl"], want_e[keep]) and np.array_equal(rows["n_match"], want_m[keep])
    finally:
        db.close()


def test_cli_train_then_run_end_to_end(hip_lib, tmp_path):
    """`yacht train` + `yacht run` through the command line entry point, as the reference's
    tests/test_workflow.py::test_full_workflow drives them: files, columns, the known answer."""
    from yacht_amd import cli

    out = tmp_path / "out"
    out.mkdir()
    ref_zip = tmp_path / "20_genomes_sketches.zip"
    sample_zip = tmp_path / "sample.sig.zip"
    shutil.copyfile(os.path.join(FX, "20_genomes_sketches.zip"), ref_zip)
    shutil.copyfile(os.path.join(FX, "sample.sig.zip"), sample_zip)
    assert cli.main(["train", "--ref_file", str(ref_zip), "--ksize", "31", "--prefix", "gtdb_ani_thresh_0.95",
                     "--ani_thresh", "0.95", "--outdir", str(out), "--num_threads", "2", "--force"]) == 0
    cfg = json.load(open(out / "gtdb_ani_thresh_0.95_config.json"))
    assert cfg["ksize"] == 31 and cfg["ani_thresh"] == 0.95 and cfg["scale"] == 1000
    work = out / "gtdb_ani_thresh_0.95_intermediate_files"
    for f in ("SOURMASH-MANIFEST.csv", "selected_result.tsv", "training_sig_files.tsv",
              "signatures/04212e93c2172d4df49dc5d8c2973d8b.sig"):
        assert (work / f).stat().st_size > 291
    man = pd.read_csv(out / "gtdb_ani_thresh_0.95_processed_manifest.tsv", sep="\t")
    assert list(man.columns) == ["organism_name", "md5sum", "num_unique_kmers_in_genome_sketch",
                                 "num_total_kmers_in_genome_sketch", "genome_scale_factor"] and len(man) == 20
    with pytest.raises(ValueError, match="already exists"):   # no --force
        cli.main(["train", "--ref_file", str(ref_zip), "--ksize", "31", "--prefix", "gtdb_ani_thresh_0.95",
                  "--outdir", str(out)])

    assert cli.main(["run", "--json", str(out / "gtdb_ani_thresh_0.95_config.json"), "--sample_file", str(sample_zip),
                     "--significance", "0.99", "--num_threads", "2", "--min_coverage_list", "0.001", "--show_all",
                     "--keep_raw", "--outdir", str(tmp_path)]) == 0
    res = tmp_path / "results"
    allr = pd.read_csv(res / "result_all.txt", sep="\t")
    sheet = pd.read_csv(res / "sheets" / "min_coverage0.001.tsv", sep="\t")
    raw = pd.read_csv(res / "sheets" / "raw_result.tsv", sep="\t")
    want_cols = ["organism_name", "num_unique_kmers_in_genome_sketch", "num_total_kmers_in_genome_sketch", "scale_factor",
                 "num_exclusive_kmers_in_sample_sketch", "num_total_kmers_in_sample_sketch", "min_coverage",
                 "in_sample_est", "p_vals", "num_exclusive_kmers_to_genome", "num_exclusive_kmers_to_genome_coverage",
                 "num_matches", "acceptance_threshold_with_coverage", "actual_confidence_with_coverage",
                 "alt_confidence_mut_rate_with_coverage"]
    assert list(allr.columns) == want_cols == list(sheet.columns)
    assert "acceptance_threshold_wo_coverage" in raw.columns and float(raw["min_coverage"].iloc[0]) == 1.0
    assert len(allr) == 1 and float(allr["min_coverage"].iloc[0]) == 0.001   # the forced 1.0 pass is not in result_all
    r = sheet[sheet["organism_name"] == "CP032507.1 Ectothiorhodospiraceae bacterium BW-2 chromosome, complete genome"].iloc[0]
    assert str(r["in_sample_est"]) == "True" and int(r["num_matches"]) == 2
    assert float(r["acceptance_threshold_with_coverage"]) == 0
    # the reference's quirk: this column carries the sample's mean abundance
    assert float(r["num_exclusive_kmers_in_sample_sketch"]) == pytest.approx(2.4032636839886794)
    assert int(r["num_total_kmers_in_sample_sketch"]) == int(np.round(2.4032636839886794 * 49821))
    # results/result.xlsx, read back as the reference's test_workflow.py:52-66 does (first sheet = the user's coverage
    # unless --keep_raw put raw_result in front; here by name), without openpyxl: the three look-ups of that test
    from yacht_amd import xlsx

    book = xlsx.read_xlsx(str(res / "result.xlsx"))
    assert list(book) == ["raw_result", "min_coverage0.001"]
    df = book["min_coverage0.001"]
    assert list(df.columns) == want_cols and len(df) == len(sheet)
    rx = df[df["organism_name"] == "CP032507.1 Ectothiorhodospiraceae bacterium BW-2 chromosome, complete genome"]
    assert str(rx["in_sample_est"].values[0]) == "True"
    assert rx["num_matches"].values[0] == 2
    assert rx["acceptance_threshold_with_coverage"].values[0] == 0
    for c in want_cols:  # the workbook and the TSV sheet carry the same table
        a, b = df[c].tolist(), sheet[c].tolist()
        assert all((x == y) or (isinstance(x, float) and isinstance(y, float) and (np.isnan(x) and np.isnan(y) or abs(x - y) <= 1e-15 * abs(y)))
                   for x, y in zip(a, b)), c


def test_cli_train_reads_an_archive_the_native_reader_refuses(hip_lib, tmp_path, caplog):
    """`yacht train` on an archive whose members use a compression method the one-pass native reader does not
    know (bzip2) falls back to Python's zipfile by itself -- the same manifest as on the archive sourmash wrote."""
    from yacht_amd import cli

    src = os.path.join(FX, "20_genomes_sketches.zip")
    odd = tmp_path / "bz.zip"
    with zipfile.ZipFile(src) as z, zipfile.ZipFile(odd, "w") as out:
        for i in z.infolist():
            out.writestr(i.filename, z.read(i.filename), compress_type=zipfile.ZIP_BZIP2 if i.filename.endswith(".sig.gz") else zipfile.ZIP_STORED)
    mans = []
    for name, ref in (("plain", src), ("bz", str(odd))):
        o = tmp_path / name
        o.mkdir()
        assert cli.main(["train", "--ref_file", ref, "--ksize", "31", "--prefix", "p", "--ani_thresh", "0.95", "--outdir", str(o),
                         "--num_threads", "2", "--force"]) == 0
        mans.append(pd.read_csv(o / "p_processed_manifest.tsv", sep="\t").sort_values("md5sum").reset_index(drop=True))
        assert (o / "p_intermediate_files" / "signatures" / "04212e93c2172d4df49dc5d8c2973d8b.sig").stat().st_size > 291
    assert len(mans[0]) == 20 and mans[0].equals(mans[1])


def test_train_config3_at_full_size_equals_the_genuine_reference(hip_lib):
    """BASELINE configs[3] at its real size -- 10 000 sketches, 5e7 hashes -- through yh_db_create / yh_pairwise /
    yh_train_select against what the genuine reference executable (oracle/_ref, `-t 8 -c 0.95**31`) wrote for the same
    input in the build container (tests/golden/make_golden.py cfg3): the selected ids in walk order, the three index
    statistics of main.cpp:242-244, and every one of the 20 000 pair lines (by digest)."""
    import hashlib

    from yacht_amd import synth
    from yacht_amd.engine import YH_DB_PAIRWISE_ONLY

    g = _load("golden_train_cfg3.json")
    values, offsets = synth.config4()
    assert hashlib.sha256(values.tobytes() + offsets.tobytes()).hexdigest() == g["input_sha256"], "the generator drifted"
    sizes = np.diff(offsets).astype(np.uint32)
    with RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY) as db:
        pi, pj, pc = db.pairwise(g["c"])
        stats = db.index_stats()
    lines = [format_pair_line(int(i), int(j), int(k), int(sizes[i]), int(sizes[j])) for i, j, k in zip(pi, pj, pc)]
    assert len(lines) == g["n_pair_lines"] and lines[:5] == g["first_pair_lines"]
    assert hashlib.sha256("\n".join(lines).encode()).hexdigest() == g["pair_lines_sha256"]
    assert stats == (g["stats"]["distinct"], g["stats"]["singletons"], g["stats"]["index"])
    assert train_select(sizes, pi, pj).tolist() == g["selected"]
