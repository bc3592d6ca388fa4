"""`yacht run` hot loops on the HIP engine — the counterpart of the reference's
src/yacht/hypothesis_recovery_src.py, same function names / arguments / return shapes:

  get_organisms_with_nonzero_overlap   (reference :30-113)  multisearch subprocess  -> RefDB.overlap
  get_exclusive_hashes                 (reference :116-206) Python set loops        -> RefDB.exclusive
  get_alt_mut_rate / single_hyp_test   (reference :209-306) same scipy arithmetic, plus a vectorised
                                                            hyp_test_batch used by hypothesis_recovery
  hypothesis_recovery                  (reference :309-417) same orchestration and result columns

The reference re-opens every reference .sig three times per run (multisearch, then twice in
get_exclusive_hashes); here the selected references come packed (refdb_cache: written by `yacht train`,
memory-mapped here) and stay resident in HBM as one RefDB for both steps.
"""
from __future__ import annotations

import glob
import os
import shutil
import sys
import warnings
import zipfile
from multiprocessing import Pool
from typing import Dict, List, Optional, Tuple

import numpy as np
import pandas as pd
from scipy.special import betaincinv
from scipy.stats import binom

from . import phases, refdb_cache, sigio
from .engine import RefDB, pack_csr
from .utils import decompress_all_sig_files, logger

warnings.filterwarnings("ignore")

SIG_SUFFIX = ".sig"
DB_CACHE_NAME = refdb_cache.DIR_NAME

GIVEN_COLUMNS = [
    "in_sample_est",
    "p_vals",
    "num_exclusive_kmers_to_genome",
    "num_exclusive_kmers_to_genome_coverage",
    "num_matches",
    "acceptance_threshold_with_coverage",
    "actual_confidence_with_coverage",
    "alt_confidence_mut_rate_with_coverage",
]

# one resident database per (genome dir, md5 list): get_organisms_with_nonzero_overlap builds it,
# get_exclusive_hashes finds it again
_DB_CACHE: Dict[Tuple[str, Tuple[str, ...]], RefDB] = {}
# The counts of the last fused `yacht run` call (RefDB.run_counts: overlap, and the exclusive counts
# relative to overlap > 0, ONE library call / one sample upload): get_organisms_with_nonzero_overlap makes
# the call, get_exclusive_hashes takes its answer from here when it is asked for exactly that subset --
# which is what hypothesis_recovery does (reference :361-378) -- and runs the general path otherwise.
_LAST_RUN: Dict[str, object] = {}


def _write_path_list(path: str, paths) -> None:
    """One path per line, quoted where a CSV reader needs it: byte for byte what `pd.DataFrame(paths).to_csv(path, header=False,
    index=False)` writes (the reference's call), without building a frame of 80 000 rows for it (0.1 s of a 0.57 s `yacht run`)."""
    import csv

    with open(path, "w", newline="") as f:
        csv.writer(f, lineterminator="\n").writerows([p] for p in paths)


def _read_mins(path_and_ksize) -> np.ndarray:
    """The sketch `yacht run` uses for a reference: the ONE signature of the given k-mer size in the file
    (reference: load_signature_with_ksize, utils.py:31-51 -- anything else is an error there too)."""
    path, ksize = path_and_ksize
    sigs = sigio.load_file_as_signatures(path, ksize=ksize)
    if len(sigs) != 1:
        raise ValueError(f"Expected exactly one signature with ksize {ksize} in {path}, found {len(sigs)}")
    return np.asarray(sigs[0].minhash.mins, dtype=np.uint64)


def load_reference_csr(md5sums: List[str], path_to_genome_temp_dir: str, ksize: int, num_threads: int = 1):
    """(values, offsets) of `{dir}/signatures/{md5}.sig` for every md5, in order: the packed copy
    under `{dir}/yacht_hip_db/` when it matches, else parsed from the signature files (and packed
    for the next run)."""
    cached = refdb_cache.load(path_to_genome_temp_dir, md5sums, ksize)
    if cached is not None:
        return cached
    paths = [(os.path.join(path_to_genome_temp_dir, "signatures", m + SIG_SUFFIX), ksize) for m in md5sums]
    if num_threads > 1 and len(paths) > 256:
        with Pool(min(num_threads, os.cpu_count() or 1)) as p:
            sketches = p.map(_read_mins, paths, chunksize=64)
    else:
        sketches = [_read_mins(p) for p in paths]
    sketches = [s if (s.size < 2 or bool(np.all(s[1:] > s[:-1]))) else np.unique(s) for s in sketches]
    values, offsets = pack_csr(sketches)
    refdb_cache.save(path_to_genome_temp_dir, md5sums, ksize, values, offsets)
    return values, offsets


def get_reference_db(manifest: pd.DataFrame, path_to_genome_temp_dir: str, ksize: int, num_threads: int = 1,
                     device: int = 0) -> RefDB:
    md5s = tuple(manifest["md5sum"].to_list())
    key = (os.path.abspath(path_to_genome_temp_dir), md5s)
    db = _DB_CACHE.get(key)
    if db is None:
        with phases.phase("load_reference_csr"):
            got = refdb_cache.load_any(path_to_genome_temp_dir, list(md5s), ksize)  # (round 6: `yacht train` leaves the packed form)
            if got is None or "packed" not in got:
                values, offsets = load_reference_csr(list(md5s), path_to_genome_temp_dir, ksize, num_threads)
        with phases.phase("upload_and_build_db"):
            if got is not None and "packed" in got:
                db = RefDB.from_packed(np.ascontiguousarray(got["packed"]), sizes=np.diff(got["offsets"]).astype(np.uint32), device=device)
            else:
                db = RefDB(values, offsets, device=device)
        _DB_CACHE.clear()  # one database resident at a time
        _DB_CACHE[key] = db
    return db


def release_reference_dbs() -> None:
    for db in _DB_CACHE.values():
        db.close()
    _DB_CACHE.clear()
    _LAST_RUN.clear()


def _sample_mins(sample_sig) -> np.ndarray:
    mh = sample_sig.minhash
    mins = getattr(mh, "mins", None)
    if mins is None:  # duck-typed signature (e.g. a sourmash object): hashes is a mapping
        mins = np.fromiter((int(h) for h in mh.hashes), dtype=np.uint64)
        mins.sort()
    return np.asarray(mins, dtype=np.uint64)


def get_organisms_with_nonzero_overlap(manifest: pd.DataFrame, sample_file: str, scale: int, ksize: int,
                                       num_threads: int, path_to_genome_temp_dir: str,
                                       path_to_sample_temp_dir: str, parsed_sample=None) -> List[str]:
    """Names of the manifest's organisms that share at least one hash with the sample
    (reference :30-113; there: `sourmash scripts multisearch ... -t 0`, `match_name` column)."""
    logger.info("Unzipping the sample signature zip file")
    with phases.phase("unpack_sample"):
        with zipfile.ZipFile(sample_file, "r") as z:
            z.extractall(path_to_sample_temp_dir)
        gz = glob.glob(f"{path_to_sample_temp_dir}/signatures/*.sig.gz")
        logger.info(f"Decompressing {len(gz)} .sig.gz files using {num_threads} threads.")
        decompress_all_sig_files(gz, num_threads)

    # the same two list files the reference hands to multisearch (kept for tooling that reads them)
    with phases.phase("write_list_files"):
        sample_sigs = [os.path.join(path_to_sample_temp_dir, "signatures", f)
                       for f in os.listdir(os.path.join(path_to_sample_temp_dir, "signatures"))]
        _write_path_list(os.path.join(path_to_sample_temp_dir, "sample_sig_file.txt"), sample_sigs)
        organism_sigs = [os.path.join(path_to_genome_temp_dir, "signatures", m + SIG_SUFFIX) for m in manifest["md5sum"]]
        _write_path_list(os.path.join(path_to_sample_temp_dir, "organism_sig_file.txt"), organism_sigs)

    # parsed_sample: the caller's already-parsed signature of this very file (hypothesis_recovery holds it); a
    # 10^6-hash sketch takes 0.2-0.3 s to parse and the archive has been read once already
    if parsed_sample is not None and len(sample_sigs) == 1:
        sigs = [parsed_sample]
    else:
        sigs = []
        for path in sample_sigs:
            sigs += sigio.load_file_as_signatures(path, ksize=ksize)
    result_csv = os.path.join(path_to_sample_temp_dir, "sample_multisearch_result.csv")
    rows = []
    if sigs and len(manifest):
        db = get_reference_db(manifest, path_to_genome_temp_dir, ksize, num_threads)
        names = manifest["organism_name"].to_list()
        md5s = manifest["md5sum"].to_list()
        sizes = db.sizes
        for q in sigs:
            mins = np.ascontiguousarray(q.minhash.mins, dtype=np.uint64)
            with phases.phase("run_counts"):
                overlap, n_excl, n_match = db.run_counts(mins)  # the fused step: R1 + R2 for the subset overlap > 0
            _LAST_RUN.clear()
            _LAST_RUN.update(db=db, mins=mins, overlap=overlap, n_excl=n_excl, n_match=n_match)
            with phases.phase("multisearch_rows"):
                nq = len(q.minhash)
                q_name, q_md5 = q.name, q.md5sum()  # (the md5 of a 10^6-hash sketch takes ~30 ms: once, not once per row)
                hit = np.flatnonzero(overlap)
                ov = overlap[hit].astype(np.float64)
                sz = np.asarray(sizes, dtype=np.float64)[hit]
                cont = ov / nq if nq else np.zeros_like(ov)
                for k, j in enumerate(hit.tolist()):
                    rows.append((q_name, q_md5, names[j], md5s[j], float(cont[k]), float(max(cont[k], ov[k] / sz[k])),
                                 float(ov[k] / (nq + sz[k] - ov[k])), int(ov[k])))
    if not rows:
        open(result_csv, "w").close()
        print("ERROR: Multisearch file is empty. Likely there are no microorganisms in your sample, or something went wrong",
              flush=True)
        sys.exit(0)
    with phases.phase("write_multisearch_csv"):
        res = pd.DataFrame(rows, columns=["query_name", "query_md5", "match_name", "match_md5", "containment",
                                          "max_containment", "jaccard", "intersect_hashes"])
        res.to_csv(result_csv, index=False)
        return res.drop_duplicates().reset_index(drop=True)["match_name"].to_list()


def get_exclusive_hashes(manifest: pd.DataFrame, nontrivial_organism_names: List[str], sample_sig, ksize: int,
                         path_to_genome_temp_dir: str) -> Tuple[List[Tuple[int, int]], pd.DataFrame]:
    """For the manifest rows whose organism_name is listed: (#hashes no other listed row has,
    #of those present in the sample), and the sub-manifest (reference :116-206)."""
    selected = manifest["organism_name"].isin(nontrivial_organism_names).to_numpy()
    sub_manifest = manifest.loc[selected, :].reset_index(drop=True)
    if not selected.any():
        return [], sub_manifest
    db = get_reference_db(manifest, path_to_genome_temp_dir, ksize)
    mins = np.ascontiguousarray(_sample_mins(sample_sig), dtype=np.uint64)
    last = _LAST_RUN
    if (last.get("db") is db and np.array_equal(last["mins"], mins) and np.array_equal(selected, last["overlap"] > 0)):
        n_excl, n_match = last["n_excl"], last["n_match"]  # asked for the subset the fused call answered
    else:  # caller-supplied name list (e.g. duplicate organism names pull in references without overlap)
        n_excl, n_match = db.exclusive(selected, mins)
    rows = np.flatnonzero(selected)
    return [(int(n_excl[j]), int(n_match[j])) for j in rows], sub_manifest


def get_alt_mut_rate(nu: int, thresh: int, ksize: int, significance: float = 0.99) -> float:
    """Mutation rate at which the false-positive rate would equal `significance`
    (reference :209-230): inverse regularized incomplete beta, NaN -> -1."""
    mut = 1 - (1 - betaincinv(nu - thresh, 1 + thresh, significance)) ** (1 / ksize)
    return -1.0 if np.isnan(mut) else mut


def hyp_test_batch(n_excl, n_match, ksize: int, significance: float = 0.99, ani_thresh: float = 0.95,
                   min_coverage: float = 1):
    """single_hyp_test over arrays: one scipy call per quantity instead of one process-pool task
    per organism (reference :393-408).  Returns the eight result columns as arrays."""
    e = np.asarray(n_excl, dtype=np.int64)
    m = np.asarray(n_match, dtype=np.int64)
    p = ani_thresh ** ksize
    n_cov = np.array([int(x * min_coverage) for x in e.tolist()], dtype=np.int64)  # int(): truncation, as the reference
    thr = binom.ppf(1 - significance, n_cov, p)
    conf = 1 - binom.cdf(thr, n_cov, p)
    with np.errstate(all="ignore"):
        alt = 1 - (1 - betaincinv(n_cov - thr, 1 + thr, significance)) ** (1 / ksize)
    alt = np.where(np.isnan(alt), -1.0, alt)
    p_val = np.where(m <= n_cov, binom.cdf(m, n_cov, p), 1.0)
    present = (m >= thr) & (m != 0)
    return present, p_val, e, n_cov, m, thr, conf, alt


def hyp_test_native(n_excl, n_match, ksize: int, significance: float = 0.99, ani_thresh: float = 0.95,
                    min_coverage: float = 1):
    """hyp_test_batch without scipy: yh_hyp_test of the C ABI (host C++: exact log-space binomial tails and a Newton
    solve of the regularized incomplete beta).  Same eight columns; decisions, integer columns and thresholds equal
    scipy's, floating columns to ~1e-13 relative (tests/test_hyp_native.py).  What a non-Python caller binds, and what
    makes in_sample_est independent of the scipy build of the machine; YACHT_HYP_NATIVE=1 makes hypothesis_recovery
    use it."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    e = np.ascontiguousarray(n_excl, dtype=np.uint32)
    m = np.ascontiguousarray(n_match, dtype=np.uint32)
    n = int(e.size)
    present = np.zeros(n, dtype=np.uint8)
    p_val, thr, conf, alt = (np.zeros(n, dtype=np.float64) for _ in range(4))
    n_cov = np.zeros(n, dtype=np.uint32)
    ptr = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    _lib.check(lib.yh_hyp_test(n, ptr(e), ptr(m), int(ksize), float(significance), float(ani_thresh), float(min_coverage),
                               ptr(present), ptr(p_val), ptr(n_cov), ptr(thr), ptr(conf), ptr(alt)))
    return present.astype(bool), p_val, e.astype(np.int64), n_cov.astype(np.int64), m.astype(np.int64), thr, conf, alt


def single_hyp_test(exclusive_hashes_info_org: Tuple[int, int], ksize: int, significance: float = 0.99,
                    ani_thresh: float = 0.95, min_coverage: int = 1):
    """Binomial presence test for one organism (reference :233-306); returns the same 8-tuple:
    (in_sample_est, p_val, n_exclusive, n_exclusive_at_coverage, n_matches, threshold, confidence,
    alt_mut_rate)."""
    n_excl, n_match = exclusive_hashes_info_org
    p = ani_thresh ** ksize
    n_cov = int(n_excl * min_coverage)
    thr = binom.ppf(1 - significance, n_cov, p)
    conf = 1 - binom.cdf(thr, n_cov, p)
    alt = get_alt_mut_rate(n_cov, thr, ksize, significance=significance)
    p_val = binom.cdf(n_match, n_cov, p) if n_match <= n_cov else 1.0
    present = (n_match >= thr) and (n_match != 0)
    return present, p_val, n_excl, n_cov, n_match, thr, conf, alt


def hypothesis_recovery(manifest: pd.DataFrame, sample_info_set, path_to_genome_temp_dir: str,
                        min_coverage_list: List[float], scale: int, ksize: int, significance: float = 0.99,
                        ani_thresh: float = 0.95, num_threads: int = 16):
    """One DataFrame per min_coverage: the sub-manifest of overlapping organisms joined with the
    eight hypothesis-test columns (reference :309-417)."""
    sample_file, sample_sig = sample_info_set
    sample_dir = os.path.dirname(sample_file)
    sample_name = os.path.basename(sample_file).replace(".sig.zip", "")
    path_to_sample_temp_dir = os.path.join(sample_dir, f"sample_{sample_name}_intermediate_files")
    with phases.phase("sample_temp_dir"):
        if os.path.exists(path_to_sample_temp_dir):
            logger.info(f"Removing existing temporary directory: {path_to_sample_temp_dir}")
            shutil.rmtree(path_to_sample_temp_dir)
        os.makedirs(path_to_sample_temp_dir)

    names = get_organisms_with_nonzero_overlap(manifest, sample_file, scale, ksize, num_threads,
                                               path_to_genome_temp_dir, path_to_sample_temp_dir,
                                               parsed_sample=sample_sig if hasattr(sample_sig, "minhash") else None)
    with phases.phase("exclusive_hashes_bookkeeping"):
        info, manifest = get_exclusive_hashes(manifest, names, sample_sig, ksize, path_to_genome_temp_dir)
        n_excl = np.array([x[0] for x in info], dtype=np.int64)
        n_match = np.array([x[1] for x in info], dtype=np.int64)

    out = []
    for min_coverage in min_coverage_list:
        logger.info(f"Computing hypothesis recovery for min_coverage={min_coverage}")
        with phases.phase("hypothesis_tests"):
            test = hyp_test_native if os.environ.get("YACHT_HYP_NATIVE") == "1" else hyp_test_batch
            cols = test(n_excl, n_match, ksize, significance, ani_thresh, min_coverage)
        with phases.phase("assemble_frames"):
            results = pd.DataFrame({name: col for name, col in zip(GIVEN_COLUMNS, cols)}, columns=GIVEN_COLUMNS)
            results["in_sample_est"] = results["in_sample_est"].astype(bool)
            manifest["min_coverage"] = min_coverage
            out.append(pd.concat([manifest, results], axis=1))
    return out
