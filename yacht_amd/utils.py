"""Host-side helpers of the hot path — the counterpart of the reference's src/yacht/utils.py
(:31-221) with the same function names, arguments, return shapes and error behaviour, so that
callers (and tests) written against the reference read the same here.

What differs is underneath: signatures are read by yacht_amd.sigio (no sourmash), and
run_yacht_train_core calls the HIP engine through the C ABI instead of shelling out to the
`run_yacht_train_core` executable (reference utils.py:143-147).
"""
from __future__ import annotations

import gzip
import logging
import os
import shutil
import sys
from glob import glob
from multiprocessing import Pool
from typing import Dict, List, Optional, Tuple

import numpy as np
import pandas as pd

from . import phases, refdb_cache, sigio, train_core

logger = logging.getLogger("yacht_amd")
if not logger.handlers:
    _h = logging.StreamHandler(sys.stdout)
    _h.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s", "%Y-%m-%d %H:%M:%S"))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)
    logger.propagate = False

COL_NOT_FOUND_ERROR = "Column not found: {}"
FILE_LOCATION = os.path.dirname(os.path.realpath(__file__))
__version__ = "1.4.0"  # the reference interface version mirrored here


class _NativeMinHash(sigio.MinHash):
    """The sketch of a one-signature archive as the library's scanner read it (yh_zip_sig_ingest): mins, md5, mean
    abundance, scaled -- what `yacht run` uses of a sample (run_YACHT.py:150-165).  The per-hash abundances, which nothing on
    this path reads, are parsed by the general reader only if somebody asks for them."""

    def __init__(self, mins, ksize, scaled, mean_abundance, md5, filename):
        self.mins = np.ascontiguousarray(mins, dtype=np.uint64)
        self.ksize = int(ksize)
        self._scaled = int(scaled)
        self.max_hash = sigio.max_hash_for_scaled(self._scaled)
        self.seed = 42
        self.moltype = "DNA"
        self.num = 0
        self._mean_abundance = mean_abundance
        self._md5 = md5
        self._filename = filename
        self._abundances = None

    @property
    def scaled(self) -> int:
        return self._scaled

    @property
    def abundances(self):
        if self._mean_abundance is None:
            return None
        if self._abundances is None:
            self._abundances = sigio.load_file_as_signatures(self._filename, ksize=self.ksize)[0].minhash.abundances
        return self._abundances

    @property
    def mean_abundance(self):
        return self._mean_abundance

    def md5sum(self) -> str:
        return self._md5


def _load_single_signature_native(filename: str, ksize: int):
    """The signature of a sourmash .zip that holds exactly ONE signature file whose first signature is the one of this k-mer
    size -- a sample -- through the library's scanner (a 10^6-hash sample: 0.03 s instead of 0.3 s of json.loads + an md5 over a
    million decimal strings); None for every other shape, which the general reader below handles (and complains about)."""
    import ctypes as C

    from . import _lib

    try:
        if not str(filename).endswith(".zip"):
            return None
        lib = _lib.load()
        h = C.c_void_p()
        if lib.yh_zip_sig_ingest(os.fsencode(filename), None, int(ksize), 2, C.byref(h)) != _lib.YH_OK:
            return None
    except Exception:
        return None
    hb = C.c_void_p()
    try:
        cnt = C.c_uint64(0)
        _lib.check(lib.yh_sig_meta_count(h, C.byref(cnt)))
        if cnt.value != 1:
            return None
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        status, n_hashes, scaled = np.zeros(1, np.uint8), np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        mean_ab, has_ab, md5, name_off = np.zeros(1, np.float64), np.zeros(1, np.uint8), np.zeros(33, np.uint8), np.zeros(2, np.uint64)
        _lib.check(lib.yh_sig_meta_get(h, vp(status), vp(n_hashes), vp(scaled), vp(mean_ab), vp(has_ab), vp(md5), vp(name_off)))
        if int(status[0]) != 0 or not (int(has_ab[0]) & 2):
            return None
        names_buf = np.zeros(max(int(name_off[1]), 1), np.uint8)
        _lib.check(lib.yh_sig_meta_names(h, vp(names_buf)))
        _lib.check(lib.yh_sig_meta_take_batch(h, C.byref(hb)))
        st = np.zeros(1, np.uint8)
        _lib.check(lib.yh_sig_batch_status(hb, vp(st)))
        off = np.zeros(2, np.uint64)
        _lib.check(lib.yh_sig_batch_sizes(hb, vp(off)))
        if int(st[0]) != 0 or int(off[1]) != int(n_hashes[0]) or int(off[1]) == 0:
            return None
        mins = np.empty(int(off[1]), np.uint64)
        _lib.check(lib.yh_sig_batch_values(hb, vp(mins)))
        name = names_buf.tobytes()[: int(name_off[1])].decode("utf-8", "replace")
        mh = _NativeMinHash(mins, ksize, int(scaled[0]), float(mean_ab[0]) if int(has_ab[0]) & 1 else None,
                            md5.tobytes()[:32].decode("ascii"), filename)
        return sigio.Signature(mh, name=name, filename="")
    except Exception:
        return None
    finally:
        if hb.value:
            lib.yh_sig_batch_destroy(hb)
        lib.yh_sig_meta_destroy(h)


def load_signature_with_ksize(filename: str, ksize: int) -> sigio.Signature:
    """Exactly one non-empty signature of the given k-mer size, else ValueError
    (reference utils.py:31-51, same messages)."""
    fast = _load_single_signature_native(filename, ksize)
    if fast is not None:
        return fast
    sketches = sigio.load_file_as_signatures(filename, ksize=ksize)
    if len(sketches) != 1:
        raise ValueError(f"Expected exactly one signature with ksize {ksize} in {filename}, found {len(sketches)}")
    if len(sketches[0].minhash) == 0:
        raise ValueError(
            "Empty sketch in signature. This may be due to too high of a scale factor, please reduce it, eg. --scaled=1, and try again."
        )
    return sketches[0]


def get_num_kmers(minhash_mean_abundance: Optional[float], minhash_hashes_len: int, minhash_scaled: int,
                  scale: bool = True) -> int:
    """Estimated total k-mers: mean abundance (1 when not tracked) x sketch size [x scaled]
    (reference utils.py:54-75)."""
    num_kmers = minhash_mean_abundance * minhash_hashes_len if minhash_mean_abundance else minhash_hashes_len
    if scale:
        num_kmers *= minhash_scaled
    return int(np.round(num_kmers))


def check_file_existence(file_path: str, error_description: str) -> None:
    if not os.path.exists(file_path):
        raise ValueError(error_description)


def get_info_from_single_sig(sig_file: str, ksize: int):
    """(path, name, md5sum, mean abundance, sketch size, scaled) or None with a warning
    (reference utils.py:89-110)."""
    try:
        sig = load_signature_with_ksize(sig_file, ksize)
        return (sig_file, sig.name, sig.md5sum(), sig.minhash.mean_abundance, len(sig.minhash), sig.minhash.scaled)
    except Exception:
        logger.warning(f"CANNOT extract the relevant info from the signature file: {sig_file}")
        return None


def _sig_meta_native(paths: List[str], ksize: int, num_threads: int, keep_sketches: bool = False):
    """(status, n_hashes, scaled, mean_abundance, has_abundance, md5s, names) of every file through the library's
    threaded reader (yh_sig_meta_*: a JSON scan + md5 in C++ instead of a Python object per hash).
    keep_sketches: the pass also keeps what the train core reads from the same files and leaves it with
    train_core.offer_parsed_sketches, so that `yacht train` reads its 85 205 files once, not twice."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    h = C.c_void_p()
    read = lib.yh_sig_meta_read_keep if keep_sketches else lib.yh_sig_meta_read
    _lib.check(read(arr, n, int(ksize), max(1, int(num_threads)), C.byref(h)))
    try:
        if keep_sketches:
            from . import train_core

            hb = C.c_void_p()
            _lib.check(lib.yh_sig_meta_take_batch(h, C.byref(hb)))
            train_core.offer_parsed_sketches(list(paths), hb)  # (owns the batch handle from here on)
        status = np.zeros(max(n, 1), np.uint8)
        n_hashes = np.zeros(max(n, 1), np.uint64)
        scaled = np.zeros(max(n, 1), np.uint64)
        mean_ab = np.zeros(max(n, 1), np.float64)
        has_ab = np.zeros(max(n, 1), np.uint8)
        md5 = np.zeros(max(n, 1) * 33, np.uint8)
        name_off = np.zeros(n + 1, np.uint64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(lib.yh_sig_meta_get(h, vp(status), vp(n_hashes), vp(scaled), vp(mean_ab), vp(has_ab), vp(md5), vp(name_off)))
        names_buf = np.zeros(max(int(name_off[-1]), 1), np.uint8)
        _lib.check(lib.yh_sig_meta_names(h, vp(names_buf)))
    finally:
        lib.yh_sig_meta_destroy(h)
    raw = names_buf.tobytes()
    names = [raw[int(name_off[i]):int(name_off[i + 1])].decode("utf-8", "replace") for i in range(n)]
    md5s = [md5[33 * i:33 * i + 32].tobytes().decode("ascii") for i in range(n)]
    return status[:n], n_hashes[:n], scaled[:n], mean_ab[:n], has_ab[:n], md5s, names


def collect_signature_info(num_threads: int, ksize: int, path_to_temp_dir: str) -> Dict[str, Tuple]:
    """name -> (md5sum, mean abundance, sketch size, scaled, path) for every file under
    {path_to_temp_dir}/signatures, in os.listdir order (reference utils.py:201-221).  Same records and the same
    warning for a file without exactly one non-empty signature of this k-mer size as get_info_from_single_sig;
    the files are read by the library's threaded scanner, and the few it defers (unsorted mins, non-integer
    fields) by get_info_from_single_sig itself."""
    sig_dir = os.path.join(path_to_temp_dir, "signatures")
    paths = [os.path.join(sig_dir, f) for f in os.listdir(sig_dir)]
    status, n_hashes, scaled, mean_ab, has_ab, md5s, names = _sig_meta_native(paths, ksize, num_threads, keep_sketches=True)
    out: Dict[str, Tuple] = {}
    for i, path in enumerate(paths):
        st = int(status[i])
        if st == 0:
            out[names[i]] = (md5s[i], float(mean_ab[i]) if has_ab[i] & 1 else None, int(n_hashes[i]), int(scaled[i]), path)
        elif st == 5:  # a shape the scanner leaves to the general reader
            rec = get_info_from_single_sig(path, ksize)
            if rec:
                out[rec[1]] = (rec[2], rec[3], rec[4], rec[5], rec[0])
        else:
            logger.warning(f"CANNOT extract the relevant info from the signature file: {path}")
    return out


class BackgroundExtraction:
    """The members of a .zip database being written to a directory by native threads (yh_zip_extract_start); wait()
    joins and raises what failed.  `yacht train` starts one, reads the archive without writing anything, runs the
    comparison, and waits at the end: the 85 205 file creations of a GTDB database are the directory's lock, not CPU."""

    def __init__(self, zip_path: str, out_dir: str, threads: int):
        import ctypes as C

        from . import _lib

        self._lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.yh_zip_extract_start(os.fsencode(zip_path), os.fsencode(out_dir), max(1, int(threads)), C.byref(self._h)))

    def wait(self) -> int:
        import ctypes as C

        from . import _lib

        if self._h is None or not self._h.value:
            return 0
        n = C.c_uint64(0)
        h, self._h = self._h, None
        _lib.check(self._lib.yh_zip_extract_wait(h, C.byref(n)))
        return int(n.value)

    def __del__(self):
        try:
            self.wait()
        except Exception:
            pass


def ingest_zip_database(zip_path: str, path_to_temp_dir: str, ksize: int, num_threads: int, write_files: bool = True,
                        background: Optional[BackgroundExtraction] = None) -> Dict[str, Tuple]:
    """`yacht train`'s unzip + gunzip + metadata passes in ONE pass over the archive (yh_zip_sig_ingest; reference
    make_training_data_from_sketches.py:107-133, utils.py:201-221, :499-509): the same dictionary collect_signature_info
    returns -- name -> (md5sum, mean abundance, sketch size, scaled, path) -- with the signature members in the archive's
    own order, and (write_files) the same files left in the working directory: every member as the reference's passes
    leave it, a `.sig.gz` as the `.sig` it inflates to.  The sketches the train core needs are kept from the same pass
    (train_core.offer_parsed_sketches).  Shapes the native scanner defers (status 5) go through the general reader, which
    needs the file: with write_files=False such a member is an error."""
    import ctypes as C

    from . import _lib, train_core

    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.yh_zip_sig_ingest(os.fsencode(zip_path), os.fsencode(path_to_temp_dir) if write_files else None, int(ksize),
                                     max(1, int(num_threads)), C.byref(h)))
    try:
        cnt = C.c_uint64(0)
        _lib.check(lib.yh_sig_meta_count(h, C.byref(cnt)))
        n = int(cnt.value)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        p_off = np.zeros(n + 1, np.uint64)
        _lib.check(lib.yh_sig_meta_paths(h, vp(p_off), None))
        p_buf = np.zeros(max(int(p_off[-1]), 1), np.uint8)
        _lib.check(lib.yh_sig_meta_paths(h, vp(p_off), vp(p_buf)))
        raw_p = p_buf.tobytes()
        paths = [os.path.join(path_to_temp_dir, os.fsdecode(raw_p[int(p_off[i]):int(p_off[i + 1])])) for i in range(n)]
        hb = C.c_void_p()
        _lib.check(lib.yh_sig_meta_take_batch(h, C.byref(hb)))
        train_core.offer_parsed_sketches(list(paths), hb)  # (owns the batch handle from here on)
        status = np.zeros(max(n, 1), np.uint8)
        n_hashes = np.zeros(max(n, 1), np.uint64)
        scaled = np.zeros(max(n, 1), np.uint64)
        mean_ab = np.zeros(max(n, 1), np.float64)
        has_ab = np.zeros(max(n, 1), np.uint8)
        md5 = np.zeros(max(n, 1) * 33, np.uint8)
        name_off = np.zeros(n + 1, np.uint64)
        _lib.check(lib.yh_sig_meta_get(h, vp(status), vp(n_hashes), vp(scaled), vp(mean_ab), vp(has_ab), vp(md5), vp(name_off)))
        names_buf = np.zeros(max(int(name_off[-1]), 1), np.uint8)
        _lib.check(lib.yh_sig_meta_names(h, vp(names_buf)))
    finally:
        lib.yh_sig_meta_destroy(h)
    raw = names_buf.tobytes()
    md5_raw = md5.tobytes()
    out: Dict[str, Tuple] = {}
    for i, path in enumerate(paths):
        st = int(status[i])
        if st == 0:
            name = raw[int(name_off[i]):int(name_off[i + 1])].decode("utf-8", "replace")
            out[name] = (md5_raw[33 * i:33 * i + 32].decode("ascii"), float(mean_ab[i]) if has_ab[i] & 1 else None, int(n_hashes[i]),
                         int(scaled[i]), path)
        elif st == 5 and (write_files or background is not None):
            if background is not None:
                background.wait()  # (the general reader opens the file)
            rec = get_info_from_single_sig(path, ksize)
            if rec:
                out[rec[1]] = (rec[2], rec[3], rec[4], rec[5], rec[0])
        else:
            logger.warning(f"CANNOT extract the relevant info from the signature file: {path}")
    return out


def _gunzip_one(path: str) -> None:
    with gzip.open(path, "rb") as f_in, open(path[: -len(".gz")], "wb") as f_out:
        shutil.copyfileobj(f_in, f_out)
    os.remove(path)


def decompress_all_sig_files(sig_files: List[str], num_threads: int) -> None:
    """gunzip every *.sig.gz next to itself and delete the .gz (reference utils.py:499-509): the library's
    threaded zlib pass; a file it could not handle goes through Python's gzip, which raises as the reference does."""
    if not sig_files:
        return
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    n = len(sig_files)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in sig_files])
    status = np.ones(n, np.uint8)
    _lib.check(lib.yh_gunzip_files(arr, n, max(1, int(num_threads)), status.ctypes.data_as(C.c_void_p)))
    for i in np.flatnonzero(status):
        _gunzip_one(sig_files[int(i)])


def run_yacht_train_core(num_threads: int, ani_thresh: float, ksize: int, path_to_temp_dir: str,
                         sig_info_dict: Dict[str, Tuple], num_genome_threshold: int = 1000000,
                         device: int = 0) -> pd.DataFrame:
    """Find the references related above `ani_thresh` and keep one per neighbourhood
    (reference utils.py:112-197).  Same files are left behind: training_sig_files.tsv,
    selected_result.tsv, comparison_files/<pass>_<thread>.txt; same manifest columns returned."""
    sig_dir = os.path.join(path_to_temp_dir, "signatures")
    # (the reference's file list is os.listdir's order -- whatever the file system gives.  When the ingest pass has the
    # database's sketches in hand, the list is ITS list in its order -- the archive's -- and nothing is read twice; the
    # files themselves may still be on their way to the directory: utils.BackgroundExtraction.)
    offered = train_core.parsed_paths()
    # (the offer is a process-wide leftover of the LAST ingest -- it is this call's list only when every path lies
    # in THIS working directory's signatures/; anything else -- an ingest whose run aborted, a direct caller of this public
    # function -- is dropped and the directory listed, as the reference does)
    sig_root = os.path.join(os.path.abspath(sig_dir), "")
    if offered is not None and offered and all(os.path.abspath(p_).startswith(sig_root) for p_ in offered):
        sig_files = list(offered)
    else:
        if offered is not None:
            train_core.drop_parsed_sketches()
        sig_files = [os.path.join(sig_dir, f) for f in os.listdir(sig_dir)]
    sig_files_path = os.path.join(path_to_temp_dir, "training_sig_files.tsv")
    with phases.phase("write_file_list"):
        import csv

        with open(sig_files_path, "w", newline="") as f:  # (what pd.DataFrame(sig_files).to_csv(..., header=False, index=False) writes)
            csv.writer(f, lineterminator="\n").writerows([p] for p in sig_files)

    containment_thresh = ani_thresh ** ksize
    total = len(sig_files)
    passes = 1 if total <= num_genome_threshold else int(total / num_genome_threshold) + 1
    selected_path = os.path.join(path_to_temp_dir, "selected_result.tsv")
    logger.info(f"Running comparison on device {device}: -t {num_threads} -c {containment_thresh} -p {passes} "
                f"{sig_files_path} {path_to_temp_dir} {selected_path}")
    try:
        core = train_core.run(sig_files_path, path_to_temp_dir, selected_path, threads=num_threads, passes=passes,
                              containment_threshold=containment_thresh, device=device)
    except Exception as exc:  # the reference raises ValueError on a non-zero exit code
        raise ValueError(f"Error running comparison algorithm: {exc}") from exc

    # The selected sketches, packed in manifest order (= the order of sig_info_dict, the archive's) for `yacht run`
    # (refdb_cache): their slices go to disk in a thread of their own, straight from the core's arrays, while the manifest is
    # put together here; nothing refers to the files until the sizes below have been checked.
    os.makedirs(os.path.join(path_to_temp_dir, "comparison_files"), exist_ok=True)
    for file in glob(os.path.join(path_to_temp_dir, "*.txt")):
        shutil.move(file, os.path.join(path_to_temp_dir, "comparison_files"))
    with phases.phase("rows_of_the_selected"):
        selected_paths = set(core["paths"][int(g)] for g in core["selected"])
        row_of_path = {p: i for i, p in enumerate(core["paths"])}
        kept_names = [name for name, info in sig_info_dict.items() if info[-1] in selected_paths]
        keep = [row_of_path[sig_info_dict[name][-1]] for name in kept_names]
    pending = refdb_cache.save_subset_async(path_to_temp_dir, ksize, core["values"], core["offsets"], keep, packed=core.get("packed"))
    with phases.phase("manifest_of_the_selected"):
        rows = []
        for name in kept_names:
            md5sum, mean_abund, n_hashes, scaled, _path = sig_info_dict[name]
            rows.append((name, md5sum, n_hashes, get_num_kmers(mean_abund, n_hashes, scaled, False), scaled))
        manifest = pd.DataFrame(rows, columns=["organism_name", "md5sum", "num_unique_kmers_in_genome_sketch",
                                               "num_total_kmers_in_genome_sketch", "genome_scale_factor"])
    # The train core reads record 0 / signature 0 of every file, whatever its k-mer size (as the reference's does,
    # main.cpp:62-84); the manifest's sizes come from the signature of THIS k-mer size.  Only when the two agree
    # is the packed copy what `yacht run` would load itself -- otherwise it is dropped and run reads the files.
    with phases.phase("write_packed_db"):
        if np.array_equal(np.diff(pending.out_offsets).astype(np.int64), manifest["num_unique_kmers_in_genome_sketch"].to_numpy(dtype=np.int64)):
            pending.publish(manifest["md5sum"].to_list())
        else:
            pending.discard()
            logger.warning("sketch sizes of the train core and of the ksize-filtered signatures differ: no packed copy written")
    return manifest
