"""In-process `run_yacht_train_core`: the reference executable's contract (src/cpp/main.cpp) with
the comparison done by the HIP engine.

    run(file_list, working_directory, output_filename, threads, passes, containment_threshold)

reads the same inputs (a text file with one sketch path per line, each sketch a sourmash JSON
whose FIRST signature's "mins" are used, main.cpp:74-78,127-139) and leaves the same outputs:

  * `<output_filename>`: the selected sketch paths, one per line, in the greedy walk's order
    (main.cpp:409-418);
  * `<working_directory>/<pass>_<thread id, 3 digits>.txt`: lines `i,j,jaccard,C(i in j),C(j in i)`
    for every kept ordered pair, rows split over passes and threads exactly as main.cpp:318-349
    splits them, numbers printed as C++ iostream prints doubles (6 significant digits).

Nothing here computes intersections on the CPU: counts come from RefDB.pairwise and the selection
from the library's yh_train_select.
"""
from __future__ import annotations

import math
import os
import sys
from multiprocessing import Pool
from typing import List, Sequence, Tuple

import numpy as np

from . import phases, sigio
from .engine import YH_DB_PAIRWISE_ONLY, RefDB, pack_csr, train_select


def format_pair_line(i: int, j: int, count: int, size_i: int, size_j: int) -> str:
    """One comparison line (main.cpp:296-305); %g is iostream's default double formatting."""
    jaccard = 1.0 * count / (size_i + size_j - count)
    c_ij = 1.0 * count / size_i
    c_ji = 1.0 * count / size_j
    return "%d,%d,%g,%g,%g" % (i, j, jaccard, c_ij, c_ji)


def row_ranges(n: int, threads: int, passes: int) -> List[Tuple[int, int, int, int]]:
    """(pass, thread, row_begin, row_end) in the reference's split (main.cpp:318-349)."""
    out = []
    per_pass = math.ceil(1.0 * n / passes) if passes else n
    for p in range(passes):
        start = p * per_pass
        end = n if p == passes - 1 else (p + 1) * per_pass
        rows = end - start
        chunk = rows // threads if rows > 0 else 0
        for t in range(threads):
            a = start + t * chunk
            b = end if t == threads - 1 else start + (t + 1) * chunk
            out.append((p, t, a, b))
    return out


def read_sketch_list(file_list: str) -> List[str]:
    with open(file_list) as f:
        return [line.rstrip("\n") for line in f]


def _read_one(path: str) -> np.ndarray:
    mins = sigio.read_mins_first_signature(path)
    if mins.size > 1 and not bool(np.all(mins[1:] > mins[:-1])):
        mins = np.unique(mins)  # sourmash writes ascending, unique mins; tolerate other writers
    return mins


def read_sketches(paths: Sequence[str], threads: int = 1) -> List[np.ndarray]:
    """The sketches as a list (python reader: json per file).  `run` uses read_sketches_csr."""
    if threads > 1 and len(paths) > 256:
        with Pool(min(threads, os.cpu_count() or 1)) as p:
            return p.map(_read_one, paths, chunksize=64)
    return [_read_one(p) for p in paths]


# Sketches the metadata pass in front of the core has parsed already (utils.collect_signature_info ->
# yh_sig_meta_read_keep): the path list they belong to and the library's batch handle.  Used once, by the next
# read_sketches_csr over exactly that list; anything else reads the files.
_PARSED = {}


def offer_parsed_sketches(paths: List[str], batch_handle) -> None:
    drop_parsed_sketches()
    _PARSED.update(paths=paths, handle=batch_handle)


def parsed_paths():
    """The path list of the sketches an ingest pass left here (None: nothing offered)."""
    return list(_PARSED["paths"]) if _PARSED.get("handle") is not None else None


def drop_parsed_sketches() -> None:
    h = _PARSED.pop("handle", None)
    _PARSED.clear()
    if h is not None:
        from . import _lib

        _lib.load().yh_sig_batch_destroy(h)


def read_sketches_csr(paths: Sequence[str], threads: int = 1):
    """(values, offsets) of all files through the library's threaded reader (yh_sig_batch_*: the
    counterpart of src/cpp/main.cpp:89-124).  A file that cannot be opened is an empty sketch and the
    reference's message is printed (main.cpp:66-71); a file that does not parse raises -- the reference's
    process dies on it (json::parse throws, main.cpp:73) and its Python caller turns that into ValueError."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    if _PARSED.get("handle") is not None and _PARSED.get("paths") == list(paths):
        h = _PARSED.pop("handle")  # the very files, parsed a moment ago by the metadata pass
        _PARSED.clear()
    else:
        drop_parsed_sketches()
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        h = C.c_void_p()
        _lib.check(lib.yh_sig_batch_read(arr, len(paths), max(1, int(threads)), C.byref(h)))
    try:
        status = np.zeros(max(len(paths), 1), dtype=np.uint8)
        _lib.check(lib.yh_sig_batch_status(h, status.ctypes.data_as(C.c_void_p)))
        for _ in np.flatnonzero(status[:len(paths)] == 1):
            print("Could not open the file!", file=sys.stderr)
        bad = np.flatnonzero(status[:len(paths)] == 2)
        if bad.size:
            raise ValueError(f"{bad.size} signature file(s) could not be parsed, first: {paths[int(bad[0])]}")
        offsets = np.zeros(len(paths) + 1, dtype=np.uint64)
        _lib.check(lib.yh_sig_batch_sizes(h, offsets.ctypes.data_as(C.c_void_p)))
        values = np.empty(int(offsets[-1]), dtype=np.uint64)  # (every element is written by the library)
        _lib.check(lib.yh_sig_batch_values(h, values.ctypes.data_as(C.c_void_p)))
    finally:
        lib.yh_sig_batch_destroy(h)
    return values, offsets


def read_sketches_packed(paths: Sequence[str], threads: int = 1):
    """(packed, offsets) of all files: the library's threaded reader as read_sketches_csr uses it, and the sketches packed straight
    from the parsed files (yh_sig_batch_pack: ~5.7 instead of 8 bytes per hash, no CSR in between) -- what RefDB.from_packed
    uploads and refdb_cache writes for `yacht run`.  (None, None) when some file's mins are not strictly ascending: the caller
    takes read_sketches_csr, which is the path that says what is wrong (or, for the Python reader, repairs it)."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    if _PARSED.get("handle") is not None and _PARSED.get("paths") == list(paths):
        h = _PARSED["handle"]  # (left in place: a fallback to read_sketches_csr takes it from there)
        own = False
    else:
        drop_parsed_sketches()
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        h = C.c_void_p()
        _lib.check(lib.yh_sig_batch_read(arr, len(paths), max(1, int(threads)), C.byref(h)))
        offer_parsed_sketches(list(paths), h)  # (so that a fallback does not read the files again)
        own = True
    status = np.zeros(max(len(paths), 1), dtype=np.uint8)
    _lib.check(lib.yh_sig_batch_status(h, status.ctypes.data_as(C.c_void_p)))
    if np.any(status[:len(paths)] == 2):
        return None, None  # (read_sketches_csr raises the reference's error for it)
    need = C.c_uint64(0)
    rc = lib.yh_sig_batch_pack(h, None, 0, C.byref(need), max(1, int(threads)))
    if rc == _lib.YH_ERR_UNSORTED:
        return None, None
    _lib.check(rc)
    packed = np.zeros((int(need.value) + 7) // 8, dtype=np.uint64)
    _lib.check(lib.yh_sig_batch_pack(h, packed.ctypes.data_as(C.c_void_p), packed.nbytes, C.byref(need), max(1, int(threads))))
    for _ in np.flatnonzero(status[:len(paths)] == 1):
        print("Could not open the file!", file=sys.stderr)
    drop_parsed_sketches()
    from .engine import packed_offsets

    return packed, np.array(packed_offsets(packed), dtype=np.uint64)


def run(file_list: str, working_directory: str, output_filename: str, threads: int = 1, passes: int = 1,
        containment_threshold: float = 0.9, device: int = 0, verbose: bool = False) -> dict:
    if threads < 1:
        raise ValueError("number of threads must be at least 1")
    if passes < 1:
        raise ValueError("number of passes must be at least 1")
    if containment_threshold < 0.0 or containment_threshold > 1.0:
        raise ValueError("containment threshold must be between 0.0 and 1.0")
    with phases.phase("read_file_list"):
        paths = read_sketch_list(file_list)
    # Round 6: the sketches go from the parsed files into the PACKED form (5.7 bytes per hash), which is what is uploaded
    # (yh_db_create_packed expands it in HBM under the upload) and what the caller writes for `yacht run`; the plain CSR only
    # for inputs the packer refuses (YACHT_TRAIN_CSR=1 forces it: tests compare the two).
    packed = values = None
    with phases.phase("read_sig_files"):
        if os.environ.get("YACHT_TRAIN_CSR") != "1":
            packed, offsets = read_sketches_packed(paths, threads)
        if packed is None:
            values, offsets = read_sketches_csr(paths, threads)
    n = len(paths)
    sizes = np.diff(offsets).astype(np.uint32)
    empty = [int(i) for i in np.flatnonzero(sizes == 0)]
    with phases.phase("upload_and_index"):
        if packed is not None:
            db = RefDB.from_packed(packed, sizes=sizes, device=device, flags=YH_DB_PAIRWISE_ONLY)
        else:
            db = RefDB(values, offsets, device=device, flags=YH_DB_PAIRWISE_ONLY)
    with db:
        stats = db.index_stats()
        with phases.phase("pairwise"):
            pi, pj, pc = db.pairwise(float(containment_threshold))
        with phases.phase("release_device_db"):
            db.close()
    if verbose:
        print(f"Total number of sketches to read: {n}")
        print(f"Number of empty sketches: {len(empty)}")
        print(f"Total number of distinct hashes: {stats[0]}")
        print(f"Total number of distinct hashes that appear in only one sketch: {stats[1]}")
        print(f"Size of the index: {stats[2]}")

    # comparison files, split like the reference's (pass, thread) row blocks
    with phases.phase("write_comparison_files"):
        starts = np.searchsorted(pi, np.arange(n + 1), side="left")
        for (p, t, a, b) in row_ranges(n, threads, passes):
            with open(os.path.join(working_directory, f"{p}_{t:03d}.txt"), "w") as f:
                lo, hi = (int(starts[a]), int(starts[b])) if b > a else (0, 0)
                for k in range(lo, hi):
                    i, j = int(pi[k]), int(pj[k])
                    f.write(format_pair_line(i, j, int(pc[k]), int(sizes[i]), int(sizes[j])) + "\n")

    with phases.phase("select"):
        selected = train_select(sizes, pi, pj)
    with phases.phase("write_selected_list"):
        with open(output_filename, "w") as f:
            for g in selected:
                f.write(paths[int(g)] + "\n")
    return {"n": n, "empty": empty, "stats": stats, "n_pairs": int(pi.size), "selected": selected.tolist(),
            "paths": paths, "values": values, "offsets": offsets, "packed": packed}
