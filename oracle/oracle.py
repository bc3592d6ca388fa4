"""Python face of the CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under yacht_amd/ does.  Two layers:

  * thin ctypes wrappers over oracle/_build/liboracle.so (yacht_oracle.cpp), fast enough for
    the 10^5..10^7-hash parity cases and for the timed CPU baseline;
  * *_py functions: independent pure-Python/numpy restatements with Python sets, written to
    read like the reference lines they cite; used on small cases to check the C++ layer.

Parity status: PINNED (see oracle/README.md for what each function was checked against).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
REF_EXE = os.path.join(HERE, "_ref", "run_yacht_train_core")
REFERENCE_ROOT = os.environ.get("YACHT_REFERENCE", "/root/reference")

_lib = None


def build(with_ref: bool = True) -> None:
    """Compile the C++ restatement, and the genuine reference core when its sources exist."""
    subprocess.run(["make", "-s", "-C", HERE, "all"], check=True)
    if with_ref and os.path.exists(os.path.join(REFERENCE_ROOT, "src", "cpp", "main.cpp")):
        subprocess.run(["make", "-s", "-C", HERE, "ref", f"REFERENCE={REFERENCE_ROOT}"], check=True)


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(with_ref=False)
        lib = C.CDLL(LIB_PATH)
        vp, u64 = C.c_void_p, C.c_uint64
        lib.oracle_overlap.restype = None
        lib.oracle_overlap.argtypes = [vp, vp, u64, vp, u64, vp, C.c_int]
        lib.oracle_exclusive.restype = None
        lib.oracle_exclusive.argtypes = [vp, vp, u64, vp, vp, u64, vp, vp]
        lib.oracle_train_pairs.restype = u64
        lib.oracle_train_pairs.argtypes = [vp, vp, u64, C.c_double, C.c_int, u64, vp, vp, vp, vp]
        lib.oracle_train_select.restype = u64
        lib.oracle_train_select.argtypes = [vp, u64, vp, vp, u64, vp]
        lib.oracle_hardware_threads.restype = C.c_int
        lib.oracle_hardware_threads.argtypes = []
        _lib = lib
    return _lib


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def hardware_threads() -> int:
    return int(_load().oracle_hardware_threads())


# ---- C++ layer ------------------------------------------------------------------------------------
def overlap(values, offsets, sample, threads: int = 1) -> np.ndarray:
    values, offsets, sample = _u64(values), _u64(offsets), _u64(sample)
    n = offsets.size - 1
    out = np.zeros(n, dtype=np.uint32)
    _load().oracle_overlap(_p(values), _p(offsets), n, _p(sample), sample.size, _p(out), threads)
    return out


def exclusive(values, offsets, mask, sample) -> Tuple[np.ndarray, np.ndarray]:
    values, offsets, sample = _u64(values), _u64(offsets), _u64(sample)
    n = offsets.size - 1
    mask = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    e = np.zeros(n, dtype=np.uint32)
    m = np.zeros(n, dtype=np.uint32)
    _load().oracle_exclusive(_p(values), _p(offsets), n, _p(mask), _p(sample), sample.size, _p(e), _p(m))
    return e, m


def train_pairs(values, offsets, c_thresh: float, threads: int = 1):
    """(pair_i, pair_j, count, (n_distinct, n_singletons, n_index)) — main.cpp:215-308."""
    values, offsets = _u64(values), _u64(offsets)
    n = offsets.size - 1
    stats = np.zeros(3, dtype=np.uint64)
    cap = max(4 * n, 1024)
    while True:
        pi = np.zeros(cap, dtype=np.uint32)
        pj = np.zeros(cap, dtype=np.uint32)
        pc = np.zeros(cap, dtype=np.uint32)
        k = int(_load().oracle_train_pairs(_p(values), _p(offsets), n, float(c_thresh), threads, cap, _p(pi), _p(pj),
                                           _p(pc), _p(stats)))
        if k <= cap:
            return pi[:k], pj[:k], pc[:k], tuple(int(x) for x in stats)
        cap = k


def train_select(sizes, pair_i, pair_j) -> np.ndarray:
    """Selected reference ids in walk order — main.cpp:371-407."""
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    pair_i = np.ascontiguousarray(pair_i, dtype=np.uint32)
    pair_j = np.ascontiguousarray(pair_j, dtype=np.uint32)
    sel = np.zeros(max(sizes.size, 1), dtype=np.uint32)
    k = int(_load().oracle_train_select(_p(sizes), sizes.size, _p(pair_i), _p(pair_j), pair_i.size, _p(sel)))
    return sel[:k]


# ---- pure-Python layer (small inputs only) ----------------------------------------------------------
def _slices(values, offsets) -> List[np.ndarray]:
    values, offsets = _u64(values), _u64(offsets)
    return [values[int(offsets[j]) : int(offsets[j + 1])] for j in range(offsets.size - 1)]


def overlap_py(values, offsets, sample) -> np.ndarray:
    """|set(R_j) & set(S)| — the quantity behind hypothesis_recovery_src.py:93-113."""
    s = set(int(x) for x in _u64(sample))
    return np.array([len(s.intersection(int(x) for x in r)) for r in _slices(values, offsets)], dtype=np.uint32)


def exclusive_py(values, offsets, mask, sample) -> Tuple[np.ndarray, np.ndarray]:
    """hypothesis_recovery_src.py:165-204 with Python sets, masked references in index order."""
    refs = _slices(values, offsets)
    mask = np.asarray(mask) != 0
    single, multiple = set(), set()
    for j, r in enumerate(refs):                      # :167-179
        if not mask[j]:
            continue
        for h in (int(x) for x in r):
            if h in multiple:
                continue
            if h in single:
                single.remove(h)
                multiple.add(h)
            else:
                single.add(h)
    s = set(int(x) for x in _u64(sample))             # :194
    e = np.zeros(len(refs), dtype=np.uint32)
    m = np.zeros(len(refs), dtype=np.uint32)
    for j, r in enumerate(refs):                      # :184-204
        if not mask[j]:
            continue
        excl = {int(x) for x in r if int(x) in single}
        e[j] = len(excl)
        m[j] = len(excl & s)
    return e, m


def train_pairs_py(values, offsets, c_thresh: float):
    """main.cpp:215-308 with dicts; (i, j, count) sorted by (i, j)."""
    refs = _slices(values, offsets)
    index = {}
    for i, r in enumerate(refs):
        for h in (int(x) for x in r):
            index.setdefault(h, []).append(i)
    n_distinct = len(index)
    index = {h: ids for h, ids in index.items() if len(ids) > 1}
    out = []
    for i, r in enumerate(refs):
        row = {}
        for h in (int(x) for x in r):
            for k in index.get(h, ()):
                row[k] = row.get(k, 0) + 1
        for j in sorted(row):
            if j == i or len(refs[i]) == 0 or len(refs[j]) == 0:
                continue
            if 1.0 * row[j] / len(refs[i]) < c_thresh:
                continue
            out.append((i, j, row[j]))
    pi = np.array([t[0] for t in out], dtype=np.uint32)
    pj = np.array([t[1] for t in out], dtype=np.uint32)
    pc = np.array([t[2] for t in out], dtype=np.uint32)
    return pi, pj, pc, (n_distinct, n_distinct - len(index), len(index))


# ---- hypothesis test (hypothesis_recovery_src.py:209-306) ---------------------------------------------
def single_hyp_test(exclusive_hashes_info_org, ksize, significance=0.99, ani_thresh=0.95, min_coverage=1):
    """Scalar restatement with the same scipy calls as the reference (scipy is the arithmetic
    there too, so equality is exact on the same scipy build)."""
    from scipy.special import betaincinv
    from scipy.stats import binom

    n_excl, n_match = exclusive_hashes_info_org
    p = ani_thresh ** ksize                                     # :254
    n_cov = int(n_excl * min_coverage)                          # :260
    thr = binom.ppf(1 - significance, n_cov, p)                 # :263-265
    conf = 1 - binom.cdf(thr, n_cov, p)                         # :267-269
    alt = 1 - (1 - betaincinv(n_cov - thr, 1 + thr, significance)) ** (1 / ksize)   # :229
    alt = -1.0 if np.isnan(alt) else alt                        # :230
    p_val = binom.cdf(n_match, n_cov, p) if n_match <= n_cov else 1.0               # :286-289
    present = (n_match >= thr) and (n_match != 0)               # :291-293
    return present, p_val, n_excl, n_cov, n_match, thr, conf, alt


# ---- the genuine reference executable (oracle/_ref) ---------------------------------------------------
def have_ref_exe() -> bool:
    return os.path.exists(REF_EXE) and os.access(REF_EXE, os.X_OK)


def write_minimal_sigs(sketches: Sequence[np.ndarray], directory: str) -> List[str]:
    """Write each sketch as the smallest JSON the reference core accepts (main.cpp:74-78)."""
    os.makedirs(directory, exist_ok=True)
    paths = []
    for k, s in enumerate(sketches):
        path = os.path.join(directory, f"sk{k:06d}.sig")
        with open(path, "w") as f:
            f.write('[{"signatures":[{"mins":[' + ",".join(str(int(x)) for x in s) + "]}]}]")
        paths.append(path)
    return paths


def run_ref_exe(sketches: Sequence[np.ndarray], c_thresh: float, workdir: str, threads: int = 1, passes: int = 1):
    """Run oracle/_ref/run_yacht_train_core; returns (selected ids in file order, sorted pair
    lines as text, stdout)."""
    paths = write_minimal_sigs(sketches, os.path.join(workdir, "sigs"))
    flist = os.path.join(workdir, "filelist.txt")
    with open(flist, "w") as f:
        f.write("\n".join(paths) + "\n")
    out = os.path.join(workdir, "selected.txt")
    proc = subprocess.run([REF_EXE, "-t", str(threads), "-c", repr(float(c_thresh)), "-p", str(passes), flist,
                           workdir, out], capture_output=True, text=True, check=True)
    index_of = {p: k for k, p in enumerate(paths)}
    with open(out) as f:
        selected = [index_of[line.strip()] for line in f if line.strip()]
    lines = []
    for name in sorted(os.listdir(workdir)):
        if name.endswith(".txt") and name[0].isdigit():
            with open(os.path.join(workdir, name)) as f:
                lines += [ln.strip() for ln in f if ln.strip()]
    lines.sort(key=lambda ln: tuple(int(x) for x in ln.split(",")[:2]))
    return selected, lines, proc.stdout
