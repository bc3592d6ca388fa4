"""Packed on-disk form of a reference set.

The reference re-opens every selected `.sig` (JSON) three times per `yacht run`
(hypothesis_recovery_src.py:93,154,168).  `yacht train` here also leaves

    {prefix}_intermediate_files/yacht_hip_db/values.npy    uint64[H]  all hashes, reference-major
                                            offsets.npy   uint64[N+1]
                                            meta.json     {"ksize", "md5sums": [...]}  (row order)

and `yacht run` memory-maps the two arrays and hands them straight to yh_db_create: no JSON
parsing, and pages are only touched by the host-to-device copy.  The cache is keyed by the md5
list (manifest order) and the k-mer size; anything else falls back to reading the signatures and
rewrites the cache.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

DIR_NAME = "yacht_hip_db"


def cache_dir(genome_dir: str) -> str:
    return os.path.join(genome_dir, DIR_NAME)


def save(genome_dir: str, md5sums: Sequence[str], ksize: int, values: np.ndarray, offsets: np.ndarray) -> bool:
    d = cache_dir(genome_dir)
    try:
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, "values.npy"), np.ascontiguousarray(values, dtype=np.uint64))
        np.save(os.path.join(d, "offsets.npy"), np.ascontiguousarray(offsets, dtype=np.uint64))
        with open(os.path.join(d, "meta.json"), "w") as f:
            json.dump({"ksize": int(ksize), "md5sums": list(md5sums)}, f)
        return True
    except OSError:
        return False  # read-only training directory: just do not cache


def load(genome_dir: str, md5sums: Sequence[str], ksize: int) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    d = cache_dir(genome_dir)
    try:
        with open(os.path.join(d, "meta.json")) as f:
            meta = json.load(f)
        if int(meta["ksize"]) != int(ksize) or list(meta["md5sums"]) != list(md5sums):
            return None
        values = np.load(os.path.join(d, "values.npy"), mmap_mode="r")
        offsets = np.load(os.path.join(d, "offsets.npy"), mmap_mode="r")
        if offsets.size != len(md5sums) + 1 or int(offsets[-1]) != values.size:
            return None
        return values, offsets
    except (OSError, ValueError, KeyError):
        return None


def subset(values: np.ndarray, offsets: np.ndarray, rows: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """CSR of the listed rows, in that order."""
    sizes = [int(offsets[r + 1] - offsets[r]) for r in rows]
    out_off = np.zeros(len(rows) + 1, dtype=np.uint64)
    if rows:
        out_off[1:] = np.cumsum(sizes, dtype=np.uint64)
    out = np.empty(int(out_off[-1]), dtype=np.uint64)
    for k, r in enumerate(rows):
        out[int(out_off[k]):int(out_off[k + 1])] = values[int(offsets[r]):int(offsets[r + 1])]
    return out, out_off
