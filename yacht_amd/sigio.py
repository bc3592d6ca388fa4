"""Reader/writer for sourmash signature files without sourmash.

YACHT loads sketches through `sourmash.load_file_as_signatures` (reference utils.py:31-51) and
only touches a handful of attributes: `sig.name`, `sig.md5sum()`, `sig.minhash.hashes` (a
hash -> abundance mapping), `sig.minhash.scaled`, `sig.minhash.mean_abundance`, plus the ksize
filter.  This module provides exactly those on top of the on-disk formats the reference's own
fixtures use ("I/O"):

  * `.sig`      JSON: a list of records {name, filename, signatures: [{ksize, seed, max_hash,
                mins[], abundances[], md5sum, molecule, num}], ...}
  * `.sig.gz`   the same, gzip-compressed
  * `.zip`      `SOURMASH-MANIFEST.csv` + `signatures/<md5>.sig.gz`

Hash arrays are kept as ascending numpy uint64 (the order sourmash writes "mins" in), which is
what the HIP engine's CSR layout wants.
"""
from __future__ import annotations

import csv
import gzip
import hashlib
import io
import json
import os
import zipfile
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

MANIFEST_NAME = "SOURMASH-MANIFEST.csv"
MANIFEST_HEADER = "# SOURMASH-MANIFEST-VERSION: 1.0"
MANIFEST_COLUMNS = ["internal_location", "md5", "md5short", "ksize", "moltype", "num", "scaled", "n_hashes",
                    "with_abundance", "name", "filename"]
_TWO64 = 2 ** 64


def scaled_for_max_hash(max_hash: int) -> int:
    """sourmash's inverse of max_hash = round(2**64 / scaled)."""
    return 0 if max_hash == 0 else int(round(_TWO64 / max_hash))


def max_hash_for_scaled(scaled: int) -> int:
    return 0 if scaled == 0 else (_TWO64 - 1 if scaled == 1 else int(round(_TWO64 / scaled)))


def md5_of_mins(ksize: int, mins: Sequence[int]) -> str:
    """sourmash's MinHash md5: md5 over str(ksize) then every hash in decimal, ascending."""
    m = hashlib.md5()
    m.update(str(int(ksize)).encode("ascii"))
    for h in mins:
        m.update(str(int(h)).encode("ascii"))
    return m.hexdigest()


class MinHash:
    """The slice of sourmash.MinHash that YACHT uses."""

    def __init__(self, mins: np.ndarray, ksize: int, max_hash: int, abundances: Optional[np.ndarray] = None,
                 seed: int = 42, molecule: str = "dna", num: int = 0):
        mins = np.asarray(mins, dtype=np.uint64)
        if mins.size > 1 and not bool(np.all(mins[1:] > mins[:-1])):
            order = np.argsort(mins, kind="stable")
            mins = mins[order]
            if abundances is not None:
                abundances = np.asarray(abundances)[order]
            keep = np.concatenate([[True], mins[1:] != mins[:-1]])
            mins = mins[keep]
            if abundances is not None:
                abundances = abundances[keep]
        self.mins = np.ascontiguousarray(mins)
        self.abundances = None if abundances is None else np.asarray(abundances, dtype=np.int64)
        self.ksize = int(ksize)
        self.max_hash = int(max_hash)
        self.seed = int(seed)
        self.moltype = "DNA" if molecule.lower() == "dna" else molecule
        self.num = int(num)

    @property
    def scaled(self) -> int:
        return scaled_for_max_hash(self.max_hash)

    @property
    def track_abundance(self) -> bool:
        return self.abundances is not None

    @property
    def hashes(self) -> Dict[int, int]:
        """hash -> abundance (1 when abundances are not tracked), ascending key order."""
        if self.abundances is None:
            return {int(h): 1 for h in self.mins}
        return {int(h): int(a) for h, a in zip(self.mins, self.abundances)}

    @property
    def mean_abundance(self) -> Optional[float]:
        if self.abundances is None:
            return None
        if self.abundances.size == 0:
            return float("nan")
        return float(np.mean(self.abundances))

    def __len__(self) -> int:
        return int(self.mins.size)

    def md5sum(self) -> str:
        return md5_of_mins(self.ksize, self.mins)


class Signature:
    def __init__(self, minhash: MinHash, name: str = "", filename: str = ""):
        self.minhash = minhash
        self.name = name
        self.filename = filename

    def md5sum(self) -> str:
        return self.minhash.md5sum()

    def __repr__(self) -> str:
        return f"Signature({self.name!r}, k={self.minhash.ksize}, n={len(self.minhash)})"


# ---- reading -----------------------------------------------------------------------------------------
def _signatures_from_json(text: str, ksize: Optional[int]) -> List[Signature]:
    out: List[Signature] = []
    for rec in json.loads(text):
        for s in rec.get("signatures", []):
            if ksize is not None and int(s["ksize"]) != int(ksize):
                continue
            mins = np.array(s.get("mins", []), dtype=np.uint64)
            ab = s.get("abundances")
            mh = MinHash(mins, s["ksize"], s.get("max_hash", 0), None if ab is None else np.array(ab, dtype=np.int64),
                         seed=s.get("seed", 42), molecule=s.get("molecule", "dna"), num=s.get("num", 0))
            out.append(Signature(mh, rec.get("name", ""), rec.get("filename", "")))
    return out


def load_file_as_signatures(filename: str, ksize: Optional[int] = None) -> List[Signature]:
    """Counterpart of sourmash.load_file_as_signatures for .sig, .sig.gz and .zip files."""
    if zipfile.is_zipfile(filename):
        sigs: List[Signature] = []
        with zipfile.ZipFile(filename) as z:
            for info in z.infolist():
                n = info.filename
                if n.endswith("/") or not (n.endswith(".sig") or n.endswith(".sig.gz")):
                    continue
                raw = z.read(info)
                if n.endswith(".gz") or raw[:2] == b"\x1f\x8b":
                    raw = gzip.decompress(raw)
                sigs += _signatures_from_json(raw.decode("utf-8"), ksize)
        return sigs
    with open(filename, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    return _signatures_from_json(raw.decode("utf-8"), ksize)


def read_mins_first_signature(filename: str) -> np.ndarray:
    """What the reference's train core reads from a file: record 0, signature 0, "mins"
    (src/cpp/main.cpp:74-78 — ksize is NOT checked there).  Unreadable -> empty."""
    try:
        with open(filename, "rb") as f:
            raw = f.read()
        if raw[:2] == b"\x1f\x8b":
            raw = gzip.decompress(raw)
        return np.array(json.loads(raw.decode("utf-8"))[0]["signatures"][0]["mins"], dtype=np.uint64)
    except Exception:
        return np.zeros(0, dtype=np.uint64)


def zip_has_manifest(filename: str) -> bool:
    with zipfile.ZipFile(filename) as z:
        return MANIFEST_NAME in z.namelist()


# ---- writing -----------------------------------------------------------------------------------------
def signature_record(sig: Signature) -> dict:
    mh = sig.minhash
    s = {
        "num": mh.num,
        "ksize": mh.ksize,
        "seed": mh.seed,
        "max_hash": mh.max_hash,
        "mins": [int(x) for x in mh.mins],
        "md5sum": mh.md5sum(),
    }
    if mh.abundances is not None:
        s["abundances"] = [int(x) for x in mh.abundances]
    s["molecule"] = "dna" if mh.moltype == "DNA" else mh.moltype
    return {
        "class": "sourmash_signature",
        "email": "",
        "hash_function": "0.murmur64",
        "filename": sig.filename,
        "name": sig.name,
        "license": "CC0",
        "signatures": [s],
        "version": 0.4,
    }


def dumps_signature(sig: Signature) -> str:
    return json.dumps([signature_record(sig)], separators=(",", ":"))


def write_sig(sig: Signature, path: str) -> None:
    data = dumps_signature(sig).encode("utf-8")
    if path.endswith(".gz"):
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def write_sig_zip(signatures: Iterable[Signature], path: str) -> None:
    """A sourmash-style .sig.zip: manifest + signatures/<md5>.sig.gz."""
    rows = []
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        for sig in signatures:
            md5 = sig.md5sum()
            loc = f"signatures/{md5}.sig.gz"
            buf = io.BytesIO()
            with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0) as g:
                g.write(dumps_signature(sig).encode("utf-8"))
            z.writestr(loc, buf.getvalue())
            mh = sig.minhash
            rows.append([loc, md5, md5[:8], mh.ksize, mh.moltype, mh.num, mh.scaled, len(mh),
                         int(mh.track_abundance), sig.name, sig.filename])
        out = io.StringIO()
        out.write(MANIFEST_HEADER + "\n")
        w = csv.writer(out, lineterminator="\n")
        w.writerow(MANIFEST_COLUMNS)
        w.writerows(rows)
        z.writestr(MANIFEST_NAME, out.getvalue())


def make_signature(mins, ksize: int = 31, scaled: int = 1000, name: str = "", filename: str = "",
                   abundances=None) -> Signature:
    return Signature(MinHash(np.asarray(mins, dtype=np.uint64), ksize, max_hash_for_scaled(scaled),
                             None if abundances is None else np.asarray(abundances, dtype=np.int64)), name, filename)
