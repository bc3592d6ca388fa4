"""Packed on-disk form of a reference set.

The reference re-opens every selected `.sig` (JSON) three times per `yacht run`
(hypothesis_recovery_src.py:93,154,168).  `yacht train` here also leaves

    {prefix}_intermediate_files/yacht_hip_db/values-<token>.npy   uint64[H]  all hashes, reference-major
                                            offsets-<token>.npy  uint64[N+1]
                                            meta.json   {"ksize", "md5sums": [...] (row order), "files", sizes, digest}

and `yacht run` memory-maps the two arrays and hands them straight to yh_db_create: no JSON
parsing, and pages are only touched by the host-to-device copy.  The cache is keyed by the md5
list (manifest order) and the k-mer size; anything else falls back to reading the signatures and
rewrites the cache.

Round 6: `yacht train` writes the PACKED form instead (`packed-<token>.npy`: yh_csr_pack's blob, ~5.7 instead of 8 bytes
per hash -- the selected sketches cut out of the blob the train core uploaded, yh_csr_subset, no hash decoded), `meta.json`
names it under "files": {"packed": ..}, and `yacht run` hands the memory-mapped blob to yh_db_create_packed
(load_any); directories written by earlier rounds (values / offsets) keep working.
"""
from __future__ import annotations

import hashlib
import json
import os
import secrets
from typing import List, Optional, Sequence, Tuple

import numpy as np

DIR_NAME = "yacht_hip_db"


def cache_dir(genome_dir: str) -> str:
    return os.path.join(genome_dir, DIR_NAME)


def _offsets_digest(offsets: np.ndarray) -> str:
    return hashlib.sha1(np.ascontiguousarray(offsets, dtype=np.uint64).tobytes()).hexdigest()


def save(genome_dir: str, md5sums: Sequence[str], ksize: int, values: np.ndarray, offsets: np.ndarray) -> bool:
    """Write the packed set so that a concurrent reader never sees a half-written one (many `yacht run`
    processes against one training directory is the normal workflow): the two arrays go into files of their
    own, named by a fresh token and complete before anything refers to them; meta.json -- written beside
    and moved into place with os.replace -- names the token, the sizes and a digest of the offsets.  A
    reader that opened the previous meta keeps reading the previous files (unlinked files stay readable)."""
    d = cache_dir(genome_dir)
    token = f"{os.getpid():x}-{secrets.token_hex(6)}"
    try:
        os.makedirs(d, exist_ok=True)
        values = np.ascontiguousarray(values, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        names = {"values": f"values-{token}.npy", "offsets": f"offsets-{token}.npy"}
        for key, arr in (("values", values), ("offsets", offsets)):
            tmp = os.path.join(d, names[key] + ".part")
            with open(tmp, "wb") as f:
                np.save(f, arr)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, os.path.join(d, names[key]))
        meta = {"ksize": int(ksize), "md5sums": list(md5sums), "files": names, "n_refs": int(offsets.size - 1),
                "n_hashes": int(values.size), "offsets_sha1": _offsets_digest(offsets)}
        tmp = os.path.join(d, f"meta-{token}.json.part")
        with open(tmp, "w") as f:
            json.dump(meta, f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, os.path.join(d, "meta.json"))
        _remove_older_generations(d, names)
        return True
    except OSError:
        return False  # read-only training directory: just do not cache


class PendingSubsetSave:
    """The packed copy of SOME rows of a CSR on its way to disk in a thread of its own (save_subset_async): the rows'
    slices are written straight from the source arrays -- no packed copy in memory first -- while the caller does other
    work (file writes release the GIL).  Nothing refers to the files until publish(); discard() removes them."""

    def __init__(self, genome_dir: str, ksize: int, values: Optional[np.ndarray], offsets: np.ndarray, rows: Sequence[int],
                 packed: Optional[np.ndarray] = None):
        import threading

        self.dir = cache_dir(genome_dir)
        self.ksize = int(ksize)
        token = f"{os.getpid():x}-{secrets.token_hex(6)}"
        self.token = token
        self.names = ({"packed": f"packed-{token}.npy"} if packed is not None else
                      {"values": f"values-{token}.npy", "offsets": f"offsets-{token}.npy"})
        rows = np.asarray(rows, dtype=np.int64)
        sizes = (offsets[rows + 1] - offsets[rows]).astype(np.uint64) if rows.size else np.zeros(0, np.uint64)
        self.out_offsets = np.zeros(rows.size + 1, dtype=np.uint64)
        if rows.size:
            self.out_offsets[1:] = np.cumsum(sizes, dtype=np.uint64)
        self.ok = False
        if packed is not None:
            self._thread = threading.Thread(target=self._write_packed, args=(packed, rows), name="yacht-hip-packed-db", daemon=True)
        else:
            self._thread = threading.Thread(target=self._write, args=(values, offsets, rows), name="yacht-hip-packed-db", daemon=True)
        self._thread.start()

    def _write_packed(self, packed: np.ndarray, rows: np.ndarray) -> None:
        """The rows cut out of the packed CSR the train core uploaded (yh_csr_subset: block entries re-based, payload words copied;
        the C call releases the GIL) and written as ONE file."""
        try:
            from .engine import csr_subset

            os.makedirs(self.dir, exist_ok=True)
            sub = csr_subset(packed, rows)
            tmp = os.path.join(self.dir, self.names["packed"] + ".part")
            with open(tmp, "wb") as f:
                np.save(f, sub)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, os.path.join(self.dir, self.names["packed"]))
            self.ok = True
        except Exception:  # noqa: BLE001  (read-only directory, a blob the library refuses: just do not cache)
            self.ok = False

    def _write(self, values: np.ndarray, offsets: np.ndarray, rows: np.ndarray) -> None:
        try:
            os.makedirs(self.dir, exist_ok=True)
            values = np.ascontiguousarray(values, dtype=np.uint64)
            n_out = int(self.out_offsets[-1])
            tmp = os.path.join(self.dir, self.names["values"] + ".part")
            with open(tmp, "wb") as f:
                np.lib.format.write_array_header_1_0(f, {"descr": "<u8", "fortran_order": False, "shape": (n_out,)})
                # runs of consecutive rows are one slice of the source
                k = 0
                while k < rows.size:
                    e = k + 1
                    while e < rows.size and rows[e] == rows[e - 1] + 1:
                        e += 1
                    f.write(memoryview(values[int(offsets[rows[k]]):int(offsets[rows[e - 1] + 1])]))
                    k = e
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, os.path.join(self.dir, self.names["values"]))
            tmp = os.path.join(self.dir, self.names["offsets"] + ".part")
            with open(tmp, "wb") as f:
                np.save(f, self.out_offsets)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, os.path.join(self.dir, self.names["offsets"]))
            self.ok = True
        except OSError:
            self.ok = False  # read-only training directory: just do not cache

    def publish(self, md5sums: Sequence[str]) -> bool:
        """Wait for the files and make them the directory's packed copy (meta.json moved into place)."""
        self._thread.join()
        if not self.ok:
            return False
        try:
            meta = {"ksize": self.ksize, "md5sums": list(md5sums), "files": self.names, "n_refs": int(self.out_offsets.size - 1),
                    "n_hashes": int(self.out_offsets[-1]), "offsets_sha1": _offsets_digest(self.out_offsets)}
            tmp = os.path.join(self.dir, f"meta-{self.token}.json.part")
            with open(tmp, "w") as f:
                json.dump(meta, f)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, os.path.join(self.dir, "meta.json"))
            _remove_older_generations(self.dir, self.names)
            return True
        except OSError:
            return False

    def discard(self) -> None:
        self._thread.join()
        for name in self.names.values():
            for path in (os.path.join(self.dir, name), os.path.join(self.dir, name + ".part")):
                try:
                    os.remove(path)
                except OSError:
                    pass


def save_subset_async(genome_dir: str, ksize: int, values: Optional[np.ndarray], offsets: np.ndarray, rows: Sequence[int],
                      packed: Optional[np.ndarray] = None) -> PendingSubsetSave:
    """Start writing the packed copy of `rows` (in that order) of the CSR -- or, given the train core's packed blob, of that; see
    PendingSubsetSave."""
    return PendingSubsetSave(genome_dir, ksize, values, offsets, rows, packed=packed)


def _remove_older_generations(d: str, names) -> None:
    for old in os.listdir(d):  # earlier generations (best effort; a reader that has them open keeps them)
        if (old.startswith("values-") or old.startswith("offsets-") or old.startswith("packed-") or old in ("values.npy", "offsets.npy")) \
                and old not in names.values() and not old.endswith(".part"):
            try:
                os.remove(os.path.join(d, old))
            except OSError:
                pass


def load_any(genome_dir: str, md5sums: Sequence[str], ksize: int):
    """{"packed": blob} (round 6: what `yacht train` writes; memory-mapped, for RefDB.from_packed) or {"values", "offsets"}
    (earlier rounds' directories, and what `yacht run` writes itself after parsing the signature files) -- or None when there is
    no copy for exactly these references, or when what is on disk does not agree with its own meta."""
    d = cache_dir(genome_dir)
    try:
        with open(os.path.join(d, "meta.json")) as f:
            meta = json.load(f)
        if int(meta["ksize"]) != int(ksize) or list(meta["md5sums"]) != list(md5sums):
            return None
        files = meta.get("files") or {}
        if "packed" in files:
            blob = np.load(os.path.join(d, files["packed"]), mmap_mode="r")
            if blob.dtype != np.uint64 or blob.ndim != 1 or blob.size < 8 + len(md5sums) + 1:
                return None
            n_refs, n_hashes = int(blob[1]), int(blob[2])
            offsets = np.asarray(blob[8: 8 + n_refs + 1])
            if n_refs != len(md5sums) or int(meta.get("n_refs", -1)) != n_refs or int(meta.get("n_hashes", -1)) != n_hashes \
                    or meta.get("offsets_sha1") != _offsets_digest(offsets):
                return None
            return {"packed": blob, "offsets": offsets}
    except (OSError, ValueError, KeyError):
        return None
    got = load(genome_dir, md5sums, ksize)
    return None if got is None else {"values": got[0], "offsets": got[1]}


def load(genome_dir: str, md5sums: Sequence[str], ksize: int) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """(values, offsets) memory-mapped, or None when there is no packed set for exactly these references --
    or when what is on disk does not agree with its own meta (sizes, digest of the offsets).  (A directory that holds the
    packed form is unpacked here: callers that want the CSR; `yacht run` itself takes load_any.)"""
    d = cache_dir(genome_dir)
    try:
        with open(os.path.join(d, "meta.json")) as f:
            meta = json.load(f)
        if int(meta["ksize"]) != int(ksize) or list(meta["md5sums"]) != list(md5sums):
            return None
        if "packed" in (meta.get("files") or {}):
            got = load_any(genome_dir, md5sums, ksize)
            if got is None:
                return None
            from .engine import csr_unpack

            return csr_unpack(np.ascontiguousarray(got["packed"]))
        files = meta.get("files") or {"values": "values.npy", "offsets": "offsets.npy"}
        values = np.load(os.path.join(d, files["values"]), mmap_mode="r")
        offsets = np.load(os.path.join(d, files["offsets"]), mmap_mode="r")
        if offsets.size != len(md5sums) + 1 or int(offsets[-1]) != values.size or int(offsets[0]) != 0:
            return None
        if "n_hashes" in meta and (int(meta["n_hashes"]) != values.size or int(meta["n_refs"]) != offsets.size - 1
                                   or meta.get("offsets_sha1") != _offsets_digest(np.asarray(offsets))):
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
