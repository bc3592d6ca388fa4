"""CPU restatement of sourmash's DNA FracMinHash sketching — TEST INFRASTRUCTURE ONLY.

What `sourmash sketch dna -p k=K,scaled=S,abund` computes (the reference shells out to it:
src/yacht/sketch_ref_genomes.py:25,61, sketch_sample.py:32,49; sourmash itself is third-party and
not installed here, so this follows its published definition, not its source):

  * every length-K window of every record that consists of A/C/G/T only (case-insensitive);
  * canonical form = the lexicographically smaller of the k-mer and its reverse complement;
  * hash = first 64 bits of MurmurHash3_x64_128 (Austin Appleby, public domain) of the canonical
    k-mer's ASCII bytes, seed 42;
  * kept iff hash <= max_hash, max_hash = floor(float(2**64 - 1) / scaled) = 18446744073709552
    for scaled = 1000 (the value in the reference's own .sig fixtures);
  * abundance = number of windows that produced the hash.

Pinned by known answers read from the reference's shipped result file
tests/testdata/standardize_output_testdata/results/result.xlsx (sketches of demo/ref_genomes made
with the reference's own pipeline): (distinct hashes, total k-mers) per genome, see
tests/test_sketch.py.  numpy-vectorised, so whole genomes take seconds.
"""
from __future__ import annotations

import gzip
from typing import Iterable, Iterator, Tuple

import numpy as np

C1 = np.uint64(0x87C37B91114253D5)
C2 = np.uint64(0x4CF5AD432745937F)


def max_hash_for_scaled(scaled: int) -> int:
    return int(float(2 ** 64 - 1) / float(scaled)) if scaled > 1 else 2 ** 64 - 1


def _rotl(x: np.ndarray, r: int) -> np.ndarray:
    return (x << np.uint64(r)) | (x >> np.uint64(64 - r))


def _fmix(k: np.ndarray) -> np.ndarray:
    k = k ^ (k >> np.uint64(33))
    k = k * np.uint64(0xFF51AFD7ED558CCD)
    k = k ^ (k >> np.uint64(33))
    k = k * np.uint64(0xC4CEB9FE1A85EC53)
    return k ^ (k >> np.uint64(33))


def _load_le(b: np.ndarray, start: int, n: int) -> np.ndarray:
    """little-endian uint64 from bytes b[:, start:start+n] (n <= 8)."""
    out = np.zeros(b.shape[0], dtype=np.uint64)
    for i in range(n):
        out |= b[:, start + i].astype(np.uint64) << np.uint64(8 * i)
    return out


def murmur3_x64_128_h1(b: np.ndarray, seed: int = 42) -> np.ndarray:
    """First 64 bits of MurmurHash3_x64_128 for every row of the uint8 matrix b (equal lengths)."""
    with np.errstate(over="ignore"):
        n, length = b.shape
        h1 = np.full(n, seed, dtype=np.uint64)
        h2 = np.full(n, seed, dtype=np.uint64)
        nblocks = length // 16
        for blk in range(nblocks):
            k1 = _load_le(b, 16 * blk, 8)
            k2 = _load_le(b, 16 * blk + 8, 8)
            k1 = _rotl(k1 * C1, 31) * C2
            h1 = h1 ^ k1
            h1 = (_rotl(h1, 27) + h2) * np.uint64(5) + np.uint64(0x52DCE729)
            k2 = _rotl(k2 * C2, 33) * C1
            h2 = h2 ^ k2
            h2 = (_rotl(h2, 31) + h1) * np.uint64(5) + np.uint64(0x38495AB5)
        tail = 16 * nblocks
        rem = length - tail
        if rem > 8:
            k2 = _load_le(b, tail + 8, rem - 8)
            h2 = h2 ^ (_rotl(k2 * C2, 33) * C1)
        if rem > 0:
            k1 = _load_le(b, tail, min(rem, 8))
            h1 = h1 ^ (_rotl(k1 * C1, 31) * C2)
        h1 = h1 ^ np.uint64(length)
        h2 = h2 ^ np.uint64(length)
        h1 = h1 + h2
        h2 = h2 + h1
        h1 = _fmix(h1)
        h2 = _fmix(h2)
        return h1 + h2


_CODE = np.full(256, 4, dtype=np.uint8)
for _c, _v in ((b"A", 0), (b"C", 1), (b"G", 2), (b"T", 3), (b"a", 0), (b"c", 1), (b"g", 2), (b"t", 3)):
    _CODE[_c[0]] = _v
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def kmer_hashes(seq: bytes, ksize: int, seed: int = 42) -> np.ndarray:
    """hashes of the canonical form of every valid window of one record, in window order."""
    s = np.frombuffer(seq, dtype=np.uint8)
    if s.size < ksize:
        return np.zeros(0, dtype=np.uint64)
    code = _CODE[s]
    win = np.lib.stride_tricks.sliding_window_view(code, ksize)
    valid = ~(win == 4).any(axis=1)
    fw = win[valid]
    if fw.shape[0] == 0:
        return np.zeros(0, dtype=np.uint64)
    rc = (3 - fw)[:, ::-1]
    # lexicographic comparison of the two code rows: first differing column decides
    diff = fw != rc
    first = np.where(diff.any(axis=1), diff.argmax(axis=1), 0)
    rows = np.arange(fw.shape[0])
    use_rc = rc[rows, first] < fw[rows, first]
    canon = np.where(use_rc[:, None], rc, fw)
    return murmur3_x64_128_h1(_BASES[canon], seed)


def read_fasta(path: str) -> Iterator[Tuple[str, bytes]]:
    opener = gzip.open if path.endswith(".gz") else open
    name, parts = None, []
    with opener(path, "rb") as f:
        for line in f:
            line = line.rstrip()
            if line.startswith(b">"):
                if name is not None:
                    yield name, b"".join(parts)
                name, parts = line[1:].decode("utf-8", "replace"), []
            elif line:
                parts.append(line)
    if name is not None:
        yield name, b"".join(parts)


def sketch_records(records: Iterable[bytes], ksize: int = 31, scaled: int = 1000, seed: int = 42):
    """(mins ascending uint64, abundances int64) of all records merged into one sketch."""
    mh = np.uint64(max_hash_for_scaled(scaled))
    kept = []
    for seq in records:
        h = kmer_hashes(seq, ksize, seed)
        kept.append(h[h <= mh])
    allh = np.concatenate(kept) if kept else np.zeros(0, np.uint64)
    mins, counts = np.unique(allh, return_counts=True)
    return mins.astype(np.uint64), counts.astype(np.int64)


def sketch_fasta(path: str, ksize: int = 31, scaled: int = 1000, seed: int = 42):
    return sketch_records((seq for _name, seq in read_fasta(path)), ksize, scaled, seed)
