"""FracMinHash sketching of DNA on the HIP engine.

The reference's `yacht sketch ref|sample` are thin wrappers around the third-party
`sourmash sketch dna -p k=K,scaled=S,abund` (src/yacht/sketch_ref_genomes.py:24-61,
sketch_sample.py:31-49).  Here the k-mer hashing runs in libyacht_hip.so (yh_sketch_dna) and this
module does the file handling: FASTA/FASTQ (optionally gzip) in, sourmash-format signatures out
(yacht_amd.sigio), one signature per file (all records merged) like `sourmash sketch` without
`--singleton`.
"""
from __future__ import annotations

import ctypes as C
import gzip
import os
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, sigio

DEFAULT_SEED = 42


def max_hash_for_scaled(scaled: int) -> int:
    """sourmash: floor(float(2**64 - 1) / scaled); 18446744073709552 for scaled = 1000."""
    return int(float(2 ** 64 - 1) / float(scaled)) if scaled > 1 else 2 ** 64 - 1


def read_sequences(path: str) -> Iterator[Tuple[str, bytes]]:
    """(name, sequence) records of a FASTA or FASTQ file, gzip or plain."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        first = f.read(1)
        if not first:
            return
        rest = f.read()
    data = first + rest
    if first == b"@":  # FASTQ: 4-line records
        lines = data.split(b"\n")
        for i in range(0, len(lines) - 1, 4):
            if lines[i].startswith(b"@"):
                yield lines[i][1:].decode("utf-8", "replace"), lines[i + 1].strip()
        return
    name, parts = None, []
    for line in data.split(b"\n"):
        line = line.rstrip()
        if line.startswith(b">"):
            if name is not None:
                yield name, b"".join(parts)
            name, parts = line[1:].decode("utf-8", "replace"), []
        elif line:
            parts.append(line)
    if name is not None:
        yield name, b"".join(parts)


def hash_kmers(sequences: Iterable[bytes], ksize: int, scaled: int, seed: int = DEFAULT_SEED,
               device: int = 0) -> np.ndarray:
    """Kept hashes (unsorted, with duplicates) of all windows of all sequences: one device call on
    the sequences joined by a separator byte (a non-ACGT byte breaks every window that spans it)."""
    lib = _lib.load()
    seqs = list(sequences)
    if len(seqs) == 1:  # (no copy: a genome of one record, or a caller that joined the records itself)
        buf = seqs[0] if isinstance(seqs[0], np.ndarray) else np.frombuffer(seqs[0], dtype=np.uint8)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
    else:
        buf = np.frombuffer(b"\n".join(bytes(s) for s in seqs), dtype=np.uint8)
    if buf.size < ksize:
        return np.zeros(0, dtype=np.uint64)
    mh = max_hash_for_scaled(scaled)
    cap = max(int(buf.size / max(scaled, 1) * 1.5) + 4096, 4096)
    n = C.c_uint64(0)
    while True:
        out = np.empty(cap, dtype=np.uint64)
        rc = lib.yh_sketch_dna(C.c_void_p(buf.ctypes.data), buf.size, ksize, seed, mh, device, cap,
                               C.c_void_p(out.ctypes.data), C.byref(n))
        if rc == _lib.YH_ERR_CAPACITY:
            cap = int(n.value) + 16
            continue
        _lib.check(rc)
        return out[: int(n.value)]


def sketch_sequences(sequences: Iterable[bytes], ksize: int = 31, scaled: int = 1000, seed: int = DEFAULT_SEED,
                     device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(mins ascending, abundances) — what a sourmash signature stores."""
    kept = hash_kmers(sequences, ksize, scaled, seed, device)
    mins, counts = np.unique(kept, return_counts=True)
    return mins.astype(np.uint64), counts.astype(np.int64)


def sketch_file(path: str, ksize: int = 31, scaled: int = 1000, name: Optional[str] = None, seed: int = DEFAULT_SEED,
                device: int = 0) -> sigio.Signature:
    """One signature for the whole file; `name` defaults to the first record's header, as
    `sourmash sketch --name-from-first` does."""
    records = list(read_sequences(path))
    mins, ab = sketch_sequences((s for _n, s in records), ksize, scaled, seed, device)
    if name is None:
        name = records[0][0] if records else os.path.basename(path)
    mh = sigio.MinHash(mins, ksize, max_hash_for_scaled(scaled), ab, seed=seed)
    return sigio.Signature(mh, name=name, filename=path)


def sketch_files(paths: Sequence[str], out_zip: str, ksize: int = 31, scaled: int = 1000, device: int = 0) -> List[sigio.Signature]:
    """`yacht sketch ref`-style: one signature per genome file, written as a sourmash .sig.zip."""
    sigs = [sketch_file(p, ksize, scaled, device=device) for p in paths]
    sigio.write_sig_zip(sigs, out_zip)
    return sigs


# ---- `yacht sketch ref` / `yacht sketch sample` (reference sketch_ref_genomes.py, sketch_sample.py) ----
GENOME_SUFFIXES = (".fasta", ".fna", ".fas", ".fa", ".fasta.gz", ".fna.gz", ".fas.gz", ".fa.gz")


def add_ref_arguments(parser) -> None:
    parser.add_argument("--infile", help="Input file or folder path.", required=True)
    parser.add_argument("--kmer", type=int, help="K-mer size.", default=31)
    parser.add_argument("--scaled", type=int, help="Scaled factor.", default=1000)
    parser.add_argument("--outfile", help="Output file name.", required=True)


def main_ref(args) -> None:
    """A file: one signature per record (the reference passes --singleton).  A folder: one
    signature per genome file found under it, named by the file name without its suffix."""
    if os.path.isfile(args.infile):
        sigs = []
        for name, seq in read_sequences(args.infile):
            mins, ab = sketch_sequences([seq], args.kmer, args.scaled)
            sigs.append(sigio.Signature(sigio.MinHash(mins, args.kmer, max_hash_for_scaled(args.scaled), ab), name=name,
                                        filename=args.infile))
    elif os.path.isdir(args.infile):
        found = []
        for root, _dirs, files in os.walk(args.infile):
            for f in sorted(files):
                for suf in GENOME_SUFFIXES:
                    if f.endswith(suf):
                        found.append((f[: -len(suf)], os.path.join(root, f)))
                        break
        sigs = [sketch_file(p, args.kmer, args.scaled, name=n) for n, p in sorted(found)]
    else:
        raise FileNotFoundError(f"Input path {args.infile} does not exist.")
    sigio.write_sig_zip(sigs, args.outfile)


def add_sample_arguments(parser) -> None:
    parser.add_argument("--infile", nargs="+", required=True,
                        help="Input FASTA/Q file(s). For paired-end reads, provide two files.")
    parser.add_argument("--kmer", type=int, help="K-mer size.", default=31)
    parser.add_argument("--scaled", type=int, help="Scaled factor.", default=1000)
    parser.add_argument("--outfile", help="Output file name.", required=True)


def main_sample(args) -> None:
    """All reads of the one or two files merged into ONE signature with abundances."""
    if len(args.infile) not in (1, 2):
        raise ValueError("Please provide either one file for single-end reads or two files for paired-end reads.")
    seqs: List[bytes] = []
    for path in args.infile:
        seqs += [s for _n, s in read_sequences(path)]
    mins, ab = sketch_sequences(seqs, args.kmer, args.scaled)
    sig = sigio.Signature(sigio.MinHash(mins, args.kmer, max_hash_for_scaled(args.scaled), ab),
                          name=os.path.basename(args.infile[0]), filename=args.infile[0])
    sigio.write_sig_zip([sig], args.outfile)
