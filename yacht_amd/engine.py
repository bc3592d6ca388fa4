"""RefDB — Python handle over the HIP containment engine (libyacht_hip.so).

A RefDB is a reference-sketch database resident in one GPU's HBM.  It answers the three
questions of YACHT's hot path:

    overlap(sample)              |S ∩ R_j| for every reference       (R1, `yacht run`)
    exclusive(mask, sample)      subset-exclusive hash counts         (R2, `yacht run`)
    pairwise(c_thresh)           above-threshold reference pairs      (T2-T4, `yacht train`)

numpy arrays in, numpy arrays out; all arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import (YH_DB_DEFAULT, YH_DB_KEEP_CSR, YH_DB_NO_DIRECTORY, YH_DB_NO_INDEX,  # noqa: F401
                   YH_DB_PAIRWISE_ONLY,
                   YachtHipError)


def pack_csr(sketches: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """Pack a list of ascending uint64 hash arrays into (values, offsets)."""
    offsets = np.zeros(len(sketches) + 1, dtype=np.uint64)
    if len(sketches):
        offsets[1:] = np.cumsum([len(s) for s in sketches], dtype=np.uint64)
    if len(sketches) and int(offsets[-1]):
        values = np.concatenate([np.asarray(s, dtype=np.uint64) for s in sketches])
    else:
        values = np.zeros(0, dtype=np.uint64)
    return np.ascontiguousarray(values, dtype=np.uint64), offsets


# one row of the compact result of a run step: a reference with overlap > 0 and its three counts (yh_run_row)
ROW_DTYPE = np.dtype([("ref", np.uint32), ("overlap", np.uint32), ("n_excl", np.uint32), ("n_match", np.uint32)])
# one row of the compact result of a BATCH (yh_batch_row): which sample, which reference, the three counts
BATCH_ROW_DTYPE = np.dtype([("sample", np.uint32), ("ref", np.uint32), ("overlap", np.uint32), ("n_excl", np.uint32),
                            ("n_match", np.uint32)])


def pack_sample(sample, out: Optional[np.ndarray] = None, threads: int = 0) -> np.ndarray:
    """A strictly ascending uint64 sketch as the packed bytes yh_run_submit_packed uploads (~4.7 bytes per hash;
    include/yacht_hip.h).  Host only: no device, any thread.  `out`: a uint8 buffer to pack into (e.g. a
    PinnedArray's array, so that the upload overlaps the kernels); the returned array is the used part of it.
    `threads`: host threads of this one call (0: up to 8 for a large sample; 1: the calling thread only -- for callers
    that pack many samples at once from threads of their own: the C call releases the GIL)."""
    lib = _lib.load()
    sample = _as_u64(sample)
    need = int(lib.yh_sample_pack_bound(sample.size))
    if out is None:
        out = np.empty(need, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.flags.c_contiguous
    n = C.c_uint64(0)
    _lib.check(lib.yh_sample_pack_threads(_ptr(sample), sample.size, _ptr(out), out.size, C.byref(n), int(threads)))
    return out[: int(n.value)]


def unpack_sample(packed: np.ndarray) -> np.ndarray:
    """The sketch a packed sample holds (host-side inverse of pack_sample)."""
    lib = _lib.load()
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n = C.c_uint64(0)
    _lib.check(lib.yh_sample_unpack(_ptr(packed), packed.size, None, 0, C.byref(n)))
    out = np.zeros(int(n.value), dtype=np.uint64)
    _lib.check(lib.yh_sample_unpack(_ptr(packed), packed.size, _ptr(out), out.size, C.byref(n)))
    return out


def _as_u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _ptr(a: Optional[np.ndarray]) -> C.c_void_p:
    return C.c_void_p(0 if a is None else a.ctypes.data)


class RefDB:
    """Reference sketches in HBM: delta stream, bucket table + presence filter, shared-hash inverted index."""

    def __init__(self, values, offsets, device: int = 0, flags: int = YH_DB_DEFAULT):
        self._h = C.c_void_p(0)
        lib = _lib.load()
        values = _as_u64(values)
        offsets = _as_u64(offsets)
        if offsets.ndim != 1 or offsets.size < 1:
            raise ValueError("offsets must be a 1-D array of length n_refs + 1")
        if int(offsets[-1]) != values.size:
            raise ValueError("offsets[-1] must equal len(values)")
        h = C.c_void_p(0)
        _lib.check(lib.yh_db_create(_ptr(values), _ptr(offsets), offsets.size - 1, device, flags, C.byref(h)))
        self._h = h
        self._lib = lib
        self.n_refs = offsets.size - 1
        self.sizes = np.diff(offsets).astype(np.uint32)

    @classmethod
    def from_sketches(cls, sketches: Sequence[np.ndarray], **kw) -> "RefDB":
        values, offsets = pack_csr(sketches)
        return cls(values, offsets, **kw)

    @classmethod
    def from_device(cls, d_values: int, d_offsets: int, n_refs: int, sizes: Optional[np.ndarray] = None,
                    device: int = 0, flags: int = YH_DB_DEFAULT) -> "RefDB":
        """Build from CSR arrays that already live in this device's HBM (raw device addresses)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(0)
        lib = _lib.load()
        h = C.c_void_p(0)
        _lib.check(lib.yh_db_create_device(C.c_void_p(d_values), C.c_void_p(d_offsets), n_refs, device, flags, C.byref(h)))
        self._h = h
        self._lib = lib
        self.n_refs = n_refs
        self.sizes = sizes
        return self

    @classmethod
    def from_packed(cls, packed: np.ndarray, sizes: Optional[np.ndarray] = None, device: int = 0, flags: int = YH_DB_DEFAULT) -> "RefDB":
        """Build from a packed CSR (csr_pack's uint64 array): 0.7 of the bytes cross the bus, the sketches are expanded in HBM."""
        assert packed.dtype == np.uint64 and packed.flags["C_CONTIGUOUS"], "csr_pack returns what this takes"
        self = cls.__new__(cls)
        self._h = C.c_void_p(0)
        lib = _lib.load()
        h = C.c_void_p(0)
        _lib.check(lib.yh_db_create_packed(_ptr(packed), packed.nbytes, device, flags, C.byref(h)))
        self._h = h
        self._lib = lib
        self.n_refs = int(self.info()["n_refs"])
        self.sizes = sizes
        return self

    # ---- lifetime ---------------------------------------------------------------------------
    def close(self, release_pool: bool = False) -> None:
        """Destroy the handle.  Its arrays go to the library's buffer cache (the next handle of this process takes them without
        the driver); release_pool=True hands the cache's idle blocks back to the driver as well (_lib.pool_release) -- for a
        process that needs the memory elsewhere next (torch, RCCL)."""
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.yh_db_destroy(h)
            self._h = C.c_void_p(0)
        if release_pool:
            _lib.pool_release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- metadata ---------------------------------------------------------------------------
    def info(self) -> dict:
        inf = _lib.DbInfo()
        _lib.check(self._lib.yh_db_get_info(self._h, C.byref(inf)))
        return {name: getattr(inf, name) for name, _ in inf._fields_}

    def timing(self) -> dict:
        t = _lib.Timing()
        _lib.check(self._lib.yh_db_get_timing(self._h, C.byref(t)))
        return {name: getattr(t, name) for name, _ in t._fields_}

    def index_stats(self) -> Tuple[int, int, int]:
        """(distinct hashes, hashes in exactly one reference, hashes kept in the index)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.yh_index_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def pairwise_row_stats(self) -> Tuple[int, int]:
        """(rows the last pairwise() summed sparsely, rows it handed back to the dense pass) -- (0, 0): every row dense."""
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.yh_pairwise_row_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_stream(self, hip_stream: int) -> None:
        _lib.check(self._lib.yh_db_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_batch_finish_stream(self, hip_stream: Optional[int]) -> None:
        """The second halves of the batched hash-range calls (run_batch_finish_range_device, words unpack, rows pack) on this
        hipStream_t beside the next block's first half; None = on the handle's stream again (include/yacht_hip.h)."""
        _lib.check(self._lib.yh_db_set_batch_finish_stream(self._h, C.c_void_p(hip_stream or 0)))

    def set_lookup(self, mode: int) -> None:
        """_lib.YH_LOOKUP_AUTO (default: by cost), YH_LOOKUP_STREAM or YH_LOOKUP_INDEXED for overlap / run queries."""
        _lib.check(self._lib.yh_db_set_lookup(self._h, mode))

    def lookup_choice(self, n_sample: int) -> int:
        rc = self._lib.yh_db_lookup_choice(self._h, n_sample)
        if rc < 0:
            _lib.check(rc)
        return rc

    def synchronize(self) -> None:
        _lib.check(self._lib.yh_db_synchronize(self._h))

    # ---- yacht run ---------------------------------------------------------------------------
    def overlap(self, sample, method: str = "tile") -> np.ndarray:
        sample = _as_u64(sample)
        out = np.zeros(self.n_refs, dtype=np.uint32)
        fn = self._lib.yh_overlap if method == "tile" else self._lib.yh_overlap_bsearch
        _lib.check(fn(self._h, _ptr(sample), sample.size, _ptr(out)))
        return out

    def exclusive(self, mask, sample) -> Tuple[np.ndarray, np.ndarray]:
        sample = _as_u64(sample)
        mask = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if mask.size != self.n_refs:
            raise ValueError("mask must have one entry per reference")
        e = np.zeros(self.n_refs, dtype=np.uint32)
        m = np.zeros(self.n_refs, dtype=np.uint32)
        _lib.check(self._lib.yh_exclusive(self._h, _ptr(mask), _ptr(sample), sample.size, _ptr(e), _ptr(m)))
        return e, m

    def run_counts(self, sample) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """overlap, and exclusive counts relative to the references with overlap > 0."""
        sample = _as_u64(sample)
        ov = np.zeros(self.n_refs, dtype=np.uint32)
        e = np.zeros(self.n_refs, dtype=np.uint32)
        m = np.zeros(self.n_refs, dtype=np.uint32)
        _lib.check(self._lib.yh_run(self._h, _ptr(sample), sample.size, _ptr(ov), _ptr(e), _ptr(m)))
        return ov, e, m

    # sharded run (dist.ShardedRefDB): the step in two halves around the exchange of the subset bits
    def set_ghosts(self, ghost_begin: int, n_ghost: int, d_ghost_src: int) -> None:
        _lib.check(self._lib.yh_db_set_ghosts(self._h, ghost_begin, n_ghost, C.c_void_p(d_ghost_src)))

    def run_local_device(self, d_sample: int, n_sample: int, d_overlap: int, d_excl: int, d_match: int,
                         d_bits_out: int, ctx: int = 0) -> None:
        _lib.check(self._lib.yh_run_local_device(self._h, ctx, C.c_void_p(d_sample), n_sample, C.c_void_p(d_overlap),
                                                 C.c_void_p(d_excl), C.c_void_p(d_match), C.c_void_p(d_bits_out)))

    def run_finish_device(self, d_global_bits: int, d_excl: int, ctx: int = 0) -> None:
        _lib.check(self._lib.yh_run_finish_device(self._h, ctx, C.c_void_p(d_global_bits), C.c_void_p(d_excl)))

    # hash-range shards (dist.HashRangeRefDB): the step in two halves around the all-gather of the ranks' subset bits
    def run_local_range_device(self, d_sample: int, n_sample: int, d_overlap: int, d_match: int, d_bits_out: int,
                               ctx: int = 0) -> None:
        _lib.check(self._lib.yh_run_local_range_device(self._h, ctx, C.c_void_p(d_sample), n_sample, C.c_void_p(d_overlap),
                                                       C.c_void_p(d_match), C.c_void_p(d_bits_out)))

    def run_finish_range_device(self, d_gathered_bits: int, n_ranks: int, stride_words: int, d_excl: int, ctx: int = 0) -> None:
        _lib.check(self._lib.yh_run_finish_range_device(self._h, ctx, C.c_void_p(d_gathered_bits), n_ranks, stride_words,
                                                        C.c_void_p(d_excl)))

    def run_batch_local_range_device(self, d_samples: int, d_offsets: int, n_samples: int, total_hashes: int, d_overlap: int,
                                     d_maskwords_out: int, slot: int = 0) -> None:
        _lib.check(self._lib.yh_run_batch_local_range_device(self._h, slot, C.c_void_p(d_samples), C.c_void_p(d_offsets), n_samples,
                                                             total_hashes, C.c_void_p(d_overlap), C.c_void_p(d_maskwords_out)))

    def run_batch_finish_range_device(self, n_samples: int, d_gathered: int, n_ranks: int, d_overlap: int, d_excl: int,
                                      d_match: int, slot: int = 0) -> None:
        _lib.check(self._lib.yh_run_batch_finish_range_device(self._h, slot, n_samples, C.c_void_p(d_gathered), n_ranks,
                                                              C.c_void_p(d_overlap), C.c_void_p(d_excl), C.c_void_p(d_match)))

    # the compact form of a batch's result: one entry per (reference, sample) of the batch's subset, in (reference, sample) order
    def run_batch_rows_pack_device(self, d_overlap: int, d_excl: int, d_match: int, d_vals: int, cap_rows: int, d_n_rows: int,
                                   slot: int = 0) -> None:
        """d_vals [cap_rows][3] uint32 = this handle's (overlap, n_excl, n_match) of every entry; d_n_rows: their number."""
        _lib.check(self._lib.yh_run_batch_rows_pack_device(self._h, slot, C.c_void_p(d_overlap), C.c_void_p(d_excl), C.c_void_p(d_match),
                                                           C.c_void_p(d_vals), cap_rows, C.c_void_p(d_n_rows)))

    def run_batch_rows_unpack_device(self, d_vals: int, cap_rows: int, d_rows: int, d_n_rows: int, slot: int = 0) -> None:
        """d_rows [cap_rows] BATCH_ROW_DTYPE = (sample, ref, overlap, n_excl, n_match) from (summed) values."""
        _lib.check(self._lib.yh_run_batch_rows_unpack_device(self._h, slot, C.c_void_p(d_vals), cap_rows, C.c_void_p(d_rows),
                                                             C.c_void_p(d_n_rows)))

    # the subset words of a block in compact form: what a hash-range rank all-gathers between the two halves
    def run_batch_words_pack_device(self, d_words: int, d_packed: int, cap_words: int, n_planes: int = 1) -> None:
        """d_packed [words_packed_len(cap_words)] uint64 = count, the non-zero words of d_words [n_planes][N], their ids
        (plane * N + reference); n_planes = batch_planes(samples of the block)."""
        _lib.check(self._lib.yh_run_batch_words_pack_device(self._h, C.c_void_p(d_words), n_planes, C.c_void_p(d_packed), cap_words))

    def run_batch_words_unpack_device(self, d_gathered: int, n_ranks: int, cap_words: int, d_words_out: int, d_overflow: int,
                                      n_planes: int = 1) -> None:
        """d_words_out [n_planes][N] uint64 = OR of the n_ranks packed buffers at d_gathered; d_overflow [1] uint32 = some rank overflowed."""
        _lib.check(self._lib.yh_run_batch_words_unpack_device(self._h, C.c_void_p(d_gathered), n_ranks, n_planes, cap_words,
                                                              C.c_void_p(d_words_out), C.c_void_p(d_overflow)))

    def run_submit(self, slot: int, sample: np.ndarray, overlap: np.ndarray, n_excl: np.ndarray,
                   n_match: np.ndarray) -> None:
        """Queue one `yacht run` count call (upload, ordering check, kernels, download) without waiting;
        run_wait(slot) completes it.  Arrays must stay alive (and unchanged) until then; page-locked
        arrays (pinned_empty) make the copies overlap the kernels of the neighbouring calls."""
        assert sample.dtype == np.uint64 and sample.flags.c_contiguous
        for a in (overlap, n_excl, n_match):
            assert a.dtype == np.uint32 and a.flags.c_contiguous and a.size >= self.n_refs
        _lib.check(self._lib.yh_run_submit(self._h, slot, _ptr(sample), sample.size, _ptr(overlap), _ptr(n_excl),
                                           _ptr(n_match)))

    def run_wait(self, slot: int) -> None:
        _lib.check(self._lib.yh_run_wait(self._h, slot))

    def run_submit_packed(self, slot: int, packed: np.ndarray, rows: np.ndarray) -> None:
        """Queue one run step from a packed sample (pack_sample); its result comes back as compact rows -- one
        ROW_DTYPE record per reference with overlap > 0 -- in `rows`; run_wait_rows(slot) returns how many."""
        assert packed.dtype == np.uint8 and packed.flags.c_contiguous
        assert rows.dtype == ROW_DTYPE and rows.flags.c_contiguous
        _lib.check(self._lib.yh_run_submit_packed(self._h, slot, _ptr(packed), packed.size, _ptr(rows), rows.size))

    def run_submit_rows(self, slot: int, sample: np.ndarray, rows: np.ndarray) -> None:
        assert sample.dtype == np.uint64 and sample.flags.c_contiguous
        assert rows.dtype == ROW_DTYPE and rows.flags.c_contiguous
        _lib.check(self._lib.yh_run_submit_rows(self._h, slot, _ptr(sample), sample.size, _ptr(rows), rows.size))

    def run_wait_rows(self, slot: int) -> int:
        n = C.c_uint64(0)
        _lib.check(self._lib.yh_run_wait_rows(self._h, slot, C.byref(n)))
        return int(n.value)

    def run_rows(self, sample, packed: bool = True) -> np.ndarray:
        """run_counts as compact rows (ROW_DTYPE, ascending by reference): only the references with overlap > 0."""
        sample = _as_u64(sample)
        cap = max(1024, self.n_refs // 16)
        while True:
            rows = np.zeros(cap, dtype=ROW_DTYPE)
            if packed:
                self.run_submit_packed(0, pack_sample(sample), rows)
            else:
                self.run_submit_rows(0, sample, rows)
            n = C.c_uint64(0)
            rc = self._lib.yh_run_wait_rows(self._h, 0, C.byref(n))
            if rc == _lib.YH_ERR_CAPACITY:
                cap = int(n.value)
                continue
            _lib.check(rc)
            return rows[: int(n.value)]

    def run_rows_device(self, d_overlap: int, d_excl: int, d_match: int, d_rows: int, cap_rows: int, d_n_rows: int) -> None:
        _lib.check(self._lib.yh_run_rows_device(self._h, C.c_void_p(d_overlap), C.c_void_p(d_excl), C.c_void_p(d_match),
                                                C.c_void_p(d_rows), cap_rows, C.c_void_p(d_n_rows)))

    def run_batch(self, samples: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """run_counts for up to BATCH_MAX_SAMPLES (256) samples in one pass (needs the directory: not YH_DB_NO_DIRECTORY): three uint32
        arrays of shape [len(samples), n_refs]."""
        values, offsets = pack_csr(samples)
        b = len(samples)
        out = [np.zeros((b, self.n_refs), dtype=np.uint32) for _ in range(3)]
        _lib.check(self._lib.yh_run_batch(self._h, _ptr(values), _ptr(offsets), b, *(_ptr(a) for a in out)))
        return out[0], out[1], out[2]

    def run_batch_device(self, d_samples: int, d_offsets: int, n_samples: int, total_hashes: int, d_overlap: int,
                         d_excl: int, d_match: int) -> None:
        _lib.check(self._lib.yh_run_batch_device(self._h, C.c_void_p(d_samples), C.c_void_p(d_offsets), n_samples,
                                                 total_hashes, C.c_void_p(d_overlap), C.c_void_p(d_excl),
                                                 C.c_void_p(d_match)))

    # device-pointer forms (async on the handle's stream; raw addresses, e.g. tensor.data_ptr())
    def overlap_device(self, d_sample: int, n_sample: int, d_overlap: int) -> None:
        _lib.check(self._lib.yh_overlap_device(self._h, C.c_void_p(d_sample), n_sample, C.c_void_p(d_overlap)))

    def overlap_bsearch_device(self, d_sample: int, n_sample: int, d_overlap: int) -> None:
        _lib.check(self._lib.yh_overlap_bsearch_device(self._h, C.c_void_p(d_sample), n_sample,
                                                       C.c_void_p(d_overlap)))

    def overlap_indexed_device(self, d_sample: int, n_sample: int, d_overlap: int) -> None:
        _lib.check(self._lib.yh_overlap_indexed_device(self._h, C.c_void_p(d_sample), n_sample, C.c_void_p(d_overlap)))

    def run_indexed_device(self, d_sample: int, n_sample: int, d_overlap: int, d_excl: int, d_match: int) -> None:
        _lib.check(self._lib.yh_run_indexed_device(self._h, C.c_void_p(d_sample), n_sample, C.c_void_p(d_overlap),
                                                   C.c_void_p(d_excl), C.c_void_p(d_match)))

    def run_device(self, d_sample: int, n_sample: int, d_overlap: int, d_excl: int = 0, d_match: int = 0) -> None:
        _lib.check(self._lib.yh_run_device(self._h, C.c_void_p(d_sample), n_sample, C.c_void_p(d_overlap),
                                           C.c_void_p(d_excl), C.c_void_p(d_match)))

    def run_device_pipelined(self, d_sample: int, n_sample: int, d_overlap: int, d_excl: int, d_match: int) -> None:
        """run_device whose tail runs beside the next call's lookup; outputs are complete after run_device_join()
        (or any other query / synchronize on the handle).  Alternate at least two output buffers."""
        _lib.check(self._lib.yh_run_device_pipelined(self._h, C.c_void_p(d_sample), n_sample, C.c_void_p(d_overlap),
                                                     C.c_void_p(d_excl), C.c_void_p(d_match)))

    def run_device_join(self) -> None:
        _lib.check(self._lib.yh_run_device_join(self._h))

    # ---- yacht train -------------------------------------------------------------------------
    def nshared_device(self, d_out: int) -> None:
        """d_out[j] = shared hashes of reference j (device array of n_refs uint32): the row weights of the pairwise pass."""
        _lib.check(self._lib.yh_db_nshared_device(self._h, C.c_void_p(d_out)))

    def pairwise(self, c_thresh: float, row_begin: int = 0, row_end: Optional[int] = None):
        """Ordered pairs (i, j, |R_i ∩ R_j|) with !(count/|R_i| < c_thresh), sorted by (i, j)."""
        if row_end is None:
            row_end = self.n_refs
        n = C.c_uint64(0)
        _lib.check(self._lib.yh_pairwise(self._h, c_thresh, row_begin, row_end, 0, None, None, None, C.byref(n)))
        cap = max(int(n.value), 1)
        pi = np.zeros(cap, dtype=np.uint32)
        pj = np.zeros(cap, dtype=np.uint32)
        pc = np.zeros(cap, dtype=np.uint32)
        _lib.check(self._lib.yh_pairwise(self._h, c_thresh, row_begin, row_end, cap, _ptr(pi), _ptr(pj), _ptr(pc),
                                         C.byref(n)))
        k = int(n.value)
        return pi[:k], pj[:k], pc[:k]


BATCH_MAX_SAMPLES = 256  # include/yacht_hip.h: YH_BATCH_MAX_SAMPLES


def batch_planes(n_samples: int) -> int:
    """Planes of 64 samples the subset words of a batch take (include/yacht_hip.h: YH_BATCH_PLANES)."""
    return (int(n_samples) + 63) // 64


def csr_pack(values: np.ndarray, offsets: np.ndarray, threads: int = 0) -> np.ndarray:
    """A CSR of strictly ascending uint64 sketches -> the packed form yh_db_create_packed / RefDB.from_packed take (a uint64
    array: 8-byte aligned); ~5.7 bytes per hash for sketches of ~5 000 hashes at scaled = 1000."""
    lib = _lib.load()
    values = np.ascontiguousarray(values, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    need = C.c_uint64(0)
    _lib.check(lib.yh_csr_pack(_ptr(values), _ptr(offsets), n, None, 0, C.byref(need), threads))
    out = np.zeros((int(need.value) + 7) // 8, dtype=np.uint64)
    _lib.check(lib.yh_csr_pack(_ptr(values), _ptr(offsets), n, _ptr(out), out.nbytes, C.byref(need), threads))
    assert int(need.value) == out.nbytes
    return out


def csr_subset(packed: np.ndarray, rows: Sequence[int]) -> np.ndarray:
    """The listed sketches of a packed CSR, in that order, as a packed CSR of their own (yh_csr_subset: no hash is decoded)."""
    lib = _lib.load()
    assert packed.dtype == np.uint64 and packed.flags["C_CONTIGUOUS"]
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    need = C.c_uint64(0)
    _lib.check(lib.yh_csr_subset(_ptr(packed), packed.nbytes, _ptr(rows), rows.size, None, 0, C.byref(need)))
    out = np.zeros((int(need.value) + 7) // 8, dtype=np.uint64)
    _lib.check(lib.yh_csr_subset(_ptr(packed), packed.nbytes, _ptr(rows), rows.size, _ptr(out), out.nbytes, C.byref(need)))
    return out


def packed_offsets(packed: np.ndarray) -> np.ndarray:
    """The offsets[N + 1] a packed CSR carries behind its 64-byte header (a view: header word 1 = N)."""
    n = int(packed[1])
    return packed[8: 8 + n + 1]


def csr_unpack(packed: np.ndarray):
    """(values, offsets) of a packed CSR, on the host."""
    lib = _lib.load()
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    h, n = C.c_uint64(0), C.c_uint64(0)
    _lib.check(lib.yh_csr_unpack(_ptr(packed), packed.nbytes, None, 0, None, 0, C.byref(h), C.byref(n)))
    values = np.zeros(max(int(h.value), 1), dtype=np.uint64)
    offsets = np.zeros(int(n.value) + 1, dtype=np.uint64)
    _lib.check(lib.yh_csr_unpack(_ptr(packed), packed.nbytes, _ptr(values), int(h.value), _ptr(offsets), int(n.value), C.byref(h), C.byref(n)))
    return values[:int(h.value)], offsets


class PinnedArray:
    """A page-locked host array (yh_host_alloc) viewed as numpy; freed with the object."""

    def __init__(self, n: int, dtype):
        lib = _lib.load()
        self._lib = lib
        dt = np.dtype(dtype)
        p = C.c_void_p(0)
        _lib.check(lib.yh_host_alloc(C.byref(p), max(int(n) * dt.itemsize, 16)))
        self._p = p
        buf = (C.c_uint8 * (int(n) * dt.itemsize)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dt, count=int(n))

    def close(self) -> None:
        if self._p is not None and self._p.value:
            self.array = None
            self._lib.yh_host_free(self._p)
            self._p = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def train_select(sizes, pair_i, pair_j) -> np.ndarray:
    """Greedy size-ordered dedup (src/cpp/main.cpp:371-407): kept reference ids in walk order."""
    lib = _lib.load()
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    pair_i = np.ascontiguousarray(pair_i, dtype=np.uint32)
    pair_j = np.ascontiguousarray(pair_j, dtype=np.uint32)
    sel = np.zeros(max(sizes.size, 1), dtype=np.uint32)
    n = C.c_uint64(0)
    _lib.check(lib.yh_train_select(_ptr(sizes), sizes.size, _ptr(pair_i), _ptr(pair_j), pair_i.size, _ptr(sel),
                                   C.byref(n)))
    return sel[: int(n.value)]
