"""The packed sample format (yh_sample_pack / yh_sample_unpack, host code of the C ABI): round trips, sizes, refusals.
No GPU needed.  The device-side expansion and the compact rows are checked in tests/test_gpu_pipeline.py."""
import numpy as np
import pytest

from yacht_amd import _lib, synth
from yacht_amd.engine import pack_sample, unpack_sample


def _roundtrip(a):
    a = np.asarray(a, dtype=np.uint64)
    p = pack_sample(a)
    assert np.array_equal(unpack_sample(p), a)
    return p


def test_roundtrip_shapes():
    rng = np.random.default_rng(1)
    mh = synth.max_hash_for_scaled(1000)
    for n in (0, 1, 2, 255, 256, 257, 511, 512, 513, 100_000):
        _roundtrip(np.unique(rng.integers(0, mh, size=n, dtype=np.uint64))[:n])
    _roundtrip([0, 1, 2, 3])                                          # gaps of 0 bits
    _roundtrip(np.arange(1000, dtype=np.uint64) * np.uint64(7))       # constant gaps
    _roundtrip([0, 2**64 - 1])                                        # a 64-bit gap
    _roundtrip([5, 2**63, 2**63 + 1, 2**64 - 2, 2**64 - 1])
    big = np.concatenate([np.arange(300, dtype=np.uint64), np.array([2**64 - 1], dtype=np.uint64)])
    _roundtrip(big)                                                   # one block of tiny gaps, then a huge one


def test_size_of_a_metagenome_sketch():
    """10^6 uniform hashes at scaled = 1000: ~37 bits per gap + 16 bytes per 256 -> under 4.8 bytes per hash."""
    rng = np.random.default_rng(2)
    a = np.unique(rng.integers(0, synth.max_hash_for_scaled(1000), size=1_000_000, dtype=np.uint64))
    p = _roundtrip(a)
    assert p.size < 4.8 * a.size, p.size / a.size
    assert p.size <= _lib.load().yh_sample_pack_bound(a.size)


def test_refusals():
    with pytest.raises(_lib.YachtHipError) as e:
        pack_sample([3, 3])
    assert e.value.code == _lib.YH_ERR_UNSORTED
    with pytest.raises(_lib.YachtHipError):
        pack_sample(np.concatenate([np.arange(300, dtype=np.uint64), np.array([10], dtype=np.uint64)]))  # descent across blocks
    good = pack_sample(np.arange(0, 5000, 3, dtype=np.uint64))
    for bad in (good[:-8], good[:20], np.concatenate([good, np.zeros(8, np.uint8)])):
        with pytest.raises(_lib.YachtHipError) as e:
            unpack_sample(bad)
        assert e.value.code == _lib.YH_ERR_INVALID_ARG
    forged = good.copy()
    forged[0] ^= 1                                                    # magic
    with pytest.raises(_lib.YachtHipError):
        unpack_sample(forged)
    forged = good.copy()
    forged[32 + 12] = 99                                              # width of block 0 > 64
    with pytest.raises(_lib.YachtHipError):
        unpack_sample(forged)
    forged = good.copy()
    forged[32 + 16: 32 + 24] = 0                                      # base of block 1 = 0: blocks not ascending
    with pytest.raises(_lib.YachtHipError) as e:
        unpack_sample(forged)
    assert e.value.code == _lib.YH_ERR_UNSORTED
    small = np.zeros(8, dtype=np.uint8)
    with pytest.raises(_lib.YachtHipError) as e:
        pack_sample(np.arange(100, dtype=np.uint64), out=small)
    assert e.value.code == _lib.YH_ERR_CAPACITY


def test_csr_pack_roundtrip_and_refusals():
    """yh_csr_pack / yh_csr_unpack (the input of yh_db_create_packed): sketches of every length around the block size, empty ones,
    gaps of every width up to 64 bits; what is refused (unsorted sketches, truncated or forged blobs)."""
    from yacht_amd.engine import csr_pack, csr_unpack

    rng = np.random.default_rng(12)
    mh = synth.max_hash_for_scaled(1000)
    refs = [synth.random_sketch(rng, k, mh) for k in (0, 1, 2, 255, 256, 257, 511, 512, 513, 5000, 0, 3)]
    refs.append(np.array([0, 1, 2**63, 2**64 - 1], dtype=np.uint64))          # a 63-bit gap and the largest hash
    refs.append(np.arange(1000, dtype=np.uint64) * np.uint64(3))               # constant narrow gaps
    refs.append(np.arange(300, dtype=np.uint64))                               # gaps of width 0
    values, offsets = synth.pack(refs)
    blob = csr_pack(values, offsets)
    assert blob.dtype == np.uint64
    v2, o2 = csr_unpack(blob)
    assert np.array_equal(v2, values) and np.array_equal(o2, offsets.astype(np.uint64))
    for t in (1, 3):
        assert np.array_equal(csr_pack(values, offsets, threads=t), blob)     # the same bytes whatever the threads
    # a database of uniform sketches of ~5 000 hashes: ~5.7 bytes per hash
    big = [synth.random_sketch(rng, 5000, mh) for _ in range(40)]
    bv, bo = synth.pack(big)
    bb = csr_pack(bv, bo)
    assert 5.3 < bb.nbytes / bv.size < 6.0, bb.nbytes / bv.size
    assert np.array_equal(csr_unpack(bb)[0], bv)
    # an empty database and one of empty sketches
    for vals, offs in ((np.zeros(0, np.uint64), np.zeros(1, np.uint64)), (np.zeros(0, np.uint64), np.zeros(4, np.uint64))):
        e = csr_pack(vals, offs)
        ev, eo = csr_unpack(e)
        assert ev.size == 0 and np.array_equal(eo, offs)
    # refusals
    bad = values.copy()
    bad[int(offsets[9]) + 7] = bad[int(offsets[9]) + 6]                        # a repeated hash inside sketch 9
    with pytest.raises(_lib.YachtHipError):
        csr_pack(bad, offsets)
    with pytest.raises(_lib.YachtHipError):
        csr_unpack(blob[:-1])                                                  # truncated
    forged = blob.copy()
    forged[1] += np.uint64(1)                                                  # n_refs off by one
    with pytest.raises(_lib.YachtHipError):
        csr_unpack(forged)
    forged = blob.copy()
    n_refs = len(refs)
    first_block_entry = 8 + (n_refs + 1)                                       # header (8 words) + offsets -> block 0: base, word_off, width
    forged[first_block_entry + 3 + 1] = np.uint64(2**40)                       # block 1's word_off far outside the payload
    with pytest.raises(_lib.YachtHipError):
        csr_unpack(forged)
    #  a block table whose word offsets are in bounds but NOT the running prefix -- two blocks swapped, one that
    # points at an earlier block's words -- is refused by the view itself: the chunked upload copies payload ranges by these
    # offsets, and a non-monotone table would leave blocks decoding words that were never copied
    blocks = [first_block_entry + 3 * b for b in range(4)]
    w = [int(blob[e + 1]) for e in blocks]
    assert w[0] == 0 and w[1] <= w[2] <= w[3] and w[3] > w[2]
    forged = blob.copy()
    forged[blocks[3] + 1] = np.uint64(w[2])                                    # block 3 re-reads block 2's words
    with pytest.raises(_lib.YachtHipError):
        csr_unpack(forged)
    forged = blob.copy()
    forged[blocks[2] + 1], forged[blocks[3] + 1] = np.uint64(w[3]), np.uint64(w[2])   # swapped
    with pytest.raises(_lib.YachtHipError):
        csr_unpack(forged)
    forged = blob.copy()
    forged[blocks[0]] = np.uint64(2**64 - 1)                                   # block 0's base above the header's largest hash
    forged[7 - 2] = np.uint64(2**63)                                           # (header word 5 = max_hash)
    with pytest.raises(_lib.YachtHipError):
        csr_unpack(forged)
