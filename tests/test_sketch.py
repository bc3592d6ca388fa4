"""FracMinHash sketching: the CPU restatement against the reference's known answers (CPU), and the
HIP kernel against the restatement (GPU)."""
import os

import numpy as np
import pytest

from oracle import sketch_oracle as so

FX = os.path.join(os.path.dirname(__file__), "golden", "fixtures")
# (distinct hashes, total k-mers) of demo/ref_genomes sketched by the reference's own pipeline
# (k=31, scaled=1000, abund): columns num_unique_kmers_in_genome_sketch / num_total_kmers_in_genome_sketch
# of tests/testdata/standardize_output_testdata/results/result.xlsx in the reference tree
KAT = {"GCF_018918235.1": (2319, 2323), "GCF_018918045.1": (2452, 2453),
       "GCF_018918095.1": (3009, 3035), "GCF_018918185.1": (2832, 2844)}


def test_sketch_oracle_reference_known_answers():
    for g, (distinct, total) in KAT.items():
        mins, ab = so.sketch_fasta(os.path.join(FX, f"{g}_genomic.fna.gz"))
        assert (len(mins), int(ab.sum())) == (distinct, total)
        assert bool(np.all(mins[1:] > mins[:-1])) and int(mins[-1]) <= 18446744073709552
    assert so.max_hash_for_scaled(1000) == 18446744073709552   # the max_hash in the reference's .sig fixtures
    # a hash VALUE, not only counts: sourmash's own published known answer for its k-mer hash (tests/test_minhash.py
    # there: hash_murmur("ACG") == 1731421407650554201; seed 42, first 64 bits of MurmurHash3_x64_128) -- the
    # third-party definition the reference's sketches come from (sketch_ref_genomes.py:25,61)
    assert int(so.murmur3_x64_128_h1(np.frombuffer(b"ACG", dtype=np.uint8).reshape(1, 3), 42)[0]) == 1731421407650554201
    assert so.kmer_hashes(b"ACG", 3).tolist() == [1731421407650554201] and so.kmer_hashes(b"CGT", 3).tolist() == [1731421407650554201]


def test_sketch_oracle_small_properties():
    # reverse complement gives the same sketch; N breaks windows; lower case is accepted
    seq = b"ACGTTGCAAGGCTTAACCGGATATCGCGATTACGGATCCGATTTAGGCATCGATCGGGATATCCGAT"
    rc = seq[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    a = np.sort(so.kmer_hashes(seq, 21))
    assert np.array_equal(a, np.sort(so.kmer_hashes(rc, 21)))
    assert np.array_equal(a, np.sort(so.kmer_hashes(seq.lower(), 21)))
    assert so.kmer_hashes(seq[:20], 21).size == 0
    withn = seq[:30] + b"N" + seq[31:]
    assert so.kmer_hashes(withn, 21).size == len(seq) - 21 + 1 - 21


@pytest.mark.gpu
def test_hip_sketch_equals_oracle(hip_lib):
    from yacht_amd import sketch

    path = os.path.join(FX, "GCF_018918235.1_genomic.fna.gz")
    # (k <= 32 / k <= 64: the 2-bit kernel on a 64- / 128-bit word, every shape of the MurmurHash3 tail; k = 70: the byte-wise kernel)
    for k, scaled in ((31, 1000), (21, 100), (51, 1000), (32, 1000), (16, 100), (15, 100), (17, 200), (8, 20), (24, 100), (25, 100),
                      (33, 500), (40, 500), (41, 500), (48, 500), (49, 500), (56, 500), (63, 500), (64, 500), (70, 500)):
        want_m, want_a = so.sketch_fasta(path, k, scaled)
        sig = sketch.sketch_file(path, k, scaled)
        assert np.array_equal(sig.minhash.mins, want_m)
        assert np.array_equal(sig.minhash.abundances, want_a)
    for g, kat in KAT.items():  # the four known answers of the reference's own sketches through the HIP kernel
        sg = sketch.sketch_file(os.path.join(FX, f"{g}_genomic.fna.gz"), 31, 1000)
        assert (len(sg.minhash), int(sg.minhash.abundances.sum())) == kat
    assert sketch.hash_kmers([b"ACG"], 3, 1).tolist() == [1731421407650554201]  # sourmash's published hash value
    sig = sketch.sketch_file(path, 31, 1000)
    assert (len(sig.minhash), int(sig.minhash.abundances.sum())) == KAT["GCF_018918235.1"]
    assert sig.minhash.scaled == 1000 and sig.name.startswith("NZ_JAHLQE010000140.1")
    # every window kept (scaled = 1): exercises the LDS-list overflow path, incl. N, lower case, short records
    recs = [b"ACGTNACGTTGCAAGGCTTAACCGGATATCGCGATTACGG", b"acgtacgtacgtacgtacgtacgtaaa", b"ACG", b""]
    got = np.sort(sketch.hash_kmers(recs, 11, 1))
    want = np.sort(np.concatenate([so.kmer_hashes(r, 11) for r in recs]))
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_host_call_hashes_the_sequence_while_it_arrives(hip_lib):
    """yh_sketch_dna uploads a long sequence in 32 MiB pieces and hashes the windows that are complete while the next piece
    is on the bus: the same multiset of kept hashes as ONE launch over the resident sequence (yh_sketch_dna_device), and as
    the oracle on a stretch across the first piece boundary."""
    import ctypes as C

    import torch

    from yacht_amd import _lib, sketch

    rng = np.random.default_rng(3)
    n = 40_000_000
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n, dtype=np.uint8)]
    seq[rng.integers(0, n, size=500)] = ord("N")
    seq[(32 << 20) - 40:(32 << 20) - 35] = ord("n")     # bad bases right at a piece boundary
    seq[(32 << 20) + 3] = ord("-")
    for k, scaled in ((31, 1000), (51, 2000), (70, 2000)):
        host = np.sort(sketch.hash_kmers([seq], k, scaled))
        lib = _lib.load()
        mh = sketch.max_hash_for_scaled(scaled)
        d_seq = torch.from_numpy(seq).cuda()
        d_out = torch.zeros(host.size + 1000, dtype=torch.int64, device="cuda")
        d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        _lib.check(lib.yh_sketch_dna_device(C.c_void_p(d_seq.data_ptr()), n, k, sketch.DEFAULT_SEED, mh, d_out.numel(),
                                            C.c_void_p(d_out.data_ptr()), C.c_void_p(d_cnt.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        dev = np.sort(d_out[: int(d_cnt.item())].cpu().numpy().view(np.uint64))
        assert np.array_equal(host, dev), (k, scaled, host.size, dev.size)
    lo, hi = (32 << 20) - 300_000, (32 << 20) + 300_000
    want = so.kmer_hashes(seq[lo:hi].tobytes(), 31)
    want = np.sort(want[want <= np.uint64(sketch.max_hash_for_scaled(1000))])
    got = np.sort(sketch.hash_kmers([seq[lo:hi]], 31, 1000))
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_sketched_genomes_through_the_engine(hip_lib, tmp_path):
    """sketch -> .sig.zip -> train -> run on the two demo genomes: a genome sampled at 50 % is
    found, the other is not reported as overlapping more than by chance."""
    from yacht_amd import cli, sigio, sketch

    paths = [os.path.join(FX, f"{g}_genomic.fna.gz") for g in KAT]
    refzip = tmp_path / "refs.sig.zip"
    sigs = sketch.sketch_files(paths, str(refzip))
    assert [len(s.minhash) for s in sigs] == [KAT[g][0] for g in KAT]
    rng = np.random.default_rng(0)
    keep = sigs[0].minhash.mins[rng.random(len(sigs[0].minhash)) < 0.5]
    sample = sigio.make_signature(keep, 31, 1000, name="sample", abundances=np.ones(keep.size, np.int64))
    sample_zip = tmp_path / "sample.sig.zip"
    sigio.write_sig_zip([sample], str(sample_zip))
    out = tmp_path / "out"
    out.mkdir()
    assert cli.main(["train", "--ref_file", str(refzip), "--ksize", "31", "--prefix", "demo", "--outdir", str(out),
                     "--num_threads", "1"]) == 0
    assert cli.main(["run", "--json", str(out / "demo_config.json"), "--sample_file", str(sample_zip),
                     "--min_coverage_list", "1", "0.1", "--outdir", str(tmp_path), "--num_threads", "1"]) == 0
    import pandas as pd

    res = pd.read_csv(tmp_path / "results" / "result_all.txt", sep="\t")
    hit = res[(res["min_coverage"] == 0.1)]
    assert list(hit["organism_name"]) == [sigs[0].name]
    assert bool(hit["in_sample_est"].iloc[0]) and int(hit["num_matches"].iloc[0]) == keep.size
