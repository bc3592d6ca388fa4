"""GPU: `yacht sketch ref|sample` on the HIP sketcher, against the reference's wrapper semantics."""
import gzip
import os

import numpy as np
import pytest

from oracle import sketch_oracle as so

pytestmark = pytest.mark.gpu
FX = os.path.join(os.path.dirname(__file__), "golden", "fixtures")


def test_sketch_cli_ref_and_sample(hip_lib, tmp_path):
    from yacht_amd import cli, sigio

    genome = os.path.join(FX, "GCF_018918045.1_genomic.fna.gz")
    # folder -> one signature per genome file, named by the file stem (sourmash sketch fromfile)
    gdir = tmp_path / "genomes" / "sub"
    gdir.mkdir(parents=True)
    os.symlink(genome, gdir / "GCF_018918045.1_genomic.fna.gz")
    assert cli.main(["sketch", "ref", "--infile", str(tmp_path / "genomes"), "--kmer", "31", "--scaled", "1000",
                     "--outfile", str(tmp_path / "ref.sig.zip")]) == 0
    sigs = sigio.load_file_as_signatures(str(tmp_path / "ref.sig.zip"), ksize=31)
    assert len(sigs) == 1 and sigs[0].name == "GCF_018918045.1_genomic"
    assert (len(sigs[0].minhash), int(sigs[0].minhash.abundances.sum())) == (2452, 2453)
    # single file -> one signature per record (--singleton)
    recs = list(so.read_fasta(genome))[:5]
    small = tmp_path / "five.fa"
    small.write_bytes(b"".join(b">" + n.encode() + b"\n" + s + b"\n" for n, s in recs))
    assert cli.main(["sketch", "ref", "--infile", str(small), "--scaled", "100", "--outfile",
                     str(tmp_path / "five.sig.zip")]) == 0
    five = {s.name: s for s in sigio.load_file_as_signatures(str(tmp_path / "five.sig.zip"), ksize=31)}
    assert set(five) == {n for n, _ in recs}
    for n, s in recs:
        m, a = so.sketch_records([s], 31, 100)
        assert np.array_equal(five[n].minhash.mins, m) and np.array_equal(five[n].minhash.abundances, a)
    # sample: two FASTQ files merged into one signature with abundances
    reads = [recs[0][1][i:i + 100] for i in range(0, 3000, 50)]
    for tag, part in (("R1", reads[::2]), ("R2", reads[1::2])):
        with gzip.open(tmp_path / f"{tag}.fq.gz", "wb") as f:
            for i, r in enumerate(part):
                f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    assert cli.main(["sketch", "sample", "--infile", str(tmp_path / "R1.fq.gz"), str(tmp_path / "R2.fq.gz"), "--scaled",
                     "10", "--outfile", str(tmp_path / "sample.sig.zip")]) == 0
    smp = sigio.load_file_as_signatures(str(tmp_path / "sample.sig.zip"), ksize=31)
    assert len(smp) == 1
    m, a = so.sketch_records(reads, 31, 10)
    assert np.array_equal(smp[0].minhash.mins, m) and np.array_equal(smp[0].minhash.abundances, a)
    assert int(a.max()) >= 2 and sigio.zip_has_manifest(str(tmp_path / "sample.sig.zip"))
