#!/usr/bin/env python3
"""Randomized parity run on the GPU box: HIP path vs the CPU oracle over many shapes.

    python tests/tools/fuzz_parity.py [--seconds 300] [--seed 0]

Every round draws a database shape (number of references, sketch sizes, clusters that share
hashes, hash range from a few thousand values to the full 64 bits, duplicates of whole sketches,
empty sketches) and a sample (noise, hits, the union of everything, nothing), and compares
overlap, the run step's exclusive counts (fused path), exclusive counts for an arbitrary subset
(general path), the independent bsearch kernel, the packed / rows / pipelined forms of the step, a batch of five
samples through the batched run and -- on small rounds -- the pairwise list with the oracle, bit for bit.  Prints one JSON line; exit code 1 on the first difference.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from oracle import oracle  # noqa: E402  (the checker)
from yacht_amd import _lib, synth  # noqa: E402
import torch  # noqa: E402

from yacht_amd.engine import RefDB, YH_DB_KEEP_CSR, YH_DB_PAIRWISE_ONLY  # noqa: E402


def draw(rng):
    n_refs = int(rng.choice([1, 2, 7, 64, 300, 1500, 6000]))
    top_bits = int(rng.choice([13, 20, 34, 54, 64]))
    top = (1 << top_bits) - 1
    median = float(rng.choice([3, 40, 400, 3000]))
    median = min(median, max(1.0, top / 4))
    refs = []
    while len(refs) < n_refs:
        kind = rng.random()
        size = int(np.clip(rng.lognormal(np.log(median), 0.7), 0, min(20000, top // 2)))
        base = np.unique(rng.integers(0, top, size=size, dtype=np.uint64, endpoint=True))
        refs.append(base)
        if kind < 0.25 and base.size:  # a cluster around it
            for _ in range(int(rng.integers(1, 6))):
                keep = base[rng.random(base.size) < rng.choice([1.0, 0.9, 0.5, 0.1])]
                extra = np.unique(rng.integers(0, top, size=int(rng.integers(0, max(2, size // 3))), dtype=np.uint64,
                                               endpoint=True))
                refs.append(np.unique(np.concatenate([keep, extra])))
        elif kind < 0.30:
            refs.append(np.zeros(0, np.uint64))
    refs = refs[:n_refs]
    values, offsets = synth.pack(refs)
    allh = np.unique(values) if values.size else values
    mode = rng.random()
    noise = np.unique(rng.integers(0, top, size=int(rng.choice([0, 10, 1000, 50000])), dtype=np.uint64, endpoint=True))
    if mode < 0.15 or allh.size == 0:
        sample = noise
    elif mode < 0.3:
        sample = np.unique(np.concatenate([allh, noise]))
    else:
        present = rng.random(len(refs)) < rng.choice([0.01, 0.1, 0.5])
        parts = [r[rng.random(r.size) < rng.choice([0.05, 0.5, 1.0])] for r, p in zip(refs, present) if p and r.size]
        sample = np.unique(np.concatenate(parts + [noise])) if parts else noise
    return refs, values, offsets, sample, top_bits


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-batch", action="store_true", help="skip the batched-run check of every round")
    ap.add_argument("--packed", action="store_true", help="every handle from the database's packed form (yh_csr_pack -> yh_db_create_packed; "
                                                          "with YH_DEBUG_TUNING=1 YH_UPLOAD_CHUNK_MIN=1 the chunked expansion on the device)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    rounds = 0
    hashes = 0
    c = 0.95 ** 31
    def make_db(values, offsets, flags):
        if not args.packed:
            return RefDB(values, offsets, flags=flags)
        from yacht_amd.engine import csr_pack, csr_unpack

        blob = csr_pack(values, offsets, threads=int(rng.integers(1, 4)))
        v2, o2 = csr_unpack(blob)
        assert np.array_equal(v2, values) and np.array_equal(o2, offsets.astype(np.uint64)), "csr_pack / csr_unpack round trip"
        return RefDB.from_packed(blob, sizes=np.diff(offsets).astype(np.uint32), flags=flags)

    while time.time() < t_end:
        refs, values, offsets, sample, top_bits = draw(rng)
        n = len(refs)
        sizes = np.diff(offsets).astype(np.uint32)
        tag = {"round": rounds, "n_refs": n, "n_hashes": int(values.size), "n_sample": int(sample.size), "top_bits": top_bits}
        try:
            with make_db(values, offsets, YH_DB_KEEP_CSR) as db:
                want = oracle.overlap(values, offsets, sample)
                assert np.array_equal(db.overlap(sample, method="bsearch"), want), "bsearch overlap"
                we, wm = oracle.exclusive(values, offsets, want > 0, sample)
                mask = rng.random(n) < 0.5
                xe, xm = oracle.exclusive(values, offsets, mask, sample)
                # every query through the streaming kernel, the sample-driven kernel, and the library's choice
                for mode, name in ((_lib.YH_LOOKUP_STREAM, "stream"), (_lib.YH_LOOKUP_INDEXED, "indexed"), (_lib.YH_LOOKUP_AUTO, "auto")):
                    db.set_lookup(mode)
                    assert np.array_equal(db.overlap(sample), want), "overlap " + name
                    ov, e, m = db.run_counts(sample)
                    assert np.array_equal(ov, want) and np.array_equal(e, we) and np.array_equal(m, wm), "run counts " + name
                    ge, gm = db.exclusive(mask, sample)
                    assert np.array_equal(ge, xe) and np.array_equal(gm, xm), "exclusive for a subset " + name
                    # the same step through the packed upload + compact rows, and as one-launch pipelined device steps
                    # (three of them: the sample's own three stages ride in three consecutive launches)
                    keep = np.flatnonzero(want)
                    for packed in (True, False):
                        rows = db.run_rows(sample, packed=packed)
                        assert np.array_equal(rows["ref"], keep) and np.array_equal(rows["overlap"], want[keep]) and \
                            np.array_equal(rows["n_excl"], we[keep]) and np.array_equal(rows["n_match"], wm[keep]), \
                            f"rows (packed={packed}) " + name
                    if n:
                        d_s = torch.from_numpy(np.ascontiguousarray(sample).view(np.int64).copy()).cuda() if sample.size else \
                            torch.zeros(1, dtype=torch.int64, device="cuda")
                        bufs = [torch.zeros(3, n, dtype=torch.int32, device="cuda") for _ in range(3)]
                        for b in bufs:
                            db.run_device_pipelined(d_s.data_ptr(), int(sample.size), b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
                        db.run_device_join()
                        db.synchronize()
                        for b in bufs:
                            g = b.cpu().numpy().view(np.uint32)
                            assert np.array_equal(g[0], want) and np.array_equal(g[1], we) and np.array_equal(g[2], wm), \
                                "pipelined steps " + name
                # a batch through the batched run (needs the bucket table: handles of a non-empty database have it):
                # the sample, a random half of it, a sample of absent hashes, an empty one and the sample again
                if n and values.size and not args.no_batch:
                    half = sample[rng.random(sample.size) < 0.5]
                    absent = np.unique(rng.integers(1, 2 ** 62, size=int(rng.integers(1, 2000)), dtype=np.uint64))
                    batch = [sample, half, absent, np.zeros(0, np.uint64), sample]
                    # round 6: up to 256 samples per pass (subset words in planes of 64 samples) -- the five patterns repeated in a
                    # random order to a random batch size; the oracle sees each pattern once
                    n_b = int(rng.choice([5, 5, 64, 65, 130, 200, 256])) if n <= 2000 else 5
                    pick = list(range(5)) + [int(x) for x in rng.integers(0, 5, size=n_b - 5)]
                    bo, be, bm = db.run_batch([batch[k] for k in pick])
                    zero = np.zeros(n, np.uint32)
                    expect = {0: (want, we, wm), 3: (zero, zero, zero), 4: (want, we, wm)}  # (the oracle saw these already)
                    for k in (1, 2):
                        w_ov = oracle.overlap(values, offsets, batch[k])
                        expect[k] = (w_ov,) + tuple(oracle.exclusive(values, offsets, w_ov > 0, batch[k]))
                    for pos, k in enumerate(pick):
                        w_ov, w_e, w_m = expect[k]
                        assert np.array_equal(bo[pos], w_ov) and np.array_equal(be[pos], w_e) and np.array_equal(bm[pos], w_m), \
                            f"batched run, sample {pos} (pattern {k}) of a batch of {n_b}"
                if values.size < 400_000:
                    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c, threads=4)
                    gi, gj, gc = db.pairwise(c)
                    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc), "pairs"
                    assert db.index_stats() == wstats, "index stats"
                    # `yacht train`'s own handle: the fused path (the sort's last pass writes the records), a row range of it,
                    # and its shared-hash counts per reference against the full handle's
                    with make_db(values, offsets, YH_DB_PAIRWISE_ONLY) as tdb:
                        ti, tj, tc = tdb.pairwise(c)
                        assert np.array_equal(ti, wi) and np.array_equal(tj, wj) and np.array_equal(tc, wc), "pairs (train handle)"
                        assert tdb.index_stats() == wstats, "index stats (train handle)"
                        if n > 2:
                            cut = int(rng.integers(1, n))
                            parts = [tdb.pairwise(c, 0, cut), tdb.pairwise(c, cut, n)]
                            for k, w_k in ((0, wi), (1, wj), (2, wc)):
                                assert np.array_equal(np.concatenate([q[k] for q in parts]), w_k), "pairs (train handle, row ranges)"
                        if n:
                            ns_t = torch.zeros(n, dtype=torch.int32, device="cuda:0")
                            ns_d = torch.zeros(n, dtype=torch.int32, device="cuda:0")
                            torch.cuda.synchronize()
                            tdb.nshared_device(ns_t.data_ptr()); tdb.synchronize()
                            db.nshared_device(ns_d.data_ptr()); db.synchronize()
                            assert torch.equal(ns_t, ns_d), "shared hashes per reference (train handle)"
        except AssertionError as ex:
            print(json.dumps({"fuzz": "FAILED", "what": str(ex), **tag, "seed": args.seed}), flush=True)
            np.savez("gpurun_out/fuzz_failure.npz", values=values, offsets=offsets, sample=sample)
            return 1
        rounds += 1
        hashes += int(values.size)
    print(json.dumps({"fuzz": "ok", "rounds": rounds, "reference_hashes_checked": hashes, "seconds": args.seconds,
                      "seed": args.seed}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
