#!/usr/bin/env python3
"""Timing probe for the exclusive-count kernels at bench.py's scale: the pass with an all-zero
mask (pure dispatch / early-exit cost), with the sample's own mask, and with every reference
masked.  Prints the kernel milliseconds reported by yh_db_get_timing."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from yacht_amd import synth
    from yacht_amd.engine import RefDB

    n = 85_205
    vals, offsets, sample = synth.config3_device(seed=1002, n_refs=n, n_sample=1_000_000, device="cuda:0")
    torch.cuda.synchronize()
    def fresh():
        return RefDB.from_device(vals.data_ptr(), offsets.data_ptr(), n)

    db = fresh()
    info = db.info()
    print({k: info[k] for k in ("n_shared_hashes", "n_shared_postings") if k in info})
    smp = sample.cpu().numpy().view(np.uint64)
    ov = db.overlap(smp)
    for name, mask in (("zero", np.zeros(n, np.uint8)), ("own", (ov > 0).astype(np.uint8)), ("all", np.ones(n, np.uint8))):
        db.close()
        db = fresh()  # the timing ring averages over a handle's calls: one handle per case
        ts = []
        for _ in range(8):
            db.exclusive(mask, smp)
            ts.append(db.timing()["ms_exclusive_kernels"])
        print(name, "masked", int(mask.sum()), "ms_exclusive_kernels (ring avg)", round(float(ts[-1]), 4))
    db.close()


if __name__ == "__main__":
    main()
