#!/usr/bin/env python3
"""Feasibility probe: do two query contexts on two streams hide one sample's small tail kernels
behind the next sample's streaming kernel?  Uses two independent handles over the same arrays
(double memory, probe only) and compares steps/s with one handle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from yacht_amd import synth
    from yacht_amd.engine import RefDB

    n = 85_205
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    vals, offsets, sample = synth.config3_device(seed=1002, n_refs=n, n_sample=1_000_000, device="cuda:0")
    torch.cuda.synchronize()
    dbs = [RefDB.from_device(vals.data_ptr(), offsets.data_ptr(), n) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [torch.zeros(3, n, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    for db, st in zip(dbs, streams):
        db.set_stream(st.cuda_stream)

    def run(nctx):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            b = i % nctx
            dbs[b].run_device(sample.data_ptr(), sample.numel(), outs[b][0].data_ptr(), outs[b][1].data_ptr(),
                              outs[b][2].data_ptr())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    for nctx in (1, 2, 1, 2):
        run(nctx)  # warm
        print("contexts", nctx, "ms/step", round(run(nctx), 4))
    print("equal outputs", bool(torch.equal(outs[0], outs[1])))


if __name__ == "__main__":
    main()
