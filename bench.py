#!/usr/bin/env python3
"""bench.py — containment queries/sec of the `yacht run` hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the device-side `yacht run` counts over one sample sketch:
overlap of the sample with every reference of the rank's shard (streaming lookup kernel, R1),
mask = overlap > 0, subset-exclusive hash counts (R2), and — for N > 1 — one RCCL all-gather of
the per-reference counts.  Inputs are resident in HBM before the timed region.

Workload (config.workload): BASELINE.json configs[2] — GTDB-rs214-representatives scale,
85 205 synthetic reference sketches (k=31, scaled=1000, sizes LogNormal(ln 3300, 0.6) in
[300, 15000]) per GPU against one ~1 M-hash sample; the metric "ref-sketch containment
queries/sec" counts one (sample, reference) intersection as one query.  N > 1 is weak scaling:
every rank holds its own 85 205-reference shard of an N x 85 205 database, the sample is
replicated, and no collective sits on the data path except the final gather of counts.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the
dominant kernel (k_stream_lookup) and `cpu_baseline` (the oracle's C++ restatement,
all host cores, same workload, also used as the full-size bit-exact parity check).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="gtdb_rs214_scale", choices=["gtdb_rs214_scale", "config2_1000refs"])
    ap.add_argument("--refs", type=int, default=0, help="override references per GPU (testing only)")
    ap.add_argument("--sample-hashes", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and its parity check)")
    ap.add_argument("--overlap-only", action="store_true", help="time R1 only (diagnostic; not the reported metric)")
    ap.add_argument("--seed", type=int, default=1002)
    ap.add_argument("--present", type=int, default=200, help="genomes present in the sample (diagnostic)")
    ap.add_argument("--no-indexed", action="store_true", help="skip the extra measurement of the sample-driven path")
    ap.add_argument("--sync-gather", action="store_true", help="N>1: blocking all-gather inside every step (no overlap)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to test the N>1 logic)")
    ap.add_argument("--share-gpu", action="store_true", help="testing: all ranks use cuda:0 (1-GPU box, gloo backend)")
    return ap.parse_args()


def main() -> int:
    args = parse_args()
    # stdout carries the ONE JSON line and nothing else: libraries that print banners at start-up
    # (RCCL prints its version block to stdout when a communicator is created) are sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks on this pool
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)", file=sys.stderr)
        return 2
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from yacht_amd import build, synth
    from yacht_amd.engine import RefDB, YH_DB_DEFAULT, YH_DB_FULL_INDEX

    if not os.path.exists(build.LIB_PATH):
        build.build_lib()

    # ---- synthetic workload, generated in HBM ---------------------------------------------------------
    if args.workload == "gtdb_rs214_scale":
        n_refs = args.refs or 85_205
        values, offsets, sample = synth.config3_device(seed=args.seed + 7919 * rank, n_refs=n_refs,
                                                       n_sample=args.sample_hashes, device=str(dev),
                                                       n_present=args.present)
        workload = f"GTDB-rs214-scale synthetic: {n_refs} refs/GPU k=31 scaled=1000 vs 1 sample"
    else:
        n_refs = args.refs or 1000
        values, offsets, sample = synth.config3_device(seed=args.seed + 7919 * rank, n_refs=n_refs,
                                                       n_sample=args.sample_hashes, device=str(dev), median=5000.0,
                                                       sigma=0.35, lo=500, hi=20000, cluster_frac=0.0, n_present=50)
        workload = f"configs[1]: {n_refs} refs/GPU (~5000 hashes) vs 1 sample"
    if world > 1:  # the sample is replicated: every rank queries rank 0's sample
        n_s = torch.tensor([sample.numel()], device=dev, dtype=torch.int64)
        dist.broadcast(n_s, 0)
        if rank != 0:
            sample = torch.empty(int(n_s.item()), device=dev, dtype=torch.int64)
        dist.broadcast(sample, 0)
    H = int(values.numel())
    n_sample = int(sample.numel())
    torch.cuda.synchronize()

    db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_refs, device=local_rank,
                           flags=YH_DB_DEFAULT if args.no_indexed else YH_DB_FULL_INDEX)
    info = db.info()
    # Everything of the timed region runs on ONE explicit stream: the library's kernels are queued
    # on it (a null handle — torch's default stream — would mean "the library's own stream"), and
    # RCCL orders the all-gather after whatever is on torch's current stream.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    db.set_stream(stream.cuda_stream)
    assert stream.cuda_stream != 0
    # Two sets of count buffers: the all-gather of sample k (async, on RCCL's stream, ordered after
    # sample k's kernels) overlaps the kernels of sample k+1; a buffer set is reused only after the
    # collective that reads it has completed (work.wait() = stream-level wait, no host block on RCCL).
    NBUF = 2 if world > 1 and not args.sync_gather else 1
    counts_b = [torch.zeros((3, n_refs), device=dev, dtype=torch.int32) for _ in range(NBUF)]  # overlap, n_excl, n_match
    gathered_b = [torch.zeros((world * 3, n_refs), device=dev, dtype=torch.int32) if world > 1 else None
                  for _ in range(NBUF)]  # concatenation layout
    pending = [None] * NBUF
    p_sample = sample.data_ptr()
    state = {"i": 0}

    def step():
        b = state["i"] % NBUF
        state["i"] += 1
        c = counts_b[b]
        with torch.cuda.stream(stream):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
            db.run_device(p_sample, n_sample, c[0].data_ptr(), 0 if args.overlap_only else c[1].data_ptr(),
                          0 if args.overlap_only else c[2].data_ptr())
            if world > 1:
                if NBUF > 1:
                    pending[b] = dist.all_gather_into_tensor(gathered_b[b], c, async_op=True)
                else:
                    dist.all_gather_into_tensor(gathered_b[b], c)

    def drain():
        with torch.cuda.stream(stream):
            for b in range(NBUF):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    fence()
    db.timing()  # drop the warm-up launches from the kernel-duration ring
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = db.timing()  # mean over the (up to 256) most recent launches of the timed region
    last = (state["i"] - 1) % NBUF
    counts, gathered = counts_b[last], gathered_b[last]
    if world > 1:  # every buffer set: the gathered block of this rank must be this rank's counts
        for b in range(NBUF):
            assert bool(torch.equal(gathered_b[b].view(world, 3, n_refs)[rank], counts_b[b])), \
                "all-gather ran ahead of the kernels"

    # ---- extra: the same step through the sample-driven path (work ~ |S| instead of streaming the
    # database).  Not the reported `value` this round: the headline stays on the streaming kernel
    # the north star describes; both are exact and both are checked against the oracle below.
    indexed = None
    counts_idx = None
    if not args.no_indexed and not args.overlap_only:
        counts_idx = torch.zeros((3, n_refs), device=dev, dtype=torch.int32)
        pi0, pi1, pi2 = (counts_idx[k].data_ptr() for k in range(3))

        def step_idx():
            with torch.cuda.stream(stream):
                db.run_indexed_device(p_sample, n_sample, pi0, pi1, pi2)
                if world > 1:
                    dist.all_gather_into_tensor(gathered, counts_idx)  # (this extra measurement keeps the blocking form)

        for _ in range(args.warmup):
            step_idx()
        fence()
        db.timing()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_idx()
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        tm_idx = db.timing()
        indexed = {"ms_per_step": round(1e3 * el / args.steps, 4),
                   "value": round(n_refs * world / (el / args.steps), 1), "unit": "queries/s",
                   "lookup_kernel_ms_avg": round(float(tm_idx["ms_overlap_kernel"]), 4),
                   "exclusive_kernels_ms_avg": round(float(tm_idx["ms_exclusive_kernels"]), 4),
                   "note": "k_index_lookup: one lane per sample hash through the distinct-hash directory "
                           "(YH_DB_FULL_INDEX); equals the streaming path bit for bit"}
        indexed["equals_streaming_path"] = bool(torch.equal(counts_idx, counts))

    ms_per_step = 1e3 * elapsed / args.steps
    total_refs = n_refs * world
    value = total_refs / (elapsed / args.steps)

    # ---- roofline of the dominant kernel (the streaming lookup) ------------------------------------------
    # Bytes one launch HAS to move in the layout the kernel reads (yh_db_info.stream_bytes: one delta
    # byte per (hash, reference) pair + an 8-byte header per 1024 for the default hash-sorted delta
    # stream; 3 bytes per pair for YH_STREAM=keys; 8 for YH_WIDE_KEYS=1) plus the 8-byte sample hashes
    # staged once.  `achieved` is that figure over the measured duration: the physical HBM rate the
    # kernel sustains, comparable with `peak` and with `traffic` (PMC).  SURVEY.md 8d's one-touch
    # formula (8 B per reference hash: 8(H+|S|) + 8(N+1) + 4N) is reported beside it; it exceeds the
    # peak because the kernel does not read 8 bytes per hash any more.
    layout = int(info.get("stream_layout", 0))
    kernel_name = {1: "k_stream_lookup", 2: "k_tile_lookup_keys", 3: "k_tile_lookup<OverlapHit>"}.get(layout, "?")
    survey_bytes = 8 * (H + n_sample) + 8 * (n_refs + 1) + 4 * n_refs
    alg_bytes = int(info.get("stream_bytes", 0)) + 8 * n_sample
    k_ms = float(timing["ms_overlap_kernel"])
    achieved = (alg_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
    survey_rate = (survey_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
    roofline = {
        "bound": "hbm",
        "kernel": kernel_name,
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None,
        "algorithmic_bytes_per_launch": alg_bytes,
        "bytes_per_ref_hash": round(int(info.get("stream_bytes", 0)) / max(H, 1), 4),
        "kernel_ms_avg": round(k_ms, 4),
        "exclusive_kernels_ms_avg": round(float(timing["ms_exclusive_kernels"]), 4),
        "survey_formula": {"bytes_per_launch": survey_bytes, "GBps": round(survey_rate, 1),
                           "frac": round(survey_rate / HBM_PEAK_GBS, 4),
                           "note": "8 B per reference hash as SURVEY.md 8d counts; the kernel streams "
                                   f"{round(int(info.get('stream_bytes', 0)) / max(H, 1), 3)} B per hash"},
    }
    traffic_file = os.path.join(ROOT, "profiles", "traffic_r01.json")  # written by scripts/make_traffic_json.py
    if os.path.exists(traffic_file):  # HBM bytes per launch from the rocprofv3 --pmc passes (profiles/README.md)
        try:
            with open(traffic_file) as f:
                tr = json.load(f)
            if tr.get("n_hashes") == H and tr.get("kernel") == roofline["kernel"]:
                roofline["traffic"] = tr.get("hbm_bytes_per_launch")
        except Exception:
            pass

    # ---- CPU baseline + full-size parity (rank 0, its own shard) ---------------------------------------
    cpu_baseline = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle  # the checker; never the thing measured as `value`

        h_values = values.cpu().numpy().view(np.uint64)
        h_offsets = offsets.cpu().numpy().view(np.uint64)
        h_sample = sample.cpu().numpy().view(np.uint64)
        cores = oracle.hardware_threads()
        t0 = time.perf_counter()
        want_ov = oracle.overlap(h_values, h_offsets, h_sample, threads=cores)
        t_ov = time.perf_counter() - t0
        t0 = time.perf_counter()
        want_e, want_m = oracle.exclusive(h_values, h_offsets, want_ov > 0, h_sample)
        t_ex = time.perf_counter() - t0
        got = counts.cpu().numpy().view(np.uint32)
        parity = bool(np.array_equal(got[0], want_ov))
        if not args.overlap_only:
            parity = parity and bool(np.array_equal(got[1], want_e)) and bool(np.array_equal(got[2], want_m))
        t_cpu = t_ov if args.overlap_only else t_ov + t_ex
        cpu_baseline = {
            "value": round(n_refs / t_cpu, 1),
            "unit": "queries/s",
            "cores": cores,
            "kind": "port",
            "sample": f"whole rank-0 workload once ({n_refs} refs, {H} hashes): overlap {t_ov:.2f} s on {cores} "
                      f"threads + exclusive {t_ex:.2f} s on 1 thread",
        }

    if rank == 0:
        out = {
            "metric": "ref-sketch containment queries/sec (yacht run)",
            "value": round(value, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "refs_per_gpu": n_refs,
                "ref_hashes_per_gpu": H,
                "sample_hashes": n_sample,
                "stream_layout": {1: "hash-sorted delta stream", 2: "packed 24-bit keys", 3: "64-bit hashes"}.get(layout, "none"),
                "stream_bytes": int(info.get("stream_bytes", 0)),
                "shared_hashes": info["n_shared_distinct"],
                "db_build_ms": round(float(timing["ms_db_build"]), 2),
                "db_hbm_bytes": info["device_bytes"],
                "step": "overlap" if args.overlap_only else "overlap + exclusive counts" + ((" + all_gather" + ("" if args.sync_gather else " (overlapped with the next sample)")) if world > 1 else ""),
                "parallelism": f"refs sharded x{world}",
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "parity_bit_exact": parity,
            "indexed_path": indexed,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    db.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and indexed is not None and not indexed["equals_streaming_path"]:
        print("bench.py: indexed path differs from the streaming path", file=sys.stderr)
        return 1
    if rank == 0 and parity is False:
        print("bench.py: GPU counts differ from the CPU oracle", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
