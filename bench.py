#!/usr/bin/env python3
"""bench.py — containment queries/sec of the `yacht run` hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: as above -- bench.py then spawns its N ranks itself as child processes, before anything touches the GPU --
     or under a launcher: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the device-side `yacht run` counts over ONE sample sketch: overlap of the
sample with every reference (R1), subset = overlap > 0, subset-exclusive hash counts (R2).  Inputs are
resident in HBM before the timed region; consecutive steps take DIFFERENT samples (--samples, default
8, rotated), so no step finds the previous one's candidates warm in cache.

Workload (config.workload): BASELINE.json configs[2] -- GTDB-rs214-representatives scale: ONE synthetic
database of 85 205 reference sketches per GPU (k=31, scaled=1000, sizes LogNormal(ln 3300, 0.6) in
[300, 15000], ~10 % of the genomes in clusters of 2-8 sharing 10-95 %) against ~1 M-hash samples; the
metric "ref-sketch containment queries/sec" counts one (sample, reference) intersection as one query.

N > 1 (default --shard hash, --block-mode batched, --scaling strong): the HASH SPACE of ONE 85 205-reference database is cut
over the ranks (dist.HashRangeRefDB: every GPU holds one hash range of all references and looks up the sample's hashes in it;
tables AND lookups divide) and the samples go in blocks of 64 through dist.BatchedRangeRunner: first halves, ONE all-gather
of the ranks' non-zero subset words, second halves, ONE sum-reduce of compact count rows to rank 0 -- 0.54 MB per rank and
block, three blocks in flight.  The counts are the exact global ones; rank 0 checks them against the CPU oracle on the WHOLE
database.  --shard refs: contiguous reference shards + ghosts (dist.ShardedRefDB, the capacity mode); --block-mode steps: one
sample per pair of half-steps; --scaling weak: 85 205 references per GPU.
An N > 1 line proves itself and carries its own like-for-like baseline:
  form, value_1gpu_same_form   what a step of `value` is (batched blocks), and rank 0 ALONE running the same blocks of the same
                               samples on the whole database through the same runner, measured after the timed region
  scaling_efficiency           value / (N x value_1gpu_same_form)
  rccl_world_size, distributed what the process group reports after init, every rank's device identity (all-gathered), and a
                               one-word all-reduce only N distinct ranks can get right (the line is refused if it is wrong)
At N = 1 `value` is ONE sample per launch (as the reference runs samples: run_YACHT.py:150); `value_batched` is the same
runner on the whole database -- what N > 1 values are to be divided by.

Rank 0 prints ONE JSON line (contract in the task statement).  Beside the contract's keys:
  device_resident   median / p10 / p90 of per-step HIP-event intervals (a separate pass)
  host_inclusive    SURVEY.md 8d's metric: pinned sample -> H2D -> kernels -> counts D2H, pipelined
                    (yh_run_submit / yh_run_wait), + the latency of one synchronous yh_run call
  real_shape        the hit shape of real runs (~29 % of the references overlap an 83 k-hash sample)
  roofline          the dominant kernel, bytes_basis says which byte count `achieved` uses
  cpu_baseline      the oracle's C++ restatement on the host cores (also the full-size parity check)
"""
from __future__ import annotations

import argparse
import math
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
REFS_PER_GPU = 85_205


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--extras", default=None, help="where the WHOLE result goes (the stdout line carries the contract keys only and names "
                                                   "this file as `extras`; default gpurun_out/bench_extras.json in the repository)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="gtdb_rs214_scale", choices=["gtdb_rs214_scale", "config2_1000refs"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N>1: strong (default) = BASELINE configs[2], ONE 85 205-reference database over the N GPUs; "
                         "weak = 85 205 references per GPU (the database grows with N)")
    ap.add_argument("--shard", default="hash", choices=["hash", "refs"],
                    help="N>1: hash (default) = every GPU holds one HASH RANGE of all references and looks up the sample's hashes "
                         "in it (dist.HashRangeRefDB: table AND lookups divide by N; counts are summed); refs = every GPU holds a "
                         "range of the REFERENCES + ghosts and looks up the whole sample (dist.ShardedRefDB: the capacity mode)")
    ap.add_argument("--no-train", action="store_true", help="N=1: skip the `yacht train` block (configs[3], bench_train.py as a child process)")
    ap.add_argument("--no-sketch", action="store_true", help="N=1: skip the sketcher block (bench_sketch.py as a child process)")
    ap.add_argument("--block-mode", default="batched", choices=["batched", "steps"],
                    help="N>1 with --shard hash: batched (default) = the samples of a block go through the batched kernels in ONE "
                         "pass per rank (a rank's share of one sample is too few lookups to fill a launch) around one exchange of "
                         "their subset words; steps = every sample is its own pair of half-steps, eight per bit exchange")
    ap.add_argument("--batch-block", type=int, default=256, help="--block-mode batched: samples per block (<= 256: four word planes of 64 samples; 64 until round 5)")
    ap.add_argument("--no-scaling-model", action="store_true",
                    help="N=1: skip the measurement of one rank's share of a G-way hash-range step (G = 2, 4, 8)")
    ap.add_argument("--refs", type=int, default=0, help="override references per GPU (testing only)")
    ap.add_argument("--sample-hashes", type=int, default=1_000_000)
    ap.add_argument("--samples", type=int, default=8, help="distinct samples rotated through the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and its parity check)")
    ap.add_argument("--parity-samples", type=int, default=2, help="how many of the samples the oracle re-computes")
    ap.add_argument("--seed", type=int, default=1002)
    ap.add_argument("--present", type=int, default=200, help="genomes present in a sample (diagnostic)")
    ap.add_argument("--no-indexed", action="store_true", help="skip the extra measurement of the sample-driven path")
    ap.add_argument("--no-host-inclusive", action="store_true")
    ap.add_argument("--plain-steps", action="store_true",
                    help="N=1: yh_run_device (three launches per sample: lookup, reduce, exclusive pass) instead of "
                         "yh_run_device_pipelined (ONE launch per sample: its lookup beside the reducer of the previous sample and "
                         "the exclusive pass of the one before) in the timed loop")
    ap.add_argument("--no-real-shape", action="store_true")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--batch-samples", type=int, default=64, help="samples per yh_run_batch_device call of the `batched` leg (<= 256)")
    ap.add_argument("--host-depth", type=int, default=4, help="host-inclusive leg: calls in flight (1..4)")
    ap.add_argument("--percentile-steps", type=int, default=200)
    ap.add_argument("--dense-reduce", action="store_true",
                    help="N>1, --block-mode batched: sum the dense [3, B, N] shares of a block (round 3's result path: 65 MB per rank "
                         "and block at rs214 scale) instead of its compact rows")
    ap.add_argument("--dense-words", action="store_true",
                    help="N>1, --block-mode batched: all-gather the dense subset-word rows of a block (8 N bytes per rank: round 4's "
                         "exchange) instead of the ranks' non-zero (word, reference) entries")
    ap.add_argument("--no-same-form", action="store_true",
                    help="N>1: skip rank 0's single-GPU pass of the same form on the same samples (value_1gpu_same_form, scaling_efficiency)")
    ap.add_argument("--min-timed-steps", type=int, default=2000,
                    help="the timed loop runs max(--steps, this) steps, and more until it lasts --min-timed-ms (SURVEY.md 8d: >= 1000 "
                         "iterations on launch-bound configurations; 20 steps are 0.75 ms of timed region)")
    ap.add_argument("--min-timed-ms", type=float, default=50.0)
    ap.add_argument("--pack-threads", default="2,4,8,16,24,32", help="host-inclusive leg with yh_sample_pack INSIDE the step: packing threads to try")
    ap.add_argument("--sync-gather", action="store_true", help="N>1: blocking gather of the count rows inside every step")
    ap.add_argument("--count-gather", choices=("root", "all"), default="root",
                    help="N>1 over RCCL: the count rows of a block go to rank 0 (where results are consumed; default) or to every rank")
    ap.add_argument("--gather-every", type=int, default=8, help="N>1: samples per all-gather of the count rows")
    ap.add_argument("--no-pipeline", action="store_true", help="N>1: finish every sample before the next one's lookup is queued")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to test the N>1 logic)")
    ap.add_argument("--share-gpu", action="store_true", help="testing: all ranks use cuda:0 (1-GPU box, gloo backend)")
    ap.add_argument("--force-dist", action="store_true", help="testing: take the N > 1 code path (process group, ShardedRefDB, gathers) with one rank")
    return ap.parse_args()


def pct(xs, q):
    return float(np.percentile(np.asarray(xs, dtype=np.float64), q)) if len(xs) else 0.0


def stats_ms(xs):
    return {"median_ms": round(pct(xs, 50), 4), "p10_ms": round(pct(xs, 10), 4), "p90_ms": round(pct(xs, 90), 4),
            "mean_ms": round(float(np.mean(xs)) if len(xs) else 0.0, 4), "n": len(xs)}


def source_tag() -> str:
    """What the PMC traffic figure is keyed on: the kernels' source."""
    h = hashlib.sha256()
    for f in ("yh_query.hip", "yh_common.h"):
        with open(os.path.join(ROOT, "yacht_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args) -> int:
    """--gpus N > 1 started the way --gpus 1 is started (no launcher, WORLD_SIZE unset): spawn the N ranks as CHILD
    processes through torch.distributed.run and relay rank 0's JSON line and the exit code.  Nothing in this process
    has touched the GPU (torch is not even imported yet), and nothing is exec'ed: the children are ordinary
    subprocesses."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode != 0 or lines else 1


def cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None: the GPU boxes of this pool show 256 hardware
    threads and grant 16 CPUs of them (`1600000 100000`) -- threads beyond the quota only take turns (profiles/r06/ingest_trace.txt)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p_ = f.read().split()[:2]
        if q != "max" and float(p_) > 0:
            return float(q) / float(p_)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p_ = float(f.read())
        if q > 0 and p_ > 0:
            return q / p_
    except (OSError, ValueError):
        pass
    return None


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def main() -> int:
    args = parse_args()
    args.pipelined_tail = not args.plain_steps
    wall = {}  # this process's wall-clock by section (seconds): a slow box or a driver stall shows up here
    t_wall = [time.perf_counter()]

    def stamp(name: str) -> None:
        now = time.perf_counter()
        wall[name] = round(wall.get(name, 0.0) + now - t_wall[0], 2)
        t_wall[0] = now

    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"  # (the same thing at N = 1)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_dist:
        return launch_ranks(args)
    # stdout carries the ONE JSON line and nothing else: libraries that print banners at start-up
    # (RCCL prints its version block to stdout when a communicator is created) are sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("YH_BENCH_TRACE_HANG"):  # diagnostics: every thread's stack to stderr after that many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["YH_BENCH_TRACE_HANG"]), repeat=False, file=sys.stderr, exit=False)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks on this pool
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (a launcher started a different number of ranks)",
                  file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)", file=sys.stderr)
        return 2
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist  # the N > 1 code path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            if args.backend == "gloo":
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (no host-name look-ups: they stall on some boxes)
            dist.init_process_group(args.backend)

    from yacht_amd import build, synth
    from yacht_amd import dist as ydist
    from yacht_amd import _lib as ylib
    from yacht_amd.engine import PinnedArray, RefDB, YH_DB_DEFAULT, YH_DB_NO_DIRECTORY

    if not os.path.exists(build.LIB_PATH):
        build.build_lib()

    stamp("imports_and_init")
    # ---- synthetic workload, generated in HBM: this rank's shard of ONE global database -----------------
    if args.workload == "gtdb_rs214_scale":
        per_gpu = args.refs or REFS_PER_GPU
        gen = dict(cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
        n_present = args.present
        wl_name = "GTDB-rs214-scale synthetic"
    else:
        per_gpu = args.refs or 1000
        gen = dict(cluster_frac=0.0, median=5000.0, sigma=0.35, lo=500, hi=20000)
        n_present = 50
        wl_name = "configs[1] synthetic (~5000-hash refs)"
    n_total = per_gpu * world if args.scaling == "weak" else per_gpu
    plan = synth.global_db_plan(args.seed, n_total, **gen)
    by_hash = multi and args.shard == "hash"
    shards = ydist.shard_plan(plan["offsets"].astype(np.uint64), world)
    max_hash_db = synth.max_hash_for_scaled(1000)
    bounds = ydist.hash_range_bounds(max_hash_db, world)
    if by_hash:
        # this rank's HASH RANGE of every reference, generated in chunks of references (any rank generates any reference)
        r_beg, r_end = 0, n_total
        n_local = n_total
        vs, sizes_l = [], []
        for c0 in range(0, n_total, 16384):
            v_c, o_c = synth.global_db_refs_device(plan, np.arange(c0, min(n_total, c0 + 16384)), device=str(dev))
            v_c, o_c = ydist.slice_to_hash_range(v_c, o_c, bounds[rank], bounds[rank + 1])
            vs.append(v_c)
            sizes_l.append(o_c[1:] - o_c[:-1])
        values = torch.cat(vs).contiguous()
        offsets = torch.zeros(n_total + 1, dtype=torch.int64, device=dev)
        offsets[1:] = torch.cumsum(torch.cat(sizes_l), 0)
        del vs, sizes_l
    else:
        r_beg, r_end = shards[rank]
        n_local = r_end - r_beg
        values, offsets = synth.global_db_refs_device(plan, np.arange(r_beg, r_end), device=str(dev))
    H = int(values.numel())
    K = max(args.samples, 1)
    hash_batched = by_hash and args.block_mode == "batched"
    BB = max(1, min(int(args.batch_block), 256))
    if hash_batched:
        K = max(K, BB)  # a block holds DISTINCT samples: a sample repeated inside one pass would find its own buckets cached
    samples = [synth.global_db_sample_device(plan, args.seed + 1000 + i, n_sample=args.sample_hashes, n_present=n_present,
                                             device=str(dev)) for i in range(K)]
    if multi:  # the samples are replicated: every rank queries rank 0's (deterministic, but make it certain)
        def bcast(t):
            c = ydist._stage(t, None)
            dist.broadcast(c, 0)
            return c.to(dev)

        for i in range(K):
            n_i = int(bcast(torch.tensor([samples[i].numel()], device=dev, dtype=torch.int64)).item())
            samples[i] = bcast(samples[i] if rank == 0 else torch.empty(n_i, device=dev, dtype=torch.int64))
    n_sample = int(np.mean([int(s.numel()) for s in samples]))
    workload = (f"{wl_name}: one database of {n_total} refs, {per_gpu if args.scaling == 'weak' else n_total // world} refs/GPU, "
                f"k=31 scaled=1000, {K} rotating ~{args.sample_hashes}-hash samples")
    torch.cuda.synchronize()

    # Everything of the timed region runs on ONE explicit stream: the library's kernels are queued on it
    # and RCCL orders its collectives after whatever is on torch's current stream.
    stream = torch.cuda.Stream(device=dev)
    sdb = None
    db_build_driver = None
    with torch.cuda.stream(stream):
        if by_hash:
            sdb = ydist.HashRangeRefDB(values, offsets, bounds, ydist.HipRangeBackend(local_rank),
                                       block=max(1, min(args.gather_every, 8)))
            db = sdb.local.handle
            row_stride = n_total
        elif multi:
            sdb = ydist.ShardedRefDB(values, offsets, ydist.HipLocalBackend(local_rank), block=max(1, min(args.gather_every, 8)))
            db = sdb.local.handle
            n_rows = torch.tensor([sdb.n_rows], device=dev, dtype=torch.int64)
            n_rows_c = ydist._stage(n_rows, None)
            dist.all_reduce(n_rows_c, op=dist.ReduceOp.MAX)
            row_stride = int(n_rows_c.item())
        else:
            mem0 = ylib.alloc_stats()
            db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_local, device=local_rank,
                                   flags=YH_DB_NO_DIRECTORY if args.no_indexed else YH_DB_DEFAULT)
            mem1 = ylib.alloc_stats()
            db_build_driver = {"allocations": mem1["driver_allocs"] - mem0["driver_allocs"],
                               "ms_inside_hipMalloc": round(mem1["ms_in_driver"] - mem0["ms_in_driver"], 1)}
            row_stride = n_local
    torch.cuda.synchronize()
    db.set_stream(stream.cuda_stream)
    assert stream.cuda_stream != 0
    info = db.info()
    # the samples stay resident: each one's span in this rank's hash range is computed once (HashRangeRefDB remembers nothing)
    spans = [sdb.slice_of(s_) for s_ in samples] if by_hash else None

    # Count rows: N = 1 alternates two buffers.  N > 1: the samples go in BLOCKS of GB = --gather-every (default 8, at
    # most 8): every sample's rank-local half (lookup + reduce) runs in its own step context of the library, the subset
    # bits of the whole block leave in ONE all-gather, then the second halves follow and the block's count rows
    # [GB, 3, row_stride] leave in ONE all-gather -- the north star's "final gather of the per-reference counts".
    # A torch collective costs the host ~30 us per call and the device two cross-queue hand-overs: per sample that
    # made the N > 1 loop host-bound (0.06-0.09 ms per step issued); per block it is an eighth of that.  Two blocks
    # alternate: while the bits of block b travel, the local halves of block b+1 are queued; a block's buffers are
    # refilled only after the collectives reading them have completed (work.wait() = a stream-level wait).
    NBUF = 2
    GB = max(1, min(args.gather_every, 8)) if multi else 1  # (two blocks in flight use 2 * GB of the library's 16 step contexts)
    counts_blk = [torch.zeros((GB, 3, row_stride), device=dev, dtype=torch.int32) for _ in range(NBUF)]
    gathered_blk = [torch.zeros((world, GB, 3, row_stride), device=dev, dtype=torch.int32) if multi else None
                    for _ in range(NBUF)]
    pending = [None] * NBUF
    staged_gather = multi and args.backend != "nccl"
    if by_hash:  # the block's rows are SUMMED over the ranks (to rank 0), not gathered
        gathered_blk = [None] * NBUF
    # all-gathering the rows hands every rank world x 8 MB per block it never reads; results are consumed on rank 0
    to_root = multi and not by_hash and not staged_gather and args.count_gather == "root"
    root_lists = [([gathered_blk[b][r] for r in range(world)] if rank == 0 else None) for b in range(NBUF)] if to_root else None
    pipelined = multi and not staged_gather and not args.sync_gather and not args.no_pipeline
    state = {"i": 0, "open": None}

    def rows_of(i):  # where sample i's three count rows go
        return counts_blk[(i // GB) % NBUF][i % GB]

    def gather_block(i):  # sample i was the last of its block: the block leaves
        blk = (i // GB) % NBUF
        if by_hash:  # one reduce of [GB, 3, N] per block; rank 0 holds the sums (its buffer is overwritten by them)
            if staged_gather:
                counts_blk[blk].copy_(sdb.reduce(counts_blk[blk], dst=0))
            else:
                w = dist.reduce(counts_blk[blk], dst=0, op=dist.ReduceOp.SUM, async_op=not args.sync_gather)
                pending[blk] = None if args.sync_gather else w
            return
        if staged_gather:
            ydist.all_gather_into(gathered_blk[blk].view(-1), counts_blk[blk].view(-1))
        elif to_root:  # rank 0 receives [world, GB, 3, row_stride]; the others only send their 8 MB
            w = dist.gather(counts_blk[blk], gather_list=root_lists[blk], dst=0, async_op=not args.sync_gather)
            pending[blk] = None if args.sync_gather else w
        elif args.sync_gather:
            dist.all_gather_into_tensor(gathered_blk[blk], counts_blk[blk])
        else:
            pending[blk] = dist.all_gather_into_tensor(gathered_blk[blk], counts_blk[blk], async_op=True)

    def finish_block(blk, i0, n_in):  # second halves of the n_in samples of block `blk` (first sample i0); then its rows leave
        for g in range(n_in):
            sdb.end(rows_of(i0 + g), blk, g)
        gather_block(i0)

    def close_block(blk, i0, n_in):  # all local halves of the block are queued: its subset bits leave in ONE all-gather
        sdb.exchange(blk)
        if pipelined:
            # ... and while they travel, the second halves of the PREVIOUS block (whose bits have arrived meanwhile)
            if state["open"] is not None:
                finish_block(*state["open"])
            state["open"] = (blk, i0, n_in)
        else:
            finish_block(blk, i0, n_in)

    # --block-mode batched: block j = samples [j * BB, (j + 1) * BB) mod K in one pass of the batched kernels.  Three blocks
    # rotate through three batch slots of the library: while the subset words of block j travel (ONE all-gather of N * 8
    # bytes per rank), the second half of block j - 1 runs, and its result leaves as COMPACT ROWS (dist.BatchRowsReducer:
    # one sum-reduce of cap * 12 bytes to rank 0 -- north_star's "final gather of the per-reference counts"; round 3 summed
    # the dense [3, BB, N] shares: 65 MB per rank and block); block j - 3's row count is read back before its slot is reused.
    NB3 = 3
    rowsx = None
    runner = None
    if hash_batched:
        packed_blocks = {}

        def on_block(tag, n_in, rows, dense):
            state["last_result"] = (tag, n_in, rows, dense)

        runner = ydist.BatchedRangeRunner(sdb, batch=BB, dst=0, nbuf=NB3, dense_rows=args.dense_reduce,
                                          compact_words=not args.dense_words, on_result=on_block,
                                          async_collectives=not staged_gather and not args.sync_gather)
        rowsx = runner.red

        def run_block(j, n_in):
            key = (j * BB) % K
            if (key, n_in) not in packed_blocks:  # (resident samples: their slices are concatenated once)
                packed_blocks[(key, n_in)] = sdb.pack_batch([samples[(key + t) % K] for t in range(n_in)],
                                                             spans=[spans[(key + t) % K] for t in range(n_in)])
            runner.submit(packed_blocks[(key, n_in)], n_in, tag=j)

    def step():
        i = state["i"]
        state["i"] += 1
        if hash_batched:
            if i % BB == BB - 1:
                run_block(i // BB, BB)
            return
        s = samples[i % K]
        c = rows_of(i)
        if sdb is None:
            # (two count buffers alternate; with --no-pipelined-tail every step's tail is waited for by the next lookup)
            (db.run_device if not args.pipelined_tail else db.run_device_pipelined)(
                s.data_ptr(), s.numel(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())
            return
        blk, g = (i // GB) % NBUF, i % GB
        if g == 0 and pending[blk] is not None:  # first sample of a block: the block's previous gather must be done
            pending[blk].wait()
            pending[blk] = None
        sdb.begin(s, c, blk, g, **({"span": spans[i % K]} if spans is not None else {}))  # lookup + reduce of this sample in its own step context
        if g == GB - 1:
            close_block(blk, i - g, GB)

    def drain():
        if sdb is None:
            db.run_device_join()  # the last tails: the stream's end is the end of every step
        i = state["i"]
        if hash_batched:
            if i % BB != 0:  # a partly filled block at the end of a loop
                run_block(i // BB, i % BB)
                state["i"] += BB - i % BB
            runner.drain()  # (the previous block's second half, then every block's rows, oldest first: the same order on every rank)
            return
        if multi and i % GB != 0:  # a partly filled block at the end of a loop leaves too
            close_block((i // GB) % NBUF, i - i % GB, i % GB)
            state["i"] += GB - i % GB  # (the next loop starts a fresh block)
        if state["open"] is not None:
            finish_block(*state["open"])
            state["open"] = None
        for b in range(NBUF):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    torch.cuda.set_stream(stream)  # everything below is queued on this stream (torch ops and collectives included)
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    db.timing()  # drop the warm-up launches from the kernel-duration ring
    # How many steps the timed region holds: --steps as the driver passes it is a handful (20 steps = 0.75 ms of timed region:
    # +-7 % from run to run), so the loop runs max(--steps, --min-timed-steps) steps -- and, should that still be shorter
    # than --min-timed-ms, as many as a short calibration pass says it takes (the same number on every rank).
    n_timed = max(args.steps, args.min_timed_steps)
    n_cal = max(min(n_timed, 256), BB if hash_batched else 1)
    t0 = time.perf_counter()
    for _ in range(n_cal):
        step()
    drain()
    fence()
    cal = torch.tensor([(time.perf_counter() - t0) / n_cal], device=dev, dtype=torch.float64)
    if multi:
        cal = ydist._stage(cal, None)
        dist.all_reduce(cal, op=dist.ReduceOp.MAX)
    n_timed = max(n_timed, int(math.ceil(args.min_timed_ms * 1e-3 / max(float(cal.item()), 1e-9))))
    if hash_batched:
        n_timed = ((n_timed + BB - 1) // BB) * BB  # whole blocks
    db.timing()
    # (HIP events on the handle's stream around the launches of the timed region -- the roofline's launch duration when a
    # step is ONE launch: see `duration_basis` below)
    ev_region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev_region[0].record(stream)
    t0 = time.perf_counter()
    for _ in range(n_timed):
        step()
    t_issued = time.perf_counter() - t0  # host time to QUEUE the steps (no waiting): host-bound when it equals `elapsed`
    ev_region[1].record(stream)
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = ydist._stage(torch.tensor([elapsed], device=dev, dtype=torch.float64), None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / n_timed
    value = n_total / (elapsed / n_timed)

    stamp("database_and_timed_region")
    # ---- per-step percentiles: a separate pass with one HIP event between steps -------------------------
    n_pct = max(args.percentile_steps, args.steps)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_pct + 1)]
    evs[0].record(stream)
    for k in range(n_pct):
        step()
        evs[k + 1].record(stream)
    drain()
    fence()
    step_ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(n_pct)]
    device_resident = dict(stats_ms(step_ms), how="HIP-event interval per step, separate pass of rotating samples; "
                                                   "`value`/`ms_per_step` come from the un-instrumented timed loop")
    timing = db.timing()  # kernel-duration ring: every 32nd launch of the timed region + percentile pass

    # last step's counts of every sample (for parity): run each sample once more into its own buffer
    results = []
    for i in range(K):
        c = torch.zeros((3, row_stride), device=dev, dtype=torch.int32)
        with torch.cuda.stream(stream):
            if sdb is None:
                db.run_device(samples[i].data_ptr(), samples[i].numel(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())
                results.append(c)
            else:
                sdb.run(samples[i], c)
                results.append(sdb.gather(c))
    torch.cuda.synchronize()
    blocks_ok = None
    if hash_batched and rank == 0:  # the timed loop's own outputs: the totals of the block finished last are on rank 0
        jl, n_last, rows_l, dense_l = state["last_result"]
        got = dense_l if rows_l is None else ydist.BatchRowsReducer.rows_to_dense(rows_l, n_last, n_total)
        blocks_ok = all(bool(torch.equal(got[:, t, :], results[((jl * BB) % K + t) % K])) for t in range(n_last))
    pipelined_ok = None
    if sdb is None and args.pipelined_tail:  # the timed loop's own (pipelined) outputs: its last two steps are still in the buffers
        last = state["i"] - 1
        pipelined_ok = all(bool(torch.equal(rows_of(last - d), results[(last - d) % K])) for d in range(min(2, NBUF)))
    if multi and not by_hash:  # the steps' own gathers must carry this rank's rows
        for b in range(NBUF):
            if not to_root or rank == 0:
                assert bool(torch.equal(gathered_blk[b][rank], counts_blk[b])), "gather ran ahead of the kernels"

    stamp("percentiles")
    # ---- N = 1 extras ------------------------------------------------------------------------------------
    indexed = host_inclusive = real_shape = None
    paths = {}
    default_choice = db.lookup_choice(n_sample)  # (the sharded step asks the same question inside yh_run_local_device)
    if not multi:
        timing_default = timing
        cpath2 = [torch.zeros((3, n_local), device=dev, dtype=torch.int32) for _ in range(2)]
        cpath = cpath2[0]

        def step_path(i, plain=False):
            s = samples[i % K]
            c = cpath if plain else cpath2[i % 2]
            with torch.cuda.stream(stream):
                db.run_device(s.data_ptr(), s.numel(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())  # (plain steps: the kernels apart)

        for name, mode in (("stream", ylib.YH_LOOKUP_STREAM), ("indexed", ylib.YH_LOOKUP_INDEXED)):
            if mode == ylib.YH_LOOKUP_INDEXED and args.no_indexed:
                continue
            db.set_lookup(mode)
            for i in range(args.warmup):
                step_path(i)
            db.run_device_join()
            fence()
            db.timing()
            t0 = time.perf_counter()
            for i in range(max(args.steps, 64)):
                step_path(i)
            db.run_device_join()
            fence()
            el = (time.perf_counter() - t0) / max(args.steps, 64)
            tm = db.timing()
            same = True
            for i in range(K):
                step_path(i, plain=True)
                torch.cuda.synchronize()
                same = same and bool(torch.equal(cpath, results[i]))
            paths[name] = {"ms_per_step": round(1e3 * el, 4), "value": round(n_local / el, 1), "unit": "queries/s",
                           "lookup_kernel_ms_avg": round(float(tm["ms_overlap_kernel"]), 4),
                           "exclusive_kernels_ms_avg": round(float(tm["ms_exclusive_kernels"]), 4),
                           "equals_default_path": same}
        db.set_lookup(ylib.YH_LOOKUP_AUTO)
        indexed = paths.get("indexed")

    if not multi and not args.no_host_inclusive:
        # SURVEY.md 8d's metric: wall time of the steady-state call INCLUDING sample H2D and counts D2H.
        # Page-locked host buffers, DEPTH calls in flight.  Two wire formats:
        #   packed_rows  the sample travels packed (yh_sample_pack: ~4.7 B per hash, made ONCE where the sketch is parsed;
        #                its cost per sample is reported, it is not inside the step) and is expanded + order-checked on the
        #                device; the result comes back as compact rows (references with overlap > 0 only)
        #   raw_dense    round 2's form: 8 B per hash up, three dense count rows down
        from yacht_amd.engine import ROW_DTYPE, pack_sample

        DEPTH = max(1, min(args.host_depth, 4))
        h_samples, h_packed, pack_ms = [], [], []
        for s_ in samples:
            pa = PinnedArray(int(s_.numel()), np.uint64)
            pa.array[:] = s_.cpu().numpy().view(np.uint64)
            h_samples.append(pa)
            pp = PinnedArray(int(ylib.load().yh_sample_pack_bound(int(s_.numel()))), np.uint8)
            t1 = time.perf_counter()
            used = pack_sample(pa.array, out=pp.array)
            pack_ms.append((time.perf_counter() - t1) * 1e3)
            h_packed.append((pp, used))
        h_blk = [PinnedArray(3 * n_local, np.uint32) for _ in range(DEPTH)]  # one contiguous [3][N] block per slot: one D2H copy
        h_rows = [PinnedArray(n_local, ROW_DTYPE) for _ in range(DEPTH)]
        h_out = [[b.array[k * n_local:(k + 1) * n_local] for k in range(3)] for b in h_blk]

        def submit_raw(slot, i):
            o = h_out[slot]
            db.run_submit(slot, h_samples[i % K].array, o[0], o[1], o[2])

        def submit_packed(slot, i):
            db.run_submit_packed(slot, h_packed[i % K][1], h_rows[slot].array)

        n_rows_seen = {}

        def wait_rows(slot):
            n_rows_seen[slot] = db.run_wait_rows(slot)

        def host_leg(submit, wait, check):
            done_t = []
            host_t = {"submit": 0.0, "wait": 0.0}

            def loop(n_steps, record):
                for i in range(n_steps + DEPTH):
                    slot = i % DEPTH
                    if i >= DEPTH:
                        ta = time.perf_counter()
                        wait(slot)
                        tb = time.perf_counter()
                        if record:
                            done_t.append(tb)
                            host_t["wait"] += tb - ta
                    if i < n_steps:
                        ta = time.perf_counter()
                        submit(slot, i)
                        if record:
                            host_t["submit"] += time.perf_counter() - ta

            loop(max(args.warmup, 4 * K), False)  # (the first copy out of every pinned buffer is slow: ~15 ms)
            n_host = max(args.steps, args.percentile_steps)
            t0 = time.perf_counter()
            loop(n_host, True)
            el = time.perf_counter() - t0
            gaps = (np.diff(np.asarray([t0] + done_t)) * 1e3).tolist()[DEPTH:]
            ok = all(check(i % DEPTH, i % K) for i in range(n_host - DEPTH, n_host))  # the last DEPTH results are still in the buffers
            return dict(stats_ms(gaps), value=round(n_local / (el / n_host), 1), unit="queries/s",
                        ms_per_step=round(1e3 * el / n_host, 4), equals_device_resident=ok,
                        host_ms_in_submit=round(1e3 * host_t["submit"] / n_host, 4),
                        host_ms_in_wait=round(1e3 * host_t["wait"] / n_host, 4))

        def check_dense(slot, k):
            want = results[k].cpu().numpy().view(np.uint32)
            return all(np.array_equal(h_out[slot][r], want[r]) for r in range(3))

        def check_rows(slot, k):
            want = results[k].cpu().numpy().view(np.uint32)
            ref = np.flatnonzero(want[0])
            got = h_rows[slot].array[: n_rows_seen.get(slot, 0)]
            return bool(got.size == ref.size and np.array_equal(got["ref"], ref) and np.array_equal(got["overlap"], want[0][ref])
                        and np.array_equal(got["n_excl"], want[1][ref]) and np.array_equal(got["n_match"], want[2][ref]))

        leg_packed = host_leg(submit_packed, wait_rows, check_rows)
        leg_raw = host_leg(submit_raw, db.run_wait, check_dense)
        # the packed form with yh_sample_pack INSIDE the step: T host threads pack the samples to come into a ring of
        # page-locked buffers (the C call releases the GIL) while the main thread submits and waits as before -- which T
        # keeps the step where the pre-packed form has it?  (One core packs a 1e6-hash sample in ~0.7 ms.)
        from concurrent.futures import ThreadPoolExecutor

        pack_legs = {}
        bound = int(ylib.load().yh_sample_pack_bound(int(max(int(s_.numel()) for s_ in samples))))
        for T in [int(x) for x in str(args.pack_threads).split(",") if x.strip()]:
            LOOK = 2 * T + 2
            R = DEPTH + LOOK + 1
            ring = [PinnedArray(bound, np.uint8) for _ in range(R)]
            used = [None] * R
            n_host = max(args.steps, args.percentile_steps)
            n_all = max(args.warmup, 2 * K) + n_host
            with ThreadPoolExecutor(max_workers=T) as pool:
                def job(j):
                    used[j % R] = pack_sample(h_samples[j % K].array, out=ring[j % R].array, threads=1)  # (one core per sample)
                futs = {j: pool.submit(job, j) for j in range(min(LOOK, n_all))}
                t0 = None
                for i in range(n_all + DEPTH):
                    slot = i % DEPTH
                    if i == n_all - n_host:
                        t0 = time.perf_counter()
                    if i >= DEPTH:
                        wait_rows(slot)
                    if i + LOOK < n_all:  # (its ring buffer belonged to call i + LOOK - R = i - DEPTH - 1: waited for above)
                        futs[i + LOOK] = pool.submit(job, i + LOOK)
                    if i < n_all:
                        futs.pop(i).result()
                        db.run_submit_packed(slot, used[i % R], h_rows[slot].array)
                el = time.perf_counter() - t0
            ok = all(check_rows(i % DEPTH, i % K) for i in range(n_all - DEPTH, n_all))
            pack_legs[str(T)] = {"ms_per_step": round(1e3 * el / n_host, 4), "equals_device_resident": bool(ok)}
            for pa in ring:
                pa.close()
        packed_bytes = int(np.mean([int(u.size) for _, u in h_packed]))
        rows_per_step = int(np.mean([int((r[0] != 0).sum().item()) for r in results]))
        leg_packed.update(h2d_bytes_per_step=packed_bytes, d2h_bytes_per_step=16 * rows_per_step + 4,
                          h2d_GBps=round(packed_bytes / (leg_packed["ms_per_step"] / 1e3) / 1e9, 1),
                          bytes_per_sample_hash=round(packed_bytes / max(n_sample, 1), 3),
                          pack_ms_per_sample_on_host=round(float(np.median(pack_ms)), 3),
                          pack_note="yh_sample_pack runs once per sample where the sketch is parsed (not inside the step); "
                                    "pack_inside_the_step: the same leg with T host threads packing inside the timed loop",
                          pack_inside_the_step=pack_legs,
                          pack_threads_to_keep_the_step=next((int(t_) for t_, v_ in sorted(pack_legs.items(), key=lambda kv: int(kv[0]))
                                                              if v_["ms_per_step"] <= 1.1 * leg_packed["ms_per_step"]), None))
        leg_raw.update(h2d_bytes_per_step=8 * n_sample, d2h_bytes_per_step=12 * n_local,
                       h2d_GBps=round(8 * n_sample / (leg_raw["ms_per_step"] / 1e3) / 1e9, 1))
        # latency of ONE synchronous host-pointer call (what the CLI pays per sample)
        lat = []
        for i in range(30):
            t1 = time.perf_counter()
            db.run_counts(h_samples[i % K].array)
            lat.append((time.perf_counter() - t1) * 1e3)
        host_inclusive = dict(leg_packed, form="packed_rows", pipeline_depth=DEPTH, raw_dense=leg_raw,
                              equals_device_resident=bool(leg_packed["equals_device_resident"] and leg_raw["equals_device_resident"]
                                                          and all(v_["equals_device_resident"] for v_ in pack_legs.values())),
                              sync_call_ms_median=round(pct(lat[5:], 50), 4),
                              how="page-locked host buffers; per step: sample H2D on a copy stream -> expansion / ordering "
                                  "check + kernels -> result D2H written by the step's own kernels (yh_run_submit_packed / "
                                  "yh_run_wait_rows; raw_dense: yh_run_submit / yh_run_wait); percentiles over the intervals "
                                  "between completed steps; sync_call = one blocking yh_run from pageable numpy arrays")
        for pa in h_samples + [pp for pp, _ in h_packed] + h_blk + h_rows:
            pa.close()
        h_out = None

    real_samples = []
    if not multi and not args.no_real_shape and args.workload == "gtdb_rs214_scale":
        # the hit shape of the reference's shipped results: ~29 % of the references overlap
        real_samples = [synth.global_db_sample_device(plan, args.seed + 5000 + i, n_sample=83_000, device=str(dev),
                                                      shape="real") for i in range(4)]
        creal2 = [torch.zeros((3, n_local), device=dev, dtype=torch.int32) for _ in range(2)]
        creal = creal2[0]
        run_step = db.run_device if not args.pipelined_tail else db.run_device_pipelined

        def step_real(i, plain=False):
            s = real_samples[i % len(real_samples)]
            c = creal if plain else creal2[i % 2]
            with torch.cuda.stream(stream):
                (db.run_device if plain else run_step)(s.data_ptr(), s.numel(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())

        def fence_real():
            db.run_device_join()
            fence()

        for i in range(args.warmup):
            step_real(i)
        fence_real()
        db.timing()
        ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(n_pct + 1)]
        n_real = max(args.steps, args.min_timed_steps)  # (a 20-step loop is 0.3 ms: anything the host does is in it)
        t0 = time.perf_counter()
        for i in range(n_real):
            step_real(i)
        fence_real()
        el = time.perf_counter() - t0
        ev2[0].record(stream)
        for k in range(n_pct):
            step_real(k)
            ev2[k + 1].record(stream)
        fence_real()
        tm = db.timing()
        step_real(0, plain=True)
        torch.cuda.synchronize()
        real_counts0 = creal.clone()
        forced = {}
        for name, mode in (("stream", ylib.YH_LOOKUP_STREAM), ("indexed", ylib.YH_LOOKUP_INDEXED)):
            if mode == ylib.YH_LOOKUP_INDEXED and args.no_indexed:
                continue
            db.set_lookup(mode)
            for i in range(args.warmup):
                step_real(i)
            fence_real()
            t0 = time.perf_counter()
            for i in range(max(args.steps, 64)):
                step_real(i)
            fence_real()
            forced[name] = round(1e3 * (time.perf_counter() - t0) / max(args.steps, 64), 4)
            step_real(0, plain=True)
            torch.cuda.synchronize()
            forced[name + "_equals_default"] = bool(torch.equal(creal, real_counts0))
        db.set_lookup(ylib.YH_LOOKUP_AUTO)
        real_shape = dict(stats_ms([ev2[k].elapsed_time(ev2[k + 1]) for k in range(n_pct)]),
                          default_lookup="indexed" if db.lookup_choice(int(real_samples[0].numel())) == ylib.YH_LOOKUP_INDEXED else "stream",
                          forced_ms_per_step=forced,
                          ms_per_step=round(1e3 * el / n_real, 4), value=round(n_local / (el / n_real), 1), steps_timed=n_real,
                          unit="queries/s", sample_hashes=int(real_samples[0].numel()),
                          refs_overlapping=int((real_counts0[0] != 0).sum().item()),
                          lookup_kernel_ms_avg=round(float(tm["ms_overlap_kernel"]), 4),
                          exclusive_kernels_ms_avg=round(float(tm["ms_exclusive_kernels"]), 4))

    stamp("n1_extras")
    # ---- batched run: many samples against the resident database in ONE call -----------
    # Not `value` (a step there is one sample, as the reference runs them): the throughput a caller gets who has
    # the samples of a whole plate in hand.  Distinct samples, so that none finds its buckets cached.
    batched = None
    if not multi and not args.no_batched and args.workload == "gtdb_rs214_scale" and not args.no_indexed:
        Bn = max(1, min(int(args.batch_samples), 256))
        bs = [samples[i] if i < K else synth.global_db_sample_device(plan, args.seed + 7000 + i, n_sample=args.sample_hashes,
                                                                     n_present=n_present, device=str(dev)) for i in range(Bn)]
        cat = torch.cat(bs).contiguous()
        soff = torch.zeros(Bn + 1, dtype=torch.int64, device=dev)
        soff[1:] = torch.cumsum(torch.tensor([int(b.numel()) for b in bs], device=dev, dtype=torch.int64), 0)
        bout = torch.zeros((3, Bn, n_local), device=dev, dtype=torch.int32)
        single = torch.zeros((3, n_local), device=dev, dtype=torch.int32)
        torch.cuda.synchronize()

        def step_batch():
            with torch.cuda.stream(stream):
                db.run_batch_device(cat.data_ptr(), soff.data_ptr(), Bn, int(cat.numel()), bout[0].data_ptr(), bout[1].data_ptr(),
                                    bout[2].data_ptr())

        for _ in range(2):
            step_batch()
        fence()
        reps_b = max(3, min(20, args.steps // Bn + 1))
        t0 = time.perf_counter()
        for _ in range(reps_b):
            step_batch()
        fence()
        el_b = (time.perf_counter() - t0) / reps_b
        same = True
        for i in (0, Bn // 2, Bn - 1):  # the same counts as the single-sample step
            with torch.cuda.stream(stream):
                db.run_device(bs[i].data_ptr(), bs[i].numel(), single[0].data_ptr(), single[1].data_ptr(), single[2].data_ptr())
            fence()
            same = same and bool(torch.equal(single, bout[:, i, :]))
        batched = {"samples_per_call": Bn, "ms_per_call": round(1e3 * el_b, 4), "ms_per_sample": round(1e3 * el_b / Bn, 4),
                   "value": round(n_local * Bn / el_b, 1), "unit": "queries/s", "equals_single_sample_step": same,
                   "how": "yh_run_batch_device: one lookup pass over the hashes of all samples + one exclusive pass with "
                          "64-bit per-sample words; device-resident, distinct samples"}
        del cat, bout, bs

    stamp("batched")
    # ---- what ONE rank of a G-way hash-range run computes per step, measured here (N = 1 only) ------------------
    # The first multi-GPU run has a prediction to be compared with: rank 0's range of the database (1 / G of the hashes
    # of every reference) is built on this GPU and the two halves of its step (yh_run_local_range_device,
    # yh_run_finish_range_device) are timed on its slice of the rotating samples -- everything but the collectives.
    scaling_model = None

    block_samples = []

    def share_of(G, v_full, o_full, n_refs_, group=None, single_steps=True):
        """Rank 0's share of a G-way hash-range run of the database (v_full, o_full), built and timed on THIS GPU: both halves
        of a block of BM distinct samples through dist.BatchedRangeRunner -- the very object the N > 1 loop drives, here with a
        one-rank exchange (its own words): packing and unpacking of the compact subset words and of the compact rows
        included, the collectives themselves not.  G = 1 is the whole database: the single-GPU figure of the SAME FORM."""
        bg = ydist.hash_range_bounds(max_hash_db, G)
        v_g, o_g = (v_full, o_full) if G == 1 else ydist.slice_to_hash_range(v_full, o_full, bg[0], bg[1])
        with torch.cuda.stream(stream):
            hr = ydist.HashRangeRefDB(v_g, o_g, [bg[0], bg[1]], ydist.HipRangeBackend(local_rank), group=group, block=1)
        torch.cuda.synchronize()
        hr.local.handle.set_stream(stream.cuda_stream)
        spans_g = [hr.slice_of(s_) for s_ in samples]  # (resident samples: their spans in this range, once)
        el = None
        if single_steps:
            cg = [hr.new_counts() for _ in range(2)]

            def step_g(i):
                with torch.cuda.stream(stream):
                    hr.begin(samples[i % K], cg[i % 2], 0, 0, span=spans_g[i % K])
                    hr.end(cg[i % 2], 0, 0)

            for i in range(max(args.warmup, K)):
                step_g(i)
            torch.cuda.synchronize()
            n_g = max(args.steps, 100)
            t0 = time.perf_counter()
            for i in range(n_g):
                step_g(i)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / n_g
            del cg
        BM = max(1, min(int(args.batch_block), 256))
        # the throughput form: BM distinct samples per block (the block of --block-mode batched), three blocks in flight
        if len(block_samples) < BM:  # (made once: the same BM distinct samples for every G -- 248 x ~1e6 hashes at BM = 256)
            block_samples.extend(samples[i] if i < K else synth.global_db_sample_device(plan, args.seed + 7000 + i, n_sample=args.sample_hashes,
                                                                                      n_present=n_present, device=str(dev))
                                 for i in range(len(block_samples), BM))
        bsamp = block_samples[:BM]
        with torch.cuda.stream(stream):
            packed_b = hr.pack_batch(bsamp)
        seen = {}

        def on_res(tag, n_in, rows, dense):
            seen["rows"] = int(rows.shape[0]) if rows is not None else -1

        with torch.cuda.stream(stream):
            run_g = ydist.BatchedRangeRunner(hr, batch=BM, dst=0, nbuf=3, on_result=on_res, compact_words=not args.dense_words,
                                             dense_rows=args.dense_reduce)
            for _ in range(3):
                run_g.submit(packed_b, BM)
            run_g.drain()
            torch.cuda.synchronize()
            nb_ = max(30, min(60, args.steps // BM + 1))  # (steady state: six blocks were mostly the fill and the drain of three in flight)
            t0 = time.perf_counter()
            for _ in range(nb_):
                run_g.submit(packed_b, BM)
            run_g.drain()
            torch.cuda.synchronize()
            el_b = (time.perf_counter() - t0) / nb_
        a_, b_ = spans_g[0]
        res = {"rank0_compute_ms_per_step": round(1e3 * el, 4) if el is not None else None, "sample_hashes_in_range": int(b_ - a_),
               "batched_rank0_ms_per_sample": round(1e3 * el_b / BM, 4), "batched_rank0_ms_per_block": round(1e3 * el_b, 4),
               "samples_per_block": BM, "rows_in_block": seen.get("rows"), "blocks_timed": nb_,
               "word_exchange_overflows": run_g.n_words_overflow, "rows_overflows": (run_g.red.n_overflow if run_g.red is not None else None),
               "collective_bytes": run_g.collective_bytes(),
               "ref_hashes_in_range": int(v_g.numel()),
               "lookup_choice": "indexed" if hr.local.handle.lookup_choice(int(b_ - a_)) == ylib.YH_LOOKUP_INDEXED else "stream"}
        hr.close()
        del bsamp, packed_b, run_g, hr
        return res

    if not multi and not args.no_scaling_model and args.workload == "gtdb_rs214_scale" and not args.no_indexed:
        per_g = {}
        for G in (1, 2, 4, 8):
            per_g[str(G)] = share_of(G, values, offsets, n_local, single_steps=G > 1)
        BM = max(1, min(int(args.batch_block), 256))
        # The collectives of a block, from their BYTES (nothing here is measured: this box has one GPU).  Stated constants:
        #   link   64 GB/s per direction and xGMI link (7 links x ~153 GB/s bidirectional per GPU, ~83 % of the wire rate as payload)
        #   lat    30 us per collective (RCCL launch, synchronisation and the first hop of a small message between 8 ranks)
        # On the fully connected xGMI mesh a rank reaches each peer over a link of its own: an all-gather moves the rank's
        # message once per link (`direct`); a ring moves (G - 1) messages over one link (`ring`, the pessimistic bound).
        LINK_GBPS, LAT_MS = 64.0, 0.030
        cbytes = per_g["1"]["collective_bytes"]        # (the same capacities at every G: they depend on N and the block size only)
        words_bytes = cbytes["subset_words_all_gather"]  # all-gather of the block's subset words, compact: 12 B per non-zero word's slot
        words_dense_bytes = cbytes["subset_words_dense_form"]  # round 4's exchange: one uint64 per reference and rank
        rows_bytes = cbytes["result"]                  # dist.BatchRowsReducer's collective: cap x (overlap, n_excl, n_match)
        dense_bytes = cbytes["result_dense_form"]      # round 3's result path: the dense shares
        def t_coll(nbytes, G, ring):
            return LAT_MS + (nbytes * ((G - 1) if ring else 1)) / (LINK_GBPS * 1e9) * 1e3
        one_gpu_same_form = per_g["1"]["batched_rank0_ms_per_sample"]  # the single-GPU figure of the SAME form (blocks of BM through the same runner)
        model = {}
        for g, v in per_g.items():
            G = int(g)
            if G == 1:
                continue
            comp = v["batched_rank0_ms_per_block"]
            c_dir = t_coll(words_bytes, G, False) + t_coll(rows_bytes, G, False)
            c_ring = t_coll(words_bytes, G, True) + t_coll(rows_bytes, G, True)
            c_dense_ring = t_coll(words_dense_bytes, G, True) + t_coll(dense_bytes, G, True)
            model[g] = {
                "compute_ms_per_block": comp,
                "collectives_ms_per_block_direct": round(c_dir, 4), "collectives_ms_per_block_ring": round(c_ring, 4),
                "collectives_ms_per_block_dense_rows_ring": round(c_dense_ring, 4),
                # three blocks in flight: the words of block j travel under the second half of block j - 1, the rows of block
                # j - 1 under the first half of block j + 1 -- a block costs the longer of its compute and its collectives;
                # `serial` = nothing overlaps (the pessimistic bound)
                "ms_per_sample_overlapped": round(max(comp, c_ring) / BM, 4),
                "ms_per_sample_serial": round((comp + c_ring) / BM, 4),
                "ms_per_sample_serial_dense_rows": round((comp + c_dense_ring) / BM, 4),
                # LIKE FOR LIKE: against one GPU running the same batched blocks (what scaling efficiency is made of) ...
                "speedup_vs_1gpu_batched_overlapped": round(one_gpu_same_form / (max(comp, c_ring) / BM), 2),
                "speedup_vs_1gpu_batched_serial": round(one_gpu_same_form / ((comp + c_ring) / BM), 2),
                "efficiency_vs_1gpu_batched_overlapped": round(one_gpu_same_form / (max(comp, c_ring) / BM) / G, 3),
                "efficiency_vs_1gpu_batched_serial": round(one_gpu_same_form / ((comp + c_ring) / BM) / G, 3),
                # ... and against `value` (one GPU, ONE sample per launch: a different form -- what the driver's own division of the
                # per-N values will show, since N = 1 reports single steps and N > 1 batched blocks)
                "speedup_vs_1gpu_single_steps_overlapped": round(ms_per_step / (max(comp, c_ring) / BM), 2),
                "speedup_vs_1gpu_single_steps_serial": round(ms_per_step / ((comp + c_ring) / BM), 2),
                "speedup_vs_1gpu_single_steps_serial_dense_rows": round(ms_per_step / ((comp + c_dense_ring) / BM), 2),
            }
        scaling_model = {
            "per_G": per_g,
            "one_gpu_same_form_ms_per_sample": one_gpu_same_form,
            "collective_bytes_per_block_and_rank": {
                "subset_words_all_gather": words_bytes, "subset_words_dense_form_round4": words_dense_bytes,
                "compact_rows_reduce": rows_bytes, "total": words_bytes + rows_bytes,
                "compact_rows_capacity": cbytes["result_capacity_rows"], "subset_words_capacity": cbytes["subset_words_capacity"],
                "rows_in_a_block_measured": {g: v["rows_in_block"] for g, v in per_g.items()},
                "dense_rows_reduce_round3": dense_bytes,
            },
            "assumed": {"xgmi_link_GBps_per_direction": LINK_GBPS, "collective_latency_ms": LAT_MS,
                        "note": "bytes are exact; the link rate and the per-collective latency are stated constants, not measurements (1-GPU box)"},
            "batched": model,
            # the per-sample half-steps (--block-mode steps): four launches of latency per sample, one exchange per block of 8
            "single_steps_predicted_ms_per_step": {g: round(v["rank0_compute_ms_per_step"] + (t_coll(8 * ((n_local + 255) // 256) * 32, int(g), True)
                                                                                            + t_coll(8 * 3 * n_local * 4, int(g), True)) / 8, 4)
                                                   for g, v in per_g.items() if v["rank0_compute_ms_per_step"] is not None},
            "how": "rank 0's hash range of the whole database built on THIS GPU (G = 1: all of it); blocks of samples_per_block distinct "
                   "samples through dist.BatchedRangeRunner -- both halves, compact words and compact rows packed and unpacked, three blocks "
                   "in flight, no collectives -- timed on its slice; the collectives from their bytes",
        }

    # ---- N > 1: what makes the scaling line like-for-like and self-proving ---------------------------------------------
    #   form                    what a step of `value` is: batched blocks (default) or single steps
    #   value_1gpu_same_form    rank 0 ALONE, after the timed region, on the WHOLE database and the same samples, through the very
    #                           same runner (a one-rank group: no collectives) -- the other ranks wait at the barrier
    #   scaling_efficiency      value / (N x value_1gpu_same_form)
    #   rccl_world_size, ranks  what the process group itself reports after init_process_group, every rank's device identity
    #                           (all-gathered) and a one-word all-reduce whose answer only N distinct ranks can produce
    same_form = None
    dist_proof = None
    if multi:
        ident = {"rank": rank, "local_rank": local_rank, "device_index": int(torch.cuda.current_device()), "pid": os.getpid()}
        try:
            pr_ = torch.cuda.get_device_properties(dev)
            ident["device_name"] = pr_.name
            ident["device_uuid"] = str(getattr(pr_, "uuid", None))
            ident["pci_bus_id"] = getattr(pr_, "pci_bus_id", None)
        except Exception as ex:  # noqa: BLE001
            ident["error"] = repr(ex)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        word = ydist._stage(torch.tensor([rank + 1], device=dev, dtype=torch.int64), None)
        dist.all_reduce(word, op=dist.ReduceOp.SUM)
        uu = [str(x.get("device_uuid")) + "/" + str(x.get("pci_bus_id")) + "/" + str(x.get("device_index")) for x in idents]
        dist_proof = {"backend": dist.get_backend(), "rccl_world_size": dist.get_world_size(), "ranks": idents,
                      "distinct_devices": len(set(uu)) if not args.share_gpu else 1,
                      "all_reduce_of_rank_plus_1": int(word.item()), "all_reduce_expected": world * (world + 1) // 2,
                      "all_reduce_ok": int(word.item()) == world * (world + 1) // 2}
        g1 = dist.new_group([0])  # (every rank takes part in making it; only rank 0 uses it)
        if hash_batched and not args.no_same_form and args.workload == "gtdb_rs214_scale":
            if rank == 0:
                vs_ = []
                for c0 in range(0, n_total, 16384):
                    v_c, _o = synth.global_db_refs_device(plan, np.arange(c0, min(n_total, c0 + 16384)), device=str(dev))
                    vs_.append(v_c)
                v_full = torch.cat(vs_).contiguous()
                o_full = torch.from_numpy(plan["offsets"].astype(np.int64)).to(dev)
                del vs_
                same_form = share_of(1, v_full, o_full, n_total, group=g1, single_steps=False)
                del v_full, o_full
            fence()
    stamp("scaling_model")
    # ---- roofline of the dominant kernel of the DEFAULT step --------------------------------------------
    # Streaming lookup (k_stream_lookup): `achieved` = bytes one launch HAS to move in the layout the kernel
    # reads (yh_db_info.stream_bytes: one delta byte per (hash, reference) pair + an 8-byte header per 1024,
    # plus the 8-byte sample hashes staged once) over the measured launch duration.  SURVEY.md 8d's one-touch
    # formula (8 B per reference hash) is reported beside it (`survey_formula`); it exceeds the peak because
    # the kernel does not read 8 bytes per hash.
    # Sample-driven lookup (k_index_lookup): one 64-byte bucket per sample hash + the 8-byte sample hash:
    # 72 B x |S|, random sectors -- the same HBM peak, a much lower practical ceiling.
    layout = int(info.get("stream_layout", 0))
    Hh = int(info["n_hashes"])
    Nh = int(info["n_refs"])
    survey_bytes = 8 * (Hh + n_sample) + 8 * (Nh + 1) + 4 * Nh

    rl_main = [False]  # (set while the roofline of the timed loop's own step is made: that one may be the fused launch)

    def roofline_stream(k_ms, excl_ms):
        alg_bytes = int(info.get("stream_bytes", 0)) + 8 * n_sample
        achieved = (alg_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
        survey_rate = (survey_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
        return {
            "bound": "hbm",
            "kernel": {1: "k_stream_lookup", 2: "k_tile_lookup_keys", 3: "k_tile_lookup<OverlapHit>"}.get(layout, "?"),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "bytes_basis": "layout: the delta stream the kernel reads (1 B per reference hash + 8 B per 1024) + 8 B per sample hash",
            "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_per_ref_hash": round(int(info.get("stream_bytes", 0)) / max(Hh, 1), 4),
            "kernel_ms_avg": round(k_ms, 4), "exclusive_kernels_ms_avg": round(excl_ms, 4),
            "survey_formula": {"bytes_per_launch": survey_bytes, "GBps": round(survey_rate, 1),
                               "frac": round(survey_rate / HBM_PEAK_GBS, 4),
                               "note": "8 B per reference hash as SURVEY.md 8d counts; the kernel streams "
                                       f"{round(int(info.get('stream_bytes', 0)) / max(Hh, 1), 3)} B per hash"},
        }

    def roofline_indexed(k_ms, excl_ms):
        # N > 1, batched blocks (the default there): ONE launch of k_batch_lookup looks up this rank's slices of all BB samples of a
        # block -- BB x |S| / world hashes --, so the launch's bytes are counted for those (round 6: the line counted one sample's
        # bytes against a block's launch -- frac 0.002)
        batched_launch = bool(multi and hash_batched and rl_main[0])
        launch_hashes = (BB * n_sample) // world if batched_launch else n_sample
        alg_bytes = 72 * launch_hashes
        achieved = (alg_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
        sv_bytes = survey_bytes + 8 * (launch_hashes - n_sample)
        survey_rate = (sv_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
        return {
            # (every sample size takes a form of k_index_lookup_tile: yh_q_overlap_indexed)
            "bound": "hbm", "kernel": "k_batch_lookup" if batched_launch else
                                      "k_step_fused" if (args.pipelined_tail and not multi and rl_main[0]) else "k_index_lookup_tile",
            "hashes_per_launch": launch_hashes,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "bytes_basis": "layout: one 64-byte bucket (random sector) + the 8-byte hash per SAMPLE hash; no reference hash is streamed. "
                           "The presence filter in front of the buckets trades most bucket reads of absent hashes for one filter "
                           "line each, so the measured traffic differs from this figure in both directions",
            "algorithmic_bytes_per_launch": alg_bytes,
            "random_sectors_per_s": round(launch_hashes / (k_ms / 1e3), 1) if k_ms > 0 else 0.0,
            "random_sector_ceiling_per_s": 4.6e10,
            "ceiling_note": "scripts/probes/gather_probe.hip on this GPU: 4.6e10 independent 64-byte reads/s (2.9 TB/s) whatever the access form",
            "kernel_ms_avg": round(k_ms, 4), "exclusive_kernels_ms_avg": round(excl_ms, 4),
            "survey_formula": {"bytes_per_launch": sv_bytes, "GBps": round(survey_rate, 1),
                               "frac": round(survey_rate / HBM_PEAK_GBS, 4),
                               "note": "8 B per reference hash as SURVEY.md 8d counts; this kernel reads none of them"},
        }

    k_ms = float(timing["ms_overlap_kernel"])
    x_ms = float(timing["ms_exclusive_kernels"])
    k_pairs_ms = k_ms
    one_launch_per_step = bool(args.pipelined_tail and not multi and default_choice == ylib.YH_LOOKUP_INDEXED)
    if one_launch_per_step:
        # The step is ONE launch (k_step_fused), so the launches of the timed region are exactly n_timed and the interval between
        # two HIP events on their stream, one before the first and one behind the last, divided by n_timed, is the kernel's
        # average duration INCLUDING the dispatch gaps between consecutive launches -- an upper bound, never flattering -- and
        # without the ~5 us a per-launch event pair adds to every sampled launch (`kernel_ms_avg_event_pairs`; rocprofv3's
        # own figure for the kernel is in profiles/: 33.4 us).
        k_ms = float(ev_region[0].elapsed_time(ev_region[1])) / n_timed
    rl_main[0] = True
    roofline = (roofline_indexed if default_choice == ylib.YH_LOOKUP_INDEXED else roofline_stream)(k_ms, x_ms)
    rl_main[0] = False
    if roofline["kernel"] == "k_step_fused":
        roofline["kernel_note"] = ("one launch per step: the lookup of this sample (the role the bytes are counted for: "
                                   "k_index_lookup_tile's body) + the reducer of the previous sample + the exclusive pass of the one before")
    roofline["duration_basis"] = (f"HIP events on the handle's stream around the {n_timed} launches of the timed region / {n_timed} "
                                  "(dispatch gaps included)" if one_launch_per_step else
                                  "HIP event pairs around every 32nd launch of the dominant kernel in the timed region (the library's ring)")
    roofline["kernel_ms_avg_event_pairs"] = round(k_pairs_ms, 4)
    roofline["default_lookup"] = "indexed" if default_choice == ylib.YH_LOOKUP_INDEXED else "stream"
    # (scalar copies of the nested survey_formula block: parsers that keep only scalars keep these)
    roofline["bytes_survey_formula"] = roofline["survey_formula"]["bytes_per_launch"]
    roofline["GBps_survey_formula"] = roofline["survey_formula"]["GBps"]
    roofline["frac_survey_formula"] = roofline["survey_formula"]["frac"]
    # What a HIP-event pair measures with NOTHING between its two records, on the busy stream of the step loop: the
    # part of `kernel_ms_avg` that is marker processing and dispatch, not kernel (rocprofv3's kernel durations in
    # profiles/ do not contain it).  `achieved` / `frac` stay on the raw interval: never flattered.
    if not multi:
        pairs = []
        for _ in range(64):
            step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            e1.record(stream)
            pairs.append((e0, e1))
        drain()
        fence()
        gaps = sorted(a.elapsed_time(b) for a, b in pairs)
        empty_ms = float(gaps[len(gaps) // 2])
        roofline["event_pair_overhead_ms"] = round(empty_ms, 4)
        roofline["kernel_ms_avg_net_of_event_overhead"] = round(max(k_pairs_ms - empty_ms, 0.0), 4)  # (of the per-launch pairs)
    other = None
    if "stream" in paths and default_choice == ylib.YH_LOOKUP_INDEXED:
        other = roofline_stream(paths["stream"]["lookup_kernel_ms_avg"], paths["stream"]["exclusive_kernels_ms_avg"])
    elif "indexed" in paths and default_choice == ylib.YH_LOOKUP_STREAM:
        other = roofline_indexed(paths["indexed"]["lookup_kernel_ms_avg"], paths["indexed"]["exclusive_kernels_ms_avg"])
    # HBM bytes per launch from the rocprofv3 --pmc passes (profiles/README.md), attached only when they
    # were taken from THIS source of the kernels on THIS workload; otherwise null.
    tag = source_tag()
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not (name.startswith("traffic_") and name.endswith(".json")):
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tr = json.load(f)
            for rl in (roofline, other):
                if rl is not None and rl["traffic"] is None and tr.get("source_tag") == tag and tr.get("n_hashes") == Hh \
                        and tr.get("kernel") == rl["kernel"]:
                    rl["traffic"] = tr.get("hbm_bytes_per_launch")
                    if tr.get("l2_requests_per_launch") and rl.get("kernel_ms_avg"):
                        # isolated reads are bound by the NUMBER of L2 requests (TCC_REQ, same PMC passes): the rate over
                        # this run's kernel interval, next to scripts/probes/gather_probe.hip's ceiling for launches of
                        # 1e6 independent 64-byte reads (3.7e10/s; 4.6e10/s at 1.6e7)
                        rl["l2_requests_per_launch"] = tr["l2_requests_per_launch"]
                        rl["l2_requests_per_s"] = round(tr["l2_requests_per_launch"] / (rl["kernel_ms_avg"] / 1e3), 1)
                    rl["traffic_provenance"] = {"file": "profiles/" + name, "source_tag": tag,
                                                "taken": tr.get("taken"), "commit": tr.get("commit")}
        except Exception:
            pass

    stamp("roofline")
    # ---- CPU baseline + full-size parity (rank 0, the WHOLE database) -----------------------------------
    cpu_baseline = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle  # the checker; never the thing measured as `value`

        if not multi:
            h_values = values.cpu().numpy().view(np.uint64)
            h_offsets = offsets.cpu().numpy().view(np.uint64)
        else:  # regenerate the other shards here, one at a time (the generator is a pure function of the plan)
            parts = []
            for (b, e) in shards:
                v, _ = synth.global_db_refs_device(plan, np.arange(b, e), device=str(dev))
                parts.append(v.cpu().numpy().view(np.uint64))
                del v
            h_values = np.concatenate(parts)
            h_offsets = plan["offsets"].astype(np.uint64)
            del parts
        hw_threads = oracle.hardware_threads()
        quota = cpu_quota()
        # (threads = what the container may actually run at once: the box shows 256 hardware threads and grants 16 CPUs)
        cores = max(1, min(hw_threads, int(quota))) if quota else hw_threads
        t_ov = t_ex = 0.0
        parity = True
        n_par = min(max(args.parity_samples, 1), K)
        for i in range(n_par):
            h_sample = samples[i].cpu().numpy().view(np.uint64)
            t0 = time.perf_counter()
            want_ov = oracle.overlap(h_values, h_offsets, h_sample, threads=cores)
            t_ov += time.perf_counter() - t0
            t0 = time.perf_counter()
            want_e, want_m = oracle.exclusive(h_values, h_offsets, want_ov > 0, h_sample)
            t_ex += time.perf_counter() - t0
            got = results[i].cpu().numpy().view(np.uint32)[:, :n_total]
            parity = parity and bool(np.array_equal(got[0], want_ov)) and bool(np.array_equal(got[1], want_e)) \
                and bool(np.array_equal(got[2], want_m))
        if real_shape is not None:
            h_sample = real_samples[0].cpu().numpy().view(np.uint64)
            want_ov = oracle.overlap(h_values, h_offsets, h_sample, threads=cores)
            want_e, want_m = oracle.exclusive(h_values, h_offsets, want_ov > 0, h_sample)
            got = real_counts0.cpu().numpy().view(np.uint32)
            real_shape["parity_bit_exact"] = bool(np.array_equal(got[0], want_ov) and np.array_equal(got[1], want_e)
                                                  and np.array_equal(got[2], want_m))
            parity = parity and real_shape["parity_bit_exact"]
        if not multi:
            cpu_baseline = {
                "value": round(n_par * n_total / (t_ov + t_ex), 1),
                "unit": "queries/s",
                "cores": cores,
                "cpu_model": cpu_model(),
                "kind": "port",
                "sample": f"{n_par} of the {K} samples against the whole database ({n_total} refs, {Hh} hashes): "
                          f"overlap {t_ov:.2f} s on {cores} threads + exclusive {t_ex:.2f} s on 1 thread"
                          + (f" (cgroup quota {quota:g} CPUs of {hw_threads} hardware threads)" if quota else ""),
                "hardware_threads": hw_threads, "cpu_quota": quota,
            }

    stamp("cpu_baseline_and_parity")
    # ---- the `yacht train` side of the path (BASELINE configs[3]) in the same driver-timed run: bench_train.py as a CHILD
    # process (its own handle, its own JSON line), after this process has released its database
    train = None
    db_rebuild_ms = None
    db_rebuild_driver = None
    if rank == 0 and not multi and not args.no_train:
        import subprocess

        if sdb is None:
            db.close()
            # ---- the database built again in this process: its arrays now come out of the library's buffer cache (the
            # handle above has just returned them), so the driver is not asked for memory at all
            if not args.no_scaling_model:
                with torch.cuda.stream(stream):
                    mem0 = ylib.alloc_stats()
                    db2 = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_local, device=local_rank,
                                            flags=YH_DB_NO_DIRECTORY if args.no_indexed else YH_DB_DEFAULT)
                    mem1 = ylib.alloc_stats()
                db_rebuild_ms = round(float(db2.timing()["ms_db_build"]), 2)
                db_rebuild_driver = {"allocations": mem1["driver_allocs"] - mem0["driver_allocs"],
                                     "ms_inside_hipMalloc": round(mem1["ms_in_driver"] - mem0["ms_in_driver"], 1)}
                db2.close()
        del values, offsets
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        try:
            tp = subprocess.run([sys.executable, os.path.join(ROOT, "bench_train.py"), "--steps", "5"]
                                + (["--no-oracle"] if args.no_cpu_baseline else []),
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
            tl = [ln for ln in tp.stdout.splitlines() if ln.startswith("{")]
            train = json.loads(tl[-1]) if tl else {"error": (tp.stderr or "")[-500:]}
            train["returncode"] = tp.returncode
        except Exception as ex:  # noqa: BLE001
            train = {"error": repr(ex)}
        train["wall_s_of_the_child"] = round(time.perf_counter() - t0, 1)

    stamp("train_child")
    # ---- the sketcher next to the path, likewise as a child process with its own JSON line
    sketch_block = None
    if rank == 0 and not multi and not args.no_sketch and not args.no_train:
        import subprocess

        t0 = time.perf_counter()
        try:
            sp = subprocess.run([sys.executable, os.path.join(ROOT, "bench_sketch.py"), "--steps", "6"],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
            sl = [ln for ln in sp.stdout.splitlines() if ln.startswith("{")]
            sketch_block = json.loads(sl[-1]) if sl else {"error": (sp.stderr or "")[-500:]}
            sketch_block["returncode"] = sp.returncode
        except Exception as ex:  # noqa: BLE001
            sketch_block = {"error": repr(ex)}
        sketch_block["wall_s_of_the_child"] = round(time.perf_counter() - t0, 1)

    stamp("sketch_child")
    form = ("batched blocks of %d samples (dist.BatchedRangeRunner)" % BB) if hash_batched else "single steps (one sample per launch)"
    value_batched = ms_per_sample_batched = value_1gpu_same_form = scaling_efficiency = None
    if not multi:
        if scaling_model is not None:  # the runner form on the whole database (scaling_model.per_G["1"])
            ms_per_sample_batched = scaling_model["one_gpu_same_form_ms_per_sample"]
        elif batched is not None:      # (--no-scaling-model: the one-call form, yh_run_batch_device)
            ms_per_sample_batched = batched["ms_per_sample"]
        if ms_per_sample_batched:
            value_batched = round(n_total / (ms_per_sample_batched / 1e3), 1)
    elif same_form is not None:
        value_1gpu_same_form = round(n_total / (same_form["batched_rank0_ms_per_sample"] / 1e3), 1)
        scaling_efficiency = round(value / (world * value_1gpu_same_form), 4)
    train_di = ((train or {}).get("device_input") or {})
    if rank == 0:
        try:
            import scipy
            scipy_version = scipy.__version__
        except Exception:
            scipy_version = None
        out = {
            "metric": "ref-sketch containment queries/sec (yacht run)",
            "value": round(value, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            # SURVEY.md 8d's wall-clock form of the same metric (sample H2D + kernels + result D2H per step, pipelined):
            # `value` above has the samples resident in HBM, as the bench contract asks
            "value_host_inclusive": (host_inclusive or {}).get("value"),
            "ms_per_step_host_inclusive": (host_inclusive or {}).get("ms_per_step"),
            "host_issue_ms_per_step": round(1e3 * t_issued / n_timed, 4),
            # what a step of `value` is.  N = 1: one sample per launch (as the reference runs them: run_YACHT.py:150); N > 1 (default):
            # blocks of --batch-block distinct samples per pass.  The like-for-like single-GPU figure of the batched form is
            # `value_batched` at N = 1 and `value_1gpu_same_form` at N > 1 (rank 0 alone, whole database, same samples, same runner).
            "form": form,
            "value_batched": value_batched, "ms_per_sample_batched": ms_per_sample_batched,
            "value_1gpu_same_form": value_1gpu_same_form,
            "ms_per_sample_1gpu_same_form": (same_form or {}).get("batched_rank0_ms_per_sample"),
            "scaling_efficiency": scaling_efficiency,
            "rccl_world_size": (dist_proof or {}).get("rccl_world_size"),
            "distributed": dist_proof,
            # the timed region itself: `steps` is what the caller asked for, `steps_timed` what the loop ran (see --min-timed-steps)
            "steps_timed": n_timed,
            "timed_region_s": round(elapsed, 6),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "refs_total": n_total,
                # the figures that decide credit, where the driver's record keeps them (VERDICT r04 "next" 2)
                "ms_per_step_host_inclusive": (host_inclusive or {}).get("ms_per_step"),
                "value_host_inclusive": (host_inclusive or {}).get("value"),
                "sample_hash_lookups_per_s": round(n_sample / (ms_per_step / 1e3), 1),  # (the work rate: a step costs per SAMPLE hash, whatever N_refs is)
                "value_batched": value_batched, "ms_per_sample_batched": ms_per_sample_batched,
                "form": form, "value_1gpu_same_form": value_1gpu_same_form, "scaling_efficiency": scaling_efficiency,
                "rccl_world_size": (dist_proof or {}).get("rccl_world_size"),
                "train_device_input_ms": (round(1e3 * train_di["seconds"]["total"], 3) if train_di.get("seconds") else None),
                "train_device_ms": ((train_di.get("roofline") or {}).get("device_ms")),
                "train_frac": ((train_di.get("roofline") or {}).get("frac")),
                "train_traffic_bytes": ((train_di.get("roofline") or {}).get("traffic")),
                "train_host_input_ms": (round(1e3 * train["seconds"]["total"], 3) if (train or {}).get("seconds") else None),
                "train_packed_input_ms": (round(1e3 * train["packed_input"]["seconds"]["total"], 3)
                                          if ((train or {}).get("packed_input") or {}).get("seconds") else None),
                "roofline_frac": roofline.get("frac"), "roofline_kernel_us": round(1e3 * float(roofline.get("kernel_ms_avg") or 0.0), 2),
                "refs_per_gpu": n_local,
                "ref_hashes_per_gpu": Hh,
                "sample_hashes": n_sample,
                "distinct_samples": K,
                "stream_layout": {1: "hash-sorted delta stream", 2: "packed 24-bit keys", 3: "64-bit hashes"}.get(layout, "none"),
                "stream_bytes": int(info.get("stream_bytes", 0)),
                "shared_hashes": info["n_shared_distinct"],
                "shared_postings": info["n_shared_postings"], "holder_set_records": info.get("n_holder_sets"),
                "filter_bytes": info.get("filter_bytes"),
                "ghost_refs_rank0": (sdb.n_ghost if sdb is not None else 0),
                "db_build_ms": round(float(timing["ms_db_build"]), 2),
                # ... without the host time inside hipMalloc (db_build_driver): on this pool a process that starts while the driver is
                # still wiping tens of GB a PREVIOUS process freed waits for that inside its own first allocations
                # (profiles/r05/malloc_modes_probe.txt, fresh_build_probe.txt) -- not the build's work
                "db_build_ms_net_of_hipMalloc": (round(max(float(timing["ms_db_build"]) - db_build_driver["ms_inside_hipMalloc"], 0.0), 2)
                                                 if db_build_driver else None),
                "db_rebuild_ms": db_rebuild_ms,
                "db_build_driver": db_build_driver, "db_rebuild_driver": db_rebuild_driver,
                "device_memory": ylib.alloc_stats(),
                "wall_s_by_section": wall,
                "db_build_note": "device input; HIP events around validation + sort + index + tables, host stalls included: a hipMalloc of a "
                                 "multi-GB block sporadically takes 0.7-4 s on this pool (profiles/r04/malloc_probe.txt, a plain HIP program) -- "
                                 "round 3's 705-815 ms; db_rebuild_ms = the same build again behind the first handle's destroy, its arrays out of "
                                 "the library's buffer cache; db_build_driver / db_rebuild_driver = the trips to the driver inside the two calls "
                                 "and the host time inside them (yh_alloc_stats): what a build above ~60 ms is made of",
                "db_hbm_bytes": info["device_bytes"],
                "pipelined_tail": bool(not multi and args.pipelined_tail),
                "step": "overlap + exclusive counts" + ((f" (blocks of {GB} samples: one all_gather of their subset bits" + (", two blocks in flight" if pipelined else "") + f") + one {'gather to rank 0' if to_root else 'all_gather'} of the count rows per {GB} samples"
                                                       + ("" if args.sync_gather else " (overlapped with the next sample)")) if multi else ""),
                "parallelism": (f"one database, hash space cut x{world}: every GPU holds one hash range of all references, looks up "
                                f"the sample's hashes in it; counts summed" if by_hash else
                                f"one database, references sharded x{world} by hash count"),
                "shard": ("hash" if by_hash else "refs") if multi else None,
                "block_mode": (args.block_mode if by_hash else "steps") if multi else None,
                "samples_per_block": (BB if hash_batched else GB) if multi else None,
                "result_path": (("dense [3, B, N] shares summed" if rowsx is None else "compact rows: value triples summed (dist.BatchRowsReducer)")
                                if hash_batched else None),
                "collective_bytes_per_block_and_rank": (dict(runner.collective_bytes(),
                                                             rows_in_last_block=getattr(rowsx, "last_n_rows", None) if rowsx is not None else None,
                                                             dense_fallbacks=rowsx.n_overflow if rowsx is not None else None,
                                                             word_exchange_repeats=runner.n_words_overflow)
                                                        if hash_batched else None),
                "one_gpu_same_form": same_form,
                "scipy": scipy_version,
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "parity_bit_exact": parity,
            "pipelined_steps_equal_plain_steps": pipelined_ok,
            "batched_blocks_equal_single_steps": blocks_ok,
            "device_resident": device_resident,
            "host_inclusive": host_inclusive,
            "scaling_model": scaling_model,
            "train": train,
            "sketch": sketch_block,
            "real_shape": real_shape,
            "batched": batched,
            "paths": paths,
            "roofline_other_path": other,
        }
        # The ONE stdout line holds the contract keys + small config / roofline / cpu_baseline (bench_line.py: <= 4 kB, strict
        # JSON); the whole result goes to gpurun_out/bench_extras.json and to stderr (a 20.7 kB line left the
        # driver's record unparsed).
        import bench_line

        extras_path = bench_line.write_extras(out, ROOT, args.extras)
        sys.stderr.write("bench.py: full result (also in %s):\n%s\n" % (extras_path, json.dumps(out, indent=1, default=str)))
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(json_fd, (bench_line.dumps(out, extras_path) + "\n").encode())
    if sdb is not None:
        sdb.close()
    else:
        db.close()  # (a second close is a no-op)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and any(not p_["equals_default_path"] for p_ in paths.values()):
        print("bench.py: the two lookup paths differ", file=sys.stderr)
        return 1
    if rank == 0 and batched is not None and not batched["equals_single_sample_step"]:
        print("bench.py: the batched run differs from the single-sample step", file=sys.stderr)
        return 1
    if rank == 0 and host_inclusive is not None and not host_inclusive["equals_device_resident"]:
        print("bench.py: host-buffer path differs from the device-resident path", file=sys.stderr)
        return 1
    if rank == 0 and train is not None and (train.get("returncode", 1) != 0 or train.get("parity_bit_exact") is False):
        print("bench.py: the train block failed or differs from its references: " + json.dumps(train)[:600], file=sys.stderr)
        return 1
    if rank == 0 and dist_proof is not None and not dist_proof["all_reduce_ok"]:
        print("bench.py: the one-word all-reduce over the ranks gave a wrong sum: " + json.dumps(dist_proof)[:400], file=sys.stderr)
        return 1
    if rank == 0 and blocks_ok is False:
        print("bench.py: the batched blocks differ from the single-sample steps", file=sys.stderr)
        return 1
    if rank == 0 and pipelined_ok is False:
        print("bench.py: the pipelined steps differ from the plain ones", file=sys.stderr)
        return 1
    if rank == 0 and parity is False:
        print("bench.py: GPU counts differ from the CPU oracle", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
