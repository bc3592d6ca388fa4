#!/usr/bin/env python3
"""bench_train.py — the `yacht train` side of the hot path (BASELINE.json configs[3]): pairwise
reference x reference containment + greedy dedup, 2 000 clusters x 5 sketches of ~5 000 hashes,
ani_thresh 0.95 (C = 0.95**31).  Secondary to bench.py (whose contract the driver reads): this
prints one JSON line with the pair-query rate, the phase times and a parity verdict.

    python bench_train.py [--clusters 2000] [--size 5000] [--steps 5] [--oracle-clusters 400]

Phases timed on the device path (inputs on the host, as `yacht train` has them):
  upload+build   yh_db_create: CSR upload, partitioned CSR, radix-sort inverted index
  pairwise       yh_pairwise: posting lists -> dense int32 row block (atomics) -> threshold -> pairs
  select         yh_train_select (host)
A query = one unordered reference pair whose intersection size is produced: N(N-1)/2 per pass.
Parity: the same pipeline on the first `--oracle-clusters` clusters against the CPU oracle
(inverted index + scatter, the reference's algorithm), bit-exact pairs / statistics / selection.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None."""
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


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clusters", type=int, default=2000)
    ap.add_argument("--size", type=int, default=5000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--oracle-clusters", type=int, default=400)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1: the work split over ranks (launch with torch.distributed.run)")
    ap.add_argument("--shard", default="hash", choices=["hash", "rows"],
                    help="N > 1: hash (default) = every rank uploads, indexes and intersects ONE HASH RANGE of all sketches, the "
                         "partial pair counts are summed (dist.hash_range_pairwise); rows = every rank holds everything and computes a "
                         "block of rows of the pair matrix (dist.sharded_pairwise)")
    ap.add_argument("--no-scaling-model", action="store_true", help="N = 1: skip the measurement of one rank's share of a G-way hash-range run")
    ap.add_argument("--no-device-input", action="store_true",
                    help="N = 1: skip the extra passes with the sketches already in HBM (yh_db_create_device): the kernels' own rate, no PCIe")
    ap.add_argument("--device-input", action="store_true", help="N = 1: ONLY the device-input passes (for profiling the kernels)")
    ap.add_argument("--no-packed-input", action="store_true", help="N = 1: skip the passes from the packed database (yh_db_create_packed)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--share-gpu", action="store_true", help="testing: all ranks on cuda:0 (gloo)")
    args = ap.parse_args()
    wall = {}  # where this process's wall-clock went (seconds per section): a slow box shows up here, not in the medians
    t_wall = [time.perf_counter()]

    def stamp(name: str) -> None:
        now = time.perf_counter()
        wall[name] = round(wall.get(name, 0.0) + now - t_wall[0], 2)
        t_wall[0] = now

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench_train.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if world > 1:
        json_fd = os.dup(1)
        os.dup2(2, 1)  # (RCCL prints its banner to stdout)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            if args.backend == "gloo":
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (no host-name look-ups: they stall on some boxes)
            dist.init_process_group(args.backend)

    from yacht_amd import _lib, synth
    from yacht_amd.engine import YH_DB_PAIRWISE_ONLY, RefDB, train_select

    if _lib.device_count() < 1:
        print("bench_train.py needs an MI355X (no CPU fallback)", file=sys.stderr)
        return 2
    stamp("imports_and_device_count")
    c = 0.95 ** 31
    values, offsets = synth.config4(seed=1003, n_clusters=args.clusters, size=args.size)
    n = offsets.size - 1
    sizes = np.diff(offsets).astype(np.uint32)

    range_slice = None
    if world > 1 and args.shard == "hash":
        from yacht_amd import dist as ydist

        bnd = ydist.hash_range_bounds(int(values.max()), world)
        range_slice = ydist.slice_csr_to_hash_range(values, offsets, bnd[rank], bnd[rank + 1])
    # ---- the sketches already in HBM (the `yacht run` -> re-train case; every rank of a multi-GPU train behind the first
    # broadcast): yh_db_create_device + yh_pairwise + selection -- no PCIe in the call, the kernels' own rate
    stamp("synthetic_sketches")
    device_input = None
    if world == 1 and not args.no_device_input:
        import torch

        d_values = torch.from_numpy(values.view(np.int64)).to(f"cuda:{local_rank}")
        d_offsets = torch.from_numpy(offsets.astype(np.int64)).to(f"cuda:{local_rank}")
        torch.cuda.synchronize()
        td_b, td_p, td_s, kb, kp = [], [], [], [], []
        for _ in range(args.steps + 2):
            t0 = time.perf_counter()
            dbd = RefDB.from_device(d_values.data_ptr(), d_offsets.data_ptr(), n, sizes=sizes, device=local_rank, flags=YH_DB_PAIRWISE_ONLY)
            t1 = time.perf_counter()
            di, dj, dc = dbd.pairwise(c)
            t2 = time.perf_counter()
            dsel = train_select(sizes, di, dj)
            t3 = time.perf_counter()
            tmd = dbd.timing()
            dstats = dbd.index_stats()
            dbd.close()
            td_b.append(t1 - t0); td_p.append(t2 - t1); td_s.append(t3 - t2)
            kb.append(tmd["ms_db_build"]); kp.append(tmd["ms_pairwise_kernels"])
        med = lambda x: float(np.median(x[2:]))  # noqa: E731
        d_total = med(td_b) + med(td_p) + med(td_s)
        alg_d = 8 * int(offsets[-1]) + 12 * int(di.size)
        k_ms_d = med(kb) + med(kp)
        device_input = {
            "value": round(n * (n - 1) / 2 / d_total, 1), "unit": "pair-queries/s",
            "seconds": {"create_device": round(med(td_b), 5), "pairwise": round(med(td_p), 5), "select": round(med(td_s), 5), "total": round(d_total, 5),
                        "db_build_kernels_ms": round(med(kb), 3), "pairwise_kernels_ms": round(med(kp), 3)},
            "roofline": {"bound": "hbm", "achieved": round(alg_d / 1e9 / (k_ms_d / 1e3), 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(alg_d / 1e9 / (k_ms_d / 1e3) / 8000.0, 4), "algorithmic_bytes": alg_d,
                         "device_ms": round(k_ms_d, 3),
                         "note": "one-touch bytes (8 H + 12 P_out, SURVEY.md 8d) over the device time of build + pairwise (HIP events around "
                                 "validation, the distribution sort with its fused last pass, and the row pass)"},
            "kernels": "k_piece_bounds (regions of every sketch, ordering check, records cleared), k_piece_part (pieces read in place -> buckets), "
                       "k_bucket_group5 (grouping by hash in LDS -> pairwise records), k_pair_rows",
            "how": "sketches resident in HBM: yh_db_create_device(PAIRWISE_ONLY) + yh_pairwise + yh_train_select; medians of the passes behind two warm-ups",
            "_results": (di, dj, dc, dsel, dstats),
        }
        # HBM bytes actually moved, from the rocprofv3 --pmc passes of scripts/pmc_train.sh -- attached only when they were taken
        # from THIS source of the kernels (profiles/traffic_<round>_train.json is keyed on its sha256)
        try:
            import hashlib

            hsh = hashlib.sha256()
            for f_ in ("yh_sort.hip", "yh_pairwise.hip", "yh_common.h"):
                with open(os.path.join(ROOT, "yacht_amd", "csrc", f_), "rb") as fh:
                    hsh.update(fh.read())
            tag_ = hsh.hexdigest()[:16]
            device_input["roofline"]["traffic"] = None
            for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if name.startswith("traffic_") and name.endswith("_train.json"):
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        tr = json.load(f)
                    if tr.get("source_tag") == tag_ and tr.get("n_hashes") == int(offsets[-1]):
                        device_input["roofline"]["traffic"] = tr["hbm_bytes_per_call"]
                        device_input["roofline"]["moved_over_algorithmic"] = tr["moved_over_algorithmic"]
                        device_input["roofline"]["traffic_per_kernel"] = {k_: v_["hbm_bytes"] for k_, v_ in tr["per_kernel"].items()}
                        device_input["roofline"]["traffic_provenance"] = {"file": "profiles/" + name, "source_tag": tag_, "taken": tr.get("taken"),
                                                                           "commit": tr.get("commit"), "read_factor": tr.get("read_factor")}
                        break
        except Exception as ex:  # noqa: BLE001
            device_input["roofline"]["traffic_error"] = repr(ex)
        del d_values, d_offsets
        stamp("device_input_passes")
    t_build, t_pair, t_sel = [], [], []
    k_pair_ms = []
    pi = pj = pc = None
    for _ in range(0 if (args.device_input and device_input is not None) else args.steps + 1):  # first pass is warm-up
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        if world > 1 and args.shard == "hash":
            # this rank's hash range of every sketch: uploaded, sorted, indexed and intersected here; partial counts summed
            from yacht_amd import dist as ydist

            v_r, o_r = range_slice  # (cut on the host once, outside the timed passes: `yacht train` has the sketches on the host)
            db = RefDB(v_r, o_r, device=local_rank, flags=YH_DB_PAIRWISE_ONLY)
            t1 = time.perf_counter()
            pi, pj, pc, stats_sum = ydist.hash_range_pairwise(lambda: (*db.pairwise(0.0), db.index_stats()), n, sizes, c,
                                                             device="cpu" if args.backend != "nccl" else f"cuda:{local_rank}")
            t2 = time.perf_counter()
            sel = train_select(sizes, pi, pj)
            t3 = time.perf_counter()
            tm = db.timing()
            info = db.info()
            stats = stats_sum
            db.close()
            t_build.append(t1 - t0)
            t_pair.append(t2 - t1)
            t_sel.append(t3 - t2)
            k_pair_ms.append(tm["ms_pairwise_kernels"])
            continue
        db = RefDB(values, offsets, device=local_rank, flags=YH_DB_PAIRWISE_ONLY)  # what `yacht train` creates (train_core.py)
        t1 = time.perf_counter()
        if world == 1:
            pi, pj, pc = db.pairwise(c)
        else:  # every rank holds the whole set and computes a block of rows cut by cumulative shared hashes
            from yacht_amd import dist as ydist

            nsh = torch.zeros(n, dtype=torch.int32, device=f"cuda:{local_rank}")
            torch.cuda.synchronize()  # (torch fills on its own stream; the library copies on the handle's)
            db.nshared_device(nsh.data_ptr())
            db.synchronize()
            plan = ydist.pair_row_plan(nsh.cpu().numpy(), world)
            pi, pj, pc = ydist.sharded_pairwise(lambda b, e: db.pairwise(c, b, e), plan,
                                                device="cpu" if args.backend != "nccl" else f"cuda:{local_rank}")
            key = pi.astype(np.int64) * n + pj
            if not bool(np.all(np.diff(key) > 0)):
                bad = np.flatnonzero(np.diff(key) <= 0)
                print(f"rank {rank}: gathered pairs not sorted: plan {plan}, {pi.size} pairs, first bad at {bad[:3]}: "
                      f"{pi[bad[0] - 1:bad[0] + 3]} {pj[bad[0] - 1:bad[0] + 3]}", file=sys.stderr, flush=True)
        t2 = time.perf_counter()
        sel = train_select(sizes, pi, pj)
        t3 = time.perf_counter()
        tm = db.timing()
        info = db.info()
        stats = db.index_stats()
        db.close()
        t_build.append(t1 - t0)
        t_pair.append(t2 - t1)
        t_sel.append(t3 - t2)
        k_pair_ms.append(tm["ms_pairwise_kernels"])
    stamp("host_input_passes")
    # ---- the same call from the PACKED database (yh_csr_pack: ~5.7 bytes per hash cross the bus instead of 8) -------------
    packed_input = None
    if world == 1 and t_build and not args.no_packed_input:
        from yacht_amd.engine import csr_pack

        t0 = time.perf_counter()
        blob = csr_pack(values, offsets)
        t_pack = time.perf_counter() - t0
        pb, pp, ps_ = [], [], []
        for _ in range(args.steps + 1):
            t0 = time.perf_counter()
            dbp = RefDB.from_packed(blob, device=local_rank, flags=YH_DB_PAIRWISE_ONLY)
            t1 = time.perf_counter()
            qi, qj, qc = dbp.pairwise(c)
            t2 = time.perf_counter()
            qsel = train_select(sizes, qi, qj)
            t3 = time.perf_counter()
            qstats = dbp.index_stats()
            tmp_ = dbp.timing()
            dbp.close()
            pb.append(t1 - t0)
            pp.append(t2 - t1)
            ps_.append(t3 - t2)
        b_, p_, s_ = (float(np.median(x[1:])) for x in (pb, pp, ps_))
        packed_input = {
            "value": round(n * (n - 1) // 2 / (b_ + p_ + s_), 1), "unit": "pair-queries/s",
            "seconds": {"create_packed": round(b_, 5), "pairwise": round(p_, 5), "select": round(s_, 5), "total": round(b_ + p_ + s_, 5),
                        "db_build_kernels_ms": round(float(tmp_["ms_db_build"]), 3), "h2d_ms": round(float(tmp_.get("ms_h2d", 0.0)), 3)},
            "packed_bytes": int(blob.nbytes), "bytes_per_hash": round(blob.nbytes / max(int(offsets[-1]), 1), 3),
            "bus_ms_at_56_GBps": round(blob.nbytes / 56e9 * 1e3, 3),
            "pack_on_host_ms": round(1e3 * t_pack, 1),
            "equals_host_input": bool(np.array_equal(qi, pi) and np.array_equal(qj, pj) and np.array_equal(qc, pc) and np.array_equal(qsel, sel)
                                      and tuple(int(x) for x in qstats) == tuple(int(x) for x in stats)),
            "how": "yh_csr_pack once (where the sketches are parsed: not inside the call), then yh_db_create_packed(PAIRWISE_ONLY) + yh_pairwise + "
                   "yh_train_select; the chunks are expanded in HBM under the upload; medians of the passes behind a warm-up",
        }
        # ... and with the blob in page-locked host memory (a database that is kept resident for many calls)
        try:
            from yacht_amd.engine import PinnedArray

            pin = PinnedArray(blob.size, np.uint64)
            pin.array[:] = blob
            tb = []
            for _ in range(args.steps + 1):
                t0 = time.perf_counter()
                dbp = RefDB.from_packed(pin.array, device=local_rank, flags=YH_DB_PAIRWISE_ONLY)
                t1 = time.perf_counter()
                qi2 = dbp.pairwise(c)[0]
                t2 = time.perf_counter()
                h2d_pin = float(dbp.timing().get("ms_h2d", 0.0))
                dbp.close()
                tb.append((t1 - t0, t2 - t1))
            packed_input["page_locked_blob"] = {"create_packed": round(float(np.median([x[0] for x in tb[1:]])), 5),
                                                "pairwise": round(float(np.median([x[1] for x in tb[1:]])), 5),
                                                "total_with_select": round(float(np.median([x[0] + x[1] for x in tb[1:]])) + s_, 5),
                                                "h2d_ms": round(h2d_pin, 3), "pairs_equal": bool(np.array_equal(qi2, pi))}
            pin.close()
        except Exception as ex:  # noqa: BLE001
            packed_input["page_locked_blob"] = {"error": repr(ex)[:200]}
        del blob
        stamp("packed_input_passes")
    if not t_build:  # --device-input: only those passes ran; the line's main figures are theirs
        pi, pj, pc, sel, stats = device_input["_results"]
        t_build = [0.0, device_input["seconds"]["create_device"]]
        t_pair = [0.0, device_input["seconds"]["pairwise"]]
        t_sel = [0.0, device_input["seconds"]["select"]]
        k_pair_ms = [0.0, device_input["seconds"]["pairwise_kernels_ms"]]
        tm = {"ms_db_build": device_input["seconds"]["db_build_kernels_ms"]}
        info = {"n_shared_postings": -1}
    t_build, t_pair, t_sel, k_pair_ms = (float(np.median(x[1:])) for x in (t_build, t_pair, t_sel, k_pair_ms))
    total = t_build + t_pair + t_sel
    if world > 1:  # the slowest rank
        tt = torch.tensor([t_build, t_pair, t_sel, total], dtype=torch.float64)
        tt = tt if args.backend != "nccl" else tt.to(f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_build, t_pair, t_sel, total = (float(x) for x in tt.tolist())
    n_pairs_unordered = n * (n - 1) // 2

    parity = None
    cpu = None
    if not args.no_oracle and rank == 0:
        from oracle import oracle

        k = min(args.oracle_clusters, args.clusters) * 5
        v2, o2 = values[: int(offsets[k])], offsets[: k + 1]
        s2 = sizes[:k]
        hw_threads = oracle.hardware_threads()
        quota = _cpu_quota()  # (the GPU boxes of this pool show 256 hardware threads and grant 16 CPUs: threads beyond that only take turns)
        cores = max(1, min(hw_threads, int(quota))) if quota else hw_threads
        t0 = time.perf_counter()
        wi, wj, wc, wstats = oracle.train_pairs(v2, o2, c, threads=cores)
        wsel = oracle.train_select(s2, wi, wj)
        t_cpu = time.perf_counter() - t0
        with RefDB(v2, o2) as db2:
            gi, gj, gc = db2.pairwise(c)
            gstats = db2.index_stats()
        gsel = train_select(s2, gi, gj)
        parity = bool(np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
                      and gstats == wstats and np.array_equal(gsel, wsel))
        if k == n:  # the oracle saw everything: the timed passes' own results (sharded over the ranks when N > 1) against it
            parity = parity and bool(np.array_equal(pi, wi) and np.array_equal(pj, wj) and np.array_equal(pc, wc)
                                     and tuple(int(x) for x in stats) == tuple(wstats) and np.array_equal(sel, wsel))
        cpu = {"value": round(k * (k - 1) / 2 / t_cpu, 1), "unit": "pair-queries/s", "cores": cores, "kind": "port",
               "cpu_model": _cpu_model(),
               "sample": f"first {k} sketches ({int(o2[-1])} hashes): index build on 1 thread + scatter on {cores} "
                         f"threads + selection, {t_cpu:.2f} s", "hardware_threads": hw_threads, "cpu_quota": quota}
        stamp("cpu_port_and_parity")
        # the GENUINE reference executable (oracle/_ref, built from /root/reference/src/cpp/main.cpp in the
        # build container and shipped as a binary): timed on the same sketches, and its files compared
        # with the HIP path's results
        if oracle.have_ref_exe():
            import re
            import tempfile

            from yacht_amd.train_core import format_pair_line

            refs2 = [v2[int(o2[j]):int(o2[j + 1])] for j in range(k)]
            with tempfile.TemporaryDirectory() as d:
                t0 = time.perf_counter()
                rsel, rlines, rout = oracle.run_ref_exe(refs2, c, d, threads=min(cores, 64))
                t_ref = time.perf_counter() - t0
            phases = {m.group(1): int(m.group(2)) for m in re.finditer(r"Time taken to ([a-z ]+): (\d+) milliseconds", rout)}
            core_s = sum(phases.get(x, 0) for x in ("build index", "compute intersection matrix", "do yacht train")) / 1e3
            glines = [format_pair_line(int(i), int(j), int(cc), int(s2[i]), int(s2[j])) for i, j, cc in zip(gi, gj, gc)]
            same = bool(rsel == gsel.tolist() and rlines == glines)
            parity = parity and same
            cpu = {"value": round(k * (k - 1) / 2 / core_s, 1), "unit": "pair-queries/s", "cores": min(cores, 64),
                   "kind": "reference", "cpu_model": _cpu_model(),
                   "sample": f"reference run_yacht_train_core -t {min(cores, 64)} on the first {k} sketches: index "
                             f"{phases.get('build index', 0)} ms + matrix {phases.get('compute intersection matrix', 0)} ms "
                             f"+ selection {phases.get('do yacht train', 0)} ms (file reading {phases.get('read all sketches', 0)} ms "
                             f"not counted; whole process incl. writing the JSON inputs {t_ref:.1f} s)",
                   "outputs_equal_hip_path": same, "hardware_threads": hw_threads, "cpu_quota": quota,
                   "port_value": round(k * (k - 1) / 2 / t_cpu, 1)}

    stamp("genuine_reference_executable")
    # configs[3] at its REAL size against the genuine reference: tests/golden/golden_train_cfg3.json holds what
    # oracle/_ref/run_yacht_train_core wrote for this very input (selected ids in walk order, the three index statistics,
    # a sha256 over the sorted pair lines) -- made once in the build container by tests/golden/make_golden.py cfg3
    golden = None
    gpath = os.path.join(ROOT, "tests", "golden", "golden_train_cfg3.json")
    if rank == 0 and args.clusters == 2000 and args.size == 5000 and os.path.exists(gpath):
        import hashlib

        from yacht_amd.train_core import format_pair_line

        with open(gpath) as f:
            g = json.load(f)
        same_input = hashlib.sha256(values.tobytes() + offsets.tobytes()).hexdigest() == g["input_sha256"]
        lines = [format_pair_line(int(i), int(j), int(cc), int(sizes[i]), int(sizes[j])) for i, j, cc in zip(pi, pj, pc)]
        golden = {"same_input": same_input,
                  "selected_equal": sel.tolist() == g["selected"],
                  "pair_lines_equal": len(lines) == g["n_pair_lines"] and hashlib.sha256("\n".join(lines).encode()).hexdigest() == g["pair_lines_sha256"],
                  "stats_equal": list(stats) == [g["stats"]["distinct"], g["stats"]["singletons"], g["stats"]["index"]]}
        golden["all_equal"] = all(golden.values())
        if parity is not False and not golden["all_equal"]:
            parity = False
        elif parity is None:
            parity = golden["all_equal"]

    stamp("full_size_golden_compare")
    # what ONE rank of a G-way hash-range run does (N = 1 only): rank 0's range of every sketch through upload + index +
    # pairwise(0.0) on this GPU; the merge of the partial lists is measured on this host with G copies of that list
    scaling_model = None
    if world == 1 and not args.no_scaling_model:
        from yacht_amd import dist as ydist

        per_g = {}
        for G in (2, 4, 8):
            bnd = ydist.hash_range_bounds(int(values.max()), G)
            v_r, o_r = ydist.slice_csr_to_hash_range(values, offsets, bnd[0], bnd[1])
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                dbg = RefDB(v_r, o_r, device=local_rank, flags=YH_DB_PAIRWISE_ONLY)
                gi_, gj_, gc_ = dbg.pairwise(0.0)
                ts.append(time.perf_counter() - t0)
                dbg.close()  # (as in the timed passes above: the handle is released behind the results)
            # the sum per pair of G such lists, on the device as the RCCL path does it (the lists arrive there)
            import torch

            one = torch.from_numpy(np.stack([x.astype(np.int64) for x in (gi_, gj_, gc_)])).to(f"cuda:{local_rank}")
            tm_ = []
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ydist.merge_partial_pairs_device(n, torch.cat([one] * G, dim=1), sizes, c)
                tm_.append(time.perf_counter() - t0)
            t_merge = float(np.median(tm_[1:]))
            per_g[str(G)] = {"rank0_upload_index_pairwise_s": round(float(np.median(ts[1:])), 5), "hashes_in_range": int(v_r.size),
                             "partial_pairs": int(gi_.size), "device_merge_of_G_lists_s": round(t_merge, 5)}
        scaling_model = {"per_G": per_g,
                         "predicted_total_s": {g: round(v["rank0_upload_index_pairwise_s"] + v["device_merge_of_G_lists_s"] + t_sel + 0.0003, 5)
                                               for g, v in per_g.items()},
                         "predicted_speedup_vs_1gpu": {g: round(total / (v["rank0_upload_index_pairwise_s"] + v["device_merge_of_G_lists_s"] + t_sel + 0.0003), 2)
                                                       for g, v in per_g.items()},
                         "how": "rank 0's hash range of every sketch: yh_db_create(PAIRWISE_ONLY) + yh_pairwise(c = 0) timed on this GPU; + "
                                "the sum-per-pair of G such lists on the device (torch.unique + scatter_add, as over RCCL) + selection + 0.3 ms assumed for "
                                "the two small all-gathers"}

    stamp("scaling_model")
    # algorithmic bytes: every reference hash once + one (i, j, count) per emitted pair
    alg = 8 * int(offsets[-1]) + 12 * int(pi.size)
    out = {
        "metric": "ref x ref containment pair-queries/sec (yacht train)",
        "value": round(n_pairs_unordered / total, 1),
        "unit": "pair-queries/s",
        "n_gpus": world,
        "scaling": ("strong (one hash range of every sketch per rank: upload, index and pairwise all divide; partial counts summed)"
                    if args.shard == "hash" else "strong (rows of one pair matrix over ranks; every rank builds the whole index)"),
        "config": {"workload": f"configs[3]: {args.clusters} clusters x 5 sketches of ~{args.size} hashes, C=0.95**31",
                   "n_refs": int(n), "n_hashes": int(offsets[-1]), "pairs_emitted": int(pi.size),
                   "selected": int(sel.size), "shared_hashes": int(stats[2]), "postings": int(info["n_shared_postings"])},
        "seconds": {"upload_and_build": round(t_build, 5), "pairwise": round(t_pair, 5), "select": round(t_sel, 5),
                    "total": round(total, 5), "pairwise_kernels_ms": round(k_pair_ms, 3),
                    "db_build_kernels_ms": round(float(tm["ms_db_build"]), 3)},
        "algorithmic_bytes": alg,
        "algorithmic_GBps_over_total": round(alg / total / 1e9, 2),
        "cpu_baseline": cpu,
        "parity_bit_exact": parity,
        "full_size_vs_genuine_reference": golden,
        "scaling_model": scaling_model,
        "wall_s_by_section": wall,
        "device_memory": _lib.alloc_stats(),
    }
    if packed_input is not None:
        if not packed_input["equals_host_input"]:
            out["parity_bit_exact"] = False
        out["packed_input"] = packed_input
    if device_input is not None:
        di, dj, dc, dsel, dstats = device_input.pop("_results")
        device_input["equals_host_input"] = bool(np.array_equal(di, pi) and np.array_equal(dj, pj) and np.array_equal(dc, pc)
                                                 and np.array_equal(dsel, sel) and tuple(int(x) for x in dstats) == tuple(int(x) for x in stats))
        if not device_input["equals_host_input"]:
            parity = False
            out["parity_bit_exact"] = False
        out["device_input"] = device_input
    # roofline-style figure of the kernels alone (B = 8 H + 12 P_out), HBM peak 8 TB/s
    k_ms = float(tm["ms_db_build"]) + k_pair_ms
    alg_upload_s = 8 * int(offsets[-1]) / 56e9  # what the bus needs for the sketches alone (measured: 56 GB/s from pageable memory)
    out["roofline"] = {"bound": "hbm", "kernels": "k_piece_bounds per chunk (under the upload), k_piece_part, k_bucket_group5 (yh_sort.hip: regions read in place as pieces of the ascending sketches, buckets grouped by hash in LDS into the pairwise records), k_pair_rows",
                       "achieved": round(alg / 1e9 / (k_ms / 1e3), 1) if k_ms > 0 else None, "peak": 8000.0, "unit": "GB/s",
                       "frac": round(alg / 1e9 / (k_ms / 1e3) / 8000.0, 4) if k_ms > 0 else None,
                       "exposed_device_ms": round(1e3 * total - 1e3 * alg_upload_s, 3),
                       "traffic": (device_input or {}).get("roofline", {}).get("traffic"),
                       "note": "one-touch bytes over the SUM of the device time of all build + pairwise kernels (k_scan_refs' ordering check of "
                               "every chunk included); the bounds pass of the chunks runs while the database crosses PCIe -- "
                               "exposed_device_ms = the call minus 8 H bytes at 56 GB/s; traffic = the device-input kernels' (the same kernels)"}
    if world > 1:
        if rank == 0:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
    else:
        print(json.dumps(out), flush=True)
    return 0 if parity in (None, True) else 1


if __name__ == "__main__":
    sys.exit(main())
