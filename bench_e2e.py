#!/usr/bin/env python3
"""bench_e2e.py — the `yacht train` and `yacht run` COMMANDS end to end, files in, files out.

The reference publishes one timing for this path (README.md:276): `yacht train` on the 85 205 GTDB
rs214 representative genomes, "around 12 minutes" and 52 GB of memory for the whole command -- unzip the
sourmash database, gunzip 85 205 `.sig.gz`, read every signature's metadata, compare, write the
manifest.  bench.py times the resident kernels; this script times the commands:

  train   python -m yacht_amd train --ref_file refs.sig.zip ...   on a synthetic database of the same
          shape (85 205 sketches, k=31, scaled=1000, sizes LogNormal(ln 3300, 0.6), ~10 % in clusters),
          written first as a real sourmash-style zip (SOURMASH-MANIFEST.csv + signatures/<md5>.sig.gz)
  run     python -m yacht_amd run --json ... --sample_file sample.sig.zip   against what train left
          behind, for a ~1 M-hash sample and for a real-hit-shape 83 k-hash sample, 5 coverages

Per command one JSON line: wall seconds, the phase breakdown (yacht_amd/phases.py), the slowest host
phase, peak resident memory of the process (and of its pool workers).  Lines go to stdout and, with
--out DIR, to DIR/e2e_train.json and DIR/e2e_run.json.

    python bench_e2e.py --refs 85205 --threads 64 --work /tmp/yacht_e2e --out profiles/r02
"""
from __future__ import annotations

import argparse
import gzip
import io
import json
import os
import resource
import shutil
import sys
import time
import zipfile
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _encode(job):
    """One signature -> (location, md5, manifest row, gzipped JSON bytes).  Runs in pool workers."""
    from yacht_amd import sigio

    name, mins = job
    sig = sigio.make_signature(mins, ksize=31, scaled=1000, name=name, filename=name + ".fna", abundances=np.ones(mins.size, np.int64))
    md5 = sig.md5sum()
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0, compresslevel=1) as g:
        g.write(sigio.dumps_signature(sig).encode("utf-8"))
    mh = sig.minhash
    row = [f"signatures/{md5}.sig.gz", md5, md5[:8], mh.ksize, mh.moltype, mh.num, mh.scaled, len(mh), int(mh.track_abundance),
           sig.name, sig.filename]
    return row, buf.getvalue()


def write_db_zip(path: str, values: np.ndarray, offsets: np.ndarray, threads: int) -> float:
    import csv

    from yacht_amd import sigio

    t0 = time.perf_counter()
    n = offsets.size - 1
    jobs = ((f"genome_{j:06d}", values[int(offsets[j]):int(offsets[j + 1])]) for j in range(n))
    rows = []
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z, Pool(threads) as pool:
        for row, data in pool.imap(_encode, jobs, chunksize=64):
            z.writestr(row[0], data)
            rows.append(row)
        out = io.StringIO()
        out.write(sigio.MANIFEST_HEADER + "\n")
        w = csv.writer(out, lineterminator="\n")
        w.writerow(sigio.MANIFEST_COLUMNS)
        w.writerows(rows)
        z.writestr(sigio.MANIFEST_NAME, out.getvalue())
    return time.perf_counter() - t0


def write_sample_zip(path: str, mins: np.ndarray, name: str) -> None:
    from yacht_amd import sigio

    sig = sigio.make_signature(mins, ksize=31, scaled=1000, name=name, filename=name + ".fq",
                               abundances=np.ones(mins.size, np.int64))
    sigio.write_sig_zip([sig], path)


def peak_rss_gb():
    kb_self = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    kb_kids = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    return round(kb_self / 1e6, 2), round(kb_kids / 1e6, 2)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=85_205)
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 8, 64))
    ap.add_argument("--more-threads", type=int, default=0, help="also run the default and the --no_sig_files train with this many host threads")
    ap.add_argument("--work", default="/tmp/yacht_e2e")
    ap.add_argument("--out", default="")
    ap.add_argument("--seed", type=int, default=1002)
    ap.add_argument("--keep", action="store_true", help="keep the work directory")
    args = ap.parse_args()

    import torch

    from yacht_amd import cli, phases, synth

    if not torch.cuda.is_available():
        print("bench_e2e.py needs an MI355X (the commands have no CPU fallback)", file=sys.stderr)
        return 2
    shutil.rmtree(args.work, ignore_errors=True)
    os.makedirs(args.work)
    free_gb = shutil.disk_usage(args.work).free / 1e9
    need_gb = args.refs * 110e3 / 1e9  # ~30 KB gz + ~70 KB JSON per sketch, + the packed copy
    if free_gb < need_gb + 2:
        print(f"bench_e2e.py: {free_gb:.1f} GB free under {args.work}, ~{need_gb:.1f} GB needed", file=sys.stderr)
        return 2

    # ---- the synthetic database as files --------------------------------------------------------------
    t0 = time.perf_counter()
    plan = synth.global_db_plan(args.seed, args.refs)
    v, o = synth.global_db_refs_device(plan, np.arange(args.refs), device="cuda:0")
    values, offsets = v.cpu().numpy().view(np.uint64), o.cpu().numpy().astype(np.uint64)
    s_big = synth.global_db_sample_device(plan, args.seed + 1, n_sample=1_000_000, n_present=200, device="cuda:0").cpu().numpy().view(np.uint64)
    s_real = synth.global_db_sample_device(plan, args.seed + 2, n_sample=83_000, device="cuda:0", shape="real").cpu().numpy().view(np.uint64)
    del v, o
    torch.cuda.empty_cache()
    t_gen = time.perf_counter() - t0
    ref_zip = os.path.join(args.work, "refs.sig.zip")
    t_zip = write_db_zip(ref_zip, values, offsets, args.threads)
    write_sample_zip(os.path.join(args.work, "sample_1M.sig.zip"), s_big, "sample_1M")
    write_sample_zip(os.path.join(args.work, "sample_real.sig.zip"), s_real, "sample_real")
    zip_gb = os.path.getsize(ref_zip) / 1e9
    print(f"# database: {args.refs} sketches, {values.size} hashes, zip {zip_gb:.2f} GB (generated {t_gen:.1f} s, written {t_zip:.1f} s)",
          file=sys.stderr, flush=True)
    del values, offsets

    lines = {}
    # ---- yacht train: the default (one native pass over the archive, the unzipped .sig files left behind as the reference
    # leaves them), the same without the files, and rounds 1-3's three Python passes -----------------------------------
    variants = [("default", [], args.threads), ("no_sig_files", ["--no_sig_files"], args.threads),
                ("python_ingest", ["--python_ingest"], args.threads)]
    if args.more_threads:
        variants += [(f"default_{args.more_threads}_threads", [], args.more_threads),
                     (f"no_sig_files_{args.more_threads}_threads", ["--no_sig_files"], args.more_threads)]
    train_lines = {}
    for tag, extra, n_threads in variants:
        out_dir = os.path.join(args.work, "out_" + tag)
        os.makedirs(out_dir)
        phases.reset()
        t0 = time.perf_counter()
        rc = cli.main(["train", "--ref_file", ref_zip, "--ksize", "31", "--ani_thresh", "0.95", "--prefix", "db", "--outdir", out_dir,
                       "--num_threads", str(n_threads), "--force"] + extra)
        wall = time.perf_counter() - t0
        ph = phases.snapshot()
        top = {k: v for k, v in ph.items() if "/" not in k}
        rss_self, rss_kids = peak_rss_gb()
        n_kept = sum(1 for _ in open(os.path.join(out_dir, "db_processed_manifest.tsv"))) - 1
        train_lines[tag] = {"rc": rc, "threads": n_threads, "wall_s": round(wall, 2), "phases_s": ph, "slowest_phase": max(top, key=top.get) if top else None,
                            "unaccounted_s": round(wall - sum(top.values()), 2), "references_kept": n_kept,
                            "peak_rss_gb": {"process": rss_self, "largest_pool_worker": rss_kids}}
        if tag != "default" and not args.keep:
            shutil.rmtree(out_dir, ignore_errors=True)
    # ---- the same command as a user starts it: a FRESH process (the in-process calls above find numpy / pandas / torch imported
    # and the library and the HIP context up already)
    import subprocess

    cold = {}
    for cmd_tag, extra in (("default", []), ("no_sig_files", ["--no_sig_files"])):
        out_dir_c = os.path.join(args.work, f"out_cold_{cmd_tag}")
        os.makedirs(out_dir_c)
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        t0 = time.perf_counter()
        rc = subprocess.run([sys.executable, "-m", "yacht_amd", "train", "--ref_file", ref_zip, "--ksize", "31", "--ani_thresh", "0.95",
                             "--prefix", "db", "--outdir", out_dir_c, "--num_threads", str(args.threads), "--force"] + extra,
                            env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        cold[cmd_tag] = {"rc": rc, "wall_s": round(time.perf_counter() - t0, 2)}
        shutil.rmtree(out_dir_c, ignore_errors=True)
    train_lines["default"]["fresh_process_wall_s"] = cold
    out_dir = os.path.join(args.work, "out_default")
    n_kept = train_lines["default"]["references_kept"]
    lines["train"] = dict(train_lines["default"],
                          command="yacht train (python -m yacht_amd train): zip -> manifest + packed DB",
                          workload=f"{args.refs} synthetic sketches k=31 scaled=1000 (rs214 shape), sourmash-style zip of {zip_gb:.2f} GB",
                          threads=args.threads, variants={k: v for k, v in train_lines.items() if k != "default"},
                          reference_published="README.md:276: 85 205 GTDB genomes, ~12 minutes, 52 GB (whole command, CPU)")
    print(json.dumps(lines["train"]), flush=True)

    # ---- yacht run ---------------------------------------------------------------------------------------
    runs = []
    for tag in ("sample_1M", "sample_real"):
        res_dir = os.path.join(args.work, "res_" + tag)
        os.makedirs(res_dir)
        phases.reset()
        t0 = time.perf_counter()
        rc = cli.main(["run", "--json", os.path.join(out_dir, "db_config.json"), "--sample_file", os.path.join(args.work, tag + ".sig.zip"),
                       "--min_coverage_list", "1", "0.5", "0.1", "0.05", "0.01", "--outdir", res_dir, "--num_threads", str(args.threads)])
        wall = time.perf_counter() - t0
        ph = phases.snapshot()
        top = {k: v for k, v in ph.items() if "/" not in k}
        rows = sum(1 for _ in open(os.path.join(res_dir, "results", "result_all.txt"))) - 1
        rss_self, rss_kids = peak_rss_gb()
        runs.append({"sample": tag, "rc": rc, "wall_s": round(wall, 3), "phases_s": ph,
                     "slowest_phase": max(top, key=top.get) if top else None,
                     "slowest_leaf_phase": max((k for k in ph if not any(o.startswith(k + "/") for o in ph)), key=ph.get),
                     "result_rows": rows, "peak_rss_gb": {"process": rss_self, "largest_pool_worker": rss_kids}})
    res_dir = os.path.join(args.work, "res_cold")
    os.makedirs(res_dir)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.perf_counter()
    rc = subprocess.run([sys.executable, "-m", "yacht_amd", "run", "--json", os.path.join(out_dir, "db_config.json"), "--sample_file",
                         os.path.join(args.work, "sample_1M.sig.zip"), "--min_coverage_list", "1", "0.5", "0.1", "0.05", "0.01",
                         "--outdir", res_dir, "--num_threads", str(args.threads)],
                        env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
    cold_run = {"rc": rc, "wall_s": round(time.perf_counter() - t0, 2)}
    lines["run"] = {
        "fresh_process_wall_s_sample_1M": cold_run,
        "command": "yacht run (python -m yacht_amd run): config + sample zip -> result_all.txt (+ xlsx when openpyxl exists), 5 coverages",
        "workload": f"the {n_kept} references `yacht train` kept, packed DB memory-mapped from disk",
        "runs": runs,
    }
    print(json.dumps(lines["run"]), flush=True)
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        for k, v in lines.items():
            with open(os.path.join(args.out, f"e2e_{k}.json"), "w") as f:
                f.write(json.dumps(v) + "\n")
    if not args.keep:
        shutil.rmtree(args.work, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
