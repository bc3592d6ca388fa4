"""GPU: dist.ShardedRefDB on the HIP backend -- the `yacht run` step over ranks (ghost references,
yh_run_local_device / yh_run_finish_device) -- with TWO PROCESSES sharing the one GPU of the test box
(gloo moves the collectives through the host; RCCL refuses two ranks on one device), checked against
the CPU oracle on the WHOLE database.  Plus bench.py's own N = 2 path on the same arrangement."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["YH_ROOT"])
import torch
import torch.distributed as dist
from oracle import oracle
from yacht_amd import dist as ydist, synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo")
n_total = int(os.environ["YH_NREFS"])
plan = synth.global_db_plan(31, n_total, cluster_frac=0.6)   # many clusters: the cuts go through some
# cuts moved forward to the next cluster MEMBER (its founder stays on the left of the cut): every cut
# goes through a cluster, so every rank has ghost references
cuts = [0]
for r in range(1, world):
    j = (n_total * r) // world
    while plan["parent"][j] == j:
        j += 1
    cuts.append(j)
cuts.append(n_total)
shards = [(cuts[r], cuts[r + 1]) for r in range(world)]
b, e = shards[rank]
values, offsets = synth.global_db_refs_device(plan, np.arange(b, e), device="cuda:0")
sdb = ydist.ShardedRefDB(values, offsets, ydist.HipLocalBackend(0))
all_v, all_o = synth.global_db_refs_device(plan, np.arange(n_total), device="cuda:0")
hv, ho = all_v.cpu().numpy().view(np.uint64), all_o.cpu().numpy().astype(np.uint64)
ok = True
for i, (shape, n_s) in enumerate((("present", 200000), ("real", 5000), ("present", 0))):
    s = synth.global_db_sample_device(plan, 77 + i, n_sample=max(n_s, 1), n_present=60, device="cuda:0", shape=shape)
    if n_s == 0:
        s = s[:0]
    c = sdb.run(s)
    torch.cuda.synchronize()
    full = sdb.gather(c).cpu().numpy().view(np.uint32)
    hs = s.cpu().numpy().view(np.uint64)
    want_ov = oracle.overlap(hv, ho, hs, threads=4)
    want_e, want_m = oracle.exclusive(hv, ho, want_ov > 0, hs)
    for name, got, want in (("overlap", full[0], want_ov), ("n_excl", full[1], want_e), ("n_match", full[2], want_m)):
        if not np.array_equal(got, want):
            ok = False
            bad = np.flatnonzero(got != want)
            print(f"rank {rank} sample {i} {name}: {bad.size} differ, first {bad[:5]} got {got[bad[:5]]} want {want[bad[:5]]}", flush=True)
print(f"rank {rank}: ghosts {sdb.n_ghost}, rows {sdb.n_rows}, ok {ok}", flush=True)
if world > 1 and sdb.n_ghost == 0:
    ok = False
    print("no ghost references: the cut did not go through a cluster", flush=True)
sdb.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def _launch(world: int, n_refs: int, tmp_path, lookup: str = "auto", worker: str = None):
    script = tmp_path / "worker.py"
    script.write_text(worker or WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), YH_ROOT=ROOT, YH_NREFS=str(n_refs), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if lookup != "auto":  # force the streaming / the sample-driven lookup inside yh_run_local_device (a tuning switch)
            env["YH_DEBUG_TUNING"] = "1"
            env["YH_LOOKUP"] = lookup
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    return [p.returncode for p in procs], outs


# (three ranks: the gloo tests on the CPU, tests/test_dist_ghost_cpu.py / test_dist_range_cpu.py; three processes starting up on
# one GPU box took 25-50 s of a 4-minute suite on some boxes)
@pytest.mark.parametrize("world,lookup", [(1, "auto"), (2, "indexed")])
def test_sharded_refdb_hip_processes_share_gpu(hip_lib, tmp_path, world, lookup):
    rcs, outs = _launch(world, 3000, tmp_path, lookup)
    assert all(rc == 0 for rc in rcs), "\n".join(outs)


def test_sharded_refdb_over_rccl_one_rank(hip_lib, tmp_path):
    """The real backend: torch.distributed "nccl" (= RCCL) with one rank, every collective of the exchange and of the
    step forced to run (YH_FORCE_EXCHANGE=1) -- what a 1-GPU box can check of the call shapes the 8-GPU run uses."""
    script = tmp_path / "worker_rccl.py"
    script.write_text(WORKER.replace('dist.init_process_group("gloo")',
                                     'dist.init_process_group("nccl", device_id=dev)'))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               YH_ROOT=ROOT, YH_NREFS="3000", HSA_ENABLE_IPC_MODE_LEGACY="0", YH_FORCE_EXCHANGE="1")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr


def _whole_result(line):
    """bench.py's stdout line carries the contract keys only (bench_line.py); the whole result is in the file it names."""
    import bench_line

    full = bench_line.read_extras(line, ROOT)
    for k in ("metric", "value", "n_gpus", "ms_per_step", "parity_bit_exact", "scaling"):
        assert full[k] == line[k] or (isinstance(full[k], float) and abs(full[k] - line[k]) < 1e-9), k
    return full


def _extras_arg():
    import tempfile

    return ["--extras", os.path.join(tempfile.mkdtemp(prefix="yh_bench_"), "extras.json")]


@pytest.mark.parametrize("shard", ["hash", "refs"])
def test_bench_over_rccl_one_rank(hip_lib, shard):
    """bench.py's N > 1 code path (sharded step + async collectives of the bits and the count rows) on RCCL with one rank."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", YH_FORCE_EXCHANGE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--shard", shard, "--steps", "6", "--min-timed-steps", "0", "--min-timed-ms", "0", "--warmup", "2",
                        "--refs", "4000", "--sample-hashes", "100000", "--samples", "3", "--percentile-steps", "8", "--present", "50"] + _extras_arg(),
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(p.stdout.strip().splitlines()[-1]) < 4096
    assert line["parity_bit_exact"] is True and _whole_result(line)["config"]["parallelism"].startswith("one database")


@pytest.mark.parametrize("shard", ["hash", "refs"])
def test_bench_two_ranks_share_gpu_strong_and_weak(hip_lib, tmp_path, shard):
    """bench.py --gpus 2 over gloo on one GPU: the driver-run N > 1 path (hash-range shards by default, reference shards
    + ghosts with --shard refs), bit-exact against the oracle."""
    for scaling in ("strong", "weak"):
        port = _free_port()
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen(
                [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--min-timed-steps", "0", "--min-timed-ms", "0", "--warmup", "2",
                 "--backend", "gloo", "--share-gpu", "--scaling", scaling, "--shard", shard, "--refs", "4000", "--sample-hashes", "100000",
                 "--samples", "3", "--percentile-steps", "8", "--present", "50", "--batch-block", "4", "--extras", str(tmp_path / f"extras_{scaling}.json")],
                env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        res = [p.communicate(timeout=900) for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(o + e for o, e in res)
        line = json.loads(res[0][0].strip().splitlines()[-1])
        assert line["config"]["shard"] == shard and (shard != "hash" or line["config"]["samples_per_block"] == 4)
        assert line["config"]["rccl_world_size"] == 2 and (shard != "hash" or line["config"]["scaling_efficiency"] > 0)
        line = _whole_result(line)
        assert line["parity_bit_exact"] is True and line["n_gpus"] == 2 and line["scaling"] == scaling
        assert line["config"]["refs_total"] == (4000 if scaling == "strong" else 8000) and line["config"]["shard"] == shard
        # the self-proving part of an N > 1 line (VERDICT r04 "next" 1)
        assert line["rccl_world_size"] == 2 and line["distributed"]["all_reduce_ok"] is True and len(line["distributed"]["ranks"]) == 2
        assert sorted(x["rank"] for x in line["distributed"]["ranks"]) == [0, 1]
        if shard == "hash":  # the default: batched blocks -- with rank 0's single-GPU pass of the same form beside it
            assert line["form"].startswith("batched blocks") and line["config"]["form"] == line["form"]
            assert line["value_1gpu_same_form"] > 0 and line["scaling_efficiency"] > 0
            assert abs(line["scaling_efficiency"] - line["value"] / (2 * line["value_1gpu_same_form"])) < 1e-3
            cb = line["config"]["collective_bytes_per_block_and_rank"]
            assert cb["total"] == cb["subset_words_all_gather"] + cb["result"] and cb["subset_words_all_gather"] < cb["subset_words_dense_form"] * 2
            assert line["batched_blocks_equal_single_steps"] is True
        else:
            assert line["form"].startswith("single steps") and line["value_1gpu_same_form"] is None


def test_bench_two_ranks_hash_range_single_steps(hip_lib, tmp_path):
    """--block-mode steps: the per-sample half-steps of the hash-range shards (the batched blocks are the default)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                        "--block-mode", "steps", "--steps", "6", "--min-timed-steps", "0", "--min-timed-ms", "0", "--warmup", "2", "--refs", "4000", "--sample-hashes", "100000",
                        "--samples", "3", "--percentile-steps", "8", "--present", "50"] + _extras_arg(),
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = _whole_result(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
    assert line["parity_bit_exact"] is True and line["config"]["block_mode"] == "steps" and line["config"]["shard"] == "hash"


def test_bench_starts_its_own_ranks(hip_lib):
    """`python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE, the way the driver starts `--gpus 1`: bench.py
    spawns its two ranks as child processes (before anything touches the GPU) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                        "--steps", "6", "--min-timed-steps", "0", "--min-timed-ms", "0", "--warmup", "2", "--refs", "4000", "--sample-hashes", "100000", "--samples", "3",
                        "--percentile-steps", "8", "--present", "50"] + _extras_arg(),
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, p.stdout
    line = _whole_result(json.loads(lines[0]))
    assert line["parity_bit_exact"] is True and line["n_gpus"] == 2
    assert line["rccl_world_size"] == 2 and line["distributed"]["all_reduce_ok"] is True and line["scaling_efficiency"] > 0


RANGE_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["YH_ROOT"])
import torch
import torch.distributed as dist
from oracle import oracle
from yacht_amd import dist as ydist, synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo")
n_total = int(os.environ["YH_NREFS"])
plan = synth.global_db_plan(31, n_total, cluster_frac=0.6)
all_v, all_o = synth.global_db_refs_device(plan, np.arange(n_total), device="cuda:0")
hv, ho = all_v.cpu().numpy().view(np.uint64), all_o.cpu().numpy().astype(np.uint64)
bounds = ydist.hash_range_bounds(int(hv.max()), world)
v, o = ydist.slice_to_hash_range(all_v, all_o, bounds[rank], bounds[rank + 1])
hr = ydist.HashRangeRefDB(v, o, bounds, ydist.HipRangeBackend(0), block=2)
ok = True
def check(full, s, what):
    global ok
    full = full.cpu().numpy().view(np.uint32)
    hs = s.cpu().numpy().view(np.uint64)
    want_ov = oracle.overlap(hv, ho, hs, threads=4)
    want_e, want_m = oracle.exclusive(hv, ho, want_ov > 0, hs)
    for name, got, want in (("overlap", full[0], want_ov), ("n_excl", full[1], want_e), ("n_match", full[2], want_m)):
        if not np.array_equal(got, want):
            ok = False
            bad = np.flatnonzero(got != want)
            print(f"rank {rank} {what} {name}: {bad.size} differ, first {bad[:5]} got {got[bad[:5]]} want {want[bad[:5]]}", flush=True)
ss = []
for i, (shape, n_s) in enumerate((("present", 200000), ("real", 5000), ("present", 0), ("present", 30000))):
    s = synth.global_db_sample_device(plan, 77 + i, n_sample=max(n_s, 1), n_present=60, device="cuda:0", shape=shape)
    ss.append(s[:0] if n_s == 0 else s)
for i, s in enumerate(ss):
    c = hr.run(s)
    torch.cuda.synchronize()
    check(hr.gather(c), s, f"sample {i}")
# blocks of two samples per exchange, two blocks in flight, one reduce per block (what bench.py --gpus N drives)
blk = [torch.zeros((2, 3, hr.n_total), dtype=torch.int32, device=dev) for _ in range(2)]
for g in range(2):
    hr.begin(ss[g], blk[0][g], 0, g)
hr.exchange(0)
for g in range(2):
    hr.begin(ss[2 + g], blk[1][g], 1, g)
hr.exchange(1)
for g in range(2):
    hr.end(blk[0][g], 0, g)
for g in range(2):
    hr.end(blk[1][g], 1, g)
torch.cuda.synchronize()
tot = [hr.reduce(b.clone()) for b in blk]
for k, s in enumerate(ss):
    check(tot[k // 2][k % 2], s, f"blocked {k}")
# the batched form: all four samples in ONE pass around one exchange of their subset words
cb = hr.run_batch(ss)
torch.cuda.synchronize()
full = hr.reduce(cb.clone())
for k, s in enumerate(ss):
    check(full[:, k, :], s, f"batched {k}")
# the result path in compact form: blocks of two samples rotate through three batch slots, their value triples are summed
# to rank 0 in ONE collective per block (BatchRowsReducer); an undersized first collective takes the dense fallback
blocks = [[ss[0], ss[1]], [ss[2], ss[3]], [ss[3], ss[0]], [ss[1]]]
for first_cap in (None, 8):
    red = ydist.BatchRowsReducer(hr, batch=2, dst=0, nbuf=3, cap_rows=first_cap)
    cnt = [torch.zeros((3, 2, hr.n_total), dtype=torch.int32, device=dev) for _ in range(3)]
    wrd = [torch.zeros(hr.n_total, dtype=torch.int64, device=dev) for _ in range(3)]
    gth = [torch.zeros((world, hr.n_total), dtype=torch.int64, device=dev) for _ in range(3)]
    got = {}
    def finish(j):
        rows, dense = red.finish(j % 3)
        if rank == 0:
            nb = len(blocks[j])
            got[j] = (dense[:, :nb] if rows is None else ydist.BatchRowsReducer.rows_to_dense(rows, nb, hr.n_total)).clone()
    def second_half(j):
        b = j % 3
        hr.batch_end(len(blocks[j]), gth[b], cnt[b], slot=b)
        red.send(b, len(blocks[j]), cnt[b], slot=b)
    for j, blk_ in enumerate(blocks):
        b = j % 3
        if j >= 3:
            finish(j - 3)
        hr.batch_begin(hr.pack_batch(blk_), cnt[b], wrd[b], slot=b)
        hr.batch_exchange(wrd[b], gth[b])
        if j:
            second_half(j - 1)
    second_half(len(blocks) - 1)
    for j in range(max(0, len(blocks) - 3), len(blocks)):
        finish(j)
    torch.cuda.synchronize()
    if first_cap is not None and not (red.n_overflow >= 1 and red.cap > first_cap):
        ok = False
        print(f"rank {rank}: the undersized collective went unnoticed", flush=True)
    if rank == 0:
        for j, blk_ in enumerate(blocks):
            for k, s in enumerate(blk_):
                check(got[j][:, k, :], s, f"compact rows (first cap {first_cap}) block {j} sample {k}")
# a second half without its first half in the slot is refused
if int(v.numel()):
    try:
        hr.batch_end(2, gth[0], cnt[0], slot=1)
        ok = False
        print(f"rank {rank}: a second half without a first half was accepted", flush=True)
    except Exception as e:
        if "YH_ERR_INVALID_ARG" not in str(e):
            raise
print(f"rank {rank}: range [{bounds[rank]}, {bounds[rank + 1]}), {int(v.numel())} of {int(all_v.numel())} hashes, ok {ok}", flush=True)
hr.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("world,lookup", [(1, "auto"), (2, "stream"), (2, "indexed")])
def test_hash_range_refdb_hip_processes_share_gpu(hip_lib, tmp_path, world, lookup):
    """dist.HashRangeRefDB on the HIP backend: every rank holds one hash range of ALL references; the reduced counts
    equal the oracle on the whole database (1, 2 and 3 processes sharing the GPU over gloo)."""
    rcs, outs = _launch(world, 3000, tmp_path, lookup, worker=RANGE_WORKER)
    assert all(rc == 0 for rc in rcs), "\n".join(outs)
