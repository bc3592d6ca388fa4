// yh_query.hip — the per-query kernels of libyacht_hip.so (gfx950 / CDNA4, wave64).
//
//   k_index_lookup_tile  the sample-driven lookup (default): one lane per SAMPLE hash -- presence bit, then one
//                        64-byte bucket of the table over the database's distinct hashes -- hits summed per
//                        reference in an LDS table, one global atomic per (workgroup, reference)
//   k_stream_lookup      the streaming lookup (databases small against the sample): all (hash, reference) pairs
//                        in hash order, one delta byte each; a workgroup stages the sample hashes of its key range
//                        in LDS, rebuilds the lanes' key spans of 16-block super-blocks (v_sad_u8 + DPP scans) and
//                        lets the sample keys probe them; candidates are confirmed against the full hash
//   k_reduce_replicas    overlap counts from the replicated counters, which it clears (zero at rest); the subset
//                        "overlap > 0" as bits; in the run step also n_match, the singleton part of n_excl and the
//                        work list of the exclusive pass
//   k_excl_pieces        subset-exclusive hash counts, one wave per work record: over a reference's DISTINCT holder
//                        sets (run step) or its postings with hit flags (arbitrary subsets); k_excl_worklist,
//                        k_excl_final (the arithmetic of hypothesis_recovery_src.py:165-204)
//   k_overlap_bsearch    one wave per reference, lanes binary-search the sample: the independent cross-check
// (the batched run is yh_batch.hip, `yacht train`'s pairwise counts yh_pairwise.hip)
#include "yh_common.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

constexpr int WAVE = 64;
constexpr int EXCL_PIECE_C = YH_EXCL_PIECE;  // postings per work record of the exclusive pass

typedef u64 u64x2 __attribute__((ext_vector_type(2)));  // one 16-byte global load per lane

// Blocks b and b+8 share an XCD (observed round-robin dispatch; speed only, never correctness).
// Map the hardware block id to a logical id so that one XCD receives CONSECUTIVE logical ids:
// the workgroups that share a partition's sample tile and offsets then share one L2.
__device__ __forceinline__ u32 xcd_remap(u32 bid, u32 nwg) {
    const u32 q = nwg >> 3, r = nwg & 7u;
    const u32 xcd = bid & 7u;
    const u32 base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// XCD-local counting (YH_XCD_ATOMICS, experiment): replica = the XCD the wave runs on, adds at WORKGROUP scope.
// A device-scope atomic bypasses the XCD's L2 (the eight L2s are not coherent with each other) and is
// executed at the memory side; an add that only waves of ONE XCD ever make to its address can stay in that
// XCD's L2 -- the kernel boundary writes the lines back before k_reduce_replicas sums the replicas.
#ifndef YH_XCD_ATOMICS
#define YH_XCD_ATOMICS 0
#endif
__device__ __forceinline__ u32 xcc_id() { return (u32)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }  // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ void count_add(u32* p, u32 v) {
#if YH_XCD_ATOMICS
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    atomicAdd(p, v);
#endif
}
__device__ __forceinline__ u32 replica_of(u32 wg, u32 rep_mask) {
#if YH_XCD_ATOMICS
    return xcc_id() & rep_mask;
#else
    return wg & rep_mask;
#endif
}

// overlap[j] = sum of the replicas, which are CLEARED as they are read (d_reps is zero at rest: no
// zeroing pass in front of the next query); optionally also the subset mask "overlap > 0" as bytes
// and as bits (one ballot per wave: references 64w .. 64w+63 -> two words), and the zeroing of the
// three exclusive accumulators [3][n] the kernels behind this one add into.
// With `fused` (the `yacht run` step on the hash-sorted stream, where the subset is "overlap > 0"):
// also the second replica set (hits on shared hashes), and the exclusive counts up to the part that
// needs the posting lists: n_match[j] = overlap - hits on shared hashes (final: a shared hash found
// in the sample has all its holders in the subset, so it is exclusive to none of them), n_excl[j] =
// |R_j| - nshared_j for the subset (k_excl_chunks adds the shared hashes whose other holders are all
// outside the subset).
struct FusedRun {
    u32* reps2;
    const u32* sizes;
    const u32* nshared;
    u32* n_excl;
    u32* n_match;
    u32* bits_out;  // may be null: a second copy of the subset bits (the sharded run hands them to the other ranks)
    // work list of the exclusive pass behind this kernel: pieces of the subset's references' holder-set records
    // (reference j owns records [rpo[j], rpo[j + 1]) -- yh_db::d_hpo)
    const u32* rpo;
    uint4* work;
    u32* work_count;  // zero when this kernel starts (the lookup kernel in front of it clears it)
    u32 work_refs;    // references below this number produce work (sharded run: the ghosts behind do not)
};

// Append the posting pieces of every reference in the subset to the work list: (reference, first posting,
// end) per <= EXCL_PIECE postings.  One atomic per workgroup.  All 256 threads of the block must call it.
__device__ __forceinline__ void append_pieces(bool in_subset, u32 j, u32 nshared_j, u32 rpo_j, uint4* __restrict__ work,
                                              u32* __restrict__ work_count, u32* lds /* [waves + 1] */) {
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const u32 np = in_subset ? (nshared_j + (u32)EXCL_PIECE_C - 1u) / (u32)EXCL_PIECE_C : 0u;
    u32 v = np;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 t = (u32)__shfl_up((int)v, off);
        if (lane >= (u32)off) v += t;
    }
    if (lane == 63) lds[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 total = 0;
        for (u32 q = 0; q < n_waves; ++q) total += lds[q];
        lds[n_waves] = total ? atomicAdd(work_count, total) : 0u;
    }
    __syncthreads();
    u32 at = lds[n_waves] + v - np;
    for (u32 q = 0; q < wv; ++q) at += lds[q];
    const u32 last = rpo_j + nshared_j;  // (= rpo[j + 1]: the record carries its end, one read less per piece)
    for (u32 i = 0; i < np; ++i) {
        const u32 first = rpo_j + i * (u32)EXCL_PIECE_C;
        work[at + i] = make_uint4(j, first, min(first + (u32)EXCL_PIECE_C, last), 0u);
    }
}

// (body: any workgroup size that is a multiple of 64; `wg` = the workgroup's index among the reducing ones; lds: waves + 1 words)
__device__ __forceinline__ void reduce_replicas_body(u32 wg, u32* lds, u32* __restrict__ reps, u32 R, u64 n,
                                                     u32* __restrict__ out, u8* __restrict__ mask,
                                                     u32* __restrict__ maskbits, u32* __restrict__ excl3,
                                                     const FusedRun& fused) {
    const u64 j = wg * (u64)blockDim.x + threadIdx.x;
    u32 acc = 0;
    // everything this thread reads is requested up front (one memory round trip, not one per dependent use)
    u32 size_j = 0, nshared_j = 0, rpo_j = 0, rpo_end = 0;
    if (j < n && fused.reps2) {
        size_j = fused.sizes[j];
        nshared_j = fused.nshared[j];
        if (fused.work) { rpo_j = fused.rpo[j]; rpo_end = fused.rpo[j + 1]; }
    }
    if (j < n) {
        for (u32 r = 0; r < R; ++r) {
            acc += reps[(u64)r * n + j];
            reps[(u64)r * n + j] = 0;
        }
        out[j] = acc;
        if (mask) mask[j] = acc ? 1 : 0;
        if (excl3) { excl3[j] = 0; excl3[n + j] = 0; excl3[2 * n + j] = 0; }
        if (fused.reps2) {
            u32 acc2 = 0;
            for (u32 r = 0; r < R; ++r) {
                acc2 += fused.reps2[(u64)r * n + j];
                fused.reps2[(u64)r * n + j] = 0;
            }
            fused.n_match[j] = acc - acc2;
            if (fused.n_excl) fused.n_excl[j] = acc ? size_j - nshared_j : 0u;
        }
    }
    if (maskbits) {
        const u64 bal = __ballot(acc != 0);
        // (whole 256-reference blocks are written, whatever the workgroup size: what the arrays are sized for)
        if ((threadIdx.x & 63) == 0 && (j >> 5) < ((n + 255) / 256) * 8) {
            maskbits[(j >> 5)] = (u32)bal;
            maskbits[(j >> 5) + 1] = (u32)(bal >> 32);
            if (fused.bits_out) {
                fused.bits_out[(j >> 5)] = (u32)bal;
                fused.bits_out[(j >> 5) + 1] = (u32)(bal >> 32);
            }
        }
    }
    if (fused.work) append_pieces(acc != 0 && j < fused.work_refs, (u32)j, rpo_end - rpo_j, rpo_j, fused.work, fused.work_count, lds);
}

__global__ void __launch_bounds__(256) k_reduce_replicas(u32* __restrict__ reps, u32 R, u64 n,
                                                         u32* __restrict__ out, u8* __restrict__ mask,
                                                         u32* __restrict__ maskbits, u32* __restrict__ excl3,
                                                         FusedRun fused) {
    __shared__ u32 lds[17];
    reduce_replicas_body(blockIdx.x, lds, reps, R, n, out, mask, maskbits, excl3, fused);
}

// The reducer of the fused launch: RPT references per thread, all their reads in flight together, ONE scan and ONE
// atomic per workgroup for the work list -- a workgroup covers RPT x blockDim references in one pass (the launch has
// only the lookup's spare wave slots for this role: looping over 1024-reference blocks serialized five round trips).
template <int RPT>
__device__ __forceinline__ void reduce_replicas_multi(u32 blk, u32* lds, u32* __restrict__ reps, u32 R, u64 n, u32* __restrict__ out,
                                                      u32* __restrict__ maskbits, const FusedRun& f) {
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    u64 j[RPT];
    u32 acc[RPT], acc2[RPT], size_j[RPT], nsh[RPT], rpo0[RPT], rpo1[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        j[r] = ((u64)blk * RPT + r) * blockDim.x + threadIdx.x;
        acc[r] = acc2[r] = size_j[r] = nsh[r] = rpo0[r] = rpo1[r] = 0;
        if (j[r] < n) {
            size_j[r] = f.sizes[j[r]];
            nsh[r] = f.nshared[j[r]];
            if (f.work) { rpo0[r] = f.rpo[j[r]]; rpo1[r] = f.rpo[j[r] + 1]; }
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        if (j[r] < n)
            for (u32 k = 0; k < R; ++k) {
                acc[r] += reps[(u64)k * n + j[r]];
                acc2[r] += f.reps2[(u64)k * n + j[r]];
            }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        if (j[r] < n) {
            for (u32 k = 0; k < R; ++k) { reps[(u64)k * n + j[r]] = 0; f.reps2[(u64)k * n + j[r]] = 0; }
            out[j[r]] = acc[r];
            f.n_match[j[r]] = acc[r] - acc2[r];
            if (f.n_excl) f.n_excl[j[r]] = acc[r] ? size_j[r] - nsh[r] : 0u;
        }
        const u64 bal = __ballot(acc[r] != 0);
        if (lane == 0 && (j[r] >> 5) < ((n + 255) / 256) * 8) {  // (whole 256-reference blocks: what the arrays are sized for)
            maskbits[(j[r] >> 5)] = (u32)bal;
            maskbits[(j[r] >> 5) + 1] = (u32)(bal >> 32);
            if (f.bits_out) { f.bits_out[(j[r] >> 5)] = (u32)bal; f.bits_out[(j[r] >> 5) + 1] = (u32)(bal >> 32); }
        }
    }
    if (!f.work) return;
    u32 np[RPT], mine = 0;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const bool in = acc[r] != 0 && j[r] < f.work_refs;
        np[r] = in ? (rpo1[r] - rpo0[r] + (u32)EXCL_PIECE_C - 1u) / (u32)EXCL_PIECE_C : 0u;
        mine += np[r];
    }
    u32 v = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 t = (u32)__shfl_up((int)v, off);
        if (lane >= (u32)off) v += t;
    }
    if (lane == 63) lds[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 total = 0;
        for (u32 q = 0; q < n_waves; ++q) total += lds[q];
        lds[n_waves] = total ? atomicAdd(f.work_count, total) : 0u;
    }
    __syncthreads();
    u32 at = lds[n_waves] + v - mine;
    for (u32 q = 0; q < wv; ++q) at += lds[q];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const u32 last = rpo1[r];
        for (u32 i = 0; i < np[r]; ++i) {
            const u32 first = rpo0[r] + i * (u32)EXCL_PIECE_C;
            f.work[at + i] = make_uint4((u32)j[r], first, min(first + (u32)EXCL_PIECE_C, last), 0u);
        }
        at += np[r];
    }
}

// the same work list for a subset that arrives as bits (general path); work_count zeroed by the caller
__global__ void __launch_bounds__(256) k_excl_worklist(u64 n, const u32* __restrict__ maskbits, const u32* __restrict__ nshared,
                                                       const u32* __restrict__ rpo, uint4* __restrict__ work,
                                                       u32* __restrict__ work_count) {
    __shared__ u32 lds[17];
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    const bool in = j < n && ((maskbits[j >> 5] >> (j & 31u)) & 1u);
    append_pieces(in, (u32)j, in ? nshared[j] : 0u, in ? rpo[j] : 0u, work, work_count, lds);
}

// Sharded run: the subset bits of the GHOST references (copies of other ranks' references that share a
// hash with one of ours, appended behind the local ones from a multiple of 64 on) come from the
// all-gathered bits of their owners: ghost k's bit is bit src[k] of `global_bits`.
__global__ void __launch_bounds__(256) k_ghost_bits(const u32* __restrict__ global_bits, const u32* __restrict__ src,
                                                    u64 ghost_begin, u64 n_ghost, u32* __restrict__ maskbits) {
    const u64 k = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    bool on = false;
    if (k < n_ghost) {
        const u32 b = src[k];
        on = (global_bits[b >> 5] >> (b & 31u)) & 1u;
    }
    const u64 bal = __ballot(on);
    if ((threadIdx.x & 63) == 0 && (k & ~63ull) < n_ghost) {
        const u64 w = (ghost_begin + k) >> 5;  // ghost_begin is a multiple of 64
        maskbits[w] = (u32)bal;
        maskbits[w + 1] = (u32)(bal >> 32);
    }
}

// Hash-range sharding (yh_run_finish_range_device): every rank holds the hashes of ONE range of every reference, so
// "reference j overlaps the sample" is the OR over the ranks of their local subset bits.  From the gathered rows
// (rank r's row starts r * stride words in) this kernel makes the global subset bits, the part of n_excl that needs
// no posting list -- |R_j| - nshared_j of THIS rank's range, for the references of the global subset -- and the work
// list of the exclusive pass over them.  work_count is zero when it starts (the lookup kernel cleared it).
__global__ void __launch_bounds__(256) k_range_mask(const u32* __restrict__ gathered, u32 n_ranks, u64 stride, u64 n,
                                                    const u32* __restrict__ sizes, const u32* __restrict__ nshared,
                                                    const u32* __restrict__ rpo, u32* __restrict__ maskbits,
                                                    u32* __restrict__ n_excl, uint4* __restrict__ work,
                                                    u32* __restrict__ work_count) {
    __shared__ u32 lds[17];
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    bool in = false;
    u32 size_j = 0, nshared_j = 0, rpo_j = 0, rpo_end = 0;
    if (j < n) {
        u32 w = 0;
        for (u32 r = 0; r < n_ranks; ++r) w |= gathered[(u64)r * stride + (j >> 5)];
        in = (w >> (j & 31u)) & 1u;
        size_j = sizes[j];
        nshared_j = nshared[j];
        if (work) { rpo_j = rpo[j]; rpo_end = rpo[j + 1]; }
        n_excl[j] = in ? size_j - nshared_j : 0u;
    }
    const u64 bal = __ballot(in);
    if ((threadIdx.x & 63) == 0) {
        maskbits[(j >> 5)] = (u32)bal;
        maskbits[(j >> 5) + 1] = (u32)(bal >> 32);
    }
    if (work) append_pieces(in, (u32)j, rpo_end - rpo_j, rpo_j, work, work_count, lds);
}

__global__ void __launch_bounds__(256) k_mask_bits(const u8* __restrict__ mask, u64 n, u32* __restrict__ maskbits) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    const u64 bal = __ballot(j < n && mask[j] != 0);
    if ((threadIdx.x & 63) == 0) {
        maskbits[(j >> 5)] = (u32)bal;
        maskbits[(j >> 5) + 1] = (u32)(bal >> 32);
    }
}

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// ---- K1 over the hash-sorted delta stream ------------------------------------------------------------
// The stream (yh_common.h) is every (hash, reference) pair in ascending hash order, the hashes
// truncated to t = hash >> sshift and stored as ONE BYTE per element, the difference to the previous
// element.  Both sides being sorted, the lookup is a merge: a wave takes a block of 1024 elements
// (16 bytes per lane, one coalesced load), rebuilds where its lanes' key spans start with byte sums
// (v_sad_u8) and one wave scan, and then lets the FEW sample keys that fall into the block's key
// range probe it -- ~3 per block for a 10^6-hash sample against 3.3 x 10^8 hashes -- instead of
// testing every stream key against the sample.  A probe that equals an element's t is a candidate
// (stream position, sample index); k_resolve_stream confirms it against the 64-bit hash of that
// position, so truncation collisions (~|S|/48 per query) and filler elements never count.
// A workgroup owns a contiguous range of blocks = a range of t; the sample hashes of that range are
// staged in LDS (sorted t values + a bucket directory) -- if there are none the range is not read.
#ifndef YH_ST_SLOTS
#define YH_ST_SLOTS 6144
#endif
constexpr int ST_SLOTS = YH_ST_SLOTS;   // sample keys per tile
constexpr int ST_PAD = 64 + 2;          // readable sentinels behind the last key (a wave reads 64 slots at once)
constexpr int ST_CAP = ST_SLOTS - ST_PAD;

struct StreamHit {
    u32 n_refs;
    u32* reps;
    u32 rep_mask;
    const uint2* srec;  // per stream position: {hash low word, reference | shared bit}
    u8* hitflag;        // may be null: hit[g] = 1 for every shared hash g found in the sample (g by search in gsh)
    const u64* gsh;     // the shared hashes ascending, and their number (only read when hitflag is set)
    u32 n_gsh;
    u32* reps2;         // may be null: a second set of replicas counting the hits ON SHARED HASHES only
    const u64* sample;
    const u32* bad;     // may be null: *bad == bad_gen = the sample failed the ordering check queued in front of
    u32 bad_gen;        // this kernel (deferred verdict of the host-buffer calls): nothing is looked up, counts stay zero
    u32* work_count;    // may be null: cleared here for the kernels behind (k_reduce_replicas appends the exclusive pass's work)
};

// Geometry of k_stream_lookup: 512 threads = 8 waves per workgroup, two workgroups per CU (~75 KB of
// LDS each) = 4 waves per SIMD and up to 128 VGPRs (no scratch).  768 threads / 6 waves per SIMD
// (<= 80 VGPRs) spills with 16-block super-blocks and measured slower (0.113 vs 0.100 ms).
#ifndef YH_STREAM_THREADS
#define YH_STREAM_THREADS 512
#endif
#ifndef YH_STREAM_WAVES_PER_SIMD
#define YH_STREAM_WAVES_PER_SIMD 4
#endif
constexpr int STREAM_THREADS = YH_STREAM_THREADS;
// Candidates (a probe equal to an element's truncated key) are queued PER WAVE in LDS and confirmed
// by the wave that found them -- no workgroup barrier anywhere in the streaming loop.  Confirmation
// needs four independent random reads per candidate (the 64-bit hash and the reference of the stream
// position, the sample hash, the shared-hash index); they are SPLIT IN TWO PHASES: a flush requests
// them, one candidate per lane, into registers, and the NEXT flush, one super-block later, compares
// and counts.  By then the wave has waited for two groups of the stream requested after them (vmcnt
// is in order), so the reads cost no stall of their own; waiting for them in place was 16 us of the
// kernel's 98 (timing-only builds: probes without queueing 74, queueing 80, + reads 96, + counting 98).
constexpr u32 STREAM_WQ = 64;  // LDS queue entries per wave
struct WaveQ {
    u32* fill;     // LDS: entries claimed in this wave's queue
    u64x2* q;      // LDS: this wave's STREAM_WQ entries (x = stream position, y = sample index)
    u32 wg;
    u32* tkey;     // LDS: the workgroup's per-reference hit sums, STREAM_TSLOTS slots (reference + 1, 0 = empty)
    u32* tcnt;     // hits: low 16 bits of the slot's increments count all hits ...
    u32* tcnt2;    // ... and this one the hits on shared hashes
};
// Confirmed hits are summed per reference in an LDS table of the workgroup and leave as ONE global
// atomic per (workgroup, reference) when the workgroup ends: in hash order a reference's hits are
// spread evenly over the workgroups, and same-address atomics are serialized in L2.
#ifndef YH_STREAM_TPROBES
#define YH_STREAM_TPROBES 2   // (8 probes: a sample holding 2 000+ genomes overflows the table, and the failing CAS chains cost 7-15 %)
#endif
#ifndef YH_STREAM_TBITS
#define YH_STREAM_TBITS 9
#endif
constexpr u32 STREAM_TSLOTS = 1u << YH_STREAM_TBITS;
__device__ __forceinline__ void table_add(const StreamHit& hit, const WaveQ& c, u32 ref, bool shared) {
    u32 slot = (ref * 2654435761u) >> (32 - YH_STREAM_TBITS);
#pragma unroll 1
    for (int probe = 0; probe < YH_STREAM_TPROBES; ++probe, slot = (slot + 1) & (STREAM_TSLOTS - 1)) {
        const u32 old = atomicCAS(&c.tkey[slot], 0u, ref + 1);
        if (old == 0 || old == ref + 1) {
            atomicAdd(&c.tcnt[slot], 1u);
            if (shared && hit.reps2) atomicAdd(&c.tcnt2[slot], 1u);
            return;
        }
    }
    const u64 at = (u64)replica_of(c.wg, hit.rep_mask) * hit.n_refs + ref;  // crowded table: count directly
    count_add(&hit.reps[at], 1u);
    if (shared && hit.reps2) count_add(&hit.reps2[at], 1u);
}
struct Pending {  // one requested confirmation per lane
    uint2 rec = make_uint2(0u, STREAM_NONE);
    u64 sv = 0;
};
#ifndef YH_ABLATE_STREAM
#define YH_ABLATE_STREAM 0  // traffic-attribution builds (results wrong): 1 = candidates are not confirmed (no srec / sample read), 2 = the probes do not read their lane's delta bytes again
#endif
__device__ __forceinline__ void pending_request(const StreamHit& hit, Pending& p, u64 pos, u32 sidx) {
    if (YH_ABLATE_STREAM & 1) { p.rec = make_uint2((u32)pos, STREAM_NONE); p.sv = sidx; return; }
    p.rec = hit.srec[pos];
    p.sv = hit.sample[sidx];
}
__device__ __forceinline__ void pending_count(const StreamHit& hit, const WaveQ& c, Pending& p) {
    // the candidate's key equals the element's (bits sshift.. of the hash, sshift <= 32): the low words decide
    if (p.rec.x == (u32)p.sv && p.rec.y != STREAM_NONE) {  // (fillers have no reference)
        const bool shared = (p.rec.y & 0x80000000u) != 0;
        if (hit.hitflag && shared) {  // which shared hash: a search (general exclusive path only; shared hits are few)
            u32 lo = 0, hi = hit.n_gsh;
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (hit.gsh[mid] < p.sv) lo = mid + 1; else hi = mid;
            }
            hit.hitflag[lo] = 1;
        }
        table_add(hit, c, p.rec.y & 0x7fffffffu, shared);
    }
    p.rec.y = STREAM_NONE;
}
__device__ __forceinline__ void push_hit(const StreamHit& hit, const WaveQ& c, u64 pos, u32 sidx) {
    const u32 slot = atomicAdd(c.fill, 1u);
    if (slot < STREAM_WQ) {
        u64x2 e;
        e.x = pos;
        e.y = sidx;
        c.q[slot] = e;
    } else {  // queue full (a round with more than STREAM_WQ matches): confirm in place
        Pending p;
        pending_request(hit, p, pos, sidx);
        pending_count(hit, c, p);
    }
}
// wave-uniform call, once per round of probes: count the previous batch, request this one
__device__ __forceinline__ void wave_flush(const StreamHit& hit, const WaveQ& c, Pending& pend) {
    const u32 lane = threadIdx.x & 63u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 f = min((u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c.fill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)), STREAM_WQ);
    if (f == 0) return;  // (a batch in flight stays in flight)
    pending_count(hit, c, pend);
    if (lane < f) {
        const u64x2 x = c.q[lane];
        pending_request(hit, pend, x.x, (u32)x.y);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(c.fill, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// wg_key[w] = t of the first element of workgroup w's block range (~0 past the end): sample-independent
__global__ void k_wg_key(const u64* __restrict__ hdr, u64 nblk, u32 wgs, u64* __restrict__ wg_key) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > wgs) return;
    const u64 per = (nblk + wgs - 1) / wgs;
    wg_key[w] = hdr[min((u64)w * per, nblk)];
}

// A workgroup's sample range [lo, hi) = the sample hashes with wg_key[w] <= t <= wg_key[w + 1]
// (inclusive on both sides: equal keys may sit on either side of a range boundary).  One WAVE per
// bound, 64-ary search: 3 dependent rounds of 64 parallel probes + a final one instead of 20
// dependent reads.  Lower bound: first t >= key; upper bound: first t > key.  Wave-uniform result.
__device__ __forceinline__ u32 wave_bound(const u64* __restrict__ sample, u32 n, u32 sshift, u64 key, bool upper) {
    const u32 lane = threadIdx.x & 63u;
    auto pred = [&](u32 i) { const u64 v = sample[i] >> sshift; return upper ? (v <= key) : (v < key); };
    u32 lo = 0, hi = n;  // the answer (first index whose pred is false) lies in [lo, hi]
    while (hi - lo > 64) {
        const u32 step = (hi - lo + 63) / 64;
        const u64 q = (u64)lo + (u64)(lane + 1) * step - 1;  // probe positions, ascending
        const bool p = q < hi && pred((u32)q);
        const u32 cnt = (u32)__popcll(__ballot(p));  // pred is monotone: the true lanes are 0..cnt-1
        const u64 nlo = (u64)lo + (u64)cnt * step;
        const u64 nhi = (u64)lo + (u64)(cnt + 1) * step - 1;  // pred is false there (or it is past hi)
        hi = (u32)min((u64)hi, nhi);
        lo = (u32)min(nlo, (u64)hi);
    }
    const u32 i = lo + lane;
    const bool p = i < hi && pred(i);
    return lo + (u32)__popcll(__ballot(p));
}

// A wave takes the stream in super-blocks of STREAM_NB = STREAM_PF x STREAM_GROUPS = 16 consecutive
// blocks (16 KB of delta bytes).  Per super-block:
//   1. four groups of 4 blocks through two register sets of 16-byte vectors (groups 0 and 1 requested
//      together, group g + 2 when group g's registers have been summed): every block's lane spans
//      -- byte sums (v_sad_u8) + the group's wave scans side by side (DPP) -- written to LDS (INC),
//      the blocks' first and last keys to HB / scalar registers;
//   2. the sample keys inside the super-block's key range, ONE PER LANE (~49 of 64 lanes busy for a
//      10^6-hash sample against 3.3 x 10^8 hashes): which block (scalar compares), which lane of it
//      (6-step binary search in INC), that lane's 16 delta bytes again (a 16-byte read that hits L2)
//      and a 16-step compare.
// Nothing of the NEXT super-block is requested before this one is probed, so the probes' L2 reads
// never queue behind a prefetch (vmcnt is in order).  89 VGPRs.  The kernel is bound by the latency
// of this chain at 4 waves per SIMD, not by instruction issue and not by HBM.
#ifndef YH_STREAM_PF
#define YH_STREAM_PF 4
#endif
#ifndef YH_STREAM_GROUPS
#define YH_STREAM_GROUPS 4
#endif

// words per block row of INC: 65, not 64 -- the lanes of a probe round sit on different blocks f at
// the same search step, i.e. at the same column: with rows of 64 words all of them hit ONE bank
// (PMC round 1: 39 % of the LDS cycles were conflict cycles); an odd stride spreads them
#ifndef YH_INC_STRIDE
#define YH_INC_STRIDE 65
#endif
constexpr int INC_STRIDE = YH_INC_STRIDE;
constexpr int STREAM_PF = YH_STREAM_PF;
constexpr int STREAM_NB = YH_STREAM_PF * YH_STREAM_GROUPS;
static_assert(STREAM_PF >= 1 && STREAM_PF <= 15, "headers of a group live in lanes 0..PF");
static_assert(STREAM_NB <= 16, "HB holds 16 block keys per wave");
static_assert(YH_STREAM_GROUPS == 4, "stream_blocks is written out for four groups over two register sets");

__device__ __forceinline__ u64 uniform_u64(u64 v) {  // a value every lane holds, moved to scalar registers
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, l);
}

// loads of the STREAM_PF blocks from block bf on (clamped into [.., bl1): harmless re-reads at the
// end); lanes 0..PF of h hold the PF + 1 block headers hdr[bf .. bf + PF]
__device__ __forceinline__ void load_group(const u32x4* __restrict__ deltas, const u64* __restrict__ hdr, u64 bl1, u64 bf,
                                           u32x4 (&d)[STREAM_PF], u64& h) {
    const u32 lane = threadIdx.x & 63u;
#pragma unroll
    for (int i = 0; i < STREAM_PF; ++i) d[i] = deltas[min(bf + i, bl1 - 1) * 64 + lane];
    h = hdr[min(bf + min((u64)lane, (u64)STREAM_PF), bl1)];
}

// S: the tile's sample keys relative to Klo (32 bits, ~0 = sentinel); cur / hcur: the wave's first
// group, already requested by the caller when `preloaded`.  nb = blocks per super-block in this call
// (STREAM_NB, or fewer whole groups when the range is short).
__device__ __forceinline__ void stream_blocks(const u32x4* __restrict__ deltas, const u64* __restrict__ hdr, u64 bl0,
                                              u64 bl1, u32 sub, u32 n, u64 Klo, u64 Khi, u32 dsh, const u32* S,
                                              const u16* E, u32* INCw, u32* HBw, const StreamHit& hit,
                                              const WaveQ& ctx, Pending& pend, bool preloaded, u32 nb, u32x4 (&bufA)[STREAM_PF], u64& hdrA,
                                              u32x4 (&bufB)[STREAM_PF], u64& hdrB) {
    constexpr u32 WAVES = STREAM_THREADS / 64;
    constexpr int PF = STREAM_PF, NB = STREAM_NB;
    const u32 lane = threadIdx.x & 63u;
    const u64 n_super = (bl1 - bl0 + nb - 1) / nb;
    const u32 wvi = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and the compiler knows it
    u64 sb = wvi;
    for (; sb < n_super; sb += WAVES) {
        const u64 b0 = bl0 + sb * nb;
        const int nvalid = (int)min((u64)nb, bl1 - b0);
        u32 hl[NB];  // last key of block i relative to base0 (scalar); invalid blocks: ~0
        u64 base0 = 0;
        u32 last_rel = 0;
        // Four groups of PF blocks through TWO register sets: groups 0 and 1 are requested together,
        // group g + 2 as soon as group g's registers have been summed, so that a request is in flight
        // while the previous group's spans are computed (a super-block costs ~one memory latency, not
        // one per group).  Nothing of the NEXT super-block is requested before this one is probed: the
        // probes' own reads must not queue behind it (vmcnt is in order).
        auto issue = [&](u32x4 (&d)[PF], u64& h, int g) {
            if (g * PF < nvalid) load_group(deltas, hdr, bl1, b0 + g * PF, d, h);
        };
        auto spans = [&](const u32x4 (&d)[PF], const u64 h, int g) {
            const int nv = min(PF, nvalid - g * PF);  // valid blocks of this group
            if (nv > 0) {
                if (g == 0) base0 = readlane_u64(h, 0);
                if (lane < (u32)PF) HBw[g * PF + lane] = ((int)lane < nv) ? (u32)(h - base0) : 0xffffffffu;
                // Lane sums of the PF blocks (v_sad_u8), then the PF inclusive wave scans step by step side
                // by side (DPP only -- gfx9: row_shr 1/2/4/8 inside the rows of 16, then row_bcast:15 into
                // rows 1 and 3, row_bcast:31 into rows 2 and 3): a DPP add needs two wait states after the
                // write of its source, which the other blocks' adds fill.
                int v[PF];
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const u32 w0 = lane ? d[i].x : (d[i].x & 0xffffff00u);  // (a block's first delta byte is not used)
                    v[i] = (int)(__builtin_amdgcn_sad_u8(w0, 0u, 0u) + __builtin_amdgcn_sad_u8(d[i].y, 0u, 0u) +
                                 __builtin_amdgcn_sad_u8(d[i].z, 0u, 0u) + __builtin_amdgcn_sad_u8(d[i].w, 0u, 0u));
                }
#define YH_SCAN_STEP(ctrl, rmask)                                                                      \
    _Pragma("unroll") for (int i = 0; i < PF; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], ctrl, rmask, 0xf, false);
                YH_SCAN_STEP(0x111, 0xf)
                YH_SCAN_STEP(0x112, 0xf)
                YH_SCAN_STEP(0x114, 0xf)
                YH_SCAN_STEP(0x118, 0xf)
                YH_SCAN_STEP(0x142, 0xa)
                YH_SCAN_STEP(0x143, 0xc)
#undef YH_SCAN_STEP
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    INCw[(g * PF + i) * INC_STRIDE + lane] = (u32)v[i];
                    const u32 end_rel = (u32)(readlane_u64(h, i) - base0) + (u32)__builtin_amdgcn_readlane(v[i], 63);
                    hl[g * PF + i] = (i < nv) ? end_rel : 0xffffffffu;
                    last_rel = (i < nv) ? end_rel : last_rel;
                }
            } else {
                if (lane < (u32)PF) HBw[g * PF + lane] = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < PF; ++i) hl[g * PF + i] = 0xffffffffu;
            }
        };
        if (!preloaded) {
            issue(bufA, hdrA, 0);
            issue(bufB, hdrB, 1);
        }
        spans(bufA, hdrA, 0);
        issue(bufA, hdrA, 2);
        spans(bufB, hdrB, 1);
        issue(bufB, hdrB, 3);
        spans(bufA, hdrA, 2);
        spans(bufB, hdrB, 3);
        preloaded = false;
        const u64 sb_last = base0 + last_rel;
        if (base0 <= Khi && sb_last >= Klo)
        {  // some sample key of the tile can lie inside
            u32 k = (base0 <= Klo) ? 0u : (u32)E[(base0 - Klo) >> dsh];  // a slot at or before the first key >= base0
            k = (u32)__builtin_amdgcn_readfirstlane((int)k);
            for (;;) {  // 64 sample slots at a time; the keys inside the super-block are a run of lanes
                const u32 s32 = S[k + lane];  // (ST_PAD sentinels ~0 behind the last key keep this in bounds)
                const u64 sk = (s32 == 0xffffffffu) ? ~0ull : Klo + s32;
                const u64 ge = __ballot(sk >= base0);
                if (!ge) { k += 64; continue; }
                const bool in = sk >= base0 && sk <= sb_last && k + lane < n;
                const u32 c = (u32)__builtin_ctzll(ge);
                const u32 np = (u32)__popcll(__ballot(in));
                // One probe per lane: its block f and lane t are found in LDS (HB, INC); lane t's 16 delta
                // bytes are read again (they hit L2: the wave has just streamed them).
                const u32 rel = (u32)(sk - base0);
                u32 f = 0;  // first block whose last key is >= the probe
#pragma unroll
                for (int i = 0; i < NB; ++i) f += (hl[i] < rel) ? 1u : 0u;
                bool act = in && f < (u32)NB;
                u32 r = 0, t = 0;
                auto locate = [&]() {  // r, t of the probe inside block f; a block that starts behind the probe ends it
                    const u32 hb = HBw[f];
                    if (hb > rel) { act = false; return; }
                    r = rel - hb;
                    const u32* inc = INCw + f * INC_STRIDE;
                    t = 0;  // first lane whose last key is >= r  (inc[63] = the block's last key >= r)
#pragma unroll
                    for (int st = 32; st >= 1; st >>= 1) t += (inc[t + st - 1] < r) ? (u32)st : 0u;
                };
                if (act) locate();
                while (__ballot(act)) {  // (more than one trip only for runs of equal keys that leave a lane)
                    if (act) {
                        const u32x4 w = (YH_ABLATE_STREAM & 2) ? u32x4{r, t, f, 1u} : deltas[(b0 + f) * 64 + t];
                        const u32 W[4] = {t ? w.x : (w.x & 0xffffff00u), w.y, w.z, w.w};
                        u32 cs = t ? INCw[f * INC_STRIDE + t - 1] : 0u, match = 0;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            cs += (W[j >> 2] >> (8 * (j & 3))) & 0xffu;
                            match |= (cs == r ? 1u : 0u) << j;
                        }
                        while (match) {
                            const u32 j = (u32)__ffs((int)match) - 1u;
                            match &= match - 1u;
                            push_hit(hit, ctx, ((b0 + f) << 10) + 16u * t + j, sub + k + lane);
                        }
                        if (cs != r) act = false;          // the lane's last key is above the probe: the run ended
                        else if (t < 63) ++t;              // the run of equal keys may go on in the next lane
                        else if (++f < (u32)NB) locate();  // ... or in the next block
                        else act = false;                  // ... or in the next super-block, which finds it itself
                    }
                }
                wave_flush(hit, ctx, pend);
                if (c + np < 64) break;  // the run of inside keys ended within these 64 slots
                k += 64;
            }
        }
    }
}

constexpr int ST_LGNB = 11;             // buckets of the tile's directory E
constexpr int ST_NB = 1 << ST_LGNB;

__global__ void __launch_bounds__(STREAM_THREADS, YH_STREAM_WAVES_PER_SIMD)
k_stream_lookup(const u32x4* __restrict__ deltas, const u64* __restrict__ hdr, u64 nblk,
                const u64* __restrict__ sample, u32 n_sample, const u64* __restrict__ wg_key, u32 sshift,
                StreamHit hit) {
    constexpr u32 WAVES = STREAM_THREADS / 64;
    __shared__ u32 S[ST_SLOTS];                 // the tile's sample keys, relative to its first one
    __shared__ u16 E[ST_NB];
    __shared__ u64x2 Q[WAVES][STREAM_WQ];
    __shared__ u32 INC[WAVES][STREAM_NB * INC_STRIDE];  // per wave: the lane-inclusive key sums of a super-block
    __shared__ u32 HB[WAVES][16];               // per wave: first key of each of its blocks, relative
    __shared__ u32 q_fill[WAVES];
    __shared__ u32 sbound[2];
    __shared__ u32 n_fit;
    __shared__ u32 tkey[STREAM_TSLOTS];
    __shared__ u32 tcnt[STREAM_TSLOTS];
    __shared__ u32 tcnt2[STREAM_TSLOTS];

    const u32 tid = threadIdx.x;
    if (hit.work_count && blockIdx.x == 0 && tid == 0) *hit.work_count = 0;
    if (hit.bad && *hit.bad == hit.bad_gen) return;  // (wave-uniform scalar load)
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const u32 lid = xcd_remap(blockIdx.x, gridDim.x);
    const u64 per = (nblk + gridDim.x - 1) / gridDim.x;
    const u64 B0 = min((u64)lid * per, nblk), B1 = min(nblk, B0 + per);
    const WaveQ ctx{&q_fill[wv], Q[wv], lid, tkey, tcnt, tcnt2};
    if (B0 >= B1) return;  // no blocks
    // The wave's first group is requested before anything else; while it is in flight, waves 0 and 1
    // find the workgroup's range of the sample and the tile is staged.  (With several tiles -- a
    // sample slice above ST_CAP hashes -- the request is wasted and made again per tile.)
    u32x4 bufA[STREAM_PF], bufB[STREAM_PF];
    u64 hdrA = 0, hdrB = 0;
    const u64 first_sb = wv;
    if (first_sb * STREAM_NB < B1 - B0) load_group(deltas, hdr, B1, B0 + first_sb * STREAM_NB, bufA, hdrA);
    if (first_sb * STREAM_NB + STREAM_PF < B1 - B0) load_group(deltas, hdr, B1, B0 + first_sb * STREAM_NB + STREAM_PF, bufB, hdrB);
    if (wv < 2) {
        const u32 bnd = wave_bound(sample, n_sample, sshift, wg_key[lid + wv], wv == 1);
        if ((tid & 63u) == 0) sbound[wv] = bnd;
    }
    if (tid < WAVES) q_fill[tid] = 0;
    for (u32 k = tid; k < STREAM_TSLOTS; k += STREAM_THREADS) { tkey[k] = 0; tcnt[k] = 0; tcnt2[k] = 0; }
    __syncthreads();
    const u32 s0 = (u32)__builtin_amdgcn_readfirstlane((int)sbound[0]), s1 = (u32)__builtin_amdgcn_readfirstlane((int)sbound[1]);
    if (s0 >= s1) return;  // no sample hash in this range of t: nothing to look up
    Pending pend;
    for (u32 sub = s0; sub < s1;) {
        u32 n = min((u32)ST_CAP, s1 - sub);
        if (sub != s0) __syncthreads();  // every wave is done with the previous tile
        // keys are staged relative to the tile's first one, in 32 bits; a tile ends early where that overflows
        const u64 Klo = uniform_u64(sample[sub] >> sshift);
        if (tid == 0) n_fit = n;
        __syncthreads();
        for (u32 k = tid; k < n; k += STREAM_THREADS) {
            const u64 d = (sample[sub + k] >> sshift) - Klo;
            if (d > 0xfffffffeull) atomicMin(&n_fit, k);
            else S[k] = (u32)d;
        }
        __syncthreads();
        n = (u32)__builtin_amdgcn_readfirstlane((int)n_fit);  // >= 1: key 0 is Klo itself
        for (u32 k = n + tid; k < n + ST_PAD; k += STREAM_THREADS) S[k] = 0xffffffffu;
        __syncthreads();
        const u32 span = (u32)__builtin_amdgcn_readfirstlane((int)S[n - 1]);
        const u64 Khi = Klo + span;
        const u32 dsh = (span >> ST_LGNB) ? (u32)(32 - __builtin_clz(span)) - ST_LGNB : 0u;  // (span >> dsh) < ST_NB
        for (u32 k = tid; k < n; k += STREAM_THREADS) {
            const u32 bk = S[k] >> dsh;
            const int bp = (k == 0) ? -1 : (int)(S[k - 1] >> dsh);
            for (int x = bp + 1; x <= (int)bk; ++x) E[x] = (u16)k;
            if (k == n - 1)
                for (u32 x = bk + 1; x < (u32)ST_NB; ++x) E[x] = (u16)n;
        }
        __syncthreads();
        const bool single = sub == s0 && n == s1 - s0;  // one tile (the usual case): the first request was right
        u64 bl0 = B0, bl1 = B1;
        if (!single) {  // several tiles: each covers a contiguous sub-range of the blocks
            u64 lo = B0, hi = B1;     // first block whose NEXT header is >= Klo
            while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (hdr[mid + 1] < Klo) lo = mid + 1; else hi = mid; }
            bl0 = lo;
            lo = bl0; hi = B1;        // first block whose header is > Khi
            while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (hdr[mid] <= Khi) lo = mid + 1; else hi = mid; }
            bl1 = lo;
        }
        // full super-blocks when every wave gets at least two of them, single groups otherwise (a tile of
        // a multi-tile sample covers few blocks: with 16-block super-blocks half the waves would idle)
        // ... and fewer blocks still when the range is so short that whole groups would leave waves idle
        // (a small database against a big sample: the probes, not the stream, are the work to spread)
        const u32 nb = (bl1 - bl0 >= 2ull * WAVES * STREAM_NB) ? (u32)STREAM_NB
                       : (u32)min((u64)STREAM_PF, max((bl1 - bl0) / WAVES, (u64)1));
        if (bl0 < bl1)
            stream_blocks(deltas, hdr, bl0, bl1, sub, n, Klo, Khi, dsh, S, E, INC[wv], HB[wv], hit, ctx, pend,
                          single && nb == (u32)STREAM_NB && first_sb * STREAM_NB < B1 - B0, nb, bufA, hdrA, bufB, hdrB);
        sub += n;
    }
    wave_flush(hit, ctx, pend);
    pending_count(hit, ctx, pend);  // the last batch: the one wait for confirmation reads the wave cannot hide
    __syncthreads();
    const u64 my = (u64)replica_of(lid, hit.rep_mask) * hit.n_refs;
    for (u32 k = tid; k < STREAM_TSLOTS; k += STREAM_THREADS)
        if (tkey[k]) {
            count_add(&hit.reps[my + tkey[k] - 1], tcnt[k]);
            if (hit.reps2 && tcnt2[k]) count_add(&hit.reps2[my + tkey[k] - 1], tcnt2[k]);
        }
}

// ---- cross-check kernel: one wave per reference over the plain CSR -------------------------------
__global__ void __launch_bounds__(256) k_overlap_bsearch(const u64* __restrict__ values,
                                                         const u64* __restrict__ offsets, u64 n_refs,
                                                         const u64* __restrict__ sample, u64 n_sample,
                                                         u32* __restrict__ overlap) {
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) / WAVE;
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    for (u64 j = wave; j < n_refs; j += n_waves) {
        const u64 b = offsets[j], e = offsets[j + 1];
        u32 total = 0;
        for (u64 k0 = b; k0 < e; k0 += WAVE) {
            const u64 k = k0 + lane;
            bool hit = false;
            if (k < e) {
                const u64 h = values[k];
                u64 lo = 0, hi = n_sample;
                while (lo < hi) {
                    const u64 mid = (lo + hi) >> 1;
                    if (sample[mid] < h) lo = mid + 1; else hi = mid;
                }
                hit = (lo < n_sample) && (sample[lo] == h);
            }
            total += (u32)__popcll(__ballot(hit));
        }
        if (lane == 0) overlap[j] = total;
    }
}

// ---- sample-driven overlap through the distinct-hash directory (YH_DB_FULL_INDEX) ----------------
// One lane per SAMPLE hash: directory bucket -> a short ascending scan of the distinct hashes ->
// its holder (or the posting list of a shared hash) -> replicated counters.  Work is proportional
// to |S|, not to the database: ~3 dependent memory round trips per sample hash instead of
// streaming every reference hash.  Also flags the shared hashes found (hit[], for R2).
// The posting list of a shared hash found in the sample: every holder counts one hit.  The holders are requested
// four at a time (a list of 8 was 8 dependent round trips: the tail of the launch for a sample of cluster members).
template <typename Add>
__device__ __forceinline__ void walk_holders(const u64* __restrict__ po, const u32* __restrict__ pr, u32 gi, Add add) {
    const u64 q0 = po[gi], qe = po[gi + 1];
    for (u64 q = q0; q < qe; q += 4) {
        u32 h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = pr[min(q + (u64)i, qe - 1)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q + (u64)i < qe) add(h[i]);
    }
}

// The same lookup for LARGE samples: a workgroup of IDX_THREADS lanes takes a tile of IDX_THREADS x U consecutive
// sample hashes, every lane has its U bucket reads in flight together, and the hits are summed per reference in an
// LDS table that leaves as ONE global atomic per (workgroup, reference) at the end.  In hash order the hits of a
// genome that is really in the sample (thousands of them) are spread evenly over the workgroups, and same-address
// atomics are serialized memory-side: with one atomic per hit the hottest counter is the tail of the launch (one
// 10^6-hash sample repeated, so that its buckets stay in the Infinity Cache: 1.9e5 hits on 284 references 35.1 us, no
// hit at all 25.1 us, tiles 29.8 us; rotating samples, buckets from HBM: 38-40 -> 36 us, no hit 32).
#ifndef YH_IDX_THREADS
#define YH_IDX_THREADS 1024
#endif
#ifndef YH_IDX_TBITS
#define YH_IDX_TBITS 10
#endif
constexpr int IDX_THREADS = YH_IDX_THREADS;
// (THREADS lanes x U hashes per workgroup, 2^TBITS slots in the hit table: <2, 1024, 10> for large samples; <1, 256, 8>
// for small ones, where the table keeps a sample that is mostly ONE genome -- an isolate -- from sending thousands of
// atomics to one counter)
struct TileLookup {
    const u64* sample;
    u64 n;
    YhDirView dv;
    const u32* filter;
    u64 filter_mul;
    const u64* po;
    const u32* pr;
    u32* reps;
    u32 rep_mask;
    u64 n_refs;
    u8* hit;
    u32* reps2;
    u32* work_count;
    const u32* bad;
    u32 bad_gen;
};
// (body: `wg` = the workgroup's index among the looking-up ones; tkey / tcnt / tcnt2: 2^TBITS words of LDS each)
template <int U, int THREADS, int TBITS>
__device__ __forceinline__ void lookup_tile_body(u32 wg, u32* tkey, u32* tcnt, u32* tcnt2, const TileLookup& q) {
    constexpr u32 TSLOTS = 1u << TBITS;
    const u64* __restrict__ sample = q.sample;
    const u64 n = q.n;
    const YhDirView& dv = q.dv;
    const u32* __restrict__ filter = q.filter;
    if (q.work_count && wg == 0 && threadIdx.x == 0) *q.work_count = 0;
    u32* my = q.reps + (u64)replica_of(wg, q.rep_mask) * q.n_refs;
    u32* my2 = q.reps2 ? q.reps2 + (u64)replica_of(wg, q.rep_mask) * q.n_refs : nullptr;
    const u64 base = wg * (u64)(THREADS * U);
    u64 h[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const u64 t = base + (u64)u * THREADS + threadIdx.x;
        h[u] = sample[min(t, n - 1)];
        ok[u] = t < n && h[u] <= dv.max_hash;
        if (!ok[u]) h[u] = 0;  // (still a valid bucket to read)
    }
#if !(defined(YH_ABLATE_LOOKUP) && (YH_ABLATE_LOOKUP & 32))
    // (the verdict of the sample's ordering check is asked for BEHIND the sample's own loads: in front of them every workgroup began
    // with a dependent round trip -- profiles/r05/ablate_step.txt: 0.2 us of the step)
    if (q.bad && *q.bad == q.bad_gen) return;
#endif
    for (u32 k = threadIdx.x; k < TSLOTS; k += THREADS) { tkey[k] = 0; tcnt[k] = 0; tcnt2[k] = 0; }
    YhDirView::v4u a[U], b[U], c[U], d[U];
    u32 r[U];
#if defined(YH_ABLATE_LOOKUP) && (YH_ABLATE_LOOKUP & 16)  // timing-only build: no presence filter read
    if (false) {
#else
    if (filter) {  // the presence bits first: a hash whose bit is clear is not in the database (yh_db::d_filter)
#endif
        u64 bit[U];
        u32 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bit[u] = yh_bucket_of(h[u], dv.bkt_lsh, q.filter_mul);
            w[u] = filter[bit[u] >> 5];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32 m = yh_filter_mask(h[u], bit[u]);
            ok[u] = ok[u] && (w[u] & m) == m;
        }
    }
    if (dv.cbkt) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = b[u] = c[u] = d[u] = YhDirView::v4u{0u, 0u, 0u, 0u};
#if defined(YH_ABLATE_LOOKUP) && (YH_ABLATE_LOOKUP & 8)  // timing-only build: no bucket is read (every hash "absent" behind the filter)
            ok[u] = ok[u] && h[u] == 0x123456789abcdefull;
#endif
            if (ok[u]) dv.cbkt_request(h[u], a[u], b[u], c[u], d[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(a[u]), "+v"(b[u]), "+v"(c[u]), "+v"(d[u]));  // (see YhDirView::find)
    }
    __syncthreads();  // the table is clear
    auto add = [&](u32 ref, bool shared) {
#if defined(YH_ABLATE_LOOKUP) && (YH_ABLATE_LOOKUP & 4)  // timing-only build: hits are not counted
        if (ref == 0x7ffffff1u) my[0] = 1;
        return;
#endif
        u32 slot = (ref * 2654435761u) >> (32 - TBITS);
#pragma unroll 1
        for (int probe = 0; probe < 2; ++probe, slot = (slot + 1) & (TSLOTS - 1)) {
            const u32 old = atomicCAS(&tkey[slot], 0u, ref + 1);
            if (old == 0 || old == ref + 1) {
                atomicAdd(&tcnt[slot], 1u);
                if (shared && my2) atomicAdd(&tcnt2[slot], 1u);
                return;
            }
        }
        count_add(&my[ref], 1u);  // crowded table: count directly
        if (shared && my2) count_add(&my2[ref], 1u);
    };
#pragma unroll
    for (int u = 0; u < U; ++u) {
        r[u] = YH_DIR_NONE;
        if (ok[u]) r[u] = dv.cbkt ? dv.cbkt_resolve(h[u], a[u], b[u], c[u], d[u]) : dv.find(h[u]);
        if (r[u] == YH_DIR_NONE) continue;
        if (!(r[u] & 0x80000000u)) {
            add(r[u], false);
        } else {
            const u32 gi = r[u] & 0x7fffffffu;
            if (q.hit) q.hit[gi] = 1;
#if !(defined(YH_ABLATE_LOOKUP) && (YH_ABLATE_LOOKUP & 1))  // timing-only builds (build.py build_variant): results are wrong
            walk_holders(q.po, q.pr, gi, [&](u32 holder) { add(holder, true); });
#endif
        }
    }
    __syncthreads();
#if defined(YH_ABLATE_LOOKUP) && (YH_ABLATE_LOOKUP & 2)
    if (n == 1) return;  // (never true: keeps the table alive)
    return;
#endif
    for (u32 k = threadIdx.x; k < TSLOTS; k += THREADS)
        if (tkey[k]) {
            count_add(&my[tkey[k] - 1], tcnt[k]);
            if (my2 && tcnt2[k]) count_add(&my2[tkey[k] - 1], tcnt2[k]);
        }
}

template <int U, int THREADS, int TBITS>
__global__ void __launch_bounds__(THREADS) k_index_lookup_tile(const TileLookup q) {
    constexpr u32 TSLOTS = 1u << TBITS;
    __shared__ u32 tkey[TSLOTS];   // reference + 1, 0 = empty
    __shared__ u32 tcnt[TSLOTS];   // hits
    __shared__ u32 tcnt2[TSLOTS];  // hits on shared hashes
    lookup_tile_body<U, THREADS, TBITS>(blockIdx.x, tkey, tcnt, tcnt2, q);
}

// ---- exclusive counts -----------------------------------------------------------------------------
// The same sums without the pass over pr[]: the reference-major view of the postings is cut into
// PIECES of <= EXCL_PIECE postings (d_chunks: (reference, first posting); most references are one
// piece), stored so that neighbouring records belong to unrelated references.  ONE WAVE PER PIECE: the
// wave tests its reference's subset bit and, if set, sweeps the piece 64 x EXCL_U postings a step, every
// lane summing for itself; one wave reduction and ONE atomic per sum and piece at the end.
// (Earlier form: one lane per 64-posting chunk record, a ballot and an atomic per chunk.  With 29 % of
// the references in the subset -- the hit shape of real runs -- 73 of its 80 us were those atomics:
// the ~30 chunks of a reference, and of its cluster mates next to it in the SAME 64-byte line, are
// device-scope atomics that the memory side serializes per line.  bench.py "real_shape".)
constexpr u32 EXCL_PIECE = YH_EXCL_PIECE;
#ifndef YH_EXCL_U
#define YH_EXCL_U 4
#endif
constexpr int EXCL_U = YH_EXCL_U;
constexpr u32 EXCL_LDS_WORDS = 12288;  // 48 KiB of subset bits = 393 216 references

__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += (u32)__shfl_xor((int)v, off);
    return v;
}

// Which holders of a posting's hash are in the subset, r itself excluded.  Through rrec / rrecx when the
// handle has them (up to seven other holders sit beside the posting: one coalesced read), else -- and
// for nine holders and more -- by walking the posting list, eight holders a step.
template <class MaskWord>
__device__ __forceinline__ u32 others_in_subset(const uint4 rec, const uint4 recx, const u32* __restrict__ pr,
                                                const MaskWord& mword) {
    if (rec.w != 0xffffffffu) {
        const u32 hid[7] = {rec.x, rec.y, rec.z, recx.x, recx.y, recx.z, recx.w};
        u32 others = 0;
#pragma unroll
        for (int t = 0; t < 7; ++t) others += ((u32)t < rec.w) ? ((mword(hid[t] >> 5) >> (hid[t] & 31u)) & 1u) : 0u;
        return others;
    }
    u32 cnt = 0;
    const u32 q0 = rec.x, len = rec.y;
    for (u32 base = 0; base < len && cnt < 2; base += 8) {
        u32 h[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) h[t] = pr[q0 + min(base + (u32)t, len - 1)];
#pragma unroll
        for (int t = 0; t < 8; ++t) cnt += (base + (u32)t < len) ? ((mword(h[t] >> 5) >> (h[t] & 31u)) & 1u) : 0u;
    }
    return cnt - 1u;  // (r is in its own list and in the subset)
}

// General form (any subset): ex_e[r] += shared hashes of r with no other holder in the subset, ex_m[r] +=
// those of them found in the sample (hit[]), ovsh[r] += shared hashes of r found in the sample.
// rrec == nullptr (posting-only handles): holders through rg -> po -> pr.  hit == nullptr: ex_e only.
// mult != nullptr (the fused run step): the records are a reference's DISTINCT holder sets and mult[k] the number
// of its shared hashes with that set (yh_db::d_hrec); n_post is then the number of set records.
#ifndef YH_EXCL_GRID
#define YH_EXCL_GRID 2048
#endif
#ifndef YH_EXCL_RECX_ALWAYS
#define YH_EXCL_RECX_ALWAYS 0
#endif
constexpr int EXCL_PIECE_THREADS = YH_EXCL_PIECE_THREADS;  // waves of a workgroup share the staging of the subset bits
struct ExclPieces {
    const u32* work_count;
    const uint4* work;
    const u32* rpo;
    const u32* rg;
    const uint4* rrec;
    const uint4* rrecx;
    const u32* mult;
    const u64* po;
    u32 n_post;
    const u32* pr;
    const u32* maskbits;
    u32 n_mask_words;
    const u8* hit;
    u32* ex_e;
    u32* ex_m;
    u32* ovsh;
};
// (body: any workgroup size that is a multiple of 64; `wg` of `n_wgs` workgroups walk the work list; lmask: n_mask_words of LDS
// when LDSMASK)
// one work record (a piece of a reference's records) by one wave
template <bool LDSMASK, int U>
__device__ __forceinline__ void excl_one_piece(const uint4 mine, const u32* lmask, const ExclPieces& q) {
    const u32 lane = threadIdx.x & 63u;
    const u32* __restrict__ rg = q.rg;
    const uint4* __restrict__ rrec = q.rrec;
    const uint4* __restrict__ rrecx = q.rrecx;
    const u32* __restrict__ mult = q.mult;
    const u32* __restrict__ pr = q.pr;
    const u8* __restrict__ hit = q.hit;
    const u32 n_post = q.n_post;
    auto mword = [&](u32 i) -> u32 { return LDSMASK ? lmask[i] : q.maskbits[i]; };
    const u32 r = mine.x;
    const u32 end = mine.z;
    u32 acc_e = 0, acc_m = 0, acc_o = 0;
    for (u32 k0 = mine.y; k0 < end; k0 += 64u * U) {
        u32 k[U];
        bool valid[U], in_s[U];
        uint4 rec[U], recx[U];
        u32 gi[U], mu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            k[u] = k0 + 64u * u + lane;
            valid[u] = k[u] < end;
            const u32 kc = min(k[u], n_post - 1);  // (clamped: branch-free reads)
            gi[u] = (hit || !rrec) ? rg[kc] : 0u;
            mu[u] = mult ? mult[kc] : 1u;
            if (rrec) rec[u] = rrec[kc];
            recx[u] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (rrec) {  // holders 3..6: read only by the waves that have a posting with more than three others
            bool more = YH_EXCL_RECX_ALWAYS != 0;
#pragma unroll
            for (int u = 0; u < U; ++u) more |= valid[u] && rec[u].w > 3u && rec[u].w != 0xffffffffu;
            if (__ballot(more)) {
#pragma unroll
                for (int u = 0; u < U; ++u) recx[u] = rrecx[min(k[u], n_post - 1)];
            }
        } else {     // handles without the inline holder records: holders through the posting lists
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u64 q0 = q.po[gi[u]], q1 = q.po[gi[u] + 1];
                rec[u] = make_uint4((u32)q0, (u32)(q1 - q0), 0u, 0xffffffffu);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) in_s[u] = hit && valid[u] && hit[gi[u]] != 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool e = valid[u] && others_in_subset(rec[u], recx[u], pr, mword) == 0;
            acc_e += e ? mu[u] : 0u;
            acc_m += (e && in_s[u]) ? 1u : 0u;
            acc_o += in_s[u] ? 1u : 0u;
        }
    }
    acc_e = wave_sum(acc_e);
    if (hit) { acc_m = wave_sum(acc_m); acc_o = wave_sum(acc_o); }
    if (lane == 0) {
        if (acc_e) atomicAdd(&q.ex_e[r], acc_e);
        if (hit && acc_m) atomicAdd(&q.ex_m[r], acc_m);
        if (hit && acc_o) atomicAdd(&q.ovsh[r], acc_o);
    }
}
__device__ __forceinline__ void stage_subset_bits(u32* lmask, const ExclPieces& q) {  // (16-byte reads; n_mask_words is a multiple of 8)
    const uint4* src = reinterpret_cast<const uint4*>(q.maskbits);
    uint4* dst = reinterpret_cast<uint4*>(lmask);
    for (u32 i = threadIdx.x; i < q.n_mask_words / 4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// (body: any workgroup size that is a multiple of 64; `wg` of `n_wgs` workgroups walk the work list, a fixed share each;
// lmask: n_mask_words of LDS when LDSMASK)
template <bool LDSMASK, int U = EXCL_U>
__device__ __forceinline__ void excl_pieces_body(u32 wg, u32 n_wgs, u32* lmask, const ExclPieces& q) {
    const u32 WPB = blockDim.x >> 6;
    const u32 n_work = *q.work_count;
    if (wg * WPB >= n_work) return;  // (the grid is sized for every piece of the database)
    // this wave's first record is requested before the subset bits are staged: one round trip for both
    const u32 w0 = wg * WPB + (threadIdx.x >> 6);
    uint4 first_rec = make_uint4(0u, 0u, 0u, 0u);
    if (w0 < n_work) first_rec = q.work[w0];
    if (LDSMASK) stage_subset_bits(lmask, q);
    for (u32 w = w0; w < n_work; w += n_wgs * WPB)
        excl_one_piece<LDSMASK, U>((w == w0) ? first_rec : q.work[w], lmask, q);
}

template <bool LDSMASK>
__global__ void __launch_bounds__(EXCL_PIECE_THREADS) k_excl_pieces(const ExclPieces q) {
    extern __shared__ u32 lmask[];
    excl_pieces_body<LDSMASK>(blockIdx.x, gridDim.x, lmask, q);
}

// ---- the whole step as ONE launch of three independent roles (yh_run_device_pipelined) ---------------------------
// reduce + exclusive pass are ~11 us of launch and round-trip latency for almost no work.  Queued behind the lookup they
// are 25 % of the step; on a second stream they do not overlap the next lookup either (its 488 workgroups of 1024 lanes
// hold every wave slot of the chip until they end: measured slower than the plain step).  So the three stages of three
// CONSECUTIVE samples share one launch: workgroups [0, E) run the exclusive pass of sample k - 2, the next R the reducer
// of sample k - 1, the rest the lookup of sample k -- no workgroup depends on another of the same launch: consecutive
// samples alternate between two sets of replica counters and rotate through three step contexts (subset bits + work
// list).  The short roles come first in dispatch order, and a step costs one launch: the lookup's.
struct StepFused {
    TileLookup look;   u32 look_wgs;
    u32* r_reps; u32 r_R; u64 r_n; u32* r_out; u32* r_maskbits; FusedRun r_fused; u32 red_wgs;
    ExclPieces excl;   u32 excl_wgs; u32 excl_lds;  // excl_lds: the subset bits fit the launch's LDS
};
// (8 waves per SIMD = two workgroups per CU, as the stand-alone lookup has: 64 VGPRs; the exclusive role unrolls by 2.)
// Dispatch order = block order: the exclusive role's workgroups come first (most of them find nothing beyond the work
// list's end and are gone after one read), then the reducer's -- 4 references per thread, 21 workgroups for 85 205
// references: it fits the wave slots the 488 lookup workgroups of a 10^6-hash sample leave free -- then the lookup's.
// (Tried and dropped: no dedicated exclusive workgroups, every workgroup walking its share of the work list BEHIND its own
// job -- the pass then starts when the lookups end instead of beside them: 49.8 us per launch against 33; and claiming
// records from a shared cursor, which hung the device on a non-uniform early exit before a barrier.)
template <int U, int THREADS, int TBITS>
__global__ void __launch_bounds__(THREADS, THREADS == 1024 ? 8 : 4) k_step_fused(const StepFused s) {
    extern __shared__ u32 smem[];
    constexpr u32 TSLOTS = 1u << TBITS;
    u32 b = blockIdx.x;
#if defined(YH_ABLATE_LOOKUP) && (YH_ABLATE_LOOKUP & 64)  // timing-only build: the launch's exclusive and reducer roles do nothing
    if (b < s.excl_wgs + s.red_wgs) return;
#endif
    if (b < s.excl_wgs) {
        if (s.excl_lds) excl_pieces_body<true, 2>(b, s.excl_wgs, smem, s.excl);
        else excl_pieces_body<false, 2>(b, s.excl_wgs, smem, s.excl);
        return;
    }
    b -= s.excl_wgs;
    if (b < s.red_wgs) {
        const u32 blocks = (u32)((s.r_n + 4ull * THREADS - 1) / (4ull * THREADS));
        for (u32 blk = b; blk < blocks; blk += s.red_wgs) {
            reduce_replicas_multi<4>(blk, smem, s.r_reps, s.r_R, s.r_n, s.r_out, s.r_maskbits, s.r_fused);
            __syncthreads();  // (the scan words in LDS are re-used by the next block)
        }
        return;
    }
    b -= s.red_wgs;
    lookup_tile_body<U, THREADS, TBITS>(b, smem, smem + TSLOTS, smem + 2 * TSLOTS, s.look);
}

// e_j = (hashes of j that no other reference of the whole database has) + ex_e[j]
// m_j = (overlap_j - overlap restricted to database-shared hashes)      + ex_m[j]
__global__ void k_excl_final(u64 n, const u8* __restrict__ mask, const u32* __restrict__ sizes,
                             const u32* __restrict__ nshared, const u32* __restrict__ overlap,
                             const u32* __restrict__ ex_e, const u32* __restrict__ ex_m,
                             const u32* __restrict__ ovsh, u32* __restrict__ out_e, u32* __restrict__ out_m,
                             uint4* __restrict__ hit16, u64 n_hit16) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    // the shared-hash flags have been consumed by the kernels in front of this one: zero at rest
    for (u64 i = j; i < n_hit16; i += (u64)gridDim.x * blockDim.x) hit16[i] = make_uint4(0, 0, 0, 0);
    if (j >= n) return;
    u32 e = 0, m = 0;
    if (mask[j]) {
        e = sizes[j] - nshared[j] + ex_e[j];
        m = overlap[j] - ovsh[j] + ex_m[j];
    }
    out_e[j] = e;
    out_m[j] = m;
}

inline u32 grid_for(u64 work_items, u32 block, u32 max_blocks = 16384) {
    u64 g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (u32)g;
}

}  // namespace

// =================================================================================================
int yh_q_check_sorted_host(const u64* v, u64 n) {
    for (u64 i = 1; i < n; ++i)
        if (!(v[i - 1] < v[i])) return YH_ERR_UNSORTED;
    return YH_OK;
}

// replicated counters: R * N * 4 bytes, at most ~8 MiB; zero at rest (k_reduce_replicas clears them)
static int ensure_reps(yh_db* db, u32& R) {
    const u64 N = db->n_refs;
    R = 32;
    while (R > 1 && (u64)R * N > (2u << 20)) R >>= 1;
    if (db->reps_cap < (u64)R * N) {
        YH_HIP(hipStreamSynchronize(db->stream));
        if (db->d_reps) { yh_dfree(db, db->d_reps); db->d_reps = nullptr; db->reps_cap = 0; }
        // (two sets back to back: the second one counts hits on shared hashes in the fused run step; and all of
        // that twice: the pipelined step alternates between the two pairs, see yh_q_overlap_indexed)
        YH_HIP(hipMalloc((void**)&db->d_reps, 4 * (u64)R * N * sizeof(u32) + 16));
        YH_HIP(hipMemsetAsync(db->d_reps, 0, 4 * (u64)R * N * sizeof(u32) + 16, db->stream));
        db->reps_cap = (u64)R * N;
    }
    return YH_OK;
}
// the shared-hash flags are zero at rest too (k_excl_final clears them behind their last reader);
// a query that was not followed by its exclusive pass leaves them set, and the next one clears them
static int claim_hit_flags(yh_db* db) {
    if (!db->hit_clean && db->d_hit && db->n_shared)
        YH_HIP(hipMemsetAsync(db->d_hit, 0, db->n_shared + 15, db->stream));
    db->hit_clean = false;  // the kernels queued next set flags
    return YH_OK;
}

// hit == nullptr: only ex_e is summed (the fused run step).  One wave per work record; the work list
// (db->d_work, db->d_work_count) was appended by k_reduce_replicas / k_excl_worklist on the same stream.
static ExclPieces excl_args(yh_db* db, const u32* work_count, const uint4* work, const u32* d_maskbits, const u8* d_hit,
                            u32* d_ex_e, u32* d_ex_m, u32* d_ovsh, bool sets) {
    // sets: the work list holds pieces of holder-set records (appended by k_reduce_replicas), not of postings
    const u32 words = (u32)(((db->n_refs + 255) / 256) * 8);  // what k_reduce_replicas / k_mask_bits write: whole 256-reference blocks
    return ExclPieces{work_count, work, db->d_rpo, db->d_rg, sets ? db->d_hrec : db->d_rrec, sets ? db->d_hrecx : db->d_rrecx,
                      sets ? db->d_hmult : nullptr, db->d_po, sets ? db->n_sets : (u32)db->n_postings, db->d_pr, d_maskbits,
                      words, d_hit, d_ex_e, d_ex_m, d_ovsh};
}
static void launch_excl_pieces(yh_db* db, const u32* d_maskbits, const u8* d_hit, u32* d_ex_e, u32* d_ex_m, u32* d_ovsh,
                               bool sets = false) {
    const ExclPieces q = excl_args(db, db->d_work_count, db->d_work, d_maskbits, d_hit, d_ex_e, d_ex_m, d_ovsh, sets);
    const u32 per_wg = EXCL_PIECE_THREADS / 64;
    const u32 grid = std::min<u32>((db->n_chunks + per_wg - 1) / per_wg, (u32)YH_EXCL_GRID);
    if (q.n_mask_words <= EXCL_LDS_WORDS)
        k_excl_pieces<true><<<grid, EXCL_PIECE_THREADS, q.n_mask_words * sizeof(u32), db->stream>>>(q);
    else
        k_excl_pieces<false><<<grid, EXCL_PIECE_THREADS, 0, db->stream>>>(q);
}

// flag_shared: also flag which database-shared hashes are in the sample (db->d_hit), fused into the
// same launch; yh_q_exclusive_partial(..., hit_ready = true) then skips its own membership pass.
// overlap through the hash-sorted delta stream (the default layout)
// d_fused_excl / d_fused_match non-null: the whole `yacht run` step (subset = overlap > 0) in three
// launches -- no shared-hash flags, no exclusive accumulators, no finalize kernel (k_reduce_replicas).
static int yh_q_overlap_stream(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, bool flag_shared,
                               bool with_index, bool make_mask, u32* d_fused_excl = nullptr,
                               u32* d_fused_match = nullptr, u32* d_bits_out = nullptr) {
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    const u64 nblk = db->slen / STREAM_BLOCK;
    // (a workgroup's set-up is two wave searches and a few KB of LDS: two blocks are enough to pay for it)
    // 512 workgroups = the two resident ones of each of the 256 CUs, one round (YH_STREAM_WGS: tests force one)
    static const u32 wgs_env = [] { const char* e = yh_tune_env("YH_STREAM_WGS"); return (e && atoi(e) > 0) ? (u32)atoi(e) : 512u; }();
    u32 wgs = (u32)std::min<u64>(wgs_env, std::max<u64>(nblk / 2, 1));
    if (db->wg_key_n != wgs) {  // the first t of every workgroup's block range: once per handle
        YH_HIP(hipStreamSynchronize(st));
        if (db->d_wg_key) { yh_dfree(db, db->d_wg_key); db->d_wg_key = nullptr; }
        YH_HIP(hipMalloc((void**)&db->d_wg_key, ((u64)wgs + 2) * sizeof(u64)));
        k_wg_key<<<(wgs + 256) / 256, 256, 0, st>>>(db->d_shdr, nblk, wgs, db->d_wg_key);
        db->wg_key_n = wgs;
    }
    u32 R;
    YH_TRY(ensure_reps(db, R));
    // hits leave a workgroup pre-summed (one atomic per workgroup and reference), so few replicas do
    static const u32 r_env = [] { const char* e = yh_tune_env("YH_STREAM_REPS"); return e ? (u32)atoi(e) : (YH_XCD_ATOMICS ? 8u : 4u); }();
    while (R > 1 && R > r_env) R >>= 1;
    const bool fused = d_fused_excl != nullptr;
    const bool flags_too = flag_shared && db->has_index && !fused;
    if (flags_too) YH_TRY(claim_hit_flags(db));
    // No kernel in front of the streaming one: the counters it adds into are zero at rest, and every
    // workgroup finds its own range of the sample (two 64-ary wave searches while its first
    // super-block is in flight).
    u32* const reps2 = db->d_reps + db->reps_cap;
    StreamHit sh{(u32)N, db->d_reps, R - 1, db->d_srec, flags_too ? db->d_hit : nullptr, db->d_g, (u32)db->n_shared,
                 fused ? reps2 : nullptr, d_sample, db->d_bad, db->bad_gen, db->d_work_count};
    yh_ring_record_begin(db, db->ev_overlap);
    k_stream_lookup<<<wgs, STREAM_THREADS, 0, st>>>(reinterpret_cast<const u32x4*>(db->d_sdelta), db->d_shdr, nblk, d_sample,
                                                    (u32)n_sample, db->d_wg_key, db->sshift, sh);
    yh_ring_record_end(db, db->ev_overlap);
    k_reduce_replicas<<<(u32)((N + 255) / 256), 256, 0, st>>>(
        db->d_reps, R, N, d_overlap, (make_mask && !fused) ? db->d_mask : nullptr, make_mask ? db->d_maskbits : nullptr,
        (with_index && !fused) ? db->d_excl_e : nullptr,
        // (range_local -- a hash-range shard's first half: n_excl and the work list wait for the global subset)
        fused ? FusedRun{reps2, db->d_sizes, db->d_nshared, db->range_local ? nullptr : d_fused_excl, d_fused_match, d_bits_out,
                         db->d_hpo, db->range_local ? nullptr : db->d_work, db->d_work_count,
                         (u32)(db->n_ghost ? db->ghost_begin : N)}
              : FusedRun{});
    YH_HIP(hipGetLastError());
    return YH_OK;
}

int yh_q_overlap(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, bool flag_shared, bool make_mask) {
    if (!db->d_sdelta && db->n_hashes) { yh_set_error("this handle has no streaming layout (YH_DB_PAIRWISE_ONLY)"); return YH_ERR_UNSUPPORTED; }
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    if (n_sample > 0xfffffff0ull) { yh_set_error("sample larger than 2^32-16 hashes"); return YH_ERR_INVALID_ARG; }
    const bool with_index = flag_shared && db->has_index;
    flag_shared = with_index && db->n_shared > 0;
    if (N == 0 || db->n_hashes == 0 || n_sample == 0) {  // nothing can match: all-zero results
        YH_HIP(hipMemsetAsync(d_overlap, 0, std::max<u64>(N, 1) * sizeof(u32), st));
        if (flag_shared) { YH_HIP(hipMemsetAsync(db->d_hit, 0, db->n_shared + 15, st)); db->hit_clean = true; }
        if (with_index && N) YH_HIP(hipMemsetAsync(db->d_excl_e, 0, 3 * N * sizeof(u32), st));
        if (make_mask && N) {
            YH_HIP(hipMemsetAsync(db->d_mask, 0, N, st));
            YH_HIP(hipMemsetAsync(db->d_maskbits, 0, ((N + 255) / 256) * 32, st));
        }
        return YH_OK;
    }
    return yh_q_overlap_stream(db, d_sample, n_sample, d_overlap, flag_shared, with_index, make_mask);
}

// The `yacht run` step on the hash-sorted stream: overlap, subset = overlap > 0, exclusive counts.
// Returns 1 when this handle cannot take the fused path (the caller then runs the general one).
int yh_q_run_fused(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, u32* d_excl, u32* d_match, int phases,
                   u32* d_bits_out, const u32* d_global_bits, bool use_indexed) {
    static const bool off = [] { const char* e = yh_tune_env("YH_NO_FUSED_RUN"); return e && e[0] == '1'; }();
    if (!db->d_sdelta || !db->has_index || db->n_refs == 0 || db->n_hashes == 0 ||
        (db->n_postings && (!db->d_hrec || !db->d_hpo || !db->d_work)) || n_sample > 0xfffffff0ull)
        return 1;
    if (phases == 3 && (off || n_sample == 0)) return 1;
    hipStream_t st = db->stream;
    if (phases & 1) {  // lookup + reduce: overlap, n_match, the singleton part of n_excl, the subset bits
        if (n_sample == 0) {  // (sharded callers only: the one-call form takes the general path)
            const u64 N = db->n_refs;
            YH_HIP(hipMemsetAsync(d_overlap, 0, N * sizeof(u32), st));
            YH_HIP(hipMemsetAsync(d_excl, 0, N * sizeof(u32), st));
            YH_HIP(hipMemsetAsync(d_match, 0, N * sizeof(u32), st));
            YH_HIP(hipMemsetAsync(db->d_maskbits, 0, ((N + 255) / 256) * 32, st));
            if (d_bits_out) YH_HIP(hipMemsetAsync(d_bits_out, 0, ((N + 255) / 256) * 32, st));  // (the extent the reducer writes)
            if (db->d_work_count) YH_HIP(hipMemsetAsync(db->d_work_count, 0, sizeof(u32), st));
        } else if (use_indexed && db->has_dir && db->d_work && db->d_hrec && db->d_hpo) {
            const int rc = yh_q_overlap_indexed(db, d_sample, n_sample, d_overlap, true, d_excl, d_match, d_bits_out, true);
            if (rc != 2) return rc == YH_OK ? YH_ERR_UNSUPPORTED : rc;
        } else {
            YH_TRY(yh_q_overlap_stream(db, d_sample, n_sample, d_overlap, true, true, true, d_excl, d_match, d_bits_out));
        }
    }
    if (phases & 2) {
        yh_ring_record_begin(db, db->ev_excl);
        if (db->n_ghost && d_global_bits)
            k_ghost_bits<<<(u32)((db->n_ghost + 255) / 256), 256, 0, st>>>(d_global_bits, db->d_ghost_src, db->ghost_begin,
                                                                        db->n_ghost, db->d_maskbits);
        if (db->n_chunks)  // + the shared hashes of the subset's references whose other holders are all outside it
            launch_excl_pieces(db, db->d_maskbits, nullptr, d_excl, nullptr, nullptr, true);
        yh_ring_record_end(db, db->ev_excl);
    }
    YH_HIP(hipGetLastError());
    return YH_OK;
}

static bool fused_possible(const yh_db* db, const u32* d_fused_excl, bool for_exclusive) {
    static const bool fused_off = [] { const char* e = yh_tune_env("YH_NO_FUSED_RUN"); return e && e[0] == '1'; }();
    return d_fused_excl && for_exclusive && db->d_work && db->d_hrec && db->d_hpo && !fused_off;
}

// Second half of a step on a hash-range shard: global subset = OR of the ranks' gathered bits, then the exclusive pass.
int yh_q_range_finish(yh_db* db, const u32* d_gathered, u32 n_ranks, u64 stride_words, u32* d_excl) {
    if (!db->has_index || db->n_refs == 0) return YH_OK;
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    const bool sets = db->n_postings && db->d_hrec && db->d_hpo && db->d_work;
    yh_ring_record_begin(db, db->ev_excl);
    k_range_mask<<<(u32)((N + 255) / 256), 256, 0, st>>>(d_gathered, n_ranks, stride_words, N, db->d_sizes, db->d_nshared,
                                                          sets ? db->d_hpo : nullptr, db->d_maskbits, d_excl,
                                                          sets ? db->d_work : nullptr, db->d_work_count);
    if (sets && db->n_chunks) launch_excl_pieces(db, db->d_maskbits, nullptr, d_excl, nullptr, nullptr, true);
    yh_ring_record_end(db, db->ev_excl);
    YH_HIP(hipGetLastError());
    return YH_OK;
}

// overlap (and, with for_exclusive, the shared-hash flags, the subset mask and zeroed exclusive
// accumulators) through the directory; same outputs as yh_q_overlap(..., flag_shared, make_mask)
// d_fused_excl / d_fused_match non-null: the whole run step in three launches (see yh_q_run_fused)
int yh_q_overlap_indexed(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, bool for_exclusive,
                         u32* d_fused_excl, u32* d_fused_match, u32* d_bits_out, bool lookup_half_only) {
    if (!db->has_dir || !db->has_index) {
        yh_set_error("this handle has no directory of its distinct hashes (YH_DB_NO_DIRECTORY, or no index)");
        return YH_ERR_UNSUPPORTED;
    }
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    if (N == 0) return YH_OK;
    u32 R;
    YH_TRY(ensure_reps(db, R));
    // (k_reduce_replicas reads every replica of both sets: 16 -> 8 replicas took 4 us off the step, with no
    // measurable change of the lookup kernel at 1.3e5 hits per sample)
    static const u32 ri_env = [] { const char* e = yh_tune_env("YH_INDEX_REPS"); return e ? (u32)atoi(e) : 0u; }();
    // tiles of IDX_THREADS x U hashes once there are enough of them for every CU (k_index_lookup_tile)
    // (measured, 10^6-hash rotating samples with 1.6e5 hits: one hash per lane in 256-lane workgroups 38-40 us, tiles of U = 2
    // 35.7-36.3, U = 4 slower; no hits at all 32.5 / 31.5; an 83 k-hash sample 13 / 23)
    // YH_INDEX_TILE (debug gate): 256 = the small aggregating form, 1/2/4 = 1024-lane tiles of U
    static const long tile_env = [] { const char* e = yh_tune_env("YH_INDEX_TILE"); return e ? atol(e) : -1L; }();
    // (step time on the bench database, us, by sample size 1e5 / 2e5 / 3e5 / 4e5 / 5e5 / 7e5: small form 27.5 / 29.9 / 34.7 /
    // 37.9 / 40.5 / 43.5; 1024-lane tiles of one hash 29.3 / 30.0 / 33.5 / 33.7 / 36.4 / 41.7; of two 35.1 / 34.9 / 35.0 / 36.0 /
    // 36.9 / 41.1)
    int U = n_sample >= 512ull * IDX_THREADS ? 2 : n_sample >= 256ull * IDX_THREADS ? 1 : 256;
    if (tile_env == 1 || tile_env == 2 || tile_env == 4 || tile_env == 256) U = (int)tile_env;
    // the aggregating forms leave one atomic per (workgroup, reference): four replicas are enough there (10^6-hash sample:
    // step 49.9 -> 48.4 us; two: 51.7; 83 k-hash real-shape sample: 28.8 / 26.3 / 25.4 us with 8 / 4 / 2)
    const u32 r_want = ri_env ? ri_env : 4u;
    while (R > 1 && R > r_want) R >>= 1;
    const bool fused = fused_possible(db, d_fused_excl, for_exclusive);
    if (for_exclusive && db->n_shared && !fused) YH_TRY(claim_hit_flags(db));
    u32* const reps1 = db->d_reps;
    u32* const reps2 = reps1 + db->reps_cap;
    // (no kernel in front of the lookup: the counters are zero at rest)
    yh_ring_record_begin(db, db->ev_overlap);
    u8* const d_hitflags = (for_exclusive && db->n_shared && !fused) ? db->d_hit : nullptr;
    u32* const d_reps2 = fused ? reps2 : nullptr;
    const u32* const d_filter = yh_filter_of(db);
#define YH_TILE_LAUNCH(UU, TT, BB, FILTER)                                                                                        \
    k_index_lookup_tile<UU, TT, BB><<<(u32)((n_sample + (u64)(TT) * (UU) - 1) / ((u64)(TT) * (UU))), TT, 0, st>>>(                 \
        TileLookup{d_sample, n_sample, yh_dir_view(db), FILTER, db->filter_mul, db->d_po, db->d_pr, reps1, R - 1, N, d_hitflags,    \
                   d_reps2, db->d_work_count, db->d_bad, db->bad_gen})
    if (n_sample && db->n_distinct && U == 256)
        // small samples: latency-bound, so no filter read in front of the bucket; 256-lane workgroups keep every CU busy
        YH_TILE_LAUNCH(1, 256, 8, nullptr);
    else if (n_sample && db->n_distinct && U == 4) YH_TILE_LAUNCH(4, IDX_THREADS, YH_IDX_TBITS, d_filter);
    else if (n_sample && db->n_distinct && U == 2) YH_TILE_LAUNCH(2, IDX_THREADS, YH_IDX_TBITS, d_filter);
    else if (n_sample && db->n_distinct && U == 1) YH_TILE_LAUNCH(1, IDX_THREADS, YH_IDX_TBITS, d_filter);
#undef YH_TILE_LAUNCH
    else if (fused && db->d_work_count)
        YH_HIP(hipMemsetAsync(db->d_work_count, 0, sizeof(u32), st));
    yh_ring_record_end(db, db->ev_overlap);
    k_reduce_replicas<<<(u32)((N + 255) / 256), 256, 0, st>>>(
        reps1, R, N, d_overlap, (for_exclusive && !fused) ? db->d_mask : nullptr,
        for_exclusive ? db->d_maskbits : nullptr, (for_exclusive && !fused) ? db->d_excl_e : nullptr,
        // (range_local -- a hash-range shard's first half: n_excl and the work list wait for the global subset)
        fused ? FusedRun{reps2, db->d_sizes, db->d_nshared, db->range_local ? nullptr : d_fused_excl, d_fused_match, d_bits_out,
                         db->d_hpo, db->range_local ? nullptr : db->d_work, db->d_work_count,
                         (u32)(db->n_ghost ? db->ghost_begin : N)}
              : FusedRun{});
    if (fused && !lookup_half_only) {  // (sharded run: the posting-list half follows the exchange of the subset bits)
        yh_ring_record_begin(db, db->ev_excl);
        if (db->n_chunks) launch_excl_pieces(db, db->d_maskbits, nullptr, d_fused_excl, nullptr, nullptr, true);
        yh_ring_record_end(db, db->ev_excl);
    }
    YH_HIP(hipGetLastError());
    return fused ? 2 : YH_OK;  // 2: the exclusive counts are done too
}

// One launch of k_step_fused: the exclusive pass of the step before the previous one, the reducer of the previous
// one, and (d_sample != nullptr) the lookup of a new one.  d_sample == nullptr: a draining launch (yh_run_device_join).
// The caller has checked yh_q_step_fused_ok and allocated the step contexts 0..2.
bool yh_q_step_fused_ok(const yh_db* db, u64 n_sample) {
    return db->has_dir && db->has_index && db->d_cbkt && fused_possible(db, reinterpret_cast<const u32*>(db), true) &&
           db->n_chunks && !db->n_ghost && n_sample >= 1 && n_sample <= 0xfffffff0ull;
}
int yh_q_step_fused(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, u32* d_excl, u32* d_match) {
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    u32 R;
    YH_TRY(ensure_reps(db, R));
    while (R > 1 && R > 4u) R >>= 1;
    // One geometry per launch: the sample's size picks it (as for the stand-alone lookup: 256-lane workgroups below 262 144
    // hashes, 1024-lane tiles of one or two hashes per lane above); a draining launch takes the small one.
    const int form = !d_sample ? 0 : n_sample >= 512ull * 1024 ? 2 : n_sample >= 256ull * 1024 ? 1 : 0;
    const u32 threads = form ? 1024u : 256u, waves = threads / 64, tslots = form ? 1024u : 256u;
    StepFused s{};
    u32 lds_words = 3 * tslots;  // the lookup role's hit table
    if (db->pend_excl >= 0) {   // exclusive pass of the step reduced by the previous launch
        const int c = db->pend_excl;
        s.excl = excl_args(db, db->ctx_count[c], db->ctx_work[c], db->ctx_bits[c], nullptr, db->pend_excl_out, nullptr, nullptr, true);
        s.excl_wgs = std::min<u32>((db->n_chunks + waves - 1) / waves, 4096u / waves);  // (most find nothing and leave after one read)
        s.excl_lds = s.excl.n_mask_words <= EXCL_LDS_WORDS ? 1u : 0u;
        if (s.excl_lds) lds_words = std::max(lds_words, s.excl.n_mask_words);
    }
    if (db->pend_red >= 0) {    // reducer of the step looked up by the previous launch
        const int c = db->pend_red;
        u32* const reps1 = db->d_reps + (u64)db->pend_red_parity * 2 * db->reps_cap;
        s.r_reps = reps1;
        s.r_R = R;
        s.r_n = N;
        s.r_out = db->pend_red_out[0];
        s.r_maskbits = db->ctx_bits[c];
        s.r_fused = FusedRun{reps1 + db->reps_cap, db->d_sizes, db->d_nshared, db->pend_red_out[1], db->pend_red_out[2], nullptr,
                             db->d_hpo, db->ctx_work[c], db->ctx_count[c], (u32)N};
        s.red_wgs = (u32)((N + 4ull * threads - 1) / (4ull * threads));  // (4 references per thread)
    }
    int c_new = -1;
    if (d_sample) {
        c_new = (int)(db->pipe_k % 3);
        const int parity = (int)(db->pipe_k & 1);
        u32* const reps1 = db->d_reps + (u64)parity * 2 * db->reps_cap;
        const u32 per_wg = threads * (form == 2 ? 2u : 1u);
        // (the small form reads no presence filter: such a launch is latency-bound)
        s.look = TileLookup{d_sample, n_sample, yh_dir_view(db), form ? yh_filter_of(db) : nullptr, db->filter_mul, db->d_po, db->d_pr,
                            reps1, R - 1, N, nullptr, reps1 + db->reps_cap, db->ctx_count[c_new], nullptr, 0u};
        s.look_wgs = (u32)((n_sample + per_wg - 1) / per_wg);
    }
    const u32 total = s.excl_wgs + s.red_wgs + s.look_wgs;
    if (total) {
        if (d_sample) yh_ring_record_begin(db, db->ev_overlap);
        if (form == 2) k_step_fused<2, 1024, 10><<<total, 1024, lds_words * sizeof(u32), st>>>(s);
        else if (form == 1) k_step_fused<1, 1024, 10><<<total, 1024, lds_words * sizeof(u32), st>>>(s);
        else k_step_fused<1, 256, 8><<<total, 256, lds_words * sizeof(u32), st>>>(s);
        if (d_sample) yh_ring_record_end(db, db->ev_overlap);
        YH_HIP(hipGetLastError());
    }
    // the pipeline moves on
    db->pend_excl = db->pend_red;
    db->pend_excl_out = db->pend_red >= 0 ? db->pend_red_out[1] : nullptr;
    db->pend_red = c_new;
    if (d_sample) {
        db->pend_red_parity = (int)(db->pipe_k & 1);
        db->pend_red_out[0] = d_overlap;
        db->pend_red_out[1] = d_excl;
        db->pend_red_out[2] = d_match;
        db->pipe_last_ctx = c_new;
        ++db->pipe_k;
    }
    return YH_OK;
}

int yh_q_overlap_bsearch(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap) {
    if (!db->d_values) { yh_set_error("yh_overlap_bsearch needs a handle created with YH_DB_KEEP_CSR"); return YH_ERR_UNSUPPORTED; }
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    YH_HIP(hipMemsetAsync(d_overlap, 0, std::max<u64>(N, 1) * sizeof(u32), st));
    if (N == 0) return YH_OK;
    k_overlap_bsearch<<<grid_for(N * WAVE, 256, 8192), 256, 0, st>>>(db->d_values, db->d_offsets, N, d_sample, n_sample,
                                                                     d_overlap);
    YH_HIP(hipGetLastError());
    return YH_OK;
}

// Exclusive counts for an arbitrary subset (yh_exclusive; yh_run on handles without holder sets).  The caller has run
// yh_q_overlap / yh_q_overlap_indexed on the SAME sample with flag_shared / for_exclusive: d_overlap holds its counts,
// db->d_hit the shared hashes found in the sample, and the three accumulators db->d_excl_e/_m/d_ovsh are zero
// (k_reduce_replicas).  d_maskbits == nullptr: the subset comes as bytes (d_mask) and is turned into bits here.
int yh_q_exclusive(yh_db* db, const u8* d_mask, const u64* d_sample, u64 n_sample, const u32* d_overlap,
                   u32* d_excl, u32* d_match, const u32* d_maskbits) {
    (void)d_sample; (void)n_sample;
    if (!db->has_index) { yh_set_error("this handle was created with YH_DB_NO_INDEX"); return YH_ERR_UNSUPPORTED; }
    if (db->flags & YH_DB_PAIRWISE_ONLY) { yh_set_error("this handle holds posting lists only"); return YH_ERR_UNSUPPORTED; }
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    if (N == 0) return YH_OK;
    yh_ring_record_begin(db, db->ev_excl);
    if (!d_maskbits) {
        k_mask_bits<<<(u32)((N + 255) / 256), 256, 0, st>>>(d_mask, N, db->d_maskbits);
        d_maskbits = db->d_maskbits;
    }
    if (db->n_shared && db->n_chunks) {  // the shared hashes of the subset's references, by the reference-major postings
        YH_HIP(hipMemsetAsync(db->d_work_count, 0, sizeof(u32), st));
        k_excl_worklist<<<(u32)((N + 255) / 256), 256, 0, st>>>(N, d_maskbits, db->d_nshared, db->d_rpo, db->d_work, db->d_work_count);
        launch_excl_pieces(db, d_maskbits, db->d_hit, db->d_excl_e, db->d_excl_m, db->d_ovsh);
    }
    const bool clean_hit = db->d_hit && db->n_shared;
    k_excl_final<<<grid_for(N, 256, 1u << 22), 256, 0, st>>>(
        N, d_mask, db->d_sizes, db->d_nshared, d_overlap, db->d_excl_e, db->d_excl_m, db->d_ovsh, d_excl, d_match,
        clean_hit ? reinterpret_cast<uint4*>(db->d_hit) : nullptr, clean_hit ? (db->n_shared + 15) / 16 : 0);
    if (clean_hit) db->hit_clean = true;
    yh_ring_record_end(db, db->ev_excl);
    YH_HIP(hipGetLastError());
    return YH_OK;
}
