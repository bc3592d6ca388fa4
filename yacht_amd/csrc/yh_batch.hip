// yh_batch.hip -- batched `yacht run`: up to 256 samples against the resident database in one pass
#include "yh_common.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

// =================================================================================================
// Batched `yacht run`: up to 64 samples against the resident database in one pass
// =================================================================================================
// Samples are looked up through the distinct-hash directory (k_index_lookup's scheme), one lane per
// sample hash of ANY sample.  Per-sample state is carried as 64-bit words: maskword[r] = samples that overlap
// reference r.  Exclusivity for all samples at once, over a reference's DISTINCT holder sets (k_batch_sets):
//     excl = maskword[r] & ~(OR of the other holders' words).
// Round 6: up to YH_BATCH_MAX_SAMPLES = 256 samples per pass.  The words come in PLANES of 64 samples -- plane w = samples
// 64 w .. 64 w + 63, maskword[w * N + r] -- so a batch of <= 64 samples is exactly the one-plane layout of rounds 3-5, and a
// hash-range rank whose share of a block is an eighth of every sample gets four times the lookups per block for the same
// dozen launches at their floors (<= 64 samples capped an eight-rank block at 8e6 lookups).
namespace {

constexpr u32 BATCH_MAX = 256;                    // samples per batch (include/yacht_hip.h: YH_BATCH_MAX_SAMPLES)
constexpr u32 BATCH_MAX_PLANES = BATCH_MAX / 64;  // word planes a slot's scratch has room for
inline u32 planes_of(u32 n_samples) { return (n_samples + 63u) / 64u; }

constexpr int EXCL_BLOCK = 256;
inline u32 grid_for(u64 work_items, u32 block, u32 max_blocks = 16384) {
    u64 g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (u32)g;
}

// A workgroup's unit of work is a TILE of BATCH_TILE consecutive hashes of ONE sample; its hits are summed per reference in
// an LDS table (two probes, overflow counts directly) and leave as one global atomic per (tile, reference hit) -- a
// 2 048-hash tile of the bench samples has ~600 hits on ~190 references (timing-only build without any counting: 0.589
// against 0.712 ms per 32-sample call).
#ifndef YH_BATCH_TILE
#define YH_BATCH_TILE 2048
#endif
constexpr u32 BATCH_TILE = YH_BATCH_TILE;
#ifndef YH_BATCH_U
#define YH_BATCH_U 1
#endif
// hashes a lane has in flight.  ONE: measured (scripts/sweep_batch_u.sh, profiles/r04/batch_u.txt) 0.0154 ms per sample at 1,
// 0.0168 at 2, 0.0233 at 4, 0.0371 at 8 -- unlike the single-sample lookup (two per lane), this pass has its parallelism
// from eight resident workgroups per CU working on the SAME presence-filter lines for different samples; more hashes per
// lane cost registers, i.e. resident waves, and spread a workgroup's reads in time.
constexpr int BATCH_U = YH_BATCH_U;

constexpr u32 BATCH_TBITS = 10;
__global__ void __launch_bounds__(256) k_batch_lookup(const u64* __restrict__ samples, const u64* __restrict__ soff,
                                                      u32 n_samples, const YhDirView dv, const u64* __restrict__ po,
                                                      const u32* __restrict__ pr, u64 n_refs,
                                                      u32* __restrict__ overlap /* [B][N] */, u32* __restrict__ ovsh /* [B][N]: hits on shared hashes */,
                                                      const u32* __restrict__ filter, u64 filter_mul) {
    constexpr u32 TSLOTS = 1u << BATCH_TBITS;
    __shared__ u64 off[BATCH_MAX + 1];
    __shared__ u64 s_max_nt;
    __shared__ u32 tkey[TSLOTS], tcnt[TSLOTS], tcnt2[TSLOTS];
    for (u32 q = threadIdx.x; q <= n_samples; q += 256) off[q] = soff[q];
    if (threadIdx.x == 0) s_max_nt = 0;
    __syncthreads();
    {   // the longest sample's tiles (one wave: a maximum over <= 256 lengths)
        u64 m = 0;
        if (threadIdx.x < 64)
            for (u32 q = threadIdx.x; q < n_samples; q += 64) m = max(m, (off[q + 1] - off[q] + BATCH_TILE - 1) / BATCH_TILE);
        if (threadIdx.x < 64 && m) atomicMax((unsigned long long*)&s_max_nt, (unsigned long long)m);
    }
    __syncthreads();
    const u64 max_nt = s_max_nt;
    // The tiles of ALL samples in the order of their place in the hash range: slot (v, s) is tile i = v * nt_s / max_nt of
    // sample s (the sorted samples are uniform over the range: tile i of a sample sits at the i / nt_s quantile).  The
    // workgroups resident at any moment then (i) work on all samples -- a sample's hits land on its few hundred present
    // references, and same-address atomics serialize at ~11 ns each -- and (ii) read the SAME presence-filter lines for all
    // of them; (iii) the slots of one quantile step sit on ONE XCD (workgroups b and b + 8 share an XCD: observed
    // round-robin dispatch; speed only, never correctness), whose L2 then serves a filter line to all samples but the
    // first: inside a group of 8 steps x n_samples slots, workgroup l takes step l % 8, sample l / 8.
    // (A multiplicative permutation of 256-hash chunks, which gave (i) only: 21.0 us per 1e6 sample hashes; with (ii)
    // 19.7, with (iii) 18.6.)
    const u64 n_slots = max_nt * n_samples;
    const u64 group = 8ull * n_samples;
    for (u64 c = blockIdx.x; c < (n_slots + group - 1) / group * group; c += gridDim.x) {  // (workgroup-uniform)
        const u64 l = c % group;
        const u64 v = (c / group) * 8 + (l & 7u);
        const u32 s = (u32)(l >> 3);
        if (v >= max_nt) continue;
        const u64 n_s = off[s + 1] - off[s], nt_s = (n_s + BATCH_TILE - 1) / BATCH_TILE;
        if (nt_s == 0) continue;
        const u64 i = v * nt_s / max_nt;
        if (v > 0 && (v - 1) * nt_s / max_nt == i) continue;  // (a shorter sample: this tile had its slot already)
        for (u32 k = threadIdx.x; k < TSLOTS; k += 256) { tkey[k] = 0; tcnt[k] = 0; tcnt2[k] = 0; }
        __syncthreads();
        u32* row = overlap + (u64)s * n_refs;
        u32* row2 = ovsh + (u64)s * n_refs;
        auto add = [&](u32 ref, bool shared) {
#if defined(YH_ABLATE_BATCH) && YH_ABLATE_BATCH  // timing-only build (build.py build_variant): no counting, results wrong
            if (ref == 0x7ffffff1u) row[0] = 1;
            return;
#endif
            u32 slot = (ref * 2654435761u) >> (32 - BATCH_TBITS);
#if defined(YH_BATCH_DIRECT_ATOMICS) && YH_BATCH_DIRECT_ATOMICS  // measurement build: one global atomic per hit, no LDS table (results right)
            atomicAdd(&row[ref], 1u);
            if (shared) atomicAdd(&row2[ref], 1u);
            return;
#endif
#pragma unroll 1
            for (int probe = 0; probe < 2; ++probe, slot = (slot + 1) & (TSLOTS - 1)) {
                const u32 old = atomicCAS(&tkey[slot], 0u, ref + 1);
                if (old == 0 || old == ref + 1) {
                    atomicAdd(&tcnt[slot], 1u);
                    if (shared) atomicAdd(&tcnt2[slot], 1u);
                    return;
                }
            }
            atomicAdd(&row[ref], 1u);  // crowded table: count directly
            if (shared) atomicAdd(&row2[ref], 1u);
        };
        const u64 k_end = min(n_s, (i + 1) * BATCH_TILE);
        auto hits_of = [&](u32 r) {  // what a found hash adds: its holder, or the holders of a shared hash
            if (r == YH_DIR_NONE) return;
            if (!(r & 0x80000000u)) {
                add(r, false);
            } else {
                const u32 gi = r & 0x7fffffffu;
                for (u64 q = po[gi], qe = po[gi + 1]; q < qe; ++q) add(pr[q], true);
            }
        };
        // (Round 6, measured and dropped -- profiles/r06/batch_share_G*_B256*.txt: the eight rounds of a tile SOFTWARE-PIPELINED -- round
        // j + 1's presence word and round j + 2's sample hash requested behind round j's bucket, so that a round is one dependent
        // latency long instead of two: k_batch_lookup 3 622 -> 3 693 us per block of 256 samples at G = 1, 496 -> 512 at G = 8 --
        // nothing: with 32 resident waves per CU taking turns the pass is not waiting for its own chain.)
        // BATCH_U hashes of a lane at a time: all presence words are requested, then the buckets of the hashes that passed,
        // then they are looked at (BATCH_U = 1: word, bucket, counts, next hash)
        for (u64 k0 = i * BATCH_TILE; k0 < k_end; k0 += 256u * BATCH_U) {  // (workgroup-uniform)
            u64 h[BATCH_U];
            bool ok[BATCH_U];
            u32 w[BATCH_U];
            u64 bit[BATCH_U];
            YhDirView::v4u a[BATCH_U], b[BATCH_U], c[BATCH_U], d[BATCH_U];
#pragma unroll
            for (int u = 0; u < BATCH_U; ++u) {
                const u64 k = k0 + (u64)u * 256u + threadIdx.x;
#if defined(YH_BATCH_NT) && (YH_BATCH_NT & 2)
                h[u] = __builtin_nontemporal_load(&samples[off[s] + min(k, k_end - 1)]);
#else
                h[u] = samples[off[s] + min(k, k_end - 1)];
#endif
                ok[u] = k < k_end && h[u] <= dv.max_hash;
                if (!ok[u]) h[u] = 0;  // (still a valid word / bucket to read)
            }
#if defined(YH_ABLATE_BATCH_READS) && (YH_ABLATE_BATCH_READS & 2)
            if (false) {
#else
            if (filter) {
#endif
#pragma unroll
                for (int u = 0; u < BATCH_U; ++u) {
                    bit[u] = yh_bucket_of(h[u], dv.bkt_lsh, filter_mul);
                    w[u] = filter[bit[u] >> 5];
                }
#pragma unroll
                for (int u = 0; u < BATCH_U; ++u) {
                    const u32 m = yh_filter_mask(h[u], bit[u]);
                    ok[u] = ok[u] && (w[u] & m) == m;
#if defined(YH_ABLATE_BATCH_READS) && (YH_ABLATE_BATCH_READS & 1)  // timing-only build: the filter word is read, no bucket is (results wrong)
                    ok[u] = ok[u] && h[u] == 0x123456789abcdefull;
#endif
                }
            }
#if defined(YH_ABLATE_BATCH_READS) && (YH_ABLATE_BATCH_READS & 2)  // timing-only build: no filter word is read; the hashes that WOULD pass a
            // perfect filter are not known, so every fourth hash reads its bucket (0.25 per hash: the bench samples' 0.27 + none of the 0.12 false positives)
#pragma unroll
            for (int u = 0; u < BATCH_U; ++u) ok[u] = ok[u] && ((h[u] >> 7) & 3ull) == 0;
#endif
            if (dv.cbkt) {
#pragma unroll
                for (int u = 0; u < BATCH_U; ++u) {
                    a[u] = b[u] = c[u] = d[u] = YhDirView::v4u{0u, 0u, 0u, 0u};
#if defined(YH_BATCH_NT) && (YH_BATCH_NT & 1)
                    if (ok[u]) dv.cbkt_request_nt(h[u], a[u], b[u], c[u], d[u]);
#else
                    if (ok[u]) dv.cbkt_request(h[u], a[u], b[u], c[u], d[u]);
#endif
                }
#pragma unroll
                for (int u = 0; u < BATCH_U; ++u) asm volatile("" : "+v"(a[u]), "+v"(b[u]), "+v"(c[u]), "+v"(d[u]));  // (see YhDirView::find)
            }
#pragma unroll
            for (int u = 0; u < BATCH_U; ++u) {
                if (!ok[u]) continue;
                hits_of(dv.cbkt ? dv.cbkt_resolve(h[u], a[u], b[u], c[u], d[u]) : dv.find(h[u]));
            }
        }
        __syncthreads();
        for (u32 k = threadIdx.x; k < TSLOTS; k += 256)
            if (tkey[k]) {
                atomicAdd(&row[tkey[k] - 1], tcnt[k]);
                if (tcnt2[k]) atomicAdd(&row2[tkey[k] - 1], tcnt2[k]);
            }
        __syncthreads();
    }
}

// (Round 6, measured and dropped: the same lookups in WINDOW-MAJOR order.  The filter's word is a monotone function of the hash, so the
// hashes of all samples that fall into a window of 2 048 consecutive filter words are a contiguous slice of each sorted sample; a
// workgroup staged its window in LDS -- the whole filter then read ONCE per pass, coalesced, instead of one L2 read per sample
// hash, which is 30 % of this pass: profiles/r06/ablate_batch_reads.txt -- and walked the samples' slices against it, a wave per
// sample (then four samples in flight per wave), cursors per sample in LDS, one global atomic per hit (a (window, sample) pair
// holds ~27 hits on as many references: nothing to combine; one atomic per hit costs the tile order +0.4 ms).  Bit-exact
// (test_gpu_batch.py with it forced) and SLOWER: 4.10-4.17 ms per block of 256 samples against 3.62-3.71, and 1.23 against 0.71
// for an eighth of the hash space: ~50 hashes of a sample per window are 512-byte reads scattered over 256 arrays, every
// workgroup starts with 512 binary searches, and the lookups of one tile no longer share an LDS hit table.  Git history:
// "batch lookup in window-major order".)
// maskword[r] = samples with overlap > 0; anybits = "some sample overlaps r" (for k_batch_worklist)
__global__ void __launch_bounds__(256) k_batch_maskwords(const u32* __restrict__ overlap, u32 n_samples, u64 n_refs,
                                                         u64* __restrict__ maskword, u32* __restrict__ anybits) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    u64 any = 0;
    for (u32 s0 = 0; s0 < n_samples; s0 += 64) {  // (uniform) one plane of 64 samples a round
        const u32 ns = min(64u, n_samples - s0);
        u64 w = 0;
        if (r < n_refs) {
            // eight rows requested together (one row a step, each load waited for before the next was asked: 29.6 us for the 21.8 MB of a
            // block of 64 -- 0.74 TB/s: profiles/r05/batch_share_kernels.txt)
            u32 s = 0;
            for (; s + 8 <= ns; s += 8) {
                u32 v[8];
#pragma unroll
                for (u32 q = 0; q < 8; ++q) v[q] = overlap[(u64)(s0 + s + q) * n_refs + r];
#pragma unroll
                for (u32 q = 0; q < 8; ++q) w |= (u64)(v[q] != 0) << (s + q);
            }
            for (; s < ns; ++s) w |= (u64)(overlap[(u64)(s0 + s) * n_refs + r] != 0) << s;
            maskword[(u64)(s0 >> 6) * n_refs + r] = w;
        }
        any |= w;
    }
    const u64 bal = __ballot(any != 0);
    if (anybits && (threadIdx.x & 63) == 0) {  // (nullptr: a first half whose second half makes its own -- k_batch_or_maskwords)
        anybits[(r >> 5)] = (u32)bal;
        anybits[(r >> 5) + 1] = (u32)(bal >> 32);
    }
}

// The work list of the exclusive pass: every reference some sample overlaps appends pieces of <= BATCH_PIECE of its
// DISTINCT holder-set records (yh_db::d_hrec, [hpo[r], hpo[r + 1])); work_count zeroed by the caller.  The piece size is
// the one yh_db::d_work was sized for at build time (k_chunk_counts: ceil(nshared[r] / YH_EXCL_PIECE) pieces per reference,
// and a reference has no more holder-set records than shared hashes), whatever a build variant sets it to.
constexpr u32 BATCH_PIECE = YH_EXCL_PIECE;
__global__ void __launch_bounds__(256) k_batch_worklist(u64 n, const u32* __restrict__ anybits, const u32* __restrict__ hpo,
                                                        uint4* __restrict__ work, u32* __restrict__ work_count) {
    __shared__ u32 lds[5];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    const bool in = j < n && ((anybits[j >> 5] >> (j & 31u)) & 1u);
    const u32 first = in ? hpo[j] : 0u, last = in ? hpo[j + 1] : 0u;
    const u32 np = (last - first + BATCH_PIECE - 1u) / BATCH_PIECE;
    u32 v = np;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 t = (u32)__shfl_up((int)v, off);
        if (lane >= (u32)off) v += t;
    }
    if (lane == 63) lds[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 total = lds[0] + lds[1] + lds[2] + lds[3];
        lds[4] = total ? atomicAdd(work_count, total) : 0u;
    }
    __syncthreads();
    u32 at = lds[4] + v - np;
    for (u32 q = 0; q < wv; ++q) at += lds[q];
    for (u32 i = 0; i < np; ++i) {
        const u32 f = first + i * BATCH_PIECE;
        work[at + i] = make_uint4((u32)j, f, min(f + BATCH_PIECE, last), 0u);
    }
}

// One wave per piece (reference r, <= 256 of its holder-set records).  For all samples at once: a shared hash of r is
// exclusive to r in sample s iff r is in s's subset and none of the OTHER holders is --
//     excl = maskword[r] & ~(OR of the other holders' mask words),
// and every set bit s of it adds the record's multiplicity (shared hashes of r with exactly these other holders) to
// ex_e[s][r]: one wave sum and one atomic per (piece, sample that overlaps r).  (Before: a pass over ALL postings to
// collect those of masked references, then bit-sliced counting per posting: 149 us per 32-sample call against ~20.)
__global__ void __launch_bounds__(256) k_batch_sets(const uint4* __restrict__ work, const u32* __restrict__ work_count,
                                                    const uint4* __restrict__ hrec, const uint4* __restrict__ hrecx,
                                                    const u32* __restrict__ hmult, const u32* __restrict__ pr,
                                                    const u64* __restrict__ maskword_all, u64 n_refs, u32* __restrict__ ex_e_all, u32 n_planes) {
    const u32 lane = threadIdx.x & 63u;
    const u32 n_work = *work_count;
    for (u32 w = blockIdx.x * 4u + (threadIdx.x >> 6); w < n_work; w += gridDim.x * 4u) {
        const uint4 piece = work[w];
        const u32 r = piece.x;
        for (u32 pl = 0; pl < n_planes; ++pl) {  // (wave-uniform) the planes of 64 samples, one after the other: the records come from the L2 again
        const u64* maskword = maskword_all + (u64)pl * n_refs;
        u32* ex_e = ex_e_all + (u64)pl * 64u * n_refs;
        const u64 wr = maskword[r];
        if (wr == 0) continue;  // no sample of this plane overlaps r
        for (u32 first = piece.y; first < piece.z; first += 256u) {  // (wave-uniform; one round unless YH_EXCL_PIECE > 256)
        u64 excl[4];
        u32 mu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 k = first + 64u * u + lane;
            excl[u] = 0;
            mu[u] = 0;
            if (k < piece.z) {
                const uint4 rec = hrec[k];
                u64 others = 0;
                if (rec.w != 0xffffffffu) {  // up to seven other holders inline
                    if (rec.w > 0) others |= maskword[rec.x];
                    if (rec.w > 1) others |= maskword[rec.y];
                    if (rec.w > 2) others |= maskword[rec.z];
                    if (rec.w > 3) {
                        const uint4 rx = hrecx[k];
                        others |= maskword[rx.x];
                        if (rec.w > 4) others |= maskword[rx.y];
                        if (rec.w > 5) others |= maskword[rx.z];
                        if (rec.w > 6) others |= maskword[rx.w];
                    }
                } else {                     // a longer list: {first index in pr, holders}
                    for (u32 q = rec.x, qe = rec.x + rec.y; q < qe; ++q) {
                        const u32 o = pr[q];
                        if (o != r) others |= maskword[o];
                    }
                }
                excl[u] = wr & ~others;
                mu[u] = hmult[k];
            }
        }
        u64 todo = wr;  // (wave-uniform: the samples that overlap r)
        while (todo) {
            const u32 s = (u32)__ffsll((long long)todo) - 1u;
            todo &= todo - 1;
            u32 v = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) v += ((excl[u] >> s) & 1ull) ? mu[u] : 0u;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += (u32)__shfl_xor((int)v, off);
            if (lane == 0 && v) atomicAdd(&ex_e[(u64)s * n_refs + r], v);
        }
        }
        }
    }
}

// hash-range shards: maskword[r] = OR over the ranks' gathered words; anybits as k_batch_maskwords makes them
__global__ void __launch_bounds__(256) k_batch_or_maskwords(const u64* __restrict__ gathered /* [n_ranks][n_planes][N] */, u32 n_ranks, u64 n_refs,
                                                            u64* __restrict__ maskword, u32* __restrict__ anybits, u32 n_planes) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    u64 any = 0;
    if (r < n_refs)
        for (u32 pl = 0; pl < n_planes; ++pl) {
            // ... and this rank's own words, which the slot still holds from its first half: they are part of every complete
            // gather anyway, and with them the subset covers every cell this rank counted into even when the gathered words are
            // INCOMPLETE (an exchange that overflowed its capacity: the block is repeated) -- k_batch_final_sparse clears the hits
            // on shared hashes of the subset's cells only, and the slot's next first half relies on all of them being zero
            u64 w = maskword[(u64)pl * n_refs + r];
            for (u32 k = 0; k < n_ranks; ++k) w |= gathered[((u64)k * n_planes + pl) * n_refs + r];
            maskword[(u64)pl * n_refs + r] = w;
            any |= w;
        }
    const u64 bal = __ballot(any != 0);
    if ((threadIdx.x & 63) == 0) {
        anybits[(r >> 5)] = (u32)bal;
        anybits[(r >> 5) + 1] = (u32)(bal >> 32);
    }
}

// ex_e -> n_excl in place, n_match written, for every (sample, reference).  A shared hash found in sample s has ALL its
// holders in s's subset (each of them overlaps s by that very hash), so it is exclusive to none of them: the matches are
// the hits on unshared hashes, n_match = overlap - hits on shared hashes (what the single-sample step does too).
// (maskword != nullptr -- a hash-range shard: the subset is the global one, a reference may be in it without an overlap
// in THIS rank's range)
__global__ void __launch_bounds__(256) k_batch_final(u32 n_samples, u64 n_refs, const u32* __restrict__ sizes,
                                                     const u32* __restrict__ nshared, const u32* __restrict__ overlap,
                                                     u32* __restrict__ ovsh, u32* __restrict__ ex_e,
                                                     u32* __restrict__ ex_m, const u64* __restrict__ maskword) {
    const u64 total = (u64)n_samples * n_refs;
    for (u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 r = t % n_refs;
        const u32 ov = overlap[t];
        // the slot's hits on shared hashes are read once, here, and left zero for the slot's next first half (which then
        // has 22 MB less to clear in front of its lookups: BatchSlot::ovsh_clean); only the few non-zero ones are written
        const u32 sh = ovsh[t];
        if (sh) ovsh[t] = 0;
        const u64 smp = t / n_refs;
        const bool in = maskword ? ((maskword[(smp >> 6) * n_refs + r] >> (smp & 63u)) & 1ull) != 0 : ov != 0;
        if (in) {
            ex_e[t] = sizes[r] - nshared[r] + ex_e[t];
            ex_m[t] = ov - sh;
        } else {
            ex_e[t] = 0;
            ex_m[t] = 0;
        }
    }
}

// The same over the ENTRIES of the batch only -- the set bits of the subset words -- (round 6): the dense pass above reads three and
// writes two [B][N] arrays, 0.44 GB and 100 us per block of 256 samples at rs214 scale whatever the rank's share of the lookups,
// for ~60 000 cells that hold anything.  Cells outside the subset: ex_e is zero already (cleared in front of k_batch_sets, which
// adds to subset cells only), ex_m is cleared by the caller (one fill), overlap / ovsh are zero there on every shard (a
// reference outside the GLOBAL subset overlaps the sample in no range).  One lane per reference, its samples plane by plane.
__global__ void __launch_bounds__(256) k_batch_final_sparse(u32 n_samples, u64 n_refs, u32 n_planes, const u32* __restrict__ sizes,
                                                            const u32* __restrict__ nshared, const u32* __restrict__ overlap,
                                                            u32* __restrict__ ovsh, u32* __restrict__ ex_e, u32* __restrict__ ex_m,
                                                            const u64* __restrict__ maskword) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (r >= n_refs) return;
    u32 base = 0xffffffffu;  // sizes[r] - nshared[r], fetched behind the first set bit (most references have none)
    for (u32 pl = 0; pl < n_planes; ++pl) {
        u64 w = maskword[(u64)pl * n_refs + r];
        while (w) {
            const u32 s = pl * 64u + (u32)__ffsll((long long)w) - 1u;
            w &= w - 1;
            if (s >= n_samples) continue;
            if (base == 0xffffffffu) base = sizes[r] - nshared[r];
            const u64 t = (u64)s * n_refs + r;
            const u32 sh = ovsh[t];
            if (sh) ovsh[t] = 0;  // (left zero for the slot's next first half: BatchSlot::ovsh_clean)
            ex_e[t] += base;
            ex_m[t] = overlap[t] - sh;
        }
    }
}

// ---- the batch's result in compact form (include/yacht_hip.h: yh_run_batch_rows_*) --------------------------------------
// One entry per set bit s of maskword[r], in (r, s) order.  ROWS_BLOCK references per workgroup: k_batch_rows_count leaves
// each block's number of entries, k_batch_rows_emit sums the counts of the blocks in front of its own (N / 256 words),
// scans its own references' popcounts and writes -- PACK: the three values of every entry from the dense rows; else the
// rows themselves from the (summed) values.
constexpr u32 ROWS_BLOCK = 256;  // references per workgroup of the row kernels (2 048: 42 workgroups for 85 205 references -- 13 us a pass; 256: 333)
__global__ void __launch_bounds__(256) k_batch_rows_count(const u64* __restrict__ maskword, u64 n_refs, u32 n_planes, u32* __restrict__ blk_count) {
    __shared__ u32 part[4];
    const u64 r0 = (u64)blockIdx.x * ROWS_BLOCK;
    u32 c = 0;
    for (u32 k = threadIdx.x; k < ROWS_BLOCK; k += 256)
        if (r0 + k < n_refs)
            for (u32 pl = 0; pl < n_planes; ++pl) c += (u32)__popcll(maskword[(u64)pl * n_refs + r0 + k]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += (u32)__shfl_xor((int)c, off);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
template <bool PACK>
__global__ void __launch_bounds__(256) k_batch_rows_emit(const u64* __restrict__ maskword, u64 n_refs, u32 n_samples,
                                                         const u32* __restrict__ blk_count, u32 n_blocks,
                                                         const u32* __restrict__ overlap, const u32* __restrict__ excl,
                                                         const u32* __restrict__ match, u32* __restrict__ vals /* PACK: out, else in */,
                                                         yh_batch_row* __restrict__ rows, u64 cap, u32* __restrict__ n_rows, u32 n_planes) {
    __shared__ u32 part[4];
    __shared__ u32 wave_base[5];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    u32 before = 0;  // entries of the blocks in front of this one
    for (u32 b = threadIdx.x; b < blockIdx.x; b += 256) before += blk_count[b];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) before += (u32)__shfl_xor((int)before, off);
    if (lane == 0) part[wv] = before;
    __syncthreads();
    u32 base = part[0] + part[1] + part[2] + part[3];
    if (blockIdx.x == n_blocks - 1 && threadIdx.x == 0) *n_rows = base + blk_count[blockIdx.x];
    const u64 r0 = (u64)blockIdx.x * ROWS_BLOCK;
    for (u32 k0 = 0; k0 < ROWS_BLOCK; k0 += 256) {  // (workgroup-uniform)
        const u64 r = r0 + k0 + threadIdx.x;
        u64 wp[BATCH_MAX_PLANES];  // the reference's words, plane by plane: its entries in (reference, sample) order
        u32 c = 0;
#pragma unroll
        for (u32 pl = 0; pl < BATCH_MAX_PLANES; ++pl) {
            wp[pl] = (pl < n_planes && r < n_refs) ? maskword[(u64)pl * n_refs + r] : 0ull;
            c += (u32)__popcll(wp[pl]);
        }
        u32 inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 t = (u32)__shfl_up((int)inc, off);
            if (lane >= (u32)off) inc += t;
        }
        __syncthreads();
        if (lane == 63) wave_base[wv] = inc;
        __syncthreads();
        u32 at = base + inc - c;
        for (u32 q = 0; q < wv; ++q) at += wave_base[q];
        base += wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
#pragma unroll
        for (u32 pl = 0; pl < BATCH_MAX_PLANES; ++pl) {
        u64 w = wp[pl];
        while (w) {
            const u32 s = pl * 64u + (u32)__ffsll((long long)w) - 1u;
            w &= w - 1;
            if (at < cap && s < n_samples) {
                if (PACK) {
                    const u64 t = (u64)s * n_refs + r;
                    vals[3ull * at + 0] = overlap[t];
                    vals[3ull * at + 1] = excl[t];
                    vals[3ull * at + 2] = match[t];
                } else {
                    yh_batch_row row;
                    row.sample = s;
                    row.ref = (u32)r;
                    row.overlap = vals[3ull * at + 0];
                    row.n_excl = vals[3ull * at + 1];
                    row.n_match = vals[3ull * at + 2];
                    rows[at] = row;
                }
            }
            ++at;
        }
        }
    }
}

// ---- the subset words of a block in compact form (include/yacht_hip.h: yh_run_batch_words_*) --------------------------
// A block of 64 samples overlaps ~15 000 of 85 205 references: the dense word row a rank all-gathers (8 N bytes) is
// five-sixths zeros.  PACK: the non-zero words of a rank as (word, reference) entries behind their count --
//   packed[0] = number of non-zero words (the TRUE number, also when it exceeds cap: the reader sees the overflow),
//   packed[1 .. cap] = the words, then cap 32-bit reference ids (two per 64-bit word) --
// in no particular order (the reader ORs them into place).  UNPACK: the entries of all ranks OR-ed into one dense row
// (zeroed by the caller), *overflow = some rank had more words than its buffer carried.
// (one reserving atomic per WORKGROUP of 1 024 references: one per wave were 1 331 adds to ONE word -- 18 us of a kernel that reads 0.7 MB)
// (n_refs here = the words of ALL planes, n_planes * N: an entry's id is its index into the plane-major array)
// (Round 6: 256 lanes, four words each -- still one reserving atomic per 1 024 words.  As a 1 024-lane workgroup the kernel, on the
// finish stream beside the next block's lookups -- eight 256-lane workgroups resident on every CU --, waited for sixteen free wave
// slots on one CU: 305 us per block of 256 samples at G = 1 for a pass over 2.7 MB, profiles/r06/batch_share_G1_B256_before_pipe.txt.)
__global__ void __launch_bounds__(256) k_batch_words_pack(const u64* __restrict__ words, u64 n_refs, u64* __restrict__ packed, u64 cap) {
    __shared__ u32 wcnt[16];
    __shared__ unsigned long long s_base;
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    u64 w[4], bal[4];
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u64 r = blockIdx.x * 1024ull + k * 256u + threadIdx.x;
        w[k] = r < n_refs ? words[r] : 0ull;
    }
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        bal[k] = __ballot(w[k] != 0);
        if (lane == 0) wcnt[k * 4u + wv] = (u32)__popcll(bal[k]);
    }
    __syncthreads();
    u32 total = 0;
    for (u32 q = 0; q < 16; ++q) total += wcnt[q];
    if (total == 0) return;  // (uniform)
    if (threadIdx.x == 0) s_base = atomicAdd((unsigned long long*)packed, (unsigned long long)total);
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 4; ++k)
        if (w[k]) {
            u32 below = 0;
            for (u32 q = 0; q < k * 4u + wv; ++q) below += wcnt[q];
            const u64 at = s_base + below + __popcll(bal[k] & ((1ull << lane) - 1ull));
            if (at < cap) {
                packed[1 + at] = w[k];
                reinterpret_cast<u32*>(packed + 1 + cap)[at] = (u32)(blockIdx.x * 1024ull + k * 256u + threadIdx.x);
            }
        }
}
__global__ void __launch_bounds__(256) k_batch_words_unpack(const u64* __restrict__ gathered, u32 n_ranks, u64 cap, u64 stride,
                                                            u64 n_refs, u64* __restrict__ words, u32* __restrict__ overflow) {
    const u32 k = blockIdx.y;
    const u64* packed = gathered + (u64)k * stride;
    const u64 n = packed[0];
    if (blockIdx.x == 0 && k == 0 && threadIdx.x == 0) {
        u32 ov = 0;
        for (u32 q = 0; q < n_ranks; ++q) ov |= (u32)(gathered[(u64)q * stride] > cap);
        *overflow = ov;
    }
    const u32* refs = reinterpret_cast<const u32*>(packed + 1 + cap);
    for (u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x; e < min(n, cap); e += (u64)gridDim.x * blockDim.x) {
        const u32 r = refs[e];
        if (r < n_refs) atomicOr((unsigned long long*)&words[r], (unsigned long long)packed[1 + e]);  // (a forged id is dropped)
    }
}

}  // namespace

// a slot's scratch: maskword [4 planes][N] (+ 2) u64 | block counts of the compact rows [ceil(N / ROWS_BLOCK) + 1, padded] u32 | ovsh [B][N] u32
// (room for all BATCH_MAX_PLANES planes whatever the batch: where the parts lie does not depend on the number of samples)
static u64 batch_mask_words(const yh_db* db) { return (u64)BATCH_MAX_PLANES * db->n_refs + 2; }
static u64 batch_blk_words(const yh_db* db) { return (((db->n_refs + ROWS_BLOCK - 1) / ROWS_BLOCK + 1) + 3) & ~(u64)3; }
static u32* batch_slot_blk_counts(yh_db* db, int slot) {
    return reinterpret_cast<u32*>(reinterpret_cast<u64*>(db->batch[slot].d_scratch) + batch_mask_words(db));
}
static int batch_slot_scratch(yh_db* db, int slot, u32 n_samples, u64** d_maskword, u32** d_ovsh, u64* need_out) {
    const u64 N = db->n_refs;
    const u64 BN = (u64)n_samples * N;
    yh_db::BatchSlot& bs = db->batch[slot];
    const u64 need = batch_mask_words(db) * sizeof(u64) + batch_blk_words(db) * sizeof(u32) + BN * sizeof(u32) + 64;  // (kept per slot, grown on demand)
    if (bs.cap < need) {
        YH_HIP(hipStreamSynchronize(db->stream));
        if (bs.d_scratch) { yh_dfree(db, bs.d_scratch); bs.d_scratch = nullptr; bs.cap = 0; }
        YH_HIP(hipMalloc((void**)&bs.d_scratch, need));
        bs.cap = need;
        bs.ovsh_clean = false;
    }
    *d_maskword = reinterpret_cast<u64*>(bs.d_scratch);
    *d_ovsh = batch_slot_blk_counts(db, slot) + batch_blk_words(db);
    if (need_out) *need_out = need;
    return YH_OK;
}

int yh_q_batch_rows_pack(yh_db* db, int slot, const u32* d_overlap, const u32* d_excl, const u32* d_match, u32* d_vals, u64 cap_rows,
                         u32* d_n_rows) {
    const u64 N = db->n_refs;
    if (N == 0) { YH_HIP(hipMemsetAsync(d_n_rows, 0, sizeof(u32), db->fin_stream ? db->fin_stream : db->stream)); return YH_OK; }
    const u64* d_maskword = reinterpret_cast<const u64*>(db->batch[slot].d_scratch);
    u32* blk = batch_slot_blk_counts(db, slot);
    const u32 nblk = (u32)((N + ROWS_BLOCK - 1) / ROWS_BLOCK);
    const u32 npl = planes_of(db->batch[slot].n_samples);
    hipStream_t st = db->fin_stream ? db->fin_stream : db->stream;  // (behind the second half that made the rows)
    k_batch_rows_count<<<nblk, 256, 0, st>>>(d_maskword, N, npl, blk);
    k_batch_rows_emit<true><<<nblk, 256, 0, st>>>(d_maskword, N, db->batch[slot].n_samples, blk, nblk, d_overlap, d_excl, d_match,
                                                d_vals, nullptr, cap_rows, d_n_rows, npl);
    YH_HIP(hipGetLastError());
    if (db->fin_stream) { YH_HIP(hipEventRecord(db->ev_fin, st)); db->fin_pending = true; }
    return YH_OK;
}
int yh_q_batch_rows_unpack(yh_db* db, int slot, const u32* d_vals, u64 cap_rows, void* d_rows, u32* d_n_rows) {
    const u64 N = db->n_refs;
    if (N == 0) { YH_HIP(hipMemsetAsync(d_n_rows, 0, sizeof(u32), db->stream)); return YH_OK; }
    const u64* d_maskword = reinterpret_cast<const u64*>(db->batch[slot].d_scratch);
    u32* blk = batch_slot_blk_counts(db, slot);
    const u32 nblk = (u32)((N + ROWS_BLOCK - 1) / ROWS_BLOCK);
    const u32 npl = planes_of(db->batch[slot].n_samples);
    k_batch_rows_count<<<nblk, 256, 0, db->stream>>>(d_maskword, N, npl, blk);
    k_batch_rows_emit<false><<<nblk, 256, 0, db->stream>>>(d_maskword, N, db->batch[slot].n_samples, blk, nblk, nullptr, nullptr, nullptr,
                                                         const_cast<u32*>(d_vals), reinterpret_cast<yh_batch_row*>(d_rows), cap_rows, d_n_rows, npl);
    YH_HIP(hipGetLastError());
    return YH_OK;
}

// phases: 1 = lookup + the samples' subset words (copied to d_maskword_out when given), 2 = exclusive pass + final
// (d_gathered: the words of n_ranks hash-range shards, OR-ed into the subset first), 3 = both (one device, one call)
int yh_q_run_batch(yh_db* db, const u64* d_samples, const u64* d_soff, u32 n_samples, u64 total_hashes,
                   u32* d_overlap, u32* d_excl, u32* d_match, int phases, u64* d_maskword_out, const u64* d_gathered,
                   u32 n_ranks, int slot) {
    if (!db->has_dir || !db->has_index) {
        yh_set_error("yh_run_batch needs the directory of the distinct hashes (handle created with YH_DB_NO_DIRECTORY?)");
        return YH_ERR_UNSUPPORTED;
    }
    if (n_samples < 1 || n_samples > BATCH_MAX) { yh_set_error("1..%u samples per batch", BATCH_MAX); return YH_ERR_INVALID_ARG; }
    const u32 npl = planes_of(n_samples);  // word planes of 64 samples (d_maskword_out / d_gathered: [ranks][npl][N])
    // a second half alone goes to the finish stream when the handle has one (yh_db_set_batch_finish_stream), behind its slot's
    // first half; a first half alone then leaves the handle's subset bits to the second halves
    const bool split_streams = db->fin_stream && phases != 3;
    hipStream_t st = (split_streams && phases == 2) ? db->fin_stream : db->stream;
    const u64 N = db->n_refs;
    if (N == 0) return YH_OK;
    const u64 BN = (u64)n_samples * N;
    const u64 G = db->n_shared;
    if (G && db->n_postings && !db->d_hrec) { yh_set_error("yh_run_batch needs the holder sets of the handle"); return YH_ERR_UNSUPPORTED; }
    yh_db::BatchSlot& bs = db->batch[slot];
    u64* d_maskword = nullptr;
    u32* d_ovsh = nullptr;
    u64 need = 0;
    YH_TRY(batch_slot_scratch(db, slot, (phases & 1) ? n_samples : std::max(n_samples, bs.n_samples), &d_maskword, &d_ovsh, &need));
    if (phases & 1) { bs.n_samples = n_samples; bs.words_valid = false; }
    if (phases & 1) {
    YH_HIP(hipMemsetAsync(d_overlap, 0, BN * sizeof(u32), st));
    // (the words and block counts in front of ovsh are written whole by the kernels that make them; what lies behind them was
    // cleared when the scratch was made and is only ever cleared again)
    if (!bs.ovsh_clean) YH_HIP(hipMemsetAsync(bs.d_scratch, 0, bs.cap, st));  // (all of it: a first half without its second may have been a larger batch)
    bs.ovsh_clean = false;
    yh_ring_record_begin(db, db->ev_overlap, st);
    if (total_hashes && db->n_distinct) {
        const u64 n_tiles = (total_hashes + BATCH_TILE - 1) / BATCH_TILE + n_samples;  // (a ragged tile per sample)
        if (n_tiles >> 31) { yh_set_error("batch too large"); return YH_ERR_INVALID_ARG; }
        static const long grid_env = [] { const char* e = yh_tune_env("YH_BATCH_GRID"); return e ? atol(e) : -1L; }();
        // (workgroups of the launch, each looping over its slots: 2 048 -- the resident set -- 19.5 us per sample of a block of 256 at
        // rs214 scale, 4 096 16.5, 8 192 15.5, 16 384 14.8 (rounds 3-5), 32 768 14.2, one per slot 14.5: profiles/r06/sweep_batch_grid.txt)
        const u64 grid_cap = grid_env < 0 ? 32768ull : grid_env == 0 ? (u64)0x7fffff00 : (u64)grid_env;  // (0: one workgroup per slot)
        k_batch_lookup<<<(u32)std::min<u64>((n_tiles + 7) / 8 * 8, grid_cap), 256, 0, st>>>(d_samples, d_soff, n_samples, yh_dir_view(db),
                                                                                       db->d_po, db->d_pr, N, d_overlap, d_ovsh,
                                                                                       yh_filter_of(db), db->filter_mul);
    }
    yh_ring_record_end(db, db->ev_overlap, st);
    // With a finish stream the first stream carries nothing but the clears and the lookups: the samples' subset words (and their
    // copy for the exchange) are made on the finish stream behind the lookups' event -- the exchange they are for, the words' pack
    // and the second half all run there, and the next block's lookups start ~20 us of launch floors earlier.
    hipStream_t sw = st;
    if (split_streams) {
        YH_HIP(hipEventRecord(bs.ev_first, st));
        sw = db->fin_stream;
        YH_HIP(hipStreamWaitEvent(sw, bs.ev_first, 0));
    }
    k_batch_maskwords<<<(u32)((N + 255) / 256), 256, 0, sw>>>(d_overlap, n_samples, N, d_maskword, split_streams ? nullptr : db->d_maskbits);
    if (d_maskword_out) YH_HIP(hipMemcpyAsync(d_maskword_out, d_maskword, (u64)npl * N * sizeof(u64), hipMemcpyDeviceToDevice, sw));
    if (split_streams) { YH_HIP(hipEventRecord(db->ev_fin, sw)); db->fin_pending = true; }
    }
    if (!(phases & 2)) { YH_HIP(hipGetLastError()); return YH_OK; }
    // (a second half alone is on the finish stream behind its slot's words, which waited for the slot's lookups)
    YH_HIP(hipMemsetAsync(d_excl, 0, BN * sizeof(u32), st));
    yh_ring_record_begin(db, db->ev_excl, st);
    if (d_gathered)
        k_batch_or_maskwords<<<(u32)((N + 255) / 256), 256, 0, st>>>(d_gathered, n_ranks, N, d_maskword, db->d_maskbits, npl);
    if (G && db->n_postings) {
        YH_HIP(hipMemsetAsync(db->d_work_count, 0, sizeof(u32), st));
        k_batch_worklist<<<(u32)((N + 255) / 256), 256, 0, st>>>(N, db->d_maskbits, db->d_hpo, db->d_work, db->d_work_count);
        const u32 sets_blocks = (u32)std::min<u64>(4096, ((u64)db->n_chunks + 3) / 4 + 1);
        k_batch_sets<<<sets_blocks, 256, 0, st>>>(db->d_work, db->d_work_count, db->d_hrec, db->d_hrecx, db->d_hmult, db->d_pr,
                                                   d_maskword, N, d_excl, npl);
    }
    static const bool dense_final = [] { const char* e = yh_tune_env("YH_BATCH_DENSE_FINAL"); return e && e[0] == '1'; }();
    if (dense_final) {
        k_batch_final<<<grid_for(BN, 256, 8192), 256, 0, st>>>(n_samples, N, db->d_sizes, db->d_nshared, d_overlap, d_ovsh,
                                                               d_excl, d_match, d_gathered ? d_maskword : nullptr);
    } else {  // the entries only (the subset words are the batch's own when nothing was gathered: overlap != 0)
        YH_HIP(hipMemsetAsync(d_match, 0, BN * sizeof(u32), st));
        k_batch_final_sparse<<<(u32)((N + 255) / 256), 256, 0, st>>>(n_samples, N, npl, db->d_sizes, db->d_nshared, d_overlap, d_ovsh,
                                                                   d_excl, d_match, d_maskword);
    }
    yh_ring_record_end(db, db->ev_excl, st);
    YH_HIP(hipGetLastError());
    if (split_streams) { YH_HIP(hipEventRecord(db->ev_fin, st)); db->fin_pending = true; }
    bs.ovsh_clean = true;  // (k_batch_final, over the same [n_samples][N] the lookups counted into)
    bs.words_valid = true;  // (the slot's words are the batch's subset -- on hash-range shards the global one)
    return YH_OK;
}

// the subset words of a block as (word, reference) entries (kernels above); both on the handle's stream, no host sync
int yh_q_batch_words_pack(yh_db* db, const u64* d_words, u32 n_planes, u64* d_packed, u64 cap) {
    hipStream_t st = db->fin_stream ? db->fin_stream : db->stream;  // (where a first half leaves its words when there is a finish stream)
    const u64 nw = (u64)n_planes * db->n_refs;  // the words of all planes: an entry names its word by its index here
    YH_HIP(hipMemsetAsync(d_packed, 0, sizeof(u64), st));
    if (nw) k_batch_words_pack<<<(u32)((nw + 1023) / 1024), 256, 0, st>>>(d_words, nw, d_packed, cap);
    YH_HIP(hipGetLastError());
    return YH_OK;
}
int yh_q_batch_words_unpack(yh_db* db, const u64* d_gathered, u32 n_ranks, u32 n_planes, u64 cap, u64* d_words_out, u32* d_overflow) {
    const u64 stride = yh_batch_words_packed_len(cap);
    hipStream_t st = db->fin_stream ? db->fin_stream : db->stream;  // (the second half that reads d_words_out runs there)
    const u64 nw = (u64)n_planes * db->n_refs;
    if (nw) YH_HIP(hipMemsetAsync(d_words_out, 0, nw * sizeof(u64), st));
    const u32 gx = (u32)std::min<u64>(std::max<u64>((cap + 255) / 256, 1), 64);
    k_batch_words_unpack<<<dim3(gx, n_ranks), 256, 0, st>>>(d_gathered, n_ranks, cap, stride, nw, d_words_out, d_overflow);
    YH_HIP(hipGetLastError());
    return YH_OK;
}
