// yh_batch.hip -- batched `yacht run`: up to 64 samples against the resident database in one pass
#include "yh_common.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

// =================================================================================================
// Batched `yacht run`: up to 64 samples against the resident database in one pass
// =================================================================================================
// Samples are looked up through the distinct-hash directory (k_index_lookup's scheme), one lane per
// sample hash of ANY sample.  Per-sample state is carried as 64-bit words: hitword[g] = samples that
// contain shared hash g, maskword[r] = samples that overlap reference r.  Exclusivity of a shared
// hash for all samples at once is bit-sliced counting over its holders' mask words:
//     ones ^= w, twos |= (ones_before & w)   ->   held by exactly one masked reference = ones & ~twos.
namespace {

// Exclusive sums from the posting lists, in two launches.
//
// k_excl_collect: one coalesced pass over pr[] (four postings per lane per step, mask probes as
// BITS: N/8 bytes stay resident in every CU's L1, while random byte reads of an N-byte mask pulled
// one cache line per posting through L2).  Postings of masked references are only COLLECTED:
// appended to a per-workgroup LDS list and flushed to a queue in HBM with one atomic per workgroup.
// (~1 % of the postings belong to masked references, which is about every second wave; walking the
// dependent chain below right there left 1-2 lanes per wave busy for several microseconds.)
//
// k_excl_apply: one lane per collected posting (r holds shared hash g), all lanes busy:
//   c = masked holders of g;  c == 1 -> g is exclusive to r inside the subset (ex_e, and ex_m
//   when g is in the sample);  g in the sample -> r's "shared overlap" grows by one (ovsh).
constexpr int EXCL_BLOCK = 256;
// Workgroup b owns the contiguous vectors [b*chunk, (b+1)*chunk) of pr[] (a vector = 4 postings)
// and the queue segment that starts at posting index 4*b*chunk: even if every posting of its range
// is collected the segment cannot overflow, so there is no global counter (10^4 atomics on one word
// cost ~120 us) and no zeroing; qcount[b] is written by every workgroup.
__global__ void __launch_bounds__(EXCL_BLOCK) k_excl_collect(u64 n_post, u64 chunk, const u32* __restrict__ pr,
                                                             const u32* __restrict__ maskbits,
                                                             u32* __restrict__ queue, u32* __restrict__ qcount) {
    __shared__ u32 lq[EXCL_BLOCK * 4];
    __shared__ u32 lfill;
    if (threadIdx.x == 0) lfill = 0;
    __syncthreads();
    auto masked = [&](u32 r) -> bool { return (maskbits[r >> 5] >> (r & 31u)) & 1u; };
    const u64 n4 = n_post >> 2;
    const uint4* __restrict__ pr4 = reinterpret_cast<const uint4*>(pr);
    const u64 v_begin = (u64)blockIdx.x * chunk;
    const u64 v_end = min(n4 + 1, v_begin + chunk);  // vector n4 stands for the 0-3 trailing postings
    u32* seg = queue + 4 * v_begin;
    u32 done = 0;  // entries already flushed to seg (same value in every thread)
    for (u64 v0 = v_begin; v0 < v_end; v0 += EXCL_BLOCK) {
        const u64 v = v0 + threadIdx.x;
        if (v < v_end && v < n4) {
            const uint4 r = pr4[v];
            const bool m0 = masked(r.x), m1 = masked(r.y), m2 = masked(r.z), m3 = masked(r.w);
            const u32 cnt = (u32)m0 + (u32)m1 + (u32)m2 + (u32)m3;
            if (cnt) {
                u32 slot = atomicAdd(&lfill, cnt);
                const u32 k = (u32)(4 * v);
                if (m0) lq[slot++] = k;
                if (m1) lq[slot++] = k + 1;
                if (m2) lq[slot++] = k + 2;
                if (m3) lq[slot++] = k + 3;
            }
        } else if (v < v_end && v == n4) {
            for (u64 k = n4 << 2; k < n_post; ++k)
                if (masked(pr[k])) lq[atomicAdd(&lfill, 1u)] = (u32)k;
        }
        __syncthreads();
        const u32 f = lfill;
        for (u32 e = threadIdx.x; e < f; e += EXCL_BLOCK) seg[done + e] = lq[e];
        done += f;
        __syncthreads();
        if (threadIdx.x == 0) lfill = 0;
        __syncthreads();
    }
    if (threadIdx.x == 0) qcount[blockIdx.x] = done;
}

inline u32 grid_for(u64 work_items, u32 block, u32 max_blocks = 16384) {
    u64 g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (u32)g;
}

__global__ void __launch_bounds__(256) k_batch_lookup(const u64* __restrict__ samples, const u64* __restrict__ soff,
                                                      u32 n_samples, const YhDirView dv, const u64* __restrict__ po,
                                                      const u32* __restrict__ pr, u64 n_refs,
                                                      u32* __restrict__ overlap /* [B][N] */, u64* __restrict__ hitword,
                                                      u64 n_chunks, u64 chunk_mul, const u32* __restrict__ filter,
                                                      u64 filter_mul) {
    __shared__ u64 off[65];
    if (threadIdx.x <= n_samples) off[threadIdx.x] = soff[threadIdx.x];
    __syncthreads();
    const u64 total = off[n_samples];
    // 256-hash chunks are visited in a multiplicative permutation (chunk_mul coprime to n_chunks), so
    // that the workgroups resident at any moment work on ALL samples: a sample's hits land on its few
    // hundred present references, and same-address atomics serialize (~11 ns each on this part)
    for (u64 c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const u64 t = ((c * chunk_mul) % n_chunks) * 256 + threadIdx.x;
        if (t >= total) continue;
        u32 lo = 0, hi = n_samples;  // sample of position t: last s with off[s] <= t
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (off[mid] <= t) lo = mid; else hi = mid;
        }
        const u32 s = lo;
        const u64 h = samples[t];
        if (filter && h <= dv.max_hash) {  // presence bit first (yh_db::d_filter): clear = not in the database
            const u64 bit = yh_bucket_of(h, dv.bkt_lsh, filter_mul);
            if (!((filter[bit >> 5] >> (bit & 31u)) & 1u)) continue;
        }
        const u32 r = dv.find(h);
        if (r == YH_DIR_NONE) continue;
        u32* row = overlap + (u64)s * n_refs;
        if (!(r & 0x80000000u)) {
            atomicAdd(&row[r], 1u);
        } else {
            const u32 gi = r & 0x7fffffffu;
            atomicOr((unsigned long long*)&hitword[gi], 1ull << s);
            for (u64 q = po[gi], qe = po[gi + 1]; q < qe; ++q) atomicAdd(&row[pr[q]], 1u);
        }
    }
}

// maskword[r] = samples with overlap > 0; anybits = "some sample overlaps r" (for k_excl_collect)
__global__ void __launch_bounds__(256) k_batch_maskwords(const u32* __restrict__ overlap, u32 n_samples, u64 n_refs,
                                                         u64* __restrict__ maskword, u32* __restrict__ anybits) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    u64 w = 0;
    if (r < n_refs)
        for (u32 s = 0; s < n_samples; ++s) w |= (u64)(overlap[(u64)s * n_refs + r] != 0) << s;
    if (r < n_refs) maskword[r] = w;
    const u64 bal = __ballot(w != 0);
    if ((threadIdx.x & 63) == 0) {
        anybits[(r >> 5)] = (u32)bal;
        anybits[(r >> 5) + 1] = (u32)(bal >> 32);
    }
}

__global__ void __launch_bounds__(EXCL_BLOCK) k_batch_apply(const u32* __restrict__ queue, const u32* __restrict__ qcount,
                                                            u64 chunk, const u64* __restrict__ po,
                                                            const u32* __restrict__ pr, const u32* __restrict__ pg,
                                                            const u64* __restrict__ maskword,
                                                            const u64* __restrict__ hitword, u64 n_refs,
                                                            u32* __restrict__ ex_e, u32* __restrict__ ex_m,
                                                            u32* __restrict__ ovsh /* each [B][N] */) {
    const u32 n = qcount[blockIdx.x];
    const u32* seg = queue + 4 * (u64)blockIdx.x * chunk;
    for (u32 e = threadIdx.x; e < n; e += EXCL_BLOCK) {
        const u32 k = seg[e];
        const u32 r = pr[k];
        const u32 gi = pg[k];
        const u64 wr = maskword[r];
        const u64 hw = hitword[gi];
        u64 ones = 0, twos = 0;
        for (u64 q = po[gi], qe = po[gi + 1]; q < qe; ++q) {
            const u64 w = maskword[pr[q]];
            twos |= ones & w;
            ones ^= w;
        }
        u64 excl = ones & ~twos & wr;  // samples in which r is the only masked holder of g
        while (excl) {
            const u32 s = (u32)__ffsll((long long)excl) - 1u;
            excl &= excl - 1;
            atomicAdd(&ex_e[(u64)s * n_refs + r], 1u);
            if ((hw >> s) & 1ull) atomicAdd(&ex_m[(u64)s * n_refs + r], 1u);
        }
        u64 sh = wr & hw;  // samples that contain g and overlap r
        while (sh) {
            const u32 s = (u32)__ffsll((long long)sh) - 1u;
            sh &= sh - 1;
            atomicAdd(&ovsh[(u64)s * n_refs + r], 1u);
        }
    }
}

// hash-range shards: maskword[r] = OR over the ranks' gathered words; anybits as k_batch_maskwords makes them
__global__ void __launch_bounds__(256) k_batch_or_maskwords(const u64* __restrict__ gathered, u32 n_ranks, u64 n_refs,
                                                            u64* __restrict__ maskword, u32* __restrict__ anybits) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    u64 w = 0;
    if (r < n_refs)
        for (u32 k = 0; k < n_ranks; ++k) w |= gathered[(u64)k * n_refs + r];
    if (r < n_refs) maskword[r] = w;
    const u64 bal = __ballot(w != 0);
    if ((threadIdx.x & 63) == 0) {
        anybits[(r >> 5)] = (u32)bal;
        anybits[(r >> 5) + 1] = (u32)(bal >> 32);
    }
}

// in place: ex_e -> n_excl, ex_m -> n_match for every (sample, reference)
// (maskword != nullptr -- a hash-range shard: the subset is the global one, a reference may be in it without an overlap
// in THIS rank's range)
__global__ void __launch_bounds__(256) k_batch_final(u32 n_samples, u64 n_refs, const u32* __restrict__ sizes,
                                                     const u32* __restrict__ nshared, const u32* __restrict__ overlap,
                                                     const u32* __restrict__ ovsh, u32* __restrict__ ex_e,
                                                     u32* __restrict__ ex_m, const u64* __restrict__ maskword) {
    const u64 total = (u64)n_samples * n_refs;
    for (u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 r = t % n_refs;
        const u32 ov = overlap[t];
        const bool in = maskword ? ((maskword[r] >> (t / n_refs)) & 1ull) != 0 : ov != 0;
        if (in) {
            ex_e[t] = sizes[r] - nshared[r] + ex_e[t];
            ex_m[t] = ov - ovsh[t] + ex_m[t];
        } else {
            ex_e[t] = 0;
            ex_m[t] = 0;
        }
    }
}

}  // namespace

// phases: 1 = lookup + the samples' subset words (copied to d_maskword_out when given), 2 = exclusive pass + final
// (d_gathered: the words of n_ranks hash-range shards, OR-ed into the subset first), 3 = both (one device, one call)
int yh_q_run_batch(yh_db* db, const u64* d_samples, const u64* d_soff, u32 n_samples, u64 total_hashes,
                   u32* d_overlap, u32* d_excl, u32* d_match, int phases, u64* d_maskword_out, const u64* d_gathered,
                   u32 n_ranks) {
    if (!db->has_dir || !db->has_index) {
        yh_set_error("yh_run_batch needs the directory of the distinct hashes (handle created with YH_DB_NO_DIRECTORY?)");
        return YH_ERR_UNSUPPORTED;
    }
    if (n_samples < 1 || n_samples > 64) { yh_set_error("1..64 samples per batch"); return YH_ERR_INVALID_ARG; }
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    if (N == 0) return YH_OK;
    const u64 BN = (u64)n_samples * N;
    const u64 G = db->n_shared;
    // scratch: ovsh [B][N] u32, hitword [G] u64, maskword [N] u64 (kept on the handle, grown on demand)
    const u64 need = BN * sizeof(u32) + (G + N + 2) * sizeof(u64) + 64;
    if (db->batch_cap < need) {
        YH_HIP(hipStreamSynchronize(st));
        if (db->d_batch) { yh_dfree(db, db->d_batch); db->d_batch = nullptr; db->batch_cap = 0; }
        YH_HIP(hipMalloc((void**)&db->d_batch, need));
        db->batch_cap = need;
    }
    u64* d_hitword = reinterpret_cast<u64*>(db->d_batch);
    u64* d_maskword = d_hitword + G + 1;
    u32* d_ovsh = reinterpret_cast<u32*>(d_maskword + N + 1);
    if (phases & 1) {
    YH_HIP(hipMemsetAsync(d_overlap, 0, BN * sizeof(u32), st));
    YH_HIP(hipMemsetAsync(db->d_batch, 0, need, st));
    yh_ring_record_begin(db, db->ev_overlap);
    if (total_hashes && db->n_distinct) {
        const u64 n_chunks = (total_hashes + 255) / 256;
        if (n_chunks >> 32) { yh_set_error("batch too large"); return YH_ERR_INVALID_ARG; }
        u64 mul = (u64)((double)n_chunks * 0.6180339887) | 1;  // golden-ratio stride, made coprime
        auto gcd = [](u64 a, u64 b) { while (b) { const u64 t = a % b; a = b; b = t; } return a; };
        while (gcd(mul, n_chunks) != 1) mul += 2;
        k_batch_lookup<<<(u32)std::min<u64>(n_chunks, 8192), 256, 0, st>>>(d_samples, d_soff, n_samples, yh_dir_view(db),
                                                                           db->d_po, db->d_pr, N, d_overlap, d_hitword,
                                                                           n_chunks, mul, yh_filter_of(db), db->filter_mul);
    }
    yh_ring_record_end(db, db->ev_overlap);
    k_batch_maskwords<<<(u32)((N + 255) / 256), 256, 0, st>>>(d_overlap, n_samples, N, d_maskword, db->d_maskbits);
    if (d_maskword_out) YH_HIP(hipMemcpyAsync(d_maskword_out, d_maskword, N * sizeof(u64), hipMemcpyDeviceToDevice, st));
    }
    if (!(phases & 2)) { YH_HIP(hipGetLastError()); return YH_OK; }
    YH_HIP(hipMemsetAsync(d_excl, 0, BN * sizeof(u32), st));
    YH_HIP(hipMemsetAsync(d_match, 0, BN * sizeof(u32), st));
    yh_ring_record_begin(db, db->ev_excl);
    if (d_gathered)
        k_batch_or_maskwords<<<(u32)((N + 255) / 256), 256, 0, st>>>(d_gathered, n_ranks, N, d_maskword, db->d_maskbits);
    if (G && db->n_postings) {
        const u64 vecs = (db->n_postings >> 2) + 1;
        const u32 blocks = (u32)std::min<u64>(EXCL_QBLOCKS, (vecs + EXCL_BLOCK - 1) / EXCL_BLOCK);
        const u64 chunk = (vecs + blocks - 1) / blocks;
        k_excl_collect<<<blocks, EXCL_BLOCK, 0, st>>>(db->n_postings, chunk, db->d_pr, db->d_maskbits, db->d_pq,
                                                      db->d_pq_count);
        k_batch_apply<<<blocks, EXCL_BLOCK, 0, st>>>(db->d_pq, db->d_pq_count, chunk, db->d_po, db->d_pr, db->d_pg,
                                                     d_maskword, d_hitword, N, d_excl, d_match, d_ovsh);
    }
    k_batch_final<<<grid_for(BN, 256, 8192), 256, 0, st>>>(n_samples, N, db->d_sizes, db->d_nshared, d_overlap, d_ovsh,
                                                           d_excl, d_match, d_gathered ? d_maskword : nullptr);
    yh_ring_record_end(db, db->ev_excl);
    YH_HIP(hipGetLastError());
    return YH_OK;
}
