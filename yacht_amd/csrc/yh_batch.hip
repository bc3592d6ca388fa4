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
// sample hash of ANY sample.  Per-sample state is carried as 64-bit words: maskword[r] = samples that overlap
// reference r.  Exclusivity for all samples at once, over a reference's DISTINCT holder sets (k_batch_sets):
//     excl = maskword[r] & ~(OR of the other holders' words).
namespace {

constexpr int EXCL_BLOCK = 256;
inline u32 grid_for(u64 work_items, u32 block, u32 max_blocks = 16384) {
    u64 g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (u32)g;
}

__global__ void __launch_bounds__(256) k_batch_lookup(const u64* __restrict__ samples, const u64* __restrict__ soff,
                                                      u32 n_samples, const YhDirView dv, const u64* __restrict__ po,
                                                      const u32* __restrict__ pr, u64 n_refs,
                                                      u32* __restrict__ overlap /* [B][N] */, u32* __restrict__ ovsh /* [B][N]: hits on shared hashes */,
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
            u32* row2 = ovsh + (u64)s * n_refs;
            for (u64 q = po[gi], qe = po[gi + 1]; q < qe; ++q) {
                const u32 holder = pr[q];
                atomicAdd(&row[holder], 1u);
                atomicAdd(&row2[holder], 1u);
            }
        }
    }
}

// maskword[r] = samples with overlap > 0; anybits = "some sample overlaps r" (for k_batch_worklist)
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

// The work list of the exclusive pass: every reference some sample overlaps appends pieces of <= BATCH_PIECE of its
// DISTINCT holder-set records (yh_db::d_hrec, [hpo[r], hpo[r + 1])); work_count zeroed by the caller.
constexpr u32 BATCH_PIECE = 256;
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
                                                    const u64* __restrict__ maskword, u64 n_refs, u32* __restrict__ ex_e) {
    const u32 lane = threadIdx.x & 63u;
    const u32 n_work = *work_count;
    for (u32 w = blockIdx.x * 4u + (threadIdx.x >> 6); w < n_work; w += gridDim.x * 4u) {
        const uint4 piece = work[w];
        const u32 r = piece.x;
        const u64 wr = maskword[r];
        u64 excl[4];
        u32 mu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 k = piece.y + 64u * u + lane;
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

// ex_e -> n_excl in place, n_match written, for every (sample, reference).  A shared hash found in sample s has ALL its
// holders in s's subset (each of them overlaps s by that very hash), so it is exclusive to none of them: the matches are
// the hits on unshared hashes, n_match = overlap - hits on shared hashes (what the single-sample step does too).
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
            ex_m[t] = ov - ovsh[t];
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
    if (G && db->n_postings && !db->d_hrec) { yh_set_error("yh_run_batch needs the holder sets of the handle"); return YH_ERR_UNSUPPORTED; }
    // scratch: maskword [N] u64, ovsh [B][N] u32 (kept on the handle, grown on demand)
    const u64 need = BN * sizeof(u32) + (N + 2) * sizeof(u64) + 64;
    if (db->batch_cap < need) {
        YH_HIP(hipStreamSynchronize(st));
        if (db->d_batch) { yh_dfree(db, db->d_batch); db->d_batch = nullptr; db->batch_cap = 0; }
        YH_HIP(hipMalloc((void**)&db->d_batch, need));
        db->batch_cap = need;
    }
    u64* d_maskword = reinterpret_cast<u64*>(db->d_batch);
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
                                                                           db->d_po, db->d_pr, N, d_overlap, d_ovsh,
                                                                           n_chunks, mul, yh_filter_of(db), db->filter_mul);
    }
    yh_ring_record_end(db, db->ev_overlap);
    k_batch_maskwords<<<(u32)((N + 255) / 256), 256, 0, st>>>(d_overlap, n_samples, N, d_maskword, db->d_maskbits);
    if (d_maskword_out) YH_HIP(hipMemcpyAsync(d_maskword_out, d_maskword, N * sizeof(u64), hipMemcpyDeviceToDevice, st));
    }
    if (!(phases & 2)) { YH_HIP(hipGetLastError()); return YH_OK; }
    YH_HIP(hipMemsetAsync(d_excl, 0, BN * sizeof(u32), st));
    yh_ring_record_begin(db, db->ev_excl);
    if (d_gathered)
        k_batch_or_maskwords<<<(u32)((N + 255) / 256), 256, 0, st>>>(d_gathered, n_ranks, N, d_maskword, db->d_maskbits);
    if (G && db->n_postings) {
        YH_HIP(hipMemsetAsync(db->d_work_count, 0, sizeof(u32), st));
        k_batch_worklist<<<(u32)((N + 255) / 256), 256, 0, st>>>(N, db->d_maskbits, db->d_hpo, db->d_work, db->d_work_count);
        const u32 sets_blocks = (u32)std::min<u64>(4096, ((u64)db->n_chunks + 3) / 4 + 1);
        k_batch_sets<<<sets_blocks, 256, 0, st>>>(db->d_work, db->d_work_count, db->d_hrec, db->d_hrecx, db->d_hmult, db->d_pr,
                                                   d_maskword, N, d_excl);
    }
    k_batch_final<<<grid_for(BN, 256, 8192), 256, 0, st>>>(n_samples, N, db->d_sizes, db->d_nshared, d_overlap, d_ovsh,
                                                           d_excl, d_match, d_gathered ? d_maskword : nullptr);
    yh_ring_record_end(db, db->ev_excl);
    YH_HIP(hipGetLastError());
    return YH_OK;
}
