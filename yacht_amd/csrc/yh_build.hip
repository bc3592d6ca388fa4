// yh_build.hip — one-time database construction on the device.
//
//   (1) validation of the CSR (every reference strictly ascending), sketch sizes, the largest hash;
//   (2) ONE stable radix sort of all (hash, reference) pairs, from which everything a query reads is cut:
//       the hash-sorted delta stream (streaming lookup), the bucket table + presence filter over the distinct
//       hashes (sample-driven lookup), and the shared-hash inverted index -- distinct hashes that occur in >= 2
//       references with their posting lists: the content of the reference's `hash_index` after singletons are
//       erased (src/cpp/main.cpp:215-246) -- with its reference-major views and holder sets.
//       Replaces the per-run re-reading of N .sig files (hypothesis_recovery_src.py:93,154,168).
#include "yh_common.h"
#include "yh_sort.h"
#include "yh_pack.h"

#include <rocprim/device/device_merge.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <stdio.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <mutex>
#include <vector>

namespace {

constexpr int WAVE = 64;

// ---- validation + extent ----------------------------------------------------------------------
// One wave per reference: flag[0] |= 1 if the slice is not strictly ascending; atomicMax of the
// last element gives the database's largest hash; sizes[j] = |R_j|.
__global__ void k_scan_refs(const u64* __restrict__ values, const u64* __restrict__ offsets, u64 n_refs,
                            u32* __restrict__ sizes, u32* __restrict__ flag, u64* __restrict__ maxv) {
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) / WAVE;
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    for (u64 j = wave; j < n_refs; j += n_waves) {
        const u64 b = offsets[j], e = offsets[j + 1];
        bool bad = e < b;
        if (!bad) {
            for (u64 k = b + lane; k + 1 < e; k += WAVE)
                if (!(values[k] < values[k + 1])) bad = true;
        }
        if (__ballot(bad) != 0ull) {
            if (lane == 0) atomicOr(flag, 1u);
        }
        if (lane == 0 && !bad) {
            const u64 n = e - b;
            sizes[j] = (u32)n;
            if (n > 0xffffffffull) atomicOr(flag, 2u);
            if (n) atomicMax(maxv, values[e - 1]);
        }
    }
}

// block-wide exclusive scan of one u32 per thread (blockDim.x threads, multiple of 64, <= 1024)
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* total_out, u32* lds /* >= 17 words */) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x / WAVE;
    const int nw = blockDim.x / WAVE;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        u32 t = __shfl_up(inc, d, WAVE);
        if (lane >= d) inc += t;
    }
    if (lane == WAVE - 1) lds[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        u32 w = (lane < nw) ? lds[lane] : 0;
        u32 winc = w;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            u32 t = __shfl_up(winc, d, WAVE);
            if (lane >= d) winc += t;
        }
        if (lane < nw) lds[lane] = winc - w;  // exclusive wave prefix
        if (lane == nw - 1) lds[16] = winc;   // block total
    }
    __syncthreads();
    const u32 res = lds[wid] + inc - v;
    *total_out = lds[16];
    __syncthreads();
    return res;
}

// ---- index build ---------------------------------------------------------------------------------
// (offsets: the entries of the references [ref_base, ref_base + n_refs); ids: indexed like values)
__global__ void k_fill_ref_ids(const u64* __restrict__ offsets, u64 n_refs, u32* __restrict__ ids, u32 ref_base = 0) {
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) / WAVE;
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    for (u64 j = wave; j < n_refs; j += n_waves) {
        const u64 b = offsets[j], e = offsets[j + 1];
        for (u64 k = b + lane; k < e; k += WAVE) ids[k] = ref_base + (u32)j;
    }
}

// (tuning / tests, YH_CHECK_SORT=1) the sorted pairs: keys ascending, and ascending references inside a run of equal keys
__global__ void k_check_sorted_pairs(const u64* __restrict__ sk, const u32* __restrict__ sv, u64 n, u32* __restrict__ bad) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i + 1 < n; i += (u64)gridDim.x * blockDim.x)
        if (sk[i] > sk[i + 1] || (sk[i] == sk[i + 1] && sv[i] >= sv[i + 1])) atomicOr(bad, 1u);
}

constexpr int IDX_THREADS = 256;
constexpr int IDX_ITEMS = 16;
constexpr int IDX_BLOCK = IDX_THREADS * IDX_ITEMS;

struct IdxFlags {
    bool head;    // first element of a run of equal hashes
    bool shared;  // member of a run of length >= 2
};
__device__ __forceinline__ IdxFlags idx_flags(const u64* __restrict__ sk, u64 n, u64 i) {
    const u64 h = sk[i];
    const bool eq_prev = (i > 0) && (sk[i - 1] == h);
    const bool eq_next = (i + 1 < n) && (sk[i + 1] == h);
    IdxFlags f;
    f.head = !eq_prev;
    f.shared = eq_prev || eq_next;
    return f;
}

// One WAVE per chunk of IDX_BLOCK consecutive sorted elements, lane = element (64 at a step, coalesced);
// ranks inside the chunk are ballot prefix counts.  (The first form gave every thread 16 consecutive
// elements: 64 different cache lines per load instruction, 21 ms at 3.3e8 elements; this one ~3 ms.)
// per-chunk counts: {distinct heads, shared heads, shared elements}
__global__ void __launch_bounds__(IDX_THREADS) k_idx_count(const u64* __restrict__ sk, u64 n,
                                                            u32* __restrict__ counts /* [nb][3] */) {
    const u32 lane = threadIdx.x & 63u;
    const u64 chunk = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6;
    const u64 base = chunk * IDX_BLOCK;
    if (base >= n) return;
    u32 c0 = 0, c1 = 0, c2 = 0;
    for (u32 it = 0; it < IDX_BLOCK / 64; ++it) {
        const u64 i = base + (u64)it * 64 + lane;
        bool head = false, shared = false;
        if (i < n) {
            const IdxFlags f = idx_flags(sk, n, i);
            head = f.head;
            shared = f.shared;
        }
        c0 += (u32)__popcll(__ballot(head));
        c1 += (u32)__popcll(__ballot(head && shared));
        c2 += (u32)__popcll(__ballot(shared));
    }
    if (lane == 0) {
        counts[chunk * 3 + 0] = c0;
        counts[chunk * 3 + 1] = c1;
        counts[chunk * 3 + 2] = c2;
    }
}

// single workgroup: exclusive scan of the per-block counts into 64-bit bases, totals at [nb]
__global__ void __launch_bounds__(1024) k_idx_scan_counts(const u32* __restrict__ counts, u64 nb,
                                                          u64* __restrict__ bases /* [nb+1][3] */) {
    __shared__ u32 lds[17];
    u64 carry[3] = {0, 0, 0};
    for (u64 base = 0; base < nb; base += blockDim.x) {
        const u64 b = base + threadIdx.x;
        for (int c = 0; c < 3; ++c) {
            const u32 v = (b < nb) ? counts[b * 3 + c] : 0u;
            u32 tot;
            const u32 ex = block_excl_scan(v, &tot, lds);
            if (b < nb) bases[b * 3 + c] = carry[c] + ex;
            carry[c] += tot;
        }
    }
    if (threadIdx.x == 0) {
        bases[nb * 3 + 0] = carry[0];
        bases[nb * 3 + 1] = carry[1];
        bases[nb * 3 + 2] = carry[2];
    }
}

// Write g[], po[], pr[], pg[] and nshared[] (and dh / dref / elem_g).  Same wave-per-chunk walk as
// k_idx_count; an element's rank = the chunk's base (bases[]) + what the wave has seen so far + the flagged
// lanes below it.
__global__ void __launch_bounds__(IDX_THREADS) k_idx_emit(const u64* __restrict__ sk, const u32* __restrict__ sv,
                                                          u64 n, const u64* __restrict__ bases,
                                                          u64* __restrict__ g, u64* __restrict__ po,
                                                          u32* __restrict__ pr, u32* __restrict__ pg,
                                                          u32* __restrict__ nshared,
                                                          u64* __restrict__ dh, u32* __restrict__ dref,
                                                          u32* __restrict__ elem_g, u32* __restrict__ prank,
                                                          const u64* __restrict__ chunk_off, u64 n_chunks) {
    // dh/dref (optional): every DISTINCT hash ascending, with its single holder, or
    // 0x80000000 | (index into g) when several references hold it
    // chunk_off (optional): the chunks are the buckets of the distribution sort, [chunk_off[c], chunk_off[c + 1]), whose
    // three counts the sort made itself; else chunks of IDX_BLOCK elements counted by k_idx_count
    const u32 lane = threadIdx.x & 63u;
    const u64 chunk = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6;
    u64 base, end;
    if (chunk_off) {
        if (chunk >= n_chunks) return;
        base = chunk_off[chunk];
        end = chunk_off[chunk + 1];
    } else {
        base = chunk * IDX_BLOCK;
        if (base >= n) return;
        end = min(n, base + (u64)IDX_BLOCK);
    }
    u64 di = bases[chunk * 3 + 0];  // distinct hashes before this step's elements
    u64 gi = bases[chunk * 3 + 1];  // shared heads
    u64 mi = bases[chunk * 3 + 2];  // shared elements
    const u64 below = (1ull << lane) - 1ull;
    for (u64 i0 = base; i0 < end; i0 += 64) {  // (wave-uniform)
        const u64 i = i0 + lane;
        bool head = false, shared = false;
        u64 h = 0;
        u32 r = 0;
        if (i < end) {
            const IdxFlags f = idx_flags(sk, n, i);
            head = f.head;
            shared = f.shared;
            h = sk[i];
            r = sv[i];
        }
        const u64 bd = __ballot(head), bg = __ballot(head && shared), bm = __ballot(shared);
        const u64 my_d = di + (u64)__popcll(bd & below);
        // shared heads up to and including this lane: the g index of the run this element belongs to
        const u64 my_g_incl = gi + (u64)__popcll(bg & (below | (1ull << lane)));
        const u64 my_m = mi + (u64)__popcll(bm & below);
        if (head && shared) {
            g[my_g_incl - 1] = h;
            po[my_g_incl - 1] = my_m;
        }
        if (dh && head) {
            dh[my_d] = h;
            dref[my_d] = shared ? (0x80000000u | (u32)(my_g_incl - 1)) : r;
        }
        if (shared) {
            pr[my_m] = r;
            pg[my_m] = (u32)(my_g_incl - 1);
            const u32 rank = atomicAdd(&nshared[r], 1u);
            if (prank) prank[my_m] = rank;  // (any order inside the reference: the pairwise pass only sums)
        }
        if (elem_g && i < end) elem_g[i] = shared ? (u32)(my_g_incl - 1) : STREAM_NONE;  // shared-hash index of every sorted element
        di += (u64)__popcll(bd);
        gi += (u64)__popcll(bg);
        mi += (u64)__popcll(bm);
    }
}

// ---- hash-sorted delta stream (layout: yh_common.h) ------------------------------------------------
// c[i] = 1 + fillers in front of sorted element i; an inclusive scan of c gives position + 1.
struct StreamCount {
    const u64* sk;
    u32 sshift;
    __device__ u64 operator()(u64 i) const {
        if (i == 0) return 1;
        const u64 d = (sk[i] >> sshift) - (sk[i - 1] >> sshift);
        return 1 + (d ? (d - 1) / 255 : 0);
    }
};
__global__ void k_stream_scatter(const u64* __restrict__ sk, const u32* __restrict__ sv, const u32* __restrict__ elem_g,
                                 u64 n, u32 sshift, const u64* __restrict__ pos1 /* position + 1 */,
                                 u8* __restrict__ sdelta, uint2* __restrict__ srec, u64* __restrict__ hdr) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 h = sk[i], t = h >> sshift, pos = pos1[i] - 1;
        u32 own = 0;
        if (i) {
            const u64 tp = sk[i - 1] >> sshift, ppos = pos1[i - 1] - 1;
            const u64 nfill = pos - ppos - 1;
            for (u64 f = 1; f <= nfill; ++f) {  // (records are pre-set: no reference)
                const u64 q = ppos + f;
                sdelta[q] = 255;
                if ((q & (STREAM_BLOCK - 1)) == 0) hdr[q >> 10] = tp + 255 * f;
            }
            own = (u32)(t - tp - 255 * nfill);
        }
        sdelta[pos] = (u8)own;
        // the hash's low 32 bits (the stream key carries the bits from sshift <= 32 up) and its reference, top bit = the
        // hash is one of the database-shared ones
        srec[pos] = make_uint2((u32)h, sv[i] | ((elem_g && elem_g[i] != STREAM_NONE) ? 0x80000000u : 0u));
        if ((pos & (STREAM_BLOCK - 1)) == 0) hdr[pos >> 10] = t;
    }
}

// Directory over the distinct hashes: dir[b] = first index whose bucket (hash >> dshift) is >= b.
__global__ void k_dir_build(const u64* __restrict__ dh, u64 D, u32 dshift, u32 NB, u32* __restrict__ dir) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < D; i += (u64)gridDim.x * blockDim.x) {
        const u64 b = dh[i] >> dshift;
        const long long bp = i ? (long long)(dh[i - 1] >> dshift) : -1;
        for (long long x = bp + 1; x <= (long long)b; ++x) dir[x] = (u32)i;
        if (i == D - 1)
            for (u64 x = b + 1; x <= NB; ++x) dir[x] = (u32)D;
    }
}

// Bucket table over the distinct hashes (layout: YhDirView in yh_common.h).  One thread per distinct
// hash: its slot is the number of predecessors in the same bucket (buckets are contiguous runs of
// the sorted array); the last hash of a run writes the bucket's header.  The table is zeroed first.
__global__ void k_bkt_build(const u64* __restrict__ dh, const u32* __restrict__ dref, u64 D, u32 lsh, u64 nb,
                            u32* __restrict__ bkt) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < D; i += (u64)gridDim.x * blockDim.x) {
        const u64 h = dh[i];
        const u64 b = yh_bucket_of(h, lsh, nb);
        u32 r = 0;
        while (r < 6 && i > r && yh_bucket_of(dh[i - 1 - r], lsh, nb) == b) ++r;
        u32* w = bkt + 16 * b;
        if (r < 5) {
            w[2 * r] = (u32)h;
            w[2 * r + 1] = (u32)(h >> 32);
            w[10 + r] = dref[i];
        }
        if (i + 1 == D || yh_bucket_of(dh[i + 1], lsh, nb) != b) w[15] = (r + 1 > 5) ? YH_BKT_OVERFLOW : r + 1;
    }
}

// ---- compact bucket table (layout: YhDirView) -------------------------------------------------------------
// rank of distinct hash i inside its bucket = predecessors with the same bucket (buckets are runs of the sorted array)
__device__ __forceinline__ u32 bucket_rank(const u64* __restrict__ dh, u64 i, u64 b, u32 lsh, u64 nb) {
    u32 r = 0;
    while (r < 64 && i > r && yh_bucket_of(dh[i - 1 - r], lsh, nb) == b) ++r;
    return r;
}
__global__ void k_cbkt_count_overflow(const u64* __restrict__ dh, u64 D, u32 lsh, u64 nb, u64* __restrict__ n_over) {
    u32 mine = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < D; i += (u64)gridDim.x * blockDim.x)
        mine += bucket_rank(dh, i, yh_bucket_of(dh[i], lsh, nb), lsh, nb) >= 7 ? 1u : 0u;
    if (mine) atomicAdd(n_over, (u64)mine);
}
// The table is zeroed first; the overflow table's values are pre-set to YH_DIR_NONE (empty).
__global__ void k_cbkt_build(const u64* __restrict__ dh, const u32* __restrict__ dref, u64 D, u32 lsh, u64 nb,
                             u32* __restrict__ bkt, u64* __restrict__ ovf_keys, u32* __restrict__ ovf_vals, u32 ovf_mask) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < D; i += (u64)gridDim.x * blockDim.x) {
        const u64 h = dh[i];
        const u64 b = yh_bucket_of(h, lsh, nb);
        const u32 r = bucket_rank(dh, i, b, lsh, nb);
        u32* w = bkt + 16 * b;
        if (r < 7) {
            w[r] = (u32)h;
            w[8 + r] = dref[i];
        } else {  // claim an empty slot (all keys are distinct: nothing to compare), then publish the key
            for (u32 s = (u32)yh_ovf_slot(h) & ovf_mask;; s = (s + 1) & ovf_mask)
                if (atomicCAS(&ovf_vals[s], YH_DIR_NONE, dref[i]) == YH_DIR_NONE) {
                    ovf_keys[s] = h;
                    break;
                }
        }
        if (i + 1 == D || yh_bucket_of(dh[i + 1], lsh, nb) != b) w[7] = (r + 1 > 7) ? (7u | YH_CBKT_OVERFLOW) : r + 1;
    }
}

// ---- holder sets (layout: yh_common.h, d_hrec) -----------------------------------------------------------
// Many shared hashes of a reference have the SAME other holders (a cluster of genomes: at most 2^(k-1) different
// sets, thousands of hashes), and "none of the other holders is in the subset" has the same answer for all of them.
// So the fused run step does not walk a reference's postings but its DISTINCT holder sets with their
// multiplicities.  Built by sorting the reference-major postings by (reference, 64-bit hash of the holder record)
// and cutting runs where the full 32-byte records differ -- a hash collision only splits a run, never merges two.
__device__ __forceinline__ u64 set_mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_set_keys(u64 n_post, const uint4* __restrict__ rrec, const uint4* __restrict__ rrecx,
                           u64* __restrict__ key, u32* __restrict__ idx) {
    for (u64 p = blockIdx.x * (u64)blockDim.x + threadIdx.x; p < n_post; p += (u64)gridDim.x * blockDim.x) {
        const uint4 a = rrec[p], b = rrecx[p];
        u64 h = set_mix(((u64)a.x << 32) | a.y);
        h = set_mix(h ^ (((u64)a.z << 32) | a.w));
        h = set_mix(h ^ (((u64)b.x << 32) | b.y));
        h = set_mix(h ^ (((u64)b.z << 32) | b.w));
        key[p] = h;
        idx[p] = (u32)p;
    }
}
__global__ void k_gather_u32(u64 n, const u32* __restrict__ src, const u32* __restrict__ idx, u32* __restrict__ out) {
    for (u64 s = blockIdx.x * (u64)blockDim.x + threadIdx.x; s < n; s += (u64)gridDim.x * blockDim.x) out[s] = src[idx[s]];
}
__device__ __forceinline__ bool same16(const uint4 a, const uint4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
// head[s] = 1 when sorted posting s starts a new (reference, holder set) run
__global__ void k_set_heads(u64 n_post, const u32* __restrict__ idx, const u32* __restrict__ pref,
                            const uint4* __restrict__ rrec, const uint4* __restrict__ rrecx, u32* __restrict__ head) {
    for (u64 s = blockIdx.x * (u64)blockDim.x + threadIdx.x; s < n_post; s += (u64)gridDim.x * blockDim.x) {
        bool h = s == 0;
        if (!h) {
            const u32 p = idx[s], q = idx[s - 1];
            h = pref[p] != pref[q] || !same16(rrec[p], rrec[q]) || !same16(rrecx[p], rrecx[q]);
        }
        head[s] = h ? 1u : 0u;
    }
}
// run[s] = inclusive scan of head (1-based run number of sorted posting s)
__global__ void k_set_emit(u64 n_post, const u32* __restrict__ idx, const u32* __restrict__ pref, const u32* __restrict__ head,
                           const u32* __restrict__ run, const uint4* __restrict__ rrec, const uint4* __restrict__ rrecx,
                           uint4* __restrict__ hrec, uint4* __restrict__ hrecx, u32* __restrict__ hmult, u32* __restrict__ hcnt) {
    for (u64 s = blockIdx.x * (u64)blockDim.x + threadIdx.x; s < n_post; s += (u64)gridDim.x * blockDim.x) {
        const u32 r = run[s] - 1u, p = idx[s];
        atomicAdd(&hmult[r], 1u);
        if (head[s]) {
            hrec[r] = rrec[p];
            hrecx[r] = rrecx[p];
            atomicAdd(&hcnt[pref[p]], 1u);
        }
    }
}

// ---- reference-major chunk view of the postings ---------------------------------------------------
__global__ void k_chunk_counts(const u32* __restrict__ nshared, u64 n, u32* __restrict__ cc) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (r < n) cc[r] = (nshared[r] + (u32)(YH_EXCL_PIECE - 1)) / (u32)YH_EXCL_PIECE;
}
// rrec / rrecx (layout: yh_common.h): the OTHER holders of the posting's hash next to it, so that the
// fused run step reaches them in one (coalesced) read instead of three dependent ones (rg -> po -> pr).
__global__ void k_fill_rg(u64 n_post, const u32* __restrict__ pr, const u32* __restrict__ pg, const u64* __restrict__ po,
                          const u32* __restrict__ rpo, u32* __restrict__ cursor, u32* __restrict__ rg,
                          uint4* __restrict__ rrec, uint4* __restrict__ rrecx, u32* __restrict__ pref) {
    for (u64 k = blockIdx.x * (u64)blockDim.x + threadIdx.x; k < n_post; k += (u64)gridDim.x * blockDim.x) {
        const u32 r = pr[k], g = pg[k];
        const u32 dst = rpo[r] + atomicAdd(&cursor[r], 1u);  // order inside a reference is irrelevant (sums)
        rg[dst] = g;
        if (pref) pref[dst] = r;
        if (rrec) {
            const u64 q0 = po[g], q1 = po[g + 1];
            uint4 rec, recx = make_uint4(0u, 0u, 0u, 0u);
            if (q1 - q0 <= 8) {  // up to seven other holders: inline, the first three in rrec, the rest in rrecx
                u32 o[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
                u32 n = 0;
                for (u64 q = q0; q < q1; ++q)
                    if (q != k) o[n++] = pr[q];
                rec = make_uint4(o[0], o[1], o[2], n);
                recx = make_uint4(o[3], o[4], o[5], o[6]);
            } else {             // a longer list: where it is
                rec = make_uint4((u32)q0, (u32)(q1 - q0), 0u, 0xffffffffu);
            }
            rrec[dst] = rec;
            rrecx[dst] = recx;
        }
    }
}
// presence filter of the distinct hashes (yh_db::d_filter): bit floor(h * n_bits / (max_hash + 1))
__global__ void k_filter_build(const u64* __restrict__ dh, u64 n, u32 lsh, u64 fmul, u32* __restrict__ filter) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 bit = __umul64hi(dh[i] << lsh, fmul);
        atomicOr(&filter[bit >> 5], yh_filter_mask(dh[i], bit));
    }
}

inline u32 grid_for(u64 work_items, u32 block, u32 max_blocks = 16384) {
    u64 g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (u32)g;
}

}  // namespace

// =================================================================================================
static int validate_begin(yh_db* db) {
    const u64 N = db->n_refs;
    YH_TRY(yh_dmalloc(db, (void**)&db->d_sizes, std::max<u64>(N, 1) * sizeof(u32)));
    YH_TRY(yh_dmalloc(db, (void**)&db->d_flag, 32));  // flag | - | largest hash (8 B) | offsets[0] | offsets[n_refs] (k_ref_extents)
    YH_HIP(hipMemsetAsync(db->d_flag, 0, 32, db->stream));
    YH_HIP(hipMemsetAsync(db->d_sizes, 0, std::max<u64>(N, 1) * sizeof(u32), db->stream));
    return YH_OK;
}
static int validate_refs(yh_db* db, const u64* d_values, const u64* d_offsets, u64 r0, u64 r1) {
    if (r1 > r0) {
        u64* d_maxv = (u64*)(db->d_flag + 2);  // 8-byte aligned slot inside the 16-byte scratch
        k_scan_refs<<<grid_for((r1 - r0) * WAVE, 256), 256, 0, db->stream>>>(d_values, d_offsets + r0, r1 - r0, db->d_sizes + r0,
                                                                            db->d_flag, d_maxv);
        YH_HIP(hipGetLastError());
    }
    return YH_OK;
}
static int validate_end(yh_db* db, bool ends = false) {
    u32 hstack[8];
    // (a `yacht train` handle: the sketch sizes come along -- yh_pairwise's exact filter reads them on the host, and fetching
    // them there was one more copy behind its kernel)
    const bool want_sizes = (db->flags & YH_DB_PAIRWISE_ONLY) && db->n_refs && db->h_sizes.empty();
    YhPin pin(32 + (want_sizes ? db->n_refs * sizeof(u32) : 0));
    u32* hflag = pin.p ? static_cast<u32*>(pin.p) : hstack;
    YH_HIP(hipMemcpyAsync(hflag, db->d_flag, 32, hipMemcpyDeviceToHost, db->stream));
    if (want_sizes && pin.p) YH_HIP(hipMemcpyAsync(hflag + 8, db->d_sizes, db->n_refs * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipStreamSynchronize(db->stream));
    if (want_sizes && pin.p) {
        db->h_sizes.assign(hflag + 8, hflag + 8 + db->n_refs);
        u32 mx = 0;
        for (const u32 v : db->h_sizes) mx = std::max(mx, v);
        db->max_ref_size = mx;
    }
    if (ends) {  // the CSR's first and last offset, read by k_ref_extents (a device CSR: the host has not seen them)
        u64 first, last;
        memcpy(&first, &hflag[4], 8);
        memcpy(&last, &hflag[6], 8);
        if (first != 0) { yh_set_error("offsets[0] must be 0"); return YH_ERR_INVALID_ARG; }
        if (!(hflag[0] & 1u)) db->n_hashes = last;
    }
    if (hflag[0] & 4u) {  // (first: what such a block should have written is whatever the buffer held)
        yh_set_error("packed CSR: a block points outside the payload or does not fit the offsets");
        return YH_ERR_INVALID_ARG;
    }
    if (hflag[0] & 1u) {
        yh_set_error("a reference sketch is not strictly ascending (or offsets are not monotone)");
        return YH_ERR_UNSORTED;
    }
    if (hflag[0] & 2u) {
        yh_set_error("a reference sketch has more than 2^32-1 hashes");
        return YH_ERR_INVALID_ARG;
    }

    u64 maxv;
    memcpy(&maxv, &hflag[2], 8);
    db->max_hash = maxv;
    return YH_OK;
}
int yh_build_validate(yh_db* db, const u64* d_values, const u64* d_offsets) {
    YH_TRY(validate_begin(db));
    YH_TRY(validate_refs(db, d_values, d_offsets, 0, db->n_refs));
    YH_TRY(validate_end(db));
    db->order_checked = true;
    return YH_OK;
}
// The same in two parts, for a database the distribution sort is going to read anyway: the sizes and the largest hash from
// the offsets and every sketch's LAST element (N reads; the largest hash of ascending sketches is the largest last element,
// and sketches that are not ascending fail the ordering check whatever this says) -- and the ordering check itself, which
// the sort's first level does on its way through (yh_psort_check_order) or, without that sort, yh_build_check_order.
__global__ void k_ref_extents(const u64* __restrict__ values, const u64* __restrict__ offsets, u64 n_refs, u32* __restrict__ sizes,
                              u32* __restrict__ flag, u64* __restrict__ maxv) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { maxv[1] = offsets[0]; maxv[2] = offsets[n_refs]; }
    for (u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x; j < n_refs; j += (u64)gridDim.x * blockDim.x) {
        const u64 b = offsets[j], e = offsets[j + 1];
        if (e < b) { atomicOr(flag, 1u); continue; }
        const u64 n = e - b;
        sizes[j] = (u32)n;
        if (n > 0xffffffffull) atomicOr(flag, 2u);
        if (n) atomicMax(maxv, values[e - 1]);
    }
}
int yh_build_validate_extents(yh_db* db, const u64* d_values, const u64* d_offsets) {
    YH_TRY(validate_begin(db));
    if (db->n_refs) {
        u64* d_maxv = (u64*)(db->d_flag + 2);
        k_ref_extents<<<grid_for(db->n_refs, 256), 256, 0, db->stream>>>(d_values, d_offsets, db->n_refs, db->d_sizes, db->d_flag, d_maxv);
        YH_HIP(hipGetLastError());
    }
    YH_TRY(validate_end(db, /*ends=*/db->n_refs > 0));
    db->order_checked = false;
    return YH_OK;
}
int yh_build_check_order(yh_db* db, const u64* d_values, const u64* d_offsets) {
    // (k_scan_refs again: it rewrites the same sizes and can only raise the same maximum)
    YH_HIP(hipMemsetAsync(db->d_flag, 0, 4, db->stream));
    YH_TRY(validate_refs(db, d_values, d_offsets, 0, db->n_refs));
    u32 hflag = 0;
    YH_HIP(hipMemcpyAsync(&hflag, db->d_flag, 4, hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipStreamSynchronize(db->stream));
    if (hflag & 1u) {
        yh_set_error("a reference sketch is not strictly ascending (or offsets are not monotone)");
        return YH_ERR_UNSORTED;
    }
    return YH_OK;
}

// Host CSR -> device CSR, validated, and the (hash, reference) pairs of the whole database sorted by hash, with the
// UPLOAD OVERLAPPED WITH THE SORT: the references go up in a few chunks on a stream of their own; while chunk c + 1
// crosses the bus, chunk c is checked (k_scan_refs) and DISTRIBUTED over the first-level regions of the sort
// (yh_sort.hip: yh_psort_add -- the regions are filled by atomically reserved runs, so pieces can arrive in any number of
// calls).  Behind the last byte remain the last chunk's distribution, the second level and the bucket sorts.  (Rounds 2-3
// sorted every chunk with rocPRIM and merged it into the sorted prefix with rocprim::merge.)  Keys that are not uniform
// enough for the distribution sort (a capacity is exceeded on the device) are sorted by rocPRIM behind the upload instead.
// *d_sk_out / *d_sv_out are the caller's to yh_tfree.  (The largest hash is needed before the first chunk is distributed:
// for ascending sketches it is the largest LAST element, which the host reads off the offsets; sketches that are not
// ascending fail the check anyway.)
static double trace_now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static bool trace_on() {
    static const bool on = [] { const char* e = yh_tune_env("YH_TRACE_BUILD"); return e && e[0] == '1'; }();
    return on;
}
#define TRACE(label) do { if (trace_on()) { const double t__ = trace_now(); fprintf(stderr, "[yh build] %-28s +%.3f ms\n", label, t__ - t_prev); t_prev = t__; } } while (0)

// ---- the fused path of a YH_DB_PAIRWISE_ONLY handle (yh_db::fz; yh_sort.hip: k_bucket_group) ---------------------
// `yacht train` wants nothing of the index but the pairwise pass's input.  Where the distribution sort takes the pairs,
// they carry their CSR POSITION instead of the reference id, and the sort's last pass -- which has every run of equal
// hashes whole and in order in LDS -- writes the 8-byte record "the other holders of this element's hash" straight to the
// element's position: a reference's records are the extent of its sketch.  Gone: the sorted pairs themselves (12 B written
// and read again), k_idx_emit with its counting atomic per posting, k_pair_transpose with its isolated store per posting,
// the posting arrays, and the host's prefix sum over nshared[] in yh_pairwise.
static bool fz_wanted(const yh_db* db) {
    static const bool off = [] { const char* e = yh_tune_env("YH_NO_FUSED_TRAIN"); return e && e[0] == '1'; }();
    return !off && (db->flags & YH_DB_PAIRWISE_ONLY) && !(db->flags & YH_DB_NO_INDEX) && db->n_refs > 0 && db->n_hashes > 0;
}
static void fz_drop(yh_db* db) {
    const u64 N = db->n_refs, H = db->n_hashes;
    if (db->d_fz_off) db->device_bytes -= (N + 1) * sizeof(u64);
    if (db->d_fz_tab) db->device_bytes -= ((H >> YH_REF_TAB_SH) + 2) * sizeof(u32);
    if (db->d_fz_rec) db->device_bytes -= H * sizeof(u64);
    yh_dfree(db, db->d_fz_off); yh_dfree(db, db->d_fz_tab); yh_dfree(db, db->d_fz_rec);
    db->d_fz_off = nullptr; db->d_fz_tab = nullptr; db->d_fz_rec = nullptr;
}
// the handle's copy of the offsets, the position -> reference look-up and the (still unwritten) records; then the sort in
// position mode
static int fz_begin(yh_db* db, const u64* d_offsets, u64 max_hash, yh_psort** ps) {
    const u64 N = db->n_refs, H = db->n_hashes;
    *ps = nullptr;
    int rc = yh_dmalloc(db, (void**)&db->d_fz_off, (N + 1) * sizeof(u64));
    if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_fz_tab, ((H >> YH_REF_TAB_SH) + 2) * sizeof(u32));
    if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_fz_rec, H * sizeof(u64));
    if (rc == YH_OK && hipMemcpyAsync(db->d_fz_off, d_offsets, (N + 1) * sizeof(u64), hipMemcpyDeviceToDevice, db->stream) != hipSuccess) {
        yh_set_error("copy of the offsets failed");
        rc = YH_ERR_HIP;
    }
    if (rc == YH_OK) rc = yh_ref_table_build(db, db->d_fz_off, N, H, db->d_fz_tab);
    if (rc == YH_OK) rc = yh_psort_begin(db, H, max_hash, ps);
    if (rc == YH_OK) yh_psort_positions(*ps, db->d_fz_tab, db->d_fz_off, db->n_refs, db->d_fz_rec);
    if (rc != YH_OK) { yh_psort_destroy(db, *ps); *ps = nullptr; fz_drop(db); }
    return rc;
}
// the sort's second level + fused last pass; *took = false: not this sort's keys, everything of the attempt is released
static int fz_finish(yh_db* db, yh_psort* ps, bool* took, bool* unsorted) {
    u64 totals[3] = {0, 0, 0};
    u32* d_list = nullptr;
    int rc = yh_psort_finish_emit(db, ps, db->d_fz_rec, db->n_refs, totals, &d_list, took, unsorted);
    yh_psort_destroy(db, ps);
    if (rc != YH_OK || !*took) { fz_drop(db); return rc; }
    db->d_fz_list = d_list;
    db->fz_list_split = ~(u64)0;
    db->sort_path = YH_SORT_TWO_LEVEL;
    db->n_distinct = totals[0];
    db->n_shared = totals[1];
    db->n_postings = totals[2];
    db->fz = true;
    return YH_OK;
}

// The same through the distribution WITHOUT a first level (yh_sort.hip: k_piece_bounds / k_piece_part): no position ->
// reference table (the pairs bring their reference), the bounds pass clears the records.
static int fzp_begin(yh_db* db, const u64* d_offsets, u64 max_hash, yh_pieces** pc) {
    const u64 N = db->n_refs, H = db->n_hashes;
    *pc = nullptr;
    int rc = yh_dmalloc(db, (void**)&db->d_fz_off, (N + 1) * sizeof(u64));
    if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_fz_rec, H * sizeof(u64));
    if (rc == YH_OK && hipMemcpyAsync(db->d_fz_off, d_offsets, (N + 1) * sizeof(u64), hipMemcpyDeviceToDevice, db->stream) != hipSuccess) {
        yh_set_error("copy of the offsets failed");
        rc = YH_ERR_HIP;
    }
    if (rc == YH_OK) rc = yh_pc_begin(db, H, max_hash, N, db->d_fz_rec, pc);
    if (rc != YH_OK) { yh_pc_destroy(db, *pc); *pc = nullptr; fz_drop(db); }
    return rc;
}
int yh_radix_sort_pairs_u64_u32(yh_db* db, const u64* k_in, u64* k_out, const u32* v_in, u32* v_out, u64 n, unsigned end_bit) {
    if (n == 0) return YH_OK;
    size_t tb = 0;
    void* d_tmp = nullptr;
    end_bit = std::min(std::max(end_bit, 1u), 64u);
    YH_HIP(rocprim::radix_sort_pairs(nullptr, tb, k_in, k_out, v_in, v_out, (size_t)n, 0u, end_bit, db->stream));
    if (yh_tmalloc(db, &d_tmp, std::max<size_t>(tb, 16)) != hipSuccess) { yh_set_error("out of device memory"); return YH_ERR_OOM; }
    const hipError_t e = rocprim::radix_sort_pairs(d_tmp, tb, k_in, k_out, v_in, v_out, (size_t)n, 0u, end_bit, db->stream);
    yh_tfree(db, d_tmp);
    if (e != hipSuccess) { yh_set_error("radix sort failed: %s", hipGetErrorString(e)); return YH_ERR_HIP; }
    return YH_OK;
}
static int fzp_finish(yh_db* db, yh_pieces* pc, const u64* d_values, bool* took, bool* unsorted) {
    u64 totals[3] = {0, 0, 0};
    u32* d_list = nullptr;
    yh_pc_spill spill;
    int rc = yh_pc_finish_emit(db, pc, d_values, db->d_fz_off, totals, &d_list, took, unsorted, &spill);
    yh_pc_destroy(db, pc);
    if (rc != YH_OK || !*took) { yh_tfree(db, spill.d_list2); fz_drop(db); return rc; }
    db->d_fz_list2 = spill.d_list2;
    db->fz_list_split = spill.d_list2 ? spill.list_split : ~(u64)0;
    db->n_spilled_pairs = spill.n_pairs;
    db->n_spilled_buckets = spill.n_buckets;
    db->sort_path = YH_SORT_PIECES;
    db->d_fz_list = d_list;
    db->n_distinct = totals[0];
    db->n_shared = totals[1];
    db->n_postings = totals[2];
    db->fz = true;
    return YH_OK;
}

// (pk: the database arrives PACKED -- yh_csr_pack's blob, ~5.7 instead of 8 bytes per hash on the bus -- and every chunk is
// expanded into d_values by k_unpack_csr in front of its ordering check; h_values is not read then)
static int upload_sorted_impl(yh_db* db, const u64* h_values, const u64* h_offsets, const YhPackedCsr* pk, u64* d_values, const u64* d_offsets,
                              u64** d_sk_out, u32** d_sv_out);
int yh_build_upload_sorted(yh_db* db, const u64* h_values, const u64* h_offsets, u64* d_values, const u64* d_offsets,
                           u64** d_sk_out, u32** d_sv_out) {
    return upload_sorted_impl(db, h_values, h_offsets, nullptr, d_values, d_offsets, d_sk_out, d_sv_out);
}
int yh_build_upload_sorted_packed(yh_db* db, const YhPackedCsr* pk, u64* d_values, const u64* d_offsets, u64** d_sk_out, u32** d_sv_out) {
    return upload_sorted_impl(db, nullptr, pk->offsets, pk, d_values, d_offsets, d_sk_out, d_sv_out);
}
static int upload_sorted_impl(yh_db* db, const u64* h_values, const u64* h_offsets, const YhPackedCsr* pk, u64* d_values, const u64* d_offsets,
                              u64** d_sk_out, u32** d_sv_out) {
    const u64 N = db->n_refs, H = db->n_hashes;
    hipStream_t st = db->stream;
    *d_sk_out = nullptr;
    *d_sv_out = nullptr;
    // chunk boundaries: shares of the hashes, moved to reference boundaries
    static const std::vector<double> shares = [] {
        std::vector<double> v;
        if (const char* e = yh_tune_env("YH_UPLOAD_SHARES")) {  // e.g. "0.4,0.3,0.2,0.1"
            for (const char* q = e; *q;) { v.push_back(atof(q)); while (*q && *q != ',') ++q; if (*q) ++q; }
        }
        if (v.empty()) v = {0.4, 0.3, 0.2, 0.1};
        return v;
    }();
    std::vector<u64> rb{0};  // reference boundaries
    double acc = 0.0;
    for (size_t c = 0; c + 1 < shares.size(); ++c) {
        acc += shares[c];
        const u64 want = (u64)((double)H * acc);
        const u64 r = (u64)(std::lower_bound(h_offsets, h_offsets + N + 1, want) - h_offsets);
        if (r > rb.back() && r < N) rb.push_back(r);
    }
    rb.push_back(N);
    const size_t C = rb.size() - 1;
    u64 max_last = 0;
    if (pk) max_last = pk->max_hash;  // (the packer's; a wrong one shows below: the largest hash the check finds must not exceed it)
    else
        for (u64 j = 0; j < N; ++j)
            if (h_offsets[j + 1] > h_offsets[j]) max_last = std::max(max_last, h_values[h_offsets[j + 1] - 1]);
    double t_prev = trace_now();
    TRACE("chunk plan + max");
    // (the upload stream and the chunk events are made once per device and kept: a dozen creates and destroys were
    // 0.4 ms of every call)
    struct UpCtx { hipStream_t up = nullptr; std::vector<hipEvent_t> ev, eb, ee; };
    static std::mutex up_mu;
    static UpCtx up_ctx[64];
    std::lock_guard<std::mutex> up_lock(up_mu);  // (one chunked upload per process at a time: they would share the bus anyway)
    UpCtx& uc = up_ctx[db->device & 63];
    hipStream_t& up = uc.up;
    std::vector<hipEvent_t>&ev = uc.ev, &eb = uc.eb, &ee = uc.ee;
    u64* K[2] = {nullptr, nullptr};
    u32* V[2] = {nullptr, nullptr};
    u32* d_ids = nullptr;
    void* d_tmp = nullptr;
    char* d_pk_tab = nullptr;   // a packed database: its block table, payload and first_block[] in HBM
    u64* d_pk_payload = nullptr;
    u64* d_pk_fb = nullptr;
    yh_psort* ps = nullptr;
    int rc = YH_OK;
#define UP_HIP(call)                                                                          \
    if (rc == YH_OK) {                                                                        \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            yh_set_error("%s failed: %s", #call, hipGetErrorString(e__));                     \
            rc = (e__ == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;                      \
        }                                                                                     \
    }
    if (!up) UP_HIP(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    while (ev.size() < C && rc == YH_OK) {
        hipEvent_t a = nullptr, b = nullptr, c2 = nullptr;
        UP_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        UP_HIP(hipEventCreate(&b));
        UP_HIP(hipEventCreate(&c2));
        if (rc == YH_OK) { ev.push_back(a); eb.push_back(b); ee.push_back(c2); }
    }
    const bool dist_sort = yh_psort_applicable(H, max_last);
    bool fused = dist_sort && fz_wanted(db) && max_last != ~0ull;  // (a YH_DB_PAIRWISE_ONLY handle: positions as values, no sorted pairs at all; hash + 1 is the table's key)
    if (!fused) {
        UP_HIP(yh_tmalloc(db, (void**)&K[0], H * sizeof(u64)));
        UP_HIP(yh_tmalloc(db, (void**)&V[0], H * sizeof(u32)));
        UP_HIP(yh_tmalloc(db, (void**)&d_ids, H * sizeof(u32)));
    }
    yh_pieces* pc = nullptr;
    const bool pieces = fused && yh_pc_applicable(H, max_last, N);  // (no first level at all: the regions are read in place)
    if (rc == YH_OK && pieces) rc = fzp_begin(db, d_offsets, max_last, &pc);
    else if (rc == YH_OK && fused) rc = fz_begin(db, d_offsets, max_last, &ps);
    else if (rc == YH_OK && dist_sort) rc = yh_psort_begin(db, H, max_last, &ps);
    if (pk) {
        UP_HIP(yh_tmalloc(db, (void**)&d_pk_tab, std::max<u64>(pk->n_blocks * yh_csr_block_bytes(), 8)));
        UP_HIP(yh_tmalloc(db, (void**)&d_pk_payload, pk->payload_words * sizeof(u64)));
        UP_HIP(yh_tmalloc(db, (void**)&d_pk_fb, (N + 1) * sizeof(u64)));
        UP_HIP(hipStreamSynchronize(st));  // (stream-ordered allocations: there before the upload stream copies into them)
    }
    TRACE("stream, events, buffers");
    if (rc == YH_OK) rc = validate_begin(db);
    const int cur = 0;  // (the buffer the sorted pairs land in)
    // The copies are issued back to back by a thread of their own (a copy from pageable memory returns when the data has
    // left the host, and queueing a chunk's sort and merge takes the host ~0.1 ms: with one thread doing both, the bus
    // idled that long behind every chunk); this thread queues chunk c's device work as soon as its event is recorded.
    std::atomic<int> copied{0};        // chunks whose copy has been issued and event recorded
    std::atomic<int> copy_failed{0};
    const int device = db->device;
    std::thread copier;
    if (rc == YH_OK) {
        copier = std::thread([&, device]() {
            if (hipSetDevice(device) != hipSuccess) { copy_failed.store(1); copied.store((int)C); return; }
            const double t_up = trace_now();
            struct UpClock { yh_db* db; double t0; ~UpClock() { db->ms_h2d_create += (float)(trace_now() - t0); } } up_clock{db, t_up};  // (yh_timing.ms_h2d)
            if (pk && hipMemcpyAsync(d_pk_fb, pk->first_block.data(), (N + 1) * sizeof(u64), hipMemcpyHostToDevice, up) != hipSuccess) copy_failed.store(1);
            for (size_t c = 0; c < C; ++c) {
                const u64 e0 = h_offsets[rb[c]], e1 = h_offsets[rb[c + 1]];
                if (pk) {  // the chunk's blocks: their table entries and their payload words (the last chunk: the spare word too)
                    const u64 b0 = pk->first_block[rb[c]], b1 = pk->first_block[rb[c + 1]];
                    const u64 w0 = yh_csr_block_word_off(pk, b0), w1 = c + 1 == C ? pk->payload_words : yh_csr_block_word_off(pk, b1);
                    const u64 bb = yh_csr_block_bytes();
                    if (b1 > b0 && hipMemcpyAsync(d_pk_tab + b0 * bb, (const char*)pk->tab + b0 * bb, (b1 - b0) * bb, hipMemcpyHostToDevice, up) != hipSuccess)
                        copy_failed.store(1);
                    if (w1 > w0 && w1 <= pk->payload_words &&
                        hipMemcpyAsync(d_pk_payload + w0, pk->payload + w0, (w1 - w0) * sizeof(u64), hipMemcpyHostToDevice, up) != hipSuccess)
                        copy_failed.store(1);
                } else if (e1 > e0 && hipMemcpyAsync(d_values + e0, h_values + e0, (e1 - e0) * sizeof(u64), hipMemcpyHostToDevice, up) != hipSuccess)
                    copy_failed.store(1);
                if (hipEventRecord(ev[c], up) != hipSuccess) copy_failed.store(1);
                copied.store((int)c + 1, std::memory_order_release);
            }
            if (C) (void)hipEventSynchronize(ev[C - 1]);  // (the clock stops when the last byte is up)
        });
    }
    for (size_t c = 0; c < C && rc == YH_OK; ++c) {
        const u64 r0 = rb[c], r1 = rb[c + 1];
        const u64 e0 = h_offsets[r0], e1 = h_offsets[r1], n = e1 - e0;
        while (copied.load(std::memory_order_acquire) <= (int)c) std::this_thread::yield();
        TRACE("chunk copy issued");
        if (copy_failed.load()) { yh_set_error("CSR upload failed"); rc = YH_ERR_HIP; break; }
        UP_HIP(hipStreamWaitEvent(st, ev[c], 0));
        UP_HIP(hipEventRecord(eb[c], st));
        if (rc == YH_OK && pk)
            rc = yh_csr_expand_device(db, d_pk_tab, d_pk_payload, pk->payload_words, d_pk_fb, d_offsets, N, pk->first_block[r0], pk->first_block[r1],
                                      d_values, db->d_flag);
        if (rc == YH_OK) rc = validate_refs(db, d_values, d_offsets, r0, r1);
        if (rc == YH_OK && n && pc) {
            rc = yh_pc_scan(db, pc, d_values, db->d_fz_off, r0, r1, n, false);  // (k_scan_refs above checked the order)
        } else if (rc == YH_OK && n && fused) {
            rc = yh_psort_add(db, ps, d_values + e0, nullptr, n, e0);  // (the value of a pair is its position)
        } else if (rc == YH_OK && n) {
            k_fill_ref_ids<<<grid_for((r1 - r0) * WAVE, 256), 256, 0, st>>>(d_offsets + r0, r1 - r0, d_ids, (u32)r0);
            if (ps) rc = yh_psort_add(db, ps, d_values + e0, d_ids + e0, n);  // this chunk's pairs into the first-level regions
        }
        UP_HIP(hipEventRecord(ee[c], st));
        TRACE("chunk work queued");
    }
    if (copier.joinable()) copier.join();
    if (rc == YH_OK && copy_failed.load()) { yh_set_error("CSR upload failed"); rc = YH_ERR_HIP; }
    if (rc != YH_OK) (void)hipStreamSynchronize(up);
    if (rc == YH_OK) rc = validate_end(db);  // (waits for the stream)
    db->order_checked = true;  // (k_scan_refs checked every chunk)
    TRACE("stream drained");
    float ms_chunks = 0.f;  // device time of the chunks' checks and distribution passes (they ran under the upload)
    for (size_t c = 0; c < C && rc == YH_OK; ++c) {
        float ms = 0.f;
        if (c < eb.size() && hipEventElapsedTime(&ms, eb[c], ee[c]) == hipSuccess) ms_chunks += ms;
        else (void)hipGetLastError();
    }
    bool sorted = false;
    const bool timed_tail = rc == YH_OK && !eb.empty();
    if (timed_tail) UP_HIP(hipEventRecord(eb[0], st));  // (pair 0 again: the device time of the tail behind the last byte)
    if (rc == YH_OK && fused) {  // second level + the fused last pass: the handle is complete behind it (yh_build_index sees db->fz)
        bool took = false;
        if (pc) { rc = fzp_finish(db, pc, d_values, &took, nullptr); pc = nullptr; }
        else { rc = fz_finish(db, ps, &took, nullptr); ps = nullptr; }
        if (rc == YH_OK && !took) {  // not this sort's keys after all: the plain way, everything behind the upload
            fused = false;
            UP_HIP(yh_tmalloc(db, (void**)&K[0], H * sizeof(u64)));
            UP_HIP(yh_tmalloc(db, (void**)&V[0], H * sizeof(u32)));
            UP_HIP(yh_tmalloc(db, (void**)&d_ids, H * sizeof(u32)));
            if (rc == YH_OK) k_fill_ref_ids<<<grid_for(N * WAVE, 256), 256, 0, st>>>(d_offsets, N, d_ids);
        }
    }
    if (rc == YH_OK && ps) rc = yh_psort_finish(db, ps, K[0], V[0], &sorted);  // second level + bucket sorts
    if (rc == YH_OK && !sorted && !fused) {  // not this sort's keys: one rocPRIM sort of everything, behind the upload
        unsigned end_bit2 = 1;
        while (end_bit2 < 64 && (db->max_hash >> end_bit2) != 0) ++end_bit2;
        size_t tb = 0;
        UP_HIP(rocprim::radix_sort_pairs(nullptr, tb, (const u64*)d_values, K[0], (const u32*)d_ids, V[0], (size_t)H, 0u, end_bit2, st));
        UP_HIP(yh_tmalloc(db, &d_tmp, std::max<size_t>(tb, 16)));
        UP_HIP(rocprim::radix_sort_pairs(d_tmp, tb, (const u64*)d_values, K[0], (const u32*)d_ids, V[0], (size_t)H, 0u, end_bit2, st));
        UP_HIP(hipStreamSynchronize(st));
    }
    float ms_tail = 0.f;
    if (timed_tail) {
        UP_HIP(hipEventRecord(ee[0], st));
        UP_HIP(hipEventSynchronize(ee[0]));
        if (rc == YH_OK) (void)hipEventElapsedTime(&ms_tail, eb[0], ee[0]);
    }
    if (rc == YH_OK && sorted && ps) { db->tmp_psort = ps; ps = nullptr; }  // (its buckets are the chunks yh_build_index walks)
    yh_psort_destroy(db, ps);
    ps = nullptr;
    TRACE("sorted");
    if (rc == YH_OK && db->max_hash > max_last) { yh_set_error("internal: largest hash above the largest last element"); rc = YH_ERR_HIP; }
    if (rc == YH_OK && !fused) {
        static const bool check = [] { const char* e = yh_tune_env("YH_CHECK_SORT"); return e && e[0] == '1'; }();
        if (check) {
            u32 bad = 0;
            UP_HIP(hipMemsetAsync(db->d_flag, 0, 4, st));
            k_check_sorted_pairs<<<8192, 256, 0, st>>>(K[cur], V[cur], H, db->d_flag);
            UP_HIP(hipMemcpyAsync(&bad, db->d_flag, 4, hipMemcpyDeviceToHost, st));
            UP_HIP(hipStreamSynchronize(st));
            if (rc == YH_OK && bad) { yh_set_error("YH_CHECK_SORT: the merged pairs are not in (hash, reference) order"); rc = YH_ERR_HIP; }
        }
    }
    db->ms_upload_kernels = ms_tail + ms_chunks;
#undef UP_HIP
    if (up) (void)hipStreamSynchronize(up);
    yh_psort_destroy(db, ps);
    if (pc) { yh_pc_destroy(db, pc); pc = nullptr; if (!db->fz) fz_drop(db); }
    yh_tfree(db, d_tmp);
    yh_tfree(db, d_ids);
    yh_tfree(db, d_pk_tab); yh_tfree(db, d_pk_payload); yh_tfree(db, d_pk_fb);
    if (rc != YH_OK) { yh_tfree(db, K[cur]); yh_tfree(db, V[cur]); return rc; }
    TRACE("frees");
    *d_sk_out = K[cur];
    *d_sv_out = V[cur];
    return YH_OK;
}

// The hash-sorted delta stream from the sorted (hash, reference) pairs (layout: yh_common.h).
static int build_stream(yh_db* db, const u64* d_sk, const u32* d_sv, const u32* d_elem_g, u64 H) {
    hipStream_t st = db->stream;
    if (H == 0) return YH_OK;
    // mean truncated gap in [32, 64): ~1 % fillers, ~|S|/48 key-only candidates per query.  At most 32 bits are
    // dropped: a candidate is confirmed by the LOW 32 BITS of the hash (the 8-byte record of its position), which
    // together with the key are the whole hash only then.  (A small database of wide hashes gets a stream that is
    // mostly fillers: at most 2^32 / 255 = 16.8 M elements.)
    u32 s = 0;
    while (s < 32 && ((db->max_hash >> (s + 1)) / H) >= 32) ++s;
    db->sshift = s;
    u64* d_pos = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    int rc = YH_OK;
#define ST_HIP(call)                                                                          \
    if (rc == YH_OK) {                                                                        \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            yh_set_error("%s failed: %s", #call, hipGetErrorString(e__));                     \
            rc = (e__ == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;                      \
        }                                                                                     \
    }
    auto counts = rocprim::make_transform_iterator(rocprim::counting_iterator<u64>(0), StreamCount{d_sk, s});
    ST_HIP(yh_tmalloc(db, (void**)&d_pos, H * sizeof(u64)));
    ST_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, counts, d_pos, (size_t)H, rocprim::plus<u64>(), st));
    ST_HIP(yh_tmalloc(db, &d_tmp, tmp_bytes + 256));
    ST_HIP(rocprim::inclusive_scan(d_tmp, tmp_bytes, counts, d_pos, (size_t)H, rocprim::plus<u64>(), st));
    u64 L = 0;
    ST_HIP(hipMemcpyAsync(&L, d_pos + (H - 1), sizeof(u64), hipMemcpyDeviceToHost, st));
    ST_HIP(hipStreamSynchronize(st));
    if (rc == YH_OK) {
        db->slen = (L + (STREAM_BLOCK - 1)) & ~(u64)(STREAM_BLOCK - 1);
        const u64 nblk = db->slen / STREAM_BLOCK;
        rc = yh_dmalloc(db, (void**)&db->d_sdelta, db->slen + 64);
        if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_shdr, (nblk + 2) * sizeof(u64));
        if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_srec, db->slen * sizeof(uint2));
        ST_HIP(hipMemsetAsync(db->d_sdelta, 0, db->slen + 64, st));
        ST_HIP(hipMemsetAsync(db->d_shdr, 0xff, (nblk + 2) * sizeof(u64), st));
        ST_HIP(hipMemsetAsync(db->d_srec, 0xff, db->slen * sizeof(uint2), st));
        if (rc == YH_OK)
            k_stream_scatter<<<grid_for(H, 256), 256, 0, st>>>(d_sk, d_sv, d_elem_g, H, s, d_pos, db->d_sdelta, db->d_srec,
                                                              db->d_shdr);
        ST_HIP(hipGetLastError());
        ST_HIP(hipStreamSynchronize(st));
    }
#undef ST_HIP
    yh_tfree(db, d_pos);
    yh_tfree(db, d_tmp);
    return rc;
}

// =================================================================================================
int yh_build_index(yh_db* db, const u64* d_values, const u64* d_offsets, u64* d_sk_pre, u32* d_sv_pre) {
    const u64 N = db->n_refs;
    const u64 H = db->n_hashes;
    hipStream_t st = db->stream;
    double t_prev = trace_now();

    YH_TRY(yh_dmalloc(db, (void**)&db->d_nshared, std::max<u64>(N, 1) * sizeof(u32)));
    YH_HIP(hipMemsetAsync(db->d_nshared, 0, std::max<u64>(N, 1) * sizeof(u32), st));

    if (db->fz) {  // (yh_build_upload_sorted went the fused way: the records are there, nothing else is wanted)
        db->has_index = true;
        return YH_OK;
    }
    db->n_distinct = 0;
    db->n_shared = 0;
    db->n_postings = 0;
    if (H == 0) {
        YH_TRY(yh_dmalloc(db, (void**)&db->d_po, sizeof(u64)));
        YH_HIP(hipMemsetAsync(db->d_po, 0, sizeof(u64), st));
        db->has_index = !(db->flags & YH_DB_NO_INDEX);
        // (an empty database answers the sample-driven queries too: nothing is ever found)
        db->has_dir = db->has_index && !(db->flags & (YH_DB_NO_DIRECTORY | YH_DB_PAIRWISE_ONLY));
        return YH_OK;
    }
    if (H > 0xfffffff0ull * 2) {
        yh_set_error("index build supports at most 2^33 hashes per device");
        return YH_ERR_UNSUPPORTED;
    }

    bool try_psort = yh_psort_applicable(H, db->max_hash);
    if (!d_sk_pre && try_psort && fz_wanted(db) && db->max_hash != ~0ull && yh_pc_applicable(H, db->max_hash, N)) {
        // the fused path from device memory WITHOUT a first level: bounds pass (ordering check, records cleared), the
        // distribution of the pieces, the grouping pass
        yh_pieces* pc = nullptr;
        bool took = false, unsorted = false;
        YH_TRY(fzp_begin(db, d_offsets, db->max_hash, &pc));
        int prc = yh_pc_scan(db, pc, d_values, db->d_fz_off, 0, N, H, !db->order_checked);
        if (prc != YH_OK) { yh_pc_destroy(db, pc); fz_drop(db); return prc; }
        YH_TRY(fzp_finish(db, pc, d_values, &took, &unsorted));
        if (!db->order_checked && unsorted) {
            yh_set_error("a reference sketch is not strictly ascending (or offsets are not monotone)");
            return YH_ERR_UNSORTED;
        }
        if (took) {
            db->order_checked = true;
            db->has_index = true;
            TRACE("index: fused records (pieces)");
            return YH_OK;
        }
    }
    if (!d_sk_pre && try_psort && fz_wanted(db) && db->max_hash != ~0ull) {  // the fused path from device memory (see fz_wanted above)
        yh_psort* fps = nullptr;
        bool took = false, unsorted = false;
        YH_TRY(fz_begin(db, d_offsets, db->max_hash, &fps));
        yh_psort_check_order(fps, !db->order_checked);  // (the first level reads every pair anyway)
        int frc = yh_psort_add(db, fps, d_values, nullptr, H, 0);
        if (frc != YH_OK) { yh_psort_destroy(db, fps); fz_drop(db); return frc; }
        YH_TRY(fz_finish(db, fps, &took, &unsorted));
        if (!db->order_checked && unsorted) {
            yh_set_error("a reference sketch is not strictly ascending (or offsets are not monotone)");
            return YH_ERR_UNSORTED;
        }
        if (took) {
            db->order_checked = true;
            db->has_index = true;
            TRACE("index: fused records");
            return YH_OK;
        }
        try_psort = false;  // (a capacity was exceeded: the plain distribution sort would refuse the same keys)
    }

    u32 *d_ids = nullptr, *d_sv = nullptr, *d_counts = nullptr;
    u64 *d_sk = nullptr, *d_bases = nullptr;
    void* d_tmp = nullptr;
    int rc = YH_OK;
    // The sorted pairs are walked in CHUNKS (count -> scan -> emit).  Sorted by the distribution sort, the chunks are its
    // buckets -- it counted them itself, in LDS -- else fixed chunks of IDX_BLOCK pairs counted by k_idx_count.
    yh_psort* ps = db->tmp_psort;  // (yh_build_upload_sorted's, when it sorted that way; ours to destroy)
    db->tmp_psort = nullptr;
    u64 nb = (H + IDX_BLOCK - 1) / IDX_BLOCK;
    const u64* d_chunk_off = nullptr;
    const u32* d_chunk_counts = nullptr;
#define IDX_HIP(call)                                                                         \
    if (rc == YH_OK) {                                                                        \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            yh_set_error("%s failed: %s", #call, hipGetErrorString(e__));                     \
            rc = (e__ == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;                      \
        }                                                                                     \
    }
    if (d_sk_pre) {  // (yh_build_upload_sorted made them; ours to free)
        d_sk = d_sk_pre;
        d_sv = d_sv_pre;
    } else {
        IDX_HIP(yh_tmalloc(db, (void**)&d_ids, H * sizeof(u32)));
        IDX_HIP(yh_tmalloc(db, (void**)&d_sv, H * sizeof(u32)));
        IDX_HIP(yh_tmalloc(db, (void**)&d_sk, H * sizeof(u64)));
    }
    if (ps) yh_psort_chunks(ps, &nb, &d_chunk_off, &d_chunk_counts);
    db->sort_path = ps ? YH_SORT_TWO_LEVEL : YH_SORT_RADIX;  // (pairs sorted behind the upload: by whichever of the two took them)
    bool order_checked = db->order_checked;
    if (rc == YH_OK && !d_sk_pre) {
        const u32* ids_src = d_ids;
        // the pairs in (hash, reference) order: the distribution of yh_sort.hip for uniform keys (FracMinHash hashes are) --
        // without a first level where the geometry allows (round 5: regions read in place as pieces of the ascending sketches;
        // buckets that overflow go through a side list) --, rocPRIM's LSD radix sort -- stable: equal hashes keep ascending
        // references -- for anything else
        bool sorted = false;
        static const bool no_pc_sort = [] { const char* e = yh_tune_env("YH_NO_PIECES_SORT"); return e && e[0] == '1'; }();
        if (rc == YH_OK && try_psort && !no_pc_sort && db->max_hash != ~0ull && yh_pc_applicable(H, db->max_hash, N)) {
            bool unsorted = false;
            u64 sp_pairs = 0, sp_buckets = 0;
            rc = yh_pc_sort(db, d_values, d_offsets, N, H, db->max_hash, !order_checked, d_sk, d_sv, &ps, &sorted, &unsorted, &sp_pairs, &sp_buckets);
            if (rc == YH_OK && !order_checked && unsorted) {
                yh_set_error("a reference sketch is not strictly ascending (or offsets are not monotone)");
                rc = YH_ERR_UNSORTED;
            }
            if (rc == YH_OK && sorted) {
                order_checked = true;
                db->sort_path = YH_SORT_PIECES;
                db->n_spilled_pairs = sp_pairs;
                db->n_spilled_buckets = sp_buckets;
                yh_psort_chunks(ps, &nb, &d_chunk_off, &d_chunk_counts);
            }
        }
        if (rc == YH_OK && !sorted) k_fill_ref_ids<<<grid_for(N * WAVE, 256), 256, 0, st>>>(d_offsets, N, d_ids);
        if (rc == YH_OK && try_psort && !sorted) {
            bool unsorted = false;
            rc = yh_psort_begin(db, H, db->max_hash, &ps);
            if (rc == YH_OK) yh_psort_check_order(ps, !order_checked);  // (the first level reads every pair anyway)
            if (rc == YH_OK) rc = yh_psort_add(db, ps, d_values, ids_src, H);
            if (rc == YH_OK) rc = yh_psort_finish(db, ps, d_sk, d_sv, &sorted, &unsorted);
            if (rc == YH_OK && !order_checked && unsorted) {
                yh_set_error("a reference sketch is not strictly ascending (or offsets are not monotone)");
                rc = YH_ERR_UNSORTED;
            }
            if (rc == YH_OK && sorted) {
                order_checked = true;
                db->sort_path = YH_SORT_TWO_LEVEL;
                yh_psort_chunks(ps, &nb, &d_chunk_off, &d_chunk_counts);
            } else {
                yh_psort_destroy(db, ps);
                ps = nullptr;
            }
        }
        if (rc == YH_OK && !order_checked) {  // (no distribution sort, or refused: the ordering check as a pass of its own)
            rc = yh_build_check_order(db, d_values, d_offsets);
            order_checked = true;
        }
        if (rc == YH_OK && !sorted) {
            unsigned end_bit = 1;
            while (end_bit < 64 && (db->max_hash >> end_bit) != 0) ++end_bit;
            size_t tmp_bytes = 0;
            IDX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const u64*)d_values, d_sk, ids_src, d_sv,
                                              (size_t)H, 0u, end_bit, st));
            IDX_HIP(yh_tmalloc(db, &d_tmp, std::max<size_t>(tmp_bytes, 16)));
            IDX_HIP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, (const u64*)d_values, d_sk, ids_src, d_sv,
                                              (size_t)H, 0u, end_bit, st));
        }
        static const bool check_sorted = [] { const char* e = yh_tune_env("YH_CHECK_SORT"); return e && e[0] == '1'; }();
        if (rc == YH_OK && check_sorted) {  // (tests / fuzzer: the order on the device, whichever sort made it)
            u32 bad = 0;
            IDX_HIP(hipMemsetAsync(db->d_flag, 0, 4, st));
            k_check_sorted_pairs<<<8192, 256, 0, st>>>(d_sk, d_sv, H, db->d_flag);
            IDX_HIP(hipMemcpyAsync(&bad, db->d_flag, 4, hipMemcpyDeviceToHost, st));
            IDX_HIP(hipStreamSynchronize(st));
            if (rc == YH_OK && bad) { yh_set_error("YH_CHECK_SORT: the sorted pairs are not in (hash, reference) order"); rc = YH_ERR_HIP; }
        }
        // (the sort's buffers are the build's largest temporaries: back to the pool before the next ones come)
        IDX_HIP(hipStreamSynchronize(st));
        yh_tfree(db, d_tmp); d_tmp = nullptr;
        yh_tfree(db, d_ids); d_ids = nullptr;
    }
    TRACE("index: sorted pairs ready");
    const bool want_stream = !(db->flags & YH_DB_PAIRWISE_ONLY);
    if (rc == YH_OK && (db->flags & YH_DB_NO_INDEX)) {  // overlap-only handle: the stream and nothing else
        if (want_stream) rc = build_stream(db, d_sk, d_sv, nullptr, H);
        yh_psort_destroy(db, ps);
        yh_tfree(db, d_ids);
        yh_tfree(db, d_sv);
        yh_tfree(db, d_sk);
        yh_tfree(db, d_counts);
        yh_tfree(db, d_bases);
        yh_tfree(db, d_tmp);
        return rc;
    }
    u32* d_elem_g = nullptr;
    if (want_stream) IDX_HIP(yh_tmalloc(db, (void**)&d_elem_g, H * sizeof(u32)));
    u64 totals[3] = {0, 0, 0};
    // (d_counts: the chunk counts when k_idx_count makes them; later a few words of scratch)
    IDX_HIP(yh_tmalloc(db, (void**)&d_counts, std::max<u64>(d_chunk_counts ? 0 : nb * 3, 16) * sizeof(u32)));
    IDX_HIP(yh_tmalloc(db, (void**)&d_bases, (nb + 1) * 3 * sizeof(u64)));
    if (rc == YH_OK) {
        if (!d_chunk_counts) k_idx_count<<<(u32)((nb * 64 + IDX_THREADS - 1) / IDX_THREADS), IDX_THREADS, 0, st>>>(d_sk, H, d_counts);
        k_idx_scan_counts<<<1, 1024, 0, st>>>(d_chunk_counts ? d_chunk_counts : d_counts, nb, d_bases);
        IDX_HIP(hipGetLastError());
        IDX_HIP(hipMemcpyAsync(totals, d_bases + nb * 3, 3 * sizeof(u64), hipMemcpyDeviceToHost, st));
        IDX_HIP(hipStreamSynchronize(st));
    }
    TRACE("index: counted");
    if (rc == YH_OK) {
        db->n_distinct = totals[0];
        db->n_shared = totals[1];
        db->n_postings = totals[2];
        if (db->n_shared > 0xfffffff0ull) { yh_set_error("more than 2^32 shared hashes"); rc = YH_ERR_UNSUPPORTED; }
    }
    if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_g, std::max<u64>(db->n_shared, 2) * sizeof(u64));
    if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_po, (db->n_shared + 1) * sizeof(u64));
    if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_pr, std::max<u64>(db->n_postings, 1) * sizeof(u32));
    if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_pg, std::max<u64>(db->n_postings, 1) * sizeof(u32));
    if (rc == YH_OK && (db->flags & YH_DB_PAIRWISE_ONLY))
        rc = yh_dmalloc(db, (void**)&db->d_prank, std::max<u64>(db->n_postings, 1) * sizeof(u32));
    if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_hit, db->n_shared + 16);  // zeroed in 16-byte units
    if (rc == YH_OK && db->n_postings > 0xfffffff0ull) { yh_set_error("more than 2^32 postings"); rc = YH_ERR_UNSUPPORTED; }
    TRACE("index: arrays allocated");
    if (rc == YH_OK) {
        IDX_HIP(hipMemsetAsync(db->d_g, 0, std::max<u64>(db->n_shared, 2) * sizeof(u64), st));
        // The bucket table over the distinct hashes (sample-driven lookups) is part of every handle that
        // answers sample queries, unless YH_DB_NO_DIRECTORY / YH_NO_DIRECTORY=1 opts out.
        static const bool dir_env_off = [] { const char* e = yh_tune_env("YH_NO_DIRECTORY"); return e && e[0] == '1'; }();
        const bool full = !(db->flags & (YH_DB_NO_DIRECTORY | YH_DB_PAIRWISE_ONLY)) && !dir_env_off && db->n_distinct > 0;
        unsigned bits = 1;
        while (bits < 64 && (db->max_hash >> bits) != 0) ++bits;
        // compact form: ~2.5 distinct hashes per bucket, and a bucket must span less than 2^32 hash values
        const u64 nb_c = std::max<u64>((db->n_distinct * 2 + 4) / 5, 1);
        // bucket(h) = floor(h * mul / 2^bits) with mul = floor(nb * 2^bits / (max_hash + 1)): monotone, and the
        // hashes [0, max_hash] cover ALL nb buckets (with mul = nb a database whose largest hash is just above a
        // power of two would use half the table at twice the load)
        auto mul_for = [&](u64 nbk) -> u64 {
            const unsigned __int128 num = (unsigned __int128)nbk << bits;
            const unsigned __int128 m = num / ((unsigned __int128)db->max_hash + 1);
            return (u64)std::min<unsigned __int128>(m, ~(u64)0);
        };
        const u64 mul_c = mul_for(nb_c);
        static const bool wide_env = [] { const char* e = yh_tune_env("YH_WIDE_BUCKETS"); return e && e[0] == '1'; }();
        // a bucket spans ceil(2^bits / mul) hash values: the low 32 bits identify a hash inside it iff that is <= 2^32
        const bool compact = full && !wide_env && nb_c <= 0xfffffff0ull && mul_c > 0 &&
                             (bits <= 32 || (((unsigned __int128)1 << bits) + mul_c - 1) / mul_c <= ((unsigned __int128)1 << 32));
        u64* d_dh_tmp = nullptr;   // compact: dh / dref are build-time temporaries
        u32* d_dref_tmp = nullptr;
        if (full && !compact && db->n_distinct > 0x7ffffff0ull) {  // (the five-entry form indexes the distinct hashes with 32 bits)
            yh_set_error("more than 2^31 distinct hashes in a database too wide for compact buckets");
            rc = YH_ERR_UNSUPPORTED;
        }
        if (full) {
            db->bkt_lsh = 64 - bits;
            if (compact) {
                IDX_HIP(yh_tmalloc(db, (void**)&d_dh_tmp, std::max<u64>(db->n_distinct, 2) * sizeof(u64)));
                IDX_HIP(yh_tmalloc(db, (void**)&d_dref_tmp, std::max<u64>(db->n_distinct, 2) * sizeof(u32)));
            } else {
                // ~4 distinct hashes per directory bucket
                u32 lg = 4;
                while (lg < 30 && (4ull << lg) < db->n_distinct) ++lg;
                db->dir_shift = bits > lg ? bits - lg : 0;
                db->dir_nb = (u32)((db->max_hash >> db->dir_shift) + 1);
                if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_dh, std::max<u64>(db->n_distinct, 2) * sizeof(u64));
                if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_dref, std::max<u64>(db->n_distinct, 2) * sizeof(u32));
                if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_dir, ((u64)db->dir_nb + 2) * sizeof(u32));
                const char* nob = yh_tune_env("YH_NO_BUCKETS");
                if (rc == YH_OK && !(nob && nob[0] == '1')) {
                    db->bkt_nb = (db->n_distinct + 1) / 2;
                    db->bkt_mul = std::max<u64>(mul_for(db->bkt_nb), 1);
                    rc = yh_dmalloc(db, (void**)&db->d_bkt, db->bkt_nb * 64);
                    if (rc == YH_OK) IDX_HIP(hipMemsetAsync(db->d_bkt, 0, db->bkt_nb * 64, st));
                }
            }
        }
        u64* const dh_out = !full ? nullptr : compact ? d_dh_tmp : db->d_dh;
        u32* const dref_out = !full ? nullptr : compact ? d_dref_tmp : db->d_dref;
        if (rc == YH_OK)
            k_idx_emit<<<(u32)((nb * 64 + IDX_THREADS - 1) / IDX_THREADS), IDX_THREADS, 0, st>>>(d_sk, d_sv, H, d_bases, db->d_g, db->d_po, db->d_pr, db->d_pg,
                                                        db->d_nshared, dh_out, dref_out, d_elem_g, db->d_prank, d_chunk_off, nb);
        if (rc == YH_OK && full && !compact)
            k_dir_build<<<grid_for(db->n_distinct, 256), 256, 0, st>>>(db->d_dh, db->n_distinct, db->dir_shift, db->dir_nb,
                                                                       db->d_dir);
        if (rc == YH_OK && full && !compact && db->d_bkt)
            k_bkt_build<<<grid_for(db->n_distinct, 256), 256, 0, st>>>(db->d_dh, db->d_dref, db->n_distinct, db->bkt_lsh,
                                                                       db->bkt_mul, reinterpret_cast<u32*>(db->d_bkt));
        if (rc == YH_OK && compact) {
            db->cbkt_nb = nb_c;
            db->bkt_mul = mul_c;
            u64 n_over = 0;
            u64* d_cnt = reinterpret_cast<u64*>(d_counts);  // (free again: the per-block counts were consumed by k_idx_emit)
            IDX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(u64), st));
            if (rc == YH_OK)
                k_cbkt_count_overflow<<<grid_for(db->n_distinct, 256), 256, 0, st>>>(d_dh_tmp, db->n_distinct, db->bkt_lsh, mul_c, d_cnt);
            IDX_HIP(hipMemcpyAsync(&n_over, d_cnt, sizeof(u64), hipMemcpyDeviceToHost, st));
            IDX_HIP(hipStreamSynchronize(st));
            u64 cap = 1024;
            while (cap < 2 * n_over + 16) cap <<= 1;
            if (cap > (1ull << 31)) { yh_set_error("overflow table too large"); rc = YH_ERR_UNSUPPORTED; }
            db->ovf_mask = (u32)(cap - 1);
            if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_cbkt, nb_c * 64);
            if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_ovf_keys, cap * sizeof(u64));
            if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_ovf_vals, cap * sizeof(u32));
            IDX_HIP(hipMemsetAsync(db->d_cbkt, 0, nb_c * 64, st));
            IDX_HIP(hipMemsetAsync(db->d_ovf_keys, 0, cap * sizeof(u64), st));
            IDX_HIP(hipMemsetAsync(db->d_ovf_vals, 0xff, cap * sizeof(u32), st));
            if (rc == YH_OK)
                k_cbkt_build<<<grid_for(db->n_distinct, 256), 256, 0, st>>>(d_dh_tmp, d_dref_tmp, db->n_distinct, db->bkt_lsh, mul_c,
                                                                            reinterpret_cast<u32*>(db->d_cbkt), db->d_ovf_keys,
                                                                            db->d_ovf_vals, db->ovf_mask);
            IDX_HIP(hipGetLastError());
            // Presence filter in front of the buckets (k_index_lookup_tile): one bit per 1 / YH_FILTER_BPH of a hash's
            // share of the range, indexed monotonically like the buckets, so that a sorted sample walks it front to back.
            // A sample hash whose bit is clear is not in the database and its bucket is never read.
            static const u32 fbph = [] { const char* e = yh_tune_env("YH_FILTER_BPH"); return e ? (u32)atoi(e) : 4u; }();
            // (below ~10^6 distinct hashes the whole table is cache resident and the filter only adds a dependent read;
            // YH_FILTER_MIN lowers the bar for tests)
            static const u64 fmin = [] { const char* e = yh_tune_env("YH_FILTER_MIN"); return e ? (u64)atoll(e) : (u64)(1u << 20); }();
            if (rc == YH_OK && fbph && db->n_distinct >= fmin) {
                const u64 fbits = ((db->n_distinct * fbph + 511) / 512) * 512;
                db->filter_mul = mul_for(fbits);
                db->filter_bits = fbits;
                rc = yh_dmalloc(db, (void**)&db->d_filter, fbits / 8 + 64);
                IDX_HIP(hipMemsetAsync(db->d_filter, 0, fbits / 8 + 64, st));
                if (rc == YH_OK && db->filter_mul)
                    k_filter_build<<<grid_for(db->n_distinct, 256), 256, 0, st>>>(d_dh_tmp, db->n_distinct, db->bkt_lsh, db->filter_mul,
                                                                                  db->d_filter);
                IDX_HIP(hipGetLastError());
            }
            IDX_HIP(hipStreamSynchronize(st));
        }
        IDX_HIP(hipStreamSynchronize(st));
        TRACE("index: emit + table + filter");
        yh_tfree(db, d_dh_tmp);
        yh_tfree(db, d_dref_tmp);
        if (rc == YH_OK && full) db->has_dir = true;
        if (rc == YH_OK && want_stream) rc = build_stream(db, d_sk, d_sv, d_elem_g, H);  // (behind the table: its temporaries are gone)
        TRACE("index: delta stream");
        yh_tfree(db, d_elem_g);
        d_elem_g = nullptr;
        IDX_HIP(hipGetLastError());
        IDX_HIP(hipMemcpyAsync(db->d_po + db->n_shared, &db->n_postings, sizeof(u64), hipMemcpyHostToDevice, st));
        if (rc == YH_OK && db->n_postings && N && !(db->flags & YH_DB_PAIRWISE_ONLY)) {
            u32 *d_cc = nullptr, *d_cpo = nullptr, *d_cur = nullptr;
            void* d_st = nullptr;
            size_t st_bytes = 0;
            rc = yh_dmalloc(db, (void**)&db->d_rpo, (N + 1) * sizeof(u32));
            if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_rg, db->n_postings * sizeof(u32));
            if (rc == YH_OK && want_stream) rc = yh_dmalloc(db, (void**)&db->d_rrec, (db->n_postings + 1) * sizeof(uint4));
            if (rc == YH_OK && want_stream) rc = yh_dmalloc(db, (void**)&db->d_rrecx, (db->n_postings + 1) * sizeof(uint4));
            IDX_HIP(yh_tmalloc(db, (void**)&d_cc, N * sizeof(u32)));
            IDX_HIP(yh_tmalloc(db, (void**)&d_cpo, (N + 1) * sizeof(u32)));
            IDX_HIP(yh_tmalloc(db, (void**)&d_cur, N * sizeof(u32)));
            IDX_HIP(rocprim::inclusive_scan(nullptr, st_bytes, db->d_nshared, db->d_rpo + 1, N, rocprim::plus<u32>(), st));
            IDX_HIP(yh_tmalloc(db, &d_st, st_bytes + 256));
            IDX_HIP(hipMemsetAsync(db->d_rpo, 0, sizeof(u32), st));
            IDX_HIP(hipMemsetAsync(d_cpo, 0, sizeof(u32), st));
            IDX_HIP(hipMemsetAsync(d_cur, 0, N * sizeof(u32), st));
            IDX_HIP(rocprim::inclusive_scan(d_st, st_bytes, db->d_nshared, db->d_rpo + 1, N, rocprim::plus<u32>(), st));
            if (rc == YH_OK) k_chunk_counts<<<(u32)((N + 255) / 256), 256, 0, st>>>(db->d_nshared, N, d_cc);
            IDX_HIP(rocprim::inclusive_scan(d_st, st_bytes, d_cc, d_cpo + 1, N, rocprim::plus<u32>(), st));
            u32 n_chunks = 0;
            IDX_HIP(hipMemcpyAsync(&n_chunks, d_cpo + N, sizeof(u32), hipMemcpyDeviceToHost, st));
            IDX_HIP(hipStreamSynchronize(st));
            db->n_chunks = n_chunks;
            if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_work, ((u64)n_chunks + 64) * sizeof(uint4));
            if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_work_count, 16);
            IDX_HIP(hipMemsetAsync(db->d_work_count, 0, 16, st));
            u32* d_pref = nullptr;  // reference of every reference-major posting position (for the holder sets)
            if (want_stream) IDX_HIP(yh_tmalloc(db, (void**)&d_pref, db->n_postings * sizeof(u32)));
            if (rc == YH_OK)
                k_fill_rg<<<grid_for(db->n_postings, 256), 256, 0, st>>>(db->n_postings, db->d_pr, db->d_pg, db->d_po, db->d_rpo,
                                                                         d_cur, db->d_rg, db->d_rrec, db->d_rrecx, d_pref);
            IDX_HIP(hipGetLastError());
            if (trace_on()) { IDX_HIP(hipStreamSynchronize(st)); TRACE("index: reference-major view"); }
            if (rc == YH_OK && want_stream) {  // distinct holder sets per reference (k_set_*)
                const u64 P = db->n_postings;
                u64 *d_k = nullptr, *d_k2 = nullptr;
                u32 *d_i = nullptr, *d_i2 = nullptr, *d_kr = nullptr, *d_kr2 = nullptr, *d_head = nullptr, *d_run = nullptr;
                void* d_t = nullptr;
                size_t tb1 = 0, tb2 = 0, tb3 = 0;
                IDX_HIP(yh_tmalloc(db, (void**)&d_k, P * sizeof(u64)));
                IDX_HIP(yh_tmalloc(db, (void**)&d_k2, P * sizeof(u64)));
                IDX_HIP(yh_tmalloc(db, (void**)&d_i, P * sizeof(u32)));
                IDX_HIP(yh_tmalloc(db, (void**)&d_i2, P * sizeof(u32)));
                IDX_HIP(yh_tmalloc(db, (void**)&d_kr, P * sizeof(u32)));
                IDX_HIP(yh_tmalloc(db, (void**)&d_kr2, P * sizeof(u32)));
                unsigned ref_bits = 1;
                while (ref_bits < 32 && (N >> ref_bits) != 0) ++ref_bits;
                IDX_HIP(rocprim::radix_sort_pairs(nullptr, tb1, d_k, d_k2, d_i, d_i2, (size_t)P, 0u, 64u, st));
                IDX_HIP(rocprim::radix_sort_pairs(nullptr, tb2, d_kr, d_kr2, d_i2, d_i, (size_t)P, 0u, ref_bits, st));
                IDX_HIP(rocprim::inclusive_scan(nullptr, tb3, d_kr, d_kr2, (size_t)P, rocprim::plus<u32>(), st));
                IDX_HIP(yh_tmalloc(db, &d_t, std::max(std::max(tb1, tb2), tb3) + 256));
                if (rc == YH_OK) k_set_keys<<<grid_for(P, 256), 256, 0, st>>>(P, db->d_rrec, db->d_rrecx, d_k, d_i);
                IDX_HIP(rocprim::radix_sort_pairs(d_t, tb1, d_k, d_k2, d_i, d_i2, (size_t)P, 0u, 64u, st));  // by holder-set hash
                if (rc == YH_OK) k_gather_u32<<<grid_for(P, 256), 256, 0, st>>>(P, d_pref, d_i2, d_kr);
                IDX_HIP(rocprim::radix_sort_pairs(d_t, tb2, d_kr, d_kr2, d_i2, d_i, (size_t)P, 0u, ref_bits, st));  // then, stably, by reference
                d_head = d_kr;   // (the reference keys are spent)
                d_run = d_kr2;
                if (rc == YH_OK) k_set_heads<<<grid_for(P, 256), 256, 0, st>>>(P, d_i, d_pref, db->d_rrec, db->d_rrecx, d_head);
                IDX_HIP(rocprim::inclusive_scan(d_t, tb3, d_head, d_run, (size_t)P, rocprim::plus<u32>(), st));
                u32 n_sets = 0;
                IDX_HIP(hipMemcpyAsync(&n_sets, d_run + (P - 1), sizeof(u32), hipMemcpyDeviceToHost, st));
                IDX_HIP(hipStreamSynchronize(st));
                db->n_sets = n_sets;
                if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_hrec, ((u64)n_sets + 1) * sizeof(uint4));
                if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_hrecx, ((u64)n_sets + 1) * sizeof(uint4));
                if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_hmult, ((u64)n_sets + 1) * sizeof(u32));
                if (rc == YH_OK) rc = yh_dmalloc(db, (void**)&db->d_hpo, (N + 2) * sizeof(u32));
                IDX_HIP(hipMemsetAsync(db->d_hmult, 0, ((u64)n_sets + 1) * sizeof(u32), st));
                IDX_HIP(hipMemsetAsync(db->d_hpo, 0, (N + 2) * sizeof(u32), st));
                if (rc == YH_OK)  // per-reference counts go to hpo[1..N], then an inclusive scan in place makes them offsets
                    k_set_emit<<<grid_for(P, 256), 256, 0, st>>>(P, d_i, d_pref, d_head, d_run, db->d_rrec, db->d_rrecx, db->d_hrec,
                                                                db->d_hrecx, db->d_hmult, db->d_hpo + 1);
                IDX_HIP(rocprim::inclusive_scan(d_st, st_bytes, db->d_hpo + 1, db->d_hpo + 1, N, rocprim::plus<u32>(), st));
                IDX_HIP(hipGetLastError());
                IDX_HIP(hipStreamSynchronize(st));
                yh_tfree(db, d_k); yh_tfree(db, d_k2); yh_tfree(db, d_i); yh_tfree(db, d_i2);
                yh_tfree(db, d_kr); yh_tfree(db, d_kr2); yh_tfree(db, d_t);
                TRACE("index: holder sets");
            }
            IDX_HIP(hipStreamSynchronize(st));
            yh_tfree(db, d_pref);
            yh_tfree(db, d_cc);
            yh_tfree(db, d_cpo);
            yh_tfree(db, d_cur);
            yh_tfree(db, d_st);
        }
        IDX_HIP(hipStreamSynchronize(st));
    }
    TRACE("index: emitted + drained");
#undef IDX_HIP
    yh_psort_destroy(db, ps);
    yh_tfree(db, d_elem_g);
    yh_tfree(db, d_ids);
    yh_tfree(db, d_sv);
    yh_tfree(db, d_sk);
    yh_tfree(db, d_counts);
    yh_tfree(db, d_bases);
    yh_tfree(db, d_tmp);
    TRACE("index: frees");
    if (rc == YH_OK) db->has_index = true;
    return rc;
}
