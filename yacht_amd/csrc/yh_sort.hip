// yh_sort.hip — the sort under T2 (compute_index_from_sketches, src/cpp/main.cpp:215-246), hand-written for gfx950.
//
// The reference builds `hash_index[h] = [ids containing h]` by inserting every hash of every sketch into a node-based
// hash map, one thread (12.1 of 16.6 s in SURVEY.md's probe).  Here: all (hash, reference) pairs of the database sorted by
// (hash, reference), from which the index is cut by run detection (yh_build.hip: k_idx_*).  Rounds 1-3 called
// rocprim::radix_sort_pairs for it: an LSD radix sort that does not know anything about the keys -- 6-7 passes over the
// 12-byte pairs, 62 % of `yacht train`'s device time.  FracMinHash hashes are UNIFORM below max_hash, so a linear function
// of the key is a perfect splitter and the sort becomes distribution + a sort in LDS:
//
//   bucket(h) = floor(h * NB / (max_hash + 1))     monotone; NB = P1 * P2 buckets of ~FILL pairs each
//   k_part<1>   every tile of 4 096 pairs: LDS histogram over the P1 first-level bins (bucket / P2), ONE global atomic per
//               (tile, bin) reserves the tile's run in the bin's region, the tile is put in bin order in LDS and leaves as
//               runs of consecutive addresses (coalesced stores; without the staging: one isolated 12-byte store per pair)
//   k_part<2>   the same over every first-level region, bin = bucket % P2
//   k_bucket_sort   one workgroup per bucket (<= CAP pairs, in LDS): a counting sort over S fine slots of the bucket's key
//               range (again linear in the key: ~0.6 pairs per slot), then every pair ranks itself among the few pairs
//               of its slot by (hash, reference) -- equal hashes end up in ascending reference order, what a stable sort
//               of the CSR would give -- and the bucket leaves in order, coalesced, at its exact place of the output
//
// Three passes over the pairs (read 12 B + write 12 B each) instead of 6-7, no global merge.  Every capacity is checked on
// the device: keys that are not uniform enough (a region or a bucket overflows, a slot holds more than SLOT_MAX pairs: a
// hash held by a thousand references) raise a flag and the caller sorts with rocPRIM instead -- slower, equally exact.
// The first level can be fed in pieces (yh_psort_add): the chunks of a host database are distributed while the next
// chunk crosses PCIe, and only levels two and three remain behind the last byte (yh_build_upload_sorted).
//
// POSITION MODE (`yacht train`'s handle, yh_db::fz): the value of a pair is its CSR position, not its reference -- the
// order is the same -- and the last pass (k_bucket_group) does not sort or write pairs at all: it groups the bucket's pairs
// by hash in an LDS hash table and stores, at the position of every element whose hash another reference holds too,
// the 8-byte record the pairwise pass reads ("the other holders").  No posting arrays, no rank per posting, no
// transposition: see yh_build.hip (fz_*) and yh_pairwise.hip (k_pair_rows<.., true>).
#include "yh_common.h"
#include "yh_sort.h"

#include <stdlib.h>

#include <algorithm>
#include <cmath>

namespace {

#ifndef YH_PART_TILE
#define YH_PART_TILE 4096
#endif
#ifndef YH_PART_THREADS
#define YH_PART_THREADS 1024
#endif
constexpr u32 PART_TILE = YH_PART_TILE;      // pairs per workgroup of the distribution passes
constexpr u32 PART_THREADS = YH_PART_THREADS;
constexpr u32 PART_ITEMS = PART_TILE / PART_THREADS;
constexpr u32 PART_MAX_BINS = 1024;  // bins per level (LDS histogram)
#ifndef YH_BKT_BITS
#define YH_BKT_BITS 12  // log2 of the pairs a final bucket may hold (tuning builds: yacht_amd.build.build_variant)
#endif
#ifndef YH_BKT_THREADS
#define YH_BKT_THREADS 1024
#endif
constexpr u32 BKT_SLOT_BITS = YH_BKT_BITS;
constexpr u32 BKT_CAP = 1u << BKT_SLOT_BITS;   // pairs a final bucket may hold
#ifndef YH_GROUP_NT
#define YH_GROUP_NT 1  // 1: k_bucket_group5 reads its pairs with non-temporal loads (adopted); 2: k_piece_part stores them so (measured worse)
#endif
#ifndef YH_BKT_FILL8
#define YH_BKT_FILL8 5   // eighths of its capacity a bucket holds on average (tuning: scripts/sweep_fill.sh)
#endif
constexpr u32 BKT_FILL = BKT_CAP / 8 * YH_BKT_FILL8;  // ... and holds on average (2 560 of 4 096; uniform keys: sd ~51; clustered references ~3x that)
constexpr u32 BKT_THREADS = YH_BKT_THREADS;
constexpr u32 BKT_ITEMS = BKT_CAP / BKT_THREADS;
constexpr u32 BKT_SLOTS = BKT_CAP;   // fine slots of the counting sort inside a bucket (= 1 << BKT_SLOT_BITS)
constexpr u32 TOT_LANES = 256;       // rows of the grouping pass's counters (k_bucket_group)
constexpr u32 SLOT_MAX = 1024;       // pairs of one slot a pair ranks itself against (a hash held by that many references: 10^6 LDS reads); more: not this sort's input

// The FINE slot of a hash -- floor(h * NB * S / (max_hash + 1)), S = 2^BKT_SLOT_BITS slots per bucket -- is the one linear
// function everything is cut from: bucket = fine >> BKT_SLOT_BITS, slot inside the bucket = fine & (S - 1).  (A multiplier for
// the BUCKET index alone has too few significant bits at a few thousand buckets -- 38 281 for configs[3] -- and disagrees
// with the fine index at the buckets' edges: 6 % of the pairs landed in a neighbour's slot range.)
__device__ __forceinline__ u32 fine_of(u64 h, u32 lsh, u64 mul_fine) { return (u32)__umul64hi(h << lsh, mul_fine); }
__device__ __forceinline__ u32 bucket_of(u64 h, u32 lsh, u64 mul_fine) { return fine_of(h, lsh, mul_fine) >> BKT_SLOT_BITS; }

// block-wide exclusive scan over `n` (<= 2 * blockDim.x ... any multiple handled by the caller) LDS words, in place;
// returns nothing: arr[i] = sum of arr[0..i).  blockDim.x threads, n <= ITEMS * blockDim.x with ITEMS consecutive words per thread.
template <u32 ITEMS>
__device__ __forceinline__ void block_scan_inplace(u32* arr, u32 n, u32* wave_tot /* >= 17 words */) {
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, nw = blockDim.x >> 6;
    u32 v[ITEMS];
    u32 sum = 0;
#pragma unroll
    for (u32 k = 0; k < ITEMS; ++k) {
        const u32 i = tid * ITEMS + k;
        v[k] = i < n ? arr[i] : 0u;
        sum += v[k];
    }
    u32 inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = (u32)__shfl_up((int)inc, d);
        if (lane >= (u32)d) inc += t;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        const u32 w = lane < nw ? wave_tot[lane] : 0u;
        u32 winc = w;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const u32 t = (u32)__shfl_up((int)winc, d);
            if (lane >= (u32)d) winc += t;
        }
        if (lane < nw) wave_tot[lane] = winc - w;
    }
    __syncthreads();
    u32 run = wave_tot[wv] + inc - sum;
#pragma unroll
    for (u32 k = 0; k < ITEMS; ++k) {
        const u32 i = tid * ITEMS + k;
        if (i < n) arr[i] = run;
        run += v[k];
    }
    __syncthreads();
}

// Which reference owns CSR position p?  tab[j] = the reference of position j << YH_REF_TAB_SH (k_ref_table below); the
// answer lies in [tab[j], tab[j + 1]] -- one entry for all but the blocks a sketch boundary crosses.
__device__ __forceinline__ u32 ref_of(u64 p, const u32* __restrict__ tab, const u64* __restrict__ off) {
    const u64 j = p >> YH_REF_TAB_SH;
    u32 lo = tab[j], hi = tab[j + 1];
    while (lo < hi) {  // the largest r in [lo, hi] with off[r] <= p
        const u32 mid = lo + (hi - lo + 1) / 2;
        if (off[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// bit p of `bits` (cleared before): CSR position p is the first of a (non-empty) sketch
__global__ void k_first_bits(const u64* __restrict__ off, u64 n_refs, unsigned long long* __restrict__ bits) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (r >= n_refs) return;
    const u64 b = off[r];
    if (off[r + 1] > b) atomicOr(&bits[b >> 6], 1ull << (b & 63u));
}
__global__ void k_ref_table(const u64* __restrict__ off, u64 n_refs, u64 n_tab, u32* __restrict__ tab) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= n_tab) return;
    const u64 p = j << YH_REF_TAB_SH;
    u64 lo = 0, hi = n_refs - 1;  // the largest r with off[r] <= p (off[0] = 0: there is one)
    while (lo < hi) {
        const u64 mid = lo + (hi - lo + 1) / 2;
        if (off[mid] <= p) lo = mid; else hi = mid - 1;
    }
    tab[j] = (u32)lo;
}

struct PartArgs {
    const u64* in_k;
    const u32* in_v;      // values; NULL (level 1): the value of a pair is its CSR POSITION, val_base + its index
    u64 val_base;
    const u64* first_bits;  // position mode + ordering check: bit p set = position p is the first of its sketch (k_first_bits)
    u64* rec_clear;       // position mode: yh_db::d_fz_rec -- every pair clears the record of its position on the way through
                          // (400 MB of streaming stores under a pass that waits for LDS and atomics, instead of a memset of their own)
    u64 n_in;             // pairs of this call's input (level 1; level 2 without in_cnt: ONE input segment of n_in pairs)
    u64 cap_in;           // level 2: capacity of an input region
    const u32* in_cnt;    // level 2: pairs in every input region
    u32 tiles_per_seg;    // level 2: workgroups per input region
    u64 mul;              // fine slot of h = umulhi(h << lsh, mul); bucket = fine >> BKT_SLOT_BITS
    u32 lsh, P2, nbins;
    u64* out_k;
    u32* out_v;
    u64 cap_out;
    u32* out_cnt;
    u32* flags;           // [0] |= 1: a region overflowed; |= 8: two neighbours with the same value (reference) not ascending
    u32 check_order;      // level 1: the input is a CSR in reference order with the reference id as value -- check every sketch's order here
};

template <int LEVEL>
__global__ void __launch_bounds__(PART_THREADS) k_part(const PartArgs a) {
    __shared__ u64 skey[PART_TILE];
    __shared__ u32 sval[PART_TILE];
    __shared__ u32 hist[PART_MAX_BINS], loc[PART_MAX_BINS], gbase[PART_MAX_BINS];  // (gbase: where the bin's run starts in its region MINUS where it starts in the tile)
    __shared__ u32 wtot[17];
    const u32 tid = threadIdx.x;
    u64 seg_n, in_base;
    u32 seg = 0, t;
    if (LEVEL == 1) {
        seg_n = a.n_in;
        in_base = 0;
        t = blockIdx.x;
    } else {
        seg = blockIdx.x / a.tiles_per_seg;
        t = blockIdx.x % a.tiles_per_seg;
        seg_n = a.in_cnt ? min((u64)a.in_cnt[seg], a.cap_in) : a.n_in;
        in_base = (u64)seg * a.cap_in;
    }
    const u64 t0 = (u64)t * PART_TILE;
    if (t0 >= seg_n) return;  // (workgroup-uniform)
    const u32 tile_n = (u32)min((u64)PART_TILE, seg_n - t0);
    for (u32 b = tid; b < a.nbins; b += PART_THREADS) hist[b] = 0;
    __syncthreads();
    // (the bin is computed twice -- when the pair is counted and when it leaves -- rather than kept in LDS beside it: the
    // LDS pipe, ~7 operations per pair now, is what these passes keep busy, a multiplication is free)
    auto bin_of = [&](u64 h) -> u32 {
        const u32 b = bucket_of(h, a.lsh, a.mul);
        const u32 x = LEVEL == 1 ? b / a.P2 : b % a.P2;
        return x >= a.nbins ? a.nbins - 1 : x;  // (keys above max_hash: cannot happen after validation; stay in range)
    };
    u64 key[PART_ITEMS];
    u32 val[PART_ITEMS], bin[PART_ITEMS], rank[PART_ITEMS];
    const bool first_pass = LEVEL == 1 || !a.in_cnt;  // (the input is the CSR itself)
    const bool positions = first_pass && !a.in_v;
    const bool check = a.check_order && first_pass;
    // Every load of the tile first, none of them depending on another.  What the ordering check needs beyond the pair is the
    // same for a whole wave -- the pair in front of its first lane (the other lanes get their neighbour's by a shuffle) and,
    // positions, the one or two words of the sketches' first positions its 64 pairs fall into -- and is read with SCALAR
    // loads: a vector load costs the address unit its 16 cycles per wave whether or not the lanes agree (two more per pair
    // were 0.05 ms of this pass).
    const u32 wave0 = (u32)__builtin_amdgcn_readfirstlane((int)tid) & ~63u;  // (uniform: the wave's first lane)
    u64 pkey[PART_ITEMS], fb0[PART_ITEMS], fb1[PART_ITEMS];
    u32 pval[PART_ITEMS];
#pragma unroll
    for (u32 k = 0; k < PART_ITEMS; ++k) {
        const u32 i = k * PART_THREADS + tid;
        const u32 iw = k * PART_THREADS + wave0;  // (uniform)
        pkey[k] = 0; fb0[k] = 0; fb1[k] = 0; pval[k] = 0;
        if (i < tile_n) {
            key[k] = a.in_k[in_base + t0 + i];
            val[k] = positions ? (u32)(a.val_base + t0 + i) : a.in_v[in_base + t0 + i];
        }
        if (check && iw < tile_n) {
            if (t0 + iw > 0) {
                pkey[k] = a.in_k[in_base + t0 + iw - 1];
                if (!positions) pval[k] = a.in_v[in_base + t0 + iw - 1];
            }
            if (positions) {
                const u64 p0 = a.val_base + t0 + iw;
                fb0[k] = a.first_bits[p0 >> 6];
                fb1[k] = a.first_bits[(p0 >> 6) + 1];  // (the map has a word to spare)
            }
        }
    }
    if (check) {
#pragma unroll
        for (u32 k = 0; k < PART_ITEMS; ++k) {  // (every lane takes part: inactive ones hand on garbage nobody uses)
            const u64 up = ((u64)(u32)__shfl_up((int)(u32)(key[k] >> 32), 1) << 32) | (u32)__shfl_up((int)(u32)key[k], 1);
            const u32 upv = (u32)__shfl_up((int)val[k], 1);
            if ((tid & 63u) != 0) { pkey[k] = up; pval[k] = upv; }
        }
    }
#pragma unroll
    for (u32 k = 0; k < PART_ITEMS; ++k) {
        const u32 i = k * PART_THREADS + tid;
        bin[k] = 0xffffffffu;
        if (i < tile_n) {
            if (positions && a.rec_clear) a.rec_clear[val[k]] = 0;
            if (check && t0 + i > 0) {
                // same reference as the element in front => strictly larger hash (positions: the same reference unless this
                // element is the first of its sketch)
                const u64 p0 = a.val_base + t0 + k * PART_THREADS + wave0;  // the position of the wave's first lane
                const u64 word = ((u64)val[k] >> 6) == (p0 >> 6) ? fb0[k] : fb1[k];
                const bool same = positions ? ((word >> (val[k] & 63u)) & 1ull) == 0ull : pval[k] == val[k];
                if (same && !(pkey[k] < key[k])) atomicOr(a.flags, 8u);
            }
            bin[k] = bin_of(key[k]);
            rank[k] = atomicAdd(&hist[bin[k]], 1u);
        }
    }
    __syncthreads();
    // reserve the tile's run in every bin's region, then turn the histogram into offsets inside the tile.  The atomics'
    // answers are not needed before the pairs leave: they stay in registers while the tile is put in bin order (the wait
    // for a contended L2 atomic -- every tile of the launch adds to the same few hundred counters -- was in the chain of
    // every tile)
    constexpr u32 BINS_PER_THREAD = (PART_MAX_BINS + PART_THREADS - 1) / PART_THREADS;
    u32 g_mine[BINS_PER_THREAD], c_mine[BINS_PER_THREAD];
#pragma unroll
    for (u32 q = 0; q < BINS_PER_THREAD; ++q) {
        const u32 b = q * PART_THREADS + tid;
        g_mine[q] = 0;
        c_mine[q] = 0;
        if (b < a.nbins) {
            const u32 c = hist[b];
            c_mine[q] = c;
            if (c) {
                const u32 region = LEVEL == 1 ? b : seg * a.P2 + b;
                g_mine[q] = atomicAdd(&a.out_cnt[region], c);
            }
            loc[b] = c;
        }
    }
    __syncthreads();
    block_scan_inplace<BINS_PER_THREAD>(loc, a.nbins, wtot);
#pragma unroll
    for (u32 k = 0; k < PART_ITEMS; ++k)
        if (bin[k] != 0xffffffffu) {
            const u32 s = loc[bin[k]] + rank[k];
            skey[s] = key[k];
            sval[s] = val[k];
        }
#pragma unroll
    for (u32 q = 0; q < BINS_PER_THREAD; ++q) {
        const u32 b = q * PART_THREADS + tid;
        if (b < a.nbins) {
            gbase[b] = g_mine[q] - loc[b];
            if (c_mine[q] && (u64)g_mine[q] + c_mine[q] > a.cap_out) atomicOr(a.flags, 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < PART_ITEMS; ++k) {
        const u32 s = k * PART_THREADS + tid;
        if (s < tile_n) {
            const u64 h = skey[s];
            const u32 b = bin_of(h);
            const u64 at = (u32)(gbase[b] + s);
            if (at < a.cap_out) {
                const u64 region = LEVEL == 1 ? b : (u64)seg * a.P2 + b;
                a.out_k[region * a.cap_out + at] = h;
                a.out_v[region * a.cap_out + at] = sval[s];
            }
        }
    }
}

// exclusive scan of min(cnt[b], cap) over the buckets into 64-bit offsets (single workgroup); flags |= 2 when a bucket is over
// (cap = 0: the counts as they are -- the pairs beyond a bucket's capacity are on the side list and come back to their places)
__global__ void __launch_bounds__(1024) k_bucket_offsets(const u32* __restrict__ cnt, u64 nb, u32 cap, u64* __restrict__ off,
                                                         u32* __restrict__ flags) {
    __shared__ u32 lds[1024];
    __shared__ u32 wtot[17];
    u64 carry = 0;
    for (u64 base = 0; base < nb; base += 1024) {
        const u64 b = base + threadIdx.x;
        u32 c = b < nb ? cnt[b] : 0u;
        if (cap && c > cap) { atomicOr(flags, 2u); c = cap; }
        lds[threadIdx.x] = c;
        __syncthreads();
        block_scan_inplace<1>(lds, 1024, wtot);
        if (b < nb) off[b] = carry + lds[threadIdx.x];
        __syncthreads();
        // the block's total: last exclusive value + last count
        if (threadIdx.x == 1023) wtot[16] = lds[1023] + c;
        __syncthreads();
        carry += wtot[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) off[nb] = carry;
}

struct BucketArgs {
    const u64* in_k;
    const u32* in_v;
    const u32* cnt;
    const u64* off;
    u64 cap_in;
    u64 mul_fine;  // fine slot of h = umulhi(h << lsh, mul_fine); the low BKT_SLOT_BITS bits = the slot inside the bucket
    u32 lsh;
    u64* out_k;
    u32* out_v;
    u32* flags;    // [0] |= 4: a slot held more than SLOT_MAX pairs
    u32* counts;   // [buckets][3] {distinct hashes, hashes held by >= 2 references, pairs of those}: what k_idx_count counts per
                   // chunk of the sorted pairs -- a run of equal hashes never leaves its bucket, so the bucket sees it whole
    // position mode (k_bucket_group): the bucket does not leave as sorted pairs at all -- every element whose hash another
    // reference holds too gets its record of the pairwise pass, stored at its own CSR position
    u64 nb;                      // buckets
    u64 n_pos;                   // CSR positions (H)
    u32 per_xcd;                 // ... of one XCD (see the kernel)
    u64* rec;                    // [H] yh_db::d_fz_rec
    u32* list;                   // [buckets][BKT_CAP] yh_db::d_fz_list -- the SAME memory as in_v: a bucket's values are in registers before anything is stored
    const u32* ref_tab;
    const u64* ref_off;
    u32 inline_ok;               // reference ids fit the 21-bit fields of an inline record
    unsigned long long* totals;  // [TOT_LANES][8] {distinct hashes, shared hashes, their pairs, pairs seen, ...}: bucket b adds into row b % TOT_LANES
                                 // (one row of four counters for all 19 600 workgroups: 78 000 atomics on ONE cache line, ~6 ns each --
                                 // the whole 0.5 ms of this kernel at configs[3], whatever else it did: profiles/r05/ablate_group.txt)
    u64 stride_v;                // packed pairs: elements between two buckets of in_v / list (in_k: cap_in)
    u32* over_bits;              // != NULL: a bucket with more pairs than its capacity (k_bucket_sort: or with a crowded slot) is not an
                                 // error: its bit is set, flags[2] counts it, and its pairs are grouped / sorted from a side list
    u32 rem_bits;                // != 0: PACKED pairs (k_piece_part): in_k = low rem_bits bits of the hash | reference << rem_bits
                                 // (inside a bucket the hashes span less than 2^rem_bits: the low bits identify them), in_v = position
    u32 write_all;               // k_bucket_group5: EVERY pair stores the record of its position (0: nobody else holds its hash) --
                                 // the records are not cleared in front of the distribution (round 6)
};

// exclusive scan of arr[0 .. ITEMS * blockDim.x) in place, ITEMS consecutive words per thread; two barriers (the second one
// behind the last store).  Every wave sums the totals of the waves below it itself instead of waiting for one wave to scan them.
template <u32 ITEMS>
__device__ __forceinline__ void block_scan_inplace2(u32* arr, u32* wave_tot /* >= 16 words */) {
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, nw = blockDim.x >> 6;
    u32 v[ITEMS];
    u32 sum = 0;
    if constexpr (ITEMS == 4) {  // (one 16-byte LDS read: arr is 16-byte aligned -- a __shared__ array of the kernel)
        const uint4 q = reinterpret_cast<const uint4*>(arr)[tid];
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        sum = q.x + q.y + q.z + q.w;
    } else {
#pragma unroll
        for (u32 k = 0; k < ITEMS; ++k) {
            v[k] = arr[tid * ITEMS + k];
            sum += v[k];
        }
    }
    u32 inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = (u32)__shfl_up((int)inc, d);
        if (lane >= (u32)d) inc += t;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    u32 below = (lane < nw && lane < wv) ? wave_tot[lane] : 0u;  // (nw <= 16)
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) below += (u32)__shfl_xor((int)below, d);
    below = (u32)__shfl((int)below, 0);
    u32 run = below + inc - sum;
    if constexpr (ITEMS == 4) {
        reinterpret_cast<uint4*>(arr)[tid] = make_uint4(run, run + v[0], run + v[0] + v[1], run + v[0] + v[1] + v[2]);
    } else {
#pragma unroll
        for (u32 k = 0; k < ITEMS; ++k) {
            arr[tid * ITEMS + k] = run;
            run += v[k];
        }
    }
    __syncthreads();
}

// The last pass: every bucket (<= BKT_CAP pairs) sorted in LDS, one workgroup per bucket, and written out coalesced at its
// exact place, with the run statistics k_idx_emit needs (the index build of every handle but `yacht train`'s own, whose
// last pass is k_bucket_group below).  ~17 LDS operations per pair: 16-byte clears and scans, a scan in which every wave
// sums the totals below it itself (two barriers), no store that would put back what is in place already.
// (Measured and dropped: PERSISTENT workgroups that fetch the next bucket into registers while they sort the current one --
// 86 registers, one workgroup per CU; held to 64 it spills: 1.02 ms for configs[3] against 0.72 ms.)
__global__ void __launch_bounds__(BKT_THREADS) k_bucket_sort(const BucketArgs a) {
    __shared__ u64 skey[BKT_CAP];
    __shared__ u32 sval[BKT_CAP];
    __shared__ __attribute__((aligned(16))) u32 start[BKT_SLOTS];  // counts, then offsets
    __shared__ u32 wtot[16];
    __shared__ u32 tot3[3];
    static_assert(BKT_THREADS <= 1024, "block_scan_inplace2 sums at most 16 wave totals");
    const u32 tid = threadIdx.x;
    const u64 b = blockIdx.x;
    if (b >= a.nb) return;
    const u32 n = min(a.cnt[b], BKT_CAP);
    if (n == 0) return;  // (workgroup-uniform; its three counts stay zero: the array is cleared before the launch)
    u64 key[BKT_ITEMS];
    u32 val[BKT_ITEMS], slot[BKT_ITEMS], rank[BKT_ITEMS];
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 i = k * BKT_THREADS + tid;
        if (i < n) { key[k] = a.in_k[b * a.cap_in + i]; val[k] = a.in_v[b * a.cap_in + i]; }
    }
    // the whole bucket to the side list: it overflowed, or -- see below -- one of its slots is crowded; the list is sorted on its own
    // and its pairs come back to this bucket's place in the output (k_spill_place)
    auto spill_bucket = [&]() {
        if (tid == 0) { atomicOr(&a.over_bits[b >> 5], 1u << (b & 31u)); atomicAdd(&a.flags[2], 1u); }
    };
    if (a.over_bits && a.cnt[b] > BKT_CAP) { spill_bucket(); return; }  // (uniform)
    if (tid < 3) tot3[tid] = 0;
    static_assert(BKT_SLOTS % (4 * BKT_THREADS) == 0 || BKT_SLOTS / BKT_THREADS < 4, "16-byte clears");
    if (BKT_SLOTS / BKT_THREADS >= 4) {
        for (u32 i = tid; i < BKT_SLOTS / 4; i += BKT_THREADS) reinterpret_cast<uint4*>(start)[i] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (u32 i = tid; i < BKT_SLOTS; i += BKT_THREADS) start[i] = 0;
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 i = k * BKT_THREADS + tid;
        slot[k] = 0xffffffffu;
        if (i < n) {
            slot[k] = fine_of(key[k], a.lsh, a.mul_fine) & (BKT_SLOTS - 1u);
            rank[k] = atomicAdd(&start[slot[k]], 1u);
        }
    }
    __syncthreads();
    block_scan_inplace2<BKT_SLOTS / BKT_THREADS>(start, wtot);
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k)
        if (slot[k] != 0xffffffffu) {
            const u32 at = start[slot[k]] + rank[k];
            skey[at] = key[k];
            sval[at] = val[k];
        }
    __syncthreads();
    // every pair ranks itself among the pairs of its slot; a crowded slot = many pairs with (nearly) the same hash, which
    // this quadratic step is not made for
    bool crowded = false;
    u32 pos[BKT_ITEMS];
    bool moved[BKT_ITEMS];
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        moved[k] = false;
        if (slot[k] != 0xffffffffu) {
            const u32 s0 = start[slot[k]];
            const u32 c = (slot[k] + 1u < BKT_SLOTS ? start[slot[k] + 1u] : n) - s0;  // pairs of the slot
            u32 less = 0;
            if (c > SLOT_MAX) {
                crowded = true;
            } else {
                for (u32 q = s0; q < s0 + c; ++q) {
                    const u64 kq = skey[q];
                    less += (kq < key[k] || (kq == key[k] && sval[q] < val[k])) ? 1u : 0u;
                }
            }
            pos[k] = s0 + less;
            moved[k] = less != rank[k];  // (the one pair of a slot -- most of them -- is where it belongs already)
        }
    }
    if (a.over_bits) {  // (uniform) a crowded slot -- a hash a thousand references hold -- sends the bucket to the side list
        if (tid == 0) tot3[0] = 0;
        __syncthreads();
        if (crowded) tot3[0] = 1u;
        __syncthreads();
        const bool any = tot3[0] != 0u;
        __syncthreads();
        if (tid == 0) tot3[0] = 0;
        if (any) { spill_bucket(); return; }
    } else if (crowded) {
        atomicOr(a.flags, 4u);
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k)
        if (slot[k] != 0xffffffffu && moved[k]) {
            skey[pos[k]] = key[k];
            sval[pos[k]] = val[k];
        }
    __syncthreads();
    u32 c0 = 0, c1 = 0, c2 = 0;
    const u64 out_base = a.off[b];
    for (u32 i0 = 0; i0 < n; i0 += BKT_THREADS) {  // (workgroup-uniform bound: the ballots)
        const u32 i = i0 + tid;
        bool head = false, shared = false;
        if (i < n) {
            const u64 h = skey[i];
            a.out_k[out_base + i] = h;
            a.out_v[out_base + i] = sval[i];
            const bool eq_prev = i > 0 && skey[i - 1] == h, eq_next = i + 1 < n && skey[i + 1] == h;
            head = !eq_prev;
            shared = eq_prev || eq_next;
        }
        c0 += (u32)__popcll(__ballot(head));
        c1 += (u32)__popcll(__ballot(head && shared));
        c2 += (u32)__popcll(__ballot(shared));
    }
    if (a.counts) {
        if ((tid & 63u) == 0) { atomicAdd(&tot3[0], c0); atomicAdd(&tot3[1], c1); atomicAdd(&tot3[2], c2); }
        __syncthreads();
        if (tid < 3) a.counts[b * 3 + tid] = tot3[tid];
    }
}

// The last pass in position mode: the pairwise records need, for every pair, the OTHER pairs of its bucket with the same
// hash -- a GROUPING, not an order.  One workgroup per bucket puts the bucket's pairs into an LDS hash table keyed by the
// hash itself (open addressing from the pair's fine slot, a linear function of the key that spreads ~0.4 distinct hashes
// per slot; a 64-bit compare-and-swap claims a slot or finds the hash there), the pairs of one hash chained through their
// slot's head word; behind ONE barrier every pair walks its chain and stores its record.  Three barriers per bucket
// instead of the eight of a counting sort + ranking + placement (k_bucket_sort, which did this job first: 0.68 ms at
// configs[3], this 0.55), and no limit on the holders of one hash short of the bucket's capacity (a chain of m pairs is
// walked m times: m^2 LDS reads).
// Hashes with more than four holders (or any, when references do not fit 21-bit fields) keep their holders in the
// bucket's list area: the chain's first pair reserves room, every pair writes its reference at its rank among the chain's
// pair indices.
// XCD x (blockIdx % 8 on this chip) takes the x-th EIGHTH of the buckets, front to back: that walk is what lets the L2s
// merge the record stores -- a sketch's elements inside a first-level region of the sort are ~36 consecutive CSR positions,
// and the region's ~140 buckets are then all processed on the same XCD within a short time of each other
// (scripts/probes/scatter_probe.hip: 27 M 8-byte records in 0.30 ms this way, 0.39 ms with bucket = blockIdx, 0.69 ms at
// isolated positions, 1.3 ms with non-temporal stores -- and the counting atomic + the store of k_idx_emit /
// k_pair_transpose before: 0.76 + 0.56 ms).
__global__ void __launch_bounds__(BKT_THREADS) k_bucket_group(const BucketArgs a) {
    constexpr u32 NONE = 0xffffffffu;
    __shared__ __attribute__((aligned(16))) u64 tkey[BKT_CAP];   // hash + 1 of the slot's group (0: free)
    __shared__ __attribute__((aligned(16))) u32 thead[BKT_CAP];  // the group's last-come pair; later, for a listed group, where its holders start in the list
    __shared__ u32 eref[BKT_CAP];     // pair -> reference
    __shared__ u16 enext[BKT_CAP];    // pair -> the pair that came before it in its group (0xffff: none)
    __shared__ u32 tot3[3];
    __shared__ u32 lcount, has_list;
    static_assert(BKT_CAP <= 65535, "pair indices in 16 bits");
    const u32 tid = threadIdx.x;
    const u64 b = (u64)(blockIdx.x & 7u) * a.per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= a.per_xcd || b >= a.nb) return;
    const u64 sv = a.stride_v ? a.stride_v : a.cap_in;
    u64 key[BKT_ITEMS];
    u32 val[BKT_ITEMS], rf[BKT_ITEMS], slot[BKT_ITEMS];
#if defined(YH_GROUP_SPEC_LOADS) && YH_GROUP_SPEC_LOADS
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        key[k] = a.in_k[b * a.cap_in + e];
        val[k] = a.in_v[b * sv + e];
    }
#endif
    if (a.cnt[b] > BKT_CAP && tid == 0) atomicOr(a.flags, 2u);
    const u32 n = min(a.cnt[b], BKT_CAP);
    if (n == 0) return;
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        slot[k] = NONE;
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 16)  // timing-only build: not even the loads
        if (e < n) { key[k] = b + e; val[k] = e; }
#else
#if !(defined(YH_GROUP_SPEC_LOADS) && YH_GROUP_SPEC_LOADS)  // (else: requested above, before the bucket's count was known)
        if (e < n) { key[k] = a.in_k[b * a.cap_in + e]; val[k] = a.in_v[b * sv + e]; }
#endif
#endif
    }
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 8)  // timing-only build: the loads and nothing else
    { u64 x = 0;
#pragma unroll
      for (u32 k = 0; k < BKT_ITEMS; ++k) if (k * BKT_THREADS + tid < n) x += key[k] + val[k];
      if (x == 0x123456789abcdefull) a.rec[0] = x;
      return; }
#endif
    for (u32 i = tid; i < BKT_CAP / 2; i += BKT_THREADS) reinterpret_cast<uint4*>(tkey)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (u32 i = tid; i < BKT_CAP / 4; i += BKT_THREADS) reinterpret_cast<uint4*>(thead)[i] = make_uint4(NONE, NONE, NONE, NONE);
    if (tid < 3) tot3[tid] = 0;
    if (tid == 3) { lcount = 0; has_list = 0; }
    __syncthreads();
    u32 c0 = 0;
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        bool won = false;
        if (e < n) {
            u32 s;
            if (a.rem_bits) {  // (uniform) the pair brings its reference; the table's key is the hash's low bits
                rf[k] = (u32)(key[k] >> a.rem_bits);
                key[k] &= (1ull << a.rem_bits) - 1ull;
                s = (u32)((key[k] * 0x9E3779B97F4A7C15ull) >> (64 - BKT_SLOT_BITS));
            } else {
                rf[k] = ref_of(val[k], a.ref_tab, a.ref_off);
                s = fine_of(key[k], a.lsh, a.mul_fine) & (BKT_CAP - 1u);
            }
            eref[e] = rf[k];
            const unsigned long long k1 = key[k] + 1ull;  // (the fused path is taken only where the largest hash is below 2^64 - 1)
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 4)  // timing-only build: no table inserts
            won = (k1 & 1ull) != 0; tkey[s] = k1; slot[k] = s; enext[e] = 0xffffu; thead[s] = e;
#else
            for (u32 probe = 0; probe < BKT_CAP; ++probe) {
                const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tkey[s]), 0ull, k1);
                if (old == 0ull) { won = true; break; }
                if (old == k1) break;
                s = (s + 1u) & (BKT_CAP - 1u);
            }
            slot[k] = s;
            enext[e] = (u16)atomicExch(&thead[s], e);  // (NONE -> 0xffff)
#endif
        }
        c0 += (u32)__popcll(__ballot(won));
    }
    __syncthreads();
    u32 c1 = 0, c2 = 0;
    u32 glen[BKT_ITEMS], grank[BKT_ITEMS];
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        bool first = false, shared = false;
        glen[k] = 0;
        grank[k] = 0;
        if (slot[k] != NONE) {
            u32 others[3] = {0u, 0u, 0u};
            u32 cnt = 0, rank = 0;
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 2)  // timing-only build: the chain is not walked
            cnt = enext[e] != 0xffffu ? 1u : 0u; others[0] = eref[e ^ 1u];
#else
            for (u32 j = thead[slot[k]]; j != 0xffffu && j != NONE; j = enext[j]) {
                if (j == e) continue;
                if (cnt < 3) others[cnt] = eref[j];
                ++cnt;
                rank += j < e ? 1u : 0u;
            }
#endif
            const u32 len = cnt + 1u;
            first = enext[e] == 0xffffu;  // (the pair that claimed the chain: one per group)
            shared = len >= 2u;
            if (shared) {
                if (len <= 4u && a.inline_ok) {
                    const u64 r = (u64)(others[0] + 1u) | (cnt > 1 ? (u64)(others[1] + 1u) << 21 : 0ull) | (cnt > 2 ? (u64)(others[2] + 1u) << 42 : 0ull);
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 1)  // timing-only build: no record stores (results wrong)
                    if (r == 0x123456789abcdefull) a.rec[val[k]] = r;
#else
                    if (val[k] < a.n_pos) a.rec[val[k]] = r;
#endif
                } else {
                    glen[k] = len;
                    grank[k] = rank;
                    has_list = 1u;
                }
            }
        }
        c1 += (u32)__popcll(__ballot(first && shared));
        c2 += (u32)__popcll(__ballot(shared));
    }
    if ((tid & 63u) == 0) { atomicAdd(&tot3[0], c0); atomicAdd(&tot3[1], c1); atomicAdd(&tot3[2], c2); }
    __syncthreads();
    unsigned long long* trow = a.totals + (size_t)(b % TOT_LANES) * 8u;
    if (tid < 3 && tot3[tid]) atomicAdd(&trow[tid], (unsigned long long)tot3[tid]);
    if (tid == 3) atomicAdd(&trow[3], (unsigned long long)n);
    if (!has_list) return;  // (uniform: read behind the barrier)
    // the listed groups: the chain's first pair reserves the group's room in the bucket's list area
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        if (slot[k] != NONE && glen[k] && enext[e] == 0xffffu) thead[slot[k]] = atomicAdd(&lcount, glen[k]);
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k)
        if (slot[k] != NONE && glen[k]) {
            const u64 start = b * sv + thead[slot[k]];
            a.list[start + grank[k]] = rf[k];
            if (val[k] < a.n_pos) a.rec[val[k]] = (1ull << 63) | ((u64)glen[k] << 40) | start;
        }
}

// The grouping pass for PACKED pairs (round 5): 32-bit table words, no chains.
// What the chained table above costs was taken apart with timing-only builds (profiles/r05/ablate_group*.txt, sweep_spec*.txt,
// ablate_g4*.txt): of ~500 us at configs[3], the pairs' loads WAITING FOR THE BUCKET'S COUNT ~300 (requested before the count
// is known -- slots beyond it hold garbage nobody uses -- the kernel without any table work falls from 496 to 194 us); the
// chain walk 155 (every pair follows its group's links: dependent LDS reads, a wave as slow as its longest chain); the
// inserts 100; the record stores 57.  With either half gone the other still fills the time, so both change.  And what an
// insert costs is the NUMBER OF RETURNING LDS ATOMICS A WAVE ISSUES, ~8-9 cycles each for 64-bit operands whatever the
// active lanes (a first chain-free form kept the holders in a second 64-bit word per slot, joined by compare-and-swap: 869 us
// -- 207 for those joins alone, 325 for the side table of its groups of more than four).  Here:
//   ekey[e]  = the pair as it came (hash remainder | reference << rem_bits)          plain 8-byte store
//   tidx[s]  = claimer's pair index + 1 | members that joined << 16                  32-bit: ONE compare-and-swap claims a
//              slot or finds it taken -- then the claimer's ekey says whether it is this pair's hash (probe on if not) --
//              and a member's rank is ONE 32-bit add
//   tmem[s]  = the pair indices of the first three members                          plain 2-byte stores
// Behind the barrier a pair reads its slot's word and members (independent reads), then the up to three other pairs' ekey:
// no loop, no divergence beyond "shared or not".  Groups of more than four holders go to the bucket's list area: every
// holder knows its rank (claimer 0, members in order of arrival), the claimer reserves the room.
// Measured (profiles/r05/sweep_g5.txt, ablate_g5.txt): 472 us against 499 for the chained table on the same pairs; of them
// the inserts 283 (with keys that never collide: ~100 -- the probing of a table at load 0.4-0.6, where a wave is as slow
// as its unluckiest lane, is what is left), the reads behind the barrier 27, the record stores 24, loads + clears +
// barriers 189.
__global__ void __launch_bounds__(BKT_THREADS) k_bucket_group5(const BucketArgs a) {
    __shared__ __attribute__((aligned(16))) u64 ekey[BKT_CAP];
    __shared__ __attribute__((aligned(16))) u32 tidx[BKT_CAP];
    __shared__ u16 tmem[BKT_CAP * 3];
    __shared__ u32 tot3[3];
    __shared__ u32 lcount, has_list;
    const u32 tid = threadIdx.x;
    const u64 b = (u64)(blockIdx.x & 7u) * a.per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= a.per_xcd || b >= a.nb) return;
    const u32 c_raw = a.cnt[b];  // (asked for first: it is back by the time the first half of the pairs is)
    u64 key[BKT_ITEMS];
    u32 val[BKT_ITEMS], st[BKT_ITEMS];  // st: slot | rank << 12 (rank 0: claimed the slot; members: 1 + order of arrival)
    // The first half of the capacity is requested before the bucket's count is known (the capacity is allocated for every bucket;
    // slots beyond the count hold garbage nobody uses) -- a bucket holds ~2 560 pairs, so these are nearly all real --, the second
    // half behind the count, which has arrived meanwhile: only the pairs that exist.  (All four quarters up front: 0.96 GB read
    // for 0.60 GB of pairs at configs[3].)
#if defined(YH_GROUP_SPEC_ALL) && YH_GROUP_SPEC_ALL
    constexpr u32 SPEC_ITEMS = BKT_ITEMS;
#elif defined(YH_GROUP_SPEC_ITEMS)
    constexpr u32 SPEC_ITEMS = YH_GROUP_SPEC_ITEMS;
#else
    constexpr u32 SPEC_ITEMS = BKT_ITEMS / 2;
#endif
#pragma unroll
    for (u32 k = 0; k < SPEC_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 16)  // timing-only build: not even the loads
        key[k] = (b * 0x9E3779B97F4A7C15ull + e * 0x7F4A7C15ull) >> 3; val[k] = e;
#else
#if YH_GROUP_NT & 1  // the bucket's pairs are read once: non-temporal, so that they do not push the record lines the XCD's L2 is
        key[k] = __builtin_nontemporal_load(&a.in_k[b * a.cap_in + e]);   // collecting out before they are whole (round 6: the pass
        // wrote 0.71 GB for 0.40 GB of records; with these 0.61 -- profiles/r06/sweep_group_nt.txt; the time is the same)
        val[k] = __builtin_nontemporal_load(&a.in_v[b * a.stride_v + e]);
#else
        key[k] = a.in_k[b * a.cap_in + e];
        val[k] = a.in_v[b * a.stride_v + e];
#endif
#endif
    }
    for (u32 i = tid; i < BKT_CAP / 4; i += BKT_THREADS) reinterpret_cast<uint4*>(tidx)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 3) tot3[tid] = 0;
    if (tid == 3) { lcount = 0; has_list = 0; }
    if (c_raw > BKT_CAP) {  // (uniform) more pairs than a bucket holds -- a hash thousands of references share: the bucket is marked
        // and ALL its pairs are grouped from the side list (k_spill_collect / k_spill_group)
        if (tid == 0) {
            if (a.over_bits) { atomicOr(&a.over_bits[b >> 5], 1u << (b & 31u)); atomicAdd(&a.flags[2], 1u); }
            else atomicOr(a.flags, 2u);
        }
        return;
    }
    const u32 n = c_raw;
#pragma unroll
    for (u32 k = SPEC_ITEMS; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        key[k] = 0; val[k] = 0;
#if YH_GROUP_NT & 1
        if (e < n) { key[k] = __builtin_nontemporal_load(&a.in_k[b * a.cap_in + e]); val[k] = __builtin_nontemporal_load(&a.in_v[b * a.stride_v + e]); }
#else
        if (e < n) { key[k] = a.in_k[b * a.cap_in + e]; val[k] = a.in_v[b * a.stride_v + e]; }
#endif
    }
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        if (e < n) ekey[e] = key[k];
    }
    __syncthreads();
    const u64 rmask = (1ull << a.rem_bits) - 1ull;
    // (Measured and dropped, profiles/r05/sweep_g5q.txt: a QUEUE PER LANE -- every step a lane probes one slot for the first of
    // its pairs that is not placed yet, so that a wave makes max-over-lanes(sum of the lane's probes) steps instead of
    // sum-over-pairs(max-over-lanes(probes)) -- 491-499 us against 472 for the four loops below.)
    u32 c0 = 0;
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        bool won = false;
        st[k] = 0xffffffffu;
        if (e < n) {
            const u64 rem = key[k] & rmask;
            const u64 mixed = rem * 0x9E3779B97F4A7C15ull;
            u32 s = (u32)(mixed >> (64 - BKT_SLOT_BITS));
#if defined(YH_GROUP_LINEAR) && YH_GROUP_LINEAR
            const u32 step = 1u;
#else
            // DOUBLE HASHING: the probe step is a second (odd) function of the hash -- a wave makes max-over-lanes(probes) steps per
            // pair, and two hashes that collide do not share the rest of their probe sequences as they do with step 1 (simulated at
            // this load: 8.9 wave steps per four pairs against 11.6; the inserts are ~25 us per wave step at configs[3])
            const u32 step = ((u32)(mixed >> 20) & (BKT_CAP - 1u)) | 1u;
#endif
            u32 rank = 0;
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 4)  // timing-only build: no table inserts
            tidx[s] = e + 1u; won = (rem & 1ull) != 0;
            for (u32 probe = 0; probe < 0; ++probe) {
#else
            for (u32 probe = 0; probe < BKT_CAP; ++probe) {
#endif
                const u32 old = atomicCAS(&tidx[s], 0u, e + 1u);
                if (old == 0u) { won = true; break; }
                if ((ekey[(old & 0xffffu) - 1u] & rmask) == rem) {  // this pair's hash: join
                    const u32 r = atomicAdd(&tidx[s], 1u << 16) >> 16;  // members in front of this one
                    if (r < 3u) tmem[s * 3u + r] = (u16)e;
                    rank = r + 1u;
                    break;
                }
                s = (s + step) & (BKT_CAP - 1u);
            }
            st[k] = s | (rank << 12);
        }
        c0 += (u32)__popcll(__ballot(won));
    }
    __syncthreads();
    u32 c1 = 0, c2 = 0;
    u32 listed = 0;
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        bool first = false, shared = false;
        if (st[k] != 0xffffffffu) {
            const u32 s = st[k] & 0xfffu, rank = st[k] >> 12;
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 2)  // timing-only build: the slot is not read again
            const u32 w = (u32)key[k] & 0x1ffffu;
#else
            const u32 w = tidx[s];
#endif
            const u32 members = w >> 16;
            first = rank == 0u;
            shared = members != 0u;
            // Round 6: a pair whose hash nobody else holds stores its (zero) record too -- every CSR position belongs to exactly one
            // pair of exactly one bucket, so the records need no clearing pass in front of the distribution (0.40 GB of zeros at
            // configs[3], written by k_piece_bounds until round 5), and a record line that this XCD's L2 collects from its buckets'
            // pairs now leaves it WHOLE (2.7e7 records scattered over 5e7 positions cost 1.9 x their bytes in partly written lines).
            if (!shared && a.write_all && val[k] < a.n_pos) a.rec[val[k]] = 0ull;
            if (shared) {
                if (members <= 3u && a.inline_ok) {
                    // the other holders: the claimer (unless that is this pair) and the members but this one
                    const u32 m0 = tmem[s * 3u], m1 = tmem[s * 3u + 1u], m2 = tmem[s * 3u + 2u];
                    u32 o[3];
                    u32 cnt = 0;
                    if (rank != 0u) o[cnt++] = (w & 0xffffu) - 1u;
                    if (rank != 1u) o[cnt++] = m0;
                    if (members > 1u && rank != 2u) o[cnt++] = m1;
                    if (members > 2u && rank != 3u) o[cnt++] = m2;
                    // (cnt = members: one of the up to four is this pair)
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 2)
                    u64 r = m0 + m1 + m2 + o[0] + 1;
#else
                    u64 r = (u64)((u32)(ekey[o[0]] >> a.rem_bits) + 1u);
                    if (cnt > 1u) r |= (u64)((u32)(ekey[o[1]] >> a.rem_bits) + 1u) << 21;
                    if (cnt > 2u) r |= (u64)((u32)(ekey[o[2]] >> a.rem_bits) + 1u) << 42;
#endif
#if defined(YH_ABLATE_GROUP) && (YH_ABLATE_GROUP & 1)  // timing-only build: no record stores (results wrong)
                    if (r == 0x123456789abcdefull) a.rec[val[k]] = r;
#else
                    if (val[k] < a.n_pos) a.rec[val[k]] = r;
#endif
                } else {
                    listed |= 1u << k;
                    has_list = 1u;
                }
            }
        }
        c1 += (u32)__popcll(__ballot(first && shared));
        c2 += (u32)__popcll(__ballot(shared));
    }
    if ((tid & 63u) == 0) { atomicAdd(&tot3[0], c0); atomicAdd(&tot3[1], c1); atomicAdd(&tot3[2], c2); }
    __syncthreads();
    unsigned long long* trow = a.totals + (size_t)(b % TOT_LANES) * 8u;
    if (tid < 3 && tot3[tid]) atomicAdd(&trow[tid], (unsigned long long)tot3[tid]);
    if (tid == 3) atomicAdd(&trow[3], (unsigned long long)n);
    if (!has_list) return;  // (uniform: read behind the barrier)
    // groups of more than four holders (or any group, when references do not fit the inline fields): the claimer reserves the
    // group's room in the bucket's list area and leaves its start where the members' indices were (nobody reads those now:
    // every pair wrote its inline record, or found out that it has none, before the barrier above)
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k)
        if (((listed >> k) & 1u) && (st[k] >> 12) == 0u) {
            const u32 s = st[k] & 0xfffu;
            const u32 start = atomicAdd(&lcount, 1u + (tidx[s] >> 16));
            tmem[s * 3u] = (u16)start;  // (a bucket's list area = its capacity: 4 096 entries, 12 bits)
        }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k)
        if ((listed >> k) & 1u) {
            const u32 s = st[k] & 0xfffu, rank = st[k] >> 12, len = 1u + (tidx[s] >> 16);
            const u64 start = b * a.stride_v + tmem[s * 3u];
            a.list[start + rank] = (u32)(key[k] >> a.rem_bits);
            if (val[k] < a.n_pos) a.rec[val[k]] = (1ull << 63) | ((u64)len << 40) | start;
        }
}

// (Round 5, measured and dropped -- profiles/r05/sweep_group_w.txt: PERSISTENT workgroups, 64 per XCD, that request the next
// bucket's pairs into registers before they group the current one, on the theory that a bucket's ~13 us are mostly the
// latency of its loads and the drain of its stores: 667 us against 505 for one workgroup per bucket (32 per XCD: 945; 128:
// 696) -- 64 registers with the prefetched pairs means spills, the table needs a fourth barrier before it is cleared, and
// the hardware's own dispatch of the next workgroup into a freed slot was already hiding what there was to hide.)
// =====================================================================================================================
// The distribution WITHOUT a first level (round 5): `yacht train`'s handle, position mode.
// Every sketch is ASCENDING and bucket(h) is monotone, so the hashes of sketch i that fall into first-level region r are a
// contiguous PIECE [bnd[r][i], bnd[r + 1][i]) of the CSR: region r does not have to be written out and read again (12 B
// out + 12 B in per pair: k_part<1> 0.45-0.53 ms + the input side of k_part<2> at configs[3]) -- it can be READ IN PLACE.
//   k_piece_bounds   one streaming pass over the CSR (a wave per sketch, 8-byte coalesced loads): where every region starts
//                    in every sketch (transposed through LDS into region-major rows), the ordering check, and the clearing
//                    of the pairwise records -- 8 B read + 8 B written per pair, nothing scattered
//   k_piece_part     a tile = (region r, a group of S sketches): their pieces of r (~36 hashes = 288 contiguous bytes each at
//                    configs[3]) are read straight from the CSR and distributed over the region's P2 buckets exactly as
//                    k_part<2> does it (LDS histogram, one reserving atomic per (tile, bin), staged, coalesced runs) -- as
//                    PACKED pairs: the low rem_bits bits of the hash (inside a bucket the hashes span less than
//                    2^rem_bits) with the REFERENCE id above them (the tile knows whose piece it reads: k_bucket_group's
//                    position -> reference look-ups are gone), and the position.  Still 12 bytes.
//                    All tiles of a region run on ONE XCD, consecutively (blockIdx % 8 = region % 8): the runs of a bucket
//                    are completed in that XCD's L2 before they go to HBM.
// Traffic per pair: 8 + 8 (bounds, clear) + 8 + 12 (distribution) against 8 + 12 + 8 and 12 + 12.
constexpr u32 PC_MAX_S = 1023;   // sketches per tile (one thread each for the prefix of their piece lengths)
#ifndef YH_PC_BOUND_THREADS
#define YH_PC_BOUND_THREADS 512
#endif
constexpr u32 PC_BOUND_THREADS = YH_PC_BOUND_THREADS;  // (16 waves: with 4 per workgroup ~5 waves per CU had 10 KB of loads in flight each -- 1.3 TB/s)
#ifndef YH_PC_BOUND_U
#define YH_PC_BOUND_U 8
#endif
// 512-byte loads a wave has in flight.  Round 6 (scripts/sweep_bounds.sh, profiles/r06/sweep_bounds.txt): without its zero-fill the
// pass read configs[3]'s 0.40 GB in ~165 us -- 2.4 TB/s: 313 workgroups of 16 waves on 256 CUs (a second, 22 %-full round of
// workgroups) and two sketches of 20 dependent load rounds per wave.  One sketch per wave (SK = 8 sketches per 8-wave workgroup:
// 1 250 workgroups) and eight loads in flight: build kernels 0.985 -> 0.920 ms.
constexpr u32 PC_BOUND_U = YH_PC_BOUND_U;
struct PieceArgs {
    const u64* values;
    const u64* off;      // CSR offsets [N + 1]
    u64 n_refs;
    u32* bnd;            // [(P1 + 1)][n_pad]: bnd[r][i] = first position of sketch i whose region is >= r (bnd[0][i] = off[i], bnd[P1][i] = off[i + 1])
    u64 n_pad;
    u32 P1, P2, S, Gn, SK;
    u64 mul;
    u32 lsh, rem_bits;
    u32 inv_p2;          // ceil(2^32 / P2): bucket / P2 = (bucket * inv_p2) >> 32 for every bucket < NB
    u64* rec_clear;
    u64* out_a;          // [NB][stride_k] packed (hash remainder | reference << rem_bits)
    u32* out_p;          // [NB][stride_v] positions
    u64 stride_k, stride_v;  // elements between two buckets: BKT_CAP + a pad (see yh_pieces)
    // the side list of the buckets that overflowed (k_spill_collect): all their pairs, fetched from the sketches once more
    u64* spill_a; u32* spill_p; u32* spill_b; u64 spill_cap;
    const u32* over_bits;  // bit b: bucket b overflowed (set by the pass behind the distribution)
    u32* out_cnt;        // [NB]
    u32* flags;          // |= 8: a sketch is not ascending (bounds pass) or a piece holds a hash of another region
    u32 check_order;
    u32 emit_ref;        // != 0: the pairs leave as (whole hash, reference) -- the input of k_bucket_sort (every handle but `yacht train`'s)
};

__global__ void __launch_bounds__(PC_BOUND_THREADS) k_piece_bounds(const PieceArgs a, u64 r0, u64 r1) {
    extern __shared__ u32 bl[];  // [SK][P1 + 1]
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nwv = PC_BOUND_THREADS / 64;
    const u64 i0 = r0 + (u64)blockIdx.x * a.SK;
    const u32 nsk = (u32)min((u64)a.SK, r1 - i0);
    const u32 W = a.P1 + 1;
    bool bad = false;
    for (u32 s = wv; s < nsk; s += nwv) {  // (wave-uniform)
        u32* row = bl + (size_t)s * W;
        const u64 b = a.off[i0 + s], e = a.off[i0 + s + 1];
        if (e < b) { bad = true; for (u32 r = lane; r < W; r += 64) row[r] = (u32)b; continue; }
        u32 carry_reg = 0xffffffffu;  // region of the element in front (none yet: -1)
        u64 carry_h = 0;
        bool have = false;
        for (u64 p0 = b; p0 < e; p0 += 64u * PC_BOUND_U) {
            u64 h[PC_BOUND_U];
#pragma unroll
            for (u32 u = 0; u < PC_BOUND_U; ++u) {
                const u64 p = p0 + u * 64u + lane;
                h[u] = p < e ? a.values[p] : 0ull;
            }
#pragma unroll
            for (u32 u = 0; u < PC_BOUND_U; ++u) {
                const u64 pw = p0 + u * 64u;  // (uniform)
                if (pw >= e) break;
                const u64 p = pw + lane;
                const bool in = p < e;
                u32 reg = (u32)(((u64)bucket_of(h[u], a.lsh, a.mul) * a.inv_p2) >> 32);  // bucket / P2 (pc_geometry checks the reciprocal)
                reg = reg >= a.P1 ? a.P1 - 1u : reg;
                u32 up_reg = (u32)__shfl_up((int)reg, 1);
                u64 up_h = ((u64)(u32)__shfl_up((int)(u32)(h[u] >> 32), 1) << 32) | (u32)__shfl_up((int)(u32)h[u], 1);
                bool up_have = true;
                if (lane == 0) { up_reg = carry_reg; up_h = carry_h; up_have = have; }
                if (in) {
                    if (a.rec_clear) a.rec_clear[p] = 0;
                    if (up_have && !(up_h < h[u])) bad = true;
                    // this element opens every region behind its predecessor's up to its own (none when the sketch is not
                    // ascending here: flagged above)
                    for (u32 r = up_reg + 1u; r <= reg; ++r) row[r] = (u32)p;
                }
                const u32 last = (u32)min((u64)63, e - pw - 1);  // (uniform) the wave's last element of this load
                carry_reg = (u32)__shfl((int)reg, (int)last);
                carry_h = ((u64)(u32)__shfl((int)(u32)(h[u] >> 32), (int)last) << 32) | (u32)__shfl((int)(u32)h[u], (int)last);
                have = true;
            }
        }
        for (u32 r = carry_reg + 1u + lane; r < W; r += 64) row[r] = (u32)e;  // the regions behind the last element (all of them: an empty sketch)
    }
    if (a.check_order && __ballot(bad) != 0ull && lane == 0) atomicOr(a.flags, 8u);
    __syncthreads();
    for (u32 idx = threadIdx.x; idx < W * nsk; idx += PC_BOUND_THREADS) {  // region-major rows: runs of nsk consecutive sketches
        const u32 r = idx / nsk, s = idx % nsk;
        a.bnd[(u64)r * a.n_pad + i0 + s] = bl[(size_t)s * W + r];
    }
}

// (EMIT_REF is a template parameter, and pairs that find their bucket full are NOT appended to the side list here: either one
// as a run-time branch inside this kernel's loops cost it a third -- 480-500 us against 340, whatever form the branch took:
// profiles/r05/sweep_part_branches.txt.  The rare case is k_spill_collect's.)
template <bool EMIT_REF>
__global__ void __launch_bounds__(PART_THREADS) k_piece_part(const PieceArgs a) {
    __shared__ u64 skey[PART_TILE];
    __shared__ u32 sval[PART_TILE];
    __shared__ u16 sbin[PART_TILE];
    __shared__ u32 hist[PART_MAX_BINS], loc[PART_MAX_BINS], gbase[PART_MAX_BINS];
    __shared__ u32 pre[PC_MAX_S + 1], pstart[PC_MAX_S + 1];
    __shared__ u32 wtot[17];
    const u32 tid = threadIdx.x;
    const u32 k = blockIdx.x >> 3;
    const u32 r = (blockIdx.x & 7u) + 8u * (k / a.Gn), g = k % a.Gn;  // XCD x takes the regions x, x + 8, ...: one after the other
    if (r >= a.P1) return;
    const u64 s0 = (u64)g * a.S;
    const u32 ns = (u32)min((u64)a.S, a.n_refs - s0);
    if (tid <= ns) {
        u32 len = 0;
        if (tid < ns) {
            const u32 pa = a.bnd[(u64)r * a.n_pad + s0 + tid], pb = a.bnd[(u64)(r + 1) * a.n_pad + s0 + tid];
            pstart[tid] = pa;
            len = pb > pa ? pb - pa : 0u;
        }
        pre[tid] = len;
    }
    __syncthreads();
    block_scan_inplace<1>(pre, ns + 1, wtot);  // pre[j] = elements of the pieces in front of j; pre[ns] = all of them
    const u32 total = pre[ns];
    const u64 rem_mask = (1ull << a.rem_bits) - 1ull;
    constexpr u32 BINS_PER_THREAD = (PART_MAX_BINS + PART_THREADS - 1) / PART_THREADS;
    bool bad = false;
    for (u32 base = 0; base < total; base += PART_TILE) {  // (workgroup-uniform; one round unless the group's pieces outgrow a tile)
        const u32 tile_n = min(PART_TILE, total - base);
        for (u32 b = tid; b < a.P2; b += PART_THREADS) hist[b] = 0;
        __syncthreads();
        u64 key[PART_ITEMS];
        u32 val[PART_ITEMS], bin[PART_ITEMS], rank[PART_ITEMS], pos[PART_ITEMS], who[PART_ITEMS];
#pragma unroll
        for (u32 q = 0; q < PART_ITEMS; ++q) {
            const u32 e = base + q * PART_THREADS + tid;
            pos[q] = 0xffffffffu;
            if (e < total) {
                u32 lo = 0, hi = ns;  // the piece of element e: the largest j with pre[j] <= e (pre[ns] = total > e)
                while (hi - lo > 1u) {
                    const u32 mid = (lo + hi) >> 1;
                    if (pre[mid] <= e) lo = mid; else hi = mid;
                }
                who[q] = lo;
                pos[q] = pstart[lo] + (e - pre[lo]);
            }
        }
#pragma unroll
        for (u32 q = 0; q < PART_ITEMS; ++q)
#if defined(YH_ABLATE_PART) && (YH_ABLATE_PART & 4)  // timing-only build: the pieces are not read (keys made up inside the region)
            if (pos[q] != 0xffffffffu) key[q] = ((u64)r * a.P2 + (pos[q] * 2654435761u) % a.P2) * ((~0ull) / ((u64)a.P1 * a.P2 * 1000ull)) + pos[q];
#else
            if (pos[q] != 0xffffffffu) key[q] = a.values[pos[q]];
#endif
#pragma unroll
        for (u32 q = 0; q < PART_ITEMS; ++q) {
            bin[q] = 0xffffffffu;
            if (pos[q] != 0xffffffffu) {
                const u32 bk = bucket_of(key[q], a.lsh, a.mul);
                u32 reg = (u32)(((u64)bk * a.inv_p2) >> 32);
                reg = reg >= a.P1 ? a.P1 - 1u : reg;
                u32 b = bk - reg * a.P2;
                if (reg != r) { bad = true; b = 0; }  // (a sketch that is not ascending: the whole attempt is refused)
                if (b >= a.P2) b = a.P2 - 1u;
                bin[q] = b;
                if constexpr (EMIT_REF) {
                    val[q] = (u32)(s0 + who[q]);
                } else {
                    val[q] = pos[q];
                    key[q] = (key[q] & rem_mask) | ((u64)(s0 + who[q]) << a.rem_bits);
                }
                rank[q] = atomicAdd(&hist[b], 1u);
            }
        }
        __syncthreads();
        u32 g_mine[BINS_PER_THREAD], c_mine[BINS_PER_THREAD];
#pragma unroll
        for (u32 q = 0; q < BINS_PER_THREAD; ++q) {
            const u32 b = q * PART_THREADS + tid;
            g_mine[q] = 0;
            c_mine[q] = 0;
            if (b < a.P2) {
                const u32 c = hist[b];
                c_mine[q] = c;
#if defined(YH_ABLATE_PART) && (YH_ABLATE_PART & 2)  // timing-only build: no reserving atomics (every tile writes at the bucket's start)
                if (c) g_mine[q] = 0;
#else
                if (c) g_mine[q] = atomicAdd(&a.out_cnt[(u64)r * a.P2 + b], c);
#endif
                loc[b] = c;
            }
        }
        __syncthreads();
        block_scan_inplace<BINS_PER_THREAD>(loc, a.P2, wtot);
#pragma unroll
        for (u32 q = 0; q < PART_ITEMS; ++q)
            if (bin[q] != 0xffffffffu) {
                const u32 s = loc[bin[q]] + rank[q];
                skey[s] = key[q];
                sval[s] = val[q];
                sbin[s] = (u16)bin[q];
            }
#pragma unroll
        for (u32 q = 0; q < BINS_PER_THREAD; ++q) {
            const u32 b = q * PART_THREADS + tid;
            if (b < a.P2) gbase[b] = g_mine[q] - loc[b];
        }
        __syncthreads();
#pragma unroll
        for (u32 q = 0; q < PART_ITEMS; ++q) {
            const u32 s = q * PART_THREADS + tid;
            if (s < tile_n) {
                const u32 b = sbin[s];
                const u64 at = (u32)(gbase[b] + s);
                if (at < BKT_CAP) {  // (a pair that finds its bucket full is dropped HERE: the bucket's count says so, the pass behind this
                    // one marks the bucket, and k_spill_collect fetches ALL its pairs from the sketches again -- the rare case pays)
                    const u64 bkt = (u64)r * a.P2 + b;
#if defined(YH_ABLATE_PART) && (YH_ABLATE_PART & 1)  // timing-only build: the pairs are not stored
                    if (skey[s] == 0x123456789abcdefull) a.out_a[bkt * a.stride_k + at] = skey[s] + sval[s];
#else
#if YH_GROUP_NT & 2  // (measured: the runs' stores no longer merge in the L2 -- 0.62 -> 1.08 GB written, +55 us)
                    __builtin_nontemporal_store(skey[s], &a.out_a[bkt * a.stride_k + at]);
                    __builtin_nontemporal_store(sval[s], &a.out_p[bkt * a.stride_v + at]);
#else
                    a.out_a[bkt * a.stride_k + at] = skey[s];
                    a.out_p[bkt * a.stride_v + at] = sval[s];
#endif
#endif
                }
            }
        }
        __syncthreads();
    }
    if (__ballot(bad) != 0ull && (tid & 63u) == 0) atomicOr(a.flags, 8u);
}

// ---- the side list of the overflowed buckets ------------------------------------------------------------------------------
// A bucket holds 4 096 pairs; a k-mer that 40 000 references share (conserved rRNA 31-mers across GTDB) puts ten times
// that into one.  Until round 4 ONE such bucket sent the whole database to rocPRIM's radix sort and `yacht train` off its
// fused path.  Now the bucket's pairs go to a side list -- the late ones from k_piece_part, the rest from k_bucket_group5 --
// which alone is sorted by (bucket, hash remainder) with rocPRIM and grouped by the two kernels below: every pair finds its
// run of equal keys by binary search (the list is small), its rank is its place in the run, the run IS the group's holder
// list (yh_db::d_fz_list2), the record names it.  Everything else stays where it was.
// every pair of the marked buckets, fetched from the sketches once more (a wave per sketch, as the bounds pass reads them):
// 8 H bytes read again (twice: see below) -- only when some bucket overflowed
template <bool EMIT_REF>
__global__ void __launch_bounds__(256) k_spill_collect(const PieceArgs a) {
    const u32 lane = threadIdx.x & 63u;
    const u64 sk = blockIdx.x * 4ull + (threadIdx.x >> 6);
    if (sk >= a.n_refs) return;
    const u64 b0 = a.off[sk], e0 = a.off[sk + 1];
    const u64 rem_mask = (1ull << a.rem_bits) - 1ull;
    auto marked = [&](u64 h) -> u32 {  // the pair's bucket when that bucket is marked, else ~0
        const u32 bk = bucket_of(h, a.lsh, a.mul);
        return ((a.over_bits[bk >> 5] >> (bk & 31u)) & 1u) ? bk : 0xffffffffu;
    };
    // Two walks over the sketch: the first COUNTS the wave's pairs of marked buckets, ONE atomic reserves their room, the second
    // writes them.  (One atomic per 64 pairs that held a marked one -- every other wave step at a database with 50 hot k-mers --
    // were 3.7e5 adds to one word: 4.0 of the build's 5.4 ms; the second walk finds the sketch in the L2.)
    u32 mine = 0;
    for (u64 p0 = b0; p0 < e0; p0 += 64u * PC_BOUND_U) {
        u64 h[PC_BOUND_U];
#pragma unroll
        for (u32 u = 0; u < PC_BOUND_U; ++u) {
            const u64 p = p0 + u * 64u + lane;
            h[u] = p < e0 ? a.values[p] : 0ull;
        }
#pragma unroll
        for (u32 u = 0; u < PC_BOUND_U; ++u) {
            const u64 p = p0 + u * 64u + lane;
            mine += (p < e0 && marked(h[u]) != 0xffffffffu) ? 1u : 0u;
        }
    }
    u32 incl = mine;  // inclusive scan over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = (u32)__shfl_up((int)incl, d);
        if (lane >= (u32)d) incl += t;
    }
    const u32 total = (u32)__shfl((int)incl, 63);
    if (total == 0) return;  // (wave-uniform)
    u32 base = 0;
    if (lane == 0) base = atomicAdd(&a.flags[1], total);
    base = (u32)__shfl((int)base, 0);
    u64 at = (u64)base + (incl - mine);  // this lane's pairs, in the order it meets them
    for (u64 p0 = b0; p0 < e0; p0 += 64u * PC_BOUND_U) {
        u64 h[PC_BOUND_U];
#pragma unroll
        for (u32 u = 0; u < PC_BOUND_U; ++u) {
            const u64 p = p0 + u * 64u + lane;
            h[u] = p < e0 ? a.values[p] : 0ull;
        }
#pragma unroll
        for (u32 u = 0; u < PC_BOUND_U; ++u) {
            const u64 p = p0 + u * 64u + lane;
            if (p >= e0) continue;
            const u32 bk = marked(h[u]);
            if (bk == 0xffffffffu) continue;
            if (at < a.spill_cap) {
                a.spill_a[at] = EMIT_REF ? h[u] : ((h[u] & rem_mask) | (sk << a.rem_bits));
                a.spill_p[at] = EMIT_REF ? (u32)sk : (u32)p;
                a.spill_b[at] = bk;
            }
            ++at;
        }
    }
}
// (sa = NULL: the key is sb[i] itself)
__global__ void k_spill_keys(const u64* __restrict__ sa, const u32* __restrict__ sb, u64 n, u32 rem_bits, u64* __restrict__ key, u32* __restrict__ idx) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = sa ? (((u64)sb[i] << rem_bits) | (sa[i] & ((1ull << rem_bits) - 1ull))) : (u64)sb[i];
    idx[i] = (u32)i;
}
__global__ void k_spill_gather(const u32* __restrict__ idx, u64 n, const u64* __restrict__ sa, const u32* __restrict__ sp, u64* __restrict__ key, u32* __restrict__ ref) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 j = idx[i];
    key[i] = sa[j];
    ref[i] = sp[j];
}
__global__ void __launch_bounds__(256) k_spill_group(const u64* __restrict__ key /* sorted */, const u32* __restrict__ idx, u64 n,
                                                     const u64* __restrict__ sa, const u32* __restrict__ sp, u32 rem_bits, u64 n_pos,
                                                     u64 list_base, u32* __restrict__ list2, u64* __restrict__ rec,
                                                     unsigned long long* __restrict__ totals) {
    __shared__ u32 tot3[3];
    if (threadIdx.x < 3) tot3[threadIdx.x] = 0;
    __syncthreads();
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    bool head = false, shared = false;
    if (i < n) {
        const u64 k = key[i];
        u64 lo = 0, hi = i;  // first entry of the run: the smallest j with key[j] == k (key[i] == k)
        while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (key[mid] < k) lo = mid + 1; else hi = mid; }
        const u64 first = lo;
        lo = i; hi = n;      // one past its last entry
        while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (key[mid] <= k) lo = mid + 1; else hi = mid; }
        const u64 len = lo - first;
        const u32 j = idx[i];
        list2[i] = (u32)(sa[j] >> rem_bits);
        head = i == first;
        shared = len >= 2;
        const u32 pos = sp[j];
        // (every pair of a marked bucket stores its record -- zero when nobody else holds its hash: k_bucket_group5 left the whole
        // bucket alone, and the records are not cleared beforehand)
        if (pos < n_pos) rec[pos] = shared ? ((1ull << 63) | (len << 40) | (list_base + first)) : 0ull;
    }
    const u32 c0 = (u32)__popcll(__ballot(head)), c1 = (u32)__popcll(__ballot(head && shared)), c2 = (u32)__popcll(__ballot(shared));
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&tot3[0], c0); atomicAdd(&tot3[1], c1); atomicAdd(&tot3[2], c2); }
    __syncthreads();
    if (threadIdx.x < 3 && tot3[threadIdx.x]) atomicAdd(&totals[threadIdx.x], (unsigned long long)tot3[threadIdx.x]);
}

// the side list of a SORTING handle, sorted by (hash, reference): every pair back to its bucket's place in the output, and the
// bucket's run statistics (what k_bucket_sort leaves for the buckets it sorts itself)
__global__ void __launch_bounds__(256) k_spill_place(const u64* __restrict__ key /* sorted */, const u32* __restrict__ ref, u64 n, u32 lsh, u64 mul_fine,
                                                     const u64* __restrict__ off, u64* __restrict__ out_k, u32* __restrict__ out_v, u32* __restrict__ counts) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 h = key[i];
    const u32 b = bucket_of(h, lsh, mul_fine);
    u64 lo = 0, hi = i;  // the first entry of bucket b (buckets are monotone in the hash)
    while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (bucket_of(key[mid], lsh, mul_fine) < b) lo = mid + 1; else hi = mid; }
    const u64 dst = off[b] + (i - lo);
    out_k[dst] = h;
    out_v[dst] = ref[i];
    const bool eq_prev = i > 0 && key[i - 1] == h, eq_next = i + 1 < n && key[i + 1] == h;
    if (!eq_prev) atomicAdd(&counts[(u64)b * 3], 1u);
    if (!eq_prev && eq_next) atomicAdd(&counts[(u64)b * 3 + 1], 1u);
    if (eq_prev || eq_next) atomicAdd(&counts[(u64)b * 3 + 2], 1u);
}

static_assert(BKT_SLOTS == (1u << BKT_SLOT_BITS), "slots per bucket");
u64 mul_for(u64 slots, u64 max_hash, unsigned bits) {
    const unsigned __int128 num = (unsigned __int128)slots << bits;
    const unsigned __int128 m = num / ((unsigned __int128)max_hash + 1);
    return (u64)std::min<unsigned __int128>(m, ~(u64)0);
}

}  // namespace

struct yh_psort {
    u64 H = 0, max_hash = 0;
    u64 NB = 0;
    u32 P1 = 0, P2 = 0, lsh = 0;
    u64 mul = 0, mul_fine = 0;
    u64 cap1 = 0;
    u64 fed = 0;
    u64* k1 = nullptr;  // [P1][cap1] first-level regions
    u32* v1 = nullptr;
    u64* k2 = nullptr;  // [NB][BKT_CAP] buckets
    u32* v2 = nullptr;
    u32* cnt = nullptr;  // [P1] + [NB] + flags[4]
    u64* off = nullptr;  // [NB + 1]
    u32* counts = nullptr;  // [NB][3] run statistics of every bucket (k_bucket_sort)
    bool check_order = false;
    const u32* ref_tab = nullptr;  // position mode (yh_psort_positions): the values are CSR positions
    const u64* ref_off = nullptr;
    u64 n_refs = 0;
    unsigned long long* first_bits = nullptr;  // position mode + ordering check: the sketches' first positions as a bit map
    u64* rec_clear = nullptr;      // position mode: the records the first level clears on its way through
    unsigned long long* totals = nullptr;  // [4] position mode: what the fused last pass counted
};

// Is this input one the distribution sort takes?  (A database of a few thousand hashes is one bucket; a key range narrower
// than the number of fine slots cannot be spread: rocPRIM sorts those.)
bool yh_psort_applicable(u64 H, u64 max_hash) {
    static const bool off = [] { const char* e = yh_tune_env("YH_NO_PSORT"); return e && e[0] == '1'; }();
    if (off || H == 0 || H > 0xfffffff0ull) return false;
    const u64 NB = (H + BKT_FILL - 1) / BKT_FILL;
    if (NB > (u64)PART_MAX_BINS * PART_MAX_BINS) return false;
    // every bucket needs a key range of at least BKT_SLOTS values, or equal "slots" would pile up whatever the keys are
    return max_hash / NB >= BKT_SLOTS || NB == 1;
}

void yh_psort_check_order(yh_psort* s, bool on) { s->check_order = on; }
void yh_psort_positions(yh_psort* s, const u32* d_ref_tab, const u64* d_offsets, u64 n_refs, u64* d_rec) {
    s->ref_tab = d_ref_tab;
    s->ref_off = d_offsets;
    s->n_refs = n_refs;
    s->rec_clear = d_rec;
}
int yh_ref_table_build(yh_db* db, const u64* d_offsets, u64 n_refs, u64 H, u32* d_tab) {
    const u64 n_tab = (H >> YH_REF_TAB_SH) + 2;
    if (n_refs == 0) return YH_OK;
    k_ref_table<<<(u32)((n_tab + 255) / 256), 256, 0, db->stream>>>(d_offsets, n_refs, n_tab, d_tab);
    YH_HIP(hipGetLastError());
    return YH_OK;
}
void yh_psort_chunks(const yh_psort* s, u64* n_chunks, const u64** d_chunk_off, const u32** d_chunk_counts) {
    *n_chunks = s->NB;
    *d_chunk_off = s->off;
    *d_chunk_counts = s->counts;
}
void yh_psort_destroy(yh_db* db, yh_psort* s) {
    if (!s) return;
    yh_tfree(db, s->k1); yh_tfree(db, s->v1); yh_tfree(db, s->k2); yh_tfree(db, s->v2);
    yh_tfree(db, s->cnt); yh_tfree(db, s->off); yh_tfree(db, s->counts); yh_tfree(db, s->totals); yh_tfree(db, s->first_bits);
    delete s;
}

int yh_psort_begin(yh_db* db, u64 H, u64 max_hash, yh_psort** out) {
    *out = nullptr;
    yh_psort* s = new yh_psort();
    s->H = H;
    s->max_hash = max_hash;
    const u64 nb = std::max<u64>((H + BKT_FILL - 1) / BKT_FILL, 1);
    u32 p2 = 1;
    while ((u64)p2 * p2 < nb) ++p2;
    s->P2 = p2;
    s->P1 = (u32)((nb + p2 - 1) / p2);
    s->NB = (u64)s->P1 * s->P2;
    unsigned bits = 1;
    while (bits < 64 && (max_hash >> bits) != 0) ++bits;
    s->lsh = 64 - bits;
    s->mul_fine = mul_for(s->NB * BKT_SLOTS, max_hash, bits);
    s->mul = s->mul_fine;  // (the distribution passes cut their bins from the same fine index)
    // a first-level region: its share of uniform keys + 3 % + a tile (skewed keys overflow it and are sorted by rocPRIM instead)
    s->cap1 = ((H / s->P1 + H / s->P1 / 32 + 2 * PART_TILE + 255) / 256) * 256;
    hipError_t e = hipSuccess;
    if (s->P1 > 1) {
        if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->k1, s->P1 * s->cap1 * sizeof(u64));
        if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->v1, s->P1 * s->cap1 * sizeof(u32));
    }
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->k2, s->NB * BKT_CAP * sizeof(u64));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->v2, s->NB * BKT_CAP * sizeof(u32));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->cnt, (s->P1 + s->NB + 4) * sizeof(u32));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->off, (s->NB + 1) * sizeof(u64));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->counts, s->NB * 3 * sizeof(u32));
    if (e == hipSuccess) e = hipMemsetAsync(s->cnt, 0, (s->P1 + s->NB + 4) * sizeof(u32), db->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->counts, 0, s->NB * 3 * sizeof(u32), db->stream);
    if (e != hipSuccess) {
        yh_set_error("distribution sort: allocation failed: %s", hipGetErrorString(e));
        yh_psort_destroy(db, s);
        return e == hipErrorOutOfMemory ? YH_ERR_OOM : YH_ERR_HIP;
    }
    *out = s;
    return YH_OK;
}

// first level for n more pairs (any order of calls; on the handle's stream)
int yh_psort_add(yh_db* db, yh_psort* s, const u64* d_keys, const u32* d_vals, u64 n, u64 pos_base) {
    if (n == 0) return YH_OK;
    if (!d_vals && !s->ref_tab) { yh_set_error("internal: positions as values without yh_psort_positions"); return YH_ERR_INVALID_ARG; }
    s->fed += n;
    u32* cnt1 = s->cnt;
    u32* cnt2 = s->cnt + s->P1;
    u32* flags = s->cnt + s->P1 + s->NB;
    PartArgs a{};
    a.in_k = d_keys;
    a.in_v = d_vals;
    a.val_base = pos_base;
    if (!d_vals && s->check_order && !s->first_bits) {  // the map of the sketches' first positions, once
        const size_t words = (size_t)(s->H >> 6) + 2;
        hipError_t e = yh_tmalloc(db, (void**)&s->first_bits, words * sizeof(unsigned long long));
        if (e != hipSuccess) { yh_set_error("distribution sort: allocation failed: %s", hipGetErrorString(e)); return YH_ERR_OOM; }
        YH_HIP(hipMemsetAsync(s->first_bits, 0, words * sizeof(unsigned long long), db->stream));
        k_first_bits<<<(u32)((s->n_refs + 255) / 256), 256, 0, db->stream>>>(s->ref_off, s->n_refs, s->first_bits);
        YH_HIP(hipGetLastError());
    }
    a.first_bits = d_vals ? nullptr : s->first_bits;
    a.rec_clear = d_vals ? nullptr : s->rec_clear;
    a.n_in = n;
    a.mul = s->mul;
    a.lsh = s->lsh;
    a.flags = flags;
    a.check_order = s->check_order ? 1u : 0u;
    const u32 tiles = (u32)((n + PART_TILE - 1) / PART_TILE);
    if (s->P1 > 1) {
        a.P2 = s->P2;
        a.nbins = s->P1;
        a.out_k = s->k1;
        a.out_v = s->v1;
        a.cap_out = s->cap1;
        a.out_cnt = cnt1;
        k_part<1><<<tiles, PART_THREADS, 0, db->stream>>>(a);
    } else {  // one first-level bin: straight into the buckets (bin = bucket % P2 = bucket)
        a.P2 = s->P2;
        a.nbins = s->P2;
        a.out_k = s->k2;
        a.out_v = s->v2;
        a.cap_out = BKT_CAP;
        a.out_cnt = cnt2;
        a.in_cnt = nullptr;  // (one input segment of n_in pairs)
        a.cap_in = n;
        a.tiles_per_seg = tiles;
        k_part<2><<<tiles, PART_THREADS, 0, db->stream>>>(a);
    }
    YH_HIP(hipGetLastError());
    return YH_OK;
}

// second level + the sort of every bucket; the sorted pairs land in d_keys_out / d_vals_out (H entries).  *took_it = false:
// the keys were not this sort's input (a capacity was exceeded) -- nothing usable was written, sort another way.
static int second_level(yh_db* db, yh_psort* s) {
    u32* cnt1 = s->cnt;
    u32* cnt2 = s->cnt + s->P1;
    u32* flags = s->cnt + s->P1 + s->NB;
    if (s->P1 > 1) {
        PartArgs a{};
        a.in_k = s->k1;
        a.in_v = s->v1;
        a.cap_in = s->cap1;
        a.in_cnt = cnt1;
        a.tiles_per_seg = (u32)((s->cap1 + PART_TILE - 1) / PART_TILE);
        a.mul = s->mul;
        a.lsh = s->lsh;
        a.P2 = s->P2;
        a.nbins = s->P2;
        a.out_k = s->k2;
        a.out_v = s->v2;
        a.cap_out = BKT_CAP;
        a.out_cnt = cnt2;
        a.flags = flags;
        const u64 grid = (u64)s->P1 * a.tiles_per_seg;
        if (grid >> 31) { yh_set_error("distribution sort: grid too large"); return YH_ERR_UNSUPPORTED; }
        k_part<2><<<(u32)grid, PART_THREADS, 0, db->stream>>>(a);
    }
    YH_HIP(hipGetLastError());
    return YH_OK;
}
static void say_verdict(const yh_psort* s, u32 flags, u64 total, bool took, const char* what) {
    static const bool trace = [] { const char* e = yh_tune_env("YH_TRACE_BUILD"); return e && e[0] == '1'; }();
    if (trace || !took) {
        // (a refusal is worth a line even without the trace switch: the caller falls back to a sort three times as slow)
        static const bool say = [] { const char* e = yh_tune_env("YH_TRACE_SORT"); return e && e[0] == '1'; }();
        if (trace || say)
            fprintf(stderr, "[yh sort] H %llu  P1 %u x P2 %u = %llu buckets  cap1 %llu  flags %u (1 region, 2 bucket, 4 slot)  %s %llu of %llu -> %s\n",
                    (u64)s->H, s->P1, s->P2, (u64)s->NB, (u64)s->cap1, flags, what, total, (u64)s->fed, took ? "taken" : "REFUSED");
    }
}

int yh_psort_finish(yh_db* db, yh_psort* s, u64* d_keys_out, u32* d_vals_out, bool* took_it, bool* unsorted) {
    *took_it = false;
    if (unsorted) *unsorted = false;
    u32* cnt2 = s->cnt + s->P1;
    u32* flags = s->cnt + s->P1 + s->NB;
    YH_TRY(second_level(db, s));
    k_bucket_offsets<<<1, 1024, 0, db->stream>>>(cnt2, s->NB, BKT_CAP, s->off, flags);
    BucketArgs b{};
    b.in_k = s->k2;
    b.in_v = s->v2;
    b.cnt = cnt2;
    b.off = s->off;
    b.cap_in = BKT_CAP;
    b.mul_fine = s->mul_fine;
    b.lsh = s->lsh;
    b.out_k = d_keys_out;
    b.out_v = d_vals_out;
    b.flags = flags;
    b.counts = s->counts;
    b.nb = s->NB;
    k_bucket_sort<<<(u32)s->NB, BKT_THREADS, 0, db->stream>>>(b);
    YH_HIP(hipGetLastError());
    u32 hflags[4] = {0, 0, 0, 0};
    u64 total = 0;
    YH_HIP(hipMemcpyAsync(hflags, flags, 3 * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipMemcpyAsync(&total, s->off + s->NB, sizeof(u64), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipStreamSynchronize(db->stream));
    *took_it = (hflags[0] & 7u) == 0 && total == s->fed;
    if (unsorted) *unsorted = (hflags[0] & 8u) != 0;
    say_verdict(s, hflags[0], total, *took_it, "sorted");
    return YH_OK;
}

// Position mode: second level + the FUSED last pass (k_bucket_group): every bucket is grouped by hash in LDS and leaves as the
// records of the pairwise pass (yh_db::d_fz_rec, H entries, cleared by the first level) instead of as sorted pairs.  totals[3] = {distinct
// hashes, hashes with >= 2 holders, their pairs}.  *d_list_out: the holders of the hashes with more than four of them
// (yh_db::d_fz_list) -- the sort's own bucket array, which is the caller's from here on (free it with yh_tfree).
// *took_it = false: a capacity was exceeded, nothing usable was written.  Synchronizes the handle's stream.
int yh_psort_finish_emit(yh_db* db, yh_psort* s, u64* d_rec, u64 n_refs, u64 totals[3], u32** d_list_out, bool* took_it, bool* unsorted) {
    *took_it = false;
    *d_list_out = nullptr;
    if (unsorted) *unsorted = false;
    if (!s->ref_tab) { yh_set_error("internal: the fused pass needs positions as values"); return YH_ERR_INVALID_ARG; }
    u32* cnt2 = s->cnt + s->P1;
    u32* flags = s->cnt + s->P1 + s->NB;
    hipError_t e = yh_tmalloc(db, (void**)&s->totals, TOT_LANES * 8 * sizeof(unsigned long long));
    if (e != hipSuccess) { yh_set_error("distribution sort: allocation failed: %s", hipGetErrorString(e)); return YH_ERR_OOM; }
    YH_HIP(hipMemsetAsync(s->totals, 0, TOT_LANES * 8 * sizeof(unsigned long long), db->stream));
    if (d_rec != s->rec_clear) { yh_set_error("internal: the records were not cleared by the first level"); return YH_ERR_INVALID_ARG; }
    YH_TRY(second_level(db, s));
    BucketArgs b{};
    b.in_k = s->k2;
    b.in_v = s->v2;
    b.cnt = cnt2;
    b.cap_in = BKT_CAP;
    b.mul_fine = s->mul_fine;
    b.lsh = s->lsh;
    b.flags = flags;
    b.nb = s->NB;
    b.n_pos = s->H;
    b.per_xcd = (u32)((s->NB + 7) / 8);
    b.rec = d_rec;
    b.list = s->v2;
    b.ref_tab = s->ref_tab;
    b.ref_off = s->ref_off;
    // (reference + 1 in 21 bits; YH_FZ_NO_INLINE=1 behind the tuning gate: every record in list form, as with >= 2^21 - 1 references)
    static const bool no_inline = [] { const char* e = yh_tune_env("YH_FZ_NO_INLINE"); return e && e[0] == '1'; }();
    b.inline_ok = (n_refs < (1u << 21) - 1 && !no_inline) ? 1u : 0u;
    b.totals = s->totals;
    k_bucket_group<<<8u * b.per_xcd, BKT_THREADS, 0, db->stream>>>(b);
    YH_HIP(hipGetLastError());
    u32 hflags[4] = {0, 0, 0, 0};
    unsigned long long ht[4] = {0, 0, 0, 0};
    static thread_local unsigned long long hrows[TOT_LANES * 8];
    YH_HIP(hipMemcpyAsync(hflags, flags, 3 * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipMemcpyAsync(hrows, s->totals, sizeof(hrows), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipStreamSynchronize(db->stream));
    for (u32 q = 0; q < TOT_LANES; ++q)
        for (u32 t = 0; t < 4; ++t) ht[t] += hrows[q * 8 + t];
    *took_it = (hflags[0] & 7u) == 0 && ht[3] == s->fed;
    if (unsorted) *unsorted = (hflags[0] & 8u) != 0;
    say_verdict(s, hflags[0], ht[3], *took_it, "fused: records of");
    if (*took_it) {
        totals[0] = ht[0]; totals[1] = ht[1]; totals[2] = ht[2];
        *d_list_out = s->v2;
        s->v2 = nullptr;
    }
    return YH_OK;
}


// =====================================================================================================================
// The distribution without a first level (kernels: k_piece_bounds, k_piece_part above) -- driver
// =====================================================================================================================
struct yh_pieces {
    u64 H = 0, max_hash = 0, n_refs = 0;
    u64 NB = 0, n_pad = 0;
    u32 P1 = 0, P2 = 0, lsh = 0, rem_bits = 0, S = 0, Gn = 0, SK = 0, inv_p2 = 0;
    u64 mul_fine = 0;
    // elements between two buckets of a2 / p2: the capacity + an optional pad (YH_PC_PAD; a stride of exactly 32 KB was
    // suspected of putting every bucket's first byte on one HBM channel -- pads of 0 ... 4 352 bytes measure the same)
    u64 stride_k = BKT_CAP, stride_v = BKT_CAP;
    u32* bnd = nullptr;   // [(P1 + 1)][n_pad]
    u64* a2 = nullptr;    // [NB][BKT_CAP] packed pairs
    u32* p2 = nullptr;    // [NB][BKT_CAP] positions; afterwards the list area of the handle (yh_db::d_fz_list)
    u32* cnt = nullptr;   // [NB] + flags[4]
    unsigned long long* totals = nullptr;
    u64* rec = nullptr;
    u64 scanned = 0;      // pairs the bounds pass has seen
    // the side list of overflowed buckets (k_piece_part / k_bucket_group5 -> yh_pc_finish_emit)
    u64* spill_a = nullptr; u32* spill_p = nullptr; u32* spill_b = nullptr;
    u64 spill_cap = 0;
    u64 n_spilled = 0, n_spilled_buckets = 0;  // (what yh_pc_finish_emit found)
    u32* list2 = nullptr;  // the holder lists of the spilled groups (handed to the handle)
};

static unsigned bitlen64(u64 x) { unsigned b = 0; while (x) { ++b; x >>= 1; } return b; }

// the geometry: NB buckets of ~BKT_FILL pairs = P1 regions x P2 buckets; false: not this path's input
static bool pc_geometry(u64 H, u64 max_hash, u64 n_refs, yh_pieces* g) {
    if (!yh_psort_applicable(H, max_hash) || n_refs == 0 || max_hash == ~0ull) return false;
    static const double p2f = [] { const char* e = yh_tune_env("YH_PC_P2F"); return e ? atof(e) : 2.0; }();
    static const u32 tile_elems = [] { const char* e = yh_tune_env("YH_PC_TILE_ELEMS"); return e ? (u32)atoi(e) : 3584u; }();
    const u64 nb = std::max<u64>((H + BKT_FILL - 1) / BKT_FILL, 1);
    u64 p2 = (u64)(p2f * std::sqrt((double)nb) + 0.5);
    p2 = std::min<u64>(std::max<u64>(p2, 1), PART_MAX_BINS);
    p2 = std::min<u64>(p2, nb);
    g->P2 = (u32)p2;
    g->P1 = (u32)((nb + p2 - 1) / p2);
    if (g->P1 > PART_MAX_BINS) return false;
    g->NB = (u64)g->P1 * g->P2;
    g->inv_p2 = p2 == 1 ? 0xffffffffu : (u32)(((1ull << 32) + p2 - 1) / p2);
    // floor(b * ceil(2^32 / P2) / 2^32) == b / P2 needs b * (P2 - 2^32 mod P2) < 2^32: true below 2^22 for P2 <= 1024; checked anyway
    if (p2 > 1 && (g->NB - 1) * (u64)((u64)g->inv_p2 * p2 - (1ull << 32)) >= (1ull << 32)) return false;
    if (p2 == 1) return false;  // (one bucket per region: a database this small goes the two-level way)
    unsigned bits = 1;
    while (bits < 64 && (max_hash >> bits) != 0) ++bits;
    g->lsh = 64 - bits;
    g->mul_fine = mul_for(g->NB * BKT_SLOTS, max_hash, bits);
    if (g->mul_fine == 0) return false;
    // the hashes of one bucket span at most ceil(BKT_SLOTS * 2^bits / mul) + 1 values: their low rem_bits bits tell them apart
    const unsigned __int128 span = (((unsigned __int128)BKT_SLOTS << bits) + g->mul_fine - 1) / g->mul_fine + 2;
    if (span >> 63) return false;
    g->rem_bits = bitlen64((u64)span);
    const unsigned ref_bits = std::max(1u, bitlen64(n_refs - 1));
    if (g->rem_bits + ref_bits > 64 || g->rem_bits >= 63) return false;
    // sketches per tile: their pieces of one region together ~tile_elems pairs
    const double avg_piece = (double)H / ((double)n_refs * g->P1);
    u64 S = (u64)(tile_elems / std::max(avg_piece, 1e-9));
    S = std::min<u64>(std::max<u64>(S, 1), PC_MAX_S);
    S = std::min<u64>(S, n_refs);
    g->S = (u32)S;
    g->Gn = (u32)((n_refs + S - 1) / S);
    // sketches per workgroup of the bounds pass: their [SK][P1 + 1] block in LDS (<= 48 KB)
    static const u32 sk_env = [] { const char* e = yh_tune_env("YH_PC_SK"); return e ? (u32)std::max(1, atoi(e)) : 0u; }();
    g->SK = (u32)std::min<u64>(sk_env ? sk_env : PC_BOUND_THREADS / 64, std::max<u64>(sk_env ? 1 : 4, 12288 / (g->P1 + 1)));  // (a sketch per wave)
    g->n_pad = (n_refs + 63) & ~(u64)63;
    const u64 tiles = 8ull * ((g->P1 + 7) / 8) * g->Gn;
    if (tiles >> 31) return false;
    static const u32 pad = [] { const char* e = yh_tune_env("YH_PC_PAD"); return e ? (u32)atoi(e) : 0u; }();  // bytes (0, 128 ... 4 352 measured the same: profiles/r05/sweep_pad.txt)
    g->stride_k = BKT_CAP + pad / 8;
    g->stride_v = BKT_CAP + pad / 4;
    g->H = H; g->max_hash = max_hash; g->n_refs = n_refs;
    return true;
}
bool yh_pc_applicable(u64 H, u64 max_hash, u64 n_refs) {
    static const bool off = [] { const char* e = yh_tune_env("YH_NO_PIECES"); return e && e[0] == '1'; }();
    yh_pieces g;
    return !off && pc_geometry(H, max_hash, n_refs, &g);
}
void yh_pc_destroy(yh_db* db, yh_pieces* s) {
    if (!s) return;
    yh_tfree(db, s->bnd); yh_tfree(db, s->a2); yh_tfree(db, s->p2); yh_tfree(db, s->cnt); yh_tfree(db, s->totals);
    yh_tfree(db, s->spill_a); yh_tfree(db, s->spill_p); yh_tfree(db, s->spill_b); yh_tfree(db, s->list2);
    delete s;
}
int yh_pc_begin(yh_db* db, u64 H, u64 max_hash, u64 n_refs, u64* d_rec, yh_pieces** out) {
    *out = nullptr;
    yh_pieces* s = new yh_pieces();
    if (!pc_geometry(H, max_hash, n_refs, s)) { delete s; yh_set_error("internal: not the input of the piece distribution"); return YH_ERR_INVALID_ARG; }
    s->rec = d_rec;
    hipError_t e = yh_tmalloc(db, (void**)&s->bnd, (u64)(s->P1 + 1) * s->n_pad * sizeof(u32));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->a2, s->NB * s->stride_k * sizeof(u64));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->p2, s->NB * s->stride_v * sizeof(u32));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->cnt, (s->NB + 4 + s->NB / 32 + 2) * sizeof(u32));  // counts | flags[4] | the overflowed buckets as bits
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->totals, TOT_LANES * 8 * sizeof(unsigned long long));
    // (the side list of overflowed buckets gets its room when a bucket HAS overflowed: a sixteenth of the pairs, at least a million --
    // more than that is not a database with a few hot k-mers but keys this distribution is not made for: refused)
    static const bool no_spill = [] { const char* e_ = yh_tune_env("YH_NO_SPILL"); return e_ && e_[0] == '1'; }();
    s->spill_cap = no_spill ? 0 : std::max<u64>(H / 16, (u64)1 << 20);
    if (e == hipSuccess) e = hipMemsetAsync(s->cnt, 0, (s->NB + 4 + s->NB / 32 + 2) * sizeof(u32), db->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->totals, 0, TOT_LANES * 8 * sizeof(unsigned long long), db->stream);
    if (e != hipSuccess) {
        yh_set_error("piece distribution: allocation failed: %s", hipGetErrorString(e));
        yh_pc_destroy(db, s);
        return e == hipErrorOutOfMemory ? YH_ERR_OOM : YH_ERR_HIP;
    }
    *out = s;
    return YH_OK;
}
// (the chained table of the two-level path, k_bucket_group, takes packed pairs too: YH_GROUP_CHAINS=1 behind the tuning gate; it
// stores the records of shared hashes only, so with it the bounds pass clears them as it did until round 5)
static bool pc_group_chains() {
    static const bool chains = [] { const char* e = yh_tune_env("YH_GROUP_CHAINS"); return e && e[0] == '1'; }();
    return chains;
}
static PieceArgs pc_args(const yh_pieces* s, const u64* d_values, const u64* d_offsets) {
    PieceArgs a{};
    a.values = d_values; a.off = d_offsets; a.n_refs = s->n_refs; a.bnd = s->bnd; a.n_pad = s->n_pad;
    a.P1 = s->P1; a.P2 = s->P2; a.S = s->S; a.Gn = s->Gn; a.SK = s->SK;
    a.mul = s->mul_fine; a.lsh = s->lsh; a.rem_bits = s->rem_bits; a.inv_p2 = s->inv_p2;
    a.rec_clear = pc_group_chains() ? s->rec : nullptr;  // (k_bucket_group5 writes every position's record itself)
    a.out_a = s->a2; a.out_p = s->p2; a.out_cnt = s->cnt; a.flags = s->cnt + s->NB;
    a.stride_k = s->stride_k; a.stride_v = s->stride_v;
    a.spill_a = s->spill_a; a.spill_p = s->spill_p; a.spill_b = s->spill_b; a.spill_cap = s->spill_cap;
    a.over_bits = s->cnt + s->NB + 4;
    return a;
}
// the bounds pass over the sketches [r0, r1) (any number of calls, any order: the chunks of an upload as they arrive)
int yh_pc_scan(yh_db* db, yh_pieces* s, const u64* d_values, const u64* d_offsets, u64 r0, u64 r1, u64 n_pairs, bool check_order) {
    if (r1 <= r0) return YH_OK;
    PieceArgs a = pc_args(s, d_values, d_offsets);
    a.check_order = check_order ? 1u : 0u;
    const u64 wgs = (r1 - r0 + s->SK - 1) / s->SK;
    if (wgs >> 31) { yh_set_error("piece distribution: grid too large"); return YH_ERR_UNSUPPORTED; }
    const size_t lds = (size_t)s->SK * (s->P1 + 1) * sizeof(u32);
    k_piece_bounds<<<(u32)wgs, PC_BOUND_THREADS, lds, db->stream>>>(a, r0, r1);
    YH_HIP(hipGetLastError());
    s->scanned += n_pairs;
    return YH_OK;
}
// some bucket overflowed: room for the side list, then every pair of the marked buckets from the sketches again; *n_spill = their
// number (more than the room: the caller refuses the attempt).  Synchronizes the stream.
static int pc_collect_spill(yh_db* db, yh_pieces* s, const u64* d_values, const u64* d_offsets, bool emit_ref, u64* n_spill) {
    hipError_t e = yh_tmalloc(db, (void**)&s->spill_a, s->spill_cap * sizeof(u64));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->spill_p, s->spill_cap * sizeof(u32));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&s->spill_b, s->spill_cap * sizeof(u32));
    if (e != hipSuccess) { yh_set_error("piece distribution: no room for the side list: %s", hipGetErrorString(e)); return YH_ERR_OOM; }
    PieceArgs a = pc_args(s, d_values, d_offsets);
    const u32 grid = (u32)((s->n_refs + 3) / 4);
    if (emit_ref) k_spill_collect<true><<<grid, 256, 0, db->stream>>>(a);
    else k_spill_collect<false><<<grid, 256, 0, db->stream>>>(a);
    YH_HIP(hipGetLastError());
    u32 got = 0;
    YH_HIP(hipMemcpyAsync(&got, s->cnt + s->NB + 1, sizeof(u32), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipStreamSynchronize(db->stream));
    *n_spill = got;
    return YH_OK;
}

// distribution + the fused last pass (k_bucket_group on packed pairs): as yh_psort_finish_emit
int yh_pc_finish_emit(yh_db* db, yh_pieces* s, const u64* d_values, const u64* d_offsets, u64 totals[3], u32** d_list_out, bool* took_it,
                      bool* unsorted, yh_pc_spill* spill) {
    *took_it = false;
    *d_list_out = nullptr;
    if (unsorted) *unsorted = false;
    PieceArgs a = pc_args(s, d_values, d_offsets);
    k_piece_part<false><<<8u * ((s->P1 + 7) / 8) * s->Gn, PART_THREADS, 0, db->stream>>>(a);
    YH_HIP(hipGetLastError());
    u32* flags = s->cnt + s->NB;
    BucketArgs b{};
    b.in_k = s->a2;
    b.in_v = s->p2;
    b.cnt = s->cnt;
    b.cap_in = s->stride_k;
    b.stride_v = s->stride_v;
    b.mul_fine = s->mul_fine;
    b.lsh = s->lsh;
    b.flags = flags;
    b.nb = s->NB;
    b.n_pos = s->H;
    b.per_xcd = (u32)((s->NB + 7) / 8);
    b.rec = s->rec;
    b.list = s->p2;
    b.rem_bits = s->rem_bits;
    b.over_bits = s->spill_cap ? s->cnt + s->NB + 4 : nullptr;
    static const bool no_inline = [] { const char* e = yh_tune_env("YH_FZ_NO_INLINE"); return e && e[0] == '1'; }();
    b.inline_ok = (s->n_refs < (1u << 21) - 1 && !no_inline) ? 1u : 0u;
    b.totals = s->totals;
    const bool chains = pc_group_chains();
    b.write_all = chains ? 0u : 1u;
    if (!chains) k_bucket_group5<<<8u * b.per_xcd, BKT_THREADS, 0, db->stream>>>(b);
    else k_bucket_group<<<8u * b.per_xcd, BKT_THREADS, 0, db->stream>>>(b);
    YH_HIP(hipGetLastError());
    unsigned long long ht[4] = {0, 0, 0, 0};
    static thread_local unsigned long long hrows_pageable[TOT_LANES * 8 + 2];
    constexpr size_t ROWS_BYTES = TOT_LANES * 8 * sizeof(unsigned long long);
    YhPin pin(ROWS_BYTES + 16);  // (both read-backs queued into page-locked memory, one wait)
    unsigned long long* hrows = pin.p ? static_cast<unsigned long long*>(pin.p) : hrows_pageable;
    u32* hflags = reinterpret_cast<u32*>(hrows + TOT_LANES * 8);
    hflags[0] = hflags[1] = hflags[2] = hflags[3] = 0;
    YH_HIP(hipMemcpyAsync(hflags, flags, 3 * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipMemcpyAsync(hrows, s->totals, ROWS_BYTES, hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipStreamSynchronize(db->stream));
    for (u32 q = 0; q < TOT_LANES; ++q)
        for (u32 t = 0; t < 4; ++t) ht[t] += hrows[q * 8 + t];
    // the side list: hflags[1] pairs of hflags[2] overflowed buckets -- sorted by (bucket, hash remainder) and grouped on their own
    u64 n_spill = 0;
    bool spill_ok = true;
    if (hflags[2] && (hflags[0] & 15u) == 0) YH_TRY(pc_collect_spill(db, s, d_values, d_offsets, false, &n_spill));
    if (n_spill && (hflags[0] & 15u) == 0) {
        unsigned nb_bits = 1;
        while (nb_bits < 32 && (s->NB >> nb_bits) != 0) ++nb_bits;
        if (n_spill > s->spill_cap || nb_bits + s->rem_bits > 64) {
            spill_ok = false;
        } else {
            u64 *k_in = nullptr, *k_out = nullptr;
            u32 *i_in = nullptr, *i_out = nullptr;
            hipError_t e2 = yh_tmalloc(db, (void**)&k_in, n_spill * sizeof(u64));
            if (e2 == hipSuccess) e2 = yh_tmalloc(db, (void**)&k_out, n_spill * sizeof(u64));
            if (e2 == hipSuccess) e2 = yh_tmalloc(db, (void**)&i_in, n_spill * sizeof(u32));
            if (e2 == hipSuccess) e2 = yh_tmalloc(db, (void**)&i_out, n_spill * sizeof(u32));
            if (e2 == hipSuccess) e2 = yh_tmalloc(db, (void**)&s->list2, n_spill * sizeof(u32));
            int rc2 = e2 == hipSuccess ? YH_OK : YH_ERR_OOM;
            if (rc2 == YH_OK) {
                k_spill_keys<<<(u32)((n_spill + 255) / 256), 256, 0, db->stream>>>(s->spill_a, s->spill_b, n_spill, s->rem_bits, k_in, i_in);
                rc2 = yh_radix_sort_pairs_u64_u32(db, k_in, k_out, i_in, i_out, n_spill, nb_bits + s->rem_bits);
            }
            if (rc2 == YH_OK) {
                const u64 list_base = s->NB * s->stride_v;  // (list records below it name the buckets' own list areas: yh_db::fz_list_split)
                k_spill_group<<<(u32)((n_spill + 255) / 256), 256, 0, db->stream>>>(k_out, i_out, n_spill, s->spill_a, s->spill_p, s->rem_bits, s->H,
                                                                                   list_base, s->list2, s->rec, s->totals);
                if (hipGetLastError() != hipSuccess) rc2 = YH_ERR_HIP;
                unsigned long long add3[3] = {0, 0, 0};
                if (rc2 == YH_OK && hipMemcpyAsync(add3, s->totals, sizeof(add3), hipMemcpyDeviceToHost, db->stream) != hipSuccess) rc2 = YH_ERR_HIP;
                if (rc2 == YH_OK && hipStreamSynchronize(db->stream) != hipSuccess) rc2 = YH_ERR_HIP;
                // (row 0 of the counters held the grouping pass's share of buckets 0, 256, ...: already summed into ht[] above)
                for (u32 t = 0; t < 3; ++t) ht[t] += add3[t] - hrows[t];
                ht[3] += n_spill;
            }
            yh_tfree(db, k_in); yh_tfree(db, k_out); yh_tfree(db, i_in); yh_tfree(db, i_out);
            if (rc2 != YH_OK) { yh_set_error("the side list of the overflowed buckets could not be grouped"); return rc2; }
            s->n_spilled = n_spill;
            s->n_spilled_buckets = hflags[2];
        }
    }
    *took_it = spill_ok && (hflags[0] & 7u) == 0 && ht[3] == s->H && s->scanned == s->H && !(hflags[0] & 8u);
    if (unsorted) *unsorted = (hflags[0] & 8u) != 0;
    static const bool trace = [] { const char* e = yh_tune_env("YH_TRACE_BUILD"); const char* f = yh_tune_env("YH_TRACE_SORT"); return (e && e[0] == '1') || (f && f[0] == '1'); }();
    if (trace)
        fprintf(stderr, "[yh pieces] H %llu  P1 %u x P2 %u = %llu buckets  S %u x Gn %u  SK %u  rem_bits %u  flags %u  records of %llu of %llu -> %s\n",
                (u64)s->H, s->P1, s->P2, (u64)s->NB, s->S, s->Gn, s->SK, s->rem_bits, hflags[0], (u64)ht[3], (u64)s->H, *took_it ? "taken" : "REFUSED");
    if (trace && n_spill) fprintf(stderr, "[yh pieces] side list: %llu pairs of %u overflowed buckets (room for %llu)\n", (u64)n_spill, hflags[2], (u64)s->spill_cap);
    if (*took_it) {
        totals[0] = ht[0]; totals[1] = ht[1]; totals[2] = ht[2];
        *d_list_out = s->p2;
        s->p2 = nullptr;
        if (spill) {
            spill->d_list2 = s->list2;
            s->list2 = nullptr;
            spill->list_split = s->NB * s->stride_v;
            spill->n_pairs = s->n_spilled;
            spill->n_buckets = s->n_spilled_buckets;
        }
    }
    return YH_OK;
}


// ---- the same distribution for every OTHER handle: sorted (hash, reference) pairs + the buckets as chunks (as yh_psort_finish) ----
// Bounds pass (ordering check) -> pieces read in place, leaving as (whole hash, reference) pairs (no array of reference ids is
// made and read: k_fill_ref_ids' 4 H bytes each way are gone with the first level's 12 + 12) -> k_bucket_sort.  Buckets that
// overflow, or hold a slot too crowded for the sort's ranking step, go to the side list: sorted by rocPRIM on their own and put
// back at their places (k_spill_place) -- the rest of the database never notices.  *chunks_out: a yh_psort that owns the bucket
// offsets / statistics for yh_psort_chunks (the caller's to yh_psort_destroy).
int yh_pc_sort(yh_db* db, const u64* d_values, const u64* d_offsets, u64 n_refs, u64 H, u64 max_hash, bool check_order,
               u64* d_keys_out, u32* d_vals_out, yh_psort** chunks_out, bool* took_it, bool* unsorted, u64* n_spilled_pairs, u64* n_spilled_buckets) {
    *took_it = false;
    *chunks_out = nullptr;
    if (unsorted) *unsorted = false;
    yh_pieces* s = nullptr;
    YH_TRY(yh_pc_begin(db, H, max_hash, n_refs, nullptr, &s));
    struct Guard { yh_db* db; yh_pieces* s; ~Guard() { yh_pc_destroy(db, s); } } guard{db, s};
    s->stride_k = s->stride_v = BKT_CAP;  // (a tuning pad between the buckets' arrays is the grouping pass's business: k_bucket_sort reads both at one stride)
    u64* off = nullptr;
    u32* counts = nullptr;
    hipError_t e = yh_tmalloc(db, (void**)&off, (s->NB + 1) * sizeof(u64));
    if (e == hipSuccess) e = yh_tmalloc(db, (void**)&counts, s->NB * 3 * sizeof(u32));
    if (e == hipSuccess) e = hipMemsetAsync(counts, 0, s->NB * 3 * sizeof(u32), db->stream);
    if (e != hipSuccess) { yh_tfree(db, off); yh_tfree(db, counts); yh_set_error("piece distribution: allocation failed: %s", hipGetErrorString(e)); return YH_ERR_OOM; }
    auto fail = [&](int rc) { yh_tfree(db, off); yh_tfree(db, counts); return rc; };
    int rc = yh_pc_scan(db, s, d_values, d_offsets, 0, n_refs, H, check_order);
    if (rc != YH_OK) return fail(rc);
    PieceArgs a = pc_args(s, d_values, d_offsets);
    a.emit_ref = 1u;
    a.rec_clear = nullptr;
    k_piece_part<true><<<8u * ((s->P1 + 7) / 8) * s->Gn, PART_THREADS, 0, db->stream>>>(a);
    u32* flags = s->cnt + s->NB;
    k_bucket_offsets<<<1, 1024, 0, db->stream>>>(s->cnt, s->NB, s->spill_cap ? 0u : BKT_CAP, off, flags);
    BucketArgs b{};
    b.in_k = s->a2; b.in_v = s->p2; b.cnt = s->cnt; b.off = off; b.cap_in = BKT_CAP;
    b.mul_fine = s->mul_fine; b.lsh = s->lsh;
    b.out_k = d_keys_out; b.out_v = d_vals_out;
    b.flags = flags; b.counts = counts; b.nb = s->NB;
    b.over_bits = s->spill_cap ? s->cnt + s->NB + 4 : nullptr;
    k_bucket_sort<<<(u32)s->NB, BKT_THREADS, 0, db->stream>>>(b);
    if (hipGetLastError() != hipSuccess) { yh_set_error("piece distribution: launch failed"); return fail(YH_ERR_HIP); }
    u32 hflags[4] = {0, 0, 0, 0};
    u64 total = 0;
    if (hipMemcpyAsync(hflags, flags, 3 * sizeof(u32), hipMemcpyDeviceToHost, db->stream) != hipSuccess ||
        hipMemcpyAsync(&total, off + s->NB, sizeof(u64), hipMemcpyDeviceToHost, db->stream) != hipSuccess ||
        hipStreamSynchronize(db->stream) != hipSuccess) { yh_set_error("piece distribution: readback failed"); return fail(YH_ERR_HIP); }
    u64 n_spill = 0;
    bool ok = (hflags[0] & 15u) == 0 && total == H && s->scanned == H;
    if (ok && hflags[2]) {
        rc = pc_collect_spill(db, s, d_values, d_offsets, true, &n_spill);
        if (rc != YH_OK) return fail(rc);
    }
    if (ok && n_spill) {
        if (n_spill > s->spill_cap) {
            ok = false;
        } else {  // by reference, then (stably) by hash: (hash, reference) order; then every pair to its bucket's place
            unsigned hbits = 1, rbits = 1;
            while (hbits < 64 && (max_hash >> hbits) != 0) ++hbits;
            while (rbits < 32 && ((n_refs - 1) >> rbits) != 0) ++rbits;
            u64 *k1 = nullptr, *k2 = nullptr;
            u32 *r1 = nullptr, *r2 = nullptr;
            e = yh_tmalloc(db, (void**)&k1, n_spill * sizeof(u64));
            if (e == hipSuccess) e = yh_tmalloc(db, (void**)&k2, n_spill * sizeof(u64));
            if (e == hipSuccess) e = yh_tmalloc(db, (void**)&r1, n_spill * sizeof(u32));
            if (e == hipSuccess) e = yh_tmalloc(db, (void**)&r2, n_spill * sizeof(u32));
            rc = e == hipSuccess ? YH_OK : YH_ERR_OOM;
            // (radix sort by the REFERENCE as key needs it as the key: sort (ref -> key u64) pairs with the hash as "value" is not what
            // the helper takes; sort by the composite in two passes instead: pass 1 orders by reference -- key = reference, value =
            // index -- pass 2 orders stably by hash)
            u64* kr = k1;  // reference as a 64-bit key
            if (rc == YH_OK) {
                k_spill_keys<<<(u32)((n_spill + 255) / 256), 256, 0, db->stream>>>(nullptr, s->spill_p, n_spill, 0u, kr, r1);  // key = reference, idx = i
                rc = yh_radix_sort_pairs_u64_u32(db, kr, k2, r1, r2, n_spill, rbits);  // r2 = indices in reference order
            }
            if (rc == YH_OK) {
                k_spill_gather<<<(u32)((n_spill + 255) / 256), 256, 0, db->stream>>>(r2, n_spill, s->spill_a, s->spill_p, k1, r1);  // k1 = hash, r1 = reference, in that order
                rc = yh_radix_sort_pairs_u64_u32(db, k1, k2, r1, r2, n_spill, hbits);  // stable: equal hashes keep ascending references
            }
            if (rc == YH_OK) {
                k_spill_place<<<(u32)((n_spill + 255) / 256), 256, 0, db->stream>>>(k2, r2, n_spill, s->lsh, s->mul_fine, off, d_keys_out, d_vals_out, counts);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(db->stream) != hipSuccess) rc = YH_ERR_HIP;
            }
            yh_tfree(db, k1); yh_tfree(db, k2); yh_tfree(db, r1); yh_tfree(db, r2);
            if (rc != YH_OK) { yh_set_error("the side list of the overflowed buckets could not be sorted"); return fail(rc); }
        }
    }
    if (unsorted) *unsorted = (hflags[0] & 8u) != 0;
    static const bool trace = [] { const char* e_ = yh_tune_env("YH_TRACE_BUILD"); const char* f = yh_tune_env("YH_TRACE_SORT"); return (e_ && e_[0] == '1') || (f && f[0] == '1'); }();
    if (trace)
        fprintf(stderr, "[yh pieces] H %llu  P1 %u x P2 %u = %llu buckets  S %u x Gn %u  flags %u  sorted %llu of %llu  side list %llu pairs of %u buckets -> %s\n",
                (u64)H, s->P1, s->P2, (u64)s->NB, s->S, s->Gn, hflags[0], (u64)total, (u64)H, (u64)n_spill, hflags[2], ok ? "taken" : "REFUSED");
    if (!ok) return fail(YH_OK);
    yh_psort* ps = new yh_psort();  // (only what yh_psort_chunks / yh_psort_destroy look at)
    ps->H = H; ps->max_hash = max_hash; ps->NB = s->NB; ps->P1 = s->P1; ps->P2 = s->P2;
    ps->off = off;
    ps->counts = counts;
    *chunks_out = ps;
    *took_it = true;
    if (n_spilled_pairs) *n_spilled_pairs = n_spill;
    if (n_spilled_buckets) *n_spilled_buckets = hflags[2];
    return YH_OK;
}
