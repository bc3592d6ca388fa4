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

#include <algorithm>

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
constexpr u32 BKT_FILL = BKT_CAP / 8 * 5;      // ... and holds on average (2 560 of 4 096; uniform keys: sd ~51; clustered references ~3x that)
constexpr u32 BKT_THREADS = YH_BKT_THREADS;
constexpr u32 BKT_ITEMS = BKT_CAP / BKT_THREADS;
constexpr u32 BKT_SLOTS = BKT_CAP;   // fine slots of the counting sort inside a bucket (= 1 << BKT_SLOT_BITS)
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
__global__ void __launch_bounds__(1024) k_bucket_offsets(const u32* __restrict__ cnt, u64 nb, u32 cap, u64* __restrict__ off,
                                                         u32* __restrict__ flags) {
    __shared__ u32 lds[1024];
    __shared__ u32 wtot[17];
    u64 carry = 0;
    for (u64 base = 0; base < nb; base += 1024) {
        const u64 b = base + threadIdx.x;
        u32 c = b < nb ? cnt[b] : 0u;
        if (c > cap) { atomicOr(flags, 2u); c = cap; }
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
    unsigned long long* totals;  // [4] {distinct hashes, shared hashes, their pairs, pairs seen}
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
    if (crowded) atomicOr(a.flags, 4u);
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
    if (a.cnt[b] > BKT_CAP && tid == 0) atomicOr(a.flags, 2u);
    const u32 n = min(a.cnt[b], BKT_CAP);
    if (n == 0) return;
    u64 key[BKT_ITEMS];
    u32 val[BKT_ITEMS], rf[BKT_ITEMS], slot[BKT_ITEMS];
#pragma unroll
    for (u32 k = 0; k < BKT_ITEMS; ++k) {
        const u32 e = k * BKT_THREADS + tid;
        slot[k] = NONE;
        if (e < n) { key[k] = a.in_k[b * a.cap_in + e]; val[k] = a.in_v[b * a.cap_in + e]; }
    }
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
            rf[k] = ref_of(val[k], a.ref_tab, a.ref_off);
            eref[e] = rf[k];
            const unsigned long long k1 = key[k] + 1ull;  // (the fused path is taken only where the largest hash is below 2^64 - 1)
            u32 s = fine_of(key[k], a.lsh, a.mul_fine) & (BKT_CAP - 1u);
            for (u32 probe = 0; probe < BKT_CAP; ++probe) {
                const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tkey[s]), 0ull, k1);
                if (old == 0ull) { won = true; break; }
                if (old == k1) break;
                s = (s + 1u) & (BKT_CAP - 1u);
            }
            slot[k] = s;
            enext[e] = (u16)atomicExch(&thead[s], e);  // (NONE -> 0xffff)
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
            for (u32 j = thead[slot[k]]; j != 0xffffu && j != NONE; j = enext[j]) {
                if (j == e) continue;
                if (cnt < 3) others[cnt] = eref[j];
                ++cnt;
                rank += j < e ? 1u : 0u;
            }
            const u32 len = cnt + 1u;
            first = enext[e] == 0xffffu;  // (the pair that claimed the chain: one per group)
            shared = len >= 2u;
            if (shared) {
                if (len <= 4u && a.inline_ok) {
                    const u64 r = (u64)(others[0] + 1u) | (cnt > 1 ? (u64)(others[1] + 1u) << 21 : 0ull) | (cnt > 2 ? (u64)(others[2] + 1u) << 42 : 0ull);
                    if (val[k] < a.n_pos) a.rec[val[k]] = r;
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
    if (tid < 3 && tot3[tid]) atomicAdd(&a.totals[tid], (unsigned long long)tot3[tid]);
    if (tid == 3) atomicAdd(&a.totals[3], (unsigned long long)n);
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
            const u64 start = b * BKT_CAP + thead[slot[k]];
            a.list[start + grank[k]] = rf[k];
            if (val[k] < a.n_pos) a.rec[val[k]] = (1ull << 63) | ((u64)glen[k] << 40) | start;
        }
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
    hipError_t e = yh_tmalloc(db, (void**)&s->totals, 4 * sizeof(unsigned long long));
    if (e != hipSuccess) { yh_set_error("distribution sort: allocation failed: %s", hipGetErrorString(e)); return YH_ERR_OOM; }
    YH_HIP(hipMemsetAsync(s->totals, 0, 4 * sizeof(unsigned long long), db->stream));
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
    YH_HIP(hipMemcpyAsync(hflags, flags, 3 * sizeof(u32), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipMemcpyAsync(ht, s->totals, sizeof(ht), hipMemcpyDeviceToHost, db->stream));
    YH_HIP(hipStreamSynchronize(db->stream));
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
