// yh_pairwise.hip -- `yacht train`: pairwise intersection counts from the posting lists (src/cpp/main.cpp:249-308)
//
// |R_i ∩ R_j| for every pair with a common hash = for every shared hash, one count for every ordered pair of its
// holders.  The counts of ONE reference a (a row of the reference's intersection matrix) only come from a's own
// postings, so the row is summed where atomics are cheap -- in the LDS of one workgroup -- and only its survivors
// (main.cpp:297-303: !(count / |R_i| < C)) ever reach HBM.  No dense count matrix: 16 bytes per posting of scratch.
//
//   k_pair_transpose   the hash-major postings (pr / pg / po) into a reference-major array of 16-byte records
//                      "the OTHER holders of this posting's hash" -- three inline as COMPACT ids, or where the list is
//                      in pr[]; the slot inside the reference's run is the posting's rank among the reference's shared
//                      hashes (k_idx_emit's counting atomic returns it: yh_db::d_prank), or a cursor atomic where the
//                      handle has no ranks
//   k_pair_rows        one workgroup per reference (and block of columns): its records, four loads in flight per lane,
//                      one LDS add per other holder; then the row's survivors, in column order, into the segment's own
//                      four slots of the output, or -- a longer segment -- into room claimed behind them with one atomic
// configs[3] (27 M postings, 10 000 rows): 0.59 + 0.28 ms, where the dense count matrix with one global atomic per
// (posting, other holder) took 2.66 + 0.53 ms.
// `yacht train`'s own handle (YH_DB_PAIRWISE_ONLY, uniform keys: yh_db::fz) skips the transposition altogether: the last
// pass of the sort has written an 8-byte record per CSR position (yh_sort.hip), a reference's records are the extent of
// its sketch, and k_pair_rows<.., true> reads them as they lie -- 0.28 ms for the whole pairwise call's device side.
// The host puts the segments in row order and applies the exact threshold (no decision depends on device floating point).
#include "yh_common.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

constexpr int WAVE = 64;
// holders of a hash up to which a posting's record names the others inline (three others)
constexpr u32 PAIR_INLINE = 4;
// columns (u32 counts) of a row block in LDS: with the attribute below 36 864 (144 KiB), else what fits 64 KiB.  Both leave
// room for k_pair_rows' STATIC LDS -- the waves' list queues, 8 KiB at 1 024 lanes, + the wave totals (15 360
// columns + the queues was 65.6 KB of a 64 KB limit): 144 + 8.1 KiB of gfx950's 160, 54 + 8.1 KiB of 64.
constexpr u32 PAIR_COLS_BIG = 36864;
constexpr u32 PAIR_COLS_SMALL = 13824;
constexpr u32 PAIR_STATIC_LDS = 16u * 64u * 8u + 256u;  // (lqueue[WAVES][64] at 16 waves + wtot + s_base, rounded up)
static_assert(PAIR_COLS_BIG * 4u + PAIR_STATIC_LDS <= 160u * 1024u, "a row block + the kernel's static LDS fit gfx950's 160 KiB");
static_assert(PAIR_COLS_SMALL * 4u + PAIR_STATIC_LDS <= 64u * 1024u, "the fallback row block + the static LDS fit 64 KiB");

__global__ void __launch_bounds__(256) k_pair_transpose(u64 n_post, const u32* __restrict__ pr, const u32* __restrict__ pg,
                                                        const u64* __restrict__ po, const u32* __restrict__ rowptr,
                                                        const u32* __restrict__ prank, u32* __restrict__ cursor,
                                                        const u32* __restrict__ cid, uint4* __restrict__ rrec) {
    for (u64 k = blockIdx.x * (u64)blockDim.x + threadIdx.x; k < n_post; k += (u64)gridDim.x * blockDim.x) {
        const u32 a = pr[k], g = pg[k];
        const u64 q0 = po[g], q1 = po[g + 1];
        const u32 slot = prank ? prank[k] : atomicAdd(&cursor[a], 1u);  // order inside a reference is irrelevant (sums)
        uint4 rec;
        if (q1 - q0 <= PAIR_INLINE) {
            u32 o[3] = {0u, 0u, 0u};
            u32 n = 0;
            for (u64 q = q0; q < q1; ++q)
                if (q != k) o[n++] = pr[q];
            // (COMPACT ids: the row pass adds them to its LDS row without another look-up; o[] beyond n is reference 0)
            rec = make_uint4(cid[o[0]], cid[o[1]], cid[o[2]], n);
        } else {
            rec = make_uint4((u32)q0, (u32)(q1 - q0), 0u, 0xffffffffu);
        }
        rrec[(u64)rowptr[a] + slot] = rec;
    }
}

__device__ __forceinline__ bool pair_keep(u32 cnt, u32 i, u32 j, const u32* __restrict__ sizes, double c_relaxed) {
    if (cnt == 0 || i == j) return false;
    const u32 si = sizes[i], sj = sizes[j];
    if (si == 0 || sj == 0) return false;
    // relaxed device-side filter; the exact `!(1.0*cnt/|R_i| < C)` of main.cpp:297-303 is applied
    // on the host to the survivors, so no decision depends on device floating point
    return !((double)cnt / (double)si < c_relaxed);
}

struct PairRows {
    const uint4* rrec;    // [postings] reference-major records
    const u32* rowptr;    // [N + 1]
    const u64* frec;      // FUSED (yh_db::fz): 8-byte records, one per CSR position (yh_db::d_fz_rec) ...
    const u64* foff;      // ... the rows = the sketches' extents (yh_db::d_fz_off); pr = yh_db::d_fz_list; cid / rid unused (identity)
    const u32* pr;        // hash-major holders (the long lists)
    const u32* pr2;       // FUSED: the lists of the groups of spilled buckets (yh_db::d_fz_list2) ...
    u64 pr_split;         // ... which list records name by a start >= pr_split (entry q of them: pr2[q - pr_split])
    const u32* cid;       // [N] compact id of a reference that holds a shared hash
    const u32* rid;       // [NC] back
    const u32* sizes;     // [N]
    u64 a0;               // first reference of the launch (blockIdx.x = a - a0)
    u64 seg0;             // its first segment
    u64 nseg;             // segments of the whole call (rows x column blocks)
    u32 NC, cols;         // compact references; columns per block (blockIdx.y)
    double c_relaxed;
    u32* segcnt;          // [nseg] survivors of the segment
    u64* segoff;          // ... and where they start in out[]
    unsigned long long* cursor;
    u64 cap;              // entries of out[] behind the fixed slots (a segment that does not fit is counted, not written)
    uint2* out;           // (reference j, count): [nseg * PAIR_SLOTS] fixed slots, then the cursor's area
    const u32* rowlist;   // != NULL: blockIdx.x names rowlist[blockIdx.x] (the rows k_pair_rows_sparse handed back), not a0 + blockIdx.x
    // k_pair_rows_sparse:
    u64 rows;             // rows of the call: a0 .. a0 + rows
    u32 ncb;              // segments per row in segcnt / segoff (the dense geometry's column blocks: a row handed back fills them all)
    u32* ovf_count;       // rows whose touched columns outgrew the table ...
    u32* ovf_rows;        // ... and which (k_pair_rows takes them: rowlist)
};

// A segment of up to PAIR_SLOTS survivors has its own place in out[] (a cluster of genomes: a handful of mates per
// row); only longer ones claim room behind the slots with an atomic (10^4 workgroups adding to ONE word: ~120 us).
constexpr u32 PAIR_SLOTS = 4;
#ifndef YH_PAIR_U
#define YH_PAIR_U 4
#endif
constexpr int PAIR_U = YH_PAIR_U;  // records a lane has in flight

#ifndef YH_ABLATE_PAIR
#define YH_ABLATE_PAIR 0  // timing-only builds (results wrong): 1 no record pass, 2 no row clear / survivor scan, 4 records read but not added, 8 records read but not decoded
#endif
// FUSED: the records are yh_db::d_fz_rec's (8 bytes per CSR position, written by the sort's last pass: yh_sort.hip) and a
// row is the extent of the reference's sketch -- entries of hashes nobody else holds are 0 and add nothing.
template <bool FUSED> struct PairRec { typedef uint4 T; };
template <> struct PairRec<true> { typedef u64 T; };
constexpr u64 FZ_LIST = 1ull << 63;
constexpr u32 FZ_FIELD = (1u << 21) - 1u;

// HALF: the row's counts are 16 bits wide, two columns to the LDS word (a count never exceeds the smaller of the two sketches:
// the host asks for this form when no sketch has more than 65 535 hashes).  Half the LDS per row is twice the resident rows: a
// row's phases -- clear, records, survivors -- come one after the other inside a workgroup and the workgroups of a CU start
// together, so few large workgroups leave the memory pipe idle while they count and the ALUs idle while they read; at
// configs[3] four 512-lane rows per CU with 40 KB each took 227 us, seven 256-lane rows with 20 KB each take 209 (profiles/r05/sweep_pair_half.txt).
template <int THREADS, bool FUSED, bool HALF>
__global__ void __launch_bounds__(THREADS) k_pair_rows(const PairRows p) {
    constexpr int WAVES = THREADS / WAVE;
    typedef typename PairRec<FUSED>::T Rec;
    extern __shared__ u32 row[];  // p.cols counts (HALF: two to the word)
    __shared__ u32 wtot[WAVES];
    __shared__ u64 lqueue[WAVES][64];  // a wave's queue of list records (below)
    __shared__ u64 s_base;
    __shared__ u32 s_write;
    const u32 tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const u64 a = p.rowlist ? (u64)p.rowlist[blockIdx.x] : p.a0 + blockIdx.x;
    const u32 c0 = blockIdx.y * p.cols;
    const u32 w = min(p.NC - c0, p.cols);
    const u64 seg = p.seg0 + (a - p.a0) * gridDim.y + blockIdx.y;
    const Rec* recs;
    u32 t1;  // records of the row: recs[0 .. t1)
    if constexpr (FUSED) {
        const u64 b = p.foff[a];
        recs = p.frec + b;
        t1 = (u32)(p.foff[a + 1] - b);
    } else {
        const u32 b = p.rowptr[a];
        recs = p.rrec + b;
        t1 = p.rowptr[a + 1] - b;
    }
    const u32 t0 = 0;
    if (t0 == t1) {  // (uniform) no shared hash: no pair
        if (tid == 0) { p.segcnt[seg] = 0; p.segoff[seg] = 0; }
        return;
    }
    auto mask_off = [](Rec& r) { if constexpr (FUSED) r = 0; else r.w = 0; };
    // the first records are in flight while the row is cleared: PAIR_U loads per lane before any is used (one load in
    // flight per wave left the pass at 1.4 TB/s of record reads -- 0.3 ms at configs[3], the adds themselves are free)
    Rec rec[PAIR_U];
#pragma unroll
    for (int u = 0; u < PAIR_U; ++u) {
        const u32 t = t0 + u * THREADS + tid;
        rec[u] = recs[min(t, t1 - 1)];  // (unconditional, masked below: a load under `if` is waited for on the spot)
        if (t >= t1) mask_off(rec[u]);
    }
    if (!(YH_ABLATE_PAIR & 2))
        for (u32 j = tid; j < (HALF ? (w + 1u) >> 1 : w); j += THREADS) row[j] = 0;
    __syncthreads();
    // (Same-address LDS adds are not what the pass waits for: counting a lane's first four columns in registers and
    // adding them once made it slower; without any add it is 15 % shorter.)
    auto addc = [&](u32 cc) {  // cc: a compact id
        const u32 c = cc - c0;
        if (c >= w) return;
        if (YH_ABLATE_PAIR & 4) { if (c == 0x12345u) row[0] = 1; return; }
        if constexpr (HALF) atomicAdd(&row[c >> 1], 1u << ((c & 1u) * 16u));
        else atomicAdd(&row[c], 1u);
    };
    // (Tried, round 5: the adds of a wave step combined per column first -- the first active lane's column by readlane, who
    // else has it by ballot, one add of the population count; up to 2 / 4 / 8 distinct columns per step, the rest on their own --
    // because without any add the pass is 110 us against 208: 323 / 447 / 443 us.  The LDS takes 64 adds to a handful of words
    // far better than the wave takes the loop: profiles/r05/sweep_pair_agg.txt.)
    auto count_of = [&](u32 j) -> u32 {  // column j's count
        if constexpr (HALF) return (row[j >> 1] >> ((j & 1u) * 16u)) & 0xffffu;
        else return row[j];
    };
    auto add = [&](u32 o) { if constexpr (FUSED) addc(o); else addc(p.cid[o]); };  // o: a reference id (the long lists)
    // LIST records -- "the other holders are entries [q0, q0 + m) of the holder array" -- are not walked where they are met:
    // a load inside the record loop is waited for with everything in flight before it (vmcnt counts in order), i.e. with the
    // next step's records, and half of configs[3]'s wave steps meet one (a hash all five genomes of a cluster hold).  A wave
    // QUEUES them (64 descriptors of its own in LDS) and walks the queue when it is full and behind the row's last record:
    // all queued lists flattened over the lanes -- entry t of the concatenation belongs to the list whose exclusive length
    // prefix is the largest <= t (six ds_bpermute steps) -- so that 64 holders are requested per round whatever the lists'
    // lengths: a cluster's five-entry lists a dozen to the round, a k-mer of 5 000 holders in 79 rounds (before: one holder
    // after the other by the record's own lane up to 32 holders, one list per round above).
    // (profiles/r05/ablate_pair.txt: the records read but not decoded 86 us, decoded with the lists walked in place 217.)
    auto flush = [&](u32 nq) {
        const u64 d = lane < nq ? lqueue[wid][lane] : 0ull;
        const u32 m = (u32)(d >> 40);  // (24 bits)
        u32 inc = m;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 t_ = (u32)__shfl_up((int)inc, off);
            if (lane >= (u32)off) inc += t_;
        }
        const u32 total = (u32)__shfl((int)inc, 63);
        const u32 exc = lane < nq ? inc - m : 0xffffffffu;  // (behind the queue's end: above every t)
        const u32 dlo = (u32)d, dhi = (u32)(d >> 32);
        for (u32 t0_ = 0; t0_ < total; t0_ += 64u) {  // (wave-uniform)
            const u32 t = t0_ + lane;
            u32 e = 0;
#pragma unroll
            for (u32 step = 32; step >= 1; step >>= 1) {
                const u32 cand = e + step;  // (<= 63)
                const u32 v = (u32)__shfl((int)exc, (int)cand);
                if (v <= t) e = cand;
            }
            const u32 e_exc = (u32)__shfl((int)exc, (int)e);
            const u64 de = ((u64)(u32)__shfl((int)dhi, (int)e) << 32) | (u32)__shfl((int)dlo, (int)e);
            if (t < total) {
                u64 q0 = de & ((1ull << 40) - 1ull);
                const u32* lsrc = p.pr;
                if constexpr (FUSED) {
                    if (q0 >= p.pr_split) { lsrc = p.pr2; q0 -= p.pr_split; }
                }
                const u32 o = lsrc[q0 + (t - e_exc)];
                if (o != (u32)a) add(o);
            }
        }
    };
    u32 nq = 0;  // descriptors in this wave's queue (wave-uniform)
    for (u32 tb = t0; tb < ((YH_ABLATE_PAIR & 1) ? t0 : t1); tb += PAIR_U * THREADS) {  // (workgroup-uniform bounds: the ballot below)
        Rec cur[PAIR_U];
#pragma unroll
        for (int u = 0; u < PAIR_U; ++u) {
            cur[u] = rec[u];
            const u32 t = tb + (PAIR_U + u) * THREADS + tid;  // the next step's records
            rec[u] = recs[min(t, t1 - 1)];
            if (t >= t1) mask_off(rec[u]);
        }
#pragma unroll
        for (int u = 0; u < PAIR_U; ++u) {
            if constexpr (FUSED && (YH_ABLATE_PAIR & 8) != 0) {  // (timing only: the records read, nothing decoded)
                if (cur[u] == 0x123456789abcull) row[0] = 1;
                continue;
            }
            bool is_list;
            u64 lq0 = 0;   // a list: its first entry in pr[] ...
            u32 lm = 0;    // ... and its length (this reference included)
            if constexpr (FUSED) {
                const u64 r = cur[u];
                is_list = (r & FZ_LIST) != 0;
                if (!is_list) {  // three 21-bit fields: reference + 1, 0 = none
                    const u32 f0 = (u32)r & FZ_FIELD, f1 = (u32)(r >> 21) & FZ_FIELD, f2 = (u32)(r >> 42) & FZ_FIELD;
                    if (f0) addc(f0 - 1u);
                    if (f1) addc(f1 - 1u);
                    if (f2) addc(f2 - 1u);
                } else {
                    lq0 = r & ((1ull << 40) - 1ull);
                    lm = (u32)(r >> 40) & ((1u << 22) - 1u);
                }
            } else {
                is_list = cur[u].w == 0xffffffffu;
                if (!is_list) {
                    if (cur[u].w > 0) addc(cur[u].x);
                    if (cur[u].w > 1) addc(cur[u].y);
                    if (cur[u].w > 2) addc(cur[u].z);
                } else {
                    lq0 = cur[u].x;
                    lm = cur[u].y;
                }
            }
            if (!FUSED && is_list && lm >= (1u << 24)) {  // (no room for its length in a descriptor: a hash 16 million references hold)
                for (u64 q = lq0, qe = lq0 + lm; q < qe; ++q) {
                    const u32 o = p.pr[q];
                    if (o != (u32)a) add(o);
                }
                is_list = false;
            }
            // a list: queued (the walk is the wave's, behind the records)
            const u64 bal = __ballot(is_list);
            if (bal) {  // (wave-uniform)
                const u32 c = (u32)__popcll(bal);
                if (nq + c > 64u) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    flush(nq);
                    nq = 0;
                }
                if (is_list) lqueue[wid][nq + (u32)__popcll(bal & ((1ull << lane) - 1ull))] = (lq0 & ((1ull << 40) - 1ull)) | ((u64)lm << 40);
                nq += c;
            }
        }
    }
    if (nq) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        flush(nq);
    }
    __syncthreads();
    // survivors in column order: wave v owns the columns [v*per, (v+1)*per)
    const u32 per = ((w + WAVES * 64u - 1) / (WAVES * 64u)) * 64u;
    const u32 jb = min(w, wid * per), je = min(w, jb + per);
    u32 mine = 0;
    for (u32 j0 = jb; j0 < ((YH_ABLATE_PAIR & 2) ? jb : je); j0 += 64u) {
        const u32 j = j0 + lane;
        const u32 cnt = j < je ? count_of(j) : 0u;  // (rows are sparse: rid / sizes only behind a count)
        const bool keep = cnt != 0 && pair_keep(cnt, (u32)a, FUSED ? c0 + j : p.rid[c0 + j], p.sizes, p.c_relaxed);
        mine += (u32)__popcll(__ballot(keep));
    }
    if (lane == 0) wtot[wid] = mine;
    __syncthreads();
    if (tid == 0) {
        u32 total = 0;
        for (int v = 0; v < WAVES; ++v) total += wtot[v];
        u64 base = seg * PAIR_SLOTS;
        bool fits = true;
        if (total > PAIR_SLOTS) {
            const u64 at = atomicAdd(p.cursor, (unsigned long long)total);
            base = p.nseg * PAIR_SLOTS + at;
            fits = at + total <= p.cap;
        }
        p.segcnt[seg] = total;
        p.segoff[seg] = base;
        s_base = base;
        s_write = (total && fits) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_write) return;
    u64 dst = s_base;
    for (u32 v = 0; v < wid; ++v) dst += wtot[v];
    if (wtot[wid] == 0) return;  // (wave-uniform)
    for (u32 j0 = jb; j0 < je; j0 += 64u) {
        const u32 j = j0 + lane;
        const u32 cnt = j < je ? count_of(j) : 0u;
        u32 rj = 0;
        bool keep = false;
        if (cnt != 0) {
            rj = FUSED ? c0 + j : p.rid[c0 + j];
            keep = pair_keep(cnt, (u32)a, rj, p.sizes, p.c_relaxed);
        }
        const u64 bal = __ballot(keep);
        if (keep) p.out[dst + (u64)__popcll(bal & ((1ull << lane) - 1ull))] = make_uint2(rj, cnt);
        dst += (u64)__popcll(bal);
    }
}

// ---- the row pass for MANY references: sparse rows (round 6) ---------------------------------------------------------------
// k_pair_rows keeps a row's counts DENSE in LDS: at the scale the reference publishes (README.md:276: 85 205 genomes) that is
// 170 KB of 16-bit columns per row -- three column blocks of 28 416, each clearing and scanning its 56 KB for the handful of
// cluster mates a genome has, each reading the row's records again: 4.7 ms, nearly all of it clears and scans
// (profiles/r05/rs214_rows_sweep.txt).  Here a row's counts live in a HASH TABLE over the columns it touches (SP_SLOTS slots:
// key = column + 1, count beside it; linear probing) with a list of the slots claimed: nothing is cleared per row -- a
// persistent workgroup resets exactly the slots its row claimed -- and nothing is scanned but that list.  What a row costs is
// its records (8 bytes per CSR position, streamed once) and its adds; a row that touches nothing (nine in ten at rs214 scale)
// costs its records alone.  A row whose touched columns outgrow the list (a hot k-mer: thousands of holders) is handed back
// through ovf_rows and takes the dense pass (k_pair_rows with rowlist); the host sorts a sparse row's survivors by column
// (they leave in the order the slots were claimed).  FUSED handles only (yh_db::fz -- `yacht train`'s).
constexpr u32 SP_THREADS = 256;
constexpr u32 SP_BITS = 12;
constexpr u32 SP_SLOTS = 1u << SP_BITS;   // 32 KB of table (key + count)
constexpr u32 SP_LIST = SP_SLOTS / 2;     // claimed slots a row may have: the table is never more than half full
constexpr int SP_U = 8;                   // records a lane has in flight
__global__ void __launch_bounds__(SP_THREADS) k_pair_rows_sparse(const PairRows p) {
    constexpr int WAVES = SP_THREADS / WAVE;
    __shared__ u32 tkey[SP_SLOTS], tcnt[SP_SLOTS];
    __shared__ u16 tlist[SP_LIST];
    __shared__ u64 lqueue[WAVES][64];
    __shared__ u32 wtot[WAVES];
    __shared__ u32 s_nt[2], s_ovf[2], s_write;  // (per row PARITY: a wave that is through with row k adds for row k + 1 while others still read row k's)
    __shared__ u64 s_base;
    const u32 tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    for (u32 k = tid; k < SP_SLOTS; k += SP_THREADS) { tkey[k] = 0; tcnt[k] = 0; }
    if (tid < 2) { s_nt[tid] = 0; s_ovf[tid] = 0; }
    __syncthreads();
    u32 par = 1;
    for (u64 rix = blockIdx.x; rix < p.rows; rix += gridDim.x) {  // (workgroup-uniform)
        par ^= 1u;
        const u64 a = p.a0 + rix;
        const u64 seg = p.seg0 + rix * p.ncb;
        const u64 fb = p.foff[a];
        const u64* recs = p.frec + fb;
        const u32 t1 = (u32)(p.foff[a + 1] - fb);
        if (tid < p.ncb) { p.segcnt[seg + tid] = 0; p.segoff[seg + tid] = 0; }  // (ncb <= 65 535 / rows of >= 1 segment; see the host)
        for (u32 k = SP_THREADS + tid; k < p.ncb; k += SP_THREADS) { p.segcnt[seg + k] = 0; p.segoff[seg + k] = 0; }
        if (t1 == 0) continue;
        auto add = [&](u32 col) {
            if (col == (u32)a) return;
            u32 sl = (col * 2654435761u) >> (32 - SP_BITS);
            for (u32 probe = 0; probe < SP_SLOTS; ++probe, sl = (sl + 1u) & (SP_SLOTS - 1u)) {
                const u32 old = atomicCAS(&tkey[sl], 0u, col + 1u);
                if (old == 0u) {  // claimed: on the row's list (a list that is full hands the row back: the claim is undone by the full reset below)
                    const u32 k = atomicAdd(&s_nt[par], 1u);
                    if (k < SP_LIST) tlist[k] = (u16)sl; else s_ovf[par] = 1u;
                    atomicAdd(&tcnt[sl], 1u);
                    return;
                }
                if (old == col + 1u) { atomicAdd(&tcnt[sl], 1u); return; }
                if (s_ovf[par]) return;  // (the row is lost already; and a table nobody reads may fill up)
            }
        };
        auto flush = [&](u32 nq) {  // (k_pair_rows' walk of a wave's queued list records, flattened over the lanes)
            const u64 d = lane < nq ? lqueue[wid][lane] : 0ull;
            const u32 m = (u32)(d >> 40);
            u32 inc = m;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 t_ = (u32)__shfl_up((int)inc, off);
                if (lane >= (u32)off) inc += t_;
            }
            const u32 total = (u32)__shfl((int)inc, 63);
            const u32 exc = lane < nq ? inc - m : 0xffffffffu;
            const u32 dlo = (u32)d, dhi = (u32)(d >> 32);
            for (u32 t0_ = 0; t0_ < total; t0_ += 64u) {
                const u32 t = t0_ + lane;
                u32 e = 0;
#pragma unroll
                for (u32 step = 32; step >= 1; step >>= 1) {
                    const u32 cand = e + step;
                    const u32 v = (u32)__shfl((int)exc, (int)cand);
                    if (v <= t) e = cand;
                }
                const u32 e_exc = (u32)__shfl((int)exc, (int)e);
                const u64 de = ((u64)(u32)__shfl((int)dhi, (int)e) << 32) | (u32)__shfl((int)dlo, (int)e);
                if (t < total) {
                    u64 q0 = de & ((1ull << 40) - 1ull);
                    const u32* lsrc = p.pr;
                    if (q0 >= p.pr_split) { lsrc = p.pr2; q0 -= p.pr_split; }
                    add(lsrc[q0 + (t - e_exc)]);
                }
            }
        };
        u64 rec[SP_U];
#pragma unroll
        for (int u = 0; u < SP_U; ++u) {
            const u32 t = u * SP_THREADS + tid;
            rec[u] = recs[min(t, t1 - 1)];
            if (t >= t1) rec[u] = 0;
        }
        u32 nq = 0;
        for (u32 tb = 0; tb < t1; tb += SP_U * SP_THREADS) {  // (workgroup-uniform bounds)
            u64 cur[SP_U];
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                cur[u] = rec[u];
                const u32 t = tb + (SP_U + u) * SP_THREADS + tid;
                rec[u] = recs[min(t, t1 - 1)];
                if (t >= t1) rec[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                const u64 r = cur[u];
                if (__ballot(r != 0) == 0ull) continue;  // (wave-uniform: most wave steps of most rows)
                const bool is_list = (r & FZ_LIST) != 0;
                if (!is_list) {
                    const u32 f0 = (u32)r & FZ_FIELD, f1 = (u32)(r >> 21) & FZ_FIELD, f2 = (u32)(r >> 42) & FZ_FIELD;
                    if (f0) add(f0 - 1u);
                    if (f1) add(f1 - 1u);
                    if (f2) add(f2 - 1u);
                }
                const u64 bal = __ballot(is_list);
                if (bal) {
                    const u32 c = (u32)__popcll(bal);
                    if (nq + c > 64u) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        flush(nq);
                        nq = 0;
                    }
                    if (is_list) lqueue[wid][nq + (u32)__popcll(bal & ((1ull << lane) - 1ull))] = r & ~FZ_LIST;  // {length << 40 | first entry}: the descriptor as it is
                    nq += c;
                }
            }
        }
        if (nq) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            flush(nq);
        }
        __syncthreads();
        const u32 nt = s_nt[par];
        if (nt == 0) continue;  // (uniform) nothing touched: s_nt and the table are as they were
        if (s_ovf[par]) {       // (uniform) handed back; the table is reset whole (rare)
            __syncthreads();
            for (u32 k = tid; k < SP_SLOTS; k += SP_THREADS) { tkey[k] = 0; tcnt[k] = 0; }
            if (tid == 0) { p.ovf_rows[atomicAdd(p.ovf_count, 1u)] = (u32)a; s_nt[par] = 0; s_ovf[par] = 0; }
            __syncthreads();
            continue;
        }
        // survivors of the relaxed threshold among the touched columns (two walks over the list: count, then write)
        u32 mine = 0;
        for (u32 k0 = 0; k0 < nt; k0 += SP_THREADS) {
            const u32 k = k0 + tid;
            bool keep = false;
            if (k < nt) { const u32 sl = tlist[k]; keep = pair_keep(tcnt[sl], (u32)a, tkey[sl] - 1u, p.sizes, p.c_relaxed); }
            mine += (u32)__popcll(__ballot(keep));
        }
        if (lane == 0) wtot[wid] = mine;
        __syncthreads();
        if (tid == 0) {
            u32 total = 0;
            for (int v = 0; v < WAVES; ++v) total += wtot[v];
            u64 base = seg * PAIR_SLOTS;
            bool fits = true;
            if (total > PAIR_SLOTS) {
                const u64 at = atomicAdd(p.cursor, (unsigned long long)total);
                base = p.nseg * PAIR_SLOTS + at;
                fits = at + total <= p.cap;
            }
            p.segcnt[seg] = total;
            p.segoff[seg] = base;
            s_base = base;
            s_write = (total && fits) ? 1u : 0u;
            s_nt[par] = 0;
        }
        __syncthreads();
        u64 dst = s_base;
        for (u32 v = 0; v < wid; ++v) dst += wtot[v];
        const bool wr = s_write != 0;
        for (u32 k0 = 0; k0 < nt; k0 += SP_THREADS) {
            const u32 k = k0 + tid;
            bool keep = false;
            u32 rj = 0, cnt = 0;
            if (k < nt) {
                const u32 sl = tlist[k];
                rj = tkey[sl] - 1u;
                cnt = tcnt[sl];
                keep = pair_keep(cnt, (u32)a, rj, p.sizes, p.c_relaxed);
                tkey[sl] = 0;  // the row's slots back to empty: nothing else is ever cleared
                tcnt[sl] = 0;
            }
            const u64 bal = __ballot(keep);
            if (keep && wr) p.out[dst + (u64)__popcll(bal & ((1ull << lane) - 1ull))] = make_uint2(rj, cnt);
            dst += (u64)__popcll(bal);
        }
        __syncthreads();  // (the reset is complete before the next row's adds)
    }
}

// a fused handle counts a reference's shared hashes only when asked (yh_db_nshared_device): its non-zero records, a wave per row
__global__ void __launch_bounds__(256) k_fz_nshared(const u64* __restrict__ rec, const u64* __restrict__ off, u64 n_refs, u32* __restrict__ nshared) {
    const u64 a = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6;
    const u32 lane = threadIdx.x & 63u;
    if (a >= n_refs) return;
    u32 c = 0;
    for (u64 i = off[a] + lane, e = off[a + 1]; i < e; i += 64) c += rec[i] != 0 ? 1u : 0u;
    for (int d = 32; d > 0; d >>= 1) c += (u32)__shfl_down((int)c, d);
    if (lane == 0) nshared[a] = c;
}

inline u32 grid_for(u64 work_items, u32 block, u32 max_blocks = 16384) {
    u64 g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (u32)g;
}

}  // namespace

int yh_q_fz_nshared(yh_db* db) {
    if (!db->fz || db->fz_nshared || db->n_refs == 0) return YH_OK;
    k_fz_nshared<<<(u32)((db->n_refs * 64 + 255) / 256), 256, 0, db->stream>>>(db->d_fz_rec, db->d_fz_off, db->n_refs, db->d_nshared);
    YH_HIP(hipGetLastError());
    db->fz_nshared = true;
    return YH_OK;
}

// Fills the handle's host-side pair cache (h_pw_*) for rows [r0, r1).
int yh_q_pairwise(yh_db* db, double c_thresh, u64 r0, u64 r1) {
    if (!db->has_index) { yh_set_error("this handle was created with YH_DB_NO_INDEX"); return YH_ERR_UNSUPPORTED; }
    hipStream_t st = db->stream;
    const u64 N = db->n_refs;
    const u64 P = db->n_postings;
    free(db->h_pw_i); free(db->h_pw_j); free(db->h_pw_c);
    db->h_pw_i = db->h_pw_j = db->h_pw_c = nullptr;
    db->pw_n = 0;
    db->pw_valid = false;
    db->pw_sparse_rows = db->pw_dense_rows = 0;  // (yh_pairwise_row_stats: of THIS call)
    if (r1 > N) r1 = N;
    auto done_empty = [&]() { db->pw_valid = true; db->pw_c = c_thresh; db->pw_r0 = r0; db->pw_r1 = r1; return YH_OK; };
    if (r0 >= r1 || P == 0) return done_empty();

    // compact ids of the references that hold a shared hash (ascending with the reference id), and where a reference's
    // records start: one staging buffer [rowptr (N + 1) | cid (N) | rid (NC)], one copy up
    // (a fused handle -- yh_db::fz -- has its records in place, one per CSR position: the rows are the sketches' extents, the
    // columns the reference ids themselves, and neither a count per reference nor a table is needed)
    const bool fz = db->fz;
    std::vector<u32> h_tab;
    u64 NC = 0;
    if (!fz) {
        std::vector<u32> h_nsh(N);
        YH_HIP(hipMemcpyAsync(h_nsh.data(), db->d_nshared, N * sizeof(u32), hipMemcpyDeviceToHost, st));
        YH_HIP(hipStreamSynchronize(st));
        h_tab.resize(2 * N + 1 + N);
        u32* h_rowptr = h_tab.data();
        u32* h_cid = h_tab.data() + N + 1;
        u32* h_rid = h_tab.data() + 2 * N + 1;
        u64 acc = 0;
        for (u64 j = 0; j < N; ++j) {
            h_rowptr[j] = (u32)acc;
            acc += h_nsh[j];
            if (h_nsh[j]) { h_cid[j] = (u32)NC; h_rid[NC++] = (u32)j; }
            else h_cid[j] = 0xffffffffu;
        }
        h_rowptr[N] = (u32)acc;
        if (acc != P) { yh_set_error("posting counts do not add up (%llu of %llu)", (u64)acc, (u64)P); return YH_ERR_HIP; }
    } else {
        NC = N;
    }
    if (NC == 0) return done_empty();

    // columns per row block: what the LDS holds
    typedef void (*RowsKernel)(const PairRows);
    // 16-bit counts (k_pair_rows<.., HALF>): a fused handle that knows its sketch sizes and has none above 65 535 hashes
    static const bool no_half = [] { const char* e = yh_tune_env("YH_PAIR_NO_HALF"); return e && e[0] == '1'; }();
    const bool half = fz && !no_half && db->h_sizes.size() == N && db->max_ref_size <= 0xffffu;
    // Lanes per row and columns per block go by what a CU can hold: a row block's LDS is its columns (2 or 4 bytes each), and the
    // WAVES resident per CU -- not the lanes of one row -- decide how well the phases of different rows overlap.  A row block is
    // kept at <= 56 KB (two per CU) by cutting the columns into as many blocks as that takes, although every block reads the row's
    // records again; and it gets the lanes that fill the CU's 32 waves at its size: 256 up to 24 KB (seven blocks per CU:
    // configs[3]'s 10 000 16-bit columns, 209 us against 227 with 512), 512 up to 40 KB, 1 024 above.  At 85 205 references:
    // three blocks of 28 416 columns, 1 024 lanes: 4.8 ms (512 lanes: 6.1); two blocks of 73 728 -- one per CU --: 19.9 / 10.8 /
    // 7.1 ms at 256 / 512 / 1 024 lanes; round 4's 32-bit rows in three blocks of 36 864, 512 lanes: 11.5
    // (profiles/r05/rs214_rows_sweep.txt, sweep_pair_half.txt).
    static const int threads_env = [] { const char* e = yh_tune_env("YH_PAIR_THREADS"); const int t = e ? atoi(e) : 0; return (t == 1024 || t == 512 || t == 256) ? t : 0; }();
    const u32 bytes_per_col = half ? 2u : 4u;
    static const bool big_lds = [] {
        bool ok = true;
        for (const RowsKernel k : {(RowsKernel)k_pair_rows<1024, false, false>, (RowsKernel)k_pair_rows<512, false, false>, (RowsKernel)k_pair_rows<256, false, false>,
                                   (RowsKernel)k_pair_rows<1024, true, false>, (RowsKernel)k_pair_rows<512, true, false>, (RowsKernel)k_pair_rows<256, true, false>,
                                   (RowsKernel)k_pair_rows<1024, true, true>, (RowsKernel)k_pair_rows<512, true, true>, (RowsKernel)k_pair_rows<256, true, true>}) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)(PAIR_COLS_BIG * sizeof(u32)));
            if (e != hipSuccess) { (void)hipGetLastError(); ok = false; }
        }
        return ok;
    }();
    const u32 cols_cap = (big_lds ? PAIR_COLS_BIG : PAIR_COLS_SMALL) * (half ? 2u : 1u);  // (columns: what the LDS words can hold)
    const u32 cols_target = std::min<u32>(cols_cap, 57344u / bytes_per_col);
    const u64 nblk_want = (NC + cols_target - 1) / cols_target;
    u32 cols = (u32)std::min<u64>(cols_cap, ((NC + nblk_want - 1) / nblk_want + 63) / 64 * 64);  // (the blocks of a row alike)
    if (const char* e = yh_tune_env("YH_PAIR_COLS")) cols = (u32)std::min<u64>(cols_cap, std::max(64, atoi(e)));  // (tests: several column blocks)
    cols = (u32)std::min<u64>(cols, (NC + 63) / 64 * 64);
    const u64 block_bytes = (u64)cols * bytes_per_col;
    const int threads = threads_env ? threads_env : (block_bytes <= 24576u ? (half ? 256 : 512) : block_bytes <= 40960u ? 512 : 1024);
    const RowsKernel kern_plain = threads == 1024 ? k_pair_rows<1024, false, false> : threads == 256 ? k_pair_rows<256, false, false> : k_pair_rows<512, false, false>;
    const RowsKernel kern_fused = threads == 1024 ? k_pair_rows<1024, true, false> : threads == 256 ? k_pair_rows<256, true, false> : k_pair_rows<512, true, false>;
    const RowsKernel kern_half = threads == 1024 ? k_pair_rows<1024, true, true> : threads == 256 ? k_pair_rows<256, true, true> : k_pair_rows<512, true, true>;
    const RowsKernel kern = half ? kern_half : fz ? kern_fused : kern_plain;
    const u32 ncb = (u32)((NC + cols - 1) / cols);
    if (ncb > 65535) { yh_set_error("too many column blocks"); return YH_ERR_UNSUPPORTED; }
    const u64 rows = r1 - r0;
    const u64 nseg = rows * ncb;
    // Sparse rows (k_pair_rows_sparse) when a dense row takes more than one column block -- more than 28 672 references at 16-bit
    // counts: the scale the reference publishes -- on a fused handle; YH_PAIR_SPARSE=0 / 1 forces either form (tests run both).
    static const int sparse_env = [] { const char* e = yh_tune_env("YH_PAIR_SPARSE"); return e ? (e[0] == '1' ? 1 : e[0] == '0' ? 0 : -1) : -1; }();
    const bool sparse = fz && (sparse_env < 0 ? ncb > 1 : sparse_env == 1);

    // scratch: [records 16 P | segoff 8 nseg | cursor 8 | tab 4 (2N + 1 + NC) | segcnt 4 nseg | cursors 4 N]
    const bool ranks = db->d_prank != nullptr || fz;
    const u64 b_rec = fz ? 0 : P * sizeof(uint4), b_off = nseg * sizeof(u64), b_tab = fz ? 0 : (2 * N + 1 + NC) * sizeof(u32);
    const u64 b_cnt = nseg * sizeof(u32), b_cur = ranks ? 0 : N * sizeof(u32);
    char* d_scr = nullptr;
    uint2* d_out = nullptr;
    u32* d_ovf = nullptr;  // sparse rows: [count | the rows handed back to the dense pass]
    int rc = YH_OK;
#define PW_HIP(call)                                                                          \
    if (rc == YH_OK) {                                                                        \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            yh_set_error("%s failed: %s", #call, hipGetErrorString(e__));                     \
            rc = (e__ == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;                      \
        }                                                                                     \
    }
    // the survivors: PAIR_SLOTS per segment, and room for a million behind them to begin with; a run whose long segments
    // need more is repeated with what it counted
    const u64 n_slots = nseg * PAIR_SLOTS;
    u64 cap = std::max<u64>(std::min<u64>(NC * (NC - 1), 1u << 20), 1);
    // One block: [records | segoff | cursor | tab | segcnt | cursors | (pad) | survivors].  On a fused handle with ranks the
    // tab and the cursors are empty, and everything the host reads -- offsets, the cursor, counts, the survivors' slots and the
    // first `spec` survivors behind them -- is ONE contiguous piece: one copy into page-locked memory, one wait.  (Five copies
    // into pageable vectors around two waits were ~0.14 of configs[3]'s 0.40 ms; five queued copies still ~13 us each.)
    const u64 b_meta = b_off + 8 + b_tab + b_cnt + b_cur;
    const u64 b_meta_pad = (b_meta + 15) & ~(u64)15;
    PW_HIP(yh_tmalloc(db, (void**)&d_scr, b_rec + b_meta_pad + (n_slots + cap) * sizeof(uint2) + 64));
    uint4* d_rrec = reinterpret_cast<uint4*>(d_scr);
    u64* d_segoff = reinterpret_cast<u64*>(d_scr + b_rec);
    unsigned long long* d_cursor = reinterpret_cast<unsigned long long*>(d_scr + b_rec + b_off);
    u32* d_tab = reinterpret_cast<u32*>(d_scr + b_rec + b_off + 8);
    u32* d_rowptr = d_tab;
    u32* d_cid = d_tab + N + 1;
    u32* d_rid = d_tab + 2 * N + 1;
    u32* d_segcnt = d_tab + b_tab / sizeof(u32);
    u32* d_cur = d_segcnt + nseg;
    uint2* d_out_own = nullptr;  // (a second attempt's larger output)
    if (sparse) PW_HIP(yh_tmalloc(db, (void**)&d_ovf, (rows + 2) * sizeof(u32)));
    d_out = reinterpret_cast<uint2*>(d_scr + b_rec + b_meta_pad);
    if (!fz) PW_HIP(hipMemcpyAsync(d_tab, h_tab.data(), b_tab, hipMemcpyHostToDevice, st));
    if (!ranks) PW_HIP(hipMemsetAsync(d_cur, 0, b_cur, st));
    const double c_relaxed = c_thresh * (1.0 - 1e-9) - 1e-300;
    const u64 spec = std::min<u64>(cap, 1u << 16);
    const bool have_sizes = db->h_sizes.size() == N;  // (a YH_DB_PAIRWISE_ONLY handle fetched them when it was made)
    bool one_piece = b_tab == 0 && b_cur == 0;
    const u64 b_piece = b_meta_pad + (n_slots + spec) * sizeof(uint2);
    YhPin pin(b_piece + (have_sizes ? 0 : N * sizeof(u32)) + 64);
    char* const hp = static_cast<char*>(pin.p);
    if (!hp) one_piece = false;
    // host side: the same layout as the device's piece when it comes in one copy
    std::vector<u32> v_segcnt(hp ? 0 : nseg);
    std::vector<u64> v_segoff(hp ? 0 : nseg);
    unsigned long long n_over_stack = 0;
    u64* const h_segoff = hp ? reinterpret_cast<u64*>(hp) : v_segoff.data();
    unsigned long long* const p_n_over = hp ? reinterpret_cast<unsigned long long*>(hp + b_off) : &n_over_stack;
    u32* const h_segcnt = hp ? reinterpret_cast<u32*>(hp + b_off + 8 + b_tab) : v_segcnt.data();
    uint2* const h_out_pin = hp ? reinterpret_cast<uint2*>(hp + b_meta_pad) : nullptr;
    u32* const h_sizes_pin = (hp && !have_sizes) ? reinterpret_cast<u32*>(hp + b_piece) : nullptr;
    unsigned long long n_over = 0;
    yh_ring_record_begin(db, db->ev_pair);
    if (rc == YH_OK && !fz)
        k_pair_transpose<<<grid_for(P, 256, 1u << 20), 256, 0, st>>>(P, db->d_pr, db->d_pg, db->d_po, d_rowptr, db->d_prank, d_cur,
                                                                     d_cid, d_rrec);
    for (int attempt = 0; attempt < 2 && rc == YH_OK; ++attempt) {
        PW_HIP(hipMemsetAsync(d_cursor, 0, 8, st));
        PairRows q{d_rrec, d_rowptr, db->d_fz_rec, db->d_fz_off, fz ? db->d_fz_list : db->d_pr, db->d_fz_list2, fz ? db->fz_list_split : ~(u64)0, d_cid, d_rid, db->d_sizes, 0, 0, nseg,
                   (u32)NC, cols, c_relaxed, d_segcnt, d_segoff, d_cursor, cap, d_out};
        const u64 step = (1u << 31) / (u32)threads;  // (2^31 threads per grid dimension)
        if (sparse) {
            q.a0 = r0; q.seg0 = 0; q.rows = rows; q.ncb = ncb; q.ovf_count = d_ovf; q.ovf_rows = d_ovf + 1;
            PW_HIP(hipMemsetAsync(d_ovf, 0, sizeof(u32), st));
            // persistent workgroups (38 KB of LDS each: four resident per CU); rows by stride
            static const u32 sp_grid = [] { const char* e = yh_tune_env("YH_PAIR_SPARSE_GRID"); return e ? (u32)std::max(1, atoi(e)) : 0u; }();
            static const u32 n_cus = [] { int dev = 0, v = 0; return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? (u32)v : 256u; }();
            const u32 grid = (u32)std::min<u64>(rows, sp_grid ? sp_grid : n_cus * 16u);  // (four resident per CU; four times that for the balance: rows differ 50-fold in length -- 0.99 -> 0.87 ms at 85 205)
            if (rc == YH_OK) k_pair_rows_sparse<<<grid, SP_THREADS, 0, st>>>(q);
            u32 n_ovf = 0;
            YhPin pin_ovf(64);
            u32* h_ovf = pin_ovf.p ? static_cast<u32*>(pin_ovf.p) : &n_ovf;
            PW_HIP(hipMemcpyAsync(h_ovf, d_ovf, sizeof(u32), hipMemcpyDeviceToHost, st));
            PW_HIP(hipStreamSynchronize(st));
            n_ovf = *h_ovf;
            db->pw_sparse_rows = rows - n_ovf;
            db->pw_dense_rows = n_ovf;
            // the rows whose touched columns outgrew the table (hot k-mers): the dense pass, all their column blocks
            q.rowlist = d_ovf + 1;
            for (u64 b0 = 0; b0 < n_ovf && rc == YH_OK; b0 += step) {
                const u64 nb = std::min<u64>(n_ovf - b0, step);
                q.rowlist = d_ovf + 1 + b0;
                kern<<<dim3((u32)nb, ncb), threads, (half ? (cols + 1) / 2 : cols) * sizeof(u32), st>>>(q);
            }
            q.rowlist = nullptr;
        } else {
        for (u64 b0 = 0; b0 < rows && rc == YH_OK; b0 += step) {
            const u64 nb = std::min<u64>(rows - b0, step);
            q.a0 = r0 + b0;
            q.seg0 = b0 * ncb;
            kern<<<dim3((u32)nb, ncb), threads, (half ? (cols + 1) / 2 : cols) * sizeof(u32), st>>>(q);
        }
        }
        PW_HIP(hipGetLastError());
        if (attempt == 0) yh_ring_record_end(db, db->ev_pair);
        if (one_piece && attempt == 0) {
            PW_HIP(hipMemcpyAsync(hp, d_segoff, b_piece, hipMemcpyDeviceToHost, st));
        } else {
            PW_HIP(hipMemcpyAsync(p_n_over, d_cursor, 8, hipMemcpyDeviceToHost, st));
            PW_HIP(hipMemcpyAsync(h_segcnt, d_segcnt, b_cnt, hipMemcpyDeviceToHost, st));
            PW_HIP(hipMemcpyAsync(h_segoff, d_segoff, b_off, hipMemcpyDeviceToHost, st));
            if (hp) PW_HIP(hipMemcpyAsync(h_out_pin, d_out, (n_slots + std::min<u64>(spec, cap)) * sizeof(uint2), hipMemcpyDeviceToHost, st));
        }
        if (h_sizes_pin && attempt == 0) PW_HIP(hipMemcpyAsync(h_sizes_pin, db->d_sizes, N * sizeof(u32), hipMemcpyDeviceToHost, st));
        PW_HIP(hipStreamSynchronize(st));
        n_over = *p_n_over;
        if (rc != YH_OK || n_over <= cap) break;
        if (attempt == 1) { yh_set_error("pairwise: the second pass found more survivors than the first"); rc = YH_ERR_HIP; break; }
        cap = n_over;
        PW_HIP(yh_tmalloc(db, (void**)&d_out_own, (n_slots + cap) * sizeof(uint2)));
        d_out = d_out_own;
    }
    const u64 n_out = n_slots + n_over;
    std::vector<uint2> v_out;
    std::vector<u32> v_sizes;
    const uint2* ho = h_out_pin;
    const u32* hsizes = have_sizes ? db->h_sizes.data() : h_sizes_pin;
    if (rc == YH_OK && (!hp || n_over > spec)) {  // the second round
        v_out.resize(n_out);
        PW_HIP(hipMemcpyAsync(v_out.data(), d_out, n_out * sizeof(uint2), hipMemcpyDeviceToHost, st));
        if (!hsizes) {
            v_sizes.resize(N);
            PW_HIP(hipMemcpyAsync(v_sizes.data(), db->d_sizes, N * sizeof(u32), hipMemcpyDeviceToHost, st));
            hsizes = v_sizes.data();
        }
        PW_HIP(hipStreamSynchronize(st));
        ho = v_out.data();
    }
#undef PW_HIP
    yh_tfree(db, d_scr); yh_tfree(db, d_out_own); yh_tfree(db, d_ovf);
    if (rc != YH_OK) return rc;

    // segments in row order (columns ascend inside a segment, column blocks inside a row), through the exact host-side
    // filter (main.cpp:297-303): keep iff !(1.0*count/|R_i| < C)
    u64 n_surv = 0;
    for (u64 s = 0; s < nseg; ++s) n_surv += h_segcnt[s];
    u32* hi = (u32*)malloc(std::max<size_t>(n_surv, 1) * sizeof(u32));
    u32* hj = (u32*)malloc(std::max<size_t>(n_surv, 1) * sizeof(u32));
    u32* hc = (u32*)malloc(std::max<size_t>(n_surv, 1) * sizeof(u32));
    if (!hi || !hj || !hc) { free(hi); free(hj); free(hc); yh_set_error("host allocation failed"); return YH_ERR_OOM; }
    size_t w = 0;
    std::vector<u64> srt;
    for (u64 s = 0; s < nseg; ++s) {
        const u32 n = h_segcnt[s];
        if (!n) continue;
        const u32 i = (u32)(r0 + s / ncb);
        const u64 off = h_segoff[s];
        if (off + n > n_out) { free(hi); free(hj); free(hc); yh_set_error("pairwise: a segment outside the output"); return YH_ERR_HIP; }
        const size_t w_row = w;
        for (u32 e = 0; e < n; ++e) {
            const uint2 v = ho[off + e];
            const double cij = 1.0 * v.y / hsizes[i];
            if (cij < c_thresh) continue;
            hi[w] = i; hj[w] = v.x; hc[w] = v.y;
            ++w;
        }
        if (sparse && w - w_row > 1) {  // a sparse row's survivors leave in the order its table's slots were claimed: by column here
            srt.resize(w - w_row);
            for (size_t e = 0; e < srt.size(); ++e) srt[e] = ((u64)hj[w_row + e] << 32) | hc[w_row + e];
            if (!std::is_sorted(srt.begin(), srt.end())) {
                std::sort(srt.begin(), srt.end());
                for (size_t e = 0; e < srt.size(); ++e) { hj[w_row + e] = (u32)(srt[e] >> 32); hc[w_row + e] = (u32)srt[e]; }
            }
        }
    }
    db->pw_n = w;
    db->h_pw_i = hi;
    db->h_pw_j = hj;
    db->h_pw_c = hc;
    db->pw_valid = true;
    db->pw_c = c_thresh;
    db->pw_r0 = r0;
    db->pw_r1 = r1;
    return YH_OK;
}
