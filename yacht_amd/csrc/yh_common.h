// yh_common.h — internal declarations shared by the HIP translation units of libyacht_hip.so.
// Not part of the public boundary (that is include/yacht_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/yacht_hip.h"

typedef unsigned long long u64;  // same width as uint64_t; the type HIP's 64-bit atomics take
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

static_assert(sizeof(u64) == sizeof(uint64_t), "u64 must be 64 bits");

// ---- error plumbing ------------------------------------------------------------------------
void yh_set_error(const char* fmt, ...);

#define YH_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            yh_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                         __LINE__);                                                         \
            return (e__ == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;                  \
        }                                                                                   \
    } while (0)

#define YH_TRY(call)               \
    do {                           \
        int rc__ = (call);         \
        if (rc__ != YH_OK) return rc__; \
    } while (0)

// Tuning switches (YH_FILTER_BPH, YH_INDEX_TILE, YH_TIMING_EVERY, ...) are read from the environment only when
// YH_DEBUG_TUNING=1: one gate, so that a production process never changes behaviour because of a stray variable.
// Returns the variable's value, or nullptr (gate closed or variable unset).
const char* yh_tune_env(const char* name);

constexpr int STREAM_BLOCK = 1024;          // elements per block of the delta stream = 16 per lane
constexpr u32 STREAM_NONE = 0xffffffffu;

// postings per work record of the reference-major exclusive pass (k_excl_pieces: one wave per record)
#ifndef YH_EXCL_PIECE
#define YH_EXCL_PIECE 256
#endif
#ifndef YH_EXCL_PIECE_THREADS
#define YH_EXCL_PIECE_THREADS 512
#endif

constexpr int TIMING_RING = 32;  // (events are created when a ring records for the first time: a handle that is never timed pays nothing)

struct EventRing {
    hipEvent_t beg[TIMING_RING];
    hipEvent_t end[TIMING_RING];
    int head = 0;      // next slot to record into
    int pending = 0;   // slots recorded since the last read
    unsigned calls = 0;  // launches seen (the ring samples every n-th, see yh_ring_record_begin)
    bool armed = false;  // the current begin/end pair is being recorded
    bool created = false;
    bool wanted = false;  // the handle records into this ring (created on first use)
};

// One in-flight host-buffer run call (yh_run_submit / yh_run_wait): its own sample and count buffers
// in HBM, so that the upload of call k+1 and the download of call k-1 overlap the kernels of call k.
struct RunSlot {
    u64* d_sample = nullptr;
    u64 cap = 0;
    void* d_packed = nullptr; // a packed sample as uploaded (yh_run_submit_packed), expanded into d_sample
    u64 packed_cap = 0;
    uint4* d_rows = nullptr;  // [N] staging of the compact rows when the caller's row buffer is pageable
    u64 rows_cap = 0;         // capacity of the caller's row buffer of the call in flight
    bool rows_mode = false;   // the call in flight returns compact rows
    u32* d_out = nullptr;     // [3][N] overlap, n_excl, n_match
    u32* d_bad = nullptr;     // [1] set by the ordering check queued in front of the kernels
    u32* h_bad = nullptr;     // page-locked host words the kernels write through PCIe: [0] ordering verdict, [1] number of rows
    u32* h_bad_dev = nullptr; // its device address
    hipEvent_t ev_up = nullptr, ev_out = nullptr;
    bool busy = false;
};

struct yh_db {
    int device = -1;
    uint32_t flags = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the stream work is queued on (own_stream unless overridden)

    // sizes
    u64 n_refs = 0;
    u64 n_hashes = 0;
    u64 max_hash = 0;
    u64 device_bytes = 0;

    // plain CSR (only with YH_DB_KEEP_CSR)
    u64* d_values = nullptr;
    u64* d_offsets = nullptr;

    u32* d_sizes = nullptr;  // [N]   |R_j|

    // shared-hash inverted index (hashes present in >= 2 references)
    bool has_index = false;
    u64 n_distinct = 0;
    u64 n_shared = 0;    // G
    u64 n_postings = 0;  // sum of posting-list lengths
    u64* d_g = nullptr;      // [G]   shared hashes ascending
    u64* d_po = nullptr;     // [G+1] posting-list offsets
    u32* d_pr = nullptr;     // [postings] reference ids, ascending inside a list
    u32* d_pg = nullptr;     // [postings] index of the hash each posting belongs to
    u32* d_prank = nullptr;  // [postings] YH_DB_PAIRWISE_ONLY: the posting's rank among its reference's shared hashes (yh_pairwise.hip)
    u32* d_nshared = nullptr;  // [N] number of shared hashes in reference j
    // YH_DB_PAIRWISE_ONLY handles whose pairs the distribution sort took (yh_sort.hip, k_bucket_group): the sort's
    // last pass writes what the pairwise pass reads, and none of g / po / pr / pg / prank exists -- ONE 8-byte record per
    // CSR position, "the other holders of this element's hash" (0: no other holder), so a reference's records are the
    // extent of its sketch and nothing is counted, ranked or transposed (yh_pairwise.hip: k_pair_rows<.., true>)
    bool fz = false;
    u64* d_fz_rec = nullptr;   // [H] bit 63 clear: up to three holders as 21-bit fields (reference + 1); set: {holders << 40 | first entry of d_fz_list}
    u32* d_fz_list = nullptr;  // [buckets x 4096] the holders of every hash with more than four of them, at the hash's place in its bucket
    u32* d_fz_list2 = nullptr; // the holders of the hashes of SPILLED buckets (more pairs than a bucket holds: a k-mer thousands of references
    u64 fz_list_split = 0;     //   share), hash by hash; a list record whose start is >= fz_list_split names d_fz_list2[start - fz_list_split]
    u64 n_spilled_pairs = 0, n_spilled_buckets = 0;  // (yh_db_info)
    u32 sort_path = 0;         // how the pairs were put in order: YH_SORT_* (yh_db_info.sort_path)
    u64* d_fz_off = nullptr;   // [N + 1] the CSR offsets = the rows of d_fz_rec
    u32* d_fz_tab = nullptr;   // [H / 256 + 2] reference of every 256th CSR position (position -> reference look-ups)
    bool fz_nshared = false;   // d_nshared has been counted from the records (on demand: yh_db_nshared_device)

    // full distinct-hash directory (only with YH_DB_FULL_INDEX): the sample-driven overlap path
    u64* d_dh = nullptr;       // [D] every distinct hash, ascending
    u32* d_dref = nullptr;     // [D] its single holder, or 0x80000000 | index into d_g
    u32* d_dir = nullptr;      // [dir_nb + 1] first index of d_dh whose (hash >> dir_shift) >= bucket
    u32 dir_shift = 0, dir_nb = 0;
    // reference-major view of the postings, cut into chunks of <= 64 (one wave each): the
    // exclusive pass visits only the chunks of masked references instead of streaming pr[]
    u32* d_rpo = nullptr;      // [N + 1] first posting of reference r in d_rg
    u32* d_rg = nullptr;       // [n_postings] shared-hash index, grouped by reference
    // work list of the exclusive pass: (reference, first posting in d_rg, end) per piece of <= YH_EXCL_PIECE postings
    // of a reference in the subset; appended per query (k_reduce_replicas / k_excl_worklist); n_chunks = capacity
    uint4* d_work = nullptr;
    u32* d_work_count = nullptr;
    uint4* d_rrec = nullptr;   // [n_postings] beside d_rg (stream layout): the other holders of the posting's hash,
                               // {o0, o1, o2, count <= 7} (others 3..6 in d_rrecx) or, for nine holders and
                               // more, {first index in d_pr, holders, 0, ~0}
    uint4* d_rrecx = nullptr;  // [n_postings] {o3, o4, o5, o6}
    // presence filter of the distinct hashes in front of the compact buckets: bit umulhi(h << bkt_lsh, filter_mul)
    u32* d_filter = nullptr;
    u64 filter_mul = 0, filter_bits = 0;
    // the DISTINCT holder sets of every reference with their multiplicities (what the fused run step walks instead of
    // the postings): records as d_rrec / d_rrecx, reference-major, reference r owns [d_hpo[r], d_hpo[r + 1])
    uint4* d_hrec = nullptr;   // [n_sets]
    uint4* d_hrecx = nullptr;  // [n_sets]
    u32* d_hmult = nullptr;    // [n_sets] shared hashes of the reference with exactly these other holders
    u32* d_hpo = nullptr;      // [N + 1]
    u32 n_sets = 0;
    u32 n_chunks = 0;
    // hash-sorted delta stream (the default layout; DESIGN.md "K1"): every (hash, reference)
    // pair of the database in ascending hash order, the hash truncated to t = hash >> sshift so that
    // consecutive t differ by ~50 on average, one BYTE per element = t minus its predecessor's t.
    // A gap above 255 is bridged by filler elements (delta 255, no reference).  Blocks of
    // STREAM_BLOCK elements; s_hdr[b] = t of block b's first element (its own delta byte is unused).
    u8* d_sdelta = nullptr;    // [slen]
    u64* d_shdr = nullptr;     // [slen / STREAM_BLOCK + 1], last = ~0
    uint2* d_srec = nullptr;   // [slen] per position {low 32 bits of the hash, reference | 0x80000000 when the hash is a
                               // database-shared one} (fillers / padding: all ones): ONE 8-byte read per candidate
    u64 slen = 0;              // multiple of STREAM_BLOCK
    u32 sshift = 0;
    u64* d_wg_key = nullptr;   // [wgs + 1] first t of each workgroup's block range (sample-independent)
    u32 wg_key_n = 0;
    uint4* d_bkt = nullptr;    // [bkt_nb] 64-byte buckets over the distinct hashes (YhDirView below)
    u64 bkt_nb = 0;
    u32 bkt_lsh = 0;
    bool has_dir = false;      // the handle answers the sample-driven (indexed) queries
    int lookup_mode = 0;       // YH_LOOKUP_*: which lookup yh_run / yh_overlap take (yh_db_set_lookup)
    uint4* d_cbkt = nullptr;   // [cbkt_nb] COMPACT 64-byte buckets (seven 32-bit hash remainders + holders; YhDirView)
    u64 cbkt_nb = 0;
    u64 bkt_mul = 0;           // bucket(h) = umulhi(h << bkt_lsh, bkt_mul): [0, max_hash] spread over ALL the buckets of the table in use
    u64* d_ovf_keys = nullptr; // [ovf_mask + 1] open-addressing table of the hashes that did not fit their bucket
    u32* d_ovf_vals = nullptr; //                 their dref words (YH_DIR_NONE = empty slot)
    u32 ovf_mask = 0;

    // per-query scratch (allocated once)
    u8* d_mask = nullptr;      // [N]
    u32* d_maskbits = nullptr; // [ceil(N/64)*2] the same mask as bits
    u8* d_hit = nullptr;       // [G]
    bool hit_clean = false;    // d_hit is all zero once everything queued on the stream has run
    u32* d_excl_e = nullptr;   // [N] shared hashes that are subset-exclusive   } one allocation of 3N words,
    u32* d_excl_m = nullptr;   // [N] ... and in the sample                      } zeroed together
    u32* d_ovsh = nullptr;     // [N] overlap restricted to shared hashes        }
    u32* d_overlap_tmp = nullptr;  // [N]
    u32* d_out_tmp = nullptr;      // [2][N] exclusive counts of the host-pointer entry points (allocated on first use)
    u64* d_sample_tmp = nullptr;   // grows on demand (host-pointer entry points)
    u64 sample_tmp_cap = 0;
    u32* d_flag = nullptr;     // [1] generic error/flag word
    std::vector<u32> h_sizes;  // the sketch sizes on the host (YH_DB_PAIRWISE_ONLY handles: yh_pairwise's exact filter reads them)
    u32 max_ref_size = 0;      // ... and the largest of them (16-bit counts in k_pair_rows when it fits)
    u32* d_bad_word = nullptr; // [1] deferred ordering verdict of yh_run (see bad_gen)
    u32* d_reps = nullptr;     // [R][N] replicated overlap counters; ZERO AT REST: k_reduce_replicas clears what it sums
    u64 reps_cap = 0;
    // scratch of the batched run, per batch slot: the samples' subset words [N + 1] u64 and the hits on shared hashes
    // [B][N] u32 -- what the first half of a hash-range batch leaves for the second (two slots: the exchange of one
    // block's words travels while the next block's lookups run)
    struct BatchSlot {
        void* d_scratch = nullptr;
        u64 cap = 0;
        u32 n_samples = 0;
        bool open = false;       // first half queued, second not yet
        bool clobbered = false;  // a whole-batch call ran in the slot meanwhile
        bool words_valid = false;  // the slot holds the GLOBAL subset words of its last batch (yh_run_batch_rows_*)
        bool ovsh_clean = false;   // the second half left the slot's hits-on-shared-hashes rows zero (k_batch_final)
        hipEvent_t ev_first = nullptr;  // end of the slot's first half on the handle's stream (made when a finish stream is set)
    } batch[YH_BATCH_SLOTS];
    // yh_db_set_batch_finish_stream: where the second halves of the batched hash-range calls run (nullptr: on `stream`)
    hipStream_t fin_stream = nullptr;
    hipEvent_t ev_fin = nullptr;   // behind the last launch queued on fin_stream
    bool fin_pending = false;      // ... which the handle's stream has not waited for yet

    // pairwise result cache (two-call sizing)
    bool pw_valid = false;
    double pw_c = 0.0;
    u64 pw_r0 = 0, pw_r1 = 0;
    u64 pw_n = 0;
    u32* h_pw_i = nullptr;
    u32* h_pw_j = nullptr;
    u32* h_pw_c = nullptr;
    u64 pw_sparse_rows = 0, pw_dense_rows = 0;  // the last sparse row pass (k_pair_rows_sparse): rows it kept / handed back to the dense pass

    // Two step contexts for the sharded run: what yh_run_local_device leaves for yh_run_finish_device (subset bits,
    // work list).  With two of them the lookup of sample k+1 runs while the bit exchange of sample k is in flight.
    // ctx_bits[0] / ctx_work[0] / ctx_count[0] are the handle's own arrays; the others are allocated on first use.
    u32* ctx_bits[YH_RUN_CONTEXTS] = {};
    uint4* ctx_work[YH_RUN_CONTEXTS] = {};
    u32* ctx_count[YH_RUN_CONTEXTS] = {};
    int ctx_now = 0;
    bool range_local = false;  // set around the first half of a hash-range shard's step (yh_run_local_range_device)
    bool ctx_open[YH_RUN_CONTEXTS] = {};       // yh_run_local_device queued, yh_run_finish_device not yet
    bool ctx_clobbered[YH_RUN_CONTEXTS] = {};  // another query ran in the context's place meanwhile: its work list is gone

    // sharded run (yh_db_set_ghosts): references [ghost_begin, ghost_begin + n_ghost) are copies of other
    // ranks' references; their subset bits come from the owners (k_ghost_bits)
    u32* d_ghost_src = nullptr;  // [n_ghost] bit index into the all-gathered subset bits
    u64 ghost_begin = 0, n_ghost = 0;

    // pipelined device-resident steps (yh_run_device_pipelined): three stages of three consecutive samples per launch
    // (k_step_fused).  pend_red: step context of the sample whose lookup ran in the last launch (-1: none), with the
    // parity of its counter set and its three output rows; pend_excl: context of the sample reduced by the last launch
    u64 pipe_k = 0;
    int pend_red = -1, pend_red_parity = 0, pend_excl = -1;
    u32* pend_red_out[3] = {nullptr, nullptr, nullptr};
    u32* pend_excl_out = nullptr;
    int pipe_last_ctx = -1;  // step context of the LAST step queued when that was a pipelined one (its subset bits: yh_run_rows_device); -1 otherwise

    // pipelined host-buffer calls
    RunSlot slots[YH_RUN_SLOTS];
    hipStream_t st_in[2] = {nullptr, nullptr};  // upload streams beside `stream` (slots alternate: two copy engines, the fixed cost of one copy hidden behind the other)
    const u32* d_bad = nullptr;  // non-null while the kernels of a call with a deferred ordering verdict are being queued:
    u32 bad_gen = 0;             // the check kernel in front of them stores bad_gen there when the sample is not ascending
                                 // (generations are unique per handle, so the word never has to be cleared)

    // timing
    EventRing ev_overlap, ev_excl, ev_pair;
    struct yh_psort* tmp_psort = nullptr;  // build time: the distribution sort that made the sorted pairs (its buckets are walked as chunks)
    bool order_checked = false;            // build time: every sketch's ordering has been verified
    float ms_upload_kernels = 0.f;  // device time of the chunk sorts / merges that ran under the upload (yh_build_upload_sorted)
    float ms_db_build = 0.f;
    // host <-> device copies (yh_timing.ms_h2d / ms_d2h): the CSR upload of yh_db_create (host clock), then HIP events
    // around the sample upload / count download of the last synchronous host-pointer query
    float ms_h2d_create = 0.f;
    hipEvent_t ev_xfer[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [h2d | d2h][begin | end], created on first use
    bool xfer_recorded[2] = {false, false};
};

// ---- implemented in yh_build.hip -------------------------------------------------------------
int yh_build_validate(yh_db* db, const u64* d_values, const u64* d_offsets);  // ordering check, sizes, largest hash
int yh_build_validate_extents(yh_db* db, const u64* d_values, const u64* d_offsets);  // sizes + largest hash only (N reads); order: later
int yh_build_check_order(yh_db* db, const u64* d_values, const u64* d_offsets);
int yh_build_index(yh_db* db, const u64* d_values, const u64* d_offsets, u64* d_sk_pre = nullptr, u32* d_sv_pre = nullptr);
// host CSR up in chunks, each checked, sorted and merged while the next one crosses the bus (yh_build.hip)
int yh_build_upload_sorted(yh_db* db, const u64* h_values, const u64* h_offsets, u64* d_values, const u64* d_offsets,
                           u64** d_sk_out, u32** d_sv_out);

// ---- implemented in yh_pack.hip ----------------------------------------------------------------
int yh_pack_validate(const void* packed, u64 bytes, u64* n_out);
int yh_pack_expand_device(yh_db* db, const void* d_packed, u64 n, u64* d_out, u32* d_bad, u32 gen, u32* h_bad_dev);
int yh_rows_compact_device(yh_db* db, const u32* d_overlap, const u32* d_excl, const u32* d_match, void* rows_dev, u64 cap,
                           u32* count_dev, u32* count_host_dev);

// ---- implemented in yh_query.hip -------------------------------------------------------------
int yh_q_overlap(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, bool flag_shared, bool make_mask);
// phases: 1 = lookup + reduce (leaves the subset bits in d_maskbits and, optionally, d_bits_out), 2 = the
// posting-list part of n_excl (ghost bits patched from d_global_bits first), 3 = both.  1 = not applicable.
int yh_q_run_fused(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, u32* d_excl, u32* d_match,
                   int phases = 3, u32* d_bits_out = nullptr, const u32* d_global_bits = nullptr, bool use_indexed = false);
int yh_q_overlap_bsearch(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap);
bool yh_q_step_fused_ok(const yh_db* db, u64 n_sample);
int yh_q_step_fused(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, u32* d_excl, u32* d_match);
int yh_q_range_finish(yh_db* db, const u32* d_gathered, u32 n_ranks, u64 stride_words, u32* d_excl);
// ---- the distinct-hash directory as the lookup kernels see it -------------------------------------
// Primary structure: a table of 64-byte buckets, bucket(h) = floor(h * bkt_nb / 2^bits(max_hash))
// (monotone in h, ~2 distinct hashes per bucket).  A bucket is 16 words = one HBM sector:
//   w[0..9] five hashes (lo, hi) ascending, w[10..14] their dref words, w[15] = entries used, or
//   YH_BKT_OVERFLOW when more than five hashes fall into it (~1.7 % of the buckets).
// One random sector read answers a lookup; overflowing buckets fall back to the two-level
// directory (dir -> dh -> dref, three dependent reads), which is also the whole path when the
// table is absent (YH_NO_BUCKETS=1 at creation, for A/B timing).
#define YH_BKT_OVERFLOW 0xffffffffu
#define YH_DIR_NONE 0xffffffffu
#if defined(__HIPCC__)
__device__ __forceinline__ u64 yh_bucket_of(u64 h, u32 lsh, u64 nb) { return __umul64hi(h << lsh, nb); }
// The presence filter (yh_db::d_filter): the 32-bit WORD of a hash is a monotone function of it (a sorted sample walks the
// filter front to back); inside the word a hash owns TWO bits picked by a multiplicative hash (round 4; before: the one bit
// the monotone index named).  Same table, same single read per sample hash; at 4 bits per distinct hash a word holds ~8
// hashes = ~13 of 32 bits set, so an absent hash finds both of its bits set with probability ~0.16 instead of ~0.22 for one
// bit -- a fifth fewer bucket reads for nothing.  A hash that IS in the database always finds its bits: counts stay exact.
#ifndef YH_FILTER_K
#define YH_FILTER_K 2  // bits per hash inside its word (1 = round 3's filter: the bit the monotone index names; 3: tuning)
#endif
__host__ __device__ __forceinline__ u32 yh_filter_mask(u64 h, u64 bit) {
#if YH_FILTER_K >= 2
    const u32 m = (u32)((h * 0x9E3779B97F4A7C15ull) >> 49);  // 15 bits: three 5-bit positions
    (void)bit;
    u32 mask = (1u << (m & 31u)) | (1u << ((m >> 5) & 31u));
#if YH_FILTER_K >= 3
    mask |= 1u << ((m >> 10) & 31u);
#endif
    return mask;
#else
    return 1u << (u32)(bit & 31u);
#endif
}

// Compact bucket (the form every database of realistic size gets): bucket(h) as above with ~2.5 distinct
// hashes per bucket; inside a bucket the hashes span less than 2^32, so their LOW 32 BITS identify them:
//   w[0..6] low words of up to seven hashes, w[7] = entries used | YH_CBKT_OVERFLOW when more fell into it,
//   w[8..14] their dref words, w[15] spare.
// The hashes beyond the seventh (0.4 % of the buckets have any) live in a small open-addressing table
// keyed by the full hash: one more random read for the few lookups that reach it.  No other structure is
// kept: 25.6 bytes per distinct hash instead of 44.  Databases too small or too wide-ranged for 32-bit
// remainders (2^bits / buckets > 2^32) keep the five-entry full-hash buckets + directory above.
#define YH_CBKT_OVERFLOW 0x80000000u
__host__ __device__ __forceinline__ u64 yh_ovf_slot(u64 h) {
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    return h;
}
struct YhDirView {
    const u64* dh;
    const u32* dref;
    const u32* dir;
    const uint4* bkt;
    u64 bkt_nb, max_hash;
    u32 dshift, NB, bkt_lsh;
    const uint4* cbkt;
    u64 bkt_mul;
    const u64* ovf_keys;
    const u32* ovf_vals;
    u32 ovf_mask;

    __device__ __forceinline__ u32 find_slow(u64 h) const {
        const u64 b = h >> dshift;
        if (b >= NB) return YH_DIR_NONE;
        u32 i = dir[b];
        const u32 e = dir[b + 1];
        u64 v = 0;
        for (; i < e; ++i) {
            v = dh[i];
            if (v >= h) break;
        }
        return (i < e && v == h) ? dref[i] : YH_DIR_NONE;
    }
    __device__ __forceinline__ u32 find_overflow(u64 h) const {
        for (u32 s = (u32)yh_ovf_slot(h) & ovf_mask;; s = (s + 1) & ovf_mask) {
            const u32 v = ovf_vals[s];
            if (v == YH_DIR_NONE) return YH_DIR_NONE;
            if (ovf_keys[s] == h) return v;
        }
    }
    // the compact bucket of h in two steps, so that a lane can have several lookups in flight: request the
    // four 16-byte words (h <= max_hash) ...
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ void cbkt_request(u64 h, v4u& a, v4u& b, v4u& c, v4u& d) const {
        const v4u* p = reinterpret_cast<const v4u*>(cbkt) + 4 * yh_bucket_of(h, bkt_lsh, bkt_mul);
        // (non-temporal loads here -- read-once lines kept out of the Infinity Cache in favour of the presence filter --
        // measured 3-10 us SLOWER per launch: profiles/r03/filter_sweep.txt)
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
    }
    // the same with non-temporal loads: the bucket is read once and should not push re-used lines (the presence filter of a
    // batched pass) out of the XCD's L2
    __device__ __forceinline__ void cbkt_request_nt(u64 h, v4u& a, v4u& b, v4u& c, v4u& d) const {
        const v4u* p = reinterpret_cast<const v4u*>(cbkt) + 4 * yh_bucket_of(h, bkt_lsh, bkt_mul);
        a = __builtin_nontemporal_load(p);
        b = __builtin_nontemporal_load(p + 1);
        c = __builtin_nontemporal_load(p + 2);
        d = __builtin_nontemporal_load(p + 3);
    }
    // ... and look at them
    __device__ __forceinline__ u32 cbkt_resolve(u64 h, const v4u a, const v4u b, const v4u c, const v4u d) const {
        const u32 lo = (u32)h, n = b.w & 0xfu;
        u32 r = YH_DIR_NONE;
        if (n > 0 && a.x == lo) r = c.x;
        if (n > 1 && a.y == lo) r = c.y;
        if (n > 2 && a.z == lo) r = c.z;
        if (n > 3 && a.w == lo) r = c.w;
        if (n > 4 && b.x == lo) r = d.x;
        if (n > 5 && b.y == lo) r = d.y;
        if (n > 6 && b.z == lo) r = d.z;
        if (r == YH_DIR_NONE && (b.w & YH_CBKT_OVERFLOW)) r = find_overflow(h);
        return r;
    }
    // dref word of h (holder id, or 0x80000000 | shared-hash index), YH_DIR_NONE when h is not in the database
    __device__ __forceinline__ u32 find(u64 h) const {
        if (h > max_hash) return YH_DIR_NONE;
        if (cbkt) {
            v4u a, b, c, d;
            cbkt_request(h, a, b, c, d);
            // All four 16-byte loads of the bucket, unconditionally, before anything looks at them: left to
            // itself the compiler sinks the loads of w[0] / w[8] under "entries > 0", which it only knows
            // after the first loads have come back -- a second dependent memory round trip per lookup
            // (k_index_lookup: 55 us instead of 30 for 10^6 lookups).
            asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            return cbkt_resolve(h, a, b, c, d);
        }
        if (!bkt) return find_slow(h);
        const v4u* p = reinterpret_cast<const v4u*>(bkt) + 4 * yh_bucket_of(h, bkt_lsh, bkt_mul);
        v4u a = p[0], b = p[1], c = p[2], d = p[3];
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));  // (see above)
        if (d.w == YH_BKT_OVERFLOW) return find_slow(h);
        const u32 lo = (u32)h, hi = (u32)(h >> 32);
        u32 r = YH_DIR_NONE;
        if (d.w > 0 && a.x == lo && a.y == hi) r = c.z;
        if (d.w > 1 && a.z == lo && a.w == hi) r = c.w;
        if (d.w > 2 && b.x == lo && b.y == hi) r = d.x;
        if (d.w > 3 && b.z == lo && b.w == hi) r = d.y;
        if (d.w > 4 && c.x == lo && c.y == hi) r = d.z;
        return r;
    }
};
inline YhDirView yh_dir_view(const yh_db* db) {
    return YhDirView{db->d_dh, db->d_dref, db->d_dir, db->d_bkt, db->bkt_nb, db->max_hash, db->dir_shift, db->dir_nb, db->bkt_lsh,
                     db->d_cbkt, db->bkt_mul, db->d_ovf_keys, db->d_ovf_vals, db->ovf_mask};
}
#endif

// returns 2 when it did the exclusive counts too (lookup_half_only: everything but the posting-list pass)
int yh_q_overlap_indexed(yh_db* db, const u64* d_sample, u64 n_sample, u32* d_overlap, bool for_exclusive,
                         u32* d_fused_excl = nullptr, u32* d_fused_match = nullptr, u32* d_bits_out = nullptr,
                         bool lookup_half_only = false);
int yh_q_run_batch(yh_db* db, const u64* d_samples, const u64* d_soff, u32 n_samples, u64 total_hashes,
                   u32* d_overlap, u32* d_excl, u32* d_match, int phases = 3, u64* d_maskword_out = nullptr,
                   const u64* d_gathered = nullptr, u32 n_ranks = 0, int slot = 0);
// the compact form of a batch's result (yh_batch.hip): pack = this rank's value triples, unpack = the rows themselves
int yh_q_batch_rows_pack(yh_db* db, int slot, const u32* d_overlap, const u32* d_excl, const u32* d_match, u32* d_vals, u64 cap_rows,
                         u32* d_n_rows);
int yh_q_batch_rows_unpack(yh_db* db, int slot, const u32* d_vals, u64 cap_rows, void* d_rows, u32* d_n_rows);
// the subset words of a block as (word, reference) entries: what a hash-range rank all-gathers instead of its dense row
inline u64 yh_batch_words_packed_len(u64 cap) { return 1 + cap + (cap + 1) / 2; }
int yh_q_batch_words_pack(yh_db* db, const u64* d_words, u32 n_planes, u64* d_packed, u64 cap);
int yh_q_batch_words_unpack(yh_db* db, const u64* d_gathered, u32 n_ranks, u32 n_planes, u64 cap, u64* d_words_out, u32* d_overflow);
int yh_q_exclusive(yh_db* db, const u8* d_mask, const u64* d_sample, u64 n_sample,
                   const u32* d_overlap, u32* d_excl, u32* d_match, const u32* d_maskbits);
int yh_q_pairwise(yh_db* db, double c_thresh, u64 r0, u64 r1);
int yh_q_fz_nshared(yh_db* db);  // (yh_pairwise.hip) d_nshared of a fused handle, counted on first use
// the presence filter a lookup may read in front of the compact buckets (null: none, or YH_NO_FILTER=1)
inline const u32* yh_filter_of(const yh_db* db) {
    static const bool filter_off = [] { const char* e = yh_tune_env("YH_NO_FILTER"); return e && e[0] == '1'; }();
    return (db->d_cbkt && db->d_filter && db->filter_mul && !filter_off) ? db->d_filter : nullptr;
}
int yh_q_check_sorted_host(const u64* v, u64 n);

// helpers (yh_api.hip)
int yh_dmalloc(yh_db* db, void** p, size_t bytes);
// Temporaries of a build or a pairwise pass (and, through yh_dmalloc, the arrays of a YH_DB_PAIRWISE_ONLY handle) come from
// a per-process cache of device buffers (yh_api.hip); yh_dfree / yh_tfree release either kind of block.
void yh_dfree(yh_db* db, void* p);
hipError_t yh_tmalloc(yh_db* db, void** p, size_t bytes);
void yh_tfree(yh_db* db, void* p);
void yh_pool_trim(yh_db* db);
// A page-locked host staging buffer of >= `bytes` from a small process-wide cache (a read-back into pageable memory is a blocking,
// staged copy of 20-40 us whatever its size; into page-locked memory it is queued in ~5 us and several wait for ONE synchronize).
// p == nullptr when it cannot be had (more than 64 MB, four already in use, no memory): the caller reads into pageable memory.
void* yh_pin_acquire(u64 bytes);
void yh_pin_release(void* p);
struct YhPin {
    void* p;
    explicit YhPin(u64 bytes) : p(yh_pin_acquire(bytes)) {}
    ~YhPin() { if (p) yh_pin_release(p); }
    YhPin(const YhPin&) = delete;
    YhPin& operator=(const YhPin&) = delete;
};
void yh_ring_record_begin(yh_db* db, EventRing& r, hipStream_t st = nullptr);  // (st: nullptr = the handle's stream)
void yh_ring_record_end(yh_db* db, EventRing& r, hipStream_t st = nullptr);
