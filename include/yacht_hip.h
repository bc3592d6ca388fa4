/*
 * yacht_hip.h — C ABI of libyacht_hip.so, the MI355X (gfx950) containment engine for YACHT.
 *
 * This is the drop-in boundary for YACHT's one data-parallel hot path: sorted-uint64
 * FracMinHash set intersection.  The reference has no in-process FFI on this path; it
 * crosses PROCESS boundaries with files.  Each entry point below names the
 * reference interface it replaces (paths relative to the YACHT repo, v1.4.0):
 *
 *   yh_db_create        replaces  read_sketches + compute_index_from_sketches
 *                                 (src/cpp/main.cpp:105-124, :215-246) and the per-run re-reading
 *                                 of every reference .sig (src/yacht/hypothesis_recovery_src.py:93,154,168)
 *   yh_overlap          replaces  `sourmash scripts multisearch` as used by
 *                                 get_organisms_with_nonzero_overlap (hypothesis_recovery_src.py:93-113)
 *   yh_exclusive        replaces  the set loops of get_exclusive_hashes (hypothesis_recovery_src.py:165-204)
 *   yh_pairwise         replaces  compute_intersection_matrix(_by_sketches) (src/cpp/main.cpp:249-366)
 *   yh_index_stats      replaces  the three statistics printed at src/cpp/main.cpp:242-244
 *   yh_train_select     replaces  do_yacht_train (src/cpp/main.cpp:371-407)
 *   yh_hyp_test         replaces  single_hyp_test / get_alt_mut_rate (hypothesis_recovery_src.py:209-306; scipy there)
 *
 * Conventions
 *   - every function returns 0 on success, a negative YH_ERR_* code on failure;
 *     yh_last_error() returns a thread-local, library-owned message for the last failure;
 *   - the caller allocates and owns every buffer it passes; the library never frees caller memory;
 *   - handles are created/destroyed in pairs; one handle must not be used from two threads at
 *     once, distinct handles are independent;
 *   - pointer parameters named d_* are DEVICE pointers (HBM of the handle's device); all other
 *     pointers are host pointers;
 *   - there is NO CPU fallback: every compute entry fails with YH_ERR_NO_DEVICE when no gfx950
 *     device is usable.  The CPU restatement of the algorithm lives in oracle/ and is test
 *     infrastructure only.
 *
 * Data model
 *   A reference database is N sketches in CSR form: `values` holds every reference's hashes
 *   back to back, each reference's slice STRICTLY ASCENDING (sourmash "mins" order), and
 *   `offsets[N+1]` delimits the slices.  A sample sketch is one strictly ascending uint64 array.
 */
#ifndef YACHT_HIP_H
#define YACHT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YH_ABI_VERSION 8

enum {
    YH_OK               = 0,
    YH_ERR_INVALID_ARG  = -1,  /* null pointer, bad size, bad handle                          */
    YH_ERR_NO_DEVICE    = -2,  /* no usable HIP device / device id out of range               */
    YH_ERR_HIP          = -3,  /* a HIP runtime call failed (message has the HIP error string) */
    YH_ERR_UNSORTED     = -4,  /* a sketch is not strictly ascending                          */
    YH_ERR_CAPACITY     = -5,  /* caller buffer too small (two-call sizing: see yh_pairwise)  */
    YH_ERR_OOM          = -6,  /* device or host allocation failed                            */
    YH_ERR_UNSUPPORTED  = -7   /* e.g. index not built for this handle                        */
};

/* yh_db_create flags */
#define YH_DB_DEFAULT      0u
#define YH_DB_NO_INDEX     1u  /* skip the shared-hash inverted index (overlap-only handle)     */
#define YH_DB_KEEP_CSR     2u  /* keep the plain CSR resident too (needed by yh_overlap_bsearch) */
#define YH_DB_PAIRWISE_ONLY 8u /* `yacht train` handle: validated sizes + what yh_pairwise reads, nothing else
                                  (yh_pairwise, yh_index_stats, yh_db_nshared_device); no lookup structures, so
                                  the overlap / exclusive / run queries return YH_ERR_UNSUPPORTED.  For uniform
                                  hashes (FracMinHash) that is one 8-byte record per CSR position, written by the
                                  last pass of the sort itself; else the inverted index                          */
#define YH_DB_NO_DIRECTORY 16u /* do not build the bucket table over the distinct hashes (25.6 B per
                                  distinct hash): the sample-driven (indexed) lookups -- yh_*_indexed_device,
                                  yh_run_batch, and the automatic choice inside yh_run / yh_overlap for
                                  samples much smaller than the database -- are then unavailable and every
                                  query streams the database                                          */

typedef struct yh_db yh_db;

typedef struct yh_db_info {
    uint64_t n_refs;             /* N                                                         */
    uint64_t n_hashes;           /* H = offsets[N]                                            */
    uint64_t max_hash;           /* largest hash in the database (0 if H == 0)                */
    uint64_t n_distinct;         /* distinct hashes over all references (index built only)    */
    uint64_t n_shared_distinct;  /* distinct hashes present in >= 2 references ("index size") */
    uint64_t n_shared_postings;  /* sum over shared hashes of their reference counts          */
    uint64_t device_bytes;       /* HBM held by the handle                                    */
    int32_t  device_id;
    uint32_t flags;
    uint32_t stream_layout;      /* what the streaming overlap kernel reads: YH_STREAM_*      */
    uint32_t stream_shift;       /* delta stream: stream key = hash >> stream_shift           */
    uint64_t stream_bytes;       /* bytes of that array = HBM bytes one query has to stream   */
    uint64_t n_holder_sets;      /* distinct (reference, set of other holders) records the run step's
                                    exclusive pass walks instead of the n_shared_postings postings  */
    uint64_t filter_bytes;       /* presence filter in front of the bucket table (0: none)    */
    uint32_t sort_path;          /* how the (hash, reference) pairs were put in order: YH_SORT_* (ABI 5)                 */
    uint32_t reserved_;
    uint64_t n_spilled_buckets;  /* buckets of the distribution that held more pairs than their capacity -- a k-mer that  */
    uint64_t n_spilled_pairs;    /* thousands of references share -- and were grouped on the side; their pairs            */
} yh_db_info;

/* yh_db_info.sort_path */
#define YH_SORT_NONE      0u  /* no index (YH_DB_NO_INDEX) or an empty database                                              */
#define YH_SORT_RADIX     1u  /* rocPRIM's radix sort of all pairs: keys the distribution could not take                       */
#define YH_SORT_TWO_LEVEL 2u  /* the hand-written two-level distribution sort (yh_sort.hip: k_part, k_bucket_sort / _group)   */
#define YH_SORT_PIECES    3u  /* `yacht train`'s handle: regions read in place as pieces of the ascending sketches            */

/* yh_db_info.stream_layout */
#define YH_STREAM_NONE   0u  /* posting-only / pairwise-only handle                              */
#define YH_STREAM_DELTA  1u  /* default: all (hash, reference) pairs in hash order, one delta BYTE
                                per pair (+ an 8-byte header per 1024) -- k_stream_lookup            */

typedef struct yh_timing {
    float ms_overlap_kernel;     /* lookup kernel (HIP events on the handle's stream; mean over the sampled launches)   */
    float ms_exclusive_kernels;  /* reducer / work list / exclusive-pass kernels behind it                               */
    float ms_pairwise_kernels;   /* yh_pairwise: transpose + row kernels                                                 */
    float ms_db_build;           /* yh_db_create: validation, sort, index, tables (device time, uploads excluded)        */
    float ms_h2d;                /* host -> device copies: the CSR upload of yh_db_create until the first host-pointer
                                    query, then the sample upload of the last synchronous host-pointer query (yh_overlap,
                                    yh_exclusive, yh_run, yh_run_batch)                                                   */
    float ms_d2h;                /* device -> host copies of that query's count rows (0 until one has run)              */
} yh_timing;

/* ---- library / device ---------------------------------------------------------------- */
const char* yh_last_error(void);
int yh_abi_version(void);
int yh_device_count(int* n_devices);
/* Process-wide counters of the device buffer cache behind every handle: how often the library had to go to the
 * driver for memory (hipMalloc), the host milliseconds spent inside those calls (on this pool a multi-GB hipMalloc
 * now and then takes seconds: profiles/r04/malloc_probe.txt), and the bytes the cache holds idle now.  Any pointer
 * may be NULL.  No GPU needed. */
int yh_alloc_stats(uint64_t* n_driver_allocs, double* ms_in_driver, uint64_t* bytes_idle);
/* The cache keeps idle blocks up to a sixth of the device's memory, and never more than half of (free + cached) -- or
 * YH_POOL_KEEP=<bytes> from the environment -- trimmed at the end of every create / destroy.  yh_pool_release gives ALL
 * idle blocks back to the driver now (for a caller that is about to need the memory elsewhere: torch, RCCL, another
 * process; e.g. on an out-of-memory error of its own); *bytes_released may be NULL.  Handles in use are not touched.   */
int yh_pool_release(uint64_t* bytes_released);

/* ---- database handle ------------------------------------------------------------------ */
/* Upload a CSR reference database to `device_id`, validate ordering, and build what the queries read: the
 * hash-sorted delta stream, the bucket table + presence filter over the distinct hashes and (unless
 * YH_DB_NO_INDEX) the shared-hash inverted index. */
int yh_db_create(const uint64_t* values, const uint64_t* offsets, uint64_t n_refs,
                 int device_id, uint32_t flags, yh_db** out);
/* Same, but `d_values`/`d_offsets` already live in the HBM of `device_id` (not modified,
 * not retained after return unless YH_DB_KEEP_CSR is set, in which case they are COPIED). */
int yh_db_create_device(const uint64_t* d_values, const uint64_t* d_offsets, uint64_t n_refs,
                        int device_id, uint32_t flags, yh_db** out);
/* The same from a PACKED database (ABI 7): `packed` = what yh_csr_pack made of a host CSR -- every sketch in blocks of 256
 * hashes, a block = its first hash + the gaps at the width of its widest gap; ~5.7 bytes per hash for sketches of ~5 000 at
 * scaled = 1000 -- so that 0.7 of the bytes cross the bus; the blocks are expanded in HBM, chunk by chunk under the upload, in
 * front of the same ordering check every database goes through.  `packed` must be 8-byte aligned and is not retained.
 *   yh_csr_pack_bound   bytes that always suffice for n_hashes hashes in n_refs sketches
 *   yh_csr_pack         values / offsets as yh_db_create takes them -> packed (two-call sizing: packed = NULL, cap_bytes = 0);
 *                       YH_ERR_UNSORTED for a sketch that is not strictly ascending; threads <= 0: up to 16 host threads
 *   yh_csr_unpack       the inverse, on the host (two-call sizing: all outputs NULL / 0 -> *n_hashes, *n_refs)
 * (Where it is used: a database kept packed in host memory or on disk; `yacht train` parses .sig files into it.  No reference
 * counterpart: main.cpp:62-124 reads JSON into vectors.)                                                                */
uint64_t yh_csr_pack_bound(uint64_t n_hashes, uint64_t n_refs);
int yh_csr_pack(const uint64_t* values, const uint64_t* offsets, uint64_t n_refs, void* packed, uint64_t cap_bytes,
                uint64_t* packed_bytes, int threads);
int yh_csr_unpack(const void* packed, uint64_t packed_bytes, uint64_t* values_out, uint64_t cap_hashes, uint64_t* offsets_out,
                  uint64_t cap_refs, uint64_t* n_hashes, uint64_t* n_refs);
/* (ABI 8) rows[0 .. n_rows) of a packed CSR, in that order, as a packed CSR of their own (block entries re-based, payload words
 * copied as they are: no hash is decoded but each row's last block, for the largest hash).  Two-call sizing as yh_csr_pack.
 * (What the reference does with the selection: src/yacht/make_training_data_from_sketches.py:107-155 keeps the selected
 * signatures' files for `yacht run`, which re-opens them -- hypothesis_recovery_src.py:93,154,168.)                       */
int yh_csr_subset(const void* packed, uint64_t packed_bytes, const uint64_t* rows, uint64_t n_rows, void* out, uint64_t cap_bytes,
                  uint64_t* out_bytes);
int yh_db_create_packed(const void* packed, uint64_t packed_bytes, int device_id, uint32_t flags, yh_db** out);
int yh_db_destroy(yh_db* db);
int yh_db_get_info(yh_db* db, yh_db_info* info);
/* Run all of the handle's work on this hipStream_t (NULL = the library's own stream).
 * The library's own stream is a blocking stream: device buffers the caller fills on the legacy default
 * stream (what torch uses unless told otherwise) are complete before the handle's kernels read them, and
 * a copy queued there after a *_device call sees its results.  A stream given here is the caller's: work
 * on OTHER streams that writes the inputs or reads the outputs is the caller's to order.               */
int yh_db_set_stream(yh_db* db, void* hip_stream);
/* Block until everything queued on the handle's stream has finished.                       */
int yh_db_synchronize(yh_db* db);
/* Mean kernel durations (HIP events on the handle's stream) over the launches recorded since the
 * previous call; synchronizes the stream.  Event records are barrier packets inside the step, so
 * only every YH_TIMING_EVERY-th launch (environment, default 32; 1 = all, 0 = none) and the first one after
 * each yh_db_get_timing is recorded: an event pair costs the stream about 5 us. */
int yh_db_get_timing(yh_db* db, yh_timing* t);

/* Which lookup kernel yh_overlap / yh_run (host and device forms) use.  Both are exact.
 *   YH_LOOKUP_AUTO     (default) by cost: the sample-driven kernel (one 64-byte bucket read per sample
 *                      hash) when the sample is small against the database, else the streaming kernel
 *                      (every reference hash, one delta byte each)
 *   YH_LOOKUP_STREAM   always stream;   YH_LOOKUP_INDEXED  always sample-driven (needs the directory) */
#define YH_LOOKUP_AUTO    0
#define YH_LOOKUP_STREAM  1
#define YH_LOOKUP_INDEXED 2
int yh_db_set_lookup(yh_db* db, int mode);
/* What a query with a sample of n_sample hashes would take now: YH_LOOKUP_STREAM or YH_LOOKUP_INDEXED (< 0: error). */
int yh_db_lookup_choice(yh_db* db, uint64_t n_sample);

/* ---- yacht run, step 1: overlap of one sample with every reference ----------------------
 * overlap[j] = |S ∩ R_j| for j in [0, N).  Host-pointer form is synchronous and validates
 * that the sample is strictly ascending.                                                    */
int yh_overlap(yh_db* db, const uint64_t* sample, uint64_t n_sample, uint32_t* overlap);
/* Device-pointer form: enqueues on the handle's stream and returns (no host sync).
 * d_overlap[N] is overwritten.  The sample must be strictly ascending (not checked).       */
int yh_overlap_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap);
/* Same result through the plain one-wave-per-reference binary-search kernel (needs
 * YH_DB_KEEP_CSR).  Kept as an independent on-device cross-check and A/B baseline.        */
int yh_overlap_bsearch(yh_db* db, const uint64_t* sample, uint64_t n_sample, uint32_t* overlap);
int yh_overlap_bsearch_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap);

/* Sample-driven forms, whatever yh_db_set_lookup says (fail on a YH_DB_NO_DIRECTORY handle): one lane per
 * SAMPLE hash looks it up in the bucket table over the database's distinct hashes, so the work is
 * proportional to |S| instead of streaming every reference hash.  Same results as the streaming kernel. */
int yh_overlap_indexed_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap);
int yh_run_indexed_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample,
                          uint32_t* d_overlap, uint32_t* d_n_excl, uint32_t* d_n_match);

/* Many samples against one resident database in one pass (needs
 * the directory; the reference runs one sample per process, run_YACHT.py:150).  `samples` holds
 * n_samples (1..YH_BATCH_MAX_SAMPLES) sketches back to back, each strictly ascending, delimited by
 * sample_offsets[n_samples + 1]; outputs are [n_samples][N] row-major.  For every sample the three
 * rows equal what yh_run returns for it alone.  total_hashes = sample_offsets[n_samples].
 * (ABI 8: 256 samples per call, 64 until ABI 7.  Per-sample state travels as 64-bit SUBSET WORDS in planes of 64
 * samples: plane w = samples 64 w .. 64 w + 63, word [w * N + r], bit s - 64 w = sample s overlaps reference r;
 * YH_BATCH_PLANES(n_samples) planes.  Up to 64 samples: the one array of N words of the earlier ABIs.)          */
#define YH_BATCH_MAX_SAMPLES 256
#define YH_BATCH_PLANES(n_samples) (((n_samples) + 63u) / 64u)
int yh_run_batch(yh_db* db, const uint64_t* samples, const uint64_t* sample_offsets, uint32_t n_samples,
                 uint32_t* overlap, uint32_t* n_excl, uint32_t* n_match);
int yh_run_batch_device(yh_db* db, const uint64_t* d_samples, const uint64_t* d_sample_offsets, uint32_t n_samples,
                        uint64_t total_hashes, uint32_t* d_overlap, uint32_t* d_n_excl, uint32_t* d_n_match);

/* ---- yacht run, step 2: exclusive hashes relative to a subset ----------------------------
 * For every j with subset_mask[j] != 0:
 *   n_excl[j]  = |{h in R_j : h is in no other masked reference}|
 *   n_match[j] = |{h in R_j : same condition, and h in S}|
 * and 0 for unmasked j.  Needs the index.                                                   */
int yh_exclusive(yh_db* db, const uint8_t* subset_mask, const uint64_t* sample, uint64_t n_sample,
                 uint32_t* n_excl, uint32_t* n_match);
/* Host-pointer form of the fused `yacht run` counts below: overlap, mask = (overlap > 0),
 * exclusive counts relative to that mask.  Synchronous; validates the sample ordering.       */
int yh_run(yh_db* db, const uint64_t* sample, uint64_t n_sample,
           uint32_t* overlap, uint32_t* n_excl, uint32_t* n_match);
/* Fused device-side `yacht run` counts: overlap, mask = (overlap > 0), exclusive counts.
 * All outputs are device arrays of N uint32 (d_n_excl/d_n_match may be NULL to stop after
 * the overlap).  Enqueues on the handle's stream, no host sync.                             */
int yh_run_device(yh_db* db, const uint64_t* d_sample, uint64_t n_sample,
                  uint32_t* d_overlap, uint32_t* d_n_excl, uint32_t* d_n_match);

/* Throughput form for a caller with many samples already in HBM: every call is ONE kernel launch that looks up this
 * sample, reduces the previous call's sample and runs the exclusive pass of the one before (three independent roles of
 * one grid: consecutive calls alternate between two sets of counters and rotate through the step contexts 0..2), so the
 * step's tail -- ~11 us of launch and round-trip latency for almost no work when queued behind the lookup -- costs
 * nothing.  The three output rows of a call are complete, in the order of the handle's stream, after TWO further
 * pipelined calls or after yh_run_device_join (one or two draining launches; the host does not block), which every
 * other query entry point, yh_db_synchronize and yh_db_set_stream perform first by themselves.  A call's output buffers
 * must stay untouched until then (three buffers that rotate are enough).  Where the split does not apply (streaming
 * lookup chosen, ghosts, a handle without holder sets) the call runs as yh_run_device does.                     */
int yh_run_device_pipelined(yh_db* db, const uint64_t* d_sample, uint64_t n_sample,
                            uint32_t* d_overlap, uint32_t* d_n_excl, uint32_t* d_n_match);
int yh_run_device_join(yh_db* db);

/* ---- references spread over several GPUs, the `yacht run` subset (overlap > 0) ------------------------
 * North star: "references shard across the GPUs with only a final gather of per-reference counts".
 * Overlap is rank-local.  Exclusivity is not -- the other holder of a hash may live on another rank --
 * so every rank's handle is built over its own references PLUS ghosts: for every hash one of its
 * references shares with a reference of another rank, that foreign reference appears as an extra
 * reference holding (only) such hashes, behind the local ones, from a multiple of 64 on (pad with empty
 * references).  The handle's index then knows every holder of every local hash, and a step needs one
 * exchange: which references overlap the sample AT ALL, one bit each (N/8 bytes, latency-bound):
 *   yh_run_local_device   lookup + reduce on this rank: d_overlap, d_n_match final; d_n_excl = the part
 *                         that needs no posting list; the subset bits of the handle's references to
 *                         d_bits_out: ceil(n_refs / 256) * 8 words are written (whole 256-reference
 *                         blocks; bits behind n_refs are 0; the ghosts' bits there are NOT valid)
 *   -- all-gather of the local references' bits (torch.distributed / RCCL, yacht_amd/dist.py) --
 *   yh_run_finish_device  ghost g takes bit ghost_src[g] of d_global_bits; then the posting-list part
 *                         of d_n_excl is added.  Rows of ghosts and padding in the outputs are garbage.
 * yh_db_set_ghosts registers the ghost range and their bit positions once per handle.
 * A step runs in one of YH_RUN_CONTEXTS contexts (its own subset bits and work list): the local halves of a
 * whole block of samples can be queued, their bits exchanged in ONE collective, and the second halves follow
 * (d_global_bits then points at that sample's row of the gathered block; ghost_src indexes from there).
 * Between the two halves of a context no OTHER query may run on the handle in that context's place: a
 * yh_overlap* / yh_exclusive / yh_run* call re-uses the current context's work list (the library returns
 * YH_ERR_INVALID_ARG from yh_run_finish_device when it sees that this happened).
 *
 * What may run on a handle between the two halves of an OPEN step context or batch slot (one handle, one thread at a time):
 *
 *   entry point                                            | an open step context (ctx)            | an open batch slot
 *   -------------------------------------------------------+---------------------------------------+--------------------------------
 *   yh_run_local*_device / yh_run_finish*_device, OTHER ctx | yes: contexts are independent         | yes
 *   yh_run_batch_local_range / _finish_range, OTHER slot    | clobbers the CURRENT context (*)      | yes: slots are independent
 *   yh_run_batch_rows_pack / _unpack_device                 | yes                                   | yes (they read a COMPLETED slot)
 *   yh_overlap* / yh_exclusive / yh_run / yh_run_device /    | clobbers the CURRENT context (*): the  | yes, except yh_run_batch /
 *     yh_run_indexed_device / yh_run_submit* / yh_run_batch* |   finish half of that context then     |   yh_run_batch_device, which run in
 *                                                            |   returns YH_ERR_INVALID_ARG           |   slot 0 and clobber a first half there
 *   yh_run_device_pipelined                                  | clobbers contexts 0..2 (it rotates     | yes
 *                                                            |   through them); others untouched      |
 *   yh_run_rows_device, yh_run_device_join,                  | yes (they only complete pending        | yes
 *     yh_db_synchronize, yh_db_get_timing, yh_db_get_info    |   pipelined stages)                    |
 *   yh_pairwise, yh_index_stats, yh_db_nshared_device        | yes                                   | yes
 *   yh_db_set_stream                                         | yes (drains the old stream first)      | yes
 *   (*) the CURRENT context = the one named by the last yh_run_local*_device / yh_run_finish*_device call (0 at start).
 * Every query entry point first completes the pending stages of pipelined steps (yh_run_device_join) by itself.          */
#define YH_RUN_CONTEXTS 16
int yh_db_set_ghosts(yh_db* db, uint64_t ghost_begin, uint64_t n_ghost, const uint32_t* d_ghost_src);
/* `ctx` (0 .. YH_RUN_CONTEXTS - 1) names the step context the two halves share: the lookup of sample k+1
 * (another context) can be queued while the exchange of sample k is still in flight.                    */
int yh_run_local_device(yh_db* db, int ctx, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap,
                        uint32_t* d_n_excl, uint32_t* d_n_match, uint32_t* d_bits_out);
int yh_run_finish_device(yh_db* db, int ctx, const uint32_t* d_global_bits, uint32_t* d_n_excl);

/* ---- the run step over several GPUs by HASH RANGE (option B) ----------------------------------------
 * Rank g's handle is built over ALL N references, each cut down to its hashes in [lo_g, hi_g) (a contiguous piece of
 * every sorted sketch; yacht_amd/dist.py: HashRangeRefDB), and looks up only the sample's hashes in that range -- so
 * both the table a lookup reads and the number of lookups shrink with the number of ranks, which sharding by
 * reference does not give (there every rank looks up the whole sample).  A hash's holders all sit on one rank:
 * multiplicity, sharedness and exclusivity are rank-local -- no ghosts -- and every count is a SUM over the ranks
 * once the subset is the global one, "some rank saw an overlap":
 *   yh_run_local_range_device   lookup + reduce of the sample's hashes in this range: d_overlap and d_n_match hold
 *                               this rank's share; its subset bits go to d_bits_out (ceil(n_refs / 256) * 8 words)
 *   -- all-gather of the ranks' bits (torch.distributed / RCCL) --
 *   yh_run_finish_range_device  global subset = OR of the n_ranks gathered rows (row r starts r * stride_words words
 *                               behind d_gathered_bits); d_n_excl = this rank's share of n_exclusive for it
 *   -- sum of the three rows over the ranks (one reduce per block of samples) --
 * Same step contexts and the same rule about other queries in between as the reference-sharded pair above.        */
int yh_run_local_range_device(yh_db* db, int ctx, const uint64_t* d_sample, uint64_t n_sample, uint32_t* d_overlap,
                              uint32_t* d_n_match, uint32_t* d_bits_out);
int yh_run_finish_range_device(yh_db* db, int ctx, const uint32_t* d_gathered_bits, uint32_t n_ranks, uint64_t stride_words,
                               uint32_t* d_n_excl);

/* The same for MANY samples per call (yh_run_batch on a hash-range shard): a rank's share of one sample is only
 * |S| / n_ranks lookups -- a launch of that size is bound by launch and round-trip latencies, not by the lookups -- so the
 * throughput form takes up to YH_BATCH_MAX_SAMPLES samples' slices at once, exchanges their subset words (one uint64 per
 * reference and plane of 64 samples: bit s = sample s of the plane overlaps it) in ONE all-gather, and leaves one
 * [n_samples, N] share per count row to be summed over the ranks (P = YH_BATCH_PLANES(n_samples)):
 *   yh_run_batch_local_range_device    lookups of the concatenated slices (d_sample_offsets[n_samples + 1] delimits them);
 *                                      d_overlap [n_samples][N] = this rank's share; d_maskwords_out [P][N] its subset words
 *   yh_run_batch_finish_range_device   subset = OR of the n_ranks gathered word arrays ([n_ranks][P][N]); d_n_excl, d_n_match
 *                                      [n_samples][N] = this rank's shares (d_overlap: what the first half left)
 * The two halves of a batch share one of YH_BATCH_SLOTS batch slots of the handle (the first half's hits on shared hashes
 * wait there for the second): with several slots the subset words of block j travel while the lookups of block j + 1 run,
 * and block j - 1's compact rows are still being read (yacht_amd/dist.py: BatchRowsReducer keeps three blocks in flight).  A
 * second half without its first half in the slot, or with a different n_samples, or after yh_run_batch / yh_run_batch_device
 * ran meanwhile (they use slot 0), fails with YH_ERR_INVALID_ARG.                                                        */
#define YH_BATCH_SLOTS 3
int yh_run_batch_local_range_device(yh_db* db, int slot, const uint64_t* d_samples, const uint64_t* d_sample_offsets,
                                    uint32_t n_samples, uint64_t total_hashes, uint32_t* d_overlap, uint64_t* d_maskwords_out);
int yh_run_batch_finish_range_device(yh_db* db, int slot, uint32_t n_samples, const uint64_t* d_gathered_maskwords,
                                     uint32_t n_ranks, const uint32_t* d_overlap, uint32_t* d_n_excl, uint32_t* d_n_match);
/* A second stream for the second halves (ABI 6).  A block's second half is a dozen small launches (OR of the words, work
 * list, exclusive pass, the dense final pass, the compact rows: ~100 us of launch floors at rs214 scale, whatever the
 * rank's share of the lookups); on the handle's one stream they stand between the lookups of block j and those of block
 * j + 1.  With a finish stream set, yh_run_batch_finish_range_device, yh_run_batch_words_pack_device / _unpack_device and
 * yh_run_batch_rows_pack_device are enqueued THERE and run beside the next block's first half -- and so is the tail of
 * yh_run_batch_local_range_device: d_maskwords_out is written on the FINISH stream (behind the lookups, which stay on the
 * handle's), so that the handle's stream carries nothing but clears and lookups:
 *   - the library orders a slot's second half behind its own first half (an event per slot), and every other query entry
 *     point, yh_db_synchronize and yh_db_destroy behind the last second half queued;
 *   - the CALLER issues what it adds between the halves on the finish stream or behind it (the exchange reads
 *     d_maskwords_out / the packed words there, and its output must be visible to the finish stream before
 *     yh_run_batch_words_unpack_device / yh_run_batch_finish_range_device), reads the second half's outputs there too, and does not start a slot's next first half before that slot's second half and rows
 *     have finished (BatchedRangeRunner reads the block's entry count back first: a host wait).
 * yh_run_batch_rows_unpack_device and everything else stay on the handle's stream.  NULL = one stream again; the call
 * drains the previous finish stream.  hip_stream must be a stream of the handle's device.                              */
int yh_db_set_batch_finish_stream(yh_db* db, void* hip_stream);

/* ---- the result of a batch in compact form: north star's "final gather of the per-reference counts" -------------------
 * The three [n_samples][N] rows of a batch are almost all zero (a 10^6-hash metagenome overlaps a few hundred of 85 205
 * references): what has to leave the GPU -- to the host, or to the rank that sums the shares of a hash-range run -- is
 * one entry per (reference r, sample s) of the batch's subset, i.e. per set bit of the slot's subset words of r, in
 * (r, s) order.  On hash-range shards the subset is the GLOBAL one after the second half, so EVERY rank has the same
 * entries in the same order and only the VALUES differ: a rank packs its three shares per entry, the value arrays are
 * summed over the ranks as they are (12 bytes per entry and rank on the wire, no keys), and the consumer unpacks rows:
 *   yh_run_batch_rows_pack_device    d_vals[3 * k + {0,1,2}] = overlap, n_excl, n_match of entry k (k < cap_rows);
 *                                    *d_n_rows = the number of entries (may exceed cap_rows: then only the first
 *                                    cap_rows were written and the caller falls back to the dense rows)
 *   yh_run_batch_rows_unpack_device  d_rows[k] = {sample, ref, overlap, n_excl, n_match} from (summed) values
 * Both read the subset words the last yh_run_batch_device (slot 0) or yh_run_batch_finish_range_device (its slot) left in
 * the slot: YH_ERR_INVALID_ARG when the slot holds none.  Enqueued on the handle's stream, no host sync.          */
typedef struct yh_batch_row { uint32_t sample, ref, overlap, n_excl, n_match; } yh_batch_row;
int yh_run_batch_rows_pack_device(yh_db* db, int slot, const uint32_t* d_overlap, const uint32_t* d_n_excl,
                                  const uint32_t* d_n_match, uint32_t* d_vals, uint64_t cap_rows, uint32_t* d_n_rows);
int yh_run_batch_rows_unpack_device(yh_db* db, int slot, const uint32_t* d_vals, uint64_t cap_rows, yh_batch_row* d_rows,
                                    uint32_t* d_n_rows);

/* ---- the subset words of a block in compact form (ABI 5) ----------------------------------------------------------------
 * Between the two halves of a batched hash-range run every rank needs the OR of all ranks' subset words.  The dense row
 * (one uint64 per reference: 8 N bytes per rank and block, 682 KB at rs214 scale) is mostly zeros -- a block of 64 samples
 * overlaps ~15 000 of 85 205 references -- so a rank all-gathers its NON-ZERO words instead:
 *   yh_run_batch_words_packed_len    uint64 words a packed buffer of capacity cap_words takes: 1 + cap + ceil(cap / 2)
 *                                    ([0] = the rank's number of non-zero words, then the words, then 32-bit reference ids)
 *   yh_run_batch_words_pack_device   d_words [n_planes][N] (what yh_run_batch_local_range_device left; n_planes =
 *                                    YH_BATCH_PLANES of the block's samples -- ABI 8) -> d_packed; an entry's 32-bit id is
 *                                    its word's index in that array (plane * N + reference); the count in [0] is the true
 *                                    one even when it exceeds cap_words (then only cap_words entries were written)
 *   yh_run_batch_words_unpack_device d_gathered = n_ranks packed buffers back to back (an all-gather's output) ->
 *                                    d_words_out [n_planes][N] = their OR -- pass it to yh_run_batch_finish_range_device
 *                                    with n_ranks = 1 --; *d_overflow = 1 when some rank had more words than cap_words (the
 *                                    OR is then incomplete and the caller repeats the exchange with a larger capacity
 *                                    or with the dense rows), else 0.  Ids >= n_planes * N are ignored.
 * No handle state is read or written but the stream and N: both are enqueued on the handle's stream (on the finish stream
 * when one is set: yh_db_set_batch_finish_stream), no host sync, and may run at any point of the interleaving table above.  (No reference counterpart: run_YACHT.py:150 runs one sample per
 * process on one machine.)                                                                                              */
uint64_t yh_run_batch_words_packed_len(uint64_t cap_words);
int yh_run_batch_words_pack_device(yh_db* db, const uint64_t* d_words, uint32_t n_planes, uint64_t* d_packed, uint64_t cap_words);
int yh_run_batch_words_unpack_device(yh_db* db, const uint64_t* d_gathered, uint32_t n_ranks, uint32_t n_planes, uint64_t cap_words,
                                     uint64_t* d_words_out, uint32_t* d_overflow);

/* Pipelined host-buffer form of yh_run: SURVEY.md 8d's steady-state call -- sample H2D, kernels,
 * counts D2H -- split in two so that consecutive samples overlap.  yh_run_submit queues, on two
 * streams of the handle, the upload of `sample`, an ordering check ON THE DEVICE (a sample that fails
 * it is not looked up), the fused run kernels and the download of the three count rows into the
 * caller's buffers (one copy when they are one contiguous [3][N] block: n_excl == overlap + N,
 * n_match == n_excl + N), and returns without waiting; yh_run_wait(slot) blocks until that call's counts
 * have landed and returns YH_ERR_UNSORTED when the check failed (the buffers are then all zero).
 * `slot` in [0, YH_RUN_SLOTS): a slot holds one call in flight and must be waited for before it is
 * submitted again; calls complete in submission order.  Copies overlap the kernels only when the host
 * buffers are page-locked (yh_host_alloc, hipHostMalloc, torch pin_memory); pageable memory works,
 * synchronously.  Buffers must stay valid until yh_run_wait returns.  Needs the default (delta
 * stream) layout and the index.                                                               */
#define YH_RUN_SLOTS 4
int yh_run_submit(yh_db* db, int slot, const uint64_t* sample, uint64_t n_sample,
                  uint32_t* overlap, uint32_t* n_excl, uint32_t* n_match);
int yh_run_wait(yh_db* db, int slot);
/* The same call with less on the wire (SURVEY.md 8d counts sample H2D + kernels + counts D2H; at 10^6-hash samples the
 * step is bound by the 8 MB of PCIe upload, not by its 45 us of kernels):
 *   packed sample   yh_sample_pack turns a strictly ascending sketch into ~4.7 bytes per hash (blocks of 256: first hash +
 *                   255 bit-packed gaps at the width of the block's largest); done ONCE where the sketch is parsed
 *                   (run_YACHT.py:150-165 loads a sample once), any thread, no device.  yh_run_submit_packed uploads
 *                   that and expands it in HBM in front of the lookup; the expansion also checks the ordering, so a
 *                   forged buffer ends in YH_ERR_UNSORTED (structure errors: YH_ERR_INVALID_ARG at submit).
 *                   yh_sample_pack_bound(n) bytes always suffice; two-call sizing with packed = NULL, cap = 0.
 *                   yh_sample_unpack is the host-side inverse (tools, tests).
 *   compact rows    instead of three dense rows of N counts: one yh_run_row per reference with overlap > 0, ascending by
 *                   reference -- what hypothesis_recovery consumes (hypothesis_recovery_src.py:361-378).  yh_run_wait_rows
 *                   returns their number; YH_ERR_CAPACITY (with *n_rows = the number needed, rows[0, cap) valid) when the
 *                   buffer was too small.  A page-locked row buffer is written by the kernel itself through PCIe.
 * yh_run_submit_rows = unpacked upload, compact rows back.  All three submit forms share the slots and complete in
 * submission order; yh_run_wait works for every form (it drops the row count).
 * yh_run_rows_device: the compact rows of the step that has just been queued on the handle (yh_run_device and its
 * relatives) from its three device count rows, on the handle's stream; d_n_rows receives their number (which may
 * exceed cap_rows: rows beyond the capacity are not written).                                                   */
typedef struct yh_run_row { uint32_t ref, overlap, n_excl, n_match; } yh_run_row;
uint64_t yh_sample_pack_bound(uint64_t n_sample);
int yh_sample_pack(const uint64_t* sample, uint64_t n_sample, void* packed, uint64_t cap_bytes, uint64_t* packed_bytes);
/* The same with the number of host threads stated: <= 0 as yh_sample_pack (up to 8 for a large sample), 1 = everything on the
 * calling thread -- for a caller that packs many samples at once, one thread each.                                      */
int yh_sample_pack_threads(const uint64_t* sample, uint64_t n_sample, void* packed, uint64_t cap_bytes, uint64_t* packed_bytes,
                           int threads);
int yh_sample_unpack(const void* packed, uint64_t packed_bytes, uint64_t* sample_out, uint64_t cap, uint64_t* n_sample);
int yh_run_submit_packed(yh_db* db, int slot, const void* packed, uint64_t packed_bytes, yh_run_row* rows, uint64_t cap_rows);
int yh_run_submit_rows(yh_db* db, int slot, const uint64_t* sample, uint64_t n_sample, yh_run_row* rows, uint64_t cap_rows);
int yh_run_wait_rows(yh_db* db, int slot, uint64_t* n_rows);
int yh_run_rows_device(yh_db* db, const uint32_t* d_overlap, const uint32_t* d_n_excl, const uint32_t* d_n_match,
                       yh_run_row* d_rows, uint64_t cap_rows, uint32_t* d_n_rows);
/* Page-locked host memory for the buffers above (hipHostMalloc / hipHostFree).                */
int yh_host_alloc(void** out, uint64_t bytes);
int yh_host_free(void* p);

/* ---- yacht train: pairwise intersections ---------------------------------------------------
 * Emits every ORDERED pair (i, j), i != j, row_begin <= i < row_end, both sketches non-empty,
 * count = |R_i ∩ R_j| > 0 and !(1.0*count/|R_i| < c_thresh), sorted by (i, j).
 * Two-call sizing: cap = 0 (or too small) stores the required length in *n_out and returns
 * YH_ERR_CAPACITY when cap != 0 is too small, YH_OK when cap == 0.                            */
int yh_pairwise(yh_db* db, double c_thresh, uint64_t row_begin, uint64_t row_end, uint64_t cap,
                uint32_t* pair_i, uint32_t* pair_j, uint32_t* pair_count, uint64_t* n_out);
/* d_out[j] (device, N words) = number of hashes of reference j that another reference holds too: the work of row j
 * of the pairwise pass (yacht_amd/dist.py cuts `yacht train`'s row blocks by it).  Enqueued on the handle's stream. */
int yh_db_nshared_device(yh_db* db, uint32_t* d_out);
/* distinct hashes, hashes seen in exactly one reference, hashes kept in the index.           */
int yh_index_stats(yh_db* db, uint64_t* n_distinct, uint64_t* n_singletons, uint64_t* n_index);
/* (ABI 8) How the last yh_pairwise of this handle summed its rows: a `yacht train` handle with more references than one
 * dense row block holds (> 28 672) keeps a row's counts in a hash table over the columns the row touches
 * (*sparse_rows); a row that touches more than 2 048 columns -- a k-mer thousands of genomes share -- takes the dense
 * pass (*dense_rows).  Both 0: every row dense (small N, or a handle that is not `yacht train`'s).  Diagnostic; the
 * pairs are the same either way (reference: the dense matrix of src/cpp/main.cpp:249-312).                        */
int yh_pairwise_row_stats(yh_db* db, uint64_t* sparse_rows, uint64_t* dense_rows);

/* Greedy size-ordered selection (host).  `pair_i/pair_j` are the pairs with C(i->j) >= C,
 * sorted by (i, j).  selected[] receives the kept reference ids in walk order.              */
int yh_train_select(const uint32_t* sizes, uint64_t n_refs,
                    const uint32_t* pair_i, const uint32_t* pair_j, uint64_t n_pairs,
                    uint32_t* selected, uint64_t* n_selected);

/* ---- yacht run, step 3: the binomial presence test (host; no device, no scipy) ----------------------------
 * single_hyp_test + get_alt_mut_rate (hypothesis_recovery_src.py:209-306) for n organisms at once, the eight
 * result columns of hypothesis_recovery (:377-391) -- n_excl and n_match are the caller's own inputs:
 *   n_excl_cov[i]    = int(n_excl[i] * min_coverage)                 ("num_exclusive_kmers_to_genome_coverage")
 *   threshold[i]     = binom.ppf(1 - significance, n_excl_cov, ani_thresh ** ksize)   ("acceptance_threshold_with_coverage")
 *   confidence[i]    = 1 - binom.cdf(threshold, n_excl_cov, p)        ("actual_confidence_with_coverage")
 *   alt_mut_rate[i]  = 1 - (1 - betaincinv(n_excl_cov - threshold, 1 + threshold, significance)) ** (1 / ksize), NaN -> -1
 *   p_val[i]         = binom.cdf(n_match, n_excl_cov, p) if n_match <= n_excl_cov else 1
 *   in_sample_est[i] = n_match >= threshold and n_match != 0
 * Exact log-space binomial tails (Loader's point probabilities) and a Newton solve of the regularized incomplete
 * beta: integer columns and decisions equal scipy's, floating columns to ~1e-13 relative (tests/test_hyp_native.py
 * against the reference's own outputs).  Needs no GPU and no handle.                                          */
int yh_hyp_test(uint64_t n, const uint32_t* n_excl, const uint32_t* n_match, int ksize, double significance,
                double ani_thresh, double min_coverage, uint8_t* in_sample_est, double* p_val,
                uint32_t* n_excl_cov, double* threshold, double* confidence, double* alt_mut_rate);

/* ---- ingest: the sketches of many .sig files (in front of yh_db_create) ------------------------------
 * What the reference's train core does before anything else (src/cpp/main.cpp:62-124,
 * read_min_hashes / read_sketches_one_chunk / read_sketches): every path is a sourmash JSON file,
 * of which record 0, signature 0, "mins" is taken (ksize is not checked there either).  A file that
 * cannot be opened is an EMPTY sketch with status 1 (the reference prints "Could not open the file!" and
 * goes on); a file that does not parse is empty with status 2 -- the reference's process dies there, so
 * callers must fail (yh_sig_batch_status: one byte per path).  Read and parsed by `threads` host threads.
 * Two-step hand-over into caller-owned arrays: sizes first (offsets[n_paths + 1], offsets[0] = 0),
 * then values[offsets[n_paths]] -- exactly the CSR yh_db_create takes.  Mins come out ascending and
 * unique (sourmash writes them so; other writers are sorted and de-duplicated).                    */
typedef struct yh_sig_batch yh_sig_batch;
int yh_sig_batch_read(const char* const* paths, uint64_t n_paths, int threads, yh_sig_batch** out);
int yh_sig_batch_status(const yh_sig_batch* batch, uint8_t* status);
int yh_sig_batch_sizes(const yh_sig_batch* batch, uint64_t* offsets);
int yh_sig_batch_values(const yh_sig_batch* batch, uint64_t* values);
/* (ABI 8) The same sketches as a PACKED CSR -- yh_db_create_packed's input, ~5.7 instead of 8 bytes per hash -- made from the
 * parsed files directly (two-call sizing as yh_csr_pack; YH_ERR_UNSORTED when some file's mins are not strictly ascending: the
 * caller takes yh_sig_batch_values then).  `yacht train` uploads this blob and writes the selected sketches of it for `yacht run`.
 * (Replaces, with yh_sig_batch_read, the reference's file stage src/cpp/main.cpp:62-124 -- JSON into vector<vector<hash_t>>.)      */
int yh_sig_batch_pack(const yh_sig_batch* batch, void* packed, uint64_t cap_bytes, uint64_t* packed_bytes, int threads);
int yh_sig_batch_destroy(yh_sig_batch* batch);

/* The two host passes `yacht train` makes over the signature files before the core (make_training_data_from_sketches.py
 * :107-133, utils.py:201-221, :499-509), threaded:
 *   yh_gunzip_files    every "x.sig.gz" -> "x.sig" next to it, the .gz removed (status[i]: 0 ok, 1 failed: left as it was)
 *   yh_sig_meta_read   what get_info_from_single_sig takes from the ONE signature of k-mer size `ksize` in each file:
 *                      record name, md5 of the sketch (md5 over str(ksize) + every hash in decimal), mean abundance,
 *                      number of hashes, scaled = round(2^64 / max_hash).  status: 0 ok, 1 cannot open, 2 malformed,
 *                      3 not exactly one signature of that k-mer size, 4 empty sketch, 5 a shape this reader leaves to
 *                      the general (Python) one -- unsorted mins, non-integer fields.
 *   yh_sig_meta_get    arrays of n entries; md5 as n x 33 bytes (NUL-terminated); name_offsets[n + 1] into the byte
 *                      buffer yh_sig_meta_names fills (UTF-8, not terminated).  has_abundance: bit 0 = the signature has
 *                      abundances; bit 1 = it is record 0 / signature 0 of its file, i.e. its mins are the sketch
 *                      yh_sig_meta_take_batch hands over for this file.
 *   yh_sig_meta_read_keep / yh_sig_meta_take_batch   the same pass also keeps what the train core reads from each file
 *                      (yh_sig_batch_read's sketches and statuses, taken from the text while it is in memory), and hands
 *                      it over as a yh_sig_batch: the 85 205 files of a GTDB training set are read once, not twice.     */
typedef struct yh_sig_meta yh_sig_meta;
int yh_gunzip_files(const char* const* paths, uint64_t n_paths, int threads, uint8_t* status);
int yh_sig_meta_read(const char* const* paths, uint64_t n_paths, int ksize, int threads, yh_sig_meta** out);
int yh_sig_meta_read_keep(const char* const* paths, uint64_t n_paths, int ksize, int threads, yh_sig_meta** out);
int yh_sig_meta_take_batch(yh_sig_meta* meta, yh_sig_batch** out);
int yh_sig_meta_get(const yh_sig_meta* meta, uint8_t* status, uint64_t* n_hashes, uint64_t* scaled, double* mean_abundance,
                    uint8_t* has_abundance, char* md5, uint64_t* name_offsets);
int yh_sig_meta_names(const yh_sig_meta* meta, char* names);
int yh_sig_meta_destroy(yh_sig_meta* meta);
/* The three passes in ONE over the archive itself (`yacht train` on a sourmash .zip database: unzip + gunzip + metadata,
 * make_training_data_from_sketches.py:107-133, utils.py:201-221, :499-509): the central directory is read once (zip64
 * too: a GTDB database has more than 65 535 members), then `threads` host threads pread and inflate the members and feed
 * every "signatures/<x>.sig[.gz]" straight into the metadata scanner and the train core's sketch reader -- as
 * yh_sig_meta_read_keep does for files -- in central-directory order.  out_dir != NULL: every member is also written
 * below out_dir as the reference's passes leave it (a .sig.gz member as the .sig it inflates to; members may not leave
 * out_dir); out_dir == NULL: nothing is written.
 *   yh_sig_meta_count   signature members found
 *   yh_sig_meta_paths   their paths relative to out_dir (path_offsets[n + 1]; paths == NULL: sizes only)            */
int yh_zip_sig_ingest(const char* zip_path, const char* out_dir, int ksize, int threads, yh_sig_meta** out);
/* The same files WITHOUT the metadata, written in the background: creating 85 205 files in one directory is mostly the
 * directory's lock, so `yacht train` reads the archive with out_dir == NULL above and lets threads of their own leave the
 * unzipped members behind while it does the rest.  yh_zip_extract_start returns at once; yh_zip_extract_wait joins, frees
 * the job and reports a member that could not be read, inflated or written (YH_ERR_INVALID_ARG).                    */
typedef struct yh_zip_job yh_zip_job;
int yh_zip_extract_start(const char* zip_path, const char* out_dir, int threads, yh_zip_job** out);
int yh_zip_extract_wait(yh_zip_job* job, uint64_t* n_members);
int yh_sig_meta_count(const yh_sig_meta* meta, uint64_t* n);
int yh_sig_meta_paths(const yh_sig_meta* meta, uint64_t* path_offsets, char* paths);

/* ---- sketching (next to the path: SURVEY.md §8f N2) --------------------------------------------
 * DNA FracMinHash as `sourmash sketch dna -p k=K,scaled=S,abund` defines it (the reference shells
 * out to it: sketch_ref_genomes.py:25,61, sketch_sample.py:32,49): every length-`ksize` window of
 * `seq` made only of A/C/G/T (either case; any other byte, e.g. a record separator, breaks the
 * window) -> canonical k-mer -> first 64 bits of MurmurHash3_x64_128 with `seed` (sourmash: 42)
 * -> kept iff <= max_hash (sourmash: floor((2^64-1)/scaled)).  hashes_out receives the kept
 * hashes unsorted WITH duplicates (sort + count them for mins/abundances); *n_out their number;
 * YH_ERR_CAPACITY (with *n_out set) when cap is too small, cap = 0 only counts.               */
int yh_sketch_dna(const uint8_t* seq, uint64_t n_bytes, int ksize, uint64_t seed, uint64_t max_hash,
                  int device_id, uint64_t cap, uint64_t* hashes_out, uint64_t* n_out);
/* The same on device buffers (a sequence that is already in HBM), enqueued on `stream` (a hipStream_t, NULL = the
 * default stream) without synchronizing: *d_count is zeroed first and receives the number of kept hashes, which may
 * exceed cap -- then only the first cap were stored. */
int yh_sketch_dna_device(const uint8_t* d_seq, uint64_t n_bytes, int ksize, uint64_t seed, uint64_t max_hash,
                         uint64_t cap, uint64_t* d_hashes_out, uint64_t* d_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YACHT_HIP_H */
