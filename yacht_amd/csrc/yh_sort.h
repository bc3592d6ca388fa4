// yh_sort.h — the distribution sort of the (hash, reference) pairs (yh_sort.hip); internal to libyacht_hip.so.
#pragma once
#include "yh_common.h"

struct yh_psort;
// uniform-enough keys of a database of H pairs with hashes in [0, max_hash]?  (false: sort with rocPRIM)
bool yh_psort_applicable(u64 H, u64 max_hash);
// regions and counters for H pairs (temporaries from the handle's buffer cache)
int yh_psort_begin(yh_db* db, u64 H, u64 max_hash, yh_psort** out);
// first level for n more pairs, on the handle's stream (the pieces of a database may arrive in any number of calls)
// d_vals = NULL (after yh_psort_positions): the value of pair i is its CSR position pos_base + i
int yh_psort_add(yh_db* db, yh_psort* s, const u64* d_keys, const u32* d_vals, u64 n, u64 pos_base = 0);
// second level + the sort of every bucket: d_keys_out / d_vals_out receive all pairs in (hash, reference) order.
// *took_it = false when a capacity was exceeded on the device (keys not uniform enough): nothing usable was written.
// Synchronizes the handle's stream.
// *unsorted (optional; with yh_psort_check_order): some sketch of the input was not strictly ascending.
int yh_psort_finish(yh_db* db, yh_psort* s, u64* d_keys_out, u32* d_vals_out, bool* took_it, bool* unsorted = nullptr);
// the input of yh_psort_add is a CSR in reference order with the reference id as value: check every sketch's ordering on the
// way through the first level (what k_scan_refs would read the whole database for once more)
void yh_psort_check_order(yh_psort* s, bool on);
// the sorted pairs in CHUNKS = buckets (valid after a finish that took the input, until destroy): chunk c = sorted positions
// [d_chunk_off[c], d_chunk_off[c + 1]); d_chunk_counts[3 c ..] = its {distinct hashes, shared hashes, pairs of shared hashes}
void yh_psort_chunks(const yh_psort* s, u64* n_chunks, const u64** d_chunk_off, const u32** d_chunk_counts);
void yh_psort_destroy(yh_db* db, yh_psort* s);

// ---- position mode: the fused last pass of a YH_DB_PAIRWISE_ONLY handle (yh_db::fz) ----------------------------------
constexpr unsigned YH_REF_TAB_SH = 8;  // one look-up entry per 256 CSR positions
// d_tab[j] = the reference that owns CSR position j << YH_REF_TAB_SH, for j in [0, (H >> YH_REF_TAB_SH) + 2)
int yh_ref_table_build(yh_db* db, const u64* d_offsets, u64 n_refs, u64 H, u32* d_tab);
// the pairs carry CSR positions instead of reference ids (equal hashes still end up in ascending reference order)
// (d_rec: the H records of yh_psort_finish_emit -- the first level clears each pair's on its way through)
void yh_psort_positions(yh_psort* s, const u32* d_ref_tab, const u64* d_offsets, u64 n_refs, u64* d_rec);
// second level + every bucket sorted in LDS and turned into the pairwise pass's records on the spot (see yh_sort.hip)
int yh_psort_finish_emit(yh_db* db, yh_psort* s, u64* d_rec, u64 n_refs, u64 totals[3], u32** d_list_out, bool* took_it, bool* unsorted = nullptr);

// ---- the same without a first level (round 5): regions are read in place as contiguous PIECES of the ascending sketches ----
struct yh_pieces;
bool yh_pc_applicable(u64 H, u64 max_hash, u64 n_refs);
// d_rec: the H records of the pairwise pass (cleared by the bounds pass on its way through)
int yh_pc_begin(yh_db* db, u64 H, u64 max_hash, u64 n_refs, u64* d_rec, yh_pieces** out);
// the bounds pass over the sketches [r0, r1) holding n_pairs hashes (the chunks of an upload as they arrive); check_order:
// flag sketches that are not strictly ascending
int yh_pc_scan(yh_db* db, yh_pieces* s, const u64* d_values, const u64* d_offsets, u64 r0, u64 r1, u64 n_pairs, bool check_order);
// distribution into the buckets + the fused last pass; semantics of yh_psort_finish_emit.  Synchronizes the stream.
// *spill (optional): what went through the side list of overflowed buckets -- their groups' holder lists (the caller's to
// yh_tfree, nullptr when nothing spilled), the start value from which a list record names that array, and the counts
struct yh_pc_spill { u32* d_list2 = nullptr; u64 list_split = 0, n_pairs = 0, n_buckets = 0; };
int yh_pc_finish_emit(yh_db* db, yh_pieces* s, const u64* d_values, const u64* d_offsets, u64 totals[3], u32** d_list_out, bool* took_it,
                      bool* unsorted = nullptr, yh_pc_spill* spill = nullptr);
// rocPRIM's radix sort of (u64 key, u32 value) pairs over the low end_bit bits (yh_build.hip, where rocPRIM lives)
int yh_radix_sort_pairs_u64_u32(yh_db* db, const u64* k_in, u64* k_out, const u32* v_in, u32* v_out, u64 n, unsigned end_bit);
void yh_pc_destroy(yh_db* db, yh_pieces* s);
// every OTHER handle through the same distribution: all pairs in (hash, reference) order in d_keys_out / d_vals_out (H entries each),
// the buckets as chunks (*chunks_out: for yh_psort_chunks, the caller's to yh_psort_destroy); buckets that overflow go through
// a side list sorted by rocPRIM and come back to their places.  *took_it = false: not this path's keys, nothing usable written.
int yh_pc_sort(yh_db* db, const u64* d_values, const u64* d_offsets, u64 n_refs, u64 H, u64 max_hash, bool check_order,
               u64* d_keys_out, u32* d_vals_out, yh_psort** chunks_out, bool* took_it, bool* unsorted, u64* n_spilled_pairs, u64* n_spilled_buckets);
