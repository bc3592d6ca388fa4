// yh_sort.h — the distribution sort of the (hash, reference) pairs (yh_sort.hip); internal to libyacht_hip.so.
#pragma once
#include "yh_common.h"

struct yh_psort;
// uniform-enough keys of a database of H pairs with hashes in [0, max_hash]?  (false: sort with rocPRIM)
bool yh_psort_applicable(u64 H, u64 max_hash);
// regions and counters for H pairs (temporaries from the handle's buffer cache)
int yh_psort_begin(yh_db* db, u64 H, u64 max_hash, yh_psort** out);
// first level for n more pairs, on the handle's stream (the pieces of a database may arrive in any number of calls)
int yh_psort_add(yh_db* db, yh_psort* s, const u64* d_keys, const u32* d_vals, u64 n);
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
