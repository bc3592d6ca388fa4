// yh_pack.h -- the packed CSR (yh_pack.hip: yh_csr_pack's format) as the build sees it
#pragma once
#include "yh_common.h"

// A packed CSR blob in HOST memory as its parts (yh_csr_view fills it)
struct YhPackedCsr {
    const u64* offsets = nullptr;   // [n_refs + 1]
    const void* tab = nullptr;      // the block table: n_blocks entries of yh_csr_block_bytes() bytes
    const u64* payload = nullptr;   // [payload_words]
    u64 n_refs = 0, n_hashes = 0, n_blocks = 0, payload_words = 0, max_hash = 0;
    std::vector<u64> first_block;   // [n_refs + 1] blocks in front of every sketch
};
int yh_csr_view(const void* packed, u64 bytes, YhPackedCsr* v);
u64 yh_csr_block_bytes();
u64 yh_csr_block_word_off(const YhPackedCsr* v, u64 b);  // first payload word of block b (b == n_blocks: the spare word)
// the blocks [b0, b1) of a packed CSR whose table, payload, first_block[] and offsets are in HBM -> d_values; *d_flag |= 4 when a
// block points outside the payload or does not fit the offsets (on the handle's stream)
int yh_csr_expand_device(yh_db* db, const void* d_tab, const u64* d_payload, u64 payload_words, const u64* d_first_block, const u64* d_offsets,
                         u64 n_refs, u64 b0, u64 b1, u64* d_values, u32* d_flag);
// the chunked, overlapped upload of yh_build.hip (yh_build_upload_sorted) for a packed database: every chunk's blocks go up and
// are expanded into d_values in front of the chunk's ordering check
int yh_build_upload_sorted_packed(yh_db* db, const YhPackedCsr* pk, u64* d_values, const u64* d_offsets, u64** d_sk_out, u32** d_sv_out);
// the packer over sketches that need not lie back to back (parts[j] = sketch j's first hash); yh_csr_pack's contract otherwise
int yh_csr_pack_parts(const u64* const* parts, const u64* offsets, u64 n_refs, void* packed, u64 cap_bytes, u64* packed_bytes, int threads);
