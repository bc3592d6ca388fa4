// yh_pack.hip — the two ends of the host-buffer `yacht run` step that cross PCIe (sample H2D + kernels +
// counts D2H), made small:
//
//   packed sample    a sorted sketch of n hashes in ~4.7 bytes per hash instead of 8.  Blocks of 256 hashes; a block is
//                    its first hash (8 bytes) and 255 gaps (h[i] - h[i-1] - 1, so "strictly ascending" is structural)
//                    bit-packed at the width of the block's largest gap.  FracMinHash hashes are uniform, so a gap is
//                    geometric around range / n: 2^34.1 for 10^6 hashes at scaled = 1000, the largest of 255 of them
//                    ~2^36.6 -> 37 bits.  (The entropy of the gaps is 35.5 bits; Elias-Fano would take 36.1.)
//                    yh_sample_pack runs where the sketch is parsed (once per sample, next to a JSON parse that costs
//                    a hundred times more); k_unpack_sample expands it in HBM in front of the lookup -- one workgroup
//                    per block, one lane per hash, a block scan -- and checks what the format cannot promise: no
//                    wrap-around inside a block, blocks ascending.
//   compact rows     the run step's three count rows are ~99 % zeros (a sample overlaps ~1 % of GTDB): k_compact_rows
//                    writes one 16-byte row (reference, overlap, n_exclusive, n_matches) per reference with overlap > 0,
//                    in reference order, straight into the caller's page-locked buffer, and the number of rows.
//                    1 MB of D2H per step becomes ~14 KB.
#include "yh_common.h"
#include "yh_pack.h"

#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace {

constexpr u32 PACK_MAGIC = 0x31504859u;  // "YHP1"
constexpr u32 PACK_BLOCK = 256;

struct PackHeader {   // 32 bytes
    u32 magic, block;
    u64 n;            // hashes
    u64 payload_words;  // u64 words behind the block table (incl. one spare word: two-word reads never overrun)
    u64 reserved;
};
struct PackBlock {    // 16 bytes
    u64 base;         // the block's first hash
    u32 word_off;     // first payload word of the block
    u32 width;        // bits per gap, 0..64
};
static_assert(sizeof(PackHeader) == 32 && sizeof(PackBlock) == 16, "packed sample layout");

inline u64 n_blocks(u64 n) { return (n + PACK_BLOCK - 1) / PACK_BLOCK; }
inline u32 bits_of(u64 v) { return v ? 64u - (u32)__builtin_clzll(v) : 0u; }
inline u64 block_words(u32 cnt, u32 width) { return ((u64)(cnt - 1) * width + 63) / 64; }

// one workgroup per block, lane i = hash i of the block
__global__ void __launch_bounds__(PACK_BLOCK) k_unpack_sample(const PackBlock* __restrict__ tab, const u64* __restrict__ payload,
                                                              u64 n, u64* __restrict__ out, u32* __restrict__ bad, u32 gen,
                                                              u32* __restrict__ host_bad) {
    __shared__ u64 wave_sum[PACK_BLOCK / 64];
    __shared__ u64 wave_last[PACK_BLOCK / 64];
    const u64 b = blockIdx.x;
    const u32 i = threadIdx.x, lane = i & 63u, wv = i >> 6;
    const PackBlock blk = tab[b];
    const u64 first = b * PACK_BLOCK;
    const u32 cnt = (u32)min((u64)PACK_BLOCK, n - first);
    u64 step = 0;  // h[i] - h[i-1]; 0 for lane 0 and for lanes behind the block's end
    if (i >= 1 && i < cnt) {
        const u64 bit = (u64)(i - 1) * blk.width;
        const u64* p = payload + blk.word_off + (bit >> 6);
        const u32 sh = (u32)(bit & 63u);
        u64 v = 0;
        if (blk.width) {
            v = p[0] >> sh;
            if (sh + blk.width > 64) v |= p[1] << (64 - sh);
            if (blk.width < 64) v &= (1ull << blk.width) - 1ull;
        }
        step = v + 1ull;
    }
    // inclusive scan of the steps: inside the wave, then across the four waves
    u64 acc = step;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 t = ((u64)(u32)__shfl_up((int)(u32)(acc >> 32), off) << 32) | (u32)__shfl_up((int)(u32)acc, off);
        if (lane >= (u32)off) acc += t;
    }
    if (lane == 63) wave_sum[wv] = acc;
    __syncthreads();
    u64 before = 0;
    for (u32 w = 0; w < wv; ++w) before += wave_sum[w];
    const u64 h = blk.base + before + acc;
    // strictly ascending: every lane against its predecessor (a wrap-around past 2^64 shows as a descent somewhere),
    // the block's last hash against the next block's first
    const u64 prev_in_wave = ((u64)(u32)__shfl_up((int)(u32)(h >> 32), 1) << 32) | (u32)__shfl_up((int)(u32)h, 1);
    if (lane == 63) wave_last[wv] = h;
    __syncthreads();
    bool wrong = false;
    if (i >= 1 && i < cnt) {
        const u64 prev = lane ? prev_in_wave : wave_last[wv - 1];
        wrong = !(prev < h) || step == 0;  // (step == 0: v + 1 wrapped, a gap of 2^64)
    }
    if (i == cnt - 1 && first + cnt < n) wrong = wrong || !(h < tab[b + 1].base);
    if (i < cnt) out[first + i] = h;
    if (wrong) {
        *bad = gen;
        if (host_bad) *host_bad = 1;
    }
}

// ---- a whole CSR packed the same way (yh_csr_pack / yh_db_create_packed) ---------------------------------------------------
// Every sketch on its own: blocks of 256 hashes, a block = its first hash + the gaps at the block's widest gap's width; sketches
// of ~5 000 hashes at scaled = 1000 take ~5.7 bytes per hash.  The blob: CsrHeader | offsets [n_refs + 1] | CsrBlock [n_blocks]
// (the blocks of sketch 0, of sketch 1, ...: ceil(size / 256) each) | payload words (one spare word at the end).
constexpr u32 CSR_MAGIC = 0x31434859u;  // "YHC1"
struct CsrHeader {   // 64 bytes
    u32 magic, block;
    u64 n_refs, n_hashes, n_blocks, payload_words, max_hash;
    u64 reserved[2];
};
struct CsrBlock {    // 24 bytes
    u64 base;        // the block's first hash
    u64 word_off;    // first payload word of the block
    u32 width;       // bits per gap, 0..64
    u32 reserved;
};
static_assert(sizeof(CsrHeader) == 64 && sizeof(CsrBlock) == 24, "packed CSR layout");

// one workgroup per block b0 + blockIdx.x: which sketch it belongs to from first_block[] (blocks in front of every sketch),
// where its hashes go from the offsets; what the format cannot promise -- payload words inside the blob -- is checked
// (flag |= 4), the ordering is the ordering check's business (k_scan_refs / k_piece_bounds read what is written here).
__global__ void __launch_bounds__(PACK_BLOCK) k_unpack_csr(const CsrBlock* __restrict__ tab, const u64* __restrict__ payload, u64 payload_words,
                                                           const u64* __restrict__ first_block, const u64* __restrict__ offsets, u64 n_refs,
                                                           u64 b0, u64* __restrict__ out, u32* __restrict__ flag) {
    __shared__ u64 wave_sum[PACK_BLOCK / 64];
    const u64 b = b0 + blockIdx.x;
    const u32 i = threadIdx.x, lane = i & 63u, wv = i >> 6;
    // the sketch: the largest j with first_block[j] <= b (empty sketches share their successor's entry and are passed over)
    u64 lo = 0, hi = n_refs;  // first_block[lo] <= b < first_block[hi] (first_block[n_refs] = all blocks)
    while (hi - lo > 1) {
        const u64 mid = (lo + hi) >> 1;
        if (first_block[mid] <= b) lo = mid; else hi = mid;
    }
    const u64 j = lo;
    const u64 k = b - first_block[j];
    const u64 beg = offsets[j] + k * PACK_BLOCK, end = offsets[j + 1];
    if (beg >= end) { if (i == 0) atomicOr(flag, 4u); return; }  // (a table that does not fit the offsets)
    const u32 cnt = (u32)min((u64)PACK_BLOCK, end - beg);
    const CsrBlock blk = tab[b];
    const u64 nw = blk.width <= 64u ? ((u64)(cnt - 1) * blk.width + 63) / 64 : ~0ull;
    if (blk.width > 64u || blk.word_off > payload_words || nw + 1 > payload_words - blk.word_off) {  // (+ 1: two-word reads)
        if (i == 0) atomicOr(flag, 4u);
        return;
    }
    u64 step = 0;  // h[i] - h[i-1]; 0 for lane 0 and for lanes behind the block's end
    if (i >= 1 && i < cnt) {
        const u64 bit = (u64)(i - 1) * blk.width;
        const u64* p = payload + blk.word_off + (bit >> 6);
        const u32 sh = (u32)(bit & 63u);
        u64 v = 0;
        if (blk.width) {
            v = p[0] >> sh;
            if (sh + blk.width > 64) v |= p[1] << (64 - sh);
            if (blk.width < 64) v &= (1ull << blk.width) - 1ull;
        }
        step = v + 1ull;
    }
    u64 acc = step;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 t = ((u64)(u32)__shfl_up((int)(u32)(acc >> 32), off) << 32) | (u32)__shfl_up((int)(u32)acc, off);
        if (lane >= (u32)off) acc += t;
    }
    if (lane == 63) wave_sum[wv] = acc;
    __syncthreads();
    u64 before = 0;
    for (u32 w = 0; w < wv; ++w) before += wave_sum[w];
    if (i < cnt) out[beg + i] = blk.base + before + acc;
}

// Rows of the references with overlap > 0, in reference order.  One workgroup per 2048 references (64 words of subset
// bits); its first row number is the popcount of all the words in front, which every workgroup sums for itself (N / 32
// words: 10 KB at 85 205 references, L2-resident).
struct RowsOut {
    uint4* rows;      // device view of the caller's buffer (page-locked host memory or HBM)
    u64 cap;
    u32* count_dev;   // may be null: number of rows (device word)
    u32* count_host;  // may be null: the same, in page-locked host memory
};
__global__ void __launch_bounds__(1024) k_compact_rows(u64 n, const u32* __restrict__ maskbits, const u32* __restrict__ overlap,
                                                       const u32* __restrict__ n_excl, const u32* __restrict__ n_match,
                                                       RowsOut o) {
    __shared__ u32 wsum[16];
    __shared__ u32 wcnt[64];   // popcounts of this workgroup's 64 words
    __shared__ u32 wpre[65];   // their exclusive prefix
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const u64 word0 = (u64)blockIdx.x * 64;          // first mask word of this workgroup
    const u64 n_words = (n + 31) / 32;
    // rows in front of this workgroup
    u32 part = 0;
    for (u64 w = tid; w < word0; w += 1024) part += (u32)__popc(maskbits[w]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += (u32)__shfl_xor((int)part, off);
    if (lane == 0) wsum[wv] = part;
    if (tid < 64) wcnt[tid] = (word0 + tid < n_words) ? (u32)__popc(maskbits[word0 + tid]) : 0u;
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (int w = 0; w < 64; ++w) { wpre[w] = run; run += wcnt[w]; }
        wpre[64] = run;
    }
    __syncthreads();
    u32 base = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) base += wsum[w];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const u32 t = tid + 1024u * k;             // reference inside the workgroup's 2048
        const u64 j = word0 * 32 + t;
        if (j < n) {
            const u32 word = maskbits[word0 + (t >> 5)];
            if ((word >> (t & 31u)) & 1u) {
                const u64 row = (u64)base + wpre[t >> 5] + (u32)__popc(word & ((1u << (t & 31u)) - 1u));
                if (row < o.cap) o.rows[row] = make_uint4((u32)j, overlap[j], n_excl[j], n_match[j]);
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const u32 total = base + wpre[64];
        if (o.count_dev) *o.count_dev = total;
        if (o.count_host) *o.count_host = total;
    }
}

}  // namespace

// ---- internal (yh_api.hip) ------------------------------------------------------------------------------------------
// structure of a packed sample held in HOST memory: sizes, widths, payload offsets in bounds.  *n_out = its hash count.
int yh_pack_validate(const void* packed, u64 bytes, u64* n_out) {
    if (!packed || bytes < sizeof(PackHeader)) { yh_set_error("packed sample: shorter than its header"); return YH_ERR_INVALID_ARG; }
    PackHeader hd;
    memcpy(&hd, packed, sizeof(hd));
    if (hd.magic != PACK_MAGIC || hd.block != PACK_BLOCK) { yh_set_error("packed sample: bad magic / block size"); return YH_ERR_INVALID_ARG; }
    if (hd.n > 0xfffffff0ull) { yh_set_error("packed sample: more than 2^32-16 hashes"); return YH_ERR_INVALID_ARG; }
    const u64 nb = n_blocks(hd.n);
    if (hd.payload_words > (1ull << 33) || bytes != sizeof(PackHeader) + nb * sizeof(PackBlock) + hd.payload_words * 8) {
        yh_set_error("packed sample: size does not match its header");
        return YH_ERR_INVALID_ARG;
    }
    const PackBlock* tab = reinterpret_cast<const PackBlock*>((const char*)packed + sizeof(PackHeader));
    for (u64 b = 0; b < nb; ++b) {
        PackBlock blk;
        memcpy(&blk, tab + b, sizeof(blk));
        const u32 cnt = (u32)std::min<u64>(PACK_BLOCK, hd.n - b * PACK_BLOCK);
        // (+1: the kernel may read the word behind a gap that ends on a word boundary)
        if (blk.width > 64 || (u64)blk.word_off + block_words(cnt, blk.width) + 1 > hd.payload_words) {
            yh_set_error("packed sample: block %llu points outside the payload", b);
            return YH_ERR_INVALID_ARG;
        }
    }
    *n_out = hd.n;
    return YH_OK;
}

// d_packed: the same bytes in HBM (8-byte aligned); expands into d_out[n] and stores `gen` into *d_bad (and 1 into
// *h_bad_dev, page-locked host memory, when given) if the hashes do not come out strictly ascending
int yh_pack_expand_device(yh_db* db, const void* d_packed, u64 n, u64* d_out, u32* d_bad, u32 gen, u32* h_bad_dev) {
    if (n == 0) return YH_OK;
    const u64 nb = n_blocks(n);
    const PackBlock* tab = reinterpret_cast<const PackBlock*>((const char*)d_packed + sizeof(PackHeader));
    const u64* payload = reinterpret_cast<const u64*>(tab + nb);
    k_unpack_sample<<<(u32)nb, PACK_BLOCK, 0, db->stream>>>(tab, payload, n, d_out, d_bad, gen, h_bad_dev);
    YH_HIP(hipGetLastError());
    return YH_OK;
}

// ---- the packed CSR (yh_csr_pack): a host view of a blob, and its blocks expanded on the device -------------------------
// Sizes against the header, offsets ascending from 0 to n_hashes, as many blocks as the sketches need; v->first_block[j] =
// blocks in front of sketch j.  (Block widths and payload offsets are the unpack kernel's to check: it reads them anyway.)
int yh_csr_view(const void* packed, u64 bytes, YhPackedCsr* v) {
    if (!packed || bytes < sizeof(CsrHeader)) { yh_set_error("packed CSR: shorter than its header"); return YH_ERR_INVALID_ARG; }
    if (reinterpret_cast<uintptr_t>(packed) & 7u) { yh_set_error("packed CSR: the buffer must be 8-byte aligned"); return YH_ERR_INVALID_ARG; }
    CsrHeader hd;
    memcpy(&hd, packed, sizeof(hd));
    if (hd.magic != CSR_MAGIC || hd.block != PACK_BLOCK) { yh_set_error("packed CSR: bad magic / block size"); return YH_ERR_INVALID_ARG; }
    if (hd.n_refs > 0x7ffffff0ull || hd.n_hashes > (1ull << 48) || hd.n_blocks > (1ull << 41) || hd.payload_words > (1ull << 48) || hd.payload_words < 1) {
        yh_set_error("packed CSR: implausible header");
        return YH_ERR_INVALID_ARG;
    }
    const u64 need = sizeof(CsrHeader) + (hd.n_refs + 1) * 8 + hd.n_blocks * sizeof(CsrBlock) + hd.payload_words * 8;
    if (bytes != need) { yh_set_error("packed CSR: %llu bytes, its header says %llu", (u64)bytes, need); return YH_ERR_INVALID_ARG; }
    v->n_refs = hd.n_refs;
    v->n_hashes = hd.n_hashes;
    v->n_blocks = hd.n_blocks;
    v->payload_words = hd.payload_words;
    v->max_hash = hd.max_hash;
    v->offsets = reinterpret_cast<const u64*>((const char*)packed + sizeof(CsrHeader));
    v->tab = v->offsets + hd.n_refs + 1;
    v->payload = reinterpret_cast<const u64*>((const char*)v->tab + hd.n_blocks * sizeof(CsrBlock));
    v->first_block.assign(hd.n_refs + 1, 0);
    if (v->offsets[0] != 0) { yh_set_error("offsets[0] must be 0"); return YH_ERR_INVALID_ARG; }
    // The block table is untrusted input as well: every block's payload starts where the one before it ends --
    // the only layout yh_csr_pack writes -- so the chunked upload's payload ranges [w0, w1) are monotone, cover what their
    // blocks read, and nothing decodes words that were never copied.  width <= 64; the running prefix + the spare word is
    // the payload's length; a block's base (its first hash) and the largest hash a block can reach stay below the header's
    // max_hash check of the build only loosely, so the base is held to it here.
    u64 nb = 0, words = 0;
    for (u64 j = 0; j < hd.n_refs; ++j) {
        const u64 a = v->offsets[j], e = v->offsets[j + 1];
        if (e < a || e > hd.n_hashes) { yh_set_error("packed CSR: offsets are not monotone"); return YH_ERR_UNSORTED; }
        if (e - a > 0xffffffffull) { yh_set_error("a reference sketch has more than 2^32-1 hashes"); return YH_ERR_INVALID_ARG; }
        v->first_block[j] = nb;
        const u64 nbj = (e - a + PACK_BLOCK - 1) / PACK_BLOCK;
        if (nb + nbj > hd.n_blocks) { yh_set_error("packed CSR: offsets and block count do not match the header"); return YH_ERR_INVALID_ARG; }
        for (u64 first = a, b = nb; first < e; first += PACK_BLOCK, ++b) {
            CsrBlock blk;
            memcpy(&blk, (const char*)v->tab + b * sizeof(CsrBlock), sizeof(blk));
            const u32 cnt = (u32)std::min<u64>(PACK_BLOCK, e - first);
            if (blk.width > 64u || blk.word_off != words) {
                yh_set_error("packed CSR: block %llu is not where the blocks in front of it end (or its width is > 64)", b);
                return YH_ERR_INVALID_ARG;
            }
            if (blk.base > hd.max_hash) { yh_set_error("packed CSR: block %llu starts above the header's largest hash", b); return YH_ERR_INVALID_ARG; }
            words += block_words(cnt, blk.width);
            if (words >= hd.payload_words) { yh_set_error("packed CSR: block %llu points outside the payload", b); return YH_ERR_INVALID_ARG; }
        }
        nb += nbj;
    }
    v->first_block[hd.n_refs] = nb;
    if (v->offsets[hd.n_refs] != hd.n_hashes || nb != hd.n_blocks) { yh_set_error("packed CSR: offsets and block count do not match the header"); return YH_ERR_INVALID_ARG; }
    if (words + 1 != hd.payload_words) { yh_set_error("packed CSR: the payload is %llu words, its blocks take %llu + the spare word", hd.payload_words, words); return YH_ERR_INVALID_ARG; }
    return YH_OK;
}
u64 yh_csr_block_bytes() { return sizeof(CsrBlock); }
u64 yh_csr_block_word_off(const YhPackedCsr* v, u64 b) {  // first payload word of block b (b == n_blocks: the spare word's index)
    if (b >= v->n_blocks) return v->payload_words - 1;
    CsrBlock blk;
    memcpy(&blk, (const char*)v->tab + b * sizeof(CsrBlock), sizeof(blk));
    return blk.word_off;
}
// the blocks [b0, b1) of a packed CSR whose table, payload, first_block[] and offsets are in HBM -> d_values; *d_flag |= 4 when
// a block points outside the payload or does not fit the offsets
int yh_csr_expand_device(yh_db* db, const void* d_tab, const u64* d_payload, u64 payload_words, const u64* d_first_block, const u64* d_offsets,
                         u64 n_refs, u64 b0, u64 b1, u64* d_values, u32* d_flag) {
    for (u64 b = b0; b < b1; b += (u64)1 << 30) {  // (2^31 - 1 workgroups per launch)
        const u64 n = std::min<u64>(b1 - b, (u64)1 << 30);
        k_unpack_csr<<<(u32)n, PACK_BLOCK, 0, db->stream>>>(reinterpret_cast<const CsrBlock*>(d_tab), d_payload, payload_words, d_first_block,
                                                           d_offsets, n_refs, b, d_values, d_flag);
    }
    YH_HIP(hipGetLastError());
    return YH_OK;
}

// rows of the references with overlap > 0 from the three count rows of the step that just ran on the handle's stream
// (its subset bits: db->d_maskbits, or -- the step was a pipelined one -- those of the step context it ran in)
int yh_rows_compact_device(yh_db* db, const u32* d_overlap, const u32* d_excl, const u32* d_match, void* rows_dev, u64 cap,
                           u32* count_dev, u32* count_host_dev) {
    const u32* maskbits = (db->pipe_last_ctx >= 0 && db->ctx_bits[db->pipe_last_ctx]) ? db->ctx_bits[db->pipe_last_ctx] : db->d_maskbits;
    const u64 N = db->n_refs;
    if (N == 0) {
        if (count_dev) YH_HIP(hipMemsetAsync(count_dev, 0, sizeof(u32), db->stream));
        return YH_OK;
    }
    RowsOut o{reinterpret_cast<uint4*>(rows_dev), cap, count_dev, count_host_dev};
    k_compact_rows<<<(u32)((N + 2047) / 2048), 1024, 0, db->stream>>>(N, maskbits, d_overlap, d_excl, d_match, o);
    YH_HIP(hipGetLastError());
    return YH_OK;
}

// ---- C ABI: packing on the host -------------------------------------------------------------------------------------
extern "C" {

uint64_t yh_sample_pack_bound(uint64_t n_sample) {
    const u64 nb = n_blocks(n_sample);
    return sizeof(PackHeader) + nb * sizeof(PackBlock) + (n_sample + nb + 2) * 8;
}

// threads <= 0: as many as the sample is worth (at most 8); 1: everything on the calling thread (a caller that packs many
// samples at once gives every sample a thread of its own: bench.py's host-inclusive leg with the packing inside the step)
int yh_sample_pack_threads(const uint64_t* sample, uint64_t n_sample, void* packed, uint64_t cap_bytes, uint64_t* packed_bytes,
                           int threads) {
    if (!packed_bytes || (n_sample && !sample)) { yh_set_error("yh_sample_pack: null argument"); return YH_ERR_INVALID_ARG; }
    if (n_sample > 0xfffffff0ull) { yh_set_error("sample larger than 2^32-16 hashes"); return YH_ERR_INVALID_ARG; }
    const u64 n = n_sample, nb = n_blocks(n);
    const u64* h = (const u64*)sample;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = threads > 0 ? (unsigned)std::min<u64>((u64)threads, nb / 64 + 1)
                                   : (unsigned)std::min<u64>(std::min<unsigned>(hw ? hw : 1, 8), nb / 512 + 1);
    // pass 1: ordering and the width of every block (blocks are independent), then their word offsets (a prefix sum)
    std::vector<PackBlock> tab(nb);
    std::atomic<int> unsorted{0};
    auto width_range = [&](u64 b0, u64 b1) {
        for (u64 b = b0; b < b1; ++b) {
            const u64 first = b * PACK_BLOCK;
            const u32 cnt = (u32)std::min<u64>(PACK_BLOCK, n - first);
            bool bad = b && !(h[first - 1] < h[first]);
            u64 widest = 0;
            for (u32 i = 1; i < cnt; ++i) {
                bad |= !(h[first + i - 1] < h[first + i]);
                widest |= h[first + i] - h[first + i - 1] - 1ull;
            }
            if (bad) unsorted.store(1, std::memory_order_relaxed);
            tab[b] = PackBlock{h[first], 0u, bits_of(widest)};
        }
    };
    auto run_ranges = [&](auto&& fn) {
        if (T <= 1) { fn((u64)0, nb); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, nb * t / T, nb * (t + 1) / T);
        for (auto& x : th) x.join();
    };
    run_ranges(width_range);
    if (unsorted.load()) { yh_set_error("the sample sketch is not strictly ascending"); return YH_ERR_UNSORTED; }
    u64 words = 0;
    for (u64 b = 0; b < nb; ++b) {
        if (words > 0xfffffff0ull) { yh_set_error("packed sample too large"); return YH_ERR_INVALID_ARG; }
        tab[b].word_off = (u32)words;
        words += block_words((u32)std::min<u64>(PACK_BLOCK, n - b * PACK_BLOCK), tab[b].width);
    }
    words += 1;  // the spare word
    const u64 need = sizeof(PackHeader) + nb * sizeof(PackBlock) + words * 8;
    *packed_bytes = need;
    if (!packed || cap_bytes < need) {
        if (!packed && cap_bytes == 0) return YH_OK;  // sizing call
        yh_set_error("yh_sample_pack: buffer of %llu bytes, %llu needed", (u64)cap_bytes, need);
        return YH_ERR_CAPACITY;
    }
    PackHeader hd{PACK_MAGIC, PACK_BLOCK, n, words, 0};
    memcpy(packed, &hd, sizeof(hd));
    if (nb) memcpy((char*)packed + sizeof(hd), tab.data(), nb * sizeof(PackBlock));
    u64* payload = reinterpret_cast<u64*>((char*)packed + sizeof(hd) + nb * sizeof(PackBlock));
    // pass 2: the gaps, block by block (blocks own disjoint words: threads need no coordination)
    auto pack_range = [&](u64 b0, u64 b1) {
        for (u64 b = b0; b < b1; ++b) {
            const u64 first = b * PACK_BLOCK;
            const u32 cnt = (u32)std::min<u64>(PACK_BLOCK, n - first);
            const u32 w = tab[b].width;
            u64* out = payload + tab[b].word_off;
            const u64 nw = block_words(cnt, w);
            for (u64 k = 0; k < nw; ++k) out[k] = 0;
            if (!w) continue;
            u64 bit = 0;
            for (u32 i = 1; i < cnt; ++i, bit += w) {
                const u64 v = h[first + i] - h[first + i - 1] - 1ull;
                const u32 sh = (u32)(bit & 63u);
                out[bit >> 6] |= v << sh;
                if (sh + w > 64) out[(bit >> 6) + 1] |= v >> (64 - sh);
            }
        }
    };
    run_ranges(pack_range);
    payload[words - 1] = 0;
    return YH_OK;
}
int yh_sample_pack(const uint64_t* sample, uint64_t n_sample, void* packed, uint64_t cap_bytes, uint64_t* packed_bytes) {
    return yh_sample_pack_threads(sample, n_sample, packed, cap_bytes, packed_bytes, 0);
}

int yh_sample_unpack(const void* packed, uint64_t packed_bytes, uint64_t* sample_out, uint64_t cap, uint64_t* n_sample) {
    if (!n_sample) { yh_set_error("yh_sample_unpack: null argument"); return YH_ERR_INVALID_ARG; }
    u64 n = 0;
    YH_TRY(yh_pack_validate(packed, packed_bytes, &n));
    *n_sample = n;
    if (!sample_out && cap == 0) return YH_OK;  // sizing call
    if (cap < n || !sample_out) { yh_set_error("yh_sample_unpack: room for %llu hashes, %llu needed", (u64)cap, n); return YH_ERR_CAPACITY; }
    const u64 nb = n_blocks(n);
    const PackBlock* tab = reinterpret_cast<const PackBlock*>((const char*)packed + sizeof(PackHeader));
    const u64* payload = reinterpret_cast<const u64*>(tab + nb);
    u64 prev = 0;
    for (u64 b = 0; b < nb; ++b) {
        PackBlock blk;
        memcpy(&blk, tab + b, sizeof(blk));
        const u64 first = b * PACK_BLOCK;
        const u32 cnt = (u32)std::min<u64>(PACK_BLOCK, n - first);
        u64 cur = blk.base;
        if (b && !(prev < cur)) { yh_set_error("packed sample: blocks are not ascending"); return YH_ERR_UNSORTED; }
        sample_out[first] = cur;
        u64 bit = 0;
        for (u32 i = 1; i < cnt; ++i, bit += blk.width) {
            u64 v = 0;
            if (blk.width) {
                const u32 sh = (u32)(bit & 63u);
                v = payload[blk.word_off + (bit >> 6)] >> sh;
                if (sh + blk.width > 64) v |= payload[blk.word_off + (bit >> 6) + 1] << (64 - sh);
                if (blk.width < 64) v &= (1ull << blk.width) - 1ull;
            }
            const u64 next = cur + v + 1ull;
            if (!(cur < next)) { yh_set_error("packed sample: a gap wraps around 2^64"); return YH_ERR_UNSORTED; }
            cur = next;
            sample_out[first + i] = cur;
        }
        prev = cur;
    }
    return YH_OK;
}

// ---- a whole CSR (yh_db_create_packed's input) ------------------------------------------------------------------------
uint64_t yh_csr_pack_bound(uint64_t n_hashes, uint64_t n_refs) {
    const u64 nb = n_hashes / PACK_BLOCK + n_refs + 1;  // (every sketch may end in a partial block)
    return sizeof(CsrHeader) + (n_refs + 1) * 8 + nb * sizeof(CsrBlock) + (n_hashes + nb + 2) * 8;
}

}  // extern "C"
// The packer itself, over sketches that need not lie back to back: parts[j] = the first hash of sketch j (offsets[j + 1] -
// offsets[j] of them).  yh_csr_pack: parts[j] = values + offsets[j]; yh_sig_batch_pack (yh_sigread.hip): the parsed files' own
// vectors -- `yacht train` goes from the archive to the packed database without a CSR in between (round 6).
int yh_csr_pack_parts(const u64* const* parts, const u64* offsets, u64 n_refs, void* packed, u64 cap_bytes, u64* packed_bytes, int threads) {
    if (!packed_bytes || !offsets) { yh_set_error("yh_csr_pack: null argument"); return YH_ERR_INVALID_ARG; }
    if (n_refs > 0x7ffffff0ull) { yh_set_error("too many references"); return YH_ERR_INVALID_ARG; }
    if (offsets[0] != 0) { yh_set_error("offsets[0] must be 0"); return YH_ERR_INVALID_ARG; }
    const u64 N = n_refs, H = offsets[N];
    if (H && !parts) { yh_set_error("values is null"); return YH_ERR_INVALID_ARG; }
    std::vector<u64> fb(N + 1);
    u64 nb = 0;
    for (u64 j = 0; j < N; ++j) {
        if (offsets[j + 1] < offsets[j]) { yh_set_error("offsets are not monotone"); return YH_ERR_UNSORTED; }
        fb[j] = nb;
        nb += (offsets[j + 1] - offsets[j] + PACK_BLOCK - 1) / PACK_BLOCK;
    }
    fb[N] = nb;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = (unsigned)std::max<u64>(1, std::min<u64>(threads > 0 ? (u64)threads : std::min<unsigned>(hw ? hw : 1, 16), N / 64 + 1));
    std::vector<CsrBlock> tab(nb);
    std::atomic<int> unsorted{0};
    std::vector<u64> tmax(T, 0);
    auto run_ranges = [&](auto&& fn) {  // sketches [j0, j1) per thread, cut by hash count
        if (T <= 1) { fn(0u, (u64)0, N); return; }
        std::vector<std::thread> th;
        u64 j0 = 0;
        for (unsigned t = 0; t < T; ++t) {
            u64 j1 = j0;
            const u64 target = H / T * (t + 1);
            while (j1 < N && (t + 1 == T || offsets[j1 + 1] <= target)) ++j1;
            th.emplace_back(fn, t, j0, j1);
            j0 = j1;
        }
        for (auto& x : th) x.join();
    };
    // pass 1: ordering, the width of every block, the largest hash
    run_ranges([&](unsigned t, u64 j0, u64 j1) {
        u64 mx = 0;
        bool bad = false;
        for (u64 j = j0; j < j1; ++j) {
            const u64 a = offsets[j], e = offsets[j + 1];
            const u64* const h = e > a ? parts[j] - a : nullptr;  // (indexed by the CSR position, as if the sketches lay back to back)
            if (e > a) mx = std::max(mx, h[e - 1]);
            for (u64 first = a, b = fb[j]; first < e; first += PACK_BLOCK, ++b) {
                const u32 cnt = (u32)std::min<u64>(PACK_BLOCK, e - first);
                bad |= first > a && !(h[first - 1] < h[first]);
                u64 widest = 0;
                for (u32 i = 1; i < cnt; ++i) {
                    bad |= !(h[first + i - 1] < h[first + i]);
                    widest |= h[first + i] - h[first + i - 1] - 1ull;
                }
                tab[b] = CsrBlock{h[first], 0ull, bits_of(widest), 0u};
            }
        }
        tmax[t] = mx;
        if (bad) unsorted.store(1, std::memory_order_relaxed);
    });
    if (unsorted.load()) { yh_set_error("a reference sketch is not strictly ascending"); return YH_ERR_UNSORTED; }
    u64 words = 0;
    for (u64 j = 0; j < N; ++j)
        for (u64 first = offsets[j], b = fb[j]; first < offsets[j + 1]; first += PACK_BLOCK, ++b) {
            tab[b].word_off = words;
            words += block_words((u32)std::min<u64>(PACK_BLOCK, offsets[j + 1] - first), tab[b].width);
        }
    words += 1;  // the spare word
    const u64 need = sizeof(CsrHeader) + (N + 1) * 8 + nb * sizeof(CsrBlock) + words * 8;
    *packed_bytes = need;
    if (!packed || cap_bytes < need) {
        if (!packed && cap_bytes == 0) return YH_OK;  // sizing call
        yh_set_error("yh_csr_pack: buffer of %llu bytes, %llu needed", (u64)cap_bytes, need);
        return YH_ERR_CAPACITY;
    }
    if (reinterpret_cast<uintptr_t>(packed) & 7u) { yh_set_error("yh_csr_pack: the buffer must be 8-byte aligned"); return YH_ERR_INVALID_ARG; }
    u64 mx = 0;
    for (const u64 v : tmax) mx = std::max(mx, v);
    CsrHeader hd{CSR_MAGIC, PACK_BLOCK, N, H, nb, words, mx, {0, 0}};
    char* p = (char*)packed;
    memcpy(p, &hd, sizeof(hd));
    memcpy(p + sizeof(hd), offsets, (N + 1) * 8);
    if (nb) memcpy(p + sizeof(hd) + (N + 1) * 8, tab.data(), nb * sizeof(CsrBlock));
    u64* payload = reinterpret_cast<u64*>(p + sizeof(hd) + (N + 1) * 8 + nb * sizeof(CsrBlock));
    // pass 2: the gaps (blocks own disjoint words)
    run_ranges([&](unsigned, u64 j0, u64 j1) {
        for (u64 j = j0; j < j1; ++j) {
            const u64* const h = offsets[j + 1] > offsets[j] ? parts[j] - offsets[j] : nullptr;
            for (u64 first = offsets[j], b = fb[j]; first < offsets[j + 1]; first += PACK_BLOCK, ++b) {
                const u32 cnt = (u32)std::min<u64>(PACK_BLOCK, offsets[j + 1] - first);
                const u32 w = tab[b].width;
                u64* out = payload + tab[b].word_off;
                const u64 nw = block_words(cnt, w);
                for (u64 k = 0; k < nw; ++k) out[k] = 0;
                if (!w) continue;
                u64 bit = 0;
                for (u32 i = 1; i < cnt; ++i, bit += w) {
                    const u64 v = h[first + i] - h[first + i - 1] - 1ull;
                    const u32 sh = (u32)(bit & 63u);
                    out[bit >> 6] |= v << sh;
                    if (sh + w > 64) out[(bit >> 6) + 1] |= v >> (64 - sh);
                }
            }
        }
    });
    payload[words - 1] = 0;
    return YH_OK;
}
extern "C" {
int yh_csr_pack(const uint64_t* values, const uint64_t* offsets, uint64_t n_refs, void* packed, uint64_t cap_bytes, uint64_t* packed_bytes,
                int threads) {
    if (!packed_bytes || !offsets) { yh_set_error("yh_csr_pack: null argument"); return YH_ERR_INVALID_ARG; }
    if (n_refs > 0x7ffffff0ull) { yh_set_error("too many references"); return YH_ERR_INVALID_ARG; }
    if (offsets[n_refs] && !values) { yh_set_error("values is null"); return YH_ERR_INVALID_ARG; }
    try {
        std::vector<const u64*> parts(n_refs);
        for (u64 j = 0; j < n_refs; ++j) parts[j] = (const u64*)values + offsets[j];
        return yh_csr_pack_parts(parts.data(), (const u64*)offsets, n_refs, packed, cap_bytes, (u64*)packed_bytes, threads);
    } catch (const std::bad_alloc&) {
        yh_set_error("yh_csr_pack: out of host memory");
        return YH_ERR_OOM;
    }
}

// Rows of a packed CSR, in the given order, as a packed CSR of their own (`yacht train` writes the SELECTED references for
// `yacht run` from the blob it uploaded): the rows' block entries with their word offsets re-based, their payload words copied
// as they are; the largest hash from the rows' last blocks.  Two-call sizing as yh_csr_pack.
int yh_csr_subset(const void* packed, uint64_t packed_bytes, const uint64_t* rows, uint64_t n_rows, void* out, uint64_t cap_bytes,
                  uint64_t* out_bytes) {
    if (!out_bytes || (n_rows && !rows)) { yh_set_error("yh_csr_subset: null argument"); return YH_ERR_INVALID_ARG; }
    try {
        YhPackedCsr v;
        YH_TRY(yh_csr_view(packed, packed_bytes, &v));
        const CsrBlock* tab = reinterpret_cast<const CsrBlock*>(v.tab);
        std::vector<u64> off(n_rows + 1, 0), fb(n_rows + 1, 0);
        u64 words = 0;
        for (u64 k = 0; k < n_rows; ++k) {
            const u64 j = rows[k];
            if (j >= v.n_refs) { yh_set_error("yh_csr_subset: row %llu of %llu", j, v.n_refs); return YH_ERR_INVALID_ARG; }
            const u64 len = v.offsets[j + 1] - v.offsets[j], b0 = v.first_block[j], b1 = v.first_block[j + 1];
            off[k + 1] = off[k] + len;
            fb[k + 1] = fb[k] + (b1 - b0);
            words += yh_csr_block_word_off(&v, b1) - yh_csr_block_word_off(&v, b0);  // (the view checked: the blocks' words lie back to back)
        }
        words += 1;  // the spare word
        const u64 nb = fb[n_rows];
        const u64 need = sizeof(CsrHeader) + (n_rows + 1) * 8 + nb * sizeof(CsrBlock) + words * 8;
        *out_bytes = need;
        if (!out || cap_bytes < need) {
            if (!out && cap_bytes == 0) return YH_OK;
            yh_set_error("yh_csr_subset: buffer of %llu bytes, %llu needed", (u64)cap_bytes, need);
            return YH_ERR_CAPACITY;
        }
        if (reinterpret_cast<uintptr_t>(out) & 7u) { yh_set_error("yh_csr_subset: the buffer must be 8-byte aligned"); return YH_ERR_INVALID_ARG; }
        char* p = (char*)out;
        memcpy(p + sizeof(CsrHeader), off.data(), (n_rows + 1) * 8);
        CsrBlock* otab = reinterpret_cast<CsrBlock*>(p + sizeof(CsrHeader) + (n_rows + 1) * 8);
        u64* opay = reinterpret_cast<u64*>(reinterpret_cast<char*>(otab) + nb * sizeof(CsrBlock));
        u64 w_at = 0, mx = 0;
        for (u64 k = 0; k < n_rows; ++k) {
            const u64 j = rows[k], b0 = v.first_block[j], b1 = v.first_block[j + 1];
            if (b1 == b0) continue;
            const u64 w0 = yh_csr_block_word_off(&v, b0), w1 = yh_csr_block_word_off(&v, b1);
            for (u64 b = b0; b < b1; ++b) {
                CsrBlock blk;
                memcpy(&blk, &tab[b], sizeof(blk));
                blk.word_off = blk.word_off - w0 + w_at;
                memcpy(&otab[fb[k] + (b - b0)], &blk, sizeof(blk));
            }
            if (w1 > w0) memcpy(opay + w_at, v.payload + w0, (w1 - w0) * 8);
            w_at += w1 - w0;
            // the sketch's last hash: its last block, decoded
            CsrBlock last;
            memcpy(&last, &tab[b1 - 1], sizeof(last));
            const u32 cnt = (u32)(v.offsets[j + 1] - v.offsets[j] - (b1 - 1 - b0) * PACK_BLOCK);
            u64 cur = last.base, bit = 0;
            for (u32 i = 1; i < cnt; ++i, bit += last.width) {
                u64 g = 0;
                if (last.width) {
                    const u32 sh = (u32)(bit & 63u);
                    g = v.payload[last.word_off + (bit >> 6)] >> sh;
                    if (sh + last.width > 64) g |= v.payload[last.word_off + (bit >> 6) + 1] << (64 - sh);
                    if (last.width < 64) g &= (1ull << last.width) - 1ull;
                }
                cur += g + 1ull;
            }
            mx = std::max(mx, cur);
        }
        opay[w_at] = 0;  // the spare word
        CsrHeader hd{CSR_MAGIC, PACK_BLOCK, n_rows, off[n_rows], nb, words, mx, {0, 0}};
        memcpy(p, &hd, sizeof(hd));
        return YH_OK;
    } catch (const std::bad_alloc&) {
        yh_set_error("yh_csr_subset: out of host memory");
        return YH_ERR_OOM;
    }
}

int yh_csr_unpack(const void* packed, uint64_t packed_bytes, uint64_t* values_out, uint64_t cap_hashes, uint64_t* offsets_out,
                  uint64_t cap_refs, uint64_t* n_hashes, uint64_t* n_refs) {
    if (!n_hashes || !n_refs) { yh_set_error("yh_csr_unpack: null argument"); return YH_ERR_INVALID_ARG; }
    YhPackedCsr v;
    YH_TRY(yh_csr_view(packed, packed_bytes, &v));
    *n_hashes = v.n_hashes;
    *n_refs = v.n_refs;
    if (!values_out && !offsets_out && cap_hashes == 0 && cap_refs == 0) return YH_OK;  // sizing call
    if (cap_hashes < v.n_hashes || cap_refs < v.n_refs || !offsets_out || (v.n_hashes && !values_out)) {
        yh_set_error("yh_csr_unpack: room for %llu hashes of %llu references, %llu of %llu needed", (u64)cap_hashes, (u64)cap_refs, v.n_hashes, v.n_refs);
        return YH_ERR_CAPACITY;
    }
    memcpy(offsets_out, v.offsets, (v.n_refs + 1) * 8);
    for (u64 j = 0; j < v.n_refs; ++j) {
        u64 prev = 0;
        for (u64 first = v.offsets[j], b = v.first_block[j]; first < v.offsets[j + 1]; first += PACK_BLOCK, ++b) {
            CsrBlock blk;
            memcpy(&blk, (const char*)v.tab + b * sizeof(CsrBlock), sizeof(blk));
            const u32 cnt = (u32)std::min<u64>(PACK_BLOCK, v.offsets[j + 1] - first);
            if (blk.width > 64 || blk.word_off > v.payload_words || block_words(cnt, blk.width) + 1 > v.payload_words - blk.word_off) {
                yh_set_error("packed CSR: block %llu points outside the payload", b);
                return YH_ERR_INVALID_ARG;
            }
            u64 cur = blk.base;
            if (first > v.offsets[j] && !(prev < cur)) { yh_set_error("packed CSR: blocks of a sketch are not ascending"); return YH_ERR_UNSORTED; }
            values_out[first] = cur;
            u64 bit = 0;
            for (u32 i = 1; i < cnt; ++i, bit += blk.width) {
                u64 g = 0;
                if (blk.width) {
                    const u32 sh = (u32)(bit & 63u);
                    g = v.payload[blk.word_off + (bit >> 6)] >> sh;
                    if (sh + blk.width > 64) g |= v.payload[blk.word_off + (bit >> 6) + 1] << (64 - sh);
                    if (blk.width < 64) g &= (1ull << blk.width) - 1ull;
                }
                const u64 next = cur + g + 1ull;
                if (!(cur < next)) { yh_set_error("packed CSR: a gap wraps around 2^64"); return YH_ERR_UNSORTED; }
                cur = next;
                values_out[first + i] = cur;
            }
            prev = cur;
        }
    }
    return YH_OK;
}

}  // extern "C"
