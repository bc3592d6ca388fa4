// yh_sketch.hip — DNA FracMinHash sketching on the device.
//
// Replaces the `sourmash sketch dna -p k=K,scaled=S,abund` subprocess of the reference
// (src/yacht/sketch_ref_genomes.py:25,61, sketch_sample.py:32,49) for the hashing part: every
// length-K window made only of A/C/G/T (either case) -> canonical form (lexicographic minimum of
// the k-mer and its reverse complement) -> first 64 bits of MurmurHash3_x64_128 (public domain
// algorithm by Austin Appleby) of its upper-case ASCII bytes, seed 42 -> kept iff <= max_hash.
// One lane per window; the kept hashes (about 1/scaled of the windows) are appended unsorted, with
// duplicates; the caller sorts and counts them (abundances).  ALU-bound, integer only.
#include "yh_common.h"

#include <algorithm>

namespace {

constexpr int SK_THREADS = 256;
constexpr int SK_ITEMS = 8;      // windows per lane
constexpr int SK_LCAP = 2048;    // kept hashes parked in LDS per workgroup before the flush

__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ u64 fmix64(u64 k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// 0..3 for A,C,G,T (either case) in ALPHABETICAL order, 4 otherwise
__device__ __forceinline__ u32 base_code(u8 c) {
    const u32 u = c & 0xDFu;
    const bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
    const u32 x = (u >> 1) & 3u;  // A 0, C 1, T 2, G 3
    return ok ? (x ^ (x >> 1)) : 4u;  // -> A 0, C 1, G 2, T 3
}
__device__ __forceinline__ u8 code_ascii(u32 code) { return (u8)((0x54474341u >> (8 * code)) & 0xffu); }

__global__ void __launch_bounds__(SK_THREADS) k_sketch_dna(const u8* __restrict__ seq, u64 n, u32 k, u64 seed,
                                                           u64 max_hash, u64 cap, u64* __restrict__ out,
                                                           u64* __restrict__ out_count) {
    __shared__ u64 lbuf[SK_LCAP];
    __shared__ u32 lfill;
    __shared__ u64 gbase;
    if (threadIdx.x == 0) lfill = 0;
    __syncthreads();
    const u64 n_win = n - k + 1;
    const u64 block_first = (u64)blockIdx.x * SK_THREADS * SK_ITEMS;
    for (int it = 0; it < SK_ITEMS; ++it) {
        const u64 i = block_first + (u64)it * SK_THREADS + threadIdx.x;
        if (i >= n_win) continue;
        const u8* s = seq + i;
        // validity + orientation: the first position where the k-mer and its reverse complement
        // differ decides which one is canonical
        bool valid = true;
        int use_rc = -1;  // -1 undecided (so far equal)
        for (u32 j = 0; j < k; ++j) {
            const u32 a = base_code(s[j]);
            if (a > 3u) { valid = false; break; }
            if (use_rc < 0) {
                const u32 b = base_code(s[k - 1 - j]);
                if (b <= 3u && (3u - b) != a) use_rc = ((3u - b) < a) ? 1 : 0;
            }
        }
        if (!valid) continue;
        const bool rc = use_rc == 1;
        auto byte_at = [&](u32 j) -> u64 {
            const u32 c = rc ? 3u - base_code(s[k - 1 - j]) : base_code(s[j]);
            return (u64)code_ascii(c);
        };
        // MurmurHash3_x64_128, first word
        u64 h1 = seed, h2 = seed;
        const u64 c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
        const u32 nblocks = k / 16;
        for (u32 blk = 0; blk < nblocks; ++blk) {
            u64 k1 = 0, k2 = 0;
            for (u32 t = 0; t < 8; ++t) {
                k1 |= byte_at(16 * blk + t) << (8 * t);
                k2 |= byte_at(16 * blk + 8 + t) << (8 * t);
            }
            k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
            h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ull;
            k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
            h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ull;
        }
        const u32 tail = 16 * nblocks, rem = k - tail;
        if (rem > 8) {
            u64 k2 = 0;
            for (u32 t = 8; t < rem; ++t) k2 |= byte_at(tail + t) << (8 * (t - 8));
            k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        }
        if (rem > 0) {
            u64 k1 = 0;
            for (u32 t = 0; t < (rem < 8 ? rem : 8u); ++t) k1 |= byte_at(tail + t) << (8 * t);
            k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        }
        h1 ^= (u64)k; h2 ^= (u64)k;
        h1 += h2; h2 += h1;
        h1 = fmix64(h1); h2 = fmix64(h2);
        h1 += h2;
        if (h1 <= max_hash) {
            const u32 slot = atomicAdd(&lfill, 1u);
            if (slot < (u32)SK_LCAP) {
                lbuf[slot] = h1;
            } else {  // scaled close to 1: more kept hashes than the LDS list holds
                const u64 g = atomicAdd((unsigned long long*)out_count, 1ull);
                if (g < cap) out[g] = h1;
            }
        }
    }
    __syncthreads();
    const u32 f = min(lfill, (u32)SK_LCAP);
    if (f) {
        if (threadIdx.x == 0) gbase = atomicAdd((unsigned long long*)out_count, (unsigned long long)f);
        __syncthreads();
        for (u32 e = threadIdx.x; e < f; e += SK_THREADS)
            if (gbase + e < cap) out[gbase + e] = lbuf[e];
    }
}

}  // namespace

extern "C" int yh_sketch_dna(const uint8_t* seq, uint64_t n_bytes, int ksize, uint64_t seed, uint64_t max_hash,
                             int device_id, uint64_t cap, uint64_t* hashes_out, uint64_t* n_out) {
    if (!n_out || (cap && !hashes_out) || (n_bytes && !seq)) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (ksize < 1 || ksize > 255) { yh_set_error("ksize must be in [1, 255]"); return YH_ERR_INVALID_ARG; }
    *n_out = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        yh_set_error("no HIP device available (libyacht_hip has no CPU fallback)");
        return YH_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= ndev) { yh_set_error("device_id %d out of range", device_id); return YH_ERR_NO_DEVICE; }
    YH_HIP(hipSetDevice(device_id));
    if (n_bytes < (uint64_t)ksize) return YH_OK;
    u8* d_seq = nullptr;
    u64 *d_out = nullptr, *d_cnt = nullptr;
    int rc = YH_OK;
#define SK_HIP(call)                                                              \
    if (rc == YH_OK) {                                                            \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            yh_set_error("%s failed: %s", #call, hipGetErrorString(e__));         \
            rc = (e__ == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;          \
        }                                                                         \
    }
    SK_HIP(hipMalloc((void**)&d_seq, n_bytes));
    SK_HIP(hipMalloc((void**)&d_out, std::max<u64>(cap, 1) * sizeof(u64)));
    SK_HIP(hipMalloc((void**)&d_cnt, sizeof(u64)));
    SK_HIP(hipMemcpy(d_seq, seq, n_bytes, hipMemcpyHostToDevice));
    SK_HIP(hipMemset(d_cnt, 0, sizeof(u64)));
    if (rc == YH_OK) {
        const u64 n_win = n_bytes - (u64)ksize + 1;
        const u64 per_block = (u64)SK_THREADS * SK_ITEMS;
        const u64 blocks = (n_win + per_block - 1) / per_block;
        if (blocks > 0x7fffffffull) { yh_set_error("sequence too long for one call"); rc = YH_ERR_INVALID_ARG; }
        else k_sketch_dna<<<(u32)blocks, SK_THREADS>>>(d_seq, n_bytes, (u32)ksize, seed, max_hash, cap, d_out, d_cnt);
    }
    SK_HIP(hipGetLastError());
    u64 total = 0;
    SK_HIP(hipMemcpy(&total, d_cnt, sizeof(u64), hipMemcpyDeviceToHost));
    if (rc == YH_OK) {
        *n_out = total;
        if (total > cap) {
            if (cap) { yh_set_error("hash buffer holds %llu entries, %llu needed", (u64)cap, total); rc = YH_ERR_CAPACITY; }
        } else if (total) {
            SK_HIP(hipMemcpy(hashes_out, d_out, total * sizeof(u64), hipMemcpyDeviceToHost));
        }
    }
#undef SK_HIP
    (void)hipFree(d_seq);
    (void)hipFree(d_out);
    (void)hipFree(d_cnt);
    return rc;
}
