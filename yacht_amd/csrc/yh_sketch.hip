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
#include <mutex>

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

// (win0: the first window of this launch -- a sequence that is still arriving is hashed piece by piece; n: bytes present)
__global__ void __launch_bounds__(SK_THREADS) k_sketch_dna(const u8* __restrict__ seq, u64 n, u32 k, u64 seed,
                                                           u64 max_hash, u64 cap, u64* __restrict__ out,
                                                           u64* __restrict__ out_count, u64 win0) {
    __shared__ u64 lbuf[SK_LCAP];
    __shared__ u32 lfill;
    __shared__ u64 gbase;
    if (threadIdx.x == 0) lfill = 0;
    __syncthreads();
    const u64 n_win = n - k + 1;
    const u64 block_first = win0 + (u64)blockIdx.x * SK_THREADS * SK_ITEMS;
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

// ---- k <= 64: the k-mer as a 2-bit string in one 64- or 128-bit word ---------------------------------------------------------
// The first kernel above reads every window's k bytes three times through L1 (23 G bases/s at k = 31: below what PCIe
// delivers).  Here a workgroup stages its 8 192 + k - 1 bases ONCE, coalesced, as 2-bit codes (16 bases per word, A C G T
// = 0 1 2 3) with one "not a base" bit each; a lane owns RL_RUN = 32 consecutive windows = 64 staged bases in four
// registers.  With the string packed little-endian (base j at bits 2j), window t is bits [2t, 2t + 2k) of it, its
// big-endian packing (the one whose numeric order is the lexicographic order) is kept by shifting one base in per window,
// and the reverse complement costs nothing:  BE(rc) = ~LE(fwd),  LE(rc) = ~BE(fwd)  (complement = 3 - code = ~code).
// The canonical k-mer's ASCII bytes, which MurmurHash3 wants, are spread out of the codes arithmetically, 8 bytes a time.
constexpr int RL_THREADS = 256;
constexpr int RL_RUN = 32;                         // windows per lane
constexpr int RL_WIN = RL_THREADS * RL_RUN;        // windows per workgroup
constexpr int RL_UNITS = RL_WIN / 16 + 4;          // staged 16-base units (the last lane reads six from its first)

// four bytes -> their 2-bit codes in the low bits of each byte, and bit 7 of a byte set when it is not A/C/G/T (either case)
__device__ __forceinline__ void codes_of4(u32 w, u32& codes, u32& bad) {
    const u32 x = w & 0xDFDFDFDFu;
    const u32 t = (x >> 1) & 0x03030303u;               // A 0, C 1, T 2, G 3
    const u32 c = t ^ ((t >> 1) & 0x01010101u);         // A 0, C 1, G 2, T 3
    const u32 lo = c & 0x01010101u, hi = (c >> 1) & 0x01010101u;
    const u32 expect = 0x41414141u + (lo & ~hi) * 2u + (hi & ~lo) * 6u + (lo & hi) * 0x13u;  // 'A' 'C' 'G' 'T'
    const u32 diff = x ^ expect;
    bad = (((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u;
    codes = c;
}
// 8 codes (16 bits, base j at bits 2j) -> 8 ASCII bytes (base j in byte j)
__device__ __forceinline__ u64 ascii_of8(u32 v16) {
    u64 x = v16;
    x = (x | (x << 24)) & 0x000000FF000000FFull;
    x = (x | (x << 12)) & 0x000F000F000F000Full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    const u64 lo = x & 0x0101010101010101ull, hi = (x >> 1) & 0x0101010101010101ull;
    return 0x4141414141414141ull + (lo & ~hi) * 2ull + (hi & ~lo) * 6ull + (lo & hi) * 0x13ull;
}

typedef unsigned __int128 u128;
template <int KW> struct RollWord;
template <> struct RollWord<1> { typedef u64 type; };
template <> struct RollWord<2> { typedef u128 type; };
__device__ __forceinline__ u32 top_bit_plus1(u64 x) { return x ? 64u - (u32)__clzll((long long)x) : 0u; }
__device__ __forceinline__ u32 top_bit_plus1(u128 x) {
    const u64 h = (u64)(x >> 64);
    return h ? 64u + top_bit_plus1(h) : top_bit_plus1((u64)x);
}

// KW = 1: k <= 32, the k-mer in one 64-bit word; KW = 2: k <= 64, in a 128-bit one (the same code on a wider word)
template <int KW>
__global__ void __launch_bounds__(RL_THREADS) k_sketch_dna_roll(const u8* __restrict__ seq, u64 n, u32 k, u64 seed,
                                                                u64 max_hash, u64 cap, u64* __restrict__ out,
                                                                u64* __restrict__ out_count, u64 win0) {
    typedef typename RollWord<KW>::type word;
    constexpr u32 BITS = 64u * KW;      // of a word = 32 KW bases
    constexpr int NW = 4 * KW;          // 8-byte ASCII words of a k-mer
    __shared__ u32 lcode[RL_UNITS];
    __shared__ u32 lbad[RL_UNITS];
    __shared__ u64 lbuf[SK_LCAP];
    __shared__ u32 lfill;
    __shared__ u64 gbase;
    if (threadIdx.x == 0) lfill = 0;
    const u64 n_win = n - k + 1;
    const u64 B0 = win0 + (u64)blockIdx.x * RL_WIN;  // first base (= first window) of the workgroup
    for (u32 u = threadIdx.x; u < (u32)RL_UNITS; u += RL_THREADS) {
        const u64 p = B0 + 16ull * u;
        u32 w[4] = {0u, 0u, 0u, 0u};  // (a zero byte is not a base)
        if (p + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4*>(seq + p);  // (B0 and 16 u are multiples of 16)
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else if (p < n) {
            for (u32 j = 0; j < 16 && p + j < n; ++j) w[j >> 2] |= (u32)seq[p + j] << (8 * (j & 3));
        }
        u32 codes = 0, bad = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32 c, b;
            codes_of4(w[q], c, b);
            codes |= (((c * 0x41041u) >> 18) & 0xffu) << (8 * q);           // 4 codes -> 8 bits
            bad |= ((((b >> 7) * 0x00204081u) >> 21) & 0xfu) << (4 * q);    // 4 flags -> 4 bits
        }
        lcode[u] = codes;
        lbad[u] = bad;
    }
    __syncthreads();
    // the lane's 32 windows reach 32 + k - 1 <= 95 bases: units 2 l .. 2 l + 5 (six of sixteen bases; four when k <= 32)
    const u32 l2 = 2u * threadIdx.x;
    word P0 = 0, P1 = 0, bad = 0;
#pragma unroll
    for (int q = 0; q < 2 * KW; ++q) {
        P0 |= (word)lcode[l2 + q] << (32 * q);
        bad |= (word)lbad[l2 + q] << (16 * q);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        P1 |= (word)lcode[l2 + 2 * KW + q] << (32 * q);
        bad |= (word)lbad[l2 + 2 * KW + q] << (16 * (2 * KW + q));
    }
    const word one = 1;
    const word mask = (2 * k == BITS) ? ~(word)0 : ((one << (2 * k)) - one);
    const u64 w0 = B0 + (u64)threadIdx.x * RL_RUN;  // the lane's first window
    // big-endian packing of the first k - 1 bases (all in P0); the first window that no bad base among them reaches
    word fbe = 0;
    for (u32 j = 0; j + 1 < k; ++j) fbe = (fbe << 2) | ((P0 >> (2 * j)) & (word)3);
    const word head_bad = (k > 1) ? (bad & ((one << (k - 1)) - one)) : (word)0;
    u32 first_ok = top_bit_plus1(head_bad);
    const u64 c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    const u32 nblocks = k / 16, rem = k & 15u;
    for (u32 t = 0; t < (u32)RL_RUN; ++t) {
        const u32 p = t + k - 1;                       // the base this window adds
        if ((u32)(bad >> p) & 1u) first_ok = p + 1;
        const word fle = (t ? ((P0 >> (2 * t)) | (P1 << (BITS - 2 * t))) : P0) & mask;
        fbe = ((fbe << 2) | ((fle >> (2 * (k - 1))) & (word)3)) & mask;
        if (t < first_ok || w0 + t >= n_win) continue;
        const word rbe = ~fle & mask;                  // BE(reverse complement)
        const word canon = (rbe < fbe) ? (~fbe & mask) : fle;   // LE packing of the canonical k-mer
        // its ASCII bytes, 8 per word, bytes from k on zero
        u64 W[NW];
#pragma unroll
        for (int m = 0; m < NW; ++m) {
            const int cnt = (int)k - 8 * m;            // bytes of this word
            u64 a = ascii_of8((u32)(canon >> (16 * m)) & 0xffffu);
            if (cnt <= 0) a = 0;
            else if (cnt < 8) a &= (1ull << (8 * cnt)) - 1ull;
            W[m] = a;
        }
        u64 h1 = seed, h2 = seed;
#pragma unroll
        for (int blk = 0; blk < NW / 2; ++blk) {
            if ((u32)blk < nblocks) {
                u64 k1 = W[2 * blk], k2 = W[2 * blk + 1];
                k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
                h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ull;
                k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
                h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ull;
            }
        }
        // the tail: the words behind the last whole block (zero when k is a multiple of 16)
        u64 t1 = 0, t2 = 0;
#pragma unroll
        for (int blk = 0; blk < NW / 2; ++blk)
            if ((u32)blk == nblocks) { t1 = W[2 * blk]; t2 = W[2 * blk + 1]; }
        if (rem > 8) { t2 *= c2; t2 = rotl64(t2, 33); t2 *= c1; h2 ^= t2; }
        if (rem > 0) { t1 *= c1; t1 = rotl64(t1, 31); t1 *= c2; h1 ^= t1; }
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
        for (u32 e = threadIdx.x; e < f; e += RL_THREADS)
            if (gbase + e < cap) out[gbase + e] = lbuf[e];
    }
}

// The launch both entry points share: the windows [win0, win_end) of a sequence of which n_bytes are present (win0 a
// multiple of SK_WIN_ALIGN); k <= 64 through the 2-bit kernel (YH_SKETCH_BYTES=1 behind the tuning gate: the first one).
constexpr u64 SK_WIN_ALIGN = RL_WIN;  // (a multiple of both kernels' windows per workgroup)
static_assert(RL_WIN % (SK_THREADS * SK_ITEMS) == 0, "one alignment for both kernels");
static int launch_sketch(const u8* d_seq, u64 n_bytes, u32 k, u64 seed, u64 max_hash, u64 cap, u64* d_out, u64* d_cnt, hipStream_t st,
                         u64 win0, u64 win_end) {
    if (win_end <= win0) return YH_OK;
    static const bool bytes_only = [] { const char* e = yh_tune_env("YH_SKETCH_BYTES"); return e && e[0] == '1'; }();
    const bool roll = k <= 64 && !bytes_only && (reinterpret_cast<uintptr_t>(d_seq) & 15u) == 0;
    const u64 per_block = roll ? (u64)RL_WIN : (u64)SK_THREADS * SK_ITEMS;
    const u64 blocks = (win_end - win0 + per_block - 1) / per_block;
    if (blocks > 0x7fffffffull) { yh_set_error("sequence too long for one call"); return YH_ERR_INVALID_ARG; }
    if (roll && k <= 32) k_sketch_dna_roll<1><<<(u32)blocks, RL_THREADS, 0, st>>>(d_seq, n_bytes, k, seed, max_hash, cap, d_out, d_cnt, win0);
    else if (roll) k_sketch_dna_roll<2><<<(u32)blocks, RL_THREADS, 0, st>>>(d_seq, n_bytes, k, seed, max_hash, cap, d_out, d_cnt, win0);
    else k_sketch_dna<<<(u32)blocks, SK_THREADS, 0, st>>>(d_seq, n_bytes, k, seed, max_hash, cap, d_out, d_cnt, win0);
    YH_HIP(hipGetLastError());
    return YH_OK;
}

}  // namespace

// device buffers, enqueued on `stream`: *d_count is zeroed first and receives the number of kept hashes (which may exceed
// cap: then only the first cap were stored)
extern "C" int yh_sketch_dna_device(const uint8_t* d_seq, uint64_t n_bytes, int ksize, uint64_t seed, uint64_t max_hash,
                                    uint64_t cap, uint64_t* d_hashes_out, uint64_t* d_count, void* stream) {
    if (!d_count || (cap && !d_hashes_out) || (n_bytes && !d_seq)) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (ksize < 1 || ksize > 255) { yh_set_error("ksize must be in [1, 255]"); return YH_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    YH_HIP(hipMemsetAsync(d_count, 0, sizeof(u64), st));
    if (n_bytes < (uint64_t)ksize) return YH_OK;
    YH_TRY(launch_sketch(d_seq, n_bytes, (u32)ksize, seed, max_hash, cap, (u64*)d_hashes_out, (u64*)d_count, st, 0, n_bytes - (u64)ksize + 1));
    return YH_OK;
}

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
    // The sequence goes up in pieces of 32 MiB on one stream while the windows that are complete are hashed on another
    // (the kernel runs at ~2.5 x the bus: the call costs the upload plus the last piece); buffers, streams and events are
    // kept per device between calls (up to 1 GiB of sequence), one call per process at a time.
    struct SkCtx { u8* d_seq = nullptr; u64 seq_cap = 0; u64* d_out = nullptr; u64 out_cap = 0; u64* d_cnt = nullptr;
                   hipStream_t s_copy = nullptr, s_comp = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; };
    static std::mutex mu;
    static SkCtx ctxs[64];
    std::lock_guard<std::mutex> lock(mu);
    SkCtx& c = ctxs[device_id & 63];
    int rc = YH_OK;
#define SK_HIP(call)                                                              \
    if (rc == YH_OK) {                                                            \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            yh_set_error("%s failed: %s", #call, hipGetErrorString(e__));         \
            rc = (e__ == hipErrorOutOfMemory) ? YH_ERR_OOM : YH_ERR_HIP;          \
        }                                                                         \
    }
    if (!c.s_copy) {
        SK_HIP(hipStreamCreateWithFlags(&c.s_copy, hipStreamNonBlocking));
        SK_HIP(hipStreamCreateWithFlags(&c.s_comp, hipStreamNonBlocking));
        SK_HIP(hipEventCreateWithFlags(&c.ev[0], hipEventDisableTiming));
        SK_HIP(hipEventCreateWithFlags(&c.ev[1], hipEventDisableTiming));
        SK_HIP(hipMalloc((void**)&c.d_cnt, sizeof(u64)));
    }
    if (rc == YH_OK && c.seq_cap < n_bytes) {
        if (c.d_seq) (void)hipFree(c.d_seq);
        c.d_seq = nullptr; c.seq_cap = 0;
        SK_HIP(hipMalloc((void**)&c.d_seq, n_bytes + 64));
        if (rc == YH_OK) c.seq_cap = n_bytes;
    }
    if (rc == YH_OK && c.out_cap < std::max<u64>(cap, 1)) {
        if (c.d_out) (void)hipFree(c.d_out);
        c.d_out = nullptr; c.out_cap = 0;
        SK_HIP(hipMalloc((void**)&c.d_out, std::max<u64>(cap, 1) * sizeof(u64)));
        if (rc == YH_OK) c.out_cap = std::max<u64>(cap, 1);
    }
    SK_HIP(hipMemsetAsync(c.d_cnt, 0, sizeof(u64), c.s_comp));
    const u64 n_win = n_bytes - (u64)ksize + 1;
    const u64 piece = 32ull << 20;
    u64 done_win = 0;
    int flip = 0;
    for (u64 p0 = 0; p0 < n_bytes && rc == YH_OK; p0 += piece, flip ^= 1) {
        const u64 p1 = std::min<u64>(n_bytes, p0 + piece);
        SK_HIP(hipMemcpyAsync(c.d_seq + p0, seq + p0, p1 - p0, hipMemcpyHostToDevice, c.s_copy));  // (pageable: returns when the piece has left the host)
        SK_HIP(hipEventRecord(c.ev[flip], c.s_copy));
        SK_HIP(hipStreamWaitEvent(c.s_comp, c.ev[flip], 0));
        // windows whose bytes are all there, in whole workgroups (the last piece: all that is left)
        u64 avail = p1 >= (u64)ksize ? p1 - (u64)ksize + 1 : 0;
        u64 upto = (p1 == n_bytes) ? n_win : (avail / SK_WIN_ALIGN) * SK_WIN_ALIGN;
        if (rc == YH_OK && upto > done_win) {
            rc = launch_sketch(c.d_seq, p1, (u32)ksize, seed, max_hash, cap, c.d_out, c.d_cnt, c.s_comp, done_win, upto);
            done_win = upto;
        }
    }
    u64 total = 0;
    SK_HIP(hipMemcpyAsync(&total, c.d_cnt, sizeof(u64), hipMemcpyDeviceToHost, c.s_comp));
    SK_HIP(hipStreamSynchronize(c.s_comp));
    if (rc == YH_OK) {
        *n_out = total;
        if (total > cap) {
            if (cap) { yh_set_error("hash buffer holds %llu entries, %llu needed", (u64)cap, total); rc = YH_ERR_CAPACITY; }
        } else if (total) {
            SK_HIP(hipMemcpyAsync(hashes_out, c.d_out, total * sizeof(u64), hipMemcpyDeviceToHost, c.s_comp));
            SK_HIP(hipStreamSynchronize(c.s_comp));
        }
    }
    if (rc != YH_OK) { (void)hipStreamSynchronize(c.s_copy); (void)hipStreamSynchronize(c.s_comp); }
#undef SK_HIP
    if (c.seq_cap > (1ull << 30)) { (void)hipFree(c.d_seq); c.d_seq = nullptr; c.seq_cap = 0; }
    return rc;
}
