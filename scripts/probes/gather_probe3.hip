// gather_probe3.hip -- the floor of a TWO-LEVEL presence structure in front of the bucket table.
//
// The lookup (k_step_fused / k_index_lookup_tile) pays one isolated read of a 162 MB presence bitmap per sample hash
// (4 bits per distinct hash: a clear bit proves absence) and one dependent 64-byte bucket read for the hashes that are
// there (13 % of a bench sample) or find a set bit (22 % of the absent ones).  Would a small first level -- resident in the
// L2s / the Infinity Cache -- that rejects absent hashes before the bitmap is read pay?  What such a level CAN reject is
// bounded by information theory, not by engineering: a structure of b bits per stored key answers "absent" for at most
// 1 - 2^-b of the absent queries (a filter with false-positive rate e needs log2(1/e) bits per key).  The database has
// 3.2e8 distinct hashes:   8 MB = 0.21 bits/key -> rejects <= 14 %;  16 MB: <= 25 %;  32 MB: <= 44 %;  64 MB: <= 69 %.
// This probe measures the access pattern itself -- 1e6 sorted keys per launch, 8 rotating key sets --
//   level 1   every key reads 4 bytes of a table of L1 MB (monotone index, as the bitmap is read)
//   level 2   the keys level 1 lets through (present + false positives: fraction `pass1`) read 4 bytes of the 162 MB bitmap
//   bucket    the keys level 2 lets through (fraction `pass2` of all keys) read a 64-byte bucket of an 8 GB table
// with the pass fractions set to what the bound allows (`bound`) and to what a one-hash bitmap of that size gives
// (`bitmap`: false-positive rate 1 - exp(-keys / bits)), next to today's single level and to the configuration the
// verdict asks about (1e6 first-level reads from <= 32 MB + 0.3e6 dependent 64-byte reads).
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe3 gather_probe3.hip && ./gather_probe3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef uint32_t u32;

__device__ __forceinline__ u64 mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_keys(u64* keys, u64 n, u64 seed) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) {
        const u64 step = (~0ull) / n;
        keys[i] = i * step + mix(i + seed) % step;
    }
}
__global__ void k_fill(uint4* t, u64 n16) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
        const u64 a = mix(2 * i), b = mix(2 * i + 1);
        t[i] = make_uint4((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32));
    }
}
// 2 keys per lane, 1024 lanes (the shape of k_step_fused<2, 1024, 10>'s lookup role).  Who passes a level is decided by a
// hash of the key against a threshold (the table's content only keeps the loads alive), so the fractions are exact.
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(1024) k_probe(const u64* __restrict__ keys, u64 n, const u32* __restrict__ l1, u64 l1_words,
                                                 const u32* __restrict__ l2, u64 l2_words, const uint4* __restrict__ big, u64 nb,
                                                 u32 thr1, u32 thr2, u32* __restrict__ out) {
    constexpr int U = 2;
    const u64 base = blockIdx.x * (u64)(1024 * U);
    u64 h[U];
    u32 w1[U], w2[U], r[U];
    bool p1[U], p2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) h[u] = keys[min(base + (u64)u * 1024 + threadIdx.x, n - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        r[u] = (u32)(mix(h[u]) >> 32);
        w1[u] = TWO_LEVEL ? l1[__umul64hi(h[u], l1_words)] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        p1[u] = !TWO_LEVEL || (r[u] + (w1[u] & 1u) * 0u) < thr1 || thr1 == 0xffffffffu;  // (w1 stays live below)
        w2[u] = p1[u] ? l2[__umul64hi(h[u], l2_words)] : 0u;
    }
    u32 acc = 0;
    uint4 a[U], b[U], c[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        p2[u] = p1[u] && ((r[u] < thr2) || thr2 == 0xffffffffu);
        a[u] = b[u] = c[u] = d[u] = make_uint4(w1[u], w2[u], 0, 0);
        if (p2[u]) {
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            const v4u* p = reinterpret_cast<const v4u*>(big) + 4 * __umul64hi(h[u], nb);
            const v4u x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
            a[u] = make_uint4(x0.x, x0.y, x0.z, x0.w); b[u] = make_uint4(x1.x, x1.y, x1.z, x1.w);
            c[u] = make_uint4(x2.x, x2.y, x2.z, x2.w); d[u] = make_uint4(x3.x, x3.y, x3.z, x3.w);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += a[u].x ^ b[u].y ^ c[u].z ^ d[u].w ^ w1[u] ^ w2[u];
    if (acc == 0x12345678u) atomicAdd(out, acc);
}

int main() {
    const u64 MB = 1ull << 20;
    const u64 big_bytes = 8ull << 30;
    uint4 *big, *filt, *l1t;
    u32* out;
    u64* keys;
    CK(hipMalloc(&big, big_bytes));
    CK(hipMalloc(&filt, 162 * MB));
    CK(hipMalloc(&l1t, 64 * MB));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(out, 0, 4096));
    k_fill<<<8192, 256>>>(big, big_bytes / 16);
    k_fill<<<8192, 256>>>(filt, 162 * MB / 16);
    k_fill<<<8192, 256>>>(l1t, 64 * MB / 16);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const u64 n = 1000000;
    const int REP = 8, ROUNDS = 6;
    CK(hipMalloc(&keys, (u64)REP * n * 8));
    for (int r = 0; r < REP; ++r) k_keys<<<(unsigned)((n + 255) / 256), 256>>>(keys + (u64)r * n, n, 1000 + r);
    CK(hipDeviceSynchronize());
    auto time_it = [&](auto launch) {
        float sum = 0;
        int cnt = 0;
        for (int round = 0; round < ROUNDS; ++round)
            for (int r = 0; r < REP; ++r) {
                CK(hipEventRecord(e0));
                launch(keys + (u64)r * n);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (round >= 1) { sum += ms; ++cnt; }
            }
        return 1e3f * sum / cnt;
    };
    auto thr = [](double frac) { return frac >= 1.0 ? 0xffffffffu : (u32)(frac * 4294967296.0); };
    const u64 nb = big_bytes / 64;
    const unsigned grid = (unsigned)((n + 2047) / 2048);
    const double D = 3.2e8, present = 0.13, fp2 = 0.22;  // distinct hashes; bench sample: 13 % of its hashes are in the database; the 4-bit bitmap lets 22 % of the absent through
    printf("1e6 sorted keys per launch, 1024 x 2 lanes; us per launch (HIP events: ~6 us of that is the empty launch)\n");
    {
        const double pass2 = present + (1 - present) * fp2;
        const float t = time_it([&](const u64* k) { k_probe<false><<<grid, 1024>>>(k, n, nullptr, 0, (const u32*)filt, 162 * MB / 4, big, nb, 0xffffffffu, thr(pass2), out); });
        printf("today: 162 MB bitmap for every key, bucket for %.0f %% of them                       %6.1f us\n", 100 * pass2, t);
        const float t0 = time_it([&](const u64* k) { k_probe<false><<<grid, 1024>>>(k, n, nullptr, 0, (const u32*)filt, 162 * MB / 4, big, nb, 0xffffffffu, thr(present), out); });
        printf("a PERFECT filter of that size (bucket only for the %.0f %% that are there)             %6.1f us\n", 100 * present, t0);
    }
    printf("%8s %9s | %28s | %28s\n", "level 1", "bits/key", "bound: pass1 pass2   us", "one-hash bitmap: pass1 pass2   us");
    for (u64 mb : {4ull, 8ull, 16ull, 32ull, 64ull}) {
        const double bits = (double)(mb * MB * 8) / D;
        const double fp_bound = pow(2.0, -bits), fp_bitmap = 1.0 - exp(-1.0 / bits);
        float t[2];
        double p1[2], p2[2];
        int q = 0;
        for (double fp1 : {fp_bound, fp_bitmap}) {
            p1[q] = present + (1 - present) * fp1;
            p2[q] = present + (1 - present) * fp1 * fp2;
            const u32 t1 = thr(p1[q]), t2 = thr(p2[q]);
            t[q] = time_it([&](const u64* k) { k_probe<true><<<grid, 1024>>>(k, n, (const u32*)l1t, mb * MB / 4, (const u32*)filt, 162 * MB / 4, big, nb, t1, t2, out); });
            ++q;
        }
        printf("%5llu MB %9.2f | %14.2f %5.2f %7.1f | %20.2f %5.2f %7.1f\n", mb, bits, p1[0], p2[0], t[0], p1[1], p2[1], t[1]);
    }
    printf("the configuration of VERDICT r03 item 4 -- every key reads a first level of <= 32 MB, 0.3e6 dependent 64-byte reads, no second level --\n"
           "which no structure of that size can deliver for 3.2e8 keys (32 MB = 0.84 bits per key lets >= 56 %% of the absent hashes through):\n");
    for (u64 mb : {16ull, 32ull}) {
        const float t = time_it([&](const u64* k) { k_probe<false><<<grid, 1024>>>(k, n, nullptr, 0, (const u32*)l1t, mb * MB / 4, big, nb, 0xffffffffu, thr(0.30), out); });
        printf("%5llu MB first level + bucket for 30 %% of the keys                                   %6.1f us\n", mb, t);
    }
    return 0;
}
