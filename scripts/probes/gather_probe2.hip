// gather_probe2.hip -- what bounds 1e6 isolated reads per launch (the presence-filter read of k_index_lookup_tile)?
// Sorted keys (a sample), a monotone word index into a table of T bytes (the filter), one 4-byte read per key, 8 rotating
// key sets.  Swept: table size (L2 / Infinity Cache / HBM), window inside ONE large allocation vs an allocation of its
// own (page / fragment size), workgroup -> key-range mapping (interleaved over the XCDs as dispatched, or XCD-local:
// each XCD's workgroups take one contiguous eighth of the key range), and a second dependent 64-byte read into a big
// table for a fraction of the keys (the bucket behind the filter).
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe2 gather_probe2.hip && ./gather_probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
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
__global__ void k_fill(uint4* t, u64 n16, u32 density_256) {  // words with each bit set with probability density/256 (roughly)
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
        u32 w[4];
        for (int k = 0; k < 4; ++k) {
            u32 v = 0;
            for (int b = 0; b < 32; ++b) v |= (u32)((mix(i * 128 + k * 32 + b) & 255u) < density_256) << b;
            w[k] = v;
        }
        t[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
__device__ __forceinline__ u32 xcd_remap(u32 bid, u32 nwg) {
    const u32 q = nwg >> 3, r = nwg & 7u;
    const u32 xcd = bid & 7u;
    const u32 base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
// one 4-byte read per key; U keys per lane; optionally a dependent 64-byte bucket read (4 x 16 B) for keys whose bit is set
template <int U, int THREADS, bool LOCAL, bool BUCKET, bool NT = false>
__global__ void __launch_bounds__(THREADS) k_probe(const u64* __restrict__ keys, u64 n, const u32* __restrict__ filt, u64 nbits,
                                                    const uint4* __restrict__ big, u64 nb, u32* __restrict__ out) {
    const u32 wg = LOCAL ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const u64 base = wg * (u64)(THREADS * U);
    u64 h[U]; u64 bit[U]; u32 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const u64* kp = keys + min(base + (u64)u * THREADS + threadIdx.x, n - 1);
        h[u] = NT ? __builtin_nontemporal_load(kp) : *kp;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { bit[u] = __umul64hi(h[u], nbits); w[u] = filt[bit[u] >> 5]; }
    u32 acc = 0;
    if (BUCKET) {
        uint4 a[U], b[U], c[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = b[u] = c[u] = d[u] = make_uint4(0, 0, 0, 0);
            if ((w[u] >> (bit[u] & 31u)) & 1u) {
                typedef u32 v4u __attribute__((ext_vector_type(4)));
                const v4u* p = reinterpret_cast<const v4u*>(big) + 4 * __umul64hi(h[u], nb);
                v4u x0, x1, x2, x3;
                if (NT) { x0 = __builtin_nontemporal_load(p); x1 = __builtin_nontemporal_load(p + 1); x2 = __builtin_nontemporal_load(p + 2); x3 = __builtin_nontemporal_load(p + 3); }
                else { x0 = p[0]; x1 = p[1]; x2 = p[2]; x3 = p[3]; }
                a[u] = make_uint4(x0.x, x0.y, x0.z, x0.w); b[u] = make_uint4(x1.x, x1.y, x1.z, x1.w);
                c[u] = make_uint4(x2.x, x2.y, x2.z, x2.w); d[u] = make_uint4(x3.x, x3.y, x3.z, x3.w);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += a[u].x ^ b[u].y ^ c[u].z ^ d[u].w;
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= w[u] >> (bit[u] & 31u);  // (not provably bounded: the loads stay)
    }
    if (acc == 0x12345678u) atomicAdd(out, acc);
}

int main(int argc, char** argv) {
    const u64 big_bytes = 12ull << 30;
    uint4* big; u32* out; u64* keys;
    CK(hipMalloc(&big, big_bytes));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(out, 0, 4096));
    k_fill<<<8192, 256>>>(big, big_bytes / 16, 64);  // every bit set with probability 1/4
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u64 n = 1000000;
    const int REP = 8, ROUNDS = 6;
    CK(hipMalloc(&keys, (u64)REP * n * 8));
    for (int r = 0; r < REP; ++r) k_keys<<<(unsigned)((n + 255) / 256), 256>>>(keys + (u64)r * n, n, 1000 + r);
    CK(hipDeviceSynchronize());
    auto time_it = [&](auto launch) {
        float sum = 0; int cnt = 0;
        for (int round = 0; round < ROUNDS; ++round)
            for (int r = 0; r < REP; ++r) {
                CK(hipEventRecord(e0));
                launch(keys + (u64)r * n);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (round >= 1) { sum += ms; ++cnt; }
            }
        return 1e3f * sum / cnt;
    };
    const u64 MB = 1ull << 20;
    printf("1e6 sorted keys, ONE 4-byte read each, table = window of a 12 GB allocation; us per launch (HIP events, includes ~5 us of event overhead)\n");
    printf("%10s %12s %12s %12s %12s\n", "table", "256x1 inter", "256x1 local", "1024x2 inter", "1024x2 local");
    for (u64 mb : {1ull, 4ull, 16ull, 32ull, 64ull, 128ull, 256ull, 512ull, 2048ull, 8192ull}) {
        const u64 nbits = mb * MB * 8;
        const u32* f = reinterpret_cast<const u32*>(big);
        const float a = time_it([&](const u64* k) { k_probe<1, 256, false, false><<<(unsigned)((n + 255) / 256), 256>>>(k, n, f, nbits, big, 0, out); });
        const float b = time_it([&](const u64* k) { k_probe<1, 256, true, false><<<(unsigned)((n + 255) / 256), 256>>>(k, n, f, nbits, big, 0, out); });
        const float c = time_it([&](const u64* k) { k_probe<2, 1024, false, false><<<(unsigned)((n + 2047) / 2048), 1024>>>(k, n, f, nbits, big, 0, out); });
        const float d = time_it([&](const u64* k) { k_probe<2, 1024, true, false><<<(unsigned)((n + 2047) / 2048), 1024>>>(k, n, f, nbits, big, 0, out); });
        printf("%8llu MB %12.1f %12.1f %12.1f %12.1f\n", mb, a, b, c, d);
    }
    printf("the same with the table in an allocation of its own\n");
    for (u64 mb : {16ull, 64ull, 128ull, 256ull}) {
        uint4* own; CK(hipMalloc(&own, mb * MB));
        k_fill<<<4096, 256>>>(own, mb * MB / 16, 64);
        CK(hipDeviceSynchronize());
        const u32* f = reinterpret_cast<const u32*>(own);
        const u64 nbits = mb * MB * 8;
        const float a = time_it([&](const u64* k) { k_probe<1, 256, false, false><<<(unsigned)((n + 255) / 256), 256>>>(k, n, f, nbits, big, 0, out); });
        const float d = time_it([&](const u64* k) { k_probe<2, 1024, true, false><<<(unsigned)((n + 2047) / 2048), 1024>>>(k, n, f, nbits, big, 0, out); });
        printf("%8llu MB %12.1f %38.1f\n", mb, a, d);
        CK(hipFree(own));
    }
    printf("filter window (bits set with probability 1/4) + a dependent 64-byte bucket read into an 8 GB table behind it for keys whose bit is set\n");
    printf("%10s %12s %12s %12s %12s\n", "filter", "256x1 inter", "256x1 local", "1024x2 inter", "1024x2 local");
    for (u64 mb : {4ull, 32ull, 64ull, 160ull, 1024ull}) {
        const u64 nbits = mb * MB * 8;
        const u32* f = reinterpret_cast<const u32*>(big) + (10ull << 30) / 4;  // the filter window lies behind the bucket table
        const u64 nb = (8ull << 30) / 64;
        const float a = time_it([&](const u64* k) { k_probe<1, 256, false, true><<<(unsigned)((n + 255) / 256), 256>>>(k, n, f, nbits, big, nb, out); });
        const float b = time_it([&](const u64* k) { k_probe<1, 256, true, true><<<(unsigned)((n + 255) / 256), 256>>>(k, n, f, nbits, big, nb, out); });
        const float c = time_it([&](const u64* k) { k_probe<2, 1024, false, true><<<(unsigned)((n + 2047) / 2048), 1024>>>(k, n, f, nbits, big, nb, out); });
        const float d = time_it([&](const u64* k) { k_probe<2, 1024, true, true><<<(unsigned)((n + 2047) / 2048), 1024>>>(k, n, f, nbits, big, nb, out); });
        const float e = time_it([&](const u64* k) { k_probe<1, 256, false, true, true><<<(unsigned)((n + 255) / 256), 256>>>(k, n, f, nbits, big, nb, out); });
        const float g = time_it([&](const u64* k) { k_probe<2, 1024, false, true, true><<<(unsigned)((n + 2047) / 2048), 1024>>>(k, n, f, nbits, big, nb, out); });
        printf("%8llu MB %12.1f %12.1f %12.1f %12.1f   nt keys+buckets: 256x1 %.1f  1024x2 %.1f\n", mb, a, b, c, d, e, g);
    }
    // empty launch + event pair for reference
    {
        const float z = time_it([&](const u64* k) { k_probe<1, 256, false, false><<<1, 256>>>(k, 256, reinterpret_cast<const u32*>(big), 1024, big, 0, out); });
        printf("one-workgroup launch: %.1f us\n", z);
    }
    return 0;
}
