// scatter_probe.hip -- what does the TRANSPOSITION of `yacht train` cost on this chip, and can the L2s merge it?
//
// The pairwise pass wants, for every reference, the records of its shared hashes side by side (reference-major); the sort
// leaves them hash-major.  Rounds 2-4 moved them with one isolated 16-byte store per posting (k_pair_transpose) behind
// one counting atomic per posting (k_idx_emit): 27 M requests each at configs[3], 0.56 + 0.76 ms -- both at the ~4.6e10
// isolated requests/s the lookup probes found.  A store per posting is not a law, though: the elements of ONE sketch that
// fall into ONE first-level region of the sort (1/140 of the hash space) are ~36 consecutive positions of its CSR, and all
// of a region's buckets are sorted within a short time of each other.  If every bucket of a region runs on the SAME XCD,
// that XCD's L2 sees all 36 stores of the run and can leave them as full lines.
// This probe replays exactly that pattern -- 10 000 sketches x 5 000 hashes, 19 536 buckets of ~2 560 pairs, a record
// stored at the element's CSR position -- under several bucket -> workgroup maps:
//   random      positions permuted over the whole array (every store isolated)
//   identity    bucket = blockIdx (consecutive buckets on different XCDs)
//   xcd         XCD x walks the x-th eighth of the hash space front to back (blockIdx % 8 = XCD on this chip)
// each with 8- and 16-byte records, all elements or the 54 % that are shared at configs[3]; plus the counting atomic, and
// the same stores NON-TEMPORAL (past the L2s: 1.3 ms instead of 0.29 -- the merging is the L2s' write-back caching).
//   hipcc --offload-arch=gfx950 -O3 -o scatter_probe scatter_probe.hip && ./scatter_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef uint32_t u32;

__host__ __device__ __forceinline__ u64 mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr u32 NSK = 10000, S = 5000;
constexpr u64 H = (u64)NSK * S;
constexpr u32 NB = 19536;  // multiple of 8

__device__ __forceinline__ u64 hash_of(u32 a, u32 e) {
    const u64 step = (~0ull) / S;
    return (u64)e * step + mix(((u64)a << 32) | e) % step;  // ascending in e: a sorted sketch of uniform hashes
}
__global__ void k_count(u32* cnt) {
    const u64 p = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (p >= H) return;
    const u32 b = (u32)__umul64hi(hash_of((u32)(p / S), (u32)(p % S)), NB);
    atomicAdd(&cnt[b], 1u);
}
__global__ void k_fill(const u64* off, u32* cur, u32* lst) {
    const u64 p = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (p >= H) return;
    const u32 b = (u32)__umul64hi(hash_of((u32)(p / S), (u32)(p % S)), NB);
    lst[off[b] + atomicAdd(&cur[b], 1u)] = (u32)p;
}
// MAP 0 identity, 1 xcd-local, 2 identity with globally permuted positions
// NT: the record leaves with a non-temporal store (does a different cache policy change what the L2s merge?)
template <int MAP>
__global__ void __launch_bounds__(1024) k_scatter_nt(const u64* __restrict__ off, const u32* __restrict__ lst, u64* __restrict__ rec, u32 keep_of_128) {
    u32 b = blockIdx.x;
    if (MAP == 1) b = (blockIdx.x & 7u) * (NB / 8) + (blockIdx.x >> 3);
    const u64 o0 = off[b], o1 = off[b + 1];
    for (u64 i = o0 + threadIdx.x; i < o1; i += 1024) {
        const u64 p = lst[i];
        if ((mix(p) & 127u) >= keep_of_128) continue;
        __builtin_nontemporal_store((unsigned long long)(p * 0x0101010101010101ull), (unsigned long long*)&rec[p]);
    }
}
template <typename REC, int MAP>
__global__ void __launch_bounds__(1024) k_scatter(const u64* __restrict__ off, const u32* __restrict__ lst, REC* __restrict__ rec, u32 keep_of_128) {
    u32 b = blockIdx.x;
    if (MAP == 1) b = (blockIdx.x & 7u) * (NB / 8) + (blockIdx.x >> 3);
    const u64 o0 = off[b], o1 = off[b + 1];
    for (u64 i = o0 + threadIdx.x; i < o1; i += 1024) {
        u64 p = lst[i];
        if ((mix(p) & 127u) >= keep_of_128) continue;
        if (MAP == 2) p = (p * 2654435761ull + 12345) % H;
        REC r;
        unsigned char* q = reinterpret_cast<unsigned char*>(&r);
        for (unsigned k = 0; k < sizeof(REC); ++k) q[k] = (unsigned char)(p >> (k & 3));
        rec[p] = r;
    }
}
template <int MAP>
__global__ void __launch_bounds__(1024) k_atomic(const u64* __restrict__ off, const u32* __restrict__ lst, u32* __restrict__ nsh, u32* __restrict__ sink, u32 keep_of_128) {
    u32 b = blockIdx.x;
    if (MAP == 1) b = (blockIdx.x & 7u) * (NB / 8) + (blockIdx.x >> 3);
    const u64 o0 = off[b], o1 = off[b + 1];
    u32 acc = 0;
    for (u64 i = o0 + threadIdx.x; i < o1; i += 1024) {
        const u64 p = lst[i];
        if ((mix(p) & 127u) >= keep_of_128) continue;
        acc += atomicAdd(&nsh[p / S], 1u);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// the row pass's side: read the records of every sketch front to back (a workgroup per sketch)
template <typename REC>
__global__ void __launch_bounds__(512) k_read(const REC* __restrict__ rec, u32* __restrict__ sink) {
    const u64 base = (u64)blockIdx.x * S;
    u32 acc = 0;
    for (u32 i = threadIdx.x; i < S; i += 512) {
        const REC r = rec[base + i];
        acc += *reinterpret_cast<const u32*>(&r);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F>
static float timed(F f, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    u32 *cnt, *cur, *lst, *nsh, *sink;
    u64* off;
    void* rec;
    CK(hipMalloc(&cnt, NB * 4)); CK(hipMalloc(&cur, NB * 4)); CK(hipMalloc(&lst, H * 4)); CK(hipMalloc(&off, (NB + 1) * 8));
    CK(hipMalloc(&nsh, NSK * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&rec, H * 16));
    CK(hipMemset(cnt, 0, NB * 4)); CK(hipMemset(cur, 0, NB * 4)); CK(hipMemset(nsh, 0, NSK * 4));
    k_count<<<(u32)((H + 255) / 256), 256>>>(cnt);
    std::vector<u32> hc(NB);
    CK(hipMemcpy(hc.data(), cnt, NB * 4, hipMemcpyDeviceToHost));
    std::vector<u64> ho(NB + 1);
    u64 acc = 0; u32 mx = 0;
    for (u32 b = 0; b < NB; ++b) { ho[b] = acc; acc += hc[b]; if (hc[b] > mx) mx = hc[b]; }
    ho[NB] = acc;
    CK(hipMemcpy(off, ho.data(), (NB + 1) * 8, hipMemcpyHostToDevice));
    k_fill<<<(u32)((H + 255) / 256), 256>>>(off, cur, lst);
    CK(hipDeviceSynchronize());
    printf("%llu pairs, %u buckets (largest %u), %u sketches x %u\n", acc, NB, mx, NSK, S);
    const u32 all = 128, part = 69;  // 54 % of the elements are shared at configs[3]
#define RUN(name, REC, MAP, keep)                                                                                     \
    {                                                                                                                 \
        const float us = timed([&] { k_scatter<REC, MAP><<<NB, 1024>>>(off, lst, (REC*)rec, keep); });                 \
        const double n = (double)H * keep / 128.0;                                                                    \
        printf("%-44s %8.1f us   %6.2f e10 stores/s   %7.1f GB/s of records\n", name, us, n / us / 1e4, n * sizeof(REC) / us / 1e3); \
    }
    RUN("8 B  all elements   random positions", u64, 2, all);
    RUN("8 B  all elements   bucket = blockIdx", u64, 0, all);
    RUN("8 B  all elements   XCD-local regions", u64, 1, all);
    RUN("8 B  54 %           random positions", u64, 2, part);
    RUN("8 B  54 %           bucket = blockIdx", u64, 0, part);
    RUN("8 B  54 %           XCD-local regions", u64, 1, part);
    {
        const float us0 = timed([&] { k_scatter_nt<0><<<NB, 1024>>>(off, lst, (u64*)rec, part); });
        printf("%-44s %8.1f us\n", "8 B  54 %  non-temporal  bucket = blockIdx", us0);
        const float us1 = timed([&] { k_scatter_nt<1><<<NB, 1024>>>(off, lst, (u64*)rec, part); });
        printf("%-44s %8.1f us\n", "8 B  54 %  non-temporal  XCD-local regions", us1);
    }
    RUN("16 B all elements   random positions", uint4, 2, all);
    RUN("16 B all elements   bucket = blockIdx", uint4, 0, all);
    RUN("16 B all elements   XCD-local regions", uint4, 1, all);
    RUN("16 B 54 %           random positions", uint4, 2, part);
    RUN("16 B 54 %           bucket = blockIdx", uint4, 0, part);
    RUN("16 B 54 %           XCD-local regions", uint4, 1, part);
    {
        const float us = timed([&] { k_atomic<0><<<NB, 1024>>>(off, lst, nsh, sink, part); });
        printf("%-44s %8.1f us   %6.2f e10 atomics/s\n", "counting atomic (returning), 54 %, blockIdx", us, (double)H * part / 128.0 / us / 1e4);
        const float us2 = timed([&] { k_atomic<1><<<NB, 1024>>>(off, lst, nsh, sink, part); });
        printf("%-44s %8.1f us   %6.2f e10 atomics/s\n", "counting atomic (returning), 54 %, XCD-local", us2, (double)H * part / 128.0 / us2 / 1e4);
    }
    {
        const float us = timed([&] { CK(hipMemsetAsync(rec, 0, H * 8, 0)); });
        printf("%-44s %8.1f us\n", "memset of 8 B x all elements", us);
        const float us8 = timed([&] { k_read<u64><<<NSK, 512>>>((const u64*)rec, sink); });
        printf("%-44s %8.1f us   %7.1f GB/s\n", "row-order read, 8 B records", us8, (double)H * 8 / us8 / 1e3);
        const float us16 = timed([&] { k_read<uint4><<<NSK, 512>>>((const uint4*)rec, sink); });
        printf("%-44s %8.1f us   %7.1f GB/s\n", "row-order read, 16 B records", us16, (double)H * 16 / us16 / 1e3);
    }
    return 0;
}
