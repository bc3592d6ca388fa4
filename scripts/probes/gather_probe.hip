// gather_probe.hip -- how fast can gfx950 answer N independent random 64-byte bucket reads (the shape of
// the sample-driven lookup), by access form?  Also PCIe H2D rates for the host-inclusive step.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef uint32_t u32;

__device__ __forceinline__ u64 mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// sorted-ish keys like a sample: key i = (i * stride + jitter); bucket = key * nb >> 64
__global__ void k_keys(u64* keys, u64 n, u64 seed) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) {
        const u64 step = (~0ull) / n;
        keys[i] = i * step + mix(i + seed) % step;
    }
}
__global__ void k_fill(uint4* t, u64 n16) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x)
        t[i] = make_uint4((u32)mix(i), (u32)mix(i + 1), (u32)mix(i + 2), (u32)i);
}
// V1: one lane per lookup, four 16-byte loads of the same 64-byte bucket
__global__ void __launch_bounds__(256) k_v1(const u64* __restrict__ keys, u64 n, const uint4* __restrict__ tab, u64 nb, u32* __restrict__ out) {
    u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 h = keys[t];
    const uint4* p = tab + 4 * __umul64hi(h, nb);
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    const u32 lo = (u32)h;
    u32 r = 0;
    if (a.x == lo) r = c.z; if (a.z == lo) r = c.w; if (b.x == lo) r = d.x; if (b.z == lo) r = d.y; if (c.x == lo) r = d.z;
    if (r) atomicAdd(&out[r & 1023], 1u);
}
// V2: one lane per lookup, ONE 16-byte load (what a 16-byte bucket would cost)
__global__ void __launch_bounds__(256) k_v2(const u64* __restrict__ keys, u64 n, const uint4* __restrict__ tab, u64 nb, u32* __restrict__ out) {
    u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 h = keys[t];
    const uint4 a = tab[4 * __umul64hi(h, nb)];
    if (a.x == (u32)h) atomicAdd(&out[a.y & 1023], 1u);
}
// V3: four lanes per lookup, 16 bytes each (a wave-instruction touches 16 buckets, fully used)
__global__ void __launch_bounds__(256) k_v3(const u64* __restrict__ keys, u64 n, const uint4* __restrict__ tab, u64 nb, u32* __restrict__ out) {
    const u64 gt = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    const u32 lane = threadIdx.x & 63u;
    const u64 wave0 = (gt & ~63ull);  // 64 lookups per wave, in four rounds of 16
    u64 hv = (wave0 + lane < n) ? keys[wave0 + lane] : 0;
    uint4 v[4];
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int src = rd * 16 + (int)(lane >> 2);
        const u64 h = __shfl(hv, src);
        v[rd] = tab[4 * __umul64hi(h, nb) + (lane & 3u)];
    }
    u32 hit = 0;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int src = rd * 16 + (int)(lane >> 2);
        const u64 h = __shfl(hv, src);
        if (wave0 + src < n && (v[rd].x == (u32)h || v[rd].z == (u32)h)) hit = v[rd].y | 1u;
    }
    if (hit) atomicAdd(&out[hit & 1023], 1u);
}
// V4: like V1 but each thread does TWO lookups (more loads in flight per wave)
__global__ void __launch_bounds__(256) k_v4(const u64* __restrict__ keys, u64 n, const uint4* __restrict__ tab, u64 nb, u32* __restrict__ out) {
    u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    const u64 half = (n + 1) / 2;
    if (t >= half) return;
    const u64 h0 = keys[t], h1 = (t + half < n) ? keys[t + half] : h0;
    const uint4* p = tab + 4 * __umul64hi(h0, nb);
    const uint4* q = tab + 4 * __umul64hi(h1, nb);
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    const uint4 a1 = q[0], b1 = q[1], c1 = q[2], d1 = q[3];
    u32 r = 0;
    if (a.x == (u32)h0 || b.x == (u32)h0 || c.x == (u32)h0) r = d.x;
    if (a1.x == (u32)h1 || b1.x == (u32)h1 || c1.x == (u32)h1) r = d1.x;
    if (r) atomicAdd(&out[r & 1023], 1u);
}
// V5: one lane per lookup, one 64-bit... two 32-byte halves as 2 x (2 x 16 B)?  -> 128-byte bucket (8 x 16 B) to see the line-size effect
__global__ void __launch_bounds__(256) k_v5(const u64* __restrict__ keys, u64 n, const uint4* __restrict__ tab, u64 nb, u32* __restrict__ out) {
    u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 h = keys[t];
    const uint4* p = tab + 8 * __umul64hi(h, nb >> 1);
    u32 r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint4 a = p[i]; if (a.x == (u32)h) r = a.y; }
    if (r) atomicAdd(&out[r & 1023], 1u);
}

// occupies every CU the way k_stream_lookup does (75 KB of LDS, 512 threads, 512 workgroups) while streaming `bytes`
__global__ void __launch_bounds__(512, 4) k_busy(const uint4* __restrict__ src, u64 n16, u32* __restrict__ out, int reps) {
    __shared__ u32 pad[75 * 256];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    u32 acc = pad[(threadIdx.x * 7) & 511];
    for (int r = 0; r < reps; ++r)
        for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
            const uint4 v = src[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    if (acc == 12345u) out[0] = acc;
}

static void overlap_probe(uint4* tab, u32* out) {
    const u64 bytes = 8ull << 20;
    void *h, *d;
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d, bytes));
    memset(h, 1, bytes);
    hipStream_t sc, sk;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const u64 n16 = (340ull << 20) / 16;  // ~340 MB streamed: ~70-90 us
    for (int mode = 0; mode < 3; ++mode) {  // 0: kernel alone, 1: copy alone, 2: both at once
        float best_k = 1e9, best_c = 1e9, best_t = 1e9;
        for (int r = 0; r < 8; ++r) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            if (mode != 1) { CK(hipEventRecord(a0, sk)); k_busy<<<512, 512, 0, sk>>>(tab, n16, out, 2); CK(hipEventRecord(a1, sk)); }
            if (mode != 0) { CK(hipEventRecord(b0, sc)); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, sc)); CK(hipEventRecord(b1, sc)); }
            CK(hipDeviceSynchronize());
            const float tot = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
            float mk = 0, mc = 0;
            if (mode != 1) CK(hipEventElapsedTime(&mk, a0, a1));
            if (mode != 0) CK(hipEventElapsedTime(&mc, b0, b1));
            best_k = std::min(best_k, mk * 1e3f); best_c = std::min(best_c, mc * 1e3f); best_t = std::min(best_t, tot);
        }
        printf("overlap probe mode %d (0 kernel, 1 copy, 2 both): kernel %.1f us, copy %.1f us, host wall %.1f us\n", mode, best_k, best_c, best_t);
    }
}

int main(int argc, char** argv) {
    const u64 table_mb = argc > 2 ? strtoull(argv[2], nullptr, 10) : 6144;  // table size in MiB
    const u64 table_bytes = table_mb << 20;
    const u64 nb = table_bytes / 64;
    uint4* tab; u32* out; u64* keys;
    CK(hipMalloc(&tab, table_bytes));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(out, 0, 4096));
    k_fill<<<4096, 256>>>(tab, table_bytes / 16);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (argc > 1 && argv[1][0] == 'x') { overlap_probe(tab, out); return 0; }
    const u64 sizes[] = {83000, 1000000, 4000000};
    const int REP = 20;
    for (u64 n : sizes) {
        CK(hipMalloc(&keys, (u64)REP * n * 8));
        for (int r = 0; r < REP; ++r) k_keys<<<(unsigned)((n + 255) / 256), 256>>>(keys + (u64)r * n, n, 1000 + r);
        CK(hipDeviceSynchronize());
        printf("n = %llu lookups, table %.1f GB:", n, table_bytes / 1e9);
        for (int v = 1; v <= 5; ++v) {
            float best = 1e9, sum = 0;
            for (int r = 0; r < REP; ++r) {
                const u64* k = keys + (u64)r * n;
                CK(hipEventRecord(e0));
                const unsigned g = (unsigned)((n + 255) / 256);
                if (v == 1) k_v1<<<g, 256>>>(k, n, tab, nb, out);
                if (v == 2) k_v2<<<g, 256>>>(k, n, tab, nb, out);
                if (v == 3) k_v3<<<g, 256>>>(k, n, tab, nb, out);
                if (v == 4) k_v4<<<(g + 1) / 2, 256>>>(k, n, tab, nb, out);
                if (v == 5) k_v5<<<g, 256>>>(k, n, tab, nb, out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) { best = std::min(best, ms); sum += ms; }
            }
            printf("  V%d %.1f us (min %.1f)", v, 1e3 * sum / (REP - 2), 1e3 * best);
        }
        printf("\n");
        if (n == 1000000) {  // the same V1 lookups with OTHER kernels between them, as inside a query step
            uint4* other; const u64 ob = 2ull << 30;
            CK(hipMalloc(&other, ob));
            k_fill<<<4096, 256>>>(other, ob / 16);
            CK(hipDeviceSynchronize());
            for (int mode = 0; mode < 3; ++mode) {  // 0: a 340 MB streaming kernel between, 1: a tiny kernel between, 2: 2 GB streamed between
                float sum = 0;
                for (int r = 0; r < REP; ++r) {
                    if (mode == 0) k_busy<<<512, 512>>>(other, (340ull << 20) / 16, out, 1);
                    if (mode == 1) k_busy<<<64, 512>>>(other, 4096, out, 1);
                    if (mode == 2) k_busy<<<512, 512>>>(other, ob / 16, out, 1);
                    CK(hipEventRecord(e0));
                    k_v1<<<(unsigned)((n + 255) / 256), 256>>>(keys + (u64)r * n, n, tab, nb, out);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) sum += ms;
                }
                printf("   V1 with another kernel in front (mode %d): %.1f us\n", mode, 1e3 * sum / (REP - 2));
            }
            CK(hipFree(other));
        }
        CK(hipFree(keys));
    }
    if (argc > 1 && argv[1][0] == 't') return 0;
    // ---- PCIe: pinned host -> device -----------------------------------------------------------------
    {
        const u64 bytes = 64ull << 20;
        void *h, *d;
        CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
        CK(hipMalloc(&d, bytes));
        memset(h, 1, bytes);
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        for (u64 sz : {1ull << 20, 8ull << 20, 64ull << 20}) {
            for (int streams = 1; streams <= 2; ++streams) {
                float best = 1e9;
                for (int r = 0; r < 6; ++r) {
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, s1));
                    if (streams == 1) CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, s1));
                    else {
                        CK(hipMemcpyAsync(d, h, sz / 2, hipMemcpyHostToDevice, s1));
                        CK(hipMemcpyAsync((char*)d + sz / 2, (char*)h + sz / 2, sz / 2, hipMemcpyHostToDevice, s2));
                        CK(hipStreamSynchronize(s2));
                    }
                    CK(hipEventRecord(e1, s1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
                printf("H2D pinned %llu MB, %d stream(s): %.1f us = %.1f GB/s\n", sz >> 20, streams, 1e3 * best, sz / (best * 1e6));
            }
        }
        // device kernel reading the pinned host buffer directly (zero-copy): a copy kernel
        // (k_fill-like read of host memory)
    }
    return 0;
}
