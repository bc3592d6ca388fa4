// segread_probe.hip -- why does k_pair_rows read its 16-byte records at 1.5 TB/s when a grid-stride kernel reads the same
// buffer at 5 TB/s?  27.2 M records (435 MB) in 10 000 segments of ~2 716 (a reference's shared postings at configs[3]).
//   linear      grid-stride over the whole buffer
//   seg         one workgroup per segment (offsets from a table), THREADS lanes, U loads in flight per lane
//   seg+lds     ... with L bytes of dynamic LDS claimed (4 workgroups per CU at 40 KB)
//   seg+clear   ... and the LDS row cleared and scanned as k_pair_rows does
//   multi       a workgroup takes R consecutive segments (fewer, longer-lived workgroups)
//   hipcc --offload-arch=gfx950 -O3 -o segread_probe segread_probe.hip && ./segread_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef uint32_t u32;

__global__ void k_fill(uint4* t, u64 n) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        t[i] = make_uint4((u32)i, (u32)(i >> 3), (u32)(i * 7), 2u);
}
__global__ void __launch_bounds__(256) k_linear(const uint4* __restrict__ r, u64 n, u32* out) {
    u32 acc = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) acc ^= r[i].w ^ r[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}
template <int THREADS, int U, int MODE>  // MODE 0: read only; 1: + clear and scan `cols` words of LDS; 2: + LDS atomics
__global__ void __launch_bounds__(THREADS) k_seg(const uint4* __restrict__ r, const u32* __restrict__ ptr, u32 nseg, u32 per_wg, u32 cols,
                                                 u32* out) {
    extern __shared__ u32 row[];
    u32 acc = 0;
    for (u32 s = blockIdx.x * per_wg; s < min(nseg, (blockIdx.x + 1) * per_wg); ++s) {
        const u32 t0 = ptr[s], t1 = ptr[s + 1];
        if (MODE >= 1) {
            for (u32 j = threadIdx.x; j < cols; j += THREADS) row[j] = 0;
            __syncthreads();
        }
        for (u32 tb = t0; tb < t1; tb += U * THREADS) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32 t = tb + u * THREADS + threadIdx.x;
                v[u] = make_uint4(0, 0, 0, 0);
                if (t < t1) v[u] = r[t];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc ^= v[u].w ^ v[u].x;
                if (MODE >= 2 && v[u].w) { atomicAdd(&row[(s * 5 + (v[u].x & 3u)) % cols], 1u); atomicAdd(&row[(s * 5 + (v[u].y & 3u)) % cols], 1u); }
            }
        }
        if (MODE >= 1) {
            __syncthreads();
            for (u32 j = threadIdx.x; j < cols; j += THREADS) acc += (u32)__popcll(__ballot(row[j] != 0));
            __syncthreads();
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const u32 nseg = 10000;
    std::vector<u32> ptr(nseg + 1);
    u64 acc = 0;
    srand(5);
    for (u32 s = 0; s < nseg; ++s) { ptr[s] = (u32)acc; acc += 2716 - 50 + rand() % 100; }
    ptr[nseg] = (u32)acc;
    const u64 n = acc;
    uint4* d_r; u32 *d_ptr, *d_out;
    CK(hipMalloc(&d_r, n * 16)); CK(hipMalloc(&d_ptr, (nseg + 1) * 4)); CK(hipMalloc(&d_out, 64));
    CK(hipMemcpy(d_ptr, ptr.data(), (nseg + 1) * 4, hipMemcpyHostToDevice));
    k_fill<<<4096, 256>>>(d_r, n);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-46s %8.1f us  %6.2f TB/s\n", name, best * 1e3, n * 16 / (best * 1e-3) / 1e12);
    };
    printf("%llu records, %.0f MB, %u segments\n", n, n * 16 / 1e6, nseg);
    time("linear 8192 x 256", [&] { k_linear<<<8192, 256>>>(d_r, n, d_out); });
#define SEG(T, U, M, LDS, PER, label) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg<T, U, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 150000)); \
    time(label, [&] { k_seg<T, U, M><<<(nseg + PER - 1) / PER, T, LDS>>>(d_r, d_ptr, nseg, PER, 10000, d_out); }); }
    SEG(256, 1, 0, 0, 1, "seg 256 thr U1, no LDS");
    SEG(256, 4, 0, 0, 1, "seg 256 thr U4, no LDS");
    SEG(512, 4, 0, 0, 1, "seg 512 thr U4, no LDS");
    SEG(1024, 2, 0, 0, 1, "seg 1024 thr U2, no LDS");
    SEG(256, 4, 0, 40000, 1, "seg 256 thr U4, 40 KB LDS claimed");
    SEG(512, 4, 0, 40000, 1, "seg 512 thr U4, 40 KB LDS claimed");
    SEG(512, 4, 0, 20000, 1, "seg 512 thr U4, 20 KB LDS claimed");
    SEG(512, 4, 1, 40000, 1, "seg 512 thr U4, 40 KB LDS clear+scan");
    SEG(512, 4, 2, 40000, 1, "seg 512 thr U4, 40 KB LDS clear+scan+atomics");
    SEG(1024, 2, 2, 40000, 1, "seg 1024 thr U2, 40 KB LDS clear+scan+atomics");
    SEG(512, 4, 0, 40000, 4, "multi 4 seg/WG 512 thr U4, 40 KB claimed");
    SEG(512, 4, 2, 40000, 4, "multi 4 seg/WG 512 thr U4, clear+scan+atomics");
    SEG(512, 4, 2, 40000, 10, "multi 10 seg/WG 512 thr U4, clear+scan+atomics");
    SEG(1024, 2, 2, 40000, 10, "multi 10 seg/WG 1024 thr U2, clear+scan+atomics");
    SEG(1024, 2, 2, 40000, 20, "multi 20 seg/WG 1024 thr U2, clear+scan+atomics");
    return 0;
}
