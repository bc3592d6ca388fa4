// malloc_modes_probe.hip -- the first tens of gigabytes a fresh process asks the driver for (VERDICT r04 "next" 7): on some boxes
// of this pool a process pays ~1-2 s ONCE inside hipMalloc around there (scripts/probes/fresh_build_probe.py).  Which call pays,
// and does another way of asking avoid it?  One mode per process (argv[1]):
//   many     48 hipMalloc calls of 0.1-3 GB (40 GB together: what yh_db_create asks for at rs214 scale), each timed
//   one      ONE hipMalloc of 40 GB
//   async    the same 48 sizes through hipMallocAsync on a stream (the device's default memory pool)
//   vmm      hipMemAddressReserve(40 GB) + hipMemCreate / hipMemMap / hipMemSetAccess in 1 GB pieces
// then (all modes) a hipMemset of the first 4 GB and a second round of the same requests after freeing everything.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/malloc_modes_probe.hip -o /tmp/mmp && for m in many one async vmm; do /tmp/mmp $m; done
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <vector>
static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "many";
    const double t_start = now_ms();
    hipSetDevice(0);
    hipFree(0);
    const double t_ctx = now_ms();
    std::vector<size_t> sizes;
    size_t total = 0;
    for (int i = 0; i < 48; ++i) { size_t s = ((size_t)(100 + (i * 977) % 2900)) << 20; if (total + s > ((size_t)40 << 30)) s = ((size_t)40 << 30) - total; if (!s) break; sizes.push_back(s); total += s; }
    hipStream_t st; hipStreamCreate(&st);
    for (int round = 0; round < 2; ++round) {
        std::vector<void*> ptrs;
        double worst = 0, sum = 0; int worst_i = -1;
        const double t0 = now_ms();
        if (!strcmp(mode, "many") || !strcmp(mode, "async")) {
            for (size_t i = 0; i < sizes.size(); ++i) {
                void* p = nullptr;
                const double a = now_ms();
                hipError_t e = !strcmp(mode, "many") ? hipMalloc(&p, sizes[i]) : hipMallocAsync(&p, sizes[i], st);
                if (!strcmp(mode, "async")) hipStreamSynchronize(st);
                const double d = now_ms() - a;
                if (e != hipSuccess) { printf("alloc %zu failed: %s\n", i, hipGetErrorString(e)); return 1; }
                ptrs.push_back(p); sum += d;
                if (d > worst) { worst = d; worst_i = (int)i; }
            }
        } else if (!strcmp(mode, "one")) {
            void* p = nullptr;
            const double a = now_ms();
            if (hipMalloc(&p, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
            worst = sum = now_ms() - a; worst_i = 0; ptrs.push_back(p);
        } else {  // vmm
            void* base = nullptr;
            const size_t piece = (size_t)1 << 30;
            const double a = now_ms();
            if (hipMemAddressReserve(&base, total, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            for (size_t off = 0; off < total; off += piece) {
                const size_t n = total - off < piece ? total - off : piece;
                hipMemGenericAllocationHandle_t h;
                const double b = now_ms();
                if (hipMemCreate(&h, n, &prop, 0) != hipSuccess || hipMemMap((char*)base + off, n, 0, h, 0) != hipSuccess ||
                    hipMemSetAccess((char*)base + off, n, &acc, 1) != hipSuccess) { printf("vmm piece at %zu failed: %s\n", off, hipGetErrorString(hipGetLastError())); return 1; }
                const double d = now_ms() - b;
                if (d > worst) { worst = d; worst_i = (int)(off / piece); }
            }
            sum = now_ms() - a; ptrs.push_back(base);
        }
        const double t1 = now_ms();
        hipMemset(ptrs[0], 0, (size_t)1 << 30);
        hipDeviceSynchronize();
        const double t2 = now_ms();
        printf("mode %-5s round %d: context %.0f ms | %zu requests, %.1f GB: %.1f ms inside the allocation calls (worst: request %d, %.1f ms) | first memset of 1 GB %.1f ms | since process start %.0f ms\n",
               mode, round, t_ctx - t_start, sizes.size(), total / 1073741824.0, sum, worst_i, worst, t2 - t1, t2 - t_start);
        (void)t0;
        if (!strcmp(mode, "vmm")) { /* (left mapped: the second round would need unmap + release; skip it) */ break; }
        for (void* p : ptrs) { if (!strcmp(mode, "async")) hipFreeAsync(p, st); else hipFree(p); }
        hipDeviceSynchronize();
    }
    return 0;
}
